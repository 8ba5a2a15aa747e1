"""GPU box: per-wave record of one wf_travq launch (RT_DEBUG_TRAV=<launch index>).  Needs a -DRT_DEBUG build of the library
(hipcc ... -DRT_DEBUG -o gpurun_out/dbg.so raytracinggpu_amd/csrc/rt_capi.hip; run with RT_LIB=gpurun_out/dbg.so): the product build has no debug records."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
p = rt.make_params(W, H, 1, 3, variant="wavefront_queue", **rt.scenes.CPU_LAUNCHER)
rows, idx = rt.interleaved_rows(H, 8, 0, 1)
local = tiling.local_buffer(H, W, 1, "cuda:0")
for _ in range(3):
    ctx.render_device(p, rows, local.data_ptr()); ctx.synchronize()
raw = np.fromfile("gpurun_out/trav_dbg.bin", dtype=np.uint64)
st = ctx.stats()
nw = st["grid_blocks"] * (st["block_threads"] // 64)
a = raw[:16 * nw].reshape(-1, 16).astype(np.int64)
t0, t1 = a[:, 0], a[:, 1]
base = t0.min(); dur = (t1 - t0) / 100.0
print("launch", os.environ["RT_DEBUG_TRAV"], "waves", nw, "blocks", st["grid_blocks"], "kernel span us %.1f" % ((t1.max() - base) / 100.0))
print("wave duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f ; start offset p50 %.1f max %.1f" % ((dur.mean(),) + tuple(np.percentile(dur, [10, 50, 90, 100])) + tuple(np.percentile((t0 - base) / 100.0, [50, 100]))))
print("end time us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((t1 - base) / 100.0, [10, 50, 90, 100])))
box, boxl, tri, tril = (a[:, k].astype(float) for k in (2, 3, 4, 5))
print("fetches %.1f max stack %d" % (a[:, 14].mean(), a[:, 15].max()))
print("fetches %.1f max stack %d" % (a[:, 14].mean(), a[:, 15].max()))
print("per wave: BOX steps %.0f (occupancy %.3f)  TRI steps %.0f (occupancy %.3f)  refill rounds %.0f rays %.0f  serial %.1f idle loops %.1f" %
      (box.mean(), boxl.sum() / 128 / max(box.sum(), 1), tri.mean(), tril.sum() / 128 / max(tri.sum(), 1), a[:, 6].mean(), a[:, 7].mean(), a[:, 8].mean(), a[:, 13].mean()))
cy = a[:, 9:12].astype(float)
tot = cy.sum()
print("cycle shares: service %.3f tri %.3f box %.3f ; cycles per wave %.0f ; per BOX step %.0f per TRI step %.0f per refill round %.0f" %
      (cy[:, 0].sum() / tot, cy[:, 1].sum() / tot, cy[:, 2].sum() / tot, cy.sum(axis=1).mean(), cy[:, 2].sum() / max(box.sum(), 1), cy[:, 1].sum() / max(tri.sum(), 1), cy[:, 0].sum() / max(a[:, 6].sum(), 1)))
td = a[:, 12]; ok = td > 0
if ok.any():
    tail = (t1[ok] - td[ok]) / 100.0
    print("pool drained -> wave end us: mean %.1f p50 %.1f p90 %.1f max %.1f ; drain time p50 %.1f max %.1f" % (tail.mean(), np.median(tail), np.percentile(tail, 90), tail.max(), np.median((td[ok] - base) / 100.0), ((td[ok] - base) / 100.0).max()))
