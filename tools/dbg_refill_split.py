import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
p = rt.make_params(1920, 1080, 1, 3, variant="wavefront_queue", **rt.scenes.CPU_LAUNCHER)
rows, idx = rt.interleaved_rows(1080, 8, 0, 1)
local = tiling.local_buffer(1080, 1920, 1, "cuda:0")
for _ in range(3):
    ctx.render_device(p, rows, local.data_ptr()); ctx.synchronize()
raw = np.fromfile("gpurun_out/trav_dbg.bin", dtype=np.uint64)
st = ctx.stats(); nw = st["grid_blocks"] * (st["block_threads"] // 64)
a = raw[:16 * nw].reshape(-1, 16).astype(np.float64)
srv, tri, box = a[:, 9].sum(), a[:, 10].sum(), a[:, 11].sum()
ret, fetch, hand = a[:, 12].sum(), a[:, 13].sum(), a[:, 8].sum()
tot = srv + tri + box + ret + fetch + hand
print("launch", os.environ["RT_DEBUG_TRAV"], "shares: retire %.3f fetch %.3f handoff %.3f rest-of-service %.3f tri %.3f box %.3f | per fetch %.0f cycles, per round handoff %.0f, fetches/wave %.1f rounds/wave %.1f" %
      (ret / tot, fetch / tot, hand / tot, srv / tot, tri / tot, box / tot, fetch / max(a[:, 14].sum(), 1), hand / max(a[:, 6].sum(), 1), a[:, 14].mean(), a[:, 6].mean()))
