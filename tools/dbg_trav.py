import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
p = rt.make_params(1920, 1080, 1, 3, variant=os.environ.get("RT_VARIANT", "wavefront"), **rt.scenes.CPU_LAUNCHER)
world = int(os.environ.get("WORLD", "1"))
import torch
from raytracinggpu_amd import tiling
rows, idx = rt.interleaved_rows(1080, 8, 0, world)
local = tiling.local_buffer(1080, 1920, world, "cuda:0")
for _ in range(3):
    ctx.render_device(p, rows, local.data_ptr()); ctx.synchronize()
raw = np.fromfile("gpurun_out/trav_dbg.bin", dtype=np.uint64)
nw = ctx.stats()["grid_blocks"] * (ctx.stats()["block_threads"] // 64)
a = raw[:6 * nw].reshape(-1, 6)
cy = raw[6 * 65536:6 * 65536 + 4 * nw].reshape(-1, 4).astype(float)
tot = cy.sum()
print("cycle shares: box %.3f expand %.3f tri %.3f retire/refill/split %.3f ; cycles per wave %.0f" % (cy[:, 0].sum() / tot, cy[:, 1].sum() / tot, cy[:, 2].sum() / tot, cy[:, 3].sum() / tot, cy.sum(axis=1).mean()))


t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
base = t0.min(); dur = (t1 - t0) / 100.0  # 100 MHz -> us
print("it", os.environ["RT_DEBUG_TRAV"], "waves", len(a), "kernel span us", (t1.max() - base) / 100.0)
print("start offset us: p50 %.1f p99 %.1f max %.1f" % tuple(np.percentile((t0 - base) / 100.0, [50, 99, 100])))
print("wave duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % ((dur.mean(),) + tuple(np.percentile(dur, [10, 50, 90, 99, 100]))))
print("end time us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((t1 - base) / 100.0, [10, 50, 90, 100])))
steps = a[:, 2].astype(float)
print("steps per wave: mean %.0f p50 %.0f p99 %.0f max %.0f ; splits mean %.1f" % (steps.mean(), np.median(steps), np.percentile(steps, 99), steps.max(), a[:, 4].mean()))
print("lane occupancy of steps: %.3f" % (a[:, 3].sum() / 64.0 / steps.sum()))
td = a[:, 5].astype(np.int64)
ok = td > 0
tail = (t1[ok] - td[ok]) / 100.0
print("time from pool drained to wave end (us): mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f ; drain time p50 %.1f max %.1f" % (tail.mean(), np.median(tail), np.percentile(tail, 90), np.percentile(tail, 99), tail.max(), np.median((td[ok] - base) / 100.0), ((td[ok] - base) / 100.0).max()))
