import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
p = rt.make_params(1920, 1080, 1, 3, variant="wavefront", **rt.scenes.CPU_LAUNCHER)
for _ in range(3): ctx.render(p)
a = np.fromfile("gpurun_out/trav_dbg.bin", dtype=np.uint64).reshape(-1, 6)
t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
base = t0.min(); dur = (t1 - t0) / 100.0  # 100 MHz -> us
print("it", os.environ["RT_DEBUG_TRAV"], "waves", len(a), "kernel span us", (t1.max() - base) / 100.0)
print("start offset us: p50 %.1f p99 %.1f max %.1f" % tuple(np.percentile((t0 - base) / 100.0, [50, 99, 100])))
print("wave duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % ((dur.mean(),) + tuple(np.percentile(dur, [10, 50, 90, 99, 100]))))
print("end time us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((t1 - base) / 100.0, [10, 50, 90, 100])))
steps = a[:, 2].astype(float)
print("steps per wave: mean %.0f p50 %.0f p99 %.0f max %.0f ; splits mean %.1f" % (steps.mean(), np.median(steps), np.percentile(steps, 99), steps.max(), a[:, 4].mean()))
print("lane occupancy of steps: %.3f" % (a[:, 3].sum() / 64.0 / steps.sum()))
print("us per step: %.3f ; tri steps frac %.3f" % (dur.sum() / steps.sum(), a[:, 5].sum() / steps.sum()))
