"""GPU box: host-visible frame time of the pipelined path (rt_render_async / rt_wait) for the float4 frame and the 8-bit image."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
W, H = 1920, 1080
p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
for rgb8 in (False, True):
    bufs = [rt.PinnedArray((H, W, 3), dtype=np.uint8) if rgb8 else rt.PinnedArray((H, W, 4)) for _ in range(2)]
    for k in range(2):
        ctx.render_async(p, bufs[k].array, slot=k, rgb8=rgb8); ctx.wait(k)
    n = 40
    t1 = time.perf_counter()
    ctx.render_async(p, bufs[0].array, slot=0, rgb8=rgb8)
    for k in range(1, n):
        ctx.render_async(p, bufs[k & 1].array, slot=k & 1, rgb8=rgb8)
        ctx.wait((k - 1) & 1)
    ctx.wait((n - 1) & 1)
    print("%s pipelined %s: %.3f ms per frame" % (os.environ.get("TAG", ""), "rgb8" if rgb8 else "float4", (time.perf_counter() - t1) / n * 1e3))
