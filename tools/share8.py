"""GPU box: rank 0's share of the 1080p frame for world 8 (or WORLD=n), under the knobs of the environment.  Prints ms per share."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
world = int(os.environ.get("WORLD", 8))
p = rt.make_params(W, H, 1, 3, variant=os.environ.get("RT_VARIANT", "auto"), **rt.scenes.CPU_LAUNCHER)
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, world)
local = tiling.local_buffer(H, W, world, "cuda:0")
res = []
for rep in range(3):
    for _ in range(5): ctx.render_device(p, rows, local.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n): ctx.render_device(p, rows, local.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / n * 1e3)
st = ctx.stats()
print("%-60s world %d: %s ms (grid %d blocks, parts %d)" % (os.environ.get("TAG", ""), world, " ".join("%.3f" % x for x in res), st["grid_blocks"], st["parts"]))
