"""GPU box (under rocprofv3 --kernel-trace --stats): rank 0's share of the 1080p frame for ONE world size, 40 frames."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
world = int(os.environ.get("WORLD", "8"))
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
W, H = 1920, 1080
p = rt.make_params(W, H, 1, 3, variant=os.environ.get("RT_VARIANT", "auto"), **rt.scenes.CPU_LAUNCHER)
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, world)
local = tiling.local_buffer(H, W, world, "cuda:0")
for _ in range(5): ctx.render_device(p, rows, local.data_ptr(), side.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): ctx.render_device(p, rows, local.data_ptr(), side.cuda_stream)
torch.cuda.synchronize()
print("world", world, "ms per share %.4f" % ((time.perf_counter() - t0) / 40 * 1e3))
