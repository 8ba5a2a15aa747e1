"""GPU box: time one rank's share of the frame for world = 1,2,4,8 on ONE GPU (no communication):
how well the kernels hold up when the per-GPU work shrinks (multi-GPU strong scaling, compute side)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
W, H = 1920, 1080
spp = int(os.environ.get("SPP", "1"))
p = rt.make_params(W, H, spp, 3, variant=os.environ.get("RT_VARIANT", "auto"), **rt.scenes.CPU_LAUNCHER)
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
for world in (1, 2, 4, 8):
    rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, world)
    local = tiling.local_buffer(H, W, world, "cuda:0")
    for _ in range(5): ctx.render_device(p, rows, local.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n): ctx.render_device(p, rows, local.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    st = ctx.stats()
    print(f"world {world}: {ms:.3f} ms per frame-share (ideal {1.0/world:.3f}x), trav {st['trav_ms']:.3f} ms in {st['trav_launches']} launches, kernels {st['kernel_ms']:.3f} ms")
