"""GPU box: time one rank's share of the frame for world = 1,2,4,8 on ONE GPU (no communication), at 1920x1080, 3840x2160 and
7680x4320: how well the kernels hold up when the per-GPU work shrinks (multi-GPU strong scaling, compute side).
usage: python tools/share_scaling.py [> profiles/roundN/share_scaling.txt]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
spp = int(os.environ.get("SPP", "1"))
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
sizes = [(1920, 1080), (3840, 2160), (7680, 4320)] if not os.environ.get("SIZES") else [tuple(int(x) for x in s.split("x")) for s in os.environ["SIZES"].split(",")]
for W, H in sizes:
    p = rt.make_params(W, H, spp, 3, variant=os.environ.get("RT_VARIANT", "auto"), **rt.scenes.CPU_LAUNCHER)
    base = None
    for world in (1, 2, 4, 8):
        rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, world)
        local = tiling.local_buffer(H, W, world, "cuda:0")
        n = 30 if W <= 1920 else 8
        for _ in range(3): ctx.render_device(p, rows, local.data_ptr(), side.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): ctx.render_device(p, rows, local.data_ptr(), side.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        base = base or ms
        print(f"{W}x{H} world {world}: {ms:.3f} ms per frame-share = {base / ms:.2f}x of the whole frame's rate (ideal {world}x), kernels {ctx.stats()['kernel_ms']:.3f} ms", flush=True)
        del local
