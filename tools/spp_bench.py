"""GPU box: frame time vs samples per pixel for the default variant, in a process that owns nothing but one context (no other streams: the runtime maps streams onto four
hardware queues, and a tool that has created a handful of them can put a context's two sub-frame streams on ONE queue -- they then run one after the other).
usage: [RT_LIB=...] python tools/spp_bench.py"""
import os, sys, statistics, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6))
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
rows, _ = rt.interleaved_rows(H, 8, 0, 1)
buf = tiling.local_buffer(H, W, 1, "cuda:0")
for variant in os.environ.get("VARIANTS", "auto").split(","):
    for spp, b in [(int(x), 3) for x in os.environ.get("SPP", "1,2,8,32,64,128,256").split(",")]:
        p = rt.make_params(W, H, spp, b, variant=variant, **rt.scenes.CPU_LAUNCHER)
        ms, wall = [], []
        for k in range(4 if spp < 128 else 3):
            t0 = time.perf_counter()
            ctx.render_device(p, rows, buf.data_ptr()); ctx.synchronize()
            if k > 0: ms.append(ctx.stats()["kernel_ms"]); wall.append((time.perf_counter() - t0) * 1e3)
        print(variant, "spp", spp, "b", b, "kernel ms %.3f" % statistics.median(ms), "per sample %.4f" % (statistics.median(ms) / spp), "| wall %.3f per sample %.4f" % (statistics.median(wall), statistics.median(wall) / spp), flush=True)
