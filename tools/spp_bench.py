"""GPU box: frame time vs samples per pixel for the default variant (RT_PATH_SAMP_MB sets how many samples a launch chain traces together)."""
import os, sys, statistics
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
    for spp, b in ((1, 3), (2, 3), (8, 3), (64, 3)):
        p = rt.make_params(W, H, spp, b, variant=variant, **rt.scenes.CPU_LAUNCHER)
        ms = []
        for k in range(5):
            ctx.render_device(p, rows, buf.data_ptr()); ctx.synchronize()
            if k > 1: ms.append(ctx.stats()["kernel_ms"])
        print(variant, "spp", spp, "b", b, "ms %.3f" % statistics.median(ms), "per sample %.3f" % (statistics.median(ms) / spp), flush=True)
