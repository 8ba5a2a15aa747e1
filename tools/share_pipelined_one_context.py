"""GPU box: rank 0's share of the 1080p frame for world = 1, 2, 4, 8 rendered back to back on ONE context with rt_ctx_set_pipelining (two
buffers, two sub-frames), next to the figures of tools/share_scaling.py (one joined frame; two contexts with one frame each)."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
st = torch.cuda.Stream()
W, H = 1920, 1080
p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
for world in (1, 2, 4, 8):
    rows, _ = rt.interleaved_rows(H, 8, 0, world)
    bufs = [tiling.local_buffer(H, W, world, "cuda:0") for _ in range(2)]
    out = []
    for mode in (False, True):
        ctx.set_pipelining(mode)
        for k in range(200):
            ctx.render_device(p, rows, bufs[k & 1].data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        K = 200
        t0 = time.perf_counter()
        for k in range(K):
            ctx.render_device(p, rows, bufs[k & 1].data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / K * 1e3)
    print("1920x1080 world %d: one context, joined frames %.3f ms | pipelined frames %.3f ms" % (world, out[0], out[1]), flush=True)
ctx.set_pipelining(False)
