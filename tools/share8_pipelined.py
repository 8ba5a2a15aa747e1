"""GPU box: rank 0's share of the 1080p frame for world 8 with K frames in flight (K contexts, each with its own streams and path state,
frames dealt round-robin): what frame-level pipelining buys when one share no longer fills the chip."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
v, t = rt.scenes.load_cat_arrays()
mesh = hostlib.build_mesh(v, t, object_slot=6)
p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
for world in (1, 2, 4, 8):
    rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, world)
    for K in (1, 2, 3):
        ctxs = [rt.Context(0) for _ in range(K)]
        for c in ctxs: c.scene_upload(rt.scenes.spheres("cpu"), mesh)
        streams = [torch.cuda.Stream() for _ in range(K)]
        bufs = [tiling.local_buffer(H, W, world, "cuda:0") for _ in range(K)]
        for k in range(3 * K): ctxs[k % K].render_device(p, rows, bufs[k % K].data_ptr(), streams[k % K].cuda_stream)
        torch.cuda.synchronize()
        n = 60
        t0 = time.perf_counter()
        for k in range(n): ctxs[k % K].render_device(p, rows, bufs[k % K].data_ptr(), streams[k % K].cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{W}x{H} world {world}, {K} frame(s) in flight: {ms:.3f} ms per frame-share", flush=True)
        for c in ctxs: c.close()
