"""GPU box: frames in flight on ONE context (rt_ctx_set_pipelining): K frames into two alternating device buffers on one stream, with and
without pipelining, at 1920x1080 and 7680x4320; every buffer compared bit for bit with a frame rendered alone."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
st = torch.cuda.Stream()
for (W, H, K) in ((1920, 1080, 40), (7680, 4320, 6)):
    p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
    rows, _ = rt.interleaved_rows(H, 8, 0, 1)
    ref = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
    ctx.set_pipelining(False)
    ctx.render_device(p, rows, ref.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    for mode in (False, True, False, True):
        ctx.set_pipelining(mode)
        for b in bufs: b.zero_()
        torch.cuda.synchronize()
        for k in range(4):
            ctx.render_device(p, rows, bufs[k & 1].data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            ctx.render_device(p, rows, bufs[k & 1].data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        same = all(bool((b.view(torch.int32) == ref.view(torch.int32)).all()) for b in bufs)
        print("%dx%d pipelining %-5s: %.4f ms per frame, frames bitwise equal to a lone frame: %s" % (W, H, mode, ms, same), flush=True)
    # one buffer only: the library must fall back to the full fork (same buffer = the previous frame's readers could be anywhere)
    ctx.set_pipelining(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        ctx.render_device(p, rows, bufs[0].data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    print("%dx%d pipelining on, ONE buffer: %.4f ms per frame, equal: %s" % (W, H, (time.perf_counter() - t0) / K * 1e3, bool((bufs[0].view(torch.int32) == ref.view(torch.int32)).all())), flush=True)
