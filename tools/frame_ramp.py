"""GPU box: per-frame GPU time of the first frames after a synchronize (is there a ramp?), and the host's enqueue time per frame."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, object_slot=6))
W, H = 1920, 1080
p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
rows, _ = rt.interleaved_rows(H, 8, 0, 1)
st = torch.cuda.Stream()
bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]
for pipe in (False, True):
    ctx.set_pipelining(pipe)
    for rep in range(2):
        with torch.cuda.stream(st):
            for k in range(5):
                ctx.render_device(p, rows, bufs[k & 1].data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            K = 20
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
            host = []
            t0 = time.perf_counter()
            ev[0].record()
            for k in range(K):
                h0 = time.perf_counter()
                ctx.render_device(p, rows, bufs[k & 1].data_ptr(), st.cuda_stream)
                host.append((time.perf_counter() - h0) * 1e3)
                ev[k + 1].record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
        per = [ev[k].elapsed_time(ev[k + 1]) for k in range(K)]
        print("pipelining %-5s: wall %.3f ms for %d frames = %.4f per frame; GPU per frame (event to event): %s ; host enqueue per frame: first %.3f, mean of the rest %.3f ms" %
              (pipe, wall, K, wall / K, " ".join("%.3f" % x for x in per), host[0], float(np.mean(host[1:]))), flush=True)
