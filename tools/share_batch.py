"""GPU box: the compute side of the N-rank job on ONE GPU, with the denominators bench.py's own lines use.
For world = 2, 4, 8 at 1920x1080 (and 3840x2160): rank 0's share of the frame (interleaved 8-row tiles)
  * as bench.py --gpus N renders it since round 6: `world` consecutive frames as ONE launch chain (rt_render_device_batch), batches alternating between two sets of buffers on one
    context and stream with rt_ctx_set_pipelining (and the same with every batch joined before the next starts);
  * as round 5 rendered it: four frames in flight on four contexts (RT_PARTS=1, RT_TRAV_MIN_GROUPS=64);
  * one frame alone (the latency of a frame on that share);
against the WHOLE frame rendered the way bench.py --gpus 1 renders it (two frames alternating on one context and stream, rt_ctx_set_pipelining) -- the `ms_per_step` a
scaling run divides by -- and against one whole frame alone.  No exchange: this is the compute side only.
usage: python tools/share_batch.py [> profiles/roundN/share_batch.txt]"""
import os, sys, time
os.environ.setdefault("RT_EXPERIMENT", "1")   # the launch-geometry knobs below are honoured only under it
os.environ.setdefault("RT_PART_PRIO", "1")
sys.path.insert(0, os.getcwd())
import torch
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
v, t = rt.scenes.load_cat_arrays()
mesh = hostlib.build_mesh(v, t, object_slot=6)
sizes = [(1920, 1080), (3840, 2160)] if not os.environ.get("SIZES") else [tuple(int(x) for x in s.split("x")) for s in os.environ["SIZES"].split(",")]
N = int(os.environ.get("FRAMES", "96"))


def contexts(k, env=None):
    old = {n: os.environ.get(n) for n in (env or {})}
    os.environ.update(env or {})
    cs = [rt.Context(0) for _ in range(k)]
    for n, val in old.items():
        if val is None:
            del os.environ[n]
        else:
            os.environ[n] = val
    for c in cs:
        c.scene_upload(rt.scenes.spheres("cpu"), mesh)
    return cs


def run(fn, frames, warm):
    for k in range(warm):
        fn(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(frames):
        fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


full = contexts(2)
lanes4 = contexts(4, {"RT_PARTS": "1", "RT_TRAV_MIN_GROUPS": "64"})
st = [torch.cuda.Stream(priority=-1 if (k & 1) else 0) for k in range(4)]
for W, H in sizes:
    p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
    rows1, _ = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, 1)
    bufs = [tiling.local_buffer(H, W, 1, "cuda:0") for _ in range(2)]
    full[0].set_pipelining(True)
    n1 = run(lambda k: full[0].render_device(p, rows1, bufs[k & 1].data_ptr(), st[0].cuda_stream), N if W <= 1920 else 16, 12)
    full[0].set_pipelining(False)
    n1_alone = run(lambda k: full[0].render_device(p, rows1, bufs[0].data_ptr(), st[0].cuda_stream), N if W <= 1920 else 16, 6)
    print(f"{W}x{H} whole frame: {n1:.4f} ms per frame as bench.py --gpus 1 renders it (two frames in flight, one context) | {n1_alone:.4f} ms one frame alone", flush=True)
    del bufs
    for world in (2, 4, 8):
        rows, _ = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, world)
        K = min(16, world)
        lb = [tiling.local_buffer(H, W, world, "cuda:0") for _ in range(2 * K)]

        descs = [[(lb[b * K + j].data_ptr(), (0.0, 0.0, 55.0), None, 123456) for j in range(K)] for b in range(2)]

        def batch(k):                                                 # as bench.py --gpus N: one context, one stream, batches alternating between two sets of buffers, pipelining on
            full[0].render_device_batch(p, rows, descs[k & 1], st[0].cuda_stream)
        nb = max(8, (N if W <= 1920 else 32) // K)
        full[0].set_pipelining(True)
        t_batch = run(batch, nb, 4) / K
        full[0].set_pipelining(False)
        t_batch_alone = run(lambda k: full[0].render_device_batch(p, rows, descs[0], st[0].cuda_stream), nb, 4) / K
        t_four = run(lambda k: lanes4[k & 3].render_device(p, rows, lb[k & 3].data_ptr(), st[k & 3].cuda_stream), N if W <= 1920 else 32, 12)
        t_alone = run(lambda k: full[0].render_device(p, rows, lb[0].data_ptr(), st[0].cuda_stream), N if W <= 1920 else 32, 6)
        print(f"{W}x{H} world {world}: batch of {K} frames in one chain {t_batch:.4f} ms per frame = {n1 / t_batch:.2f}x of bench --gpus 1 (ideal {world}x; one batch at a time {t_batch_alone:.4f} ms per frame) | "
              f"four frames in flight (round 5) {t_four:.4f} ms = {n1 / t_four:.2f}x | one frame alone (latency) {t_alone:.4f} ms = {n1_alone / t_alone:.2f}x of a lone whole frame", flush=True)
        del lb
