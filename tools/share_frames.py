"""GPU box: rank 0's share of a 1920x1080 frame for world = 8 (and 4) on ONE GPU with K whole frames in flight, K = 1 .. 6 (one context and one stream per frame
slot, sub-frames off), next to one frame cut into two sub-frames: how many frames a rank must keep in flight before the chip is full when its share is small
(VERDICT round 4 item 2).  The environment of the calling shell applies to every context (RT_TRAVQ_QW, GPU_MAX_HW_QUEUES ...).
usage: python tools/share_frames.py [> profiles/roundN/share_frames.txt]"""
import os, sys, time
os.environ.setdefault("RT_EXPERIMENT", "1")   # the launch-geometry knobs below are honoured only under it
os.environ.setdefault("RT_PART_PRIO", "1")
sys.path.insert(0, os.getcwd())
import torch
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
v, t = rt.scenes.load_cat_arrays()
mesh = hostlib.build_mesh(v, t, object_slot=6)
W, H = [int(x) for x in os.environ.get("SIZE", "1920x1080").split("x")]
KS = [int(x) for x in os.environ.get("KS", "1,2,3,4,6").split(",")]
WORLDS = [int(x) for x in os.environ.get("WORLDS", "8,4").split(",")]


def contexts(k, parts):
    old = os.environ.get("RT_PARTS")
    os.environ["RT_PARTS"] = str(parts)
    cs = [rt.Context(0) for _ in range(k)]
    if old is None:
        del os.environ["RT_PARTS"]
    else:
        os.environ["RT_PARTS"] = old
    for c in cs:
        c.scene_upload(rt.scenes.spheres("cpu"), mesh)
    return cs


def measure(cs, streams, p, rows, world, n):
    K = len(cs)
    bufs = [tiling.local_buffer(H, W, world, "cuda:0") for _ in range(K)]
    for k in range(3 * K):
        cs[k % K].render_device(p, rows, bufs[k % K].data_ptr(), streams[k % K].cuda_stream)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(n):
            cs[k % K].render_device(p, rows, bufs[k % K].data_ptr(), streams[k % K].cuda_stream)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
whole_rows, _ = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, 1)
one = contexts(1, 2)
whole = measure(one, [torch.cuda.Stream()], p, whole_rows, 1, 60)
print(f"env: " + " ".join(f"{k}={os.environ[k]}" for k in ("RT_TRAVQ_QW", "GPU_MAX_HW_QUEUES", "RT_PART_PRIO") if k in os.environ))
print(f"whole frame, one frame in two sub-frames: {whole:.3f} ms")
pool = contexts(max(KS), 1)
streams = [torch.cuda.Stream(priority=-1 if (k & 1) else 0) for k in range(max(KS))]
for world in WORLDS:
    rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, world)
    a = measure(one, streams[:1], p, rows, world, 120)
    line = f"world {world}: one frame in two sub-frames {a:.3f} ms ({whole / a:.2f}x)"
    for K in KS:
        b = measure(pool[:K], streams[:K], p, rows, world, 120)
        line += f" | {K} in flight {b:.3f} ms ({whole / b:.2f}x)"
    print(line, flush=True)
