"""GPU box: time one rank's share of the frame for world = 1,2,4,8 on ONE GPU (no communication), at 1920x1080, 3840x2160 and
7680x4320: how well the kernels hold up when the per-GPU work shrinks (multi-GPU strong scaling, compute side).  Two columns:
the library's default for a lone frame (one frame, two concurrent sub-frames), two whole frames in flight (one context each, sub-frames off: what bench.py ran for
small shares until round 4), and round 5's rule for small shares: FOUR frames in flight on contexts created under RT_TRAV_MIN_GROUPS=64 (fewer, fuller workgroups).
usage: python tools/share_scaling.py [> profiles/roundN/share_scaling.txt]"""
import os, sys, time
os.environ.setdefault("RT_EXPERIMENT", "1")   # the launch-geometry knobs below are honoured only under it
os.environ.setdefault("RT_PART_PRIO", "1")       # three contexts live in this process (as in a bench.py rank with N > 1): see Knobs::part_prio in rt_host_ctx.hip.h
sys.path.insert(0, os.getcwd())
import torch
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling
v, t = rt.scenes.load_cat_arrays()
mesh = hostlib.build_mesh(v, t, object_slot=6)
spp = int(os.environ.get("SPP", "1"))
sizes = [(1920, 1080), (3840, 2160), (7680, 4320)] if not os.environ.get("SIZES") else [tuple(int(x) for x in s.split("x")) for s in os.environ["SIZES"].split(",")]


def contexts(k, parts, min_groups=None):
    old = os.environ.get("RT_PARTS")
    if parts:
        os.environ["RT_PARTS"] = str(parts)
    if min_groups:
        os.environ["RT_TRAV_MIN_GROUPS"] = str(min_groups)
    cs = [rt.Context(0) for _ in range(k)]
    if min_groups:
        del os.environ["RT_TRAV_MIN_GROUPS"]
    if parts:
        if old is None:
            del os.environ["RT_PARTS"]
        else:
            os.environ["RT_PARTS"] = old
    for c in cs:
        c.scene_upload(rt.scenes.spheres("cpu"), mesh)
    return cs


def measure(cs, p, rows, H, W, world, n):
    K = len(cs)
    streams = [torch.cuda.Stream(priority=-1 if (k & 1) else 0) for k in range(K)]   # as bench.py's lanes: the odd lane in the high-priority queue pool
    bufs = [tiling.local_buffer(H, W, world, "cuda:0") for _ in range(K)]
    for k in range(3 * K):
        cs[k % K].render_device(p, rows, bufs[k % K].data_ptr(), streams[k % K].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        cs[k % K].render_device(p, rows, bufs[k % K].data_ptr(), streams[k % K].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = contexts(1, 0)
two = contexts(2, 1)
four = contexts(4, 1, 64)
for W, H in sizes:
    p = rt.make_params(W, H, spp, 3, variant=os.environ.get("RT_VARIANT", "auto"), **rt.scenes.CPU_LAUNCHER)
    base = None
    for world in (1, 2, 4, 8):
        rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, 0, world)
        n = 60 if W <= 1920 else 8
        a = measure(one, p, rows, H, W, world, n)
        b = measure(two, p, rows, H, W, world, n) if W <= 3840 else float("nan")
        c = measure(four, p, rows, H, W, world, n) if (W <= 3840 and world > 1) else float("nan")
        base = base or a
        best = min(x for x in (a, b, c) if x == x)
        print(f"{W}x{H} world {world}: one frame, two sub-frames {a:.3f} ms | two frames in flight {b:.3f} ms | four in flight, 256 ray slots per wave {c:.3f} ms | best = {base / best:.2f}x of the whole frame's rate (ideal {world}x)", flush=True)
