"""GPU box: SURVEY 8d measurement table -- BASELINE configs 2..5 on one GPU and the benchmark.py grid subset
(spp in {1,8,64} x num_bounce in {1,3,5}) on the cat at 1920x1080.  Kernel time from HIP events (rt_get_stats), median of
`reps` frames after a warm-up; rays counted exactly (framebuffer .w).  Prints a markdown table."""
import os, sys, statistics, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch  # noqa: F401  (one HIP runtime)
import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib, tiling

ctx = rt.Context(0)
v, t = rt.scenes.load_cat_arrays()
cat = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)


def run(scene, W, H, spp, b, reps=7, variant="auto"):
    ctx.scene_upload(rt.scenes.spheres(scene), cat if scene == "cpu" else None)
    p = rt.make_params(W, H, spp, b, variant=variant, **rt.scenes.CPU_LAUNCHER)
    rows, _ = rt.interleaved_rows(H, 8, 0, 1)
    buf = tiling.local_buffer(H, W, 1, "cuda:0")
    ms = []
    for k in range(reps + 2):
        ctx.render_device(p, rows, buf.data_ptr())
        ctx.synchronize()
        if k >= 2:
            ms.append(ctx.stats()["kernel_ms"])
    rays = float(buf[..., 3].double().sum().item())
    m = statistics.median(ms)
    return m, rays / m / 1e3


out = []
rows_ = [("config 2: walls + 4 demo spheres, no mesh", "demo10", 1920, 1080, 1, 3),
         ("config 3(i): cat, direct lighting", "cpu", 1920, 1080, 1, 0),
         ("config 3(ii): cat (headline)", "cpu", 1920, 1080, 1, 3),
         ("config 4: cat 3840x2160", "cpu", 3840, 2160, 1, 3),
         ("config 5 on ONE GPU: cat 7680x4320", "cpu", 7680, 4320, 1, 3)]
print("| workload | W x H | spp | b | ms/frame | Mrays/s |\n|---|---|---|---|---|---|")
for name, sc, W, H, spp, b in rows_:
    m, r = run(sc, W, H, spp, b)
    print(f"| {name} | {W}x{H} | {spp} | {b} | {m:.3f} | {r:,.0f} |", flush=True)
    out.append(dict(name=name, W=W, H=H, spp=spp, b=b, ms=m, mrays=r))
m, r = run("cpu", 3840, 2160, 1, 3, variant="wavefront_lds")
print(f"| config 4, every BVH node staged in LDS (variant wavefront_lds) | 3840x2160 | 1 | 3 | {m:.3f} | {r:,.0f} |", flush=True)
out.append(dict(name="config4_wavefront_lds", ms=m, mrays=r))
print("\n| cat 1920x1080 | b=1 | b=3 | b=5 |\n|---|---|---|---|")
for spp in (1, 8, 64):
    cells = []
    for b in (1, 3, 5):
        m, r = run("cpu", 1920, 1080, spp, b, reps=3 if spp == 64 else 5)
        cells.append(f"{m:.2f} ms, {r:,.0f} Mrays/s")
        out.append(dict(name="grid", spp=spp, b=b, ms=m, mrays=r))
    print(f"| spp={spp} | " + " | ".join(cells) + " |", flush=True)
json.dump(out, open("gpurun_out/grid_bench.json", "w"), indent=1)
