"""GPU box: SURVEY 8d measurement table -- BASELINE configs 2..5 on one GPU and the benchmark.py grid subset
(num_rays in 1..256 x num_bounce in 1..10, the whole sweep of the reference's benchmark.py) on the cat at 512x512 and 1920x1080.  Kernel time from HIP events (rt_get_stats), median of
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
for variant, what in (("lds_verts", "vertex array staged in LDS (optimized_vertices-in-shared.cu:681-686)"), ("lds_top", "top of the BVH staged in LDS"),
                      ("lds_all", "vertices + top of the BVH staged in LDS"), ("wavefront_lds", "per-lane walk, every node staged in LDS"),
                      ("path", "one persistent launch (wf_path)")):
    m, r = run("cpu", 3840, 2160, 1, 3, variant=variant)
    print(f"| config 4, variant {variant}: {what} | 3840x2160 | 1 | 3 | {m:.3f} | {r:,.0f} |", flush=True)
    out.append(dict(name="config4_" + variant, ms=m, mrays=r))
# benchmark.py's sweep (/root/reference/benchmark.py:10-33): num_rays in 1, 2, 4 .. 256 (rows) x num_bounce in 1 .. 10 (columns); the
# reference times its whole 512x512 program, here: kernel ms per frame, at the reference's own size and at the BASELINE size
for W, H in ((512, 512), (1920, 1080)):
    print(f"\ncat {W}x{H}: ms per frame, rows = num_rays, columns = num_bounce 1..10\n")
    print("| num_rays | " + " | ".join(str(b) for b in range(1, 11)) + " |\n|---|" + "---|" * 10)
    for spp in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        cells = []
        for b in range(1, 11):
            m, r = run("cpu", W, H, spp, b, reps=5 if spp <= 4 else 3 if spp <= 32 else 1)
            cells.append(f"{m:.2f}")
            out.append(dict(name="grid", W=W, H=H, spp=spp, b=b, ms=m, mrays=r))
        print(f"| {spp} | " + " | ".join(cells) + " |", flush=True)
json.dump(out, open("gpurun_out/grid_bench.json", "w"), indent=1)
