#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path (BASELINE.json: Mrays/s + ms/frame, cat mesh
1920x1080, at 1/2/4/8 MI355X).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one frame of the workload: the cat scene of cpu_launcher.cpp (walls + cat mesh + array BVH),
1920x1080, num_rays=1, num_bounce=3 (CPU convention: 4 segments), rendered by the HIP kernel through the
C-ABI with the scene already resident in HBM.  With N > 1 the frame is split into interleaved 8-row tiles
(tile k -> rank k mod N), every rank renders its tiles, and ONE RCCL gather per frame brings the float4
tiles to rank 0, which de-interleaves them (all inside the timed region).  Total work is fixed => "strong".

Rank 0 prints one JSON line.  `value` = rays traced per second (1 ray = 1 Scene::intersect_all call:
primary, shadow or bounce segment; counted exactly by the kernel in the framebuffer's .w channel).
`roofline` prices the render kernel against the HBM roofline with the ALGORITHMIC bytes of SURVEY 8d
(24 B/box test + 16 B/node + 48 B/triangle test + 16 B/pixel).  `cpu_baseline` is the CPU restatement of
cpu_launcher.cpp (oracle/, OpenMP schedule(dynamic,1) over rows like cpu:695) timed on this host.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
TILE_ROWS = 8              # == raytracinggpu_amd.tiling.TILE_ROWS

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--scene", default="cpu", choices=["cpu", "spheres", "demo10"])
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--gather", default="f32", choices=["f32", "rgb8"],
                    help="N > 1: what rank 0 gathers -- the float4 tiles (parity path, default) or the tonemapped RGB8 tiles (PNG path, 3 B/pixel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def build_scene(rt, ctx, scene):
    """Upload the preset through the C-ABI.  The mesh arrays are the reference's own layouts."""
    from raytracinggpu_amd import hostlib
    mesh = None
    if scene == "cpu":
        verts, tris = rt.scenes.load_cat_arrays()
        mesh = hostlib.build_mesh(verts, tris, albedo=rt.scenes.CAT_ALBEDO, object_slot=rt.scenes.mesh_slot(scene))
    ctx.scene_upload(rt.scenes.spheres(scene), mesh)


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(np.ceil(int(q) / int(per)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, rays_per_frame):
    """The oracle (CPU restatement of cpu_launcher.cpp: recursive getColor, pointer BVH + explicit stack,
    OpenMP schedule(dynamic,1) over rows) on this host's cores; rank 0, N=1 only; bounded sample."""
    from oracle import oracle_py as orc
    import raytracinggpu_amd as rt
    threads = args.cpu_threads or host_cores()
    mesh = None
    if args.scene == "cpu":
        verts, tris = rt.scenes.load_cat_arrays()
        mesh = orc.Mesh.from_arrays(verts, tris).build_bvh()
    sc = orc.Scene.preset(args.scene, mesh)
    W, H = args.width, args.height
    kw = dict(threads=threads, want_rgb8=False, tile_rows=TILE_ROWS)
    # probe: every 16th 8-row tile (interleaved => representative of the frame), one parallel region
    t0 = time.perf_counter()
    _, _, c = sc.render(W, H, args.spp, args.bounces, tile_step=16, **kw)
    probe = time.perf_counter() - t0
    est_full = probe * rays_per_frame / max(c["rays"], 1)
    step = 1 if est_full <= 10.0 else int(np.ceil(est_full / 10.0))
    reps = 3 if est_full * 3 <= 12.0 else 1
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _, _, c = sc.render(W, H, args.spp, args.bounces, tile_step=step, **kw)
        times.append(time.perf_counter() - t0)
    sec = statistics.median(times)
    what = f"full {W}x{H} frame" if step == 1 else f"every {step}th 8-row tile of the {W}x{H} frame"
    out = {"value": round(c["rays"] / sec / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"{what}, num_rays={args.spp}, num_bounce={args.bounces}, {c['rays']} rays, median of {reps} run(s), pixel loop only",
           "seconds": round(sec, 4), "ms_per_frame_equiv": round(1e3 * rays_per_frame / (c["rays"] / sec), 2)}
    if args.scene == "cpu":
        try:
            out["reference_check"] = reference_check(sc, threads)
        except Exception as e:  # never lets the check take the baseline down
            out["reference_check"] = {"skipped": str(e)}
    return out


def reference_check(sc, threads):
    """The port is the timed baseline because the reference program is hard-wired to 512x512 and to an OBJ path.  Where
    the build container compiled the reference itself (oracle/_ref/cpu, `make -C oracle ref`; it travels with the
    snapshot), run THAT program at its own size next to the port: same host, same threads, `cpu 8 3` in a directory that
    holds the cat as an OBJ rebuilt from the fixture (6-number vertex lines are not transformed by readOBJ, cpu:344-350)."""
    import subprocess, tempfile
    import raytracinggpu_amd as rt
    exe = os.path.join(ROOT, "oracle", "_ref", "cpu")
    if not os.path.exists(exe):
        return {"skipped": "oracle/_ref/cpu not built"}
    g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
    with tempfile.TemporaryDirectory() as d:
        od = os.path.join(d, "cadnav.com_model", "Models_F0202A090")
        os.makedirs(od)
        with open(os.path.join(od, "cat.obj"), "w") as f:
            for v in g["vertices"]:
                f.write("v %.9g %.9g %.9g 1 1 1\r\n" % tuple(float(x) for x in v))
            for t in g["tri_obj_order"]:
                f.write("f %d/1/1 %d/1/1 %d/1/1\r\n" % tuple(int(x) + 1 for x in t))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        ref_times = []
        for _ in range(3):
            r = subprocess.run([exe, "8", "3"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            ref_times.append(float(r.stdout.split("Rendering time:")[1].split()[0]))
    port_times = []
    for _ in range(3):
        t0 = time.perf_counter()
        sc.render(512, 512, 8, 3, threads=threads, want_rgb8=False)
        port_times.append(time.perf_counter() - t0)
    return {"config": "512x512, num_rays=8, num_bounce=3 (the reference program's own size)", "threads": threads,
            "reference_program_s": round(statistics.median(ref_times), 4), "port_pixel_loop_s": round(statistics.median(port_times), 4),
            "note": "the reference times its whole program (OBJ parse, BVH build, PNG) and draws from a clock()-seeded mt19937"}


def single_stream_launch_ms(rt, args, p, rows, local, stream):
    """Average duration of ONE launch of the dominant kernel when it owns the chip: a second context created under
    RT_PARTS=1 (knobs are read once per context) renders the same frames as one sub-frame on one stream; the library
    brackets the traversal launches with HIP events on the stream they run on."""
    old = os.environ.get("RT_PARTS")
    os.environ["RT_PARTS"] = "1"
    try:
        c1 = rt.Context(int(os.environ.get("LOCAL_RANK", "0")))
    finally:
        if old is None:
            del os.environ["RT_PARTS"]
        else:
            os.environ["RT_PARTS"] = old
    build_scene(rt, c1, args.scene)
    ms, launches, frame = [], 0, []
    for k in range(8):
        c1.render_device(p, rows, local.data_ptr(), stream)
        st = c1.stats()
        if k >= 3 and st["trav_launches"] > 0:
            ms.append(st["trav_ms"] / st["trav_launches"]); launches = st["trav_launches"]; frame.append(st["kernel_ms"])
    c1.close()
    return (statistics.median(ms), launches, statistics.median(frame)) if ms else (None, 0, None)


def roofline(rt, ctx, args, p, rows, local, stream, counts, world, W, H, kernel_ms_max, workload):
    """nominal HBM roofline of the dominant kernel from ALGORITHMIC bytes, plus what really binds it.

    achieved = algorithmic bytes of one launch / that launch's duration, measured live with HIP events by the library on
    the stream the kernel runs on, in a single-stream context (one launch owns the chip).  The scene (~255 KB) is cache
    resident, so the nominal fraction is not a utilisation: `binding` quotes the measured HBM traffic and the VALU / SALU
    issue figures of the same kernel from the committed rocprofv3 summary (profiles/round2/summary.json, produced by
    tools/round_profile.sh + tools/make_profile_summary.py)."""
    st = ctx.stats()
    trav_bytes = 24 * counts["box_tests"] + 16 * counts["nodes"] + 48 * counts["tri_tests"]
    fb_bytes = 16 * W * H
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
           "frac_is": "nominal: SURVEY 8d algorithmic bytes (24 B/box test + 16 B/node + 48 B/triangle test) of a cache-resident scene over the "
                      "HBM peak; it may exceed 1 and is NOT a utilisation -- see `binding`",
           "frame_algorithmic_bytes": int(trav_bytes + fb_bytes), "frame_kernels_ms": round(kernel_ms_max, 4),
           "per_ray": {k: round(counts[k] / counts["rays"], 3) for k in ("box_tests", "nodes", "tri_tests")}}
    if st["trav_launches"] > 0:
        kname = {6: "rtk::wf_trav<false, false>", 7: "rtk::wf_trav<false, true>"}.get(st["variant"], "rtk::wf_travq<false, 64, false, false>")
        parts = max(st.get("parts", 1), 1)
        k_ms_conc = st["trav_ms"] / st["trav_launches"]
        single_ms, single_launches, single_frame = (None, 0, None)
        if world == 1:
            single_ms, single_launches, single_frame = single_stream_launch_ms(rt, args, p, rows, local, stream)
        if single_ms:
            alg_launch = trav_bytes / single_launches
            k_ms = single_ms
            out.update({"kernel_ms": round(single_ms, 4), "launches_per_frame": single_launches, "concurrent_launches": 1,
                        "single_stream_frame_ms": round(single_frame, 4)})
        else:
            alg_launch = trav_bytes / world / (st["trav_launches"] * parts)
            k_ms = k_ms_conc
            out.update({"kernel_ms": round(k_ms, 4), "launches_per_frame": st["trav_launches"] * parts, "concurrent_launches": parts})
        out["kernel_ms_two_streams"] = round(k_ms_conc, 4)            # the default configuration: the twin launch of the other sub-frame shares the chip
    else:
        kname = {1: "rtk::render_persistent<false>", 9: "rtk::wf_path<false>"}.get(st["variant"], "rtk::render_kernel<false>")
        k_ms, alg_launch = kernel_ms_max, (trav_bytes + fb_bytes) / world
        out.update({"kernel_ms": round(k_ms, 4), "launches_per_frame": 1, "concurrent_launches": 1})
    ach = alg_launch / (k_ms * 1e-3) / 1e9
    out.update({"kernel": kname, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(alg_launch)})
    spath = os.path.join(ROOT, "profiles", "round2", "summary.json")
    if os.path.exists(spath) and workload == "cat_1920x1080_spp1_b3" and st["trav_launches"] > 0 and st["variant"] == 8:
        ks = json.load(open(spath))["kernels"]
        t, a = ks["wf_travq"], ks["wf_advance"]
        hbm = t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"]
        out["traffic"] = hbm                                          # PMC, per single-stream launch like `achieved` (FETCH_SIZE x 2 + WRITE_SIZE)
        out["traffic_source"] = "profiles/round2/summary.json (rocprofv3 --pmc, RT_PARTS=1)"
        out["binding"] = {
            "resource": "VALU issue (wf_travq: bookkeeping of the work stack, ~2/3 of its vector instructions) and, for wf_advance, HBM",
            "wf_travq": {k: t[k] for k in ("rocprof_avg_us_single_stream", "rocprof_avg_us_two_streams", "share_of_gpu_time", "hbm_frac_of_8TBps", "l2_hit_rate",
                                           "valu_pipe_busy_frac", "valu_issue_per_simd_cycle", "salu_issue_per_cu_cycle", "valu_lane_utilization",
                                           "wave_cycles_waiting_frac", "valu_wave_insts_per_launch", "salu_wave_insts_per_launch")},
            "wf_advance": {k: a[k] for k in ("rocprof_avg_us_single_stream", "rocprof_avg_us_two_streams", "share_of_gpu_time", "hbm_read_bytes_per_launch",
                                             "hbm_write_bytes_per_launch", "hbm_GBps_single_stream", "hbm_frac_of_8TBps", "valu_pipe_busy_frac", "valu_lane_utilization")},
            "source": "profiles/round2/summary.json <- pmc_wf_travq.json, pmc_wf_advance.json, bench_kernel_stats.csv, single_stream_kernel_stats.csv"}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import raytracinggpu_amd as rt
    from raytracinggpu_amd import tiling

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctx = rt.Context(local_rank)
    build_scene(rt, ctx, args.scene)
    W, H = args.width, args.height
    p = rt.make_params(W, H, args.spp, args.bounces, variant=args.variant, **rt.scenes.CPU_LAUNCHER)
    rows, idx = rt.interleaved_rows(H, TILE_ROWS, rank, world)
    local = tiling.local_buffer(H, W, world, dev)
    rgb8 = args.gather == "rgb8" and world > 1
    local8 = tiling.local_buffer(H, W, world, dev, rgb8=True) if rgb8 else None
    gathered = tiling.gather_buffer(local8 if rgb8 else local, world) if (world > 1 and rank == 0) else None
    frame = None
    # a non-default torch stream: its handle is non-NULL (NULL means "the context's own stream" in the C-ABI),
    # so the render kernel, torch's timing events and the RCCL gather are all ordered on ONE stream
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream != 0

    def exchange():
        if rgb8:                                                      # tonemap this rank's tiles (cpu:714-716), gather 3 bytes per pixel
            ctx.tonemap_device(local.data_ptr(), rows.n_rows * W, local8.data_ptr(), stream)
            return tiling.gather_frame(local8, H, world, rank, gathered)
        return tiling.gather_frame(local, H, world, rank, gathered)

    def step():
        nonlocal frame
        ctx.render_device(p, rows, local.data_ptr(), stream)
        frame = exchange()

    # exact ray count of one frame (deterministic; outside the timed region)
    step()
    torch.cuda.synchronize()
    rays_local = torch.tensor([float(local[:rows.n_rows, :, 3].double().sum().item())], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(rays_local)
    rays_per_frame = int(rays_local.item())
    # traversal work of one frame from the counting instantiation of the kernel, rank 0
    counts = ctx.count_work(p) if rank == 0 else None

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        ctx.render_device(p, rows, local.data_ptr(), stream)
        ev[k][1].record()
        frame = exchange()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = statistics.mean(a.elapsed_time(b) for a, b in ev)
    tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = float(tmax[0]), float(tmax[1])

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = rays_per_frame / (elapsed / args.steps) / 1e6
        workload = f"cat_{W}x{H}_spp{args.spp}_b{args.bounces}" if args.scene == "cpu" else f"{args.scene}_{W}x{H}_spp{args.spp}_b{args.bounces}"
        res = {"metric": "Mrays/s, cat mesh 1920x1080 (ms/frame in ms_per_step)", "value": round(value, 2), "unit": "Mrays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "scene": "cpu_launcher.cpp walls + cat.obj (3954 tris, 2019-node array BVH)",
                          "num_rays": args.spp, "num_bounce": args.bounces, "depth_convention": "cpu_launcher (b+1 segments)",
                          "rays_per_frame": rays_per_frame, "tiling": f"{TILE_ROWS}-row tiles interleaved over {world} rank(s)"
                          + (f", RCCL gather of the {'RGB8' if rgb8 else 'float4'} tiles to rank 0 per frame" if world > 1 else ""),
                          "variant": ctx.stats()["variant"], "device": ctx.device_name,
                          "primary_Msamples_per_s": round(W * H * args.spp / (elapsed / args.steps) / 1e6, 1)}}
        if world == 1:
            # like-for-like with a host caller -- rt_render (kernels + the 16 B/pixel D2H copy over PCIe), untimed region
            t1 = time.perf_counter()
            for _ in range(3):
                ctx.render(p)
            res["config"]["host_frame_ms_incl_d2h"] = round((time.perf_counter() - t1) / 3 * 1e3, 3)
            pin = rt.PinnedArray((H, W, 4))                          # the same into a buffer from rt_host_alloc: the copy is one DMA
            ctx.render(p, out=pin.array)
            t1 = time.perf_counter()
            for _ in range(3):
                ctx.render(p, out=pin.array)
            res["config"]["host_frame_ms_incl_d2h_pinned"] = round((time.perf_counter() - t1) / 3 * 1e3, 3)
            pin.close()
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args, rays_per_frame)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if counts is not None:
            assert counts["rays"] == rays_per_frame, (counts, rays_per_frame)
            res["roofline"] = roofline(rt, ctx, args, p, rows, local, stream, counts, world, W, H, kernel_ms_max, workload)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
