#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path (BASELINE.json: Mrays/s + ms/frame, cat mesh
1920x1080, at 1/2/4/8 MI355X).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one frame of the workload: the cat scene of cpu_launcher.cpp (walls + cat mesh + array BVH),
1920x1080, num_rays=1, num_bounce=3 (CPU convention: 4 segments), rendered by the HIP kernel through the
C-ABI with the scene already resident in HBM.  With N > 1 the frame is split into interleaved 8-row tiles
(tile k -> rank k mod N), every rank renders its tiles, and ONE RCCL gather per frame brings the float4
tiles to the root, which de-interleaves them (all inside the timed region).  Total work is fixed => "strong".
A rank whose share is small (<= 1.3 Mpixel: every share of the 1080p frame) renders N consecutive frames of its
share as ONE launch chain (rt_render_device_batch) and, by default, ships them in one gather per batch; the line
then carries the single-frame latency beside the throughput (config.frame_latency_ms, config.batch, config.exchange).

Rank 0 prints one JSON line.  `value` = rays traced per second (1 ray = 1 Scene::intersect_all call:
primary, shadow or bounce segment; counted exactly by the kernel in the framebuffer's .w channel).
`roofline` prices the dominant kernel (wf_travq) against vector-instruction issue: the vector wave-instructions of one launch --
step counts of the counting instantiation in THIS run x the static per-step instruction counts of the production code object
(tools/static_counts.py, written by build()), each instruction weighted by its measured issue cost -- over that launch's
duration (HIP events on its stream), against 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (the guide's SIMD-32 issue
rate).  `frac` is that WEIGHTED fraction; the unweighted count (`frac_unweighted`) and the nominal HBM figure of SURVEY 8d (24 B/box
test + 16 B/node + 48 B/triangle test of a cache-resident scene, `nominal_hbm_frac`) stay beside it, and `roofline.kernels` prices
the second hot kernel (wf_advance) against the HBM peak.  `config.end_to_end` times the whole program (rt_launcher 8 3 at 512x512:
context, OBJ parse, BVH build, upload, render, D2H, PNG) as the reference times its own (cpu_launcher.cpp:660,721-723).  `cpu_baseline` is the CPU restatement of
cpu_launcher.cpp (oracle/, OpenMP schedule(dynamic,1) over rows like cpu:695) timed on this host.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6290.0   # what a streaming kernel reaches on this part (same guide: "~6.3 TB/s achievable")
VALU_PEAK_GINST = 1024 * 2.4 / 2   # vector wave-instructions per ns the chip can issue: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64
                                   # instruction (MI355X_MICROARCH.md: "issues each VALU instruction over 2 cycles"; tools/ubench/issue_table.hip
                                   # agrees for fma / mul / add with 4+ waves per SIMD; min / max / compares / cndmask / conversions take ~1.75x
                                   # that, transcendentals ~3.3x: the WEIGHTED count over this peak is `roofline.frac`)
ADV_LAYOUT_BYTES_PER_PATH = 122   # wf_advance, HBM bytes per path and launch of the record layout (profiles/round4/pmc_wf_advance.json: 252 MB per 2.07 M paths; rounds 2-3: 154)
TILE_ROWS = 8              # == raytracinggpu_amd.tiling.TILE_ROWS

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed frames of the same workload rendered before the W warm-up steps, worth about this much GPU time: after idling the "
                         "device needs 20-40 ms of load to reach its sustained clock (tools/frame_ramp.py: the first 20 frames after an idle "
                         "period run 1.13 -> 1.01 ms); 0 = none")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--scene", default="cpu", choices=["cpu", "spheres", "demo10"])
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--gather", default="auto", choices=["auto", "f32", "rgb8"],
                    help="N > 1: what the root gathers -- the float4 tiles (parity path) or the tonemapped RGB8 tiles (PNG path, 3 B/pixel).  auto (default): float4 unless the "
                         "float4 tiles of ONE peer would need more than --link-budget-gbs of its xGMI link at the frame rate the ranks' compute side reaches (measured before "
                         "the timed region), then RGB8")
    ap.add_argument("--link-budget-gbs", type=float, default=40.0,
                    help="--gather auto: GB/s one peer may push through its xGMI link to the root (a quarter of the link's ~153 GB/s peak: the gather is many small messages)")
    ap.add_argument("--root", default="0", choices=["0", "rotate"],
                    help="N > 1: the rank that assembles a frame -- always rank 0 (the reference copies every image to one host, optimized.cu:849-856) or frame k -> rank k mod N "
                         "(one gather per frame either way; rotate spreads the inbound traffic over every rank's links)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "frame", "batch"],
                    help="N > 1 with batches: one gather per FRAME, or one per BATCH (the tiles of a peer's `batch` frames travel as one message: an eighth of the collectives, the same "
                         "bytes; every frame is still assembled).  auto = batch whenever frames are batched and the exchange is torch.distributed's")
    ap.add_argument("--batch", type=int, default=0,
                    help="N > 1: frames of a small share rendered as ONE launch chain (rt_render_device_batch); 0 = auto (about a whole frame's worth: the number of ranks, for shares of at most 1.3 Mpixel), 1 = off")
    ap.add_argument("--transport", default="torch", choices=["torch", "capi"],
                    help="N > 1: the gather goes through torch.distributed (nccl = RCCL; default) or through the product's own C-ABI "
                         "(libraytrace_rccl.so: grouped ncclSend / ncclRecv, every tile received straight into its place in rank 0's frame)")
    ap.add_argument("--comm-plan", default="auto", choices=["auto", "tile", "coalesced"],
                    help="--transport capi: the exchange plan of rt_comm_gather_tiles (auto: coalesced -- one message per peer + one placement kernel -- "
                         "when a tile is smaller than 256 KiB, else one receive per tile straight into the frame)")
    ap.add_argument("--gather-only", action="store_true",
                    help="N > 1: time the EXCHANGE alone -- the tiles are rendered once, the timed steps repeat the gather of the same buffers (no render); "
                         "the line's value is null and config.gather_only holds ms per gather and GB/s into the root")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip config.end_to_end (three runs of the rt_launcher program, ~1 s each)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--large-steps", type=int, default=4, help="frames of the second timed point (cat 7680x4320, BASELINE config 5) in the same run; 0 = skip")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST: every rank renders on GPU 0 and the exchange runs over gloo (RCCL refuses several ranks on one device); rehearses the N-rank path on a one-GPU box")
    ap.add_argument("--renderer", default="hip", choices=["hip", "oracle"],
                    help="TEST: 'oracle' renders every rank's tiles with the CPU restatement over gloo (launcher / partition / gather plumbing without a GPU); its line carries value = null")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="whole frames kept in flight (0 = auto).  One GPU: 2 = frames alternate between two device buffers on ONE context and "
                         "stream with rt_ctx_set_pipelining (frame k+1's sub-frames follow frame k's directly; default), 1 = every frame joined before "
                         "the next starts.  Several ranks: see --batch")
    ap.add_argument("--dump-frame", default="", help="rank 0 saves the (gathered) float4 frame of the first step as .npy (tests)")
    ap.add_argument("--check-frame", action="store_true", help="rank 0 renders the whole frame alone as well and reports whether the gathered frame equals it bit for bit")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start the N ranks ourselves, BEFORE anything
    in this process touches the GPU (the reference's harness starts its workers itself too, benchmark.py:19-33), relay rank 0's
    JSON line and the exit code.  Nothing here imports torch.cuda or the library."""
    n = args.gpus
    if args.renderer == "hip" and not args.share_gpu:
        import torch                                                  # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible to this process; one rank per GPU is required "
                             f"(--share-gpu rehearses the {n}-rank path on one device over gloo)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif r.stdout:
        sys.stderr.write(r.stdout)
    sys.exit(r.returncode if r.returncode else (0 if lines else 1))


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


def end_to_end(threads):
    """SURVEY 8d / VERDICT round 3 item 1: the GPU-side counterpart of `reference_check.reference_program_s` -- the whole PROGRAM, timed by
    itself the way the reference times its own (cpu_launcher.cpp:660,721-723: first line of main to after the PNG is written):
    `rt_launcher 8 3` at the reference's hard-wired 512x512, in a directory that holds the cat as an OBJ.  Includes HIP start-up,
    context creation, OBJ parse, host BVH build, scene upload, the render, the device-to-host copy and the PNG encoder."""
    import tempfile
    import raytracinggpu_amd as rt
    exe = os.path.join(ROOT, "raytracinggpu_amd", "rt_launcher")
    if not os.path.exists(exe):
        return {"skipped": "raytracinggpu_amd/rt_launcher not built"}
    g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
    with tempfile.TemporaryDirectory() as d:
        od = os.path.join(d, "cadnav.com_model", "Models_F0202A090")
        os.makedirs(od)
        with open(os.path.join(od, "cat.obj"), "w") as f:
            for v in g["vertices"]:
                f.write("v %.9g %.9g %.9g 1 1 1\r\n" % tuple(float(x) for x in v))
            for t in g["tri_obj_order"]:
                f.write("f %d/1/1 %d/1/1 %d/1/1\r\n" % tuple(int(x) + 1 for x in t))
        times, kernel = [], []
        for _ in range(3):
            r = subprocess.run([exe, "8", "3"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
            if r.returncode != 0 or "Rendering time:" not in r.stdout:
                return {"skipped": f"rt_launcher failed ({r.returncode}): {r.stderr.strip()[-200:]}"}
            times.append(float(r.stdout.split("Rendering time:")[1].split()[0]))
            if "kernel " in r.stderr:
                kernel.append(float(r.stderr.split("kernel ")[1].split()[0]))
    return {"config": "rt_launcher 8 3: 512x512, num_rays=8, num_bounce=3, cat scene (the reference program's own size and arguments)",
            "program_s": round(statistics.median(times), 4), "program_s_runs": [round(t, 4) for t in times],
            "render_kernels_ms": round(statistics.median(kernel), 3) if kernel else None,
            "includes": "HIP start-up + context, OBJ parse, host BVH build, scene upload, render, D2H, PNG encode (the reference's timer spans the same: cpu_launcher.cpp:660,721-723)"}


def single_stream_launch_ms(rt, args, p, rows, local, stream):
    """Average duration of ONE launch of the dominant kernel when it owns the chip: a second context created under
    RT_PARTS=1 (knobs are read once per context) renders the same frames as one sub-frame on one stream; the library
    brackets the traversal launches with HIP events on the stream they run on (rt_stats_enable: production frames record none)."""
    old = os.environ.get("RT_PARTS")
    os.environ["RT_PARTS"] = "1"
    try:
        c1 = rt.Context(int(os.environ.get("LOCAL_RANK", "0")) if not args.share_gpu else 0)
    finally:
        if old is None:
            del os.environ["RT_PARTS"]
        else:
            os.environ["RT_PARTS"] = old
    build_scene(rt, c1, args.scene)
    c1.stats_enable(True)
    ms, launches, frame, adv = [], 0, [], []
    adv_info = None
    for k in range(8):
        c1.render_device(p, rows, local.data_ptr(), stream)
        st = c1.stats()
        if k >= 3 and st["trav_launches"] > 0:
            ms.append(st["trav_ms"] / st["trav_launches"]); launches = st["trav_launches"]; frame.append(st["kernel_ms"])
            if st.get("adv_launches", 0) > 0:
                adv.append(st["adv_ms"] / st["adv_launches"]); adv_info = (st["adv_launches"], st["adv_paths"])
    c1.close()
    single_stream_launch_ms.adv = (statistics.median(adv), adv_info[0], adv_info[1]) if adv else None
    return (statistics.median(ms), launches, statistics.median(frame)) if ms else (None, 0, None)


def static_counts():
    path = os.path.join(ROOT, "raytracinggpu_amd", "static_counts.json")
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return None


def qw_step_counts(rt, args, p):
    """Step counters of the 4-wide traversal kernel (the production kernel wherever its node format fits): a context created under RT_TRAVQ_QW_COUNT=1
    runs THAT kernel's counting instantiation in rt_count_work (the default counting run is the binary instantiation: the reference's own box / node
    counts, which the algorithmic-bytes figure and the parity tests need)."""
    old = os.environ.get("RT_TRAVQ_QW_COUNT")
    os.environ["RT_TRAVQ_QW_COUNT"] = "1"
    try:
        c2 = rt.Context(int(os.environ.get("LOCAL_RANK", "0")) if not args.share_gpu else 0)
    finally:
        if old is None:
            del os.environ["RT_TRAVQ_QW_COUNT"]
        else:
            os.environ["RT_TRAVQ_QW_COUNT"] = old
    build_scene(rt, c2, args.scene)
    out = c2.count_work(p, detail=True)
    c2.close()
    return out


def travq_instructions(counts, sc, section="wf_travq"):
    """Vector / scalar wave-instructions of the frame's wf_travq launches: the step counters of the counting instantiation (this run)
    x the static per-region counts of the production code object (tools/static_counts.py).  section "wf_travq_qw": the 4-wide kernel, whose
    leaf_push / leaf_push2 counters are the leaf / internal push blocks its BOX steps entered."""
    st, t = counts["steps"], sc[section]
    weights = (("loop_head", st["iterations"]), ("retire", st["refill_passes"]), ("round", st["refill_rounds"]),
               ("fetch", st["fetches"]), ("tri", st["tri_steps"]), ("tdiv", st["tdiv_blocks"]), ("box", st["box_steps"]),
               ("lpush", st["leaf_push_blocks"]), ("lpush2", st["leaf_push2_blocks"]))
    if section == "wf_travq_qw" and "lflag" in t:    # the 4-wide kernel counts the flagged-leaf check blocks its TRI steps entered where the float pairs count their literal box blocks
        weights += (("lflag", st["literal_box_fallbacks"]),)
    out = {k: int(sum(t[r][k] * n for r, n in weights)) for k in ("valu", "valu_weight", "salu")}
    # the step dispatch between the refill and the steps: scalar instructions every iteration, its vector ones only in front of a TRI step
    out["salu"] += int(t["dispatch"]["salu"] * st["iterations"])
    out["valu"] += int(t["dispatch"]["valu"] * st["tri_steps"]); out["valu_weight"] += int(t["dispatch"]["valu_weight"] * st["tri_steps"])
    out["by_region_valu"] = {r: int(t[r]["valu"] * n) for r, n in weights}
    return out


def roofline(rt, ctx, args, p, rows, local, stream, counts, world, W, H, kernel_ms_max, workload):
    """What binds the dominant kernel, measured in this run; the nominal HBM roofline of SURVEY 8d beside it.

    wf_travq is bound by vector-instruction issue (HBM 9 % of peak, scene cache resident).  achieved = vector
    wave-instructions of one launch / that launch's duration (HIP events by the library on the stream the kernel runs on, in a
    single-stream context: one launch owns the chip); peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction.  The instruction count is the
    counting instantiation's step counters of THIS run x the static per-step counts of the production code object; it is checked
    against rocprofv3's SQ_INSTS_VALU in profiles/ (tools/round_profile.sh).  `traffic` quotes the measured HBM bytes of the same
    launch from the committed PMC summary when that summary was taken from this code (library hash), else null."""
    ctx.stats_enable(True)                                            # one more frame, outside the timed region, with the traversal launches bracketed by events
    ctx.render_device(p, rows, local.data_ptr(), stream)
    st = ctx.stats()
    ctx.stats_enable(False)
    trav_bytes = 24 * counts["box_tests"] + 16 * counts["nodes"] + 48 * counts["tri_tests"]
    fb_bytes = 16 * W * H
    out = {"bound": "valu_issue", "peak": round(VALU_PEAK_GINST, 1), "unit": "Gwave-inst/s", "traffic": None,
           "frame_algorithmic_bytes": int(trav_bytes + fb_bytes), "frame_kernels_ms": round(kernel_ms_max, 4),
           "per_ray": {k: round(counts[k] / counts["rays"], 3) for k in ("box_tests", "nodes", "tri_tests")}}
    sc = static_counts()
    is_travq = st["variant"] == 8 and st["trav_launches"] > 0
    if not (is_travq and sc and "steps" in counts):
        # other variants: the nominal HBM roofline only
        k_ms = kernel_ms_max
        ach = (trav_bytes + fb_bytes) / world / (k_ms * 1e-3) / 1e9
        out.update({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "whole frame", "kernel_ms": round(k_ms, 4), "achieved": round(ach, 1),
                    "frac": round(ach / HBM_PEAK_GBS, 4), "frac_is": "nominal: SURVEY 8d algorithmic bytes of a cache-resident scene over the HBM peak; not a utilisation"})
        return out
    parts = max(st.get("parts", 1), 1)
    k_ms_conc = st["trav_ms"] / st["trav_launches"]
    single_ms, single_launches, single_frame = (None, 0, None)
    if world == 1:
        single_ms, single_launches, single_frame = single_stream_launch_ms(rt, args, p, rows, local, stream)
    mode = st.get("travq_mode", 0)                                    # 2: the 4-wide BOX step is the production kernel (its own counting instantiation, its own static counts)
    steps_of = counts
    if mode == 2 and "wf_travq_qw" in sc:
        steps_of = qw_step_counts(rt, args, p)
        assert steps_of["rays"] == counts["rays"], (steps_of["rays"], counts["rays"])
    ins = travq_instructions(steps_of, sc, "wf_travq_qw" if (mode == 2 and "wf_travq_qw" in sc) else "wf_travq")
    kname = "rtk::wf_travq<false, 64, false, false, true, true> (4-wide BOX step on 16-bit fixed-point nodes)" if mode == 2 else \
            "rtk::wf_travq<false, 64, false, false, true, false> (16-bit fixed-point sibling pairs)" if mode == 1 else "rtk::wf_travq<false, 64, false, false, false, false>"
    boxes_per_step = 256.0 if mode == 2 else 128.0
    if single_ms:
        launches, k_ms = single_launches, single_ms
        out.update({"kernel_ms": round(single_ms, 4), "launches_per_frame": single_launches, "concurrent_launches": 1, "single_stream_frame_ms": round(single_frame, 4)})
    else:
        launches, k_ms = st["trav_launches"] * parts * world, k_ms_conc
        out.update({"kernel_ms": round(k_ms, 4), "launches_per_frame": launches, "concurrent_launches": parts})
    out["kernel_ms_two_streams"] = round(k_ms_conc, 4)                # the default configuration: the twin launch of the other sub-frame shares the chip
    valu_launch = ins["valu"] / launches
    weighted_launch = ins["valu_weight"] / launches                   # in units of one full-rate instruction (2 cycles of one SIMD)
    ach = weighted_launch / (k_ms * 1e-3) / 1e9
    hbm_ach = trav_bytes / launches / (k_ms * 1e-3) / 1e9
    out.update({"kernel": kname, "achieved": round(ach, 1), "frac": round(ach / VALU_PEAK_GINST, 4),
                "frac_is": "issue-cost-weighted vector wave-instructions of one launch (step counters of this run x static per-step counts of the code object; "
                           "weight 1 = fma / mul / add / logic, 1.75-2 = min / max / compare / cndmask / conversion / binary64, 4 = rcp / sqrt) / launch duration, "
                           "over 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction = 1228.8 G/s; the rest of a wave's time is s_waitcnt (vector-memory and LDS "
                           "round trips of its dependent chain) and issue contention (profiles/round5/pmc_ab_wide_nodes.txt: with the 4-wide step 37 % of the wave-cycles "
                           "wait for memory and 18 % for an issue slot; with the sibling pairs 49 % and 11 %)",
                "achieved_unweighted": round(valu_launch / (k_ms * 1e-3) / 1e9, 1), "frac_unweighted": round(valu_launch / (k_ms * 1e-3) / 1e9 / VALU_PEAK_GINST, 4),
                "valu_wave_insts_per_launch": int(valu_launch), "valu_weighted_insts_per_launch": int(weighted_launch), "salu_wave_insts_per_launch": int(ins["salu"] / launches),
                "valu_by_region_per_frame": ins["by_region_valu"], "steps_per_frame": steps_of["steps"],
                "steps_are": "the production kernel's own counting instantiation" + (" (4-wide: box_steps = 64 quads of 4 boxes; leaf_push / leaf_push2 = leaf / internal push blocks entered); "
                             "per_ray and the byte figures stay the reference-equivalent counts of the binary instantiation" if mode == 2 else ""),
                "box_step_lane_occupancy": round((steps_of["box_tests"] - steps_of["rays"]) / (boxes_per_step * max(steps_of["steps"]["box_steps"], 1)), 4),   # the root-box test of every ray belongs to the uniform kernel
                "tri_step_lane_occupancy": round(steps_of["tri_tests"] / (128.0 * max(steps_of["steps"]["tri_steps"], 1)), 4),
                "literal_box_tests": counts["box_literal"], "literal_tri_tests": counts["tri_literal"],
                "algorithmic_bytes_per_launch": int(trav_bytes / launches), "nominal_hbm_GBps": round(hbm_ach, 1), "nominal_hbm_frac": round(hbm_ach / HBM_PEAK_GBS, 4),
                "nominal_hbm_frac_is": "SURVEY 8d: 24 B/box test + 16 B/node + 48 B/triangle test of a cache-resident scene over 8 TB/s; may exceed 1, NOT a utilisation"})
    # measured HBM traffic of the same launch: only from a PMC summary taken from THIS library
    # the committed PMC summary of THIS code: every profiles/roundN/summary.json is looked at, newest round first, and the one whose code_hash equals the hash of the sources
    # being run is quoted; if none matches, the newest one is named as stale and no traffic is quoted
    import glob
    import re as _re
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "summary.json")), key=lambda q: -int((_re.search(r"round(\d+)", q) or [0, 0])[1]))
    rnd, match, newest = None, None, None
    for spath in cands:
        try:
            summ = json.load(open(spath))
        except (OSError, ValueError):
            continue
        newest = newest or os.path.basename(os.path.dirname(spath))
        if summ.get("code_hash") == code_hash() and workload == "cat_1920x1080_spp1_b3":   # (the profile is of the headline workload)
            rnd, match = os.path.basename(os.path.dirname(spath)), summ
            break
    if match is not None:
        t = match["kernels"]["wf_travq"]
        out["traffic"] = int(t["hbm_read_bytes_per_launch"] + t["hbm_write_bytes_per_launch"])
        out["traffic_source"] = f"profiles/{rnd}/summary.json (rocprofv3 --pmc, RT_PARTS=1, same source hash)"
        out["pmc"] = {"valu_wave_insts_per_launch": t.get("valu_wave_insts_per_launch"), "salu_wave_insts_per_launch": t.get("salu_wave_insts_per_launch"),
                      "valu_lane_utilization": t.get("valu_lane_utilization"), "l2_hit_rate": t.get("l2_hit_rate"),
                      "rocprof_avg_us_single_stream": t.get("rocprof_avg_us_single_stream"), "rocprof_avg_us_two_streams": t.get("rocprof_avg_us_two_streams")}
        if t.get("valu_wave_insts_per_launch") and t.get("rocprof_avg_us_single_stream"):
            out["pmc"]["frac_pmc"] = round(t["valu_wave_insts_per_launch"] / (t["rocprof_avg_us_single_stream"] * 1e-6) / 1e9 / VALU_PEAK_GINST, 4)
            out["pmc"]["frac_pmc_is"] = "SQ_INSTS_VALU per launch / rocprofv3's average launch duration (one sub-frame at a time) over the same 1228.8 G wave-inst/s: the unweighted hardware count"
    else:
        rnd = newest or "round0"
        out["traffic_note"] = f"no profiles/round*/summary.json carries the hash of the sources being run (newest: {rnd}): profile stale, traffic not quoted"
    # both hot kernels, each against the roofline that bounds it
    kern = [{"kernel": "wf_travq", "bound": "valu_issue", "kernel_ms": out.get("kernel_ms"), "launches_per_frame": out.get("launches_per_frame"),
             "achieved": out.get("achieved"), "peak": out["peak"], "unit": out["unit"], "frac": out.get("frac"), "traffic": out.get("traffic")}]
    adv = getattr(single_stream_launch_ms, "adv", None)
    adv_ms, adv_launches, adv_paths, conc = (adv[0], adv[1], adv[2], 1) if adv else (st["adv_ms"] / st["adv_launches"] if st.get("adv_launches") else None, st.get("adv_launches", 0), st.get("adv_paths", 0), parts)
    if adv_ms:
        # bytes one launch moves: measured (PMC, same source hash) when the committed profile belongs to this code, else the record layout's
        # figure (the continuation ray's 32-byte record -- which also holds the path's flag word -- read back and written,
        # the shadow ray's record written only if it passed the mesh's root box, 8-16 B of traversal results, 5 B of shading terms per live path)
        layout_b = ADV_LAYOUT_BYTES_PER_PATH * adv_paths
        measured, valu_busy = None, None
        if out.get("traffic") is not None:
            try:
                a = json.load(open(os.path.join(ROOT, "profiles", rnd, "summary.json")))["kernels"]["wf_advance"]
                valu_busy = a.get("valu_pipe_busy_frac")
                measured = int((a["hbm_read_bytes_per_launch"] + a["hbm_write_bytes_per_launch"]) * adv_paths / max(a.get("paths_per_launch", adv_paths), 1))
            except (OSError, KeyError, ValueError):
                measured = None
        b = measured if measured is not None else layout_b
        gbs = b / (adv_ms * 1e-3) / 1e9
        kern.append({"kernel": "wf_advance", "bound": "hbm", "kernel_ms": round(adv_ms, 4), "launches_per_frame": adv_launches * conc, "concurrent_launches": conc,
                     "paths_per_launch": adv_paths, "bytes_per_launch": int(b), "bytes_per_path": round(b / max(adv_paths, 1), 1),
                     "bytes_are": "measured: rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE of this source hash" if measured is not None else "the record layout's figure (no PMC summary of this source hash)",
                     "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "frac_of_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 4), "traffic": measured,
                     **({"valu_pipe_busy_frac": valu_busy, "co_bound": "the kernel's vector pipes are this busy when it runs alone (PMC, same source hash): twelve correctly rounded "
                         "square roots and two binary64 islands per path; it is bound by both, and the byte diet of round 4 moved it towards the VALU side"} if valu_busy else {})})
    out["kernels"] = kern
    return out


def code_hash():
    """Hash of the kernel sources: ties a committed PMC summary to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracinggpu_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


class OracleRenderer:
    """TEST stand-in for the HIP path (--renderer oracle): the CPU restatement renders a rank's tiles.  Plumbing only."""
    def __init__(self, args):
        from oracle import oracle_py as orc
        import raytracinggpu_amd as rt
        mesh = None
        if args.scene == "cpu":
            verts, tris = rt.scenes.load_cat_arrays()
            mesh = orc.Mesh.from_arrays(verts, tris).build_bvh()
        self.sc = orc.Scene.preset(args.scene, mesh)

    def render(self, W, H, spp, b, rank, world, local):
        import torch
        part, _, _ = self.sc.render(W, H, spp, b, rows=(rank * TILE_ROWS, H), tile_rows=TILE_ROWS, tile_step=world, threads=2, want_rgb8=False)
        local[:part.shape[0]] = torch.from_numpy(part)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                             # does not return
    import torch
    import torch.distributed as dist
    import raytracinggpu_amd as rt
    from raytracinggpu_amd import tiling

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cpu_only = args.renderer == "oracle"
    gloo = cpu_only or args.share_gpu
    dev_index = 0 if args.share_gpu else local_rank
    if not cpu_only:
        if torch.cuda.device_count() <= dev_index:
            raise SystemExit(f"rank {rank}: GPU {dev_index} not visible ({torch.cuda.device_count()} device(s)); one rank per GPU is required")
        torch.cuda.set_device(dev_index)
    dev = torch.device("cpu") if cpu_only else torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    xdev = torch.device("cpu") if gloo else dev                       # where the exchange runs (gloo moves host tensors)
    comm = None
    if args.transport == "capi" and world > 1:
        if gloo or cpu_only:
            raise SystemExit("--transport capi is RCCL between GPUs: it has no gloo / CPU leg")
        from raytracinggpu_amd import _rccl
        box = [_rccl.unique_id() if rank == 0 else None]               # rank 0's communicator id reaches the others through the process group
        dist.broadcast_object_list(box, src=0)
        comm = _rccl.Comm(dev_index, rank, world, box[0])
        comm.set_plan(args.comm_plan)                                  # the same on every rank: both sides derive their message sizes from it

    W, H = args.width, args.height
    # Frames in flight.  One GPU renders a whole frame as two concurrent sub-frames (the library's default) and keeps two frames pipelined on one context and stream
    # (rt_ctx_set_pipelining).  A rank of an N-rank job owns a SHARE of every frame; a share of at most ~1.3 Mpixel no longer fills the chip with its eleven dependent launches,
    # so the bench renders `batch` consecutive frames of the share as ONE launch chain (rt_render_device_batch: about one whole frame's worth of paths per chain), batches
    # pipelined like frames, every frame still gathered and assembled on its own.  Measured on one MI355X for rank 0's share (tools/share_batch.py,
    # profiles/round6/share_batch.txt): 1/8 of 1080p 0.282 ms one frame at a time -> 0.109 ms per frame in batches of 8 (round 5: four contexts with four frames in flight,
    # 0.13-0.19 ms); 1/8 of 3840x2160 0.536 -> 0.489 ms; a share above ~1.3 Mpixel is fastest one frame at a time (1/4 of 3840x2160: 0.878 ms alone, 0.973 in batches).
    SMALL_SHARE_PX = 1.3e6
    ctx = None
    pools = {}                                                         # "full": [contexts with the default knobs], each with a stream of its own

    def pool(kind, n):
        if kind in pools and len(pools[kind][0]) < n:                 # a later point wants more contexts: build the pool again
            del pools[kind]
        if kind not in pools:
            cs, ts = [], []
            for _ in range(n):
                c = rt.Context(dev_index)
                build_scene(rt, c, args.scene)
                cs.append(c)
                # a non-default torch stream per context: its handle is non-NULL (NULL means "the context's own stream" in the C-ABI),
                # so the render kernels, torch's timing events and the gather are all ordered on ONE stream
                ts.append(torch.cuda.Stream(device=dev, priority=-1 if (len(ts) & 1) else 0))
            pools[kind] = (cs, ts)
        return pools[kind]

    if cpu_only:
        oren = OracleRenderer(args)
        stream = None
    else:
        ctx = pool("full", 1)[0][0]
        torch.cuda.set_stream(pool("full", 1)[1][0])
        stream = pool("full", 1)[1][0].cuda_stream
        assert stream != 0
    rgb8 = args.gather == "rgb8" and world > 1 and not cpu_only        # --gather auto: decided below, before the timed point is built
    rotate = args.root == "rotate" and world > 1

    class Lane:
        """One frame in flight: this rank's tile buffer and the buffers of its exchange."""
        def __init__(self, k, W, H, local=None, local8=None):
            self.k = k
            self.local = tiling.local_buffer(H, W, world, dev) if local is None else local
            self.local8 = (tiling.local_buffer(H, W, world, dev, rgb8=True) if local8 is None else local8) if rgb8 else None
            if local is not None:                                      # a frame of a batch whose exchange moves the whole batch: the batch owns the exchange buffers
                self.xlocal = self.gathered = self.frame = None
                return
            src = self.local8 if rgb8 else self.local
            self.xlocal = torch.empty(src.shape, dtype=src.dtype, device=xdev, pin_memory=not cpu_only) if (gloo and world > 1 and not cpu_only) else src
            is_root = rank == 0 or rotate                              # a rotating root: every rank assembles its share of the frames
            self.gathered = tiling.gather_buffer(self.xlocal, world) if (world > 1 and is_root and comm is None) else None
            self.frame = torch.empty((H, W) + tuple(src.shape[2:]), dtype=src.dtype, device=dev) if (comm is not None and is_root) else None

    class Point:
        """One timed workload: its parameters, this rank's tiles, and one Lane per frame in flight.
        batch > 1 (small shares of an N-rank job): `batch` consecutive frames are rendered as ONE launch chain (rt_render_device_batch: frames x pixels are the chain's
        items, so a 1/8 share fills the chip the way the whole frame does on one GPU) into `batch` lanes, two batches in flight on two contexts; every frame is still
        exchanged and assembled on its own, one gather per frame."""
        def __init__(self, W, H, lanes=0, batch=None):
            self.W, self.H = W, H
            self.p = rt.make_params(W, H, args.spp, args.bounces, variant=args.variant, **rt.scenes.CPU_LAUNCHER)
            self.rows, self.idx = rt.interleaved_rows(H, TILE_ROWS, rank, world)
            full_share = tiling.tiles_per_rank(H, world) * TILE_ROWS * W   # (the same on every rank: all ranks must batch alike, the steps hold a collective)
            if batch is None:
                batch = args.batch if args.batch > 0 else (min(16, max(2, round(W * H / max(full_share, 1)))) if (world > 1 and full_share <= SMALL_SHARE_PX) else 1)
            self.batch = 1 if (cpu_only or world == 1 or args.spp != 1 or lanes == 1) else max(1, min(16, batch))
            want = lanes if lanes > 0 else (args.frames_in_flight if (args.frames_in_flight > 0 and world == 1) else 2 if world == 1 else 1)
            if self.batch > 1:
                want = 2 * self.batch
            self.n_lanes = 1 if cpu_only else want
            # one GPU renders the whole frame: the frames in flight share ONE context and stream (two buffers, rt_ctx_set_pipelining);
            # a rank with a small share keeps one context per frame in flight
            self.pipelined = (not cpu_only) and world == 1 and self.n_lanes > 1 and hasattr(pool("full", 1)[0][0], "set_pipelining")
            if not cpu_only:
                if world == 1:
                    full = pool("full", 1)
                    self.ctxs, self.tstreams = [full[0][0]] * self.n_lanes, [full[1][0]] * self.n_lanes
                    if hasattr(full[0][0], "set_pipelining"):
                        full[0][0].set_pipelining(self.pipelined)
                elif self.batch > 1:                                    # ONE context with the library's defaults (two sub-frames, each taking half of a batch's frames), batches alternating
                    full = pool("full", 1)                              # between two sets of buffers on one stream with rt_ctx_set_pipelining: batch b + 1's chains follow batch b's directly
                    self.ctxs, self.tstreams = [full[0][0]] * self.n_lanes, [full[1][0]] * self.n_lanes
                    full[0][0].set_pipelining(True)
                else:                                                   # a big share: one frame at a time, two sub-frames
                    self.ctxs, self.tstreams = pool("full", 1)
                    self.ctxs[0].set_pipelining(False)
            # one gather per batch: the frames of a batch live in ONE tensor [batch, rows, W, C], so a peer's tiles of all of them are one contiguous message
            self.bx = self.batch > 1 and comm is None and args.exchange != "frame"
            if self.bx:
                K, rows_pad = self.batch, tiling.tiles_per_rank(H, world) * TILE_ROWS
                self.bbuf = [torch.zeros((K, rows_pad, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
                self.bbuf8 = [torch.zeros((K, rows_pad, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)] if rgb8 else None
                src = self.bbuf8 if rgb8 else self.bbuf
                self.bhost = [torch.empty(src[0].shape, dtype=src[0].dtype, device=xdev, pin_memory=True) for _ in range(2)] if gloo else None
                is_root = rank == 0 or rotate
                self.bgath = [torch.empty((world,) + tuple(src[0].shape), dtype=src[0].dtype, device=xdev) for _ in range(2)] if is_root else None
                self.lanes = [Lane(k, W, H, local=self.bbuf[k // K][k % K], local8=self.bbuf8[k // K][k % K] if rgb8 else None) for k in range(self.n_lanes)]
                self.nb = 0                                           # batches exchanged since the point was built (decides the rotating root of a batch)
            else:
                self.lanes = [Lane(k, W, H) for k in range(self.n_lanes)]
            self.local = self.lanes[0].local
            self.frame = None
            self.n = 0                                                # frames stepped since the point was built (decides the rotating root: the same count on every rank)
            self.pos, self.total = 0, 1                               # position in the current run of steps and its length (a batch never renders frames the run does not step)

        def begin(self, total):
            self.pos, self.total = 0, max(int(total), 1)

        def render(self, ln):
            if cpu_only:
                oren.render(self.W, self.H, args.spp, args.bounces, rank, world, ln.local)
            else:
                self.ctxs[ln.k].render_device(self.p, self.rows, ln.local.data_ptr(), self.tstreams[ln.k].cuda_stream)

        def render_batch(self, first_lane, count):
            """`count` frames of the run into lanes first_lane .. first_lane + count - 1 as one launch chain.  Every frame of the metric is the same frame (as on one GPU, where
            the bench renders one frame K times): same camera, same seed; the entry point takes a camera and a seed per frame (tests/test_gpu_parity.py renders a dolly with it)."""
            lns = self.lanes[first_lane:first_lane + count]
            frames = [(ln.local.data_ptr(), (0.0, 0.0, 55.0), None, self.p.seed) for ln in lns]
            self.ctxs[first_lane].render_device_batch(self.p, self.rows, frames, self.tstreams[first_lane].cuda_stream)

        def exchange(self, ln, root):
            src = ln.local
            if rgb8:                                                  # tonemap this rank's tiles (cpu:714-716), gather 3 bytes per pixel
                self.ctxs[ln.k].tonemap_device(ln.local.data_ptr(), self.rows.n_rows * self.W, ln.local8.data_ptr(), self.tstreams[ln.k].cuda_stream)
                src = ln.local8
            if comm is not None:                                      # the product's own transport: tiles land in place, nothing to assemble
                comm.gather_tiles(src.data_ptr(), self.W, self.H, src.shape[2] * src.element_size(), ln.frame.data_ptr() if rank == root else None,
                                  tile_rows=TILE_ROWS, root=root, stream=self.tstreams[ln.k].cuda_stream)
                self.frame = ln.frame if rank == root else None
                return
            if ln.xlocal is not src:                                  # --share-gpu: the exchange runs over gloo on host tensors
                ln.xlocal.copy_(src, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            self.frame = tiling.gather_frame(ln.xlocal, self.H, world, rank, ln.gathered, root=root)

        def exchange_batch(self, b, count):
            """ONE gather for the `count` frames of batch buffer b: every peer sends the tiles of all of them as one message; the root assembles each frame."""
            root = tiling.root_of(self.nb, world, args.root)
            self.nb += 1
            k0 = b * self.batch
            if rgb8:
                for j in range(count):
                    ln = self.lanes[k0 + j]
                    self.ctxs[ln.k].tonemap_device(ln.local.data_ptr(), self.rows.n_rows * self.W, ln.local8.data_ptr(), self.tstreams[ln.k].cuda_stream)
            src = (self.bbuf8 if rgb8 else self.bbuf)[b][:count]      # contiguous: the leading frames of the batch tensor
            if self.bhost is not None:                                # --share-gpu: the exchange runs over gloo on host tensors
                host = self.bhost[b][:count]
                host.copy_(src, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                src = host
            g = self.bgath[b][:, :count] if rank == root else None
            dist.gather(src, [g[r] for r in range(world)] if rank == root else None, dst=root)
            self.frame = None
            if rank == root:
                for j in range(count):
                    self.frame = tiling.assemble(g[:, j], self.H)
                if count > 1:
                    self.frame = tiling.assemble(g[:, 0], self.H)     # (--dump-frame / --check-frame look at the batch's first frame)

        def step(self, ev=None, mode="both"):
            """mode: "both" = the metric's step; "render" / "exchange" = one side only (the compute-side / exchange-side accounting after the timed region)."""
            root = tiling.root_of(self.n, world, args.root)
            if self.batch > 1:
                b, j = divmod(self.pos, self.batch)
                ln = self.lanes[(b % 2) * self.batch + j]
            else:
                j, ln = 0, self.lanes[self.n % self.n_lanes]
            self.n += 1
            self.pos += 1
            if cpu_only:
                if mode != "exchange":
                    self.render(ln)
                if mode != "render":
                    self.exchange(ln, root)
                return
            with torch.cuda.stream(self.tstreams[ln.k]):
                if ev:
                    ev[0].record()
                if mode != "exchange" and not (args.gather_only and self.n > self.n_lanes):  # --gather-only: every lane's tiles are rendered once, then only exchanged
                    if self.batch > 1:
                        if j == 0:                                    # the batch's first step issues the chain for all of its frames; the others find their frame rendered
                            self.render_batch(ln.k, min(self.batch, self.total - (self.pos - 1)))
                    else:
                        self.render(ln)
                if ev:
                    ev[1].record()
                if mode != "render":
                    if self.bx:                                       # the batch's LAST step moves the whole batch
                        b_, j_ = divmod(self.pos - 1, self.batch)
                        cnt = min(self.batch, self.total - b_ * self.batch)
                        if j_ == cnt - 1:
                            self.exchange_batch(b_ % 2, cnt)
                    else:
                        self.exchange(ln, root)

    def sync():
        if not cpu_only:
            torch.cuda.synchronize()

    def timed(pt, steps, warmup, mode="both"):
        """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        if not cpu_only and args.prewarm_ms > 0:                        # same count on every rank (the steps hold a collective)
            est_ms = max(pt.W * pt.H / 2.0e6 / world, 0.2)
            n_pre = max(1, min(1000, int(args.prewarm_ms / est_ms)))
            pt.begin(n_pre)
            for _ in range(n_pre):
                pt.step()
        pt.begin(warmup)
        for _ in range(warmup):
            pt.step(mode=mode)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if not cpu_only else []
        if world > 1:
            dist.barrier()
        sync()
        pt.begin(steps)
        t0 = time.perf_counter()
        for k in range(steps):
            pt.step(ev[k] if ev else None, mode)
        sync()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = statistics.mean(a.elapsed_time(b) for a, b in ev) if ev else 0.0
        tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=xdev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax[0]), float(tmax[1])

    def rays_of(pt):
        r = torch.tensor([float(pt.local[:pt.rows.n_rows, :, 3].double().sum().item())], dtype=torch.float64, device=xdev)
        if world > 1:
            dist.all_reduce(r)
        return int(r.item())

    gather_choice = None
    if args.gather == "auto" and world > 1 and not cpu_only:
        # the exchange budget: every peer pushes its float4 tiles through ONE xGMI link into the root; at the frame rate the ranks' compute side
        # reaches that is  W * H * 16 / N bytes x frames/s  per link.  Measured here, before the timed point exists: a few frames with the exchange skipped.
        probe = Point(W, H)
        e_p, _ = timed(probe, 8, 2, "render")
        fps = 8 / max(e_p, 1e-9)
        per_link = W * H * 16 / world * fps / 1e9
        rgb8 = per_link > args.link_budget_gbs
        gather_choice = {"float4_GBps_per_link_at_compute_rate": round(per_link, 2), "budget_GBps_per_link": args.link_budget_gbs, "chosen": "rgb8" if rgb8 else "f32",
                         "compute_side_frames_per_s": round(fps, 1)}
        del probe
    main_pt = Point(W, H)
    # exact ray count of one frame (deterministic; outside the timed region)
    main_pt.begin(1)
    main_pt.step()
    sync()
    rays_per_frame = rays_of(main_pt)
    # traversal work of one frame from the counting instantiation of the kernel, rank 0
    counts = ctx.count_work(main_pt.p, detail=True) if (rank == 0 and not cpu_only) else None
    if args.dump_frame and rank == 0:
        np.save(args.dump_frame, (main_pt.frame if world > 1 else main_pt.local[:H]).cpu().numpy())
    frame_ok = None
    if args.check_frame and rank == 0 and not cpu_only and not rgb8:
        full = ctx.render(main_pt.p)
        got = main_pt.frame.cpu().numpy() if world > 1 else main_pt.local[:H].cpu().numpy()
        frame_ok = bool((got.view(np.uint32) == full.view(np.uint32)).all())

    elapsed, kernel_ms_max = timed(main_pt, args.steps, args.warmup)
    sides = None
    if world > 1 and not args.gather_only:
        # the same steps once more with one side only, outside the timed region (same lanes, same buffers; every rank takes part: the exchange is collective):
        # what the ranks' own shares cost (the compute side of the strong scaling) and what the gather costs when nothing is rendered beside it
        pw, args.prewarm_ms = args.prewarm_ms, 0
        e_r, _ = timed(main_pt, args.steps, 2, "render")
        e_x, _ = timed(main_pt, args.steps, 2, "exchange")
        args.prewarm_ms = pw
        sides = {"compute_ms_per_step": round(1e3 * e_r / args.steps, 4), "exchange_ms_per_step": round(1e3 * e_x / args.steps, 4),
                 "is": "the timed steps again with the exchange skipped / with the render skipped (max over ranks, same frames in flight): the two sides of ms_per_step, which overlaps them"}

    latency_ms = None
    if world > 1 and not args.gather_only and not cpu_only:
        # what the throughput figure does not say: ONE frame alone on this rank's share (rendered, gathered, assembled, then the next) -- the latency of a frame.  With
        # frames batched or in flight side by side `ms_per_step` is the rate frames COMPLETE at; a single frame of a 1/8 share cannot fill the chip and takes longer than 1/8
        lat_pt = Point(W, H, lanes=1)
        pw, args.prewarm_ms = args.prewarm_ms, 0
        e_l, _ = timed(lat_pt, args.steps, 2)
        args.prewarm_ms = pw
        latency_ms = round(1e3 * e_l / args.steps, 4)
        del lat_pt

    large = None
    if args.large_steps > 0 and args.scene == "cpu" and not cpu_only and (W, H) == (1920, 1080):
        # BASELINE config 5's size in the same run (the >= 7x tile-scaling claim is feasible there; `value` stays the 1080p metric)
        lp = Point(7680, 4320)
        lp.step()
        sync()
        lrays = rays_of(lp)
        lel, lk = timed(lp, args.large_steps, 1)
        large = {"workload": f"cat_7680x4320_spp{args.spp}_b{args.bounces}", "steps": args.large_steps, "ms_per_step": round(1e3 * lel / args.large_steps, 4),
                 "value": round(lrays / (lel / args.large_steps) / 1e6, 2), "unit": "Mrays/s", "rays_per_frame": lrays, "kernels_ms_max_over_ranks": round(lk, 4),
                 "frames_in_flight": lp.n_lanes}
        del lp
    one_in_flight_ms = None
    if world == 1 and not cpu_only and main_pt.n_lanes > 1:           # the same frames with every frame joined before the next one starts
        p1 = Point(W, H, lanes=1)
        e1, _ = timed(p1, args.steps, 2)
        one_in_flight_ms = round(1e3 * e1 / args.steps, 4)
        del p1
        main_pt.ctxs[0].set_pipelining(False)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = rays_per_frame / (elapsed / args.steps) / 1e6
        workload = f"cat_{W}x{H}_spp{args.spp}_b{args.bounces}" if args.scene == "cpu" else f"{args.scene}_{W}x{H}_spp{args.spp}_b{args.bounces}"
        backend = "gloo (test: ranks share GPU 0)" if args.share_gpu else "gloo (test: CPU stand-in renderer)" if cpu_only else "RCCL through libraytrace_rccl.so, tiles received in place" if comm is not None else "nccl (RCCL)"
        res = {"metric": "Mrays/s, cat mesh 1920x1080 (ms/frame in ms_per_step)", "value": None if cpu_only else round(value, 2), "unit": "Mrays/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "scene": "cpu_launcher.cpp walls + cat.obj (3954 tris, 2019-node array BVH)",
                          "num_rays": args.spp, "num_bounce": args.bounces, "depth_convention": "cpu_launcher (b+1 segments)",
                          "rays_per_frame": rays_per_frame, "ranks": dist.get_world_size() if world > 1 else 1,
                          "frames_in_flight": main_pt.n_lanes, "prewarm_ms": args.prewarm_ms,
                          **({"frames_in_flight_is": "frames alternate between two device buffers on one context and one stream; each of frame k+1's two sub-frames "
                                                     "follows the same sub-frame of frame k without a join in between (rt_ctx_set_pipelining); every frame is rendered in full",
                              "ms_per_step_one_frame_in_flight": one_in_flight_ms} if one_in_flight_ms is not None else {}),
                          "tiling": f"{TILE_ROWS}-row tiles interleaved over {world} rank(s)"
                          + (f", one gather ({backend}) of the {'RGB8' if rgb8 else 'float4'} tiles to {'a rotating root' if rotate else 'rank 0'} per "
                             f"{'batch of ' + str(main_pt.batch) + ' frames' if getattr(main_pt, 'bx', False) else 'frame'}" if world > 1 else ""),
                          "variant": ctx.stats()["variant"] if ctx else None, "device": ctx.device_name if ctx else "cpu (oracle stand-in: not a measurement)",
                          "primary_Msamples_per_s": round(W * H * args.spp / (elapsed / args.steps) / 1e6, 1),
                          "shadow_rays": ("traced to the end (RT_TRAVQ_ANYHIT=0)" if os.environ.get("RT_TRAVQ_ANYHIT", "1") == "0" else
                                          "any-hit: a shadow ray stops at the first accepted triangle that certainly lies before the light (cpu:615 is monotone in the nearest hit's t: "
                                          "and a shadow ray whose answer cannot reach the pixel -- a sphere shades it already, or the segment's direct term is +0 either way -- is not traced through the mesh; frames bit-identical to RT_TRAVQ_ANYHIT=0, which costs +4.5 %: profiles/round6/ab_anyhit.txt); every ray is still one intersect_all call of the count")}}
        if frame_ok is not None:
            res["config"]["frame_equals_single_device_frame"] = frame_ok
        if comm is not None:
            res["config"]["comm_plan"] = comm.last_plan
        if sides is not None:
            res["config"]["sides"] = sides
        if world > 1:
            res["config"]["root"] = "frame k is assembled on rank k mod N (one gather per frame)" if rotate else "rank 0 assembles every frame"
            res["config"]["gather"] = "rgb8" if rgb8 else "f32"
            if gather_choice is not None:
                res["config"]["gather_auto"] = gather_choice
            if main_pt.batch > 1:
                res["config"]["batch"] = main_pt.batch
                res["config"]["exchange"] = ("one gather per BATCH: a peer's tiles of the batch's frames travel as one message, the root assembles every frame" if main_pt.bx
                                             else "one gather per frame")
                res["config"]["batch_is"] = (f"{main_pt.batch} consecutive frames of this rank's share are rendered as ONE launch chain (rt_render_device_batch), two batches in flight on one context and two "
                                             "sets of buffers (rt_ctx_set_pipelining); every frame is assembled on its own.  ms_per_step is a THROUGHPUT figure: the frames of a batch finish together")
            if latency_ms is not None:
                res["config"]["frame_latency_ms"] = latency_ms
                res["config"]["frame_latency_is"] = "one frame alone on the ranks' shares: rendered, gathered and assembled before the next starts (max over ranks); the other side of the batched / in-flight throughput"
        if args.gather_only and world > 1:
            px_bytes = 3 if rgb8 else 16
            moved = W * H * px_bytes * (world - 1) / world               # bytes that cross the fabric into the root per gather
            res["value"] = None
            res["config"]["gather_only"] = {"ms_per_gather": round(ms_per_step, 4), "bytes_into_root": int(moved), "GBps_into_root": round(moved / (ms_per_step * 1e-3) / 1e9, 2),
                                            "note": "render skipped inside the timed region on purpose: this line measures the exchange, not the metric"}
        if large:
            res["config"]["large"] = large
        if world == 1 and not cpu_only:
            res["config"].update(host_frame_timings(rt, ctx, main_pt.p, W, H))
            if args.scene == "cpu" and not args.no_end_to_end:
                try:
                    res["config"]["end_to_end"] = end_to_end(args.cpu_threads or host_cores())
                except Exception as e:  # a timing extra: never takes the line down
                    res["config"]["end_to_end"] = {"skipped": str(e)}
        if world == 1 and not args.no_cpu_baseline and not cpu_only:
            try:
                res["cpu_baseline"] = cpu_baseline(args, rays_per_frame)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if counts is not None:
            assert counts["rays"] == rays_per_frame, (counts, rays_per_frame)
            res["roofline"] = roofline(rt, ctx, args, main_pt.p, main_pt.rows, main_pt.local, stream, counts, world, W, H, kernel_ms_max, workload)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_frame_timings(rt, ctx, p, W, H):
    """what a host caller sees (kernels + the device-to-host copy over PCIe), outside the timed region: rt_render into
    pageable and pinned memory, and the pipelined forms (rt_render_async / rt_wait: frame k's copy runs beside frame k+1's
    kernels) for the float4 frame and for the 8-bit image a PNG writer needs."""
    out = {}
    t1 = time.perf_counter()
    for _ in range(3):
        ctx.render(p)
    out["host_frame_ms_incl_d2h"] = round((time.perf_counter() - t1) / 3 * 1e3, 3)
    pin = rt.PinnedArray((H, W, 4))                                  # the same into a buffer from rt_host_alloc: the copy is one DMA
    ctx.render(p, out=pin.array)
    t1 = time.perf_counter()
    for _ in range(3):
        ctx.render(p, out=pin.array)
    out["host_frame_ms_incl_d2h_pinned"] = round((time.perf_counter() - t1) / 3 * 1e3, 3)
    if hasattr(ctx, "render_async"):
        n = 12
        pins = [pin, rt.PinnedArray((H, W, 4))]
        for rgb8, key in ((False, "host_frame_ms_pipelined_pinned"), (True, "host_frame_ms_pipelined_rgb8")):
            bufs = [rt.PinnedArray((H, W, 3), dtype=np.uint8) for _ in range(2)] if rgb8 else pins
            for k in range(2):                                        # warm both slots
                ctx.render_async(p, bufs[k].array, slot=k, rgb8=rgb8)
                ctx.wait(k)
            t1 = time.perf_counter()
            ctx.render_async(p, bufs[0].array, slot=0, rgb8=rgb8)
            for k in range(1, n):
                ctx.render_async(p, bufs[k & 1].array, slot=k & 1, rgb8=rgb8)   # frame k starts ...
                ctx.wait((k - 1) & 1)                                           # ... while frame k-1's copy completes
            ctx.wait((n - 1) & 1)
            out[key] = round((time.perf_counter() - t1) / n * 1e3, 3)
            if rgb8:
                for b in bufs:
                    b.close()
        pins[1].close()
    pin.close()
    return out


if __name__ == "__main__":
    main()
