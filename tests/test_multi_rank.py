"""N-rank row-tile path: world_size-2/3 torch.distributed runs.

CPU (gloo): the partition / gather / reassembly code bench.py uses, with the CPU oracle standing in for the
renderer of each rank's tiles -- the gathered frame must equal the full frame bit for bit.
GPU (-m gpu): the same with the HIP render path on cuda:0 in every rank (<= 3 ranks on the one card), gloo for
the exchange; RCCL itself needs one GPU per rank and runs in bench.py on the 8-GPU node.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, W, H, spp, b, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracinggpu_amd as rt
    from raytracinggpu_amd import tiling
    g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
    rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, rank, world)
    local = tiling.local_buffer(H, W, world, "cpu")
    if mode == "oracle":
        from oracle import oracle_py as orc
        mesh = orc.Mesh.from_arrays(g["vertices"], g["tri_obj_order"]).build_bvh()
        sc = orc.Scene.preset("cpu", mesh)
        if rows.n_rows:
            part, _, _ = sc.render(W, H, spp, b, rows=(rank * tiling.TILE_ROWS, H), tile_rows=tiling.TILE_ROWS, tile_step=world,
                                   threads=2, want_rgb8=False)
            assert part.shape[0] == rows.n_rows
            local[:rows.n_rows] = torch.from_numpy(part)
    else:
        from raytracinggpu_amd import hostlib
        ctx = rt.Context(0)
        ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(g["vertices"], g["tri_obj_order"], object_slot=6))
        dev = torch.zeros(local.shape, dtype=torch.float32, device="cuda:0")
        ctx.render_device(rt.make_params(W, H, spp, b, **rt.scenes.CPU_LAUNCHER), rows, dev.data_ptr())
        ctx.synchronize()
        local.copy_(dev.cpu())
        ctx.close()
    frame = tiling.gather_frame(local, H, world, rank)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_tiles_equal_full_frame_gloo_cpu(oracle, oracle_cat, tmp_path, world):
    W, H, spp, b = 96, 50, 1, 2          # H is not a multiple of the tile height: padding tiles in play
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), "oracle", W, H, spp, b, out), nprocs=world, join=True)
    got = np.load(out)
    exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(W, H, spp, b, want_rgb8=False)
    assert got.shape == exp.shape
    np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))


def _worker_rotating_root(rank, world, port, W, H, b, out_dir):
    """frames 0 .. world: frame k is gathered to and assembled on rank k mod world (tiling.root_of), each frame with its own seed so that a frame landing on the wrong rank shows"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracinggpu_amd as rt
    from raytracinggpu_amd import tiling
    from oracle import oracle_py as orc
    g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
    sc = orc.Scene.preset("cpu", orc.Mesh.from_arrays(g["vertices"], g["tri_obj_order"]).build_bvh())
    rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, rank, world)
    for k in range(world + 1):
        local = tiling.local_buffer(H, W, world, "cpu")
        if rows.n_rows:
            part, _, _ = sc.render(W, H, 1, b, rows=(rank * tiling.TILE_ROWS, H), tile_rows=tiling.TILE_ROWS, tile_step=world, threads=2, seed=100 + k, want_rgb8=False)
            local[:rows.n_rows] = torch.from_numpy(part)
        root = tiling.root_of(k, world, "rotate")
        frame = tiling.gather_frame(local, H, world, rank, root=root)
        assert (frame is not None) == (rank == root)
        if frame is not None:
            np.save(os.path.join(out_dir, f"frame{k}_rank{rank}.npy"), frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rotating_root_assembles_frame_k_on_rank_k_mod_world_gloo_cpu(oracle, oracle_cat, tmp_path, world):
    """bench.py --root rotate / tiling.root_of: one gather per frame, the root moving over the ranks frame by frame (the inbound traffic of a stream of frames is spread over
    every rank's links instead of rank 0's).  Every frame, wherever it was assembled, is bit for bit the full frame."""
    W, H, b = 64, 42, 1
    mp.spawn(_worker_rotating_root, args=(world, _free_port(), W, H, b, str(tmp_path)), nprocs=world, join=True)
    from raytracinggpu_amd import tiling
    assert [tiling.root_of(k, world, "rotate") for k in range(world + 1)] == [k % world for k in range(world + 1)] and tiling.root_of(5, world, "0") == 0
    for k in range(world + 1):
        exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(W, H, 1, b, seed=100 + k, want_rgb8=False)
        got = np.load(tmp_path / f"frame{k}_rank{k % world}.npy")
        np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
        assert not (tmp_path / f"frame{k}_rank{(k + 1) % world}.npy").exists()


def test_bench_rotating_root_and_batch_arguments_gloo_cpu(oracle, oracle_cat, tmp_path):
    """`bench.py --gpus 2 --root rotate` over gloo with the CPU stand-in: three steps, three roots (0, 1, 0), the line says which policy ran; frame 0 (root 0) is dumped and equal."""
    W, H, b = 96, 50, 1
    out = str(tmp_path / "frame.npy")
    r, line = _bench("--gpus", "2", "--renderer", "oracle", "--root", "rotate", "--width", str(W), "--height", str(H), "--bounces", str(b), "--steps", "3", "--warmup", "0",
                     "--large-steps", "0", "--dump-frame", out)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["steps"] == 3 and "rank k mod N" in line["config"]["root"] and line["config"]["gather"] == "f32" and line["value"] is None
    exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(W, H, 1, b, want_rgb8=False)
    np.testing.assert_array_equal(np.load(out).view(np.uint32), exp.view(np.uint32))


def _worker_rgb8(rank, world, port, W, H, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracinggpu_amd as rt
    from raytracinggpu_amd import tiling
    from oracle import oracle_py as orc
    g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
    rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, rank, world)
    local8 = tiling.local_buffer(H, W, world, "cpu", rgb8=True)
    mesh = orc.Mesh.from_arrays(g["vertices"], g["tri_obj_order"]).build_bvh()
    if rows.n_rows:
        _, part8, _ = orc.Scene.preset("cpu", mesh).render(W, H, 1, 0, rows=(rank * tiling.TILE_ROWS, H), tile_rows=tiling.TILE_ROWS, tile_step=world, threads=2)
        local8[:rows.n_rows] = torch.from_numpy(part8)
    frame = tiling.gather_frame(local8, H, world, rank)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_rgb8_gather_equals_full_image_gloo_cpu(oracle, oracle_cat, tmp_path):
    """The 8-bit exchange of bench.py --gather rgb8 (3 bytes per pixel): gathered tiles == the full tonemapped image."""
    W, H = 96, 50
    out = str(tmp_path / "frame8.npy")
    mp.spawn(_worker_rgb8, args=(2, _free_port(), W, H, out), nprocs=2, join=True)
    _, exp8, _ = oracle.Scene.preset("cpu", oracle_cat).render(W, H, 1, 0)
    np.testing.assert_array_equal(np.load(out), exp8)


def _bench(*args, timeout=600):
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")})
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_starts_its_own_ranks_gloo_cpu(oracle, oracle_cat, tmp_path):
    """`python bench.py --gpus 2` (the shape of the driver's command, no torch.distributed environment) starts its two ranks itself,
    relays rank 0's ONE JSON line and the exit code.  Here with the CPU stand-in renderer over gloo: launcher, partition, gather and
    reassembly are the ones the GPU run uses; the gathered frame equals the full oracle frame bit for bit and the line refuses to
    carry a throughput (value null)."""
    W, H, b = 96, 50, 2
    out = str(tmp_path / "frame.npy")
    r, line = _bench("--gpus", "2", "--renderer", "oracle", "--width", str(W), "--height", str(H), "--bounces", str(b), "--steps", "1", "--warmup", "0",
                     "--large-steps", "0", "--dump-frame", out)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["value"] is None and line["steps"] == 1
    exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(W, H, 1, b, want_rgb8=False)
    np.testing.assert_array_equal(np.load(out).view(np.uint32), exp.view(np.uint32))
    assert line["config"]["rays_per_frame"] == int(exp[..., 3].sum())


def test_bench_refuses_more_ranks_than_gpus_with_a_clear_message():
    if torch.cuda.device_count() >= 16:
        pytest.skip("needs fewer than 16 GPUs")
    r, line = _bench("--gpus", "16", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and line is None
    assert "GPU(s) visible" in r.stderr and "--share-gpu" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_the_one_gpu_frame_equals_single_device_frame(tmp_path):
    """--share-gpu: both ranks render on GPU 0 through the HIP path, gloo moves the tiles; the line reports the ranks the process group
    saw and that the gathered frame is bitwise the frame one context renders alone."""
    r, line = _bench("--gpus", "2", "--share-gpu", "--check-frame", "--width", "640", "--height", "356", "--steps", "2", "--warmup", "1", "--large-steps", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["config"]["frame_equals_single_device_frame"] is True
    assert line["value"] > 0 and "roofline" in line
    # a share this small is rendered as batches of `ranks` frames in one launch chain (rt_render_device_batch); the line says so and carries the latency of a lone frame beside it
    assert line["config"]["batch"] == 2 and line["config"]["frame_latency_ms"] > 0 and line["config"]["gather"] in ("f32", "rgb8")
    assert line["config"]["exchange"].startswith("one gather per BATCH")          # ... and its frames' tiles travel in one gather per batch (--exchange auto)
    r, line = _bench("--gpus", "2", "--share-gpu", "--check-frame", "--exchange", "frame", "--gather", "rgb8", "--width", "640", "--height", "356", "--steps", "5", "--warmup", "1", "--large-steps", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["config"]["exchange"] == "one gather per frame" and line["config"]["gather"] == "rgb8" and line["config"]["batch"] == 2
    # three ranks, a rotating root, a step count that is no multiple of the batch (7 = 3 + 3 + 1): frame 0 (root 0) still equals the single-device frame
    r, line = _bench("--gpus", "3", "--share-gpu", "--check-frame", "--root", "rotate", "--gather", "f32", "--width", "640", "--height", "356", "--steps", "7", "--warmup", "2", "--large-steps", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["config"]["batch"] == 3 and line["steps"] == 7 and line["config"]["frame_equals_single_device_frame"] is True and "rank k mod N" in line["config"]["root"]


def test_assemble_is_the_inverse_of_the_partition():
    from raytracinggpu_amd import tiling
    import raytracinggpu_amd as rt
    for H, W, G in ((1080, 8, 8), (50, 4, 3), (7, 3, 2), (64, 2, 4)):
        frame = torch.arange(H * W * 4, dtype=torch.float32).view(H, W, 4)
        locs = []
        for r in range(G):
            rows, idx = rt.interleaved_rows(H, tiling.TILE_ROWS, r, G)
            loc = tiling.local_buffer(H, W, G, "cpu")
            loc[:len(idx)] = frame[torch.as_tensor(idx, dtype=torch.long)]
            locs.append(loc)
        assert torch.equal(tiling.assemble(torch.stack(locs), H), frame)


@pytest.mark.gpu
def test_gathered_gpu_tiles_equal_single_gpu_frame(tmp_path):
    import raytracinggpu_amd as rt
    from raytracinggpu_amd import hostlib
    W, H, spp, b = 640, 356, 1, 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(3, _free_port(), "gpu", W, H, spp, b, out), nprocs=3, join=True)
    got = np.load(out)
    g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
    ctx = rt.Context(0)
    ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(g["vertices"], g["tri_obj_order"], object_slot=6))
    full = ctx.render(rt.make_params(W, H, spp, b, **rt.scenes.CPU_LAUNCHER))
    np.testing.assert_array_equal(got.view(np.uint32), full.view(np.uint32))


@pytest.mark.gpu
def test_single_process_multi_device_frame_is_bitwise_the_single_device_frame():
    """rt_render_multi (one host process, several device contexts, peer copies + one de-interleave kernel on the
    root): three contexts on the one GPU of the test box == the single-context frame, bit for bit; ray count and
    per-device statistics are reported."""
    import raytracinggpu_amd as rt
    from .conftest import load_golden
    g = load_golden("cat_mesh.npz")
    mesh = dict(vertices=g["vertices"], indices=g["tri_bvh_order"], bvh_arr10=g["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    one = rt.Context(0)
    one.scene_upload(rt.scenes.spheres("cpu"), mesh)
    for ids, (W, H) in (([0, 0, 0], (400, 250)), ([0], (333, 77)), ([0, 0, 0, 0, 0, 0, 0, 0], (640, 360))):
        p = rt.make_params(W, H, 2, 2, **rt.scenes.CPU_LAUNCHER)
        ref = one.render(p)
        m = rt.MultiContext(ids)
        m.scene_upload(rt.scenes.spheres("cpu"), mesh)
        got = m.render(p)
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
        st = m.stats()
        assert st["n_devices"] == len(ids) and st["rays"] == int(ref[..., 3].sum())
        assert all(k > 0 for k in st["kernel_ms"]) and st["gather_ms"] >= 0 and st["frame_ms"] > 0
        import torch
        dev = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
        m.render_device(p, dev.data_ptr())
        np.testing.assert_array_equal(dev.cpu().numpy().view(np.uint32), ref.view(np.uint32))
        m.close()
    with pytest.raises(rt.RtError):
        rt.MultiContext([0, 99])
    one.close()


@pytest.mark.gpu
def test_multi_device_host_path_submits_all_devices_at_once():
    """rt_render_multi drives every device from its own submit thread: with eight contexts (on the one GPU of the test box) at
    1920x1080 the host's wall clock of a frame exceeds the summed kernel time of the devices by at most 60 us, and the time to
    ISSUE all eight devices' work (rt_multi_stats.submit_ms) is a fraction of the frame.  The frame stays bitwise the single-device one."""
    import raytracinggpu_amd as rt
    from tests.conftest import load_golden
    g = load_golden("cat_mesh.npz")
    mesh = dict(vertices=g["vertices"], indices=g["tri_bvh_order"], bvh_arr10=g["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    p = rt.make_params(1920, 1080, 1, 3, **rt.scenes.CPU_LAUNCHER)
    one = rt.Context(0)
    one.scene_upload(rt.scenes.spheres("cpu"), mesh)
    ref = one.render(p)
    one.close()
    m = rt.MultiContext([0] * 8)
    m.scene_upload(rt.scenes.spheres("cpu"), mesh)
    import torch
    dev = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda:0")
    for _ in range(3):
        m.render_device(p, dev.data_ptr())
    best = None
    for _ in range(10):
        m.render_device(p, dev.data_ptr())
        st = m.stats()
        over = st["frame_ms"] - sum(st["kernel_ms"])
        if best is None or over < best[0]:
            best = (over, st["frame_ms"], st["submit_ms"], sum(st["kernel_ms"]))
    print(f"8 contexts, 1080p: frame {best[1]:.3f} ms, submit {best[2]:.3f} ms, sum of kernel times {best[3]:.3f} ms, frame - sum = {best[0] * 1e3:.0f} us")
    assert best[0] <= 0.060                    # (on ONE GPU the eight contexts' kernels overlap, so the sum exceeds the wall clock: the bound is met with room)
    assert best[2] <= 0.5 and best[2] < best[1]  # issuing eight devices' launch chains takes a fraction of a millisecond: they are submitted concurrently
    np.testing.assert_array_equal(dev.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    m.close()


@pytest.mark.gpu
def test_multi_device_rgb8_gather_equals_single_device_png_bytes():
    """rt_render_multi_rgb8: every device tonemaps its tiles and the exchange moves the 8-bit image (3 bytes per pixel instead
    of 16).  The assembled bytes equal rt_render_rgb8's on one device -- at 400x250 and, with eight contexts, at 7680x4320
    (BASELINE config 5) -- and the statistics report the gathered bytes and the peer paths."""
    import raytracinggpu_amd as rt
    from tests.conftest import load_golden
    g = load_golden("cat_mesh.npz")
    mesh = dict(vertices=g["vertices"], indices=g["tri_bvh_order"], bvh_arr10=g["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    one = rt.Context(0)
    one.scene_upload(rt.scenes.spheres("cpu"), mesh)
    for world, W, H in ((3, 400, 250), (8, 7680, 4320), (2, 333, 77)):
        p = rt.make_params(W, H, 1, 1, **rt.scenes.CPU_LAUNCHER)
        exp = one.render_rgb8(p)
        m = rt.MultiContext([0] * world)
        m.scene_upload(rt.scenes.spheres("cpu"), mesh)
        got = m.render_rgb8(p)
        np.testing.assert_array_equal(got, exp)
        st = m.stats()
        f32 = m.render(p)
        assert st["rays"] == int(f32[..., 3].astype(np.float64).sum())
        rows0 = len(rt.interleaved_rows(H, 8, 0, world)[1])
        assert st["gather_bytes"] == (H - rows0) * W * 3                  # everything but the root's own tiles, 3 bytes per pixel
        assert m.stats()["gather_bytes"] == (H - rows0) * W * 16          # the float4 exchange moves 16
        assert st["peer_access"] == [-1] * world                          # all contexts on the one GPU of the test box
        m.close()
    one.close()


def test_bench_capi_transport_has_no_cpu_leg():
    """`--transport capi` is RCCL between GPUs through libraytrace_rccl.so: with the CPU stand-in renderer it stops with a message
    instead of silently taking another transport."""
    r, line = _bench("--gpus", "2", "--renderer", "oracle", "--transport", "capi", "--width", "64", "--height", "32", "--steps", "1", "--warmup", "0", "--large-steps", "0")
    assert r.returncode != 0 and line is None
    assert "no gloo / CPU leg" in (r.stderr + r.stdout)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Tests that wake up on a second GPU.  The build's own test box has ONE MI355X: there they are collected and
# skipped with that reason; on a box with two or more devices they run two real RCCL ranks / two real devices with nothing stubbed.
# ---------------------------------------------------------------------------------------------------------------------------------------
def _n_gpus():
    return torch.cuda.device_count()


needs_two_gpus = pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: this box shows %d (one-GPU boxes cover the N-rank path with --share-gpu / gloo and a one-rank communicator)" % _n_gpus())


@pytest.mark.gpu
@needs_two_gpus
@pytest.mark.parametrize("transport,gather,plan", [("torch", "f32", "auto"), ("torch", "rgb8", "auto"), ("capi", "f32", "tile"), ("capi", "f32", "coalesced"), ("capi", "rgb8", "auto")])
def test_two_real_rccl_ranks_through_bench(transport, gather, plan):
    """`bench.py --gpus 2`: one rank per GPU over RCCL -- through torch.distributed (nccl) and through the product's own transport
    (libraytrace_rccl.so: per-tile receives in place, and the coalesced plan with its placement kernel) -- H not a multiple of 8; the
    gathered float4 frame is bitwise the frame one device renders alone."""
    args = ["--gpus", "2", "--transport", transport, "--gather", gather, "--comm-plan", plan, "--width", "640", "--height", "356", "--steps", "3", "--warmup", "1", "--large-steps", "0"]
    if gather == "f32":
        args.append("--check-frame")
    r, line = _bench(*args)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["value"] > 0
    if gather == "f32":
        assert line["config"]["frame_equals_single_device_frame"] is True
    if transport == "capi":
        assert line["config"]["comm_plan"] in ("tile", "coalesced") and (plan == "auto" or line["config"]["comm_plan"] == plan)


@pytest.mark.gpu
@needs_two_gpus
def test_two_real_rccl_ranks_more_ranks_than_tiles():
    """n_tiles < world: a 640x8 frame is ONE tile, rank 1 holds nothing and sends nothing; both plans."""
    for plan in ("tile", "coalesced"):
        r, line = _bench("--gpus", "2", "--transport", "capi", "--comm-plan", plan, "--check-frame", "--width", "640", "--height", "8", "--steps", "2", "--warmup", "1", "--large-steps", "0")
        assert r.returncode == 0, r.stderr[-3000:]
        assert line["config"]["frame_equals_single_device_frame"] is True


@pytest.mark.gpu
@needs_two_gpus
def test_launcher_two_processes_rccl_png_equals_the_reference_bytes(tmp_path):
    """`rt_launcher 1 0 --tile-rank r --tile-world 2 --rccl-id FILE --device r`: two C++ processes, one per GPU, the id through a file tied to
    the launch by its nonce, one RCCL gather of the tone-mapped tiles; rank 0's PNG == the bytes of the reference's `./cpu 1 0`."""
    from PIL import Image
    from .conftest import load_golden
    cat = load_golden("cat_mesh.npz")
    g = load_golden("ref_cpu_png_1_0.npz")
    launcher = os.path.join(ROOT, "raytracinggpu_amd", "rt_launcher")
    d = tmp_path / "cadnav.com_model" / "Models_F0202A090"
    d.mkdir(parents=True)
    with open(d / "cat.obj", "w") as f:
        for v in cat["vertices"]:
            f.write("v %.9g %.9g %.9g 1 1 1\r\n" % tuple(float(x) for x in v))
        for t in cat["tri_obj_order"]:
            f.write("f %d/1/1 %d/1/1 %d/1/1\r\n" % tuple(int(x) + 1 for x in t))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    ps = [subprocess.Popen([launcher, "1", "0", "--tile-rank", str(rk), "--tile-world", "2", "--device", str(rk), "--rccl-id", str(tmp_path / "id"),
                            "--rccl-nonce", "two-gpu-test", "--rccl-timeout", "120", "--out", "two.png"], cwd=tmp_path, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for rk in (1, 0)]      # rank 1 first: it waits for the id
    try:
        outs = [p.communicate(timeout=300) for p in ps]
    finally:
        for p in ps:
            if p.poll() is None:
                p.kill(); p.communicate()
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, se[-2000:]
        assert "over RCCL" in se
    assert not (tmp_path / "id").exists()
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "two.png").convert("RGB")), g["cat"])


@pytest.mark.gpu
@needs_two_gpus
def test_single_process_two_real_devices_peer_access():
    """rt_render_multi over devices {0, 1}: the peer pushes its tiles into the root's HBM over xGMI (peer_access == 1 for device 1, -1 = not
    needed for the root); float4 frame and RGB8 image bitwise the single-device ones, twice in a row."""
    import raytracinggpu_amd as rt
    from .conftest import load_golden
    g = load_golden("cat_mesh.npz")
    mesh = dict(vertices=g["vertices"], indices=g["tri_bvh_order"], bvh_arr10=g["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    one = rt.Context(0)
    one.scene_upload(rt.scenes.spheres("cpu"), mesh)
    m = rt.MultiContext([0, 1])
    m.scene_upload(rt.scenes.spheres("cpu"), mesh)
    for W, H in ((640, 356), (1920, 1080)):
        p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
        ref, ref8 = one.render(p), one.render_rgb8(p)
        for _ in range(2):
            np.testing.assert_array_equal(m.render(p).view(np.uint32), ref.view(np.uint32))
            np.testing.assert_array_equal(m.render_rgb8(p), ref8)
        st = m.stats()
        assert st["n_devices"] == 2 and list(st["peer_access"][:2]) == [-1, 1], st
    m.close(); one.close()
