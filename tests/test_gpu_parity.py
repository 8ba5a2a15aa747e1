"""HIP render path (through the C-ABI) vs the CPU oracle on the same inputs.  -m gpu.

Tolerance (BASELINE.json / SURVEY 8d): per-channel L-inf <= 1e-4 on g = min(pow(c,1/2.2),255)/255.
For sigma == 0 the linear float colour IS bit-identical -- same single IEEE roundings everywhere, and the binary64
sin / cos of the bounce direction (rt_sincos.h) gives the same binary32 products as glibc over all 2^24 arguments
(tools/check_sincos.cpp) -- so every such comparison asserts exact equality of every channel.  Only frames with
pixel jitter (sigma != 0: device logf vs glibc) are held to the 1e-4 bound with a stated identical fraction.
"""
import os

import numpy as np
import pytest

import raytracinggpu_amd as rt
from .conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    yield c
    c.close()


def upload(ctx, scene, cat_golden, ref_arrays=True):
    mesh = None
    if scene in ("cpu", "optimized"):
        mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"],
                    albedo=rt.scenes.CAT_ALBEDO, object_slot=rt.scenes.mesh_slot(scene))
    ctx.scene_upload(rt.scenes.spheres(scene), mesh)


def linf(oracle, a, b):
    return float(np.abs(oracle.gamma_unit(a[..., :3]) - oracle.gamma_unit(b[..., :3])).max())


def values_equal(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


VARIANTS = ["path", "wavefront_queue", "wavefront_lds", "wavefront", "global", "lockstep"]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("scene,W,H", [("cpu", 512, 512), ("spheres", 512, 512), ("demo10", 256, 256), ("cpu", 333, 77)])
def test_direct_lighting_bit_exact(ctx, oracle, oracle_cat, cat_golden, scene, W, H, variant):
    """num_bounce=0, sigma=0: deterministic in the reference too -> bit-exact linear colour."""
    upload(ctx, scene, cat_golden)
    got = ctx.render(rt.make_params(W, H, 1, 0, variant=variant, **rt.scenes.CPU_LAUNCHER))
    exp, exp8, _ = oracle.Scene.preset(scene, oracle_cat if scene == "cpu" else None).render(W, H, 1, 0)
    assert values_equal(got[..., :3], exp[..., :3]).all()
    np.testing.assert_array_equal(got[..., 3], exp[..., 3])          # rays per pixel
    assert linf(oracle, got, exp) == 0.0
    got8 = ctx.render_rgb8(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER))
    np.testing.assert_array_equal(got8, exp8)


def test_png_bytes_equal_unmodified_reference_binary(ctx, cat_golden):
    """GPU 8-bit image == bytes of `./cpu 1 0` (the reference program itself)."""
    g = load_golden("ref_cpu_png_1_0.npz")
    upload(ctx, "cpu", cat_golden)
    np.testing.assert_array_equal(ctx.render_rgb8(rt.make_params(512, 512, 1, 0, **rt.scenes.CPU_LAUNCHER)), g["cat"])
    upload(ctx, "spheres", cat_golden)
    np.testing.assert_array_equal(ctx.render_rgb8(rt.make_params(512, 512, 1, 0, **rt.scenes.CPU_LAUNCHER)), g["spheres"])


def test_reference_getColor_floats_1080p(ctx, cat_golden):
    """Linear floats of the reference's own Scene::getColor at 1920x1080 (every 8th pixel), bit-exact."""
    g = load_golden("ref_render.npz")
    W, H, spp, b, stride = (int(x) for x in g["cpu_1080p_direct_cfg"])
    upload(ctx, "cpu", cat_golden)
    got = ctx.render(rt.make_params(W, H, spp, b, **rt.scenes.CPU_LAUNCHER))
    np.testing.assert_array_equal(got[::stride, ::stride, :3], g["cpu_1080p_direct_color"])


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("scene,W,H,spp,b", [("cpu", 512, 512, 2, 3), ("demo10", 256, 256, 2, 5), ("spheres", 320, 200, 4, 2),
                                             ("cpu", 256, 256, 1, 10)])
def test_bounces_within_tolerance(ctx, oracle, oracle_cat, cat_golden, scene, W, H, spp, b, variant):
    upload(ctx, scene, cat_golden)
    got = ctx.render(rt.make_params(W, H, spp, b, variant=variant, **rt.scenes.CPU_LAUNCHER))
    exp, _, _ = oracle.Scene.preset(scene, oracle_cat if scene == "cpu" else None).render(W, H, spp, b, want_rgb8=False)
    same = values_equal(got[..., :3], exp[..., :3]).mean()
    err = linf(oracle, got, exp)
    print(f"{scene} {W}x{H} spp={spp} b={b}: bit-identical channels {same:.6f}, Linf(gamma) {err:.3g}")
    assert err <= TOL
    assert values_equal(got[..., :3], exp[..., :3]).all()              # sigma == 0: every channel bit-identical
    np.testing.assert_array_equal(got[..., 3], exp[..., 3])


def test_optimized_cu_conventions(ctx, oracle, cat_golden):
    """optimized.cu's scene: mesh at slot 1 rescaled 0.6/(0,-4,0), eps 1e-4, t>0, b segments."""
    m = oracle.Mesh.from_arrays(cat_golden["vertices"], cat_golden["tri_obj_order"])
    m.rescale(0.6, (0, -4, 0))
    m.build_bvh()
    ctx.scene_upload(rt.scenes.spheres("optimized"),
                     dict(vertices=m.vertices, indices=m.triangles, bvh_arr10=m.bvh_array(), albedo=rt.scenes.CAT_ALBEDO, object_slot=1))
    kw = dict(rt.scenes.OPTIMIZED_CU, sigma=0.0)
    got = ctx.render(rt.make_params(384, 384, 2, 3, **kw))
    exp, _, _ = oracle.Scene.preset("optimized", m).render(384, 384, 2, 2, eps=1e-4, tri_tmin=0.0, want_rgb8=False)
    assert linf(oracle, got, exp) <= TOL
    assert values_equal(got[..., :3], exp[..., :3]).all()              # sigma == 0: every channel bit-identical, not merely within the bound
    np.testing.assert_array_equal(got[..., 3], exp[..., 3])


def test_row_ranges_and_interleaved_tiles_are_bitwise_the_full_frame(ctx, cat_golden):
    import torch
    upload(ctx, "cpu", cat_golden)
    W, H = 400, 250          # neither a multiple of the 32x8 workgroup tile
    p = rt.make_params(W, H, 2, 2, **rt.scenes.CPU_LAUNCHER)
    full = ctx.render(p)
    part = ctx.render(p, 37, 121)
    np.testing.assert_array_equal(part.view(np.uint32), full[37:121].view(np.uint32))
    assert ctx.render(p, 5, 5).shape == (0, W, 4)
    for world in (2, 3, 8):
        frame = np.zeros_like(full)
        for rank in range(world):
            rows, idx = rt.interleaved_rows(H, 8, rank, world)
            buf = torch.empty((max(rows.n_rows, 1), W, 4), dtype=torch.float32, device="cuda:0")
            ctx.render_device(p, rows, buf.data_ptr())
            ctx.synchronize()
            frame[idx] = buf[:rows.n_rows].cpu().numpy()
        np.testing.assert_array_equal(frame.view(np.uint32), full.view(np.uint32))


def test_pinned_frame_buffer_gives_the_same_frame(ctx, cat_golden):
    """rt_host_alloc: a frame buffer the D2H copy reaches by DMA; same bits as the pageable path, for rt_render and row ranges."""
    upload(ctx, "cpu", cat_golden)
    p = rt.make_params(400, 250, 2, 2, **rt.scenes.CPU_LAUNCHER)
    ref = ctx.render(p)
    pin = rt.PinnedArray((250, 400, 4))
    got = ctx.render(p, out=pin.array)
    assert got is pin.array
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    part = rt.PinnedArray((84, 400, 4))
    np.testing.assert_array_equal(ctx.render(p, 37, 121, out=part.array).view(np.uint32), ref[37:121].view(np.uint32))
    pin.close(); part.close()


def test_pinned_views_outlive_close(ctx, cat_golden):
    """A view of a PinnedArray keeps the allocation alive after close() (the memory is released with the last view)."""
    upload(ctx, "cpu", cat_golden)
    p = rt.make_params(64, 40, 1, 1, **rt.scenes.CPU_LAUNCHER)
    pin = rt.PinnedArray((40, 64, 4))
    view = ctx.render(p, out=pin.array)[3:7]
    keep = view.copy()
    pin.close()
    del pin
    import gc; gc.collect()
    other = rt.PinnedArray((40, 64, 4)); other.array[:] = -1.0          # would land on the freed block if it had been freed
    np.testing.assert_array_equal(view.view(np.uint32), keep.view(np.uint32))
    other.close()


def test_async_pipelined_frames_equal_the_synchronous_ones(ctx, cat_golden):
    """rt_render_async / rt_wait: frame k's device-to-host copy runs on the copy stream beside frame k+1's kernels (two device
    slots); every delivered frame -- float4 and the 8-bit image -- is bitwise what rt_render / rt_render_rgb8 return, including
    when a slot is re-used without waiting for it and when the frames differ from one another."""
    upload(ctx, "cpu", cat_golden)
    W, H = 400, 250
    params = [rt.make_params(W, H, 1, b, **rt.scenes.CPU_LAUNCHER) for b in (0, 2, 1, 3, 2)]
    ref = [ctx.render(p) for p in params]
    ref8 = [ctx.render_rgb8(p) for p in params]
    pins = [rt.PinnedArray((H, W, 4)) for _ in range(2)]
    got = []
    ctx.render_async(params[0], pins[0].array, slot=0)
    for k in range(1, len(params)):
        ctx.render_async(params[k], pins[k & 1].array, slot=k & 1)       # frame k is submitted ...
        ctx.wait((k - 1) & 1)                                            # ... before frame k-1 has been collected
        got.append(pins[(k - 1) & 1].array.copy())
    ctx.wait((len(params) - 1) & 1)
    got.append(pins[(len(params) - 1) & 1].array.copy())
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g.view(np.uint32), r.view(np.uint32))
    pins8 = [rt.PinnedArray((H, W, 3), dtype=np.uint8) for _ in range(2)]
    for k, p in enumerate(params):
        ctx.render_async(p, pins8[k & 1].array, slot=k & 1, rgb8=True)
        ctx.wait(k & 1)
        np.testing.assert_array_equal(pins8[k & 1].array, ref8[k])
    # a slot re-used without rt_wait: its kernels queue behind the pending copy, the second frame arrives intact
    ctx.render_async(params[1], pins[0].array, slot=0)
    ctx.render_async(params[3], pins[0].array, slot=0)
    ctx.wait(0)
    np.testing.assert_array_equal(pins[0].array.view(np.uint32), ref[3].view(np.uint32))
    with pytest.raises(rt.RtError):
        ctx.wait(1)                                                      # nothing in flight there
    with pytest.raises(rt.RtError):
        ctx.render_async(params[0], pins[0].array, slot=2)
    # pageable memory works too (the runtime stages the copy)
    out = np.empty((H, W, 4), np.float32)
    ctx.render_async(params[2], out, slot=1); ctx.wait(1)
    np.testing.assert_array_equal(out.view(np.uint32), ref[2].view(np.uint32))
    for pn in pins + pins8:
        pn.close()


def test_error_paths(ctx, cat_golden):
    upload(ctx, "cpu", cat_golden)
    with pytest.raises(rt.RtError):
        ctx.render(rt.make_params(0, 10))
    with pytest.raises(rt.RtError):
        ctx.render(rt.make_params(64, 64, 0, 0))
    with pytest.raises(rt.RtError):
        ctx.render(rt.make_params(64, 64, 1, 40))
    with pytest.raises(rt.RtError):
        ctx.render(rt.make_params(64, 64), 10, 200)
    bad = np.array(cat_golden["bvh_arr10"]); bad[0, 0] = 5000
    with pytest.raises(rt.RtError):
        ctx.scene_upload(rt.scenes.spheres("cpu"), dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=bad))
    fresh = rt.Context(0)
    with pytest.raises(rt.RtError):
        fresh.render(rt.make_params(64, 64))
    fresh.close()
    # the context is still usable after errors
    upload(ctx, "spheres", cat_golden)
    assert np.isfinite(ctx.render(rt.make_params(64, 64))).all()
    # and everything it allocated (scene, path state, queues, scratch) lives on its own device
    upload(ctx, "cpu", cat_golden)
    ctx.render(rt.make_params(128, 96, 3, 2, **rt.scenes.CPU_LAUNCHER))
    ctx.render(rt.make_params(128, 96, 2, 1, variant="path", **rt.scenes.CPU_LAUNCHER))
    ctx.mesh_rebuild(len(cat_golden["tri_bvh_order"]))
    ctx.selfcheck()


def test_launcher_cli_reproduces_reference_png(tmp_path, cat_golden):
    """`rt_launcher 1 0` in a directory holding the OBJ == decoded bytes of the reference's `./cpu 1 0`."""
    import os
    import subprocess
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    launcher = os.path.join(root, "raytracinggpu_amd", "rt_launcher")
    g = load_golden("ref_cpu_png_1_0.npz")
    # OBJ rebuilt from the fixture: 6-number vertex lines are NOT transformed by readOBJ (cpu:344-350), and
    # %.9g round-trips binary32, so the parser reproduces the fixture's already-transformed vertices exactly
    d = tmp_path / "cadnav.com_model" / "Models_F0202A090"
    d.mkdir(parents=True)
    with open(d / "cat.obj", "w") as f:
        for v in cat_golden["vertices"]:
            f.write("v %.9g %.9g %.9g 1 1 1\r\n" % tuple(float(x) for x in v))
        for t in cat_golden["tri_obj_order"]:
            f.write("f %d/1/1 %d/1/1 %d/1/1\r\n" % tuple(int(x) + 1 for x in t))
    r = subprocess.run([launcher, "1", "0"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("Rendering time: ") and r.stdout.rstrip().endswith(" s")
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "image.png").convert("RGB")), g["cat"])
    # the same through rt_render_multi_rgb8 (three contexts on the one GPU, tonemapped tiles exchanged)
    r = subprocess.run([launcher, "1", "0", "--devices", "0,0,0", "--out", "multi.png"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "multi.png").convert("RGB")), g["cat"])
    # one process per GPU from C++: three launcher processes render their interleaved tiles (here on the one GPU),
    # a fourth assembles the PNG without touching a GPU: the reference's bytes again
    files = []
    for r in range(3):
        fn = str(tmp_path / f"tiles{r}.rgb")
        q = subprocess.run([launcher, "1", "0", "--tile-rank", str(r), "--tile-world", "3", "--tiles", fn], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert q.returncode == 0, q.stderr
        files.append(fn)
    q = subprocess.run([launcher, "1", "0", "--assemble", ",".join(files), "--out", "ranks.png"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert q.returncode == 0, q.stderr
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "ranks.png").convert("RGB")), g["cat"])
    # no OBJ in the working directory: "Error opening file!" and the spheres-only image (cpu:322-325)
    e = tmp_path / "empty"
    e.mkdir()
    r = subprocess.run([launcher, "1", "0"], cwd=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "Error opening file!" in r.stdout
    np.testing.assert_array_equal(np.array(Image.open(e / "image.png").convert("RGB")), g["spheres"])


def test_product_builder_feeds_the_kernel(ctx, oracle, oracle_cat, cat_golden):
    """End to end with the product's own host code: fixture arrays -> C++ buildBVH/bvhTreeToArray ->
    stride-10 TriangleIndices -> upload -> render == oracle."""
    from raytracinggpu_amd import hostlib
    mesh = hostlib.build_mesh(cat_golden["vertices"], cat_golden["tri_obj_order"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    assert mesh["indices"].shape[1] == 10
    ctx.scene_upload(rt.scenes.spheres("cpu"), mesh)
    got = ctx.render(rt.make_params(640, 360, 1, 1, **rt.scenes.CPU_LAUNCHER))
    exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(640, 360, 1, 1, want_rgb8=False)
    assert linf(oracle, got, exp) <= TOL
    assert values_equal(got[..., :3], exp[..., :3]).all()


def test_work_counters_equal_oracle_counters(ctx, oracle, oracle_cat, cat_golden):
    """rt_count_work (counting instantiation of the kernel) == the oracle's counting pass: same rays, box tests,
    nodes and triangle tests, i.e. the stackless traversal visits exactly what cpu:277-311 visits."""
    upload(ctx, "cpu", cat_golden)
    for W, H, spp, b in ((512, 512, 1, 0), (320, 180, 2, 3)):
        _, _, exp = oracle.Scene.preset("cpu", oracle_cat).render(W, H, spp, b, want_rgb8=False)
        for variant in VARIANTS:
            got = ctx.count_work(rt.make_params(W, H, spp, b, variant=variant, **rt.scenes.CPU_LAUNCHER))
            assert got == {k: exp[k] for k in ("rays", "box_tests", "nodes", "tri_tests")}, variant


def test_headline_config_full_frame_against_the_oracle(ctx, oracle, oracle_cat, cat_golden):
    """BASELINE config 3 (the bench workload: cat, 1920x1080, num_rays 1, num_bounce 3 = 4 segments): the WHOLE frame
    against the oracle -- ray count per pixel equal, EVERY channel of every pixel bit-identical (sigma = 0: asserted with
    .all(), the 1e-4 bound beside it) -- every kernel variant writes the same bits, and the work counters equal the oracle's."""
    upload(ctx, "cpu", cat_golden)
    W, H = 1920, 1080
    exp, _, cnt = oracle.Scene.preset("cpu", oracle_cat).render(W, H, 1, 3, want_rgb8=False)
    ref = None
    for variant in ["auto"] + VARIANTS:
        got = ctx.render(rt.make_params(W, H, 1, 3, variant=variant, **rt.scenes.CPU_LAUNCHER))
        assert np.isfinite(got).all()
        if ref is None:
            ref = got
        else:
            np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    np.testing.assert_array_equal(ref[..., 3], exp[..., 3])
    assert int(ref[..., 3].astype(np.float64).sum()) == cnt["rays"]
    same = values_equal(ref[..., :3], exp[..., :3]).mean()
    err = linf(oracle, ref, exp)
    print(f"cat 1920x1080 b=3: rays {cnt['rays']}, bit-identical channels {same:.6f}, Linf(gamma) {err:.3g}")
    assert err <= TOL and values_equal(ref[..., :3], exp[..., :3]).all()
    assert ctx.count_work(rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)) == {k: cnt[k] for k in ("rays", "box_tests", "nodes", "tri_tests")}


def test_config2_spheres_only_full_size(ctx, oracle, cat_golden):
    """BASELINE config 2: walls + the four demo spheres (mirror, refraction, nested refraction), no mesh, 1920x1080,
    3 bounces: whole frame against the oracle."""
    upload(ctx, "demo10", cat_golden)
    W, H = 1920, 1080
    exp, _, cnt = oracle.Scene.preset("demo10", None).render(W, H, 1, 3, want_rgb8=False)
    for variant in ("auto", "wavefront_queue", "lockstep"):
        got = ctx.render(rt.make_params(W, H, 1, 3, variant=variant, **rt.scenes.CPU_LAUNCHER))
        np.testing.assert_array_equal(got[..., 3], exp[..., 3])
        assert linf(oracle, got, exp) <= TOL
        assert values_equal(got[..., :3], exp[..., :3]).all()
    assert int(got[..., 3].astype(np.float64).sum()) == cnt["rays"]


def test_auto_variant_is_the_lock_step_kernel_without_a_mesh_and_the_work_stack_pipeline_with_one(ctx, cat_golden):
    """RT_VARIANT_AUTO (round 5): a scene without a mesh has no traversal to feed, so one lane per pixel for the whole frame (variant 5) renders it; a scene with
    a mesh, or a posed camera (the wavefront family only), takes the wavefront pipeline with the work-stack traversal (8).  Both are checked against the oracle by the
    tests above; this one pins which kernel AUTO means (rt_stats.variant)."""
    small = rt.make_params(96, 64, 1, 2, **rt.scenes.CPU_LAUNCHER)
    upload(ctx, "demo10", cat_golden)
    assert ctx.stats_after_render(small)["variant"] == rt._capi.VARIANTS["lockstep"]
    ctx.render_pose(small, rt.make_pose(yaw=0.2))
    assert ctx.stats()["variant"] == rt._capi.VARIANTS["wavefront_queue"]
    upload(ctx, "cpu", cat_golden)
    assert ctx.stats_after_render(small)["variant"] == rt._capi.VARIANTS["wavefront_queue"]


@pytest.mark.parametrize("env", [{"RT_TRAVQ_CAP": "128"}, {"RT_TRAVQ_R": "32"}, {"RT_TRAVQ_R": "32", "RT_TRAVQ_CAP": "128"},
                                 {"RT_TRAVQ_LDS": "12"}, {"RT_TRAVQ_LDS": "12", "RT_TRAVQ_R": "32"}, {"RT_TRAVQ_LDS": "16"},
                                 {"RT_TRAVQ_LDS": "8", "RT_TRAVQ_CAP": "128"}, {"RT_TRAVQ_QW": "0", "RT_TRAVQ_Q16": "1"}, {"RT_TRAVQ_QW": "0", "RT_TRAVQ_Q16": "1", "RT_TRAVQ_CAP": "128"},
                                 {"RT_TRAVQ_QW": "0"}, {"RT_TRAVQ_QW": "0", "RT_TRAVQ_CAP": "128"}, {"RT_TRAVQ_QW": "1"}, {"RT_TRAVQ_QW": "1", "RT_TRAVQ_CAP": "128"}, {"RT_TRAVQ_QSEL": "0"}, {"RT_TRAVQ_QSEL": "0", "RT_TRAVQ_CAP": "128"}])
def test_work_stack_traversal_bounded_stack_and_slot_counts(ctx, cat_golden, monkeypatch, env):
    """wf_travq with a 128-entry stack (forces the serial-drain path that keeps LDS bounded for any tree), with
    32 ray slots per wave, with all / the top 15 BVH nodes staged in LDS (RT_TRAVQ_LDS = waves per CU), with the BOX step reading
    16-bit fixed-point sibling pairs (RT_TRAVQ_Q16, rt_qnodes.hip.h), with the float sibling pairs the default kernel was until round 5
    (RT_TRAVQ_QW=0), with the 4-wide BOX step named explicitly (RT_TRAVQ_QW=1: the default for the cat; with a 128-entry stack its steps
    predict an overflow and walk their pairs serially) and with its quads taking every other level of the tree instead of the cuts the surface-area DP
    picks (RT_TRAVQ_QSEL=0: any cut is exact): same bits and same work counters as the stackless-walk kernel."""
    upload(ctx, "cpu", cat_golden)
    p = rt.make_params(640, 360, 2, 3, variant="wavefront_queue", **rt.scenes.CPU_LAUNCHER)
    ref = ctx.render(rt.make_params(640, 360, 2, 3, variant="wavefront", **rt.scenes.CPU_LAUNCHER))
    work = ctx.count_work(rt.make_params(640, 360, 2, 3, variant="wavefront", **rt.scenes.CPU_LAUNCHER))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tuned = rt.Context(0)                    # the knobs are read once, when a context is created
    upload(tuned, "cpu", cat_golden)
    got = tuned.render(p)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert tuned.count_work(p) == work
    tuned.close()


def test_samples_as_parallel_items_equal_the_serial_sample_loop(ctx, oracle, oracle_cat, cat_golden, monkeypatch):
    """The samples of a pixel traced together as items of one launch chain (default while the chain's state fits the Infinity
    Cache), one chain per sample (RT_PATH_SAMP_MB=1), and uneven chains (5 samples, 2 per chain): the ordered reduction gives the
    same bits as the reference's serial loop (cpu:701-713), through the wavefront pipeline and through wf_path."""
    upload(ctx, "cpu", cat_golden)
    W, H, spp, b = 320, 200, 5, 2
    exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(W, H, spp, b, sigma=0.2, want_rgb8=False)
    kw = dict(rt.scenes.CPU_LAUNCHER, sigma=0.2)
    ref = ctx.render(rt.make_params(W, H, spp, b, **kw))
    assert linf(oracle, ref, exp) <= TOL
    np.testing.assert_array_equal(ref[..., 3], exp[..., 3])
    per_item_mb = W * H * 132 / 2**20
    for mb in (1, int(2.2 * per_item_mb) + 1):
        monkeypatch.setenv("RT_PATH_SAMP_MB", str(mb))
        tuned = rt.Context(0)
        upload(tuned, "cpu", cat_golden)
        for variant in ("auto", "path"):
            got = tuned.render(rt.make_params(W, H, spp, b, variant=variant, **kw))
            np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
        tuned.close()
    monkeypatch.delenv("RT_PATH_SAMP_MB")


def test_jitter_sigma_0p2_matches_oracle(ctx, oracle, oracle_cat, cat_golden):
    """anti-aliasing jitter sigma = 0.2 (optimized.cu:753; cpu:705-707 with its sigma line enabled),
    Box-Muller from the counter RNG's dims 2,3: same estimator as the oracle, within the stated tolerance."""
    upload(ctx, "cpu", cat_golden)
    kw = dict(rt.scenes.CPU_LAUNCHER, sigma=0.2)
    got = ctx.render(rt.make_params(320, 200, 4, 2, **kw))
    exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(320, 200, 4, 2, sigma=0.2, want_rgb8=False)
    same = values_equal(got[..., :3], exp[..., :3]).mean()
    print(f"sigma=0.2: bit-identical channels {same:.6f}, Linf(gamma) {linf(oracle, got, exp):.3g}")
    assert linf(oracle, got, exp) <= TOL
    assert same > 0.99
    np.testing.assert_array_equal(got[..., 3], exp[..., 3])
    # jitter really moves the samples: the image differs from the sigma = 0 one along edges
    plain = ctx.render(rt.make_params(320, 200, 4, 2, **rt.scenes.CPU_LAUNCHER))
    assert (plain[..., :3] != got[..., :3]).any()


def test_config4_3840x2160_lds_staged_variants_and_oracle_bands(ctx, oracle, oracle_cat, cat_golden):
    """BASELINE config 4 (cat, 3840x2160, LDS-staged variant): the default kernel, the work-stack kernel with the top
    of the BVH staged in LDS and the stackless kernel with every node staged in LDS write the same bits; row bands of
    the deterministic (num_bounce = 0) frame equal the oracle bit for bit."""
    upload(ctx, "cpu", cat_golden)
    W, H = 3840, 2160
    p3 = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
    ref = ctx.render(p3)
    assert np.isfinite(ref).all()
    lds_all = ctx.render(rt.make_params(W, H, 1, 3, variant="wavefront_lds", **rt.scenes.CPU_LAUNCHER))
    np.testing.assert_array_equal(lds_all.view(np.uint32), ref.view(np.uint32))
    del lds_all
    for variant in ("lds_top", "lds_verts", "lds_all"):
        staged = ctx.render(rt.make_params(W, H, 1, 3, variant=variant, **rt.scenes.CPU_LAUNCHER))
        np.testing.assert_array_equal(staged.view(np.uint32), ref.view(np.uint32))
        del staged
    direct = ctx.render(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER))
    sc = oracle.Scene.preset("cpu", oracle_cat)
    for a, b in ((0, 4), (1076, 1092), (1600, 1604), (2156, 2160)):
        exp, _, _ = sc.render(W, H, 1, 0, rows=(a, b), want_rgb8=False)
        np.testing.assert_array_equal(direct[a:b].view(np.uint32), exp.view(np.uint32))
    exp, _, _ = sc.render(W, H, 1, 3, rows=(1080, 1084), want_rgb8=False)
    assert linf(oracle, ref[1080:1084], exp) <= TOL
    np.testing.assert_array_equal(ref[1080:1084, :, 3], exp[..., 3])


def test_config5_7680x4320_eight_way_tiles_are_bitwise_the_full_frame(ctx, oracle, oracle_cat, cat_golden):
    """BASELINE config 5 (cat, 7680x4320, row-tiled over 8 GPUs): the eight ranks' interleaved 8-row tiles, rendered
    one after the other on the one GPU of the test box, are bitwise the single-device frame (the RNG and the camera
    are keyed by the global pixel); row bands equal the oracle; rt_render_multi with eight contexts gives the same
    frame (size-independent properties at the full BASELINE size)."""
    import torch
    upload(ctx, "cpu", cat_golden)
    W, H = 7680, 4320
    B = 3                                                               # the bench's num_bounce (config.large of bench.py), not a lighter stand-in
    p = rt.make_params(W, H, 1, B, **rt.scenes.CPU_LAUNCHER)
    full = ctx.render(p)
    assert int(full[..., 3].astype(np.float64).sum()) > W * H
    for rank in range(8):
        rows, idx = rt.interleaved_rows(H, 8, rank, 8)
        buf = torch.empty((rows.n_rows, W, 4), dtype=torch.float32, device="cuda:0")
        ctx.render_device(p, rows, buf.data_ptr())
        ctx.synchronize()
        np.testing.assert_array_equal(buf.cpu().numpy().view(np.uint32), full[idx].view(np.uint32))
        del buf
    sc = oracle.Scene.preset("cpu", oracle_cat)
    for a, b in ((0, 2), (2158, 2162), (4316, 4320)):
        exp, _, _ = sc.render(W, H, 1, B, rows=(a, b), want_rgb8=False)
        assert linf(oracle, full[a:b], exp) <= TOL
        assert values_equal(full[a:b, :, :3], exp[..., :3]).all()      # sigma == 0: bit for bit
        np.testing.assert_array_equal(full[a:b, :, 3], exp[..., 3])
    mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"],
                albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    m = rt.MultiContext([0] * 8)
    m.scene_upload(rt.scenes.spheres("cpu"), mesh)
    got = m.render(p)
    np.testing.assert_array_equal(got.view(np.uint32), full.view(np.uint32))
    assert m.stats()["rays"] == int(full[..., 3].astype(np.float64).sum())
    m.close()


def test_posed_camera_and_progressive_accumulation(ctx, oracle, oracle_cat, cat_golden):
    """SURVEY 8f2 (headless realtime_render.cu): posed camera {C, yaw, pitch} with the reference's ray generation and
    per-sample averaging, frame seeds WangHash(frame), accumbuffer / framenumber display.  Checked against the oracle's
    restatement (the CUDA + GL program itself cannot run here: this row's parity is unpinned)."""
    upload(ctx, "cpu", cat_golden)
    W, H, spp, b = 256, 160, 2, 2
    kw = dict(rt.scenes.CPU_LAUNCHER, sigma=0.2)
    sc = oracle.Scene.preset("cpu", oracle_cat)
    for pos, yaw, pitch in (((0.0, 0.0, 55.0), 0.0, 0.3), ((-6.0, 4.0, 49.0), 0.35, 0.1)):
        pose = rt.make_pose(pos, yaw, pitch)
        ctx.progressive_reset()
        accum = np.zeros((H, W, 4), np.float32)
        for frame in (1, 2, 3):
            seed = oracle.wang_hash(frame)
            got = ctx.render_pose(rt.make_params(W, H, spp, b, **dict(kw, seed=seed)), pose)
            exp, _, _ = sc.render(W, H, spp, b, sigma=0.2, seed=seed, fov=np.float32(np.pi / 2), cam=pos, pose=(yaw, pitch), want_rgb8=False)
            assert linf(oracle, got, exp) <= TOL
            assert values_equal(got[..., :3], exp[..., :3]).mean() > 0.99
            np.testing.assert_array_equal(got[..., 3], exp[..., 3])
            # the library's own accumulation of the SAME frames, replayed by the oracle's accumulate: bit-exact floats
            disp, rgb8 = ctx.progressive_frame(rt.make_params(W, H, spp, b, **kw), pose)
            assert ctx.progressive_frames() == frame
            edisp, ergb8 = oracle.progressive_accumulate(accum, got, frame)
            np.testing.assert_array_equal(disp.view(np.uint32), edisp.view(np.uint32))
            assert np.abs(rgb8.astype(int) - ergb8.astype(int)).max() <= 1      # device powf vs glibc powf at a truncation boundary
            assert (rgb8 != ergb8).mean() < 1e-3
    # the pose really is used: yawing the camera changes the image; variants without the pose path refuse
    a = ctx.render_pose(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER), rt.make_pose(yaw=0.0))
    b2 = ctx.render_pose(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER), rt.make_pose(yaw=0.5))
    assert (a != b2).any()
    with pytest.raises(rt.RtError):
        ctx.render_pose(rt.make_params(W, H, 1, 0, variant="lockstep", **rt.scenes.CPU_LAUNCHER), rt.make_pose())


def _synthetic_mesh(kind, rng):
    """Meshes that reach the corners of the traversal code the cat does not."""
    if kind == "three_triangles":          # fewer than 5 triangles: the BVH is a single leaf (root-is-a-leaf path)
        v = rng.uniform(-12, 12, (9, 3)).astype(np.float32)
        t = np.arange(9, dtype=np.int32).reshape(3, 3)
    elif kind == "axis_aligned_quads":     # flat, axis-aligned geometry: zero-thickness boxes, the strict '>' of cpu:156
        vs, ts = [], []
        for k in range(40):
            c = rng.uniform(-15, 15, 3)
            ax = k % 3
            a, b = [(1, 2), (0, 2), (0, 1)][ax]
            e = np.zeros((4, 3)); s = rng.uniform(1, 6, 2)
            e[1, a] = s[0]; e[2, a] = s[0]; e[2, b] = s[1]; e[3, b] = s[1]
            base = len(vs)
            vs += list(np.round(c) + e)     # integer coordinates: rays through corners and edges do occur
            ts += [[base, base + 1, base + 2], [base, base + 2, base + 3]]
        v = np.array(vs, np.float32); t = np.array(ts, np.int32)
    elif kind == "soup":                   # random triangle soup with duplicates and degenerate (zero-area) triangles
        v = rng.uniform(-20, 20, (300, 3)).astype(np.float32)
        t = rng.integers(0, 300, (400, 3)).astype(np.int32)
        t[::50, 1] = t[::50, 0]             # zero-area
        t = np.concatenate([t, t[:20]])     # duplicates: exact t ties, the earliest must win
    elif kind == "deep_strip":             # a long spiral strip: 6000 triangles, deep unbalanced BVH, many big leaves
        n = 3001
        s = np.linspace(0, 1, n)
        ang = 40 * s
        r = 3 + 14 * s
        p0 = np.stack([r * np.cos(ang), -8 + 20 * s, r * np.sin(ang)], 1)
        p1 = p0 + np.array([0, 1.5, 0])
        v = np.empty((2 * n, 3), np.float32); v[0::2] = p0; v[1::2] = p1
        i = np.arange(n - 1) * 2
        t = np.concatenate([np.stack([i, i + 1, i + 2], 1), np.stack([i + 1, i + 3, i + 2], 1)]).astype(np.int32)
    elif kind == "geometric_chain":        # triangles at x = 24 * 0.7^k, each 0.7 the size of the previous: every midpoint split
        k = np.arange(100)                 # peels off two triangles => a BVH far deeper than optimized.cu's s[30]
        x = 24.0 * 0.7 ** k
        c = np.stack([x - 10, np.full(100, -6.0), np.zeros(100)], 1)
        d = rng.uniform(-0.1, 0.1, (100, 3, 3)) * x[:, None, None]
        v = (c[:, None, :] + d).reshape(-1, 3).astype(np.float32)
        t = np.arange(300, dtype=np.int32).reshape(100, 3)
    elif kind == "flat_faces":             # the six faces of an axis-aligned box, 4 x 4 quads each: every leaf of a face is a ZERO-THICKNESS box lying on a face of the
        vs, ts = [], []                    # root box (minimum AND maximum face of every axis; ADVICE r5: half extent 0 on a grid point of the fixed-point nodes) -- which
        lo, hi = np.array([-14.0, -9.0, -12.0]), np.array([10.0, 11.0, 8.0])   # BoundingBox::intersect never hits (strict '>', cpu:156): those triangles are invisible in
        for ax in range(3):                # the reference, through the middle of the face as well as along its rim; 40 ordinary triangles inside give the rays something to hit
            a, b = [(1, 2), (0, 2), (0, 1)][ax]
            for side in (lo, hi):
                for i in range(4):
                    for j in range(4):
                        q = np.zeros((4, 3)); q[:, ax] = side[ax]
                        fa = lo[a] + (hi[a] - lo[a]) * np.array([i, i + 1, i + 1, i]) / 4
                        fb = lo[b] + (hi[b] - lo[b]) * np.array([j, j, j + 1, j + 1]) / 4
                        q[:, a] = fa; q[:, b] = fb
                        base = len(vs); vs += list(q)
                        ts += [[base, base + 1, base + 2], [base, base + 2, base + 3]]
        inner = rng.uniform(-8, 7, (120, 3))
        base = len(vs); vs += list(inner)
        ts += [[base + 3 * k, base + 3 * k + 1, base + 3 * k + 2] for k in range(40)]
        v = np.array(vs, np.float32); t = np.array(ts, np.int32)
    else:
        raise ValueError(kind)
    return v, t


@pytest.mark.parametrize("kind", ["three_triangles", "axis_aligned_quads", "soup", "deep_strip", "geometric_chain", "flat_faces"])
def test_synthetic_meshes_bit_exact(ctx, oracle, kind, monkeypatch):
    """Random / degenerate / deep meshes through the product's own BVH builder: direct lighting bit-exact against the
    oracle (which builds its own BVH from the same arrays), bounces within tolerance, all traversal kernels and the
    bounded-stack / LDS-staged configurations of the work-stack kernel writing the same bits and counting the same work."""
    from raytracinggpu_amd import hostlib
    rng = np.random.default_rng({"three_triangles": 1, "axis_aligned_quads": 2, "soup": 3, "deep_strip": 4, "geometric_chain": 5, "flat_faces": 6}[kind])
    v, t = _synthetic_mesh(kind, rng)
    om = oracle.Mesh.from_arrays(v, t).build_bvh()
    osc = oracle.Scene.preset("cpu", om)
    ctx.scene_upload(rt.scenes.spheres("cpu"), hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6))
    W, H = 320, 200
    exp0, _, cnt0 = osc.render(W, H, 1, 0, want_rgb8=False)
    exp2, _, _ = osc.render(W, H, 2, 2, want_rgb8=False)
    ref0 = None
    for variant in ("wavefront_queue", "wavefront", "lockstep"):
        got = ctx.render(rt.make_params(W, H, 1, 0, variant=variant, **rt.scenes.CPU_LAUNCHER))
        assert values_equal(got[..., :3], exp0[..., :3]).all(), variant
        np.testing.assert_array_equal(got[..., 3], exp0[..., 3])
        got_work = ctx.count_work(rt.make_params(W, H, 1, 0, variant=variant, **rt.scenes.CPU_LAUNCHER))
        assert got_work == {k: cnt0[k] for k in ("rays", "box_tests", "nodes", "tri_tests")}, variant
        ref0 = got if ref0 is None else ref0
    got2 = ctx.render(rt.make_params(W, H, 2, 2, **rt.scenes.CPU_LAUNCHER))
    assert linf(oracle, got2, exp2) <= TOL
    np.testing.assert_array_equal(got2[..., 3], exp2[..., 3])
    mesh = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    for env in ({"RT_TRAVQ_CAP": "128"}, {"RT_TRAVQ_LDS": "16"}, {"RT_TRAVQ_LDS": "8", "RT_TRAVQ_R": "32"}):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        tuned = rt.Context(0)                # the knobs are read once, when a context is created
        tuned.scene_upload(rt.scenes.spheres("cpu"), mesh)
        alt = tuned.render(rt.make_params(W, H, 2, 2, variant="wavefront_queue", **rt.scenes.CPU_LAUNCHER))
        np.testing.assert_array_equal(alt.view(np.uint32), got2.view(np.uint32))
        tuned.close()
        for k in env:
            monkeypatch.delenv(k)


def test_device_mesh_transform_and_refit(ctx, oracle, cat_golden):
    """rt_mesh_transform = the reference's `transform` kernel (global_launcher.cu:340-365) on the uploaded
    vertices + triangle precompute + BVH refit, all on the device.  Against the oracle doing the same on the host
    (or_mesh_transform + or_mesh_refit: same tree, boxes by compute_bbox): bit-exact direct lighting and equal work
    counters on every node layout (pre-order SoA / interleaved, breadth-first), also after a second transform."""
    def rot(ax, a):
        c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
        m = np.eye(3, dtype=np.float32)
        i, j = [(1, 2), (0, 2), (0, 1)][ax]
        m[i, i] = c; m[j, j] = c; m[i, j] = -s; m[j, i] = s
        return m
    om = oracle.Mesh.from_arrays(cat_golden["vertices"], cat_golden["tri_obj_order"]).build_bvh()
    upload(ctx, "cpu", cat_golden)
    W, H = 400, 250
    for R, t in ((rot(1, 0.4) @ rot(0, 0.2), (1.5, -2.0, 0.5)), (rot(2, -0.3), (-3.0, 1.0, 2.0))):
        om.transform(R, t).refit()
        ctx.mesh_transform(R, t)
        osc = oracle.Scene.preset("cpu", om)
        exp, _, cnt = osc.render(W, H, 1, 0, want_rgb8=False)
        for variant in ("wavefront_queue", "wavefront", "lockstep"):
            p = rt.make_params(W, H, 1, 0, variant=variant, **rt.scenes.CPU_LAUNCHER)
            got = ctx.render(p)
            assert values_equal(got[..., :3], exp[..., :3]).all(), variant
            np.testing.assert_array_equal(got[..., 3], exp[..., 3])
            assert ctx.count_work(p) == {k: cnt[k] for k in ("rays", "box_tests", "nodes", "tri_tests")}, variant
        exp2, _, _ = osc.render(W, H, 2, 2, want_rgb8=False)
        got2 = ctx.render(rt.make_params(W, H, 2, 2, **rt.scenes.CPU_LAUNCHER))
        assert linf(oracle, got2, exp2) <= TOL
    # the moved cat is a different picture, and a scene without a mesh accepts the call
    upload(ctx, "cpu", cat_golden)
    still = ctx.render(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER))
    assert (still[..., :3] != exp[..., :3]).any()
    upload(ctx, "spheres", cat_golden)
    ctx.mesh_transform(np.eye(3), (0, 0, 0))


def test_smooth_normals(ctx, oracle, cat_golden):
    """interpolated vertex normals (get_smooth_normal, realtime_render.cu:221-245) replace the flat normal of
    the winning triangle.  Against the oracle's restatement (parity unpinned: the reference programs that do this are
    CUDA-only): bit-exact direct lighting, bounces within tolerance, also after a device-side transform (which moves the
    normals the way the reference's kernel does, translation included)."""
    v = cat_golden["vertices"].astype(np.float64)
    t_obj, t_bvh = cat_golden["tri_obj_order"], cat_golden["tri_bvh_order"]
    fn = np.cross(v[t_obj[:, 1]] - v[t_obj[:, 0]], v[t_obj[:, 2]] - v[t_obj[:, 0]])
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, t_obj[:, k], fn)
    vn = (vn / np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-20)).astype(np.float32)
    om = oracle.Mesh.from_arrays(cat_golden["vertices"], t_obj).set_normals(vn, t_obj).build_bvh()
    osc = oracle.Scene.preset("cpu", om)
    upload(ctx, "cpu", cat_golden)
    W, H = 400, 250
    flat = ctx.render(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER))
    ctx.mesh_set_normals(vn, t_bvh)
    for variant in ("wavefront_queue", "wavefront"):
        got = ctx.render(rt.make_params(W, H, 1, 0, variant=variant, **rt.scenes.CPU_LAUNCHER))
        exp, _, _ = osc.render(W, H, 1, 0, want_rgb8=False)
        assert values_equal(got[..., :3], exp[..., :3]).all(), variant
        np.testing.assert_array_equal(got[..., 3], exp[..., 3])
    assert (got[..., :3] != flat[..., :3]).any()
    got2 = ctx.render(rt.make_params(W, H, 2, 2, **rt.scenes.CPU_LAUNCHER))
    exp2, _, _ = osc.render(W, H, 2, 2, want_rgb8=False)
    assert linf(oracle, got2, exp2) <= TOL
    np.testing.assert_array_equal(got2[..., 3], exp2[..., 3])
    with pytest.raises(rt.RtError):
        ctx.render(rt.make_params(W, H, 1, 0, variant="lockstep", **rt.scenes.CPU_LAUNCHER))
    R = np.array([[0.9553365, 0, 0.29552022], [0, 1, 0], [-0.29552022, 0, 0.9553365]], np.float32)
    om.transform(R, (0.5, 0.25, -0.5)).refit()
    ctx.mesh_transform(R, (0.5, 0.25, -0.5))
    got3 = ctx.render(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER))
    exp3, _, _ = oracle.Scene.preset("cpu", om).render(W, H, 1, 0, want_rgb8=False)
    assert values_equal(got3[..., :3], exp3[..., :3]).all()
    ctx.mesh_set_normals(None, None)
    upload(ctx, "cpu", cat_golden)
    again = ctx.render(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER))
    np.testing.assert_array_equal(again.view(np.uint32), flat.view(np.uint32))


def test_pipelined_frames_on_one_context_equal_lone_frames(ctx, cat_golden):
    """rt_ctx_set_pipelining: frames rendered back to back on one stream into two (and three) alternating buffers, each sub-frame
    following the previous frame's directly, with a consumer of every frame (the tone mapping) on the same stream in between -- every
    frame and every 8-bit image bit for bit what a lone frame gives; one buffer only falls back to the joined form; a chunked frame
    (RT_CHUNK_MPX far below the frame) runs its chunks the same way."""
    import torch
    upload(ctx, "cpu", cat_golden)
    W, H = 640, 360
    st = torch.cuda.Stream()
    rows, _ = rt.interleaved_rows(H, 8, 0, 1)
    ps = [rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER), rt.make_params(W, H, 2, 1, **rt.scenes.CPU_LAUNCHER)]
    refs = [ctx.render(p) for p in ps]
    refs8 = [ctx.render_rgb8(p) for p in ps]
    try:
        ctx.set_pipelining(True)
        for n_buf in (2, 3, 1):
            bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(n_buf)]
            img = [torch.zeros((H * W * 3 + 16,), dtype=torch.uint8, device="cuda:0") for _ in range(7)]
            torch.cuda.synchronize()
            which = []
            for k in range(7):                                   # parameters change on the way (another layout: full fork for that frame)
                w = 1 if k in (3, 4) else 0
                which.append(w)
                ctx.render_device(ps[w], rows, bufs[k % n_buf].data_ptr(), st.cuda_stream)
                ctx.tonemap_device(bufs[k % n_buf].data_ptr(), H * W, img[k].data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            for k in range(7):
                np.testing.assert_array_equal(img[k][:H * W * 3].cpu().numpy().reshape(H, W, 3), refs8[which[k]], err_msg=f"{n_buf} buffers, frame {k}")
            for k in range(7 - n_buf, 7):
                np.testing.assert_array_equal(bufs[k % n_buf].cpu().numpy().view(np.uint32), refs[which[k]].view(np.uint32))
    finally:
        ctx.set_pipelining(False)
    ctx.selfcheck()


def test_pipelining_hazard_the_library_can_see(ctx, cat_golden, tmp_path):
    """rt_ctx_set_pipelining puts an ordering rule on the caller.  The part of it the library can see it now
    enforces: render(A), render(B), tonemap(A -> img), render(A) -- the third frame would start behind the SECOND call and so not wait for the
    tone mapping that still reads A.  The product build takes the full fork for that frame (every image and frame stays exact); the
    -DRT_DEBUG build refuses the call with RT_ERR_INVALID so that a test run shows the sequence breaks the rule."""
    import subprocess
    import sys
    import torch
    upload(ctx, "cpu", cat_golden)
    W, H = 640, 360
    st = torch.cuda.Stream()
    rows, _ = rt.interleaved_rows(H, 8, 0, 1)
    p0, p1 = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER), rt.make_params(W, H, 1, 3, seed=7, **dict(rt.scenes.CPU_LAUNCHER, sigma=0.2))
    ref0, ref1, ref8 = ctx.render(p0), ctx.render(p1), ctx.render_rgb8(p0)
    A, B = (torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2))
    img = torch.zeros((H * W * 3 + 16,), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    try:
        ctx.set_pipelining(True)
        for _ in range(5):                                        # the race, had it been there, is a matter of timing: a few rounds
            ctx.render_device(p0, rows, A.data_ptr(), st.cuda_stream)
            ctx.render_device(p0, rows, B.data_ptr(), st.cuda_stream)
            ctx.tonemap_device(A.data_ptr(), H * W, img.data_ptr(), st.cuda_stream)
            ctx.render_device(p1, rows, A.data_ptr(), st.cuda_stream)     # other pixels: an early start would show in img
            torch.cuda.synchronize()
            np.testing.assert_array_equal(img[:H * W * 3].cpu().numpy().reshape(H, W, 3), ref8)
            np.testing.assert_array_equal(A.cpu().numpy().view(np.uint32), ref1.view(np.uint32))
            np.testing.assert_array_equal(B.cpu().numpy().view(np.uint32), ref0.view(np.uint32))
    finally:
        ctx.set_pipelining(False)
    dbg = os.path.join(os.path.dirname(rt.__file__), "libraytrace_hip_debug.so")
    assert os.path.exists(dbg), "build() compiles the -DRT_DEBUG library"
    script = tmp_path / "break_the_rule.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(rt.__file__))!r})
import numpy as np, torch
import raytracinggpu_amd as rt
g = np.load(rt.scenes.CAT_FIXTURE, allow_pickle=False)
ctx = rt.Context(0)
ctx.scene_upload(rt.scenes.spheres("cpu"), dict(vertices=g["vertices"], indices=g["tri_bvh_order"], bvh_arr10=g["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6))
W, H = 640, 360
st = torch.cuda.Stream()
rows, _ = rt.interleaved_rows(H, 8, 0, 1)
p = rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER)
A, B = (torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2))
img = torch.zeros((H * W * 3 + 16,), dtype=torch.uint8, device="cuda:0")
ctx.set_pipelining(True)
ctx.render_device(p, rows, A.data_ptr(), st.cuda_stream)
ctx.render_device(p, rows, B.data_ptr(), st.cuda_stream)
ctx.tonemap_device(B.data_ptr(), H * W, img.data_ptr(), st.cuda_stream)      # reads B: fine for a frame into A
ctx.render_device(p, rows, A.data_ptr(), st.cuda_stream)
ctx.render_device(p, rows, B.data_ptr(), st.cuda_stream)
ctx.tonemap_device(A.data_ptr(), H * W, img.data_ptr(), st.cuda_stream)      # reads A ...
try:
    ctx.render_device(p, rows, A.data_ptr(), st.cuda_stream)                 # ... which this frame would overwrite without waiting
    print("ACCEPTED")
except rt.RtError as e:
    print("REFUSED", e.code, e)
torch.cuda.synchronize()
""")
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, RT_LIB=dbg), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "REFUSED -1" in r.stdout and "pipelining rule broken" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("mpx", ["0.05", "0.11"])
def test_chunked_frames_equal_unchunked_frames(cat_golden, mpx):
    """launch_render's cut into cache-sized chunks (RT_CHUNK_MPX) never ran for contiguous rows (rt_render, rt_render_rgb8,
    rt_render_async passed one tile of H rows and the chunk unit was scaled by it).  A context created under a small RT_CHUNK_MPX renders
    several chunks per call -- the last one shorter, so its layout differs and the chains re-join -- for contiguous rows (rt_render), for a row
    range, and for interleaved tiles (rt_render_device); every frame bit for bit the unchunked context's."""
    import torch
    W, H = 400, 250
    ps = [rt.make_params(W, H, 1, 3, **rt.scenes.CPU_LAUNCHER), rt.make_params(W, H, 2, 1, **rt.scenes.CPU_LAUNCHER)]
    plain = rt.Context(0)
    upload(plain, "cpu", cat_golden)
    refs = [plain.render(p) for p in ps]
    old = os.environ.get("RT_CHUNK_MPX")
    os.environ["RT_CHUNK_MPX"] = mpx                                # knobs are read when the context is created
    try:
        c = rt.Context(0)
    finally:
        if old is None:
            del os.environ["RT_CHUNK_MPX"]
        else:
            os.environ["RT_CHUNK_MPX"] = old
    upload(c, "cpu", cat_golden)
    for p, ref in zip(ps, refs):
        np.testing.assert_array_equal(c.render(p).view(np.uint32), ref.view(np.uint32))                       # contiguous, whole frame
        np.testing.assert_array_equal(c.render(p, 13, 241).view(np.uint32), ref[13:241].view(np.uint32))        # a row range
        for world in (1, 3):                                                                                 # interleaved 8-row tiles
            for rank in range(world):
                rows, idx = rt.interleaved_rows(H, 8, rank, world)
                buf = torch.empty((rows.n_rows, W, 4), dtype=torch.float32, device="cuda:0")
                c.render_device(p, rows, buf.data_ptr())
                c.synchronize()
                np.testing.assert_array_equal(buf.cpu().numpy().view(np.uint32), ref[idx].view(np.uint32))
    assert c.stats()["pixels"] > 0
    c.selfcheck()
    c.close(); plain.close()


def test_pipelining_randomized_soak():
    """tools/pipeline_stress.py: 250 frames of five sizes / parameter sets back to back with rt_ctx_set_pipelining, buffers rotating 1-3
    deep, two streams, pipelining toggled and asynchronous host frames mixed in, a consumer behind every frame: all images exact."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pipeline_stress.py"), "250", "7"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "differing 0" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("n", [161, 513])
def test_large_mesh_bit_exact(ctx, oracle, n):
    """Meshes well beyond the cat: a displaced grid of 160 x 160 (51 200 triangles) and 512 x 512 quads (524 288 triangles, ~260 000 BVH
    nodes: 25 MB of triangle records and 8 MB of nodes, past the L1s and an XCD's L2; traversal stacks several times deeper).  Direct
    lighting bit-exact against the oracle, equal work counters, bounce frame exact, through the default pipeline and the per-lane walk;
    the device-built tree equals the host's."""
    from raytracinggpu_amd import hostlib
    rng = np.random.default_rng(11)
    gx, gz = np.meshgrid(np.linspace(-18, 18, n), np.linspace(-14, 22, n), indexing="ij")
    gy = -9.0 + 3.0 * np.sin(gx * 0.45) * np.cos(gz * 0.38) + 0.15 * rng.standard_normal((n, n))
    v = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).reshape(-1)
    t = np.concatenate([np.stack([a, a + 1, a + n], 1), np.stack([a + 1, a + n + 1, a + n], 1)]).astype(np.int32)
    assert len(t) == 2 * (n - 1) ** 2
    om = oracle.Mesh.from_arrays(v, t).build_bvh()
    osc = oracle.Scene.preset("cpu", om)
    mesh = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    ctx.scene_upload(rt.scenes.spheres("cpu"), mesh)
    W, H = 384, 216
    exp0, _, cnt0 = osc.render(W, H, 1, 0, want_rgb8=False)
    for variant in ("auto", "wavefront"):
        p0 = rt.make_params(W, H, 1, 0, variant=variant, **rt.scenes.CPU_LAUNCHER)
        got = ctx.render(p0)
        assert values_equal(got[..., :3], exp0[..., :3]).all(), variant
        np.testing.assert_array_equal(got[..., 3], exp0[..., 3])
        assert ctx.count_work(p0) == {k: cnt0[k] for k in ("rays", "box_tests", "nodes", "tri_tests")}, variant
    exp2, _, _ = osc.render(W, H, 2, 3, want_rgb8=False)
    got2 = ctx.render(rt.make_params(W, H, 2, 3, **rt.scenes.CPU_LAUNCHER))
    assert values_equal(got2[..., :3], exp2[..., :3]).all()
    np.testing.assert_array_equal(got2[..., 3], exp2[..., 3])
    tris_up = np.asarray(mesh["indices"])[:, :3]                      # rt_mesh_rebuild: the same tree, built on the device from the uploaded order
    arr, order = ctx.mesh_rebuild(len(tris_up))
    again = hostlib.build_mesh(v, tris_up, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    np.testing.assert_array_equal(arr.view(np.uint32), np.ascontiguousarray(again["bvh_arr10"], np.float32).view(np.uint32))
    assert sorted(order.tolist()) == list(range(len(tris_up)))
    got3 = ctx.render(rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER))
    assert values_equal(got3[..., :3], exp0[..., :3]).all()
    ctx.selfcheck()


def test_batched_frames_are_bitwise_the_lone_frames(ctx, oracle, oracle_cat, cat_golden):
    """rt_render_device_batch (ABI 6): K frames of a rank's share -- interleaved 8-row tiles of rank 3 of 8 at 1920x1080, the share a rank of the 8-GPU job owns -- traced as the
    items of ONE launch chain, each frame with its OWN camera (a dolly along x and z, one with another field of view), its own seed and its own buffer.  Every batched frame is, word
    for word, the frame rt_render_device writes for the scene uploaded with that camera and that seed (per-pixel arithmetic does not depend on what else is in the launch); one of
    them is held against the oracle rendering the same rows with that camera.  Also: a batch of one, the full 16, whole small frames (contiguous rows), and the refusals."""
    import torch
    mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    W, H, b = 1920, 1080, 3
    rows, idx = rt.interleaved_rows(H, 8, 3, 8)
    st = torch.cuda.Stream()
    cams = [((0.5 * k - 1.0, 0.25 * (k % 3), 55.0 - 0.75 * k), None if k != 2 else 1.2, 1000 + 17 * k) for k in range(5)]
    outs = [torch.zeros((rows.n_rows, W, 4), dtype=torch.float32, device="cuda:0") for _ in cams]
    p = rt.make_params(W, H, 1, b, **rt.scenes.CPU_LAUNCHER)
    ctx.scene_upload(rt.scenes.spheres("cpu"), mesh)
    torch.cuda.synchronize()
    ctx.render_device_batch(p, rows, [(o.data_ptr(), c[0], c[1], c[2]) for o, c in zip(outs, cams)], st.cuda_stream)
    st.synchronize()
    got = [o.cpu().numpy() for o in outs]
    assert ctx.stats()["travq_mode"] == 2
    lone = torch.zeros_like(outs[0])
    for k, (pos, fov, seed) in enumerate(cams):
        ctx.scene_upload(rt.scenes.spheres("cpu"), mesh, camera=(pos, fov))
        pk = rt.make_params(W, H, 1, b, seed=seed, **rt.scenes.CPU_LAUNCHER)
        ctx.render_device(pk, rows, lone.data_ptr(), st.cuda_stream)
        st.synchronize()
        np.testing.assert_array_equal(got[k].view(np.uint32), lone.cpu().numpy().view(np.uint32), err_msg=f"frame {k}")
        assert k == 0 or (got[k][..., :3] != got[0][..., :3]).any()     # a sequence of different frames, not one frame K times
    exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(W, H, 1, b, rows=(3 * 8, H), tile_rows=8, tile_step=8, cam=cams[2][0], fov=cams[2][1], seed=cams[2][2], want_rgb8=False)
    np.testing.assert_array_equal(got[2][..., :3].view(np.uint32), exp[..., :3].view(np.uint32))
    # an EVEN number of frames on the same interleaved rows (the two sub-frames then take half of the frames each, not half of the tiles), with pixel jitter (sigma 0.2: the
    # frames' seeds reach the jitter's random numbers too), against the lone frames of the uploaded camera
    ctx.scene_upload(rt.scenes.spheres("cpu"), mesh)
    pj = rt.make_params(W, H, 1, 2, **dict(rt.scenes.CPU_LAUNCHER, sigma=0.2))
    ctx.render_device_batch(pj, rows, [(o.data_ptr(), (0.0, 0.0, 55.0), None, 2000 + k) for k, o in enumerate(outs[:4])], st.cuda_stream)
    st.synchronize()
    for k in range(4):
        ctx.render_device(rt.make_params(W, H, 1, 2, seed=2000 + k, **dict(rt.scenes.CPU_LAUNCHER, sigma=0.2)), rows, lone.data_ptr(), st.cuda_stream)
        st.synchronize()
        np.testing.assert_array_equal(outs[k].cpu().numpy().view(np.uint32), lone.cpu().numpy().view(np.uint32), err_msg=f"jittered frame {k}")
    # a scene without a mesh: the batch keeps the wavefront pipeline (the lock-step kernel that AUTO picks for lone frames of such scenes has no batch form); same frames
    ctx.scene_upload(rt.scenes.spheres("demo10"), None)
    pd = rt.make_params(W, H, 1, 5, **rt.scenes.CPU_LAUNCHER)
    ctx.render_device_batch(pd, rows, [(o.data_ptr(), (0.0, 0.0, 55.0), None, 77) for o in outs[:2]], st.cuda_stream)
    st.synchronize()
    ctx.render_device(rt.make_params(W, H, 1, 5, seed=77, **rt.scenes.CPU_LAUNCHER), rows, lone.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert ctx.stats()["variant"] == 5                                 # the lone frame: lock-step
    for k in range(2):
        np.testing.assert_array_equal(outs[k].cpu().numpy().view(np.uint32), lone.cpu().numpy().view(np.uint32))
    # whole small frames, 16 of them and one alone; then what a batch cannot be
    ctx.scene_upload(rt.scenes.spheres("cpu"), mesh)
    W2, H2 = 256, 144
    rows2, _ = rt.interleaved_rows(H2, 8, 0, 1)
    p2 = rt.make_params(W2, H2, 1, 2, **rt.scenes.CPU_LAUNCHER)
    bufs = [torch.zeros((H2, W2, 4), dtype=torch.float32, device="cuda:0") for _ in range(16)]
    torch.cuda.synchronize()
    ctx.render_device_batch(p2, rows2, [(o.data_ptr(), (0.0, 0.0, 55.0), None, 500 + k) for k, o in enumerate(bufs)], st.cuda_stream)
    st.synchronize()
    for k in (0, 7, 15):
        np.testing.assert_array_equal(bufs[k].cpu().numpy().view(np.uint32), ctx.render(rt.make_params(W2, H2, 1, 2, seed=500 + k, **rt.scenes.CPU_LAUNCHER)).view(np.uint32))
    one = torch.zeros_like(bufs[0])
    ctx.render_device_batch(p2, rows2, [(one.data_ptr(), (0.0, 0.0, 55.0), None, 507)], st.cuda_stream)
    st.synchronize()
    np.testing.assert_array_equal(one.cpu().numpy().view(np.uint32), bufs[7].cpu().numpy().view(np.uint32))
    for bad, code in (([(bufs[0].data_ptr(), (0, 0, 55), None, 1)] * 2, -1),                                   # two frames into one buffer
                      ([(o.data_ptr(), (0, 0, 55), None, 1) for o in bufs] + [(one.data_ptr(), (0, 0, 55), None, 1)], -1),   # 17 frames
                      ([], -1)):
        with pytest.raises(rt.RtError) as e:
            ctx.render_device_batch(p2, rows2, bad, st.cuda_stream)
        assert e.value.code == code
    for kw in (dict(num_rays=2), dict(variant="lockstep"), dict(variant="path")):
        with pytest.raises(rt.RtError) as e:
            q = rt.make_params(W2, H2, kw.get("num_rays", 1), 2, variant=kw.get("variant", "auto"), **rt.scenes.CPU_LAUNCHER)
            ctx.render_device_batch(q, rows2, [(one.data_ptr(), (0, 0, 55), None, 1)], st.cuda_stream)
        assert e.value.code == -5
