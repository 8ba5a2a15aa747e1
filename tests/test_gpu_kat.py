"""The reference's own known answers through the DEVICE primitives (rt_kat_*: same library, same inlined device
functions as the render kernels).  tests/golden/kat.npz was produced by the reference's Sphere::intersect,
BoundingBox::intersect, moller_trumbore and TriangleMesh::intersect (oracle/ref_harness.cpp).  -m gpu.

Bit-exact: hit flags, t and normals.  The counters show that the rare routes are really taken on the device: the
literal divisions behind the error-bounded filters for zero / denormal direction components and for rays through
edges and vertices.
"""
import numpy as np
import pytest

import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
from .conftest import load_golden

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx(cat_golden):
    c = rt.Context(0)
    mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"],
                albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    c.scene_upload(rt.scenes.spheres("cpu"), mesh)
    yield c
    c.close()


def test_square_root_is_correctly_rounded_on_device(ctx):
    """rt_sqrtf: v_sqrt_f32 (1 ulp) + the two-residual fix-up for arguments in [2^-96, inf), the compiler's full sequence for the rest
    (zero, tiny, denormal, negative, inf, NaN) behind a wave vote -- against IEEE-754 (numpy's float32 sqrt is correctly rounded): 4 M
    random bit patterns over the whole range, every power of two with its neighbours, perfect squares +- 1 ulp, and the special values,
    mixed inside waves so that both routes run together."""
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 1 << 32, 4_000_000, dtype=np.uint64).astype(np.uint32)
    e = (np.arange(1, 255, dtype=np.uint32) << 23)
    near = np.concatenate([e, e + 1, e - 1, e + 0x400000, e + 0x7fffff])
    k = rng.integers(1, 1 << 12, 100_000).astype(np.float32)
    sq = (k * k).view(np.uint32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 1.2621774e-29, 1.26217737e-29, 3.4028235e38, -1.0, 1.0, 2.0, 0.25], np.float32).view(np.uint32)
    x = np.concatenate([bits, near, sq, sq + 1, sq - 1, np.tile(special, 64)]).view(np.float32)
    rng.shuffle(x)
    got = ctx.kat_sqrt(x)
    with np.errstate(invalid="ignore"):
        exp = np.sqrt(x)
    nan = np.isnan(exp)
    assert (np.isnan(got) == nan).all()
    assert (got[~nan].view(np.uint32) == exp[~nan].view(np.uint32)).all()
    assert ((x >= 2.0 ** -96) & np.isfinite(x)).sum() > 1_000_000 and ((x < 2.0 ** -96) & (x > 0)).sum() > 100_000


def test_kat_sphere_on_device(ctx):
    g = load_golden("kat.npz")
    got = ctx.kat_sphere(g["sphere_in"])
    exp = g["sphere_out"]
    np.testing.assert_array_equal(got[:, 0], exp[:, 0])
    hit = exp[:, 0] != 0
    np.testing.assert_array_equal(bits(got[hit, 1:5]), bits(exp[hit, 1:5]))
    assert 0.05 < hit.mean() < 0.95


@pytest.mark.parametrize("route", [0, 1, 2, 3])
def test_kat_box_on_device(ctx, route):
    """BoundingBox::intersect incl. u_k = 0 (+-inf / NaN slabs): literal code, slab_filtered, qbox_filter + fall-back,
    and wf_travq's centre / half-extent filter (cbox_filter) + fall-back."""
    g = load_golden("kat.npz")
    rows = g["box_in"]
    got, cnt = ctx.kat_box(rows, route)
    np.testing.assert_array_equal(got, g["box_out"])
    zero_dir = (rows[:, 9:12] == 0).any(axis=1)
    assert zero_dir.sum() > 500
    if route == 0:
        assert cnt["box_decided"] == 0 and cnt["box_literal"] == len(rows)
    else:
        assert cnt["box_decided"] + cnt["box_literal"] == len(rows)
        # every ray with a zero direction component MUST take the literal route (the filter may not decide it) ...
        _, czero = ctx.kat_box(rows[zero_dir], route)
        assert czero["box_literal"] == zero_dir.sum() and czero["box_decided"] == 0
        # ... and the filter decides the bulk of the ordinary ones
        _, cnorm = ctx.kat_box(rows[~zero_dir], route)
        assert cnorm["box_decided"] > 0.5 * (~zero_dir).sum()      # the KAT rows are rich in grazing and degenerate boxes
        print(f"route {route}: decided {cnt['box_decided']}, literal {cnt['box_literal']} of {len(rows)}")


def test_kat_triangle_on_device(ctx):
    """moller_trumbore: vertex / edge / parallel / degenerate rows reach the literal divisions; results bit-exact."""
    g = load_golden("kat.npz")
    got, cnt = ctx.kat_triangle(g["tri_in"])
    exp = g["tri_out"]
    np.testing.assert_array_equal(got[:, 0], exp[:, 0])
    np.testing.assert_array_equal(bits(got[:, 2:5]), bits(exp[:, 2:5]))         # N = e1 x e2 is written for every row
    hit = exp[:, 0] != 0
    np.testing.assert_array_equal(bits(got[hit, 1]), bits(exp[hit, 1]))
    assert (exp[hit, 1] < 1e9).all()                                            # no row beyond the leaf loop's INF (cpu:283)
    assert cnt["tri_decided"] + cnt["tri_literal"] == len(exp)
    assert cnt["tri_literal"] > 0 and cnt["tri_decided"] > cnt["tri_literal"]
    print(f"triangle: filter decided {cnt['tri_decided']}, literal divisions {cnt['tri_literal']}")


@pytest.mark.parametrize("route", [0, 1])
def test_kat_mesh_on_device(ctx, route):
    """TriangleMesh::intersect on the cat: nearest accepted t and its normal, bit-exact, through both device walks."""
    g = load_golden("kat.npz")
    got, cnt = ctx.kat_mesh(g["mesh_in"], 1e-4, route)
    exp = g["mesh_out"]
    np.testing.assert_array_equal(got[:, 0], exp[:, 0])
    hit = exp[:, 0] != 0
    assert hit.sum() > 1500
    np.testing.assert_array_equal(bits(got[hit, 1:5]), bits(exp[hit, 1:5]))
    if route == 0:
        assert cnt["box_decided"] > 0 and cnt["tri_decided"] > 0
        print(f"mesh: box decided/literal {cnt['box_decided']}/{cnt['box_literal']}, triangle {cnt['tri_decided']}/{cnt['tri_literal']}")


def _check_mesh_rows(got, exp):
    np.testing.assert_array_equal(got[:, 0], exp[:, 0])
    hit = exp[:, 0] != 0
    np.testing.assert_array_equal(bits(got[hit, 1:5]), bits(exp[hit, 1:5]))
    return hit


@pytest.mark.parametrize("variant", ["wavefront_queue", "path", "wavefront"])
def test_reference_mesh_vectors_through_the_production_traversal_kernels(ctx, variant):
    """rt_trace_rays: the reference's own TriangleMesh::intersect vectors (tests/golden/kat.npz, 6 600 rays incl. origins inside the
    mesh and grazing rays) written into the traversal queue and traced by the PRODUCTION launches -- wf_travq (stack, leaf queue,
    refill), wf_path, wf_trav (work splitting) -- not by a test kernel's own loop: nearest t and normal bit-exact."""
    g = load_golden("kat.npz")
    got = ctx.trace_rays(g["mesh_in"], 1e-4, variant)
    hit = _check_mesh_rows(got, g["mesh_out"])
    assert hit.sum() > 1500


def test_reference_mesh_vectors_through_the_serial_drain(cat_golden, monkeypatch):
    """The same under RT_TRAVQ_CAP=128 (a context reads its knobs once): the work stack overflows and wf_travq drains popped pairs by
    the serial skip-pointer walk."""
    mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    g = load_golden("kat.npz")
    for qw in ("0", "1"):                                              # the sibling-pair kernel's drain, and the 4-wide step's (a step that predicts an overflow walks its pairs serially)
        monkeypatch.setenv("RT_TRAVQ_CAP", "128"); monkeypatch.setenv("RT_TRAVQ_QW", qw); monkeypatch.setenv("RT_TRAVQ_QW_COUNT", qw)
        c = rt.Context(0)
        monkeypatch.delenv("RT_TRAVQ_CAP"); monkeypatch.delenv("RT_TRAVQ_QW"); monkeypatch.delenv("RT_TRAVQ_QW_COUNT")
        c.scene_upload(rt.scenes.spheres("cpu"), mesh)
        for variant in ("wavefront_queue", "path"):
            _check_mesh_rows(c.trace_rays(g["mesh_in"], 1e-4, variant), g["mesh_out"])
        p = rt.make_params(640, 360, 1, 2, **rt.scenes.CPU_LAUNCHER)
        assert c.stats_after_render(p)["travq_mode"] == (2 if qw == "1" else 0)
        drains = c.count_work(p, detail=True)["steps"]["serial_drains"]
        assert drains > 0, qw                                          # the capacity really forces the drain on this mesh
        c.close()


def test_reference_mesh_vectors_through_the_fixed_point_box_step(cat_golden, monkeypatch):
    """RT_TRAVQ_Q16=1: the BOX step decides on 16-bit fixed-point boxes rounded outwards (a superset of the reference's visits), flags leaves it
    cannot be sure of and lets the reference's test of the leaf's real box decide when a triangle is accepted there (rt_qnodes.hip.h): the
    reference's 6 600 mesh vectors, bit for bit, and the frame of the default kernel."""
    monkeypatch.setenv("RT_TRAVQ_Q16", "1"); monkeypatch.setenv("RT_TRAVQ_QW", "0")
    c = rt.Context(0)
    monkeypatch.delenv("RT_TRAVQ_Q16")
    mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    c.scene_upload(rt.scenes.spheres("cpu"), mesh)
    g = load_golden("kat.npz")
    hit = _check_mesh_rows(c.trace_rays(g["mesh_in"], 1e-4, "wavefront_queue"), g["mesh_out"])
    assert hit.sum() > 1500
    d = rt.Context(0)                                                  # RT_TRAVQ_QW=0 still set: the float sibling pairs
    monkeypatch.delenv("RT_TRAVQ_QW")
    d.scene_upload(rt.scenes.spheres("cpu"), mesh)
    p = rt.make_params(960, 540, 1, 3, **rt.scenes.CPU_LAUNCHER)
    np.testing.assert_array_equal(c.render(p).view(np.uint32), d.render(p).view(np.uint32))
    assert c.stats()["travq_mode"] == 1 and d.stats()["travq_mode"] == 0
    c.close(); d.close()


def test_reference_mesh_vectors_through_the_4_wide_box_step(ctx, cat_golden, monkeypatch):
    """RT_TRAVQ_QW (round 5; the default wherever its node format fits, named here so that the test keeps meaning what it says): the BOX step tests the four boxes two
    levels below a sibling pair's parent on 16-bit fixed-point records rounded outwards, skips every other level of the tree, flags the leaves it hits by less than the
    boxes' enlargement, and a triangle accepted in a flagged leaf meets the reference's test of the leaf's REAL box (rt_travq.hip.h).  The reference's 6 600 mesh vectors bit for bit through the production launches, the frame of
    the float sibling-pair kernel (RT_TRAVQ_QW=0) word for word at b = 3, and rt_stats says which kernel ran."""
    mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    monkeypatch.setenv("RT_TRAVQ_QW", "1")
    c = rt.Context(0)
    monkeypatch.setenv("RT_TRAVQ_QW", "0")
    d = rt.Context(0)
    monkeypatch.delenv("RT_TRAVQ_QW")
    g = load_golden("kat.npz")
    for x in (c, d):
        x.scene_upload(rt.scenes.spheres("cpu"), mesh)
        hit = _check_mesh_rows(x.trace_rays(g["mesh_in"], 1e-4, "wavefront_queue"), g["mesh_out"])
        assert hit.sum() > 1500
    p = rt.make_params(960, 540, 1, 3, **rt.scenes.CPU_LAUNCHER)
    np.testing.assert_array_equal(c.render(p).view(np.uint32), d.render(p).view(np.uint32))
    assert c.stats()["travq_mode"] == 2 and d.stats()["travq_mode"] == 0
    ctx.scene_upload(rt.scenes.spheres("cpu"), mesh)                   # and the default IS the 4-wide step for this tree
    ctx.render(p)
    assert ctx.stats()["travq_mode"] == 2
    c.close(); d.close()


def _degenerate_rays(oracle_cat):
    rng = np.random.default_rng(20260410)
    n = 1200
    O = rng.uniform(-30, 30, (n, 3)).astype(np.float32)
    O[:300] = rng.uniform(-8, 8, (300, 3)).astype(np.float32) + np.float32([0, 5, 0])       # inside / near the cat
    u = rng.normal(size=(n, 3)).astype(np.float32)
    k = rng.integers(0, 3, n)
    u[np.arange(600), k[:600]] = 0.0                                   # one zero component
    u[np.arange(600, 700), k[600:700]] = np.float32(1e-42)             # denormal
    u[np.arange(700, 760), k[700:760]] = np.float32(-0.0)
    u[760:800] = 0.0; u[np.arange(760, 800), k[760:800]] = rng.choice([-1.0, 1.0], 40)      # axis-parallel
    u[800:840] *= np.float32(1e20)                                     # huge, unnormalised
    u[840:880] *= np.float32(1e-20)
    bb = oracle_cat.bvh_array()[0]
    O[880:940, 0] = bb[2]; O[940:1000, 1] = bb[6]                     # on the root box's faces
    return O, u, np.concatenate([O, u], axis=1)


def _check_degenerate(c, oracle_cat, variant):
    O, u, rays = _degenerate_rays(oracle_cat)
    n = len(rays)
    for tmin in (1e-4, 0.0):
        exp = np.zeros((n, 5), np.float32)
        for i in range(n):
            h, t, N = oracle_cat.intersect(O[i], u[i], tmin)
            exp[i, 0] = 1.0 if h else 0.0
            exp[i, 1] = t; exp[i, 2:5] = N
        got = c.trace_rays(rays, tmin, variant)
        hit = _check_mesh_rows(got, exp)
        assert hit.sum() > 50 and (~hit).sum() > 50
    assert ((rays[:, 3:6] == 0).any(axis=1)).sum() >= 500


@pytest.mark.parametrize("variant", ["wavefront_queue", "path", "wavefront"])
def test_degenerate_rays_through_the_production_traversal_kernels(ctx, oracle, oracle_cat, variant):
    """Rays a camera or a bounce never produces -- zero, denormal, huge and axis-parallel direction components, origins inside the
    mesh and on its box faces, unnormalised directions -- against the oracle's TriangleMesh::intersect (pinned to the reference by
    tests/test_oracle_pinned.py): hit flag, t and normal bit-exact through every production traversal kernel.  These rays reach
    wf_travq's c0 = +inf route (literal box tests for every pair) and, with tri_tmin = 0, moller_trumbore's own t > 0."""
    _check_degenerate(ctx, oracle_cat, variant)


def test_degenerate_rays_through_the_fixed_point_box_step(oracle, oracle_cat, cat_golden, monkeypatch):
    """The same rays with RT_TRAVQ_Q16=1: the box test is not monotone for a ray with a zero / denormal / huge component, so such a ray never meets the
    fixed-point pairs -- it is walked serially with the literal test when it is handed its slot (rt_qnodes.hip.h) -- while the rest of the batch does."""
    mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    for env in ({"RT_TRAVQ_Q16": "1", "RT_TRAVQ_QW": "0"}, {"RT_TRAVQ_QW": "1"}, {"RT_TRAVQ_QW": "0"}):   # fixed-point pairs, the 4-wide step (same rule: such rays are walked serially), float pairs
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = rt.Context(0)
        for k in env:
            monkeypatch.delenv(k)
        c.scene_upload(rt.scenes.spheres("cpu"), mesh)
        _check_degenerate(c, oracle_cat, "wavefront_queue")
        c.close()


def _grazing_rays(arr, rng, n):
    """Rays aimed at points ON the faces, edges and corners of leaf boxes (exact float coordinates of the tree's own bounds), from origins all around: the rays for which a box test
    decides by its last bits -- where a fixed-point box rounded outwards and the reference's test of the real box part company.  Half of them are normalised, the others are not."""
    leaves = arr[arr[:, 0] < 0]
    pick = leaves[rng.integers(0, len(leaves), n)]
    lo, hi = pick[:, 2:5], pick[:, 5:8]
    w = rng.integers(0, 3, (n, 3))                                     # per axis: 0 = lo face, 1 = hi face, 2 = somewhere between
    tt = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    target = np.where(w == 0, lo, np.where(w == 1, hi, lo + tt * (hi - lo))).astype(np.float32)
    O = (target + rng.normal(size=(n, 3)).astype(np.float32) * np.float32(25.0)).astype(np.float32)
    u = (target - O).astype(np.float32)
    nrm = np.linalg.norm(u, axis=1, keepdims=True).astype(np.float32)
    u[: n // 2] = (u[: n // 2] / nrm[: n // 2]).astype(np.float32)
    k = rng.integers(0, 3, n // 8)                                     # some exactly axis-parallel ones along a face
    idx = rng.integers(0, n, n // 8)
    u[idx, k] = 0.0
    return np.concatenate([O, u], axis=1).astype(np.float32)


@pytest.mark.parametrize("kind,seed", [("cat", 77), ("soup", 77), ("axis_aligned_quads", 77), ("geometric_chain", 77),
                                       ("axis_aligned_quads", 0), ("axis_aligned_quads", 3), ("axis_aligned_quads", 42), ("flat_faces", 6)])
def test_rays_grazing_leaf_boxes_through_every_box_step_form(oracle, cat_golden, monkeypatch, kind, seed):
    """The 4-wide BOX step and the fixed-point pairs decide internal nodes on boxes rounded OUTWARDS, flag the leaves they hit by less than the boxes' enlargement, and a triangle
    ACCEPTED in a flagged leaf counts only if the reference's own test of the leaf's real box says hit (rt_travq.hip.h).  4 000 rays through points on the faces, edges and corners of
    the tree's own leaf boxes -- the band in which the two box tests disagree -- must give the oracle's TriangleMesh::intersect bit for bit through the production launches of all three
    forms (and the float pairs, which have no such band), and rt_stats must say that the form asked for is the one that ran.
    Seeds 0 / 3 / 42 of the axis-aligned quads and `flat_faces` hold ZERO-THICKNESS leaves on the root box's minimum face (half extent 0 on a grid point of the fixed-point nodes:
    ADVICE r5, q16_axis): the reference never hits such a box (strict '>', cpu:156), however squarely the ray goes through it."""
    from .test_gpu_parity import _synthetic_mesh
    rng = np.random.default_rng(seed)
    if kind == "cat":
        v, t = np.array(cat_golden["vertices"], np.float32), np.array(cat_golden["tri_obj_order"], np.int32)
    else:
        v, t = _synthetic_mesh(kind, rng)
    mesh = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    om = oracle.Mesh.from_arrays(v, t).build_bvh()
    arr = np.asarray(mesh["bvh_arr10"], np.float32).reshape(-1, 10)
    if seed != 77:                                                     # the case the test is there for: a flat leaf lying on the root's minimum face
        leaves = arr[arr[:, 1] < 0]
        assert ((leaves[:, 2:5] == leaves[:, 5:8]) & (leaves[:, 2:5] == arr[0, 2:5])).any()
    rays = _grazing_rays(arr, rng, 4000)
    exp = np.zeros((len(rays), 5), np.float32)
    for i in range(len(rays)):
        h, tt, N = om.intersect(rays[i, :3], rays[i, 3:], 1e-4)
        exp[i, 0] = 1.0 if h else 0.0
        exp[i, 1] = tt; exp[i, 2:5] = N
    for env, want in (({"RT_TRAVQ_QW": "1"}, 2), ({"RT_TRAVQ_QW": "0", "RT_TRAVQ_Q16": "1"}, 1), ({"RT_TRAVQ_QW": "0", "RT_TRAVQ_Q16": "0"}, 0)):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        c = rt.Context(0)
        for k_ in env:
            monkeypatch.delenv(k_)
        c.scene_upload(rt.scenes.spheres("cpu"), mesh)
        hit = _check_mesh_rows(c.trace_rays(rays, 1e-4, "wavefront_queue"), exp)
        assert hit.sum() > 50 and (~hit).sum() > 50, (env, int(hit.sum()))
        # every one of these trees nests and has leaves of at most 127 triangles: each takes the form it was asked for (a silent fall-back to the float pairs would pass the rows above)
        assert c.stats_after_render(rt.make_params(64, 64, 1, 0, **rt.scenes.CPU_LAUNCHER))["travq_mode"] == want, (kind, env)
        c.close()


def test_trace_rays_error_paths_and_empty_scene(ctx):
    assert ctx.trace_rays(np.zeros((0, 6), np.float32)).shape == (0, 5)
    with pytest.raises(rt.RtError):
        ctx.trace_rays(np.zeros((4, 6), np.float32), variant="lockstep")
    fresh = rt.Context(0)
    with pytest.raises(rt.RtError):
        fresh.trace_rays(np.zeros((4, 6), np.float32))
    fresh.scene_upload(rt.scenes.spheres("cpu"), None)                 # no mesh: every ray misses it
    out = fresh.trace_rays(np.ones((5, 6), np.float32))
    assert (out[:, 0] == 0).all()
    fresh.close()


def test_kat_error_paths(ctx):
    with pytest.raises(rt.RtError):
        ctx.kat_box(np.zeros((4, 12), np.float32), 7)
    fresh = rt.Context(0)
    with pytest.raises(rt.RtError):
        fresh.kat_mesh(np.zeros((4, 6), np.float32))
    fresh.close()
    out, cnt = ctx.kat_box(np.zeros((0, 12), np.float32), 1)
    assert out.shape == (0,) and cnt["n"] == 0


@pytest.mark.parametrize("light,ball", [((-10.0, 20.0, 40.0), None), ((0.0, 6.0, 9.0), None), ((3.0, -9.5, 12.0), None), ((0.0, 3e18, 0.0), None), ((-10.0, 20.0, 40.0), ((-6.0, 9.0, 24.0), 5.0))],
                         ids=["reference", "beside_the_cat", "on_the_floor", "far_away", "ball_before_the_cat"])
def test_shadow_rays_stop_at_the_first_hit_that_certainly_shades(oracle, cat_golden, monkeypatch, light, ball):
    """Any-hit (round 6): cpu:615 looks at nothing but whether the shadow ray's NEAREST hit lies before the light, a comparison that is monotone in t -- so the fixed-point
    traversal kernels stop a shadow ray at the first accepted triangle whose t is certainly inside (wf_anyhit_bound) instead of tracing it to the end.  For four lights -- the
    reference's, one beside the cat (many triangles BEHIND the light: they must not shade), one a hand above the floor, one 3e18 away (outside the room: the ceiling shades everything) -- the frames are word for word those of
    a context that never stops early (RT_TRAVQ_ANYHIT=0) through the 4-wide step and through the fixed-point pairs, the direct-lighting frame is the oracle's bit for bit, and
    the step counters say the rays did stop (and never do with the knob off, nor in the float-pair kernel whose work counters are the oracle's).  Fifth case: a ball between
    the light and the cat -- shadow rays the ball shades already are not traced through the mesh at all with any-hit on (the reference-equivalent work counters still count them)."""
    mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    monkeypatch.setenv("RT_TRAVQ_QW_COUNT", "1")                       # rt_count_work reports the production kernel's own steps
    on = rt.Context(0)
    monkeypatch.setenv("RT_TRAVQ_QW", "0"); monkeypatch.setenv("RT_TRAVQ_Q16", "1")
    pairs = rt.Context(0)
    monkeypatch.delenv("RT_TRAVQ_QW"); monkeypatch.delenv("RT_TRAVQ_Q16")
    monkeypatch.setenv("RT_TRAVQ_ANYHIT", "0")
    off = rt.Context(0)
    monkeypatch.delenv("RT_TRAVQ_ANYHIT"); monkeypatch.delenv("RT_TRAVQ_QW_COUNT")
    W, H = 480, 270
    spheres = list(rt.scenes.spheres("cpu")) + ([(ball[0], ball[1], (0.6, 0.5, 0.4))] if ball else [])
    for c in (on, pairs, off):
        c.scene_upload(spheres, mesh, light=(light, 3e10))
    om = oracle.Mesh.from_arrays(cat_golden["vertices"], cat_golden["tri_obj_order"]).build_bvh()
    osc = oracle.Scene.preset("cpu", om); osc.set_light(light, 3e10)
    if ball:
        osc.add_sphere(ball[0], ball[1], (0.6, 0.5, 0.4))               # object 7, after the walls (0-5) and the mesh (6): the order scene_upload gives it too
    exp0, _, _ = osc.render(W, H, 1, 0, want_rgb8=False)
    p0 = rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER)
    p3 = rt.make_params(W, H, 2, 3, **rt.scenes.CPU_LAUNCHER)
    ref3 = off.render(p3)
    for c in (on, pairs):
        got0 = c.render(p0)
        np.testing.assert_array_equal(got0.view(np.uint32), off.render(p0).view(np.uint32))
        assert ((got0[..., :3] == exp0[..., :3]) | (np.isnan(got0[..., :3]) & np.isnan(exp0[..., :3]))).all()
        np.testing.assert_array_equal(got0[..., 3], exp0[..., 3])
        np.testing.assert_array_equal(c.render(p3).view(np.uint32), ref3.view(np.uint32))
    assert on.stats()["travq_mode"] == 2 and pairs.stats()["travq_mode"] == 1 and off.stats()["travq_mode"] == 2
    s_on, s_off = on.count_work(p3, detail=True)["steps"], off.count_work(p3, detail=True)["steps"]
    assert s_off["anyhit_stop_steps"] == 0
    if ball:                                                            # the reference's work on this scene, whatever the product skips
        plain = rt.Context(0)                                           # (the knobs are read when a context is created: this one counts the reference-equivalent traversal)
        plain.scene_upload(spheres, mesh, light=(light, 3e10))
        _, _, cnt = osc.render(W, H, 1, 1, want_rgb8=False)
        assert plain.count_work(rt.make_params(W, H, 1, 1, **rt.scenes.CPU_LAUNCHER)) == {k: cnt[k] for k in ("rays", "box_tests", "nodes", "tri_tests")}
        np.testing.assert_array_equal(plain.render(p3).view(np.uint32), ref3.view(np.uint32))
        plain.close()
        assert s_on["fetches"] > 0 and s_on["box_steps"] < s_off["box_steps"]
    if light[1] > 1e6:                                                  # a light outside the room: the ceiling shades every shadow ray, none is traced through the mesh
        assert s_on["anyhit_stop_steps"] == 0 and s_on["box_steps"] < s_off["box_steps"] and s_on["tri_steps"] < s_off["tri_steps"]
    else:
        assert s_on["anyhit_stop_steps"] > 0 and s_on["tri_steps"] < s_off["tri_steps"]
    for c in (on, pairs, off):
        c.close()


def test_any_hit_changes_no_bit_on_random_scenes(monkeypatch):
    """Differential run of the any-hit rule: twelve random scenes -- the synthetic meshes of test_gpu_parity (single-leaf trees, zero-thickness boxes, triangle soups, deep
    chains, flat faces), a light anywhere in the room including inside the mesh's box and a hair above a wall, sometimes a ball in the way, sometimes a mirror or glass mesh
    (then few shadow rays leave it) -- each rendered at 2 samples, 3 bounces by a default context and by one created under RT_TRAVQ_ANYHIT=0: every word of every frame equal."""
    from .test_gpu_parity import _synthetic_mesh
    on = rt.Context(0)
    monkeypatch.setenv("RT_TRAVQ_ANYHIT", "0")
    off = rt.Context(0)
    monkeypatch.delenv("RT_TRAVQ_ANYHIT")
    rng = np.random.default_rng(20261005)
    kinds = ["three_triangles", "axis_aligned_quads", "soup", "deep_strip", "geometric_chain", "flat_faces"]
    p = rt.make_params(256, 144, 2, 3, **rt.scenes.CPU_LAUNCHER)
    differing_lights, modes = 0, []
    for k in range(12):
        v, t = _synthetic_mesh(kinds[k % 6], np.random.default_rng(100 + k))
        mesh = hostlib.build_mesh(v, t, albedo=(0.3, 0.4, 0.5), object_slot=6)
        if k % 4 == 3:
            mesh.update(mirror=int(k % 8 == 3), in_refraction_index=1.0 if k % 8 == 3 else 1.4, out_refraction_index=1.0)
        light = [(-10.0, 20.0, 40.0), tuple(rng.uniform(-8, 8, 3)), (float(rng.uniform(-20, 20)), -9.99, float(rng.uniform(0, 30))), tuple(v[rng.integers(len(v))] + np.float32(0.01))][k % 4]
        spheres = list(rt.scenes.spheres("cpu")) + ([(tuple(rng.uniform(-10, 10, 3)), float(rng.uniform(1, 4)), (0.5, 0.5, 0.5))] if k % 3 == 0 else [])
        for c in (on, off):
            c.scene_upload(spheres, mesh, light=(tuple(float(x) for x in light), 3e10))
        a, b = on.render(p), off.render(p)
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=f"scene {k}: {kinds[k % 6]}, light {light}")
        differing_lights += int(np.isfinite(a[..., :3]).all())
        modes.append(on.stats()["travq_mode"])
    print("traversal kernel form per scene (2 = 4-wide fixed-point step, 1 = fixed-point pairs, 0 = float pairs: no any-hit there):", modes)
    assert differing_lights >= 8                                        # (a light ON a vertex may give inf / NaN pixels in both: still equal word for word)
    assert sum(m >= 1 for m in modes) >= 8                              # the rule was in play: only the single-leaf trees run the float pairs
    on.close(); off.close()
