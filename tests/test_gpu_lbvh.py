"""rt_mesh_rebuild_mode(RT_BVH_LBVH) (SURVEY 8f3 "and a GPU LBVH build"): a DIFFERENT tree than the reference's,
built on the device in parallel -- Morton sort, binary radix tree, leaves cut by the surface-area heuristic (at most 32 triangles).  Parity is "HIP == oracle ON THE
SAME TREE": the tree and the triangle order the device returns are handed to the oracle (or_mesh_set_bvh), whose traversal
(cpu_launcher.cpp:277-311 restated) then walks it; frames must be bit-identical and the work counters equal.  Against the REFERENCE tree
the image may differ only where two triangles are hit at bit-equal t (the scan order breaks the tie); counted and bounded.
-m gpu."""
import numpy as np
import pytest

import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
from .test_gpu_parity import _synthetic_mesh, values_equal

pytestmark = pytest.mark.gpu
KEYS = ("rays", "box_tests", "nodes", "tri_tests")


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    yield c
    c.close()


def _check_tree(arr, order, n_tris):
    """A proper tree in the reference's flat layout: every node reachable once, leaves of 1..32 triangles covering [0, n) exactly."""
    n = len(arr)
    assert sorted(order.tolist()) == list(range(n_tris))
    seen = np.zeros(n, bool)
    covered = np.zeros(n_tris, np.int32)
    stack = [0]
    leaves = 0
    while stack:
        k = stack.pop()
        assert 0 <= k < n and not seen[k]
        seen[k] = True
        l, r, s, e = int(arr[k, 0]), int(arr[k, 1]), int(arr[k, 8]), int(arr[k, 9])
        assert 0 <= s < e <= n_tris
        assert (arr[k, 2:5] <= arr[k, 5:8]).all()
        if l < 0:
            assert r < 0 and e - s <= 32
            covered[s:e] += 1
            leaves += 1
        else:
            assert e - s > 2
            for c in (l, r):                                            # children nest inside the parent, ranges partition the parent's
                assert (arr[c, 2:5] >= arr[k, 2:5]).all() and (arr[c, 5:8] <= arr[k, 5:8]).all()
            assert {int(arr[l, 8]), int(arr[r, 8])} >= {s} and {int(arr[l, 9]), int(arr[r, 9])} >= {e}
            assert int(arr[l, 9]) - int(arr[l, 8]) + int(arr[r, 9]) - int(arr[r, 8]) == e - s
            stack += [l, r]
    assert seen.all() and (covered == 1).all()
    return leaves


def _lbvh_against_oracle(ctx, oracle, v, tris_uploaded, W, H, bounces=(0, 2), counters=True):
    nt = len(tris_uploaded)
    arr, order = ctx.mesh_rebuild(nt, mode="lbvh")
    st = ctx.build_stats()
    om = oracle.Mesh.from_arrays(v, tris_uploaded)
    if nt <= 4:                                                          # a single leaf in either mode: the reference builder ran
        assert st["mode"] == 0 and len(arr) == 1
    else:
        assert st["mode"] == 1 and st["n_nodes"] == len(arr) and st["max_leaf_tris"] <= 32 and st["n_triangles"] == nt
        assert _check_tree(arr, order, nt) == st["n_leaves"]
    om.set_bvh(arr, order)
    osc = oracle.Scene.preset("cpu", om)
    for b in bounces:
        exp, _, cnt = osc.render(W, H, 1, b, want_rgb8=False)
        for variant in ("auto", "wavefront", "lockstep") if b == 0 else ("auto",):
            p = rt.make_params(W, H, 1, b, variant=variant, **rt.scenes.CPU_LAUNCHER)
            got = ctx.render(p)
            assert values_equal(got[..., :3], exp[..., :3]).all(), (b, variant)   # sigma == 0: every channel bit-identical
            np.testing.assert_array_equal(got[..., 3], exp[..., 3])
            if counters:
                assert ctx.count_work(p) == {k: cnt[k] for k in KEYS}, (b, variant)
    return arr, order, st


@pytest.mark.parametrize("kind", ["cat", "three_triangles", "axis_aligned_quads", "soup", "deep_strip", "geometric_chain"])
def test_lbvh_frame_and_work_counters_equal_the_oracle_on_the_same_tree(ctx, oracle, cat_golden, kind):
    if kind == "cat":
        v, t = np.array(cat_golden["vertices"], np.float32), np.array(cat_golden["tri_obj_order"], np.int32)
    else:
        rng = np.random.default_rng({"three_triangles": 1, "axis_aligned_quads": 2, "soup": 3, "deep_strip": 4, "geometric_chain": 5}[kind])
        v, t = _synthetic_mesh(kind, rng)
    first = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)        # what a caller uploads: the reference tree
    ctx.scene_upload(rt.scenes.spheres("cpu"), first)
    up = np.ascontiguousarray(first["indices"][:, :3])
    W, H = (640, 360) if kind == "cat" else (320, 200)
    p0 = rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER)
    on_reference_tree = ctx.render(p0)
    work_ref = ctx.count_work(p0)
    arr, order, st = _lbvh_against_oracle(ctx, oracle, v, up, W, H, bounces=(0, 3) if kind == "cat" else (0, 2))
    on_lbvh = ctx.render(p0)
    # the two trees hold the same triangles: the images agree except where the nearest hit is a bit-exact tie between two triangles
    diff = (~values_equal(on_lbvh[..., :3], on_reference_tree[..., :3])).any(-1)
    print(f"{kind}: {len(up)} triangles, LBVH {st['n_nodes']} nodes / {st['n_leaves']} leaves (max {st['max_leaf_tris']}, depth {st['max_depth']}), build {st['device_build_ms']:.2f} ms + "
          f"install {st['install_ms']:.1f} ms; pixels differing from the reference tree's image: {int(diff.sum())}; work per frame {ctx.count_work(p0)} vs reference tree {work_ref}")
    np.testing.assert_array_equal(on_lbvh[..., 3], on_reference_tree[..., 3])            # rays per pixel do not depend on the tree
    limit = {"soup": 0.05, "axis_aligned_quads": 0.10}.get(kind, 0.001)                   # duplicated triangles / coplanar quads sharing a diagonal tie on purpose
    assert diff.mean() <= limit, (kind, int(diff.sum()))
    # a second LBVH build starts from the order the first one left and gives the same frame; the reference builder still works afterwards
    _lbvh_against_oracle(ctx, oracle, v, up[order], W, H, bounces=(0,), counters=False)
    arr2, order2 = ctx.mesh_rebuild(len(up), mode="reference")
    assert ctx.build_stats()["mode"] == 0
    np.testing.assert_array_equal(ctx.render(p0)[..., 3], on_reference_tree[..., 3])


def test_lbvh_after_device_transform_and_refit(ctx, oracle, cat_golden):
    """LBVH, then the device-side transform with its bottom-up refit (global_launcher.cu:340-365): the tree's topology stays, the boxes move;
    the oracle does the same to the same tree."""
    v, t = np.array(cat_golden["vertices"], np.float32), np.array(cat_golden["tri_obj_order"], np.int32)
    first = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    ctx.scene_upload(rt.scenes.spheres("cpu"), first)
    up = np.ascontiguousarray(first["indices"][:, :3])
    arr, order = ctx.mesh_rebuild(len(up), mode="lbvh")
    c, s = np.float32(np.cos(0.4)), np.float32(np.sin(0.4))
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    tr = (1.5, -0.5, 2.0)
    ctx.mesh_transform(R, tr)
    om = oracle.Mesh.from_arrays(v, up)
    om.set_bvh(arr, order)
    om.transform(R, tr).refit()
    W, H = 400, 250
    exp, _, cnt = oracle.Scene.preset("cpu", om).render(W, H, 1, 1, want_rgb8=False)
    p = rt.make_params(W, H, 1, 1, **rt.scenes.CPU_LAUNCHER)
    got = ctx.render(p)
    assert values_equal(got[..., :3], exp[..., :3]).all()
    assert ctx.count_work(p) == {k: cnt[k] for k in KEYS}


@pytest.mark.parametrize("n", [513])
def test_lbvh_large_mesh_bit_exact_and_much_less_work(ctx, oracle, n, monkeypatch):
    """524 288 triangles (the displaced grid of test_large_mesh_bit_exact): frame and work counters == the oracle on the LBVH tree; the
    triangle tests per ray fall by an order of magnitude against the reference's tree (whose leaves grow with the mesh, cpu:217)."""
    v, t = _displaced_grid(n)
    first = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    ctx.scene_upload(rt.scenes.spheres("cpu"), first)
    W, H = 480, 270
    p0 = rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER)
    work_ref = ctx.count_work(p0)
    img_ref = ctx.render(p0)
    up = np.ascontiguousarray(first["indices"][:, :3])
    arr, order, st = _lbvh_against_oracle(ctx, oracle, v, up, W, H, bounces=(0, 2))
    work = ctx.count_work(p0)
    diff = (~values_equal(ctx.render(p0)[..., :3], img_ref[..., :3])).any(-1)
    print(f"{len(up)} triangles: LBVH {st['n_nodes']} nodes, depth {st['max_depth']}, build {st['device_build_ms']:.2f} ms + install {st['install_ms']:.0f} ms; "
          f"triangle tests per ray {work['tri_tests'] / work['rays']:.1f} (reference tree {work_ref['tri_tests'] / work_ref['rays']:.1f}), "
          f"box tests per ray {work['box_tests'] / work['rays']:.1f} ({work_ref['box_tests'] / work_ref['rays']:.1f}); tie pixels {int(diff.sum())}")
    assert work["tri_tests"] * 2 < work_ref["tri_tests"]
    assert diff.mean() <= 0.001
    assert st["device_build_ms"] < 50.0
    # the 4-wide kernel's OWN counting instantiation on this tree (RT_TRAVQ_QW_COUNT=1: every index that reaches an address is checked, rt_count_work fails if one is out of range):
    # the rays it retires are the frame's, its BOX steps are four boxes wide (fewer than a quarter of the reference-equivalent box tests per 64 lanes), nothing was walked serially
    # (with shadow rays traced to the end, RT_TRAVQ_ANYHIT=0: it enters a superset of the reference's leaves; with any-hit, the default, it queues fewer triangles than that)
    monkeypatch.setenv("RT_TRAVQ_QW_COUNT", "1")
    owns = {}
    for anyhit in ("0", "1"):
        monkeypatch.setenv("RT_TRAVQ_ANYHIT", anyhit)
        c2 = rt.Context(0)
        c2.scene_upload(rt.scenes.spheres("cpu"), first)
        c2.mesh_rebuild(len(up), mode="lbvh")
        owns[anyhit] = c2.count_work(p0, detail=True)
        c2.close()
    monkeypatch.delenv("RT_TRAVQ_ANYHIT")
    own = owns["0"]
    assert own["rays"] == work["rays"] and own["tri_tests"] >= work["tri_tests"]
    assert 0 < own["steps"]["box_steps"] * 64 * 4 < 2 * work["box_tests"] and own["steps"]["serial_drains"] == 0
    assert owns["1"]["rays"] == work["rays"] and owns["1"]["tri_tests"] <= own["tri_tests"] and owns["1"]["steps"]["serial_drains"] == 0


def _displaced_grid(n, seed=11):
    rng = np.random.default_rng(seed)
    gx, gz = np.meshgrid(np.linspace(-18, 18, n), np.linspace(-14, 22, n), indexing="ij")
    gy = -9.0 + 3.0 * np.sin(gx * 0.45) * np.cos(gz * 0.38) + 0.15 * rng.standard_normal((n, n))
    v = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    a = (i * n + j).reshape(-1)
    t = np.concatenate([np.stack([a, a + 1, a + n], 1), np.stack([a + 1, a + n + 1, a + n], 1)]).astype(np.int32)
    return v, t


@pytest.mark.parametrize("qw", ["default", "0"])
def test_lbvh_two_million_triangles_node_indices_beyond_2_pow_20(oracle, monkeypatch, qw):
    """2 097 152 triangles (n = 1025): the LBVH has more than 2^20 nodes, so stack entries of wf_travq carry node indices that need the top bits of their
    22-bit field (kQNodeShift = 10: the lowest node bit shares bit 10 with the seventh slot bit).  Twice: with the BOX step the library chooses (the 4-wide
    step on fixed-point quads: the tree has fewer than 2^21 nodes) and with RT_TRAVQ_QW=0 (the 16-bit fixed-point sibling pairs, automatic from 16 384
    nodes).  Frame and work counters (binary instantiation) == the oracle walking the SAME tree at a small frame, and explicit rays aimed at
    triangles all over the mesh -- in a breadth-first array most leaves sit at the deepest levels, i.e. at the highest indices -- through the production
    traversal launches (rt_trace_rays) against the oracle's TriangleMesh::intersect.  Reference twin of the builder: global_launcher.cu:298-331."""
    if qw != "default":
        monkeypatch.setenv("RT_TRAVQ_QW", qw)
    ctx = rt.Context(0)                                   # knobs are read when the context is created
    v, t = _displaced_grid(1025)
    first = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    ctx.scene_upload(rt.scenes.spheres("cpu"), first)
    up = np.ascontiguousarray(first["indices"][:, :3])
    arr, order = ctx.mesh_rebuild(len(up), mode="lbvh")
    st = ctx.build_stats()
    assert ctx.stats_after_render(rt.make_params(64, 64, 1, 0, **rt.scenes.CPU_LAUNCHER))["travq_mode"] == (2 if qw == "default" else 1)
    print(f"{len(up)} triangles: LBVH {st['n_nodes']} nodes (2^20 = {1 << 20}, 2^21 = {1 << 21}), depth {st['max_depth']}, {st['n_leaves']} leaves")
    assert st["mode"] == 1 and st["n_nodes"] == len(arr) and st["n_nodes"] > (1 << 20)
    om = oracle.Mesh.from_arrays(v, up)
    om.set_bvh(arr, order)
    osc = oracle.Scene.preset("cpu", om)
    W, H = 320, 180
    for b in (0, 2):
        exp, _, cnt = osc.render(W, H, 1, b, want_rgb8=False)
        p = rt.make_params(W, H, 1, b, **rt.scenes.CPU_LAUNCHER)
        got = ctx.render(p)
        assert values_equal(got[..., :3], exp[..., :3]).all(), b
        np.testing.assert_array_equal(got[..., 3], exp[..., 3])
        assert ctx.count_work(p) == {k: cnt[k] for k in KEYS}, b
    # explicit rays: from above and from the sides towards triangles drawn uniformly from the uploaded order (every part of the node array), plus rays that miss
    rng = np.random.default_rng(5)
    n = 3000
    tri = up[order[rng.integers(0, len(up), n)]]
    target = v[tri].mean(axis=1) + rng.normal(scale=0.002, size=(n, 3)).astype(np.float32)
    O = (target + np.float32([0, 25, 0]) + rng.uniform(-12, 12, (n, 3))).astype(np.float32)
    u = (target - O).astype(np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True).astype(np.float32)
    u[-300:] = rng.normal(size=(300, 3)).astype(np.float32)                    # arbitrary directions (unnormalised): most miss
    rays = np.concatenate([O, u], axis=1).astype(np.float32)
    exp = np.zeros((n, 5), np.float32)
    for i in range(n):
        h, tt, N = om.intersect(rays[i, :3], rays[i, 3:], 1e-4)
        exp[i, 0] = 1.0 if h else 0.0
        exp[i, 1] = tt; exp[i, 2:5] = N
    got = ctx.trace_rays(rays, 1e-4, "wavefront_queue")
    hit = exp[:, 0] != 0
    np.testing.assert_array_equal(got[:, 0], exp[:, 0])
    np.testing.assert_array_equal(got[hit].view(np.uint32), exp[hit].view(np.uint32))
    assert hit.sum() > 2000 and (~hit).sum() > 50
    ctx.close()


def test_lbvh_device_install_equals_host_install(oracle, cat_golden, monkeypatch):
    """The render kernels' formats follow from the builder's arrays ON THE DEVICE (closed forms: subtree sizes from a prefix sum of leaf starts,
    traversal order from one walk up per node, visit ranks from the leaf ranges, breadth-first pairs from a sort by (depth, path)).  A context
    under RT_LBVH_HOST_INSTALL=1 takes the old road -- read the flat tree back, re-lay it out on the host as rt_scene_upload does, upload --
    and must render the same bits with the same work counters through every traversal kernel; smooth normals (host-side tables) and a later
    reference-mode rebuild (which starts from the host copies of the orders) still work after a device-side install."""
    v, t = np.array(cat_golden["vertices"], np.float32), np.array(cat_golden["tri_obj_order"], np.int32)
    first = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    up = np.ascontiguousarray(first["indices"][:, :3])
    dev = rt.Context(0)
    monkeypatch.setenv("RT_LBVH_HOST_INSTALL", "1")
    host = rt.Context(0)
    monkeypatch.delenv("RT_LBVH_HOST_INSTALL")
    outs = []
    for c in (dev, host):
        c.scene_upload(rt.scenes.spheres("cpu"), first)
        outs.append(c.mesh_rebuild(len(up), mode="lbvh"))
    assert dev.build_stats()["install_on_device"] == 1 and host.build_stats()["install_on_device"] == 0
    np.testing.assert_array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))     # the same tree ...
    np.testing.assert_array_equal(outs[0][1], outs[1][1])                                     # ... and order
    W, H = 400, 250
    for variant in ("auto", "wavefront", "path", "lockstep", "lds_top"):
        for b in (0, 2):
            p = rt.make_params(W, H, 1, b, variant=variant, **rt.scenes.CPU_LAUNCHER)
            np.testing.assert_array_equal(dev.render(p).view(np.uint32), host.render(p).view(np.uint32), err_msg=f"{variant} b={b}")
            if b == 0:
                assert dev.count_work(p) == host.count_work(p), variant
    # device-side transform + refit on the device-installed layouts (levels, left children) == the same on the host-installed ones
    c_, s_ = np.float32(np.cos(0.3)), np.float32(np.sin(0.3))
    R = np.array([[c_, 0, s_], [0, 1, 0], [-s_, 0, c_]], np.float32)
    for c in (dev, host):
        c.mesh_transform(R, (0.5, 0.25, -1.0))
    p = rt.make_params(W, H, 1, 1, **rt.scenes.CPU_LAUNCHER)
    np.testing.assert_array_equal(dev.render(p).view(np.uint32), host.render(p).view(np.uint32))
    # smooth normals need the host copy of the visit order: fetched from the device on demand
    nrm = np.random.default_rng(5).normal(size=(len(v), 3)).astype(np.float32)
    nidx = up[outs[0][1]]                                                                      # per-triangle normal indices = vertex indices, in the NEW uploaded order
    for c in (dev, host):
        c.mesh_set_normals(nrm, nidx)
    np.testing.assert_array_equal(dev.render(p).view(np.uint32), host.render(p).view(np.uint32))
    for c in (dev, host):
        c.mesh_set_normals(None, None)
    # and a reference-mode rebuild afterwards starts from the right order on both
    a0, o0 = dev.mesh_rebuild(len(up), mode="reference")
    a1, o1 = host.mesh_rebuild(len(up), mode="reference")
    np.testing.assert_array_equal(a0.view(np.uint32), a1.view(np.uint32))
    np.testing.assert_array_equal(o0, o1)
    np.testing.assert_array_equal(dev.render(p).view(np.uint32), host.render(p).view(np.uint32))
    dev.close(); host.close()


@pytest.mark.parametrize("kind", ["cat", "deep_strip", "geometric_chain", "lbvh_524288"])
def test_node_layouts_are_a_function_of_the_tree_alone(oracle, cat_golden, kind):
    """The quads of the 4-wide BOX step hold the cut a surface-area DP picks ON THE DEVICE (rt_qnodes.hip.h: one thread per leaf climbing, the second arrival at a node computes it
    from values the first one published behind a fence).  Two contexts given the same tree must end with byte-identical layouts -- float pairs, fixed-point pairs, quads, leaf boxes --
    so that the work a frame does (BOX steps, and with them the headline's ms) does not vary from upload to upload; and a second upload into the SAME context reproduces them too.
    deep_strip / geometric_chain: chains far deeper than a wave is wide (the climb's guard is the tree's size, not a constant)."""
    if kind == "cat":
        mesh = dict(vertices=cat_golden["vertices"], indices=cat_golden["tri_bvh_order"], bvh_arr10=cat_golden["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    elif kind == "lbvh_524288":
        v, t = _displaced_grid(513)
        mesh, first = None, hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    else:
        v, t = _synthetic_mesh(kind, np.random.default_rng(4))
        mesh = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    hashes = []
    for rep in range(2):
        c = rt.Context(0)
        if mesh is None:                                               # the LBVH builder's tree (Morton sort + parallel hierarchy + SAH leaf cut, all on the device) and its device-side install
            c.scene_upload(rt.scenes.spheres("cpu"), first)
            c.mesh_rebuild(len(t), mode="lbvh")
            hashes.append(c.layout_hash())
        else:
            c.scene_upload(rt.scenes.spheres("cpu"), mesh)
            hashes.append(c.layout_hash())
            c.scene_upload(rt.scenes.spheres("cpu"), mesh)
            hashes.append(c.layout_hash())
        c.close()
    assert all(h == hashes[0] for h in hashes), hashes
    assert hashes[0]["quads"] != 0 and hashes[0]["fixed_pairs"] != 0 and hashes[0]["pairs"] != 0
