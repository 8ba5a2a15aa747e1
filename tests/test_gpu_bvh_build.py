"""rt_mesh_rebuild: the reference's buildBVH (cpu_launcher.cpp:190-224) as a level-by-level device build.
The device tree must equal the host builder's (include/raytracer.hpp buildFlatBVH, itself pinned to the reference's tree
bit for bit in tests/test_host_api.py): boxes, bvhTreeToArray numbering and the order the partition leaves the triangles in.
-m gpu."""
import numpy as np
import pytest

import raytracinggpu_amd as rt
from raytracinggpu_amd import hostlib
from .test_gpu_parity import _synthetic_mesh, values_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    yield c
    c.close()


def _check_rebuild(ctx, v, tris_uploaded):
    """Device build from the uploaded order == host build from the same order."""
    arr, order = ctx.mesh_rebuild(len(tris_uploaded))
    exp = hostlib.build_mesh(v, tris_uploaded, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    np.testing.assert_array_equal(arr.view(np.uint32), np.ascontiguousarray(exp["bvh_arr10"], np.float32).view(np.uint32))
    assert sorted(order.tolist()) == list(range(len(tris_uploaded)))
    np.testing.assert_array_equal(np.asarray(tris_uploaded)[order], exp["indices"][:, :3])
    return exp


@pytest.mark.parametrize("kind", ["cat", "three_triangles", "axis_aligned_quads", "soup", "deep_strip", "geometric_chain"])
def test_device_tree_equals_host_tree(ctx, cat_golden, kind):
    if kind == "cat":
        v, t = np.array(cat_golden["vertices"], np.float32), np.array(cat_golden["tri_obj_order"], np.int32)
    else:
        rng = np.random.default_rng({"three_triangles": 1, "axis_aligned_quads": 2, "soup": 3, "deep_strip": 4, "geometric_chain": 5}[kind])
        v, t = _synthetic_mesh(kind, rng)
    first = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)        # host tree from OBJ order: what a caller uploads
    ctx.scene_upload(rt.scenes.spheres("cpu"), first)
    if kind == "cat":
        # the device tree against the REFERENCE's own tree (tests/golden/cat_mesh.npz: TriangleMesh::buildBVH + bvhTreeToArray of the
        # reference program, oracle/ref_harness.cpp), not only against this repo's host builder: built from OBJ order, bit for bit
        ctx.scene_upload(rt.scenes.spheres("cpu"), dict(vertices=v, indices=t, bvh_arr10=first["bvh_arr10"], albedo=rt.scenes.CAT_ALBEDO, object_slot=6))
        arr, order = ctx.mesh_rebuild(len(t))
        np.testing.assert_array_equal(arr.view(np.uint32), np.ascontiguousarray(cat_golden["bvh_arr10"], np.float32).view(np.uint32))
        np.testing.assert_array_equal(t[order], np.asarray(cat_golden["tri_bvh_order"])[:, :3])
        ctx.scene_upload(rt.scenes.spheres("cpu"), first)
    p = rt.make_params(320, 200, 1, 1, **rt.scenes.CPU_LAUNCHER)
    exp = _check_rebuild(ctx, v, first["indices"][:, :3])
    got = ctx.render(p)
    # the rebuilt scene renders exactly what a fresh upload of the host-built tree renders, through every traversal kernel
    fresh = rt.Context(0)
    fresh.scene_upload(rt.scenes.spheres("cpu"), exp)
    for variant in ("wavefront_queue", "path", "lockstep"):
        pv = rt.make_params(320, 200, 1, 1, variant=variant, **rt.scenes.CPU_LAUNCHER)
        np.testing.assert_array_equal(ctx.render(pv).view(np.uint32), fresh.render(pv).view(np.uint32))
        assert ctx.count_work(pv) == fresh.count_work(pv)
    fresh.close()
    assert np.isfinite(got).all()
    # a second rebuild starts from the order the first one left (the reference partitions `indices` in place)
    _check_rebuild(ctx, v, exp["indices"][:, :3])


def test_rebuild_after_device_transform(ctx, oracle, cat_golden):
    """transform on the device (global_launcher.cu:340-365), then the tree buildBVH builds for the moved mesh -- what a refit cannot give."""
    v, t = np.array(cat_golden["vertices"], np.float32), np.array(cat_golden["tri_obj_order"], np.int32)
    first = hostlib.build_mesh(v, t, albedo=rt.scenes.CAT_ALBEDO, object_slot=6)
    ctx.scene_upload(rt.scenes.spheres("cpu"), first)
    c, s = np.float32(np.cos(0.7)), np.float32(np.sin(0.7))
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    tr = (2.0, -1.0, 0.5)
    ctx.mesh_transform(R, tr)
    om = oracle.Mesh.from_arrays(v, first["indices"][:, :3]).transform(R, tr)          # the same float operations on the host
    exp = _check_rebuild(ctx, np.array(om.vertices, np.float32), first["indices"][:, :3])
    mesh = oracle.Mesh.from_arrays(np.array(om.vertices, np.float32), exp["indices"][:, :3]).build_bvh()
    W, H = 400, 250
    ref, _, cnt = oracle.Scene.preset("cpu", mesh).render(W, H, 1, 0, want_rgb8=False)
    p = rt.make_params(W, H, 1, 0, **rt.scenes.CPU_LAUNCHER)
    got = ctx.render(p)
    assert values_equal(got[..., :3], ref[..., :3]).all()
    assert ctx.count_work(p) == {k: cnt[k] for k in ("rays", "box_tests", "nodes", "tri_tests")}
    # a scene without a mesh accepts the call
    ctx.scene_upload(rt.scenes.spheres("spheres"), None)
    arr, order = ctx.mesh_rebuild(0)
    assert len(arr) == 0
