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


def test_kat_sphere_on_device(ctx):
    g = load_golden("kat.npz")
    got = ctx.kat_sphere(g["sphere_in"])
    exp = g["sphere_out"]
    np.testing.assert_array_equal(got[:, 0], exp[:, 0])
    hit = exp[:, 0] != 0
    np.testing.assert_array_equal(bits(got[hit, 1:5]), bits(exp[hit, 1:5]))
    assert 0.05 < hit.mean() < 0.95


@pytest.mark.parametrize("route", [0, 1, 2])
def test_kat_box_on_device(ctx, route):
    """BoundingBox::intersect incl. u_k = 0 (+-inf / NaN slabs): literal code, slab_filtered, qbox_filter + fall-back."""
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


def test_kat_error_paths(ctx):
    with pytest.raises(rt.RtError):
        ctx.kat_box(np.zeros((4, 12), np.float32), 7)
    fresh = rt.Context(0)
    with pytest.raises(rt.RtError):
        fresh.kat_mesh(np.zeros((4, 6), np.float32))
    fresh.close()
    out, cnt = ctx.kat_box(np.zeros((0, 12), np.float32), 1)
    assert out.shape == (0,) and cnt["n"] == 0
