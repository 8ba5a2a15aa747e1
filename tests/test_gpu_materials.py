"""Scenes the reference's CLASSES accept and its main() never builds, through the C-ABI vs the CPU oracle.  -m gpu.

A TriangleMesh is a Geometry: Scene::getColor branches on objects[id]->mirror / the refraction indices of WHICHEVER object was hit
(cpu_launcher.cpp:573-606), and Scene::objects is a std::vector<Geometry*> scanned in insertion order with a strict '<' (cpu:538-564), so a scene
may hold several meshes at any positions.  The oracle is pinned on exactly these scenes by the reference TU itself
(tests/golden/ref_materials.npz, tests/test_oracle_pinned.py); here the HIP path must give the oracle's frames: sigma == 0, so every channel bit for bit.
"""
import os
import subprocess

import numpy as np
import pytest

import raytracinggpu_amd as rt
from . import material_scenes as ms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = rt.Context(0)
    yield c
    c.close()


def _frames_equal(got, exp):
    np.testing.assert_array_equal(got[..., :3].view(np.uint32), exp[..., :3].view(np.uint32))
    np.testing.assert_array_equal(got[..., 3], exp[..., 3])          # rays traced per pixel


@pytest.mark.parametrize("name", ms.NAMES)
def test_mesh_materials_and_several_meshes_equal_the_oracle(ctx, oracle, cat_golden, name, monkeypatch):
    """mirror cat / glass cat / two cats (one in the middle of the object order, one mirror at the end) / the same cat twice (every hit an exact tie): frames of
    b = 0, 1 and 5 bounces equal the oracle's in every channel and ray count, through the default pipeline (4-wide BOX step), the fixed-point pairs, the float pairs
    and the per-lane walk (wf_trav); and rt_stats says which traversal kernel ran."""
    v, t = cat_golden["vertices"], cat_golden["tri_obj_order"]
    osc = ms.oracle_scene(oracle, name, v, t)
    spheres, meshes = ms.capi_scene(name, v, t)
    W, H = 448, 256
    exp = {b: osc.render(W, H, 1, b, want_rgb8=False)[0] for b in (0, 1, 5)}
    assert (exp[5][..., :3] != exp[0][..., :3]).any()
    ctx.scene_upload(spheres, meshes)
    for b in (0, 1, 5):
        for variant in ("auto", "wavefront", "lockstep", "global", "path"):   # five structures, each with its own way of replaying the object order: all give the oracle's frame
            _frames_equal(ctx.render(rt.make_params(W, H, 1, b, variant=variant, **rt.scenes.CPU_LAUNCHER)), exp[b])
    ctx.render(rt.make_params(W, H, 1, 1, **rt.scenes.CPU_LAUNCHER))
    assert ctx.stats()["travq_mode"] == 2                              # forests nest (unions of their children): the 4-wide step takes them
    two = sum(1 for o in ms.describe(name, v) if o[0] == "mesh") > 1
    for env, want in (({"RT_TRAVQ_QW": "0", "RT_TRAVQ_Q16": "1"}, 1), ({"RT_TRAVQ_QW": "0", "RT_TRAVQ_Q16": "0"}, 0)):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        c = rt.Context(0)
        for k_ in env:
            monkeypatch.delenv(k_)
        c.scene_upload(spheres, meshes)
        _frames_equal(c.render(rt.make_params(W, H, 1, 5, **rt.scenes.CPU_LAUNCHER)), exp[5])
        assert c.stats()["travq_mode"] == want
        if two:                                                        # operations that address ONE mesh are refused on a forest, not guessed
            with pytest.raises(rt.RtError) as e:
                c.mesh_rebuild(len(t))
            assert e.value.code == -5
            with pytest.raises(rt.RtError) as e:
                c.mesh_set_normals(np.zeros((4, 3), np.float32), np.zeros((len(t), 3), np.int32))
            assert e.value.code == -5
        c.close()
    for variant in ("wavefront_lds", "lds_all"):                       # ... and the LDS-staged forms of the two traversal kernels
        if two and variant == "wavefront_lds":                         # two cats = 4 039 nodes: more than the per-lane walk can stage in 160 KB of LDS -- refused, as for any big tree
            with pytest.raises(rt.RtError) as e:
                ctx.render(rt.make_params(W, H, 1, 5, variant=variant, **rt.scenes.CPU_LAUNCHER))
            assert e.value.code == -5
            continue
        _frames_equal(ctx.render(rt.make_params(W, H, 1, 5, variant=variant, **rt.scenes.CPU_LAUNCHER)), exp[5])


def test_two_meshes_full_size_and_samples(ctx, oracle, cat_golden):
    """two_cats at 1920x1080 b = 3 (the headline's size: chunking, two sub-frames) and with 4 samples per pixel at 640x360 (samples as parallel items)."""
    v, t = cat_golden["vertices"], cat_golden["tri_obj_order"]
    osc = ms.oracle_scene(oracle, "two_cats", v, t)
    ctx.scene_upload(*ms.capi_scene("two_cats", v, t))
    exp, _, _ = osc.render(1920, 1080, 1, 3, want_rgb8=False)
    _frames_equal(ctx.render(rt.make_params(1920, 1080, 1, 3, **rt.scenes.CPU_LAUNCHER)), exp)
    exp, _, cnt = osc.render(640, 360, 4, 2, want_rgb8=False)
    got = ctx.render(rt.make_params(640, 360, 4, 2, **rt.scenes.CPU_LAUNCHER))
    _frames_equal(got, exp)
    # the counting instantiation on the forest: rays and triangle tests are the reference's (a mesh's triangles are tested iff the reference's own walk of THAT mesh reaches their
    # leaf); the box / node counts include the synthetic union nodes above the two roots and are not compared (raytrace_hip.h, rt_count_work)
    work = ctx.count_work(rt.make_params(640, 360, 4, 2, **rt.scenes.CPU_LAUNCHER))
    assert work["rays"] == cnt["rays"] and work["tri_tests"] == cnt["tri_tests"]


def test_two_meshes_move_together_under_the_device_transform(ctx, oracle, cat_golden):
    """rt_mesh_transform on a forest: every mesh's vertices move, every node is refitted bottom-up (the synthetic nodes above the roots become the unions of the moved roots' boxes).
    Against the oracle transforming and refitting each mesh (or_mesh_transform + or_mesh_refit: same trees, boxes by compute_bbox): direct lighting and two bounces bit for bit."""
    v, t = cat_golden["vertices"], cat_golden["tri_obj_order"]
    R = np.array([[0.9553365, 0, 0.29552022], [0, 1, 0], [-0.29552022, 0, 0.9553365]], np.float32)
    tr = (0.5, 0.25, -0.5)
    osc = oracle.Scene()
    desc = ms.describe("two_cats", v)
    for o in desc:
        if o[0] == "sphere":
            osc.add_sphere(o[1], o[2], o[3])
        else:
            m = oracle.Mesh.from_arrays(o[1], t, albedo=o[2]).set_material(o[3], o[4], o[5]).build_bvh()
            m.transform(R, tr).refit()
            osc.add_mesh(m)
    ctx.scene_upload(*ms.capi_scene("two_cats", v, t))
    ctx.mesh_transform(R, tr)
    for b in (0, 2):
        exp, _, _ = osc.render(448, 256, 1, b, want_rgb8=False)
        _frames_equal(ctx.render(rt.make_params(448, 256, 1, b, **rt.scenes.CPU_LAUNCHER)), exp)
    assert ctx.stats()["travq_mode"] == 2


def _f32_point(O, t, u):
    return (O + (np.float32(t) * u).astype(np.float32)).astype(np.float32)


def test_rays_through_a_forest_of_two_meshes(ctx, oracle, cat_golden):
    """rt_trace_rays over the forest of two cats (rt_host_scene.hip.h build_forest) against the oracle's loop over the two meshes in object order (intersect_all of a scene
    holding only them): hit flag, P = O + t u and the normal bit for bit -- camera-like rays, rays between the cats, and the degenerate ones (zero / denormal / huge
    components, axis-parallel) for which the box test is least forgiving (the synthetic union nodes above the two roots must never hide a root the reference enters)."""
    v, t = np.asarray(cat_golden["vertices"], np.float32), cat_golden["tri_obj_order"]
    v2 = ms.cat2_vertices(v)
    from raytracinggpu_amd import hostlib
    meshes = [hostlib.build_mesh(v, t, object_slot=0), hostlib.build_mesh(v2, t, object_slot=1)]
    ctx.scene_upload([], meshes)
    osc = oracle.Scene()
    osc.add_mesh(oracle.Mesh.from_arrays(v, t).build_bvh())
    osc.add_mesh(oracle.Mesh.from_arrays(v2, t).build_bvh())
    rng = np.random.default_rng(606)
    n = 3000
    O = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
    O[:800] = np.float32([0, 0, 55])
    tgt = np.where(rng.random((n, 1)) < 0.5, v[rng.integers(0, len(v), n)], v2[rng.integers(0, len(v2), n)]) + rng.normal(scale=0.2, size=(n, 3))
    u = (tgt - O).astype(np.float32)
    u[: n // 2] = (u[: n // 2] / np.linalg.norm(u[: n // 2], axis=1, keepdims=True)).astype(np.float32)
    k = rng.integers(0, 3, n)
    u[np.arange(2000, 2300), k[2000:2300]] = 0.0
    u[np.arange(2300, 2400), k[2300:2400]] = np.float32(1e-42)
    u[2400:2500] = 0.0; u[np.arange(2400, 2500), k[2400:2500]] = rng.choice([-1.0, 1.0], 100)
    u[2500:2550] *= np.float32(1e20)
    root = np.minimum(v.min(0), v2.min(0))
    O[2550:2650, 0] = root[0]; u[2550:2650, 0] = 0.0                   # origin ON the union box's face, travelling along it
    rays = np.concatenate([O, u], axis=1).astype(np.float32)
    for variant in ("wavefront_queue", "wavefront"):
        got = ctx.trace_rays(rays, 1e-4, variant)
        nh = 0
        for i in range(n):
            hit, oid, P, N = osc.intersect_all(rays[i, :3], rays[i, 3:], 1e-4)
            assert bool(got[i, 0]) == hit, (variant, i)
            if hit:
                nh += 1
                np.testing.assert_array_equal(_f32_point(rays[i, :3], got[i, 1], rays[i, 3:]).view(np.uint32), P.view(np.uint32))
                np.testing.assert_array_equal(got[i, 2:5].view(np.uint32), N.view(np.uint32))
        assert nh > 500 and n - nh > 300


def test_a_mesh_without_triangles_is_an_object_that_is_never_hit(ctx, oracle, oracle_cat, cat_golden):
    """readOBJ on a missing file leaves an empty TriangleMesh that main() still adds (cpu:322-325, 685): it holds its place in Scene::objects and nothing else.  With a
    second, real mesh in the scene the frame is the plain cat scene's (the spheres keep their relative order; only exact ties could tell, and none involve the hole)."""
    from raytracinggpu_amd import hostlib
    cat = hostlib.build_mesh(cat_golden["vertices"], cat_golden["tri_obj_order"], albedo=rt.scenes.CAT_ALBEDO, object_slot=7)
    empty = dict(vertices=np.zeros((0, 3), np.float32), indices=np.zeros((0, 3), np.int32), bvh_arr10=np.zeros((0, 10), np.float32), object_slot=2)
    ctx.scene_upload(rt.scenes.spheres("cpu"), [empty, cat])
    exp, _, _ = oracle.Scene.preset("cpu", oracle_cat).render(384, 216, 1, 3, want_rgb8=False)
    for variant in ("auto", "wavefront", "lockstep", "global", "path"):
        _frames_equal(ctx.render(rt.make_params(384, 216, 1, 3, variant=variant, **rt.scenes.CPU_LAUNCHER)), exp)
    # ... and alone it is the spheres-only scene, through every kernel family
    ctx.scene_upload(rt.scenes.spheres("cpu"), [dict(empty, object_slot=6)])
    exp, _, _ = oracle.Scene.preset("spheres").render(384, 216, 1, 3, want_rgb8=False)
    for variant in ("auto", "wavefront_queue", "lockstep", "global", "path"):
        _frames_equal(ctx.render(rt.make_params(384, 216, 1, 3, variant=variant, **rt.scenes.CPU_LAUNCHER)), exp)


def test_upload_refuses_what_it_cannot_represent(ctx, cat_golden):
    from raytracinggpu_amd import hostlib
    cat = hostlib.build_mesh(cat_golden["vertices"], cat_golden["tri_obj_order"], object_slot=6)
    for bad in ([dict(cat, object_slot=3), dict(cat, object_slot=3)], [dict(cat, object_slot=9)], [dict(cat, object_slot=-1)]):
        with pytest.raises(rt.RtError) as e:
            ctx.scene_upload(rt.scenes.spheres("cpu"), bad)
        assert e.value.code == -1
    with pytest.raises(rt.RtError):                                    # 6 spheres + 11 meshes: more than RT_MAX_OBJECTS
        ctx.scene_upload(rt.scenes.spheres("cpu"), [dict(cat, object_slot=6 + k) for k in range(11)])


def test_launcher_renders_the_material_scenes_through_the_cpp_host_api(tmp_path, oracle, cat_golden):
    """include/raytracer.hpp: `mesh_ptr->mirror = true` and a second TriangleMesh in Scene::objects reach the device (SceneArrays -> rt_scene_upload_meshes); the PNG
    of `rt_launcher 1 3 --mesh-material mirror` / `--second-cat 1` holds the oracle's tonemapped bytes."""
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    launcher = os.path.join(root, "raytracinggpu_amd", "rt_launcher")
    d = tmp_path / "cadnav.com_model" / "Models_F0202A090"
    d.mkdir(parents=True)
    with open(d / "cat.obj", "w") as f:                               # 6-number vertex lines are not transformed by readOBJ (cpu:344-350); %.9g round-trips binary32
        for v in cat_golden["vertices"]:
            f.write("v %.9g %.9g %.9g 1 1 1\r\n" % tuple(float(x) for x in v))
        for t in cat_golden["tri_obj_order"]:
            f.write("f %d/1/1 %d/1/1 %d/1/1\r\n" % tuple(int(x) + 1 for x in t))
    v, t = cat_golden["vertices"], cat_golden["tri_obj_order"]
    cases = {"cpu_mirror": ["--mesh-material", "mirror"], "cpu_glass": ["--mesh-material", "glass"]}
    for name, extra in cases.items():
        r = subprocess.run([launcher, "1", "3", "--out", name + ".png", *extra], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        _, rgb8, _ = ms.oracle_scene(oracle, name, v, t).render(512, 512, 1, 3)
        np.testing.assert_array_equal(np.array(Image.open(tmp_path / (name + ".png")).convert("RGB")), rgb8)
    # walls, cat (object 6), second cat (object 7, mirror): the launcher's own order
    osc = oracle.Scene.preset("cpu", oracle.Mesh.from_arrays(v, t).build_bvh())
    osc.add_mesh(oracle.Mesh.from_arrays(ms.cat2_vertices(v), t, albedo=(0.6, 0.3, 0.1)).set_material(1, 1.0, 1.0).build_bvh())
    r = subprocess.run([launcher, "1", "3", "--out", "two.png", "--second-cat", "1"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    _, rgb8, _ = osc.render(512, 512, 1, 3)
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "two.png").convert("RGB")), rgb8)


def test_sixteen_objects_three_meshes_every_material_and_the_rare_traversal_paths(oracle, cat_golden, monkeypatch):
    """The full house: 16 objects (RT_MAX_OBJECTS) -- thirteen spheres (the demo scene's mirror, glass, nested glass and walls + three more) and three meshes at positions 0, 8 and 15: a GLASS
    single-leaf mesh of three triangles (its root is a leaf hanging directly below a synthetic node), a mirror soup with duplicated and degenerate triangles, the diffuse cat.
    Frames of 5 bounces equal the oracle's in every channel and ray count through the default pipeline, the per-lane walk, the LDS-staged work-stack kernel, and the work-stack
    kernel with a stack so small that popped pairs are walked serially (RT_TRAVQ_CAP=128: the drain crosses the synthetic nodes); 4 samples per pixel as parallel items too."""
    from raytracinggpu_amd import hostlib
    from .test_gpu_parity import _synthetic_mesh
    v, t = np.asarray(cat_golden["vertices"], np.float32), np.asarray(cat_golden["tri_obj_order"], np.int32)
    v3, t3 = _synthetic_mesh("three_triangles", np.random.default_rng(1))
    v3 = (v3 * np.float32(0.6) + np.float32([-14, 4, 14])).astype(np.float32)
    vs, ts = _synthetic_mesh("soup", np.random.default_rng(3))
    vs = (vs * np.float32(0.35) + np.float32([15, 2, 10])).astype(np.float32)
    spheres = rt.scenes.spheres("demo10") + [((-8.0, -6.0, 18.0), 3.0, (0.8, 0.8, 0.1)), ((6.0, 12.0, 5.0), 2.5, (0.1, 0.1, 0.1), 1, 1.0, 1.0),
                                             ((0.0, -7.0, 28.0), 2.0, (1.0, 1.0, 1.0), 0, 1.3, 1.0)]   # 4 demo spheres + 6 walls + a diffuse, a mirror and a glass ball: 13
    geo = [(v3, t3, (0.9, 0.9, 0.9), 0, 1.5, 1.0, 0), (vs, ts, (0.2, 0.7, 0.3), 1, 1.0, 1.0, 8), (v, t, rt.scenes.CAT_ALBEDO, 0, 1.0, 1.0, 15)]
    meshes = []
    for vv, tt, alb, mir, ni, no, slot in geo:
        d = hostlib.build_mesh(vv, tt, albedo=alb, object_slot=slot)
        d.update(mirror=mir, in_refraction_index=ni, out_refraction_index=no)
        meshes.append(d)
    osc = oracle.Scene()
    it, n_obj = iter(spheres), len(spheres) + len(geo)
    slots = {g[6]: g for g in geo}
    for pos in range(n_obj):
        if pos in slots:
            vv, tt, alb, mir, ni, no, _ = slots[pos]
            osc.add_mesh(oracle.Mesh.from_arrays(vv, tt, albedo=alb).set_material(mir, ni, no).build_bvh())
        else:
            s = next(it)
            osc.add_sphere(s[0], s[1], s[2], *(s[3:] if len(s) > 3 else ()))
    W, H = 416, 240
    exp5, _, _ = osc.render(W, H, 1, 5, want_rgb8=False)
    exp4, _, _ = osc.render(W, H, 4, 2, want_rgb8=False)
    assert n_obj == 16
    for env, variants in (({}, ("auto", "wavefront", "lds_all", "wavefront_lds", "lockstep", "global", "path")), ({"RT_TRAVQ_CAP": "128"}, ("auto", "path")), ({"RT_TRAVQ_CAP": "128", "RT_TRAVQ_QW": "0"}, ("auto",))):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        c = rt.Context(0)
        for k_ in env:
            monkeypatch.delenv(k_)
        c.scene_upload(spheres, meshes)
        for variant in variants:
            _frames_equal(c.render(rt.make_params(W, H, 1, 5, variant=variant, **rt.scenes.CPU_LAUNCHER)), exp5)
        _frames_equal(c.render(rt.make_params(W, H, 4, 2, **rt.scenes.CPU_LAUNCHER)), exp4)
        if env:
            assert c.count_work(rt.make_params(W, H, 1, 5, **rt.scenes.CPU_LAUNCHER), detail=True)["steps"]["serial_drains"] > 0
        c.close()
    with pytest.raises(rt.RtError) as e:                              # a seventeenth object
        c = rt.Context(0)
        c.scene_upload(spheres + [((0, 0, 0), 1, (1, 1, 1))], meshes)
    assert e.value.code == -1
