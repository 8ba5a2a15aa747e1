"""The scenes of tests/golden/ref_materials.npz (oracle/ref_harness.cpp add_material_scene), built twice from the same description: for the CPU oracle and for the
C-ABI (rt_scene_upload_meshes).  A TriangleMesh is a Geometry: its mirror / refraction members are read by Scene::getColor like a sphere's (cpu_launcher.cpp:573-606),
and Scene::objects takes any number of meshes at any positions (cpu_launcher.cpp:538-564)."""
import numpy as np

WALLS = [((0, 0, -1000), 940, (0, 1, 0)), ((0, -1000, 0), 990, (0, 0, 1)), ((0, 1000, 0), 940, (1, 0, 0)),
         ((-1000, 0, 0), 940, (0, 1, 1)), ((1000, 0, 0), 940, (1, 1, 0)), ((0, 0, 1000), 940, (1, 0, 1))]
NAMES = ("cpu_mirror", "cpu_glass", "two_cats", "two_cats_diffuse")


def cat2_vertices(verts):
    """the second cat of ref_harness.cpp load_cat2: v * 0.5f + (16, -5, 20) in binary32 (one rounding per operator, as the reference's Vector)"""
    v = np.asarray(verts, np.float32)
    return ((v * np.float32(0.5)).astype(np.float32) + np.array([16, -5, 20], np.float32)).astype(np.float32)


def describe(name, cat_verts):
    """-> list of objects in Scene::objects order: ("sphere", C, R, albedo) or ("mesh", vertices, albedo, mirror, n_in, n_out)"""
    cat = np.asarray(cat_verts, np.float32)
    walls = [("sphere",) + w for w in WALLS]
    if name == "cpu_mirror":
        return walls + [("mesh", cat, (0.25, 0.25, 0.25), 1, 1.0, 1.0)]
    if name == "cpu_glass":
        return walls + [("mesh", cat, (0.25, 0.25, 0.25), 0, 1.5, 1.0)]
    if name == "two_cats":
        return walls[:3] + [("mesh", cat, (0.25, 0.25, 0.25), 0, 1.0, 1.0)] + walls[3:] + [("mesh", cat2_vertices(cat), (0.6, 0.3, 0.1), 1, 1.0, 1.0)]
    if name == "two_cats_diffuse":
        return [("mesh", cat, (0.9, 0.1, 0.1), 0, 1.0, 1.0)] + walls + [("mesh", cat, (0.1, 0.9, 0.1), 0, 1.0, 1.0)]
    raise ValueError(name)


def oracle_scene(oracle, name, cat_verts, cat_tris):
    s = oracle.Scene()
    for o in describe(name, cat_verts):
        if o[0] == "sphere":
            s.add_sphere(o[1], o[2], o[3])
        else:
            m = oracle.Mesh.from_arrays(o[1], cat_tris, albedo=o[2]).set_material(o[3], o[4], o[5]).build_bvh()
            s.add_mesh(m)
    return s


def capi_scene(name, cat_verts, cat_tris):
    """-> (spheres, meshes) for Context.scene_upload_meshes: the product's own BVH builder per mesh, object_slot = position in the list"""
    from raytracinggpu_amd import hostlib
    spheres, meshes = [], []
    for pos, o in enumerate(describe(name, cat_verts)):
        if o[0] == "sphere":
            spheres.append((o[1], o[2], o[3]))
        else:
            d = hostlib.build_mesh(o[1], cat_tris, albedo=o[2], object_slot=pos)
            d.update(mirror=o[3], in_refraction_index=o[4], out_refraction_index=o[5])
            meshes.append(d)
    return spheres, meshes
