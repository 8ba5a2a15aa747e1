"""Scene presets of the reference programs, as data for rt_scene_upload.

The reference hard-codes its scene in main()/in the kernel; constants below are
cpu_launcher.cpp:668-685 (walls, commented demo spheres), :650-651 (light),
:666,:691 (camera) and optimized.cu:679-726 (object order of the CUDA program).
"""
import os

import numpy as np

# (center, radius, albedo[, mirror, n_in, n_out])
WALLS = [((0, 0, -1000), 940, (0, 1, 0)), ((0, -1000, 0), 990, (0, 0, 1)), ((0, 1000, 0), 940, (1, 0, 0)),
         ((-1000, 0, 0), 940, (0, 1, 1)), ((1000, 0, 0), 940, (1, 1, 0)), ((0, 0, 1000), 940, (1, 0, 1))]
DEMO = [((0, 0, 0), 10, (0, 0, 0), 0, 1.5, 1.0), ((-20, 0, 0), 10, (0, 0, 0), 1, 1.0, 1.0),
        ((20, 0, 0), 9, (0, 0, 0), 0, 1.0, 1.5), ((20, 0, 0), 10, (0, 0, 0), 0, 1.5, 1.0)]
LIGHT = ((-10.0, 20.0, 40.0), 3e10)
CAMERA = ((0.0, 0.0, 55.0), None)          # fov None => float(PI/3)
CAT_ALBEDO = (0.25, 0.25, 0.25)

# per-program render constants
CPU_LAUNCHER = dict(depth_convention=0, sigma=0.0, eps=1e-3, tri_tmin=1e-4)
OPTIMIZED_CU = dict(depth_convention=1, sigma=0.2, eps=1e-4, tri_tmin=0.0)

# the benchmark mesh as arrays (vertices after readOBJ's v*0.8+(0,-10,0), faces in OBJ order, the reference's BVH): package data,
# produced from the reference's asset by oracle/make_golden.py; tests and bench.py read it from here
CAT_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cat_mesh.npz")


def load_cat_arrays(path=CAT_FIXTURE):
    """The cat as cpu_launcher.cpp's readOBJ leaves it (v*0.8+(0,-10,0)), OBJ face order."""
    g = np.load(path, allow_pickle=False)
    return np.array(g["vertices"], np.float32), np.array(g["tri_obj_order"], np.int32)


def spheres(name):
    if name == "demo10":
        return list(DEMO) + list(WALLS)
    return list(WALLS)


def mesh_slot(name):
    """cpu_launcher adds the mesh last (cpu:685); optimized.cu puts it at index 1 (optimized.cu:690-700)."""
    return 1 if name == "optimized" else len(spheres(name))
