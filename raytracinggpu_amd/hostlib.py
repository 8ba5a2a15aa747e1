"""ctypes binding of libraytrace_host.so (include/raytracer_host.h): the product's own OBJ reader, BVH
builder/flattener and PNG writer (C++ host API of include/raytracer.hpp).  No GPU code."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HOST_LIB") or os.path.join(HERE, "libraytrace_host.so")   # RT_HOST_LIB: the sanitizer build (tools/sanitize_cpu.sh)
EXPORTS = ["rth_mesh_new", "rth_mesh_free", "rth_mesh_read_obj", "rth_mesh_set_arrays", "rth_mesh_rescale",
           "rth_mesh_build_bvh", "rth_mesh_num_vertices", "rth_mesh_num_triangles", "rth_mesh_num_nodes",
           "rth_mesh_get_vertices", "rth_mesh_get_indices", "rth_mesh_get_bvh_array", "rth_write_png"]
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.rth_mesh_new.restype = vp
        L.rth_mesh_free.argtypes = [vp]
        L.rth_mesh_read_obj.argtypes = [vp, C.c_char_p, C.c_float, fp]
        L.rth_mesh_set_arrays.argtypes = [vp, fp, C.c_int, ip, C.c_int]
        L.rth_mesh_rescale.argtypes = [vp, C.c_float, fp]
        L.rth_mesh_build_bvh.argtypes = [vp]
        for n in ("rth_mesh_num_vertices", "rth_mesh_num_triangles", "rth_mesh_num_nodes"):
            getattr(L, n).argtypes = [vp]
        L.rth_mesh_get_vertices.argtypes = [vp, fp]
        L.rth_mesh_get_indices.argtypes = [vp, ip]
        L.rth_mesh_get_bvh_array.argtypes = [vp, fp]
        L.rth_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        _lib = L
    return _lib


class HostMesh:
    """TriangleMeshHost (optimized.cu:293-535) through the C exports."""

    def __init__(self):
        self._L = load()
        self._h = self._L.rth_mesh_new()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rth_mesh_free(self._h)
            self._h = None

    @classmethod
    def from_arrays(cls, verts, tris):
        m = cls()
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
        m._L.rth_mesh_set_arrays(m._h, v.ctypes.data_as(C.POINTER(C.c_float)), len(v), t.ctypes.data_as(C.POINTER(C.c_int32)), len(t))
        return m

    @classmethod
    def from_obj(cls, path, scale=0.8, offset=(0.0, -10.0, 0.0)):
        m = cls()
        o = np.asarray(offset, np.float32)
        m.status = m._L.rth_mesh_read_obj(m._h, os.fsencode(path), scale, o.ctypes.data_as(C.POINTER(C.c_float)))
        return m

    def rescale(self, scale, offset):
        o = np.asarray(offset, np.float32)
        self._L.rth_mesh_rescale(self._h, scale, o.ctypes.data_as(C.POINTER(C.c_float)))

    def build_bvh(self):
        self._L.rth_mesh_build_bvh(self._h)
        return self

    @property
    def vertices(self):
        a = np.zeros((self._L.rth_mesh_num_vertices(self._h), 3), np.float32)
        self._L.rth_mesh_get_vertices(self._h, a.ctypes.data_as(C.POINTER(C.c_float)))
        return a

    @property
    def indices10(self):
        """TriangleIndices[nt] as int32[nt,10] (vtxi,vtxj,vtxk first), in the current (BVH) order."""
        a = np.full((self._L.rth_mesh_num_triangles(self._h), 10), -1, np.int32)
        self._L.rth_mesh_get_indices(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)))
        return a

    @property
    def bvh_arr10(self):
        a = np.zeros((self._L.rth_mesh_num_nodes(self._h), 10), np.float32)
        self._L.rth_mesh_get_bvh_array(self._h, a.ctypes.data_as(C.POINTER(C.c_float)))
        return a


def build_mesh(verts, tris, albedo=(0.25, 0.25, 0.25), object_slot=None, rescale=None):
    """OBJ-order arrays -> the dict Context.scene_upload takes (reference array layouts, stride-10 indices)."""
    m = HostMesh.from_arrays(verts, tris)
    if rescale is not None:
        m.rescale(*rescale)
    m.build_bvh()
    d = dict(vertices=m.vertices, indices=m.indices10, bvh_arr10=m.bvh_arr10, albedo=albedo)
    if object_slot is not None:
        d["object_slot"] = object_slot
    return d


def write_png(path, rgb8):
    a = np.ascontiguousarray(rgb8, np.uint8)
    rc = load().rth_write_png(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != 0:
        raise OSError(f"cannot write {path}")
