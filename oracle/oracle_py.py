"""TEST INFRASTRUCTURE -- ctypes binding of oracle/liboracle.so (rt_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under raytracinggpu_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_ORACLE_LIB") or os.path.join(HERE, "liboracle.so")   # RT_ORACLE_LIB: the sanitizer build (tools/sanitize_cpu.sh)


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("mesh_rays", C.c_uint64), ("box_tests", C.c_uint64),
                ("nodes", C.c_uint64), ("tri_tests", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Params(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("num_rays", C.c_int32), ("num_bounce", C.c_int32),
                ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("sigma", C.c_float), ("eps", C.c_float), ("tri_tmin", C.c_float), ("fov", C.c_float),
                ("cam", C.c_float * 3), ("seed", C.c_uint32), ("threads", C.c_int32),
                ("rng_mode", C.c_int32), ("stride", C.c_int32), ("tile_rows", C.c_int32), ("tile_step", C.c_int32),
                ("cam_mode", C.c_int32), ("yaw", C.c_float), ("pitch", C.c_float)]


def build(force=False):
    if os.environ.get("RT_ORACLE_LIB"):
        return LIB_PATH
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "rt_oracle.c")):
        subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    vp = C.c_void_p
    L.or_mesh_new.restype = vp
    L.or_mesh_free.argtypes = [vp]
    L.or_mesh_read_obj.argtypes = [vp, C.c_char_p, C.c_float, fp]
    L.or_mesh_set_arrays.argtypes = [vp, fp, C.c_int, C.POINTER(C.c_int32), C.c_int]
    L.or_mesh_rescale.argtypes = [vp, C.c_float, fp]
    L.or_mesh_build_bvh.argtypes = [vp]
    L.or_mesh_set_bvh.argtypes = [vp, fp, C.c_int, C.POINTER(C.c_int32)]
    L.or_mesh_set_normals.argtypes = [vp, fp, C.c_int, C.POINTER(C.c_int32)]
    L.or_mesh_transform.argtypes = [vp, fp, fp]
    L.or_mesh_refit.argtypes = [vp]
    for n in ("or_mesh_num_vertices", "or_mesh_num_triangles", "or_mesh_num_nodes", "or_mesh_max_depth"):
        getattr(L, n).argtypes = [vp]
    L.or_mesh_get_vertices.argtypes = [vp, fp]
    L.or_mesh_get_triangles.argtypes = [vp, C.POINTER(C.c_int32)]
    L.or_mesh_bvh_to_array.argtypes = [vp, fp]
    L.or_mesh_set_albedo.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    L.or_mesh_set_material.argtypes = [vp, C.c_int, C.c_float, C.c_float]
    L.or_mesh_intersect.argtypes = [vp, fp, fp, C.c_float, fp, fp, C.POINTER(Counters)]
    L.or_sphere_intersect.argtypes = [fp, C.c_float, fp, fp, fp, fp]
    L.or_box_intersect.argtypes = [fp, fp, fp, fp]
    L.or_moller_trumbore.argtypes = [fp, fp, fp, fp, fp, fp, fp]
    L.or_scene_new.restype = vp
    L.or_scene_free.argtypes = [vp]
    L.or_scene_add_sphere.argtypes = [vp, fp, C.c_float, fp, C.c_int, C.c_float, C.c_float]
    L.or_scene_add_mesh.argtypes = [vp, vp]
    L.or_scene_set_light.argtypes = [vp, fp, C.c_float]
    L.or_scene_intersect_all.argtypes = [vp, fp, fp, C.c_float, fp, fp, C.POINTER(C.c_int), C.POINTER(Counters)]
    L.or_scene_get_color.argtypes = [vp, fp, fp, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32,
                                     fp, C.POINTER(Counters)]
    L.or_uniform.argtypes = [C.c_uint32] * 5
    L.or_uniform.restype = C.c_float
    L.or_render.argtypes = [vp, C.POINTER(Params), fp, C.POINTER(C.c_uint8), C.POINTER(Counters)]
    L.or_tonemap.argtypes = [fp, C.c_int, C.POINTER(C.c_uint8)]
    L.or_max_threads.restype = C.c_int
    L.or_camera_basis.argtypes = [C.c_float, C.c_float, fp, fp, fp]
    L.or_wang_hash.argtypes = [C.c_uint32]
    L.or_wang_hash.restype = C.c_uint32
    L.or_progressive_accumulate.argtypes = [fp, fp, C.c_int, C.c_int, fp, C.POINTER(C.c_uint8)]
    _lib = L
    return L


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


# scene constants: SURVEY Appendix B (cpu_launcher.cpp:668-685, 650-651, 666, 691)
WALLS = [((0, 0, -1000), 940, (0, 1, 0)), ((0, -1000, 0), 990, (0, 0, 1)), ((0, 1000, 0), 940, (1, 0, 0)),
         ((-1000, 0, 0), 940, (0, 1, 1)), ((1000, 0, 0), 940, (1, 1, 0)), ((0, 0, 1000), 940, (1, 0, 1))]
DEMO = [((0, 0, 0), 10, (0, 0, 0), 0, 1.5, 1.0), ((-20, 0, 0), 10, (0, 0, 0), 1, 1.0, 1.0),
        ((20, 0, 0), 9, (0, 0, 0), 0, 1.0, 1.5), ((20, 0, 0), 10, (0, 0, 0), 0, 1.5, 1.0)]


class Mesh:
    def __init__(self):
        self.h = lib().or_mesh_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().or_mesh_free(self.h)
            self.h = None

    @classmethod
    def from_arrays(cls, verts, tris, albedo=(0.25, 0.25, 0.25)):
        m = cls()
        v, vp = _f(verts)
        t = np.ascontiguousarray(tris, np.int32)
        lib().or_mesh_set_arrays(m.h, vp, len(v), t.ctypes.data_as(C.POINTER(C.c_int32)), len(t))
        lib().or_mesh_set_albedo(m.h, *albedo)
        return m

    @classmethod
    def from_obj(cls, path, scale=0.8, offset=(0, -10, 0), albedo=(0.25, 0.25, 0.25)):
        m = cls()
        o, op = _f(offset)
        m.status = lib().or_mesh_read_obj(m.h, path.encode(), scale, op)
        lib().or_mesh_set_albedo(m.h, *albedo)
        return m

    def rescale(self, scale, offset):
        o, op = _f(offset)
        lib().or_mesh_rescale(self.h, scale, op)

    def set_material(self, mirror=0, n_in=1.0, n_out=1.0):
        """Geometry's mirror / refraction indices of the mesh (cpu:113-116); call before Scene.add_mesh."""
        lib().or_mesh_set_material(self.h, int(mirror), n_in, n_out)
        return self

    def build_bvh(self):
        lib().or_mesh_build_bvh(self.h)
        return self

    def set_bvh(self, arr10, order=None):
        """A caller-supplied tree (flat bvhTreeToArray layout, node 0 = root) and the triangle order its ranges refer to (order[k] = current
        index of the triangle that moves to position k): what rt_mesh_rebuild_mode returns, so that the traversal here walks the SAME tree."""
        a = np.ascontiguousarray(arr10, np.float32).reshape(-1, 10)
        o = None if order is None else np.ascontiguousarray(order, np.int32)
        rc = lib().or_mesh_set_bvh(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), len(a), None if o is None else o.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise ValueError("or_mesh_set_bvh: malformed tree or order")
        return self

    def set_normals(self, normals, nidx):
        """Vertex normals + per-triangle (ni, nj, nk) in the mesh's CURRENT triangle order: smooth shading (realtime:221-245)."""
        if normals is None:
            lib().or_mesh_set_normals(self.h, None, 0, None)
            return self
        n, np_ = _f(np.asarray(normals, np.float32).reshape(-1, 3))
        ix = np.ascontiguousarray(nidx, np.int32).reshape(-1, 3)
        lib().or_mesh_set_normals(self.h, np_, len(n), ix.ctypes.data_as(C.POINTER(C.c_int32)))
        return self

    def transform(self, rotation, translation):
        """global_launcher.cu's `transform` kernel on the vertices (row-major 3x3, then translation)."""
        r, rp = _f(np.asarray(rotation, np.float32).reshape(9)); t, tp = _f(translation)
        lib().or_mesh_transform(self.h, rp, tp)
        return self

    def refit(self):
        lib().or_mesh_refit(self.h)
        return self

    @property
    def vertices(self):
        n = lib().or_mesh_num_vertices(self.h)
        a = np.zeros((n, 3), np.float32)
        lib().or_mesh_get_vertices(self.h, a.ctypes.data_as(C.POINTER(C.c_float)))
        return a

    @property
    def triangles(self):
        n = lib().or_mesh_num_triangles(self.h)
        a = np.zeros((n, 3), np.int32)
        lib().or_mesh_get_triangles(self.h, a.ctypes.data_as(C.POINTER(C.c_int32)))
        return a

    @property
    def num_nodes(self):
        return lib().or_mesh_num_nodes(self.h)

    @property
    def max_depth(self):
        return lib().or_mesh_max_depth(self.h)

    def bvh_array(self):
        a = np.zeros((self.num_nodes, 10), np.float32)
        lib().or_mesh_bvh_to_array(self.h, a.ctypes.data_as(C.POINTER(C.c_float)))
        return a

    def intersect(self, O, u, tri_tmin=1e-4):
        O_, Op = _f(O); u_, up = _f(u)
        t = C.c_float(0); N = np.zeros(3, np.float32)
        hit = lib().or_mesh_intersect(self.h, Op, up, tri_tmin, C.byref(t), N.ctypes.data_as(C.POINTER(C.c_float)), None)
        return bool(hit), t.value, N


class Scene:
    def __init__(self):
        self.h = lib().or_scene_new()
        self._meshes = []

    def __del__(self):
        if getattr(self, "h", None):
            lib().or_scene_free(self.h)
            self.h = None

    def add_sphere(self, Cc, Rr, albedo, mirror=0, n_in=1.0, n_out=1.0):
        c, cp = _f(Cc); a, ap = _f(albedo)
        return lib().or_scene_add_sphere(self.h, cp, Rr, ap, int(mirror), n_in, n_out)

    def add_mesh(self, mesh):
        self._meshes.append(mesh)
        return lib().or_scene_add_mesh(self.h, mesh.h)

    def set_light(self, L, intensity):
        l, lp = _f(L)
        lib().or_scene_set_light(self.h, lp, intensity)

    @classmethod
    def preset(cls, name, mesh=None):
        """scene_cpu (cpu:673-685) / spheres / demo10 / optimized (opt:679-726)."""
        s = cls()
        if name == "demo10":
            for c, r, a, m, ni, no in DEMO:
                s.add_sphere(c, r, a, m, ni, no)
        if name == "optimized":
            c, r, a = WALLS[0]
            s.add_sphere(c, r, a)
            if mesh is not None:
                s.add_mesh(mesh)
            for c, r, a in WALLS[1:]:
                s.add_sphere(c, r, a)
            return s
        for c, r, a in WALLS:
            s.add_sphere(c, r, a)
        if name == "cpu" and mesh is not None:
            s.add_mesh(mesh)
        return s

    def intersect_all(self, O, u, tri_tmin=1e-4):
        O_, Op = _f(O); u_, up = _f(u)
        P = np.zeros(3, np.float32); N = np.zeros(3, np.float32); oid = C.c_int(-1)
        hit = lib().or_scene_intersect_all(self.h, Op, up, tri_tmin, P.ctypes.data_as(C.POINTER(C.c_float)),
                                           N.ctypes.data_as(C.POINTER(C.c_float)), C.byref(oid), None)
        return bool(hit), oid.value, P, N

    def render(self, W, H, num_rays=1, num_bounce=0, rows=None, sigma=0.0, eps=1e-3, tri_tmin=1e-4,
               fov=None, cam=(0, 0, 55), seed=123456, threads=0, rng_mode=0, stride=1, want_rgb8=True,
               tile_rows=0, tile_step=0, pose=None):
        """pose = (yaw, pitch): realtime_render.cu's camera and per-sample averaging."""
        p = Params()
        if pose is not None:
            p.cam_mode, p.yaw, p.pitch = 1, pose[0], pose[1]
        p.W, p.H, p.num_rays, p.num_bounce = W, H, num_rays, num_bounce
        p.row_begin, p.row_end = rows if rows else (0, H)
        p.sigma, p.eps, p.tri_tmin = sigma, eps, tri_tmin
        # float alpha = PI/3 (cpu:666): double quotient narrowed to float
        p.fov = np.float32(np.pi / 3) if fov is None else np.float32(fov)
        p.cam[:] = cam
        p.seed, p.threads, p.rng_mode, p.stride = seed, threads, rng_mode, stride
        st = max(stride, 1)
        p.tile_rows, p.tile_step = tile_rows, tile_step
        if tile_step > 1 and tile_rows > 0:
            nr = sum(1 for r in range(p.row_begin, p.row_end, st) if ((r - p.row_begin) // tile_rows) % tile_step == 0)
        else:
            nr = (p.row_end - p.row_begin + st - 1) // st
        nc = (W + st - 1) // st
        rgba = np.zeros((nr, nc, 4), np.float32)
        rgb8 = np.zeros((nr, nc, 3), np.uint8) if want_rgb8 else None
        cnt = Counters()
        rc = lib().or_render(self.h, C.byref(p), rgba.ctypes.data_as(C.POINTER(C.c_float)),
                             rgb8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_rgb8 else None, C.byref(cnt))
        if rc != 0:
            raise ValueError("or_render: bad parameters")
        return rgba, rgb8, cnt.as_dict()


def camera_basis(yaw, pitch):
    """Camera::rotate() of realtime_render.cu:823-846 -> (bx, by, bz)."""
    out = [np.zeros(3, np.float32) for _ in range(3)]
    lib().or_camera_basis(yaw, pitch, *[o.ctypes.data_as(C.POINTER(C.c_float)) for o in out])
    return out


def wang_hash(a):
    return int(lib().or_wang_hash(a & 0xffffffff))


def progressive_accumulate(accum, frame, framenumber):
    """accum (H,W,4 float32, updated in place) += frame; returns (display float32, rgb8) per realtime:1136-1147."""
    assert accum.dtype == np.float32 and accum.flags.c_contiguous
    frame = np.ascontiguousarray(frame, np.float32)
    disp = np.zeros_like(accum)
    rgb8 = np.zeros(accum.shape[:-1] + (3,), np.uint8)
    fpt = C.POINTER(C.c_float)
    lib().or_progressive_accumulate(accum.ctypes.data_as(fpt), frame.ctypes.data_as(fpt), accum.size // 4, framenumber,
                                    disp.ctypes.data_as(fpt), rgb8.ctypes.data_as(C.POINTER(C.c_uint8)))
    return disp, rgb8


def tonemap(rgba):
    a = np.ascontiguousarray(rgba, np.float32).reshape(-1, 4)
    out = np.zeros((len(a), 3), np.uint8)
    lib().or_tonemap(a.ctypes.data_as(C.POINTER(C.c_float)), len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(rgba.shape[:-1] + (3,))


def uniform(seed, pixel, sample, depth, dim):
    return float(lib().or_uniform(seed, pixel, sample, depth, dim))


def gamma_unit(rgb_linear):
    """g = min(pow(c,1/2.2),255)/255 in [0,1]: the scale SURVEY 8d states the L-inf tolerance on."""
    c = np.asarray(rgb_linear, np.float64)
    with np.errstate(invalid="ignore"):
        return np.minimum(np.power(c, 1 / 2.2), 255.0) / 255.0


def algorithmic_bytes(cnt, npix):
    """24 B/box test + 16 B/node visit + 48 B/triangle test + 16 B/pixel framebuffer."""
    return 24 * cnt["box_tests"] + 16 * cnt["nodes"] + 48 * cnt["tri_tests"] + 16 * npix
