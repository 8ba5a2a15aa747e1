"""ctypes binding of libraytrace_hip.so (include/raytrace_hip.h).

Plumbing only: the render path is the HIP library.  There is no CPU fallback --
if the shared library is missing or no gfx950 device is visible this module
raises instead of rendering with something else.
"""
import ctypes as C
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(HERE, "libraytrace_hip.so")   # RT_LIB: an experimental build (tools/)

RT_OK = 0
VARIANT_AUTO, VARIANT_GLOBAL, VARIANT_LDS_VERTS, VARIANT_LDS_TOP, VARIANT_LDS_ALL, VARIANT_LOCKSTEP, VARIANT_WAVEFRONT, VARIANT_WAVEFRONT_LDS, VARIANT_WAVEFRONT_QUEUE, VARIANT_PATH = range(10)
VARIANTS = {"auto": 0, "global": 1, "lds_verts": 2, "lds_top": 3, "lds_all": 4, "lockstep": 5, "wavefront": 6, "wavefront_lds": 7, "wavefront_queue": 8, "path": 9}

# every symbol include/raytrace_hip.h declares (tests check the .so exports each)
EXPORTS = ["rt_abi_version", "rt_device_count", "rt_ctx_create", "rt_ctx_destroy", "rt_last_error",
           "rt_device_name", "rt_scene_upload", "rt_scene_upload_meshes", "rt_render", "rt_render_device", "rt_render_device_batch", "rt_tonemap_device",
           "rt_render_rgb8", "rt_synchronize", "rt_get_stats", "rt_ctx_selfcheck", "rt_count_work",
           "rt_mesh_transform", "rt_mesh_set_normals", "rt_camera_basis", "rt_render_pose", "rt_render_pose_device", "rt_progressive_reset", "rt_progressive_frame",
           "rt_progressive_frames",
           "rt_multi_create", "rt_multi_destroy", "rt_multi_last_error", "rt_multi_scene_upload", "rt_multi_scene_upload_meshes", "rt_render_multi",
           "rt_render_multi_device", "rt_render_multi_rgb8", "rt_multi_get_stats",
           "rt_stats_enable", "rt_ctx_set_pipelining", "rt_render_async", "rt_wait", "rt_trace_rays", "rt_mesh_rebuild", "rt_mesh_rebuild_mode", "rt_mesh_build_stats", "rt_host_alloc", "rt_host_free", "rt_device_alloc", "rt_device_free", "rt_device_to_host", "rt_kat_sphere", "rt_kat_sqrt", "rt_kat_box", "rt_kat_triangle", "rt_kat_mesh", "rt_kat_layout_hash"]
MAX_DEVICES = 16


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libraytrace_hip: status {code}: {msg}")
        self.code = code


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("albedo", C.c_float * 3), ("mirror", C.c_int32),
                ("in_refraction_index", C.c_float), ("out_refraction_index", C.c_float)]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("n_vertices", C.c_int32),
                ("indices", C.POINTER(C.c_int32)), ("index_stride", C.c_int32), ("n_triangles", C.c_int32),
                ("bvh_arr10", C.POINTER(C.c_float)), ("n_nodes", C.c_int32),
                ("albedo", C.c_float * 3), ("object_slot", C.c_int32),
                ("mirror", C.c_int32), ("in_refraction_index", C.c_float), ("out_refraction_index", C.c_float)]   # ABI 6: Geometry's other fields (cpu:113-116)


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("intensity", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("fov", C.c_float)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("num_rays", C.c_int32), ("num_bounce", C.c_int32),
                ("depth_convention", C.c_int32), ("sigma", C.c_float), ("eps", C.c_float), ("tri_tmin", C.c_float),
                ("seed", C.c_uint32), ("variant", C.c_int32)]


class Rows(C.Structure):
    _fields_ = [("row0", C.c_int32), ("n_rows", C.c_int32), ("tile_rows", C.c_int32), ("tile_step", C.c_int32)]


class FrameDesc(C.Structure):
    _fields_ = [("camera", Camera), ("seed", C.c_uint32), ("reserved", C.c_uint32), ("out_rgba_dev", C.c_void_p)]


MAX_BATCH = 16


class Work(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("nodes", C.c_uint64), ("tri_tests", C.c_uint64),
                ("box_literal", C.c_uint64), ("tri_literal", C.c_uint64), ("steps", C.c_uint64 * 12)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("tonemap_ms", C.c_float), ("pixels", C.c_uint64), ("variant", C.c_int32),
                ("lds_bytes", C.c_int32), ("block_threads", C.c_int32), ("grid_blocks", C.c_int32),
                ("trav_ms", C.c_float), ("trav_launches", C.c_int32), ("parts", C.c_int32), ("adv_launches", C.c_int32),
                ("adv_ms", C.c_float), ("adv_paths", C.c_int32), ("travq_mode", C.c_int32), ("reserved", C.c_int32)]


class BuildStats(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_triangles", C.c_int32), ("n_nodes", C.c_int32), ("n_leaves", C.c_int32), ("max_leaf_tris", C.c_int32),
                ("max_depth", C.c_int32), ("device_build_ms", C.c_float), ("install_ms", C.c_float), ("install_on_device", C.c_int32), ("reserved", C.c_int32)]


BVH_MODES = {"reference": 0, "lbvh": 1}


class KatCounts(C.Structure):
    _fields_ = [("n", C.c_uint64), ("box_decided", C.c_uint64), ("box_literal", C.c_uint64), ("tri_decided", C.c_uint64), ("tri_literal", C.c_uint64)]


class CameraPose(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("yaw", C.c_float), ("pitch", C.c_float), ("fov", C.c_float)]


def make_pose(position=(0.0, 0.0, 55.0), yaw=0.0, pitch=0.3, fov=None):
    """Camera() of realtime_render.cu:805-810 (C = (0,0,55), yaw 0, pitch 0.3); Scene::pov = PI / 2 (realtime:1021)."""
    q = CameraPose()
    q.position[:] = position
    q.yaw, q.pitch = yaw, pitch
    q.fov = np.float32(np.pi / 2) if fov is None else np.float32(fov)
    return q


class MultiStats(C.Structure):
    _fields_ = [("n_devices", C.c_int32), ("device_id", C.c_int32 * MAX_DEVICES), ("kernel_ms", C.c_float * MAX_DEVICES),
                ("gather_ms", C.c_float), ("frame_ms", C.c_float), ("rays", C.c_uint64), ("gather_bytes", C.c_uint64),
                ("peer_access", C.c_int32 * MAX_DEVICES), ("submit_ms", C.c_float)]


_lib = None


def load():
    """Load libraytrace_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and
    # its libraries ask for it as "libamdhip64.so", so a system copy loaded first is NOT reused and the second
    # runtime then finds no GPU.  Loading torch first makes this library's NEEDED libamdhip64.so.7 resolve to
    # the copy torch already mapped.  (The C++ launcher, which has no torch, uses /opt/rocm's.)
    if os.environ.get("RT_WITHOUT_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    if os.environ.get("RT_LIB"):      # an experimental / older build may lack the newest entry points: give them inert stand-ins
        class _Missing:
            argtypes = restype = None
        for name in EXPORTS:
            if not hasattr(L, name):
                setattr(L, name, _Missing())
    vp = C.c_void_p
    L.rt_abi_version.restype = C.c_int
    L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.rt_ctx_create.argtypes = [C.POINTER(vp), C.c_int]
    L.rt_ctx_destroy.argtypes = [vp]
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_device_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.rt_scene_upload.argtypes = [vp, C.POINTER(Sphere), C.c_int, C.POINTER(Mesh), C.POINTER(Light), C.POINTER(Camera)]
    L.rt_scene_upload_meshes.argtypes = [vp, C.POINTER(Sphere), C.c_int, C.POINTER(Mesh), C.c_int, C.POINTER(Light), C.POINTER(Camera)]
    L.rt_render.argtypes = [vp, C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.rt_render_device.argtypes = [vp, C.POINTER(Params), C.POINTER(Rows), vp, vp]
    L.rt_render_device_batch.argtypes = [vp, C.POINTER(Params), C.POINTER(Rows), C.POINTER(FrameDesc), C.c_int, vp]
    L.rt_tonemap_device.argtypes = [vp, vp, C.c_int64, vp, vp]
    L.rt_render_rgb8.argtypes = [vp, C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    L.rt_count_work.argtypes = [vp, C.POINTER(Params), C.c_int, C.c_int, C.POINTER(Work)]
    L.rt_synchronize.argtypes = [vp]
    L.rt_ctx_selfcheck.argtypes = [vp]
    L.rt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rt_stats_enable.argtypes = [vp, C.c_int]
    L.rt_ctx_set_pipelining.argtypes = [vp, C.c_int]
    L.rt_render_async.argtypes = [vp, C.POINTER(Params), C.c_int, vp, C.c_int]
    L.rt_wait.argtypes = [vp, C.c_int]
    L.rt_trace_rays.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_int, C.POINTER(C.c_float)]
    fp3 = C.POINTER(C.c_float)
    L.rt_mesh_transform.argtypes = [vp, fp3, fp3]
    L.rt_mesh_set_normals.argtypes = [vp, fp3, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int]
    L.rt_camera_basis.argtypes = [C.POINTER(CameraPose), fp3, fp3, fp3]
    L.rt_render_pose.argtypes = [vp, C.POINTER(Params), C.POINTER(CameraPose), fp3]
    L.rt_render_pose_device.argtypes = [vp, C.POINTER(Params), C.POINTER(CameraPose), C.POINTER(Rows), vp, vp]
    L.rt_progressive_reset.argtypes = [vp]
    L.rt_progressive_frame.argtypes = [vp, C.POINTER(Params), C.POINTER(CameraPose), fp3, C.POINTER(C.c_uint8)]
    L.rt_progressive_frames.argtypes = [vp, C.POINTER(C.c_int)]
    L.rt_multi_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    L.rt_multi_destroy.argtypes = [vp]
    L.rt_multi_last_error.argtypes = [vp]
    L.rt_multi_last_error.restype = C.c_char_p
    L.rt_multi_scene_upload.argtypes = [vp, C.POINTER(Sphere), C.c_int, C.POINTER(Mesh), C.POINTER(Light), C.POINTER(Camera)]
    L.rt_multi_scene_upload_meshes.argtypes = [vp, C.POINTER(Sphere), C.c_int, C.POINTER(Mesh), C.c_int, C.POINTER(Light), C.POINTER(Camera)]
    L.rt_render_multi.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_float)]
    L.rt_render_multi_device.argtypes = [vp, C.POINTER(Params), vp]
    L.rt_render_multi_rgb8.argtypes = [vp, C.POINTER(Params), C.POINTER(C.c_uint8)]
    L.rt_multi_get_stats.argtypes = [vp, C.POINTER(MultiStats)]
    L.rt_mesh_rebuild.argtypes = [vp, fp3, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rt_mesh_rebuild_mode.argtypes = [vp, C.c_int, fp3, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rt_mesh_build_stats.argtypes = [vp, C.POINTER(BuildStats)]
    L.rt_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.rt_host_free.argtypes = [vp]
    L.rt_kat_sphere.argtypes = [vp, fp3, C.c_int, fp3]
    L.rt_kat_sqrt.argtypes = [vp, fp3, C.c_int, fp3]
    L.rt_kat_box.argtypes = [vp, fp3, C.c_int, C.c_int, fp3, C.POINTER(KatCounts)]
    L.rt_kat_triangle.argtypes = [vp, fp3, C.c_int, fp3, C.POINTER(KatCounts)]
    L.rt_kat_mesh.argtypes = [vp, fp3, C.c_int, C.c_float, C.c_int, fp3, C.POINTER(KatCounts)]
    L.rt_kat_layout_hash.argtypes = [vp, C.POINTER(C.c_uint64)]
    _lib = L
    return L


class PinnedArray:
    """A float32 / uint8 numpy array over rt_host_alloc memory (frame buffer for rt_render*: the D2H copy is one DMA).

    The allocation lives as long as ANY view of it: `array`, its slices and whatever a render call returned all keep the underlying
    ctypes buffer alive, and the pinned memory is released when that buffer is collected -- close() only drops this object's own
    reference, it never frees memory somebody still looks at."""

    def __init__(self, shape, dtype=np.float32):
        L = load()
        p = C.c_void_p()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        rc = L.rt_host_alloc(C.byref(p), max(n, 16))
        if rc != RT_OK:
            raise RtError(rc, L.rt_last_error(None).decode())
        buf = (C.c_uint8 * max(n, 16)).from_address(p.value)
        weakref.finalize(buf, L.rt_host_free, C.c_void_p(p.value))    # runs when the last numpy view of `buf` is gone
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        self.array = None


def camera_basis(pose):
    """Camera::rotate() (realtime_render.cu:823-846) as the library computes it -> (bx, by, bz)."""
    out = [np.zeros(3, np.float32) for _ in range(3)]
    rc = load().rt_camera_basis(C.byref(pose), *[o.ctypes.data_as(C.POINTER(C.c_float)) for o in out])
    if rc != RT_OK:
        raise RtError(rc, "rt_camera_basis")
    return out


def device_count():
    n = C.c_int(0)
    rc = load().rt_device_count(C.byref(n))
    return n.value if rc == RT_OK else 0


def make_params(width, height, num_rays=1, num_bounce=0, depth_convention=0, sigma=0.0, eps=1e-3, tri_tmin=1e-4,
                seed=123456, variant="auto"):
    p = Params()
    p.width, p.height, p.num_rays, p.num_bounce = width, height, num_rays, num_bounce
    p.depth_convention, p.sigma, p.eps, p.tri_tmin, p.seed = depth_convention, sigma, eps, tri_tmin, seed
    p.variant = VARIANTS[variant] if isinstance(variant, str) else int(variant)
    return p


def interleaved_rows(height, tile_rows, rank, world):
    """Row tiles k*world+rank of `tile_rows` rows each.  Returns (Rows, image row indices)."""
    n_tiles = (height + tile_rows - 1) // tile_rows
    mine = list(range(rank, n_tiles, world))
    idx = np.concatenate([np.arange(t * tile_rows, min((t + 1) * tile_rows, height)) for t in mine]) if mine \
        else np.zeros(0, np.int64)
    r = Rows(rank * tile_rows, len(idx), tile_rows, world)
    return r, idx


def _marshal_scene(spheres, mesh, light, camera):
    """C structs of a scene description (shared by Context and MultiContext)."""
    arr = (Sphere * max(len(spheres), 1))()
    for i, s in enumerate(spheres):
        c, r, a = s[0], s[1], s[2]
        arr[i].center[:] = c
        arr[i].radius = r
        arr[i].albedo[:] = a
        arr[i].mirror = int(s[3]) if len(s) > 3 else 0
        arr[i].in_refraction_index = s[4] if len(s) > 4 else 1.0
        arr[i].out_refraction_index = s[5] if len(s) > 5 else 1.0
    meshes = [] if mesh is None else (list(mesh) if isinstance(mesh, (list, tuple)) else [mesh])
    marr, keep = (Mesh * max(len(meshes), 1))(), []
    taken = {d["object_slot"] for d in meshes if d.get("object_slot") is not None}
    free = (k for k in range(len(spheres) + len(meshes)) if k not in taken)
    for m, d in zip(marr, meshes):
        v = np.ascontiguousarray(d["vertices"], np.float32).reshape(-1, 3)
        ix = np.ascontiguousarray(d["indices"], np.int32)
        stride = ix.shape[1] if ix.ndim == 2 else 3
        bv = np.ascontiguousarray(d["bvh_arr10"], np.float32).reshape(-1, 10)
        m.vertices = v.ctypes.data_as(C.POINTER(C.c_float)); m.n_vertices = len(v)
        m.indices = ix.ctypes.data_as(C.POINTER(C.c_int32)); m.index_stride = stride
        m.n_triangles = ix.size // stride
        m.bvh_arr10 = bv.ctypes.data_as(C.POINTER(C.c_float)); m.n_nodes = len(bv)
        m.albedo[:] = d.get("albedo", (0.25, 0.25, 0.25))
        # a lone mesh without a slot is added last (cpu:685); several meshes without slots follow the spheres in list order
        m.object_slot = d["object_slot"] if d.get("object_slot") is not None else (len(spheres) if len(meshes) == 1 else next(free))
        m.mirror = int(d.get("mirror", 0))
        m.in_refraction_index = d.get("in_refraction_index", 1.0)
        m.out_refraction_index = d.get("out_refraction_index", 1.0)
        keep.append((v, ix, bv))
    lt = Light(); lt.position[:] = light[0]; lt.intensity = light[1]
    cam = Camera(); cam.position[:] = camera[0]
    # float alpha = PI/3 (cpu:666)
    cam.fov = np.float32(np.pi / 3) if camera[1] is None else np.float32(camera[1])
    return arr, len(spheres), marr, len(meshes), lt, cam, keep


class Context:
    """One rt_ctx = one GPU.  Methods mirror the C-ABI one to one."""

    def __init__(self, device_id=0):
        self._L = load()
        self._h = C.c_void_p()
        rc = self._L.rt_ctx_create(C.byref(self._h), device_id)
        if rc != RT_OK:
            raise RtError(rc, self._L.rt_last_error(None).decode())
        self._keep = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rt_ctx_destroy(self._h)                           # synchronises the copy streams: no frame is in flight afterwards
            self._h = C.c_void_p()
        if getattr(self, "_async_out", None):
            self._async_out.clear()

    __del__ = close

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self._L.rt_last_error(self._h).decode())

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self._L.rt_device_name(self._h, buf, 256))
        return buf.value.decode()

    def scene_upload(self, spheres, mesh=None, light=((-10.0, 20.0, 40.0), 3e10), camera=((0.0, 0.0, 55.0), None)):
        """spheres: iterable of (center, radius, albedo[, mirror, n_in, n_out]);
        mesh: dict(vertices, indices, bvh_arr10, albedo, object_slot[, mirror, in_refraction_index, out_refraction_index]) with the reference's array
        layouts, or a list of such dicts (several TriangleMesh objects in Scene::objects: rt_scene_upload_meshes)."""
        arr, n, marr, nm, lt, cam, self._keep = _marshal_scene(spheres, mesh, light, camera)
        if nm <= 1 and not isinstance(mesh, (list, tuple)):
            self._check(self._L.rt_scene_upload(self._h, arr, n, marr if nm else None, C.byref(lt), C.byref(cam)))
        else:
            self._check(self._L.rt_scene_upload_meshes(self._h, arr, n, marr, nm, C.byref(lt), C.byref(cam)))

    def render(self, params, row_begin=0, row_end=None, out=None):
        """out: optional preallocated [rows, W, 4] float32 array (e.g. PinnedArray(...).array)."""
        row_end = params.height if row_end is None else row_end
        if out is None:
            out = np.empty((max(row_end - row_begin, 0), params.width, 4), np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == max(row_end - row_begin, 0) * params.width * 4
        self._check(self._L.rt_render(self._h, C.byref(params), row_begin, row_end, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def render_rgb8(self, params, row_begin=0, row_end=None):
        row_end = params.height if row_end is None else row_end
        out = np.empty((max(row_end - row_begin, 0), params.width, 3), np.uint8)
        self._check(self._L.rt_render_rgb8(self._h, C.byref(params), row_begin, row_end, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def render_device(self, params, rows, out_ptr, stream=None):
        """Asynchronous render into device memory (e.g. a torch tensor's data_ptr())."""
        self._check(self._L.rt_render_device(self._h, C.byref(params), C.byref(rows), C.c_void_p(out_ptr),
                                             C.c_void_p(stream) if stream else None))

    def render_device_batch(self, params, rows, frames, stream=None):
        """rt_render_device_batch: `frames` = iterable of (out_ptr, camera_position, fov or None, seed); ONE launch chain traces them all (num_rays == 1)."""
        frames = list(frames)
        arr = (FrameDesc * max(len(frames), 1))()
        for d, (ptr, pos, fov, seed) in zip(arr, frames):
            d.camera.position[:] = pos
            d.camera.fov = np.float32(np.pi / 3) if fov is None else np.float32(fov)
            d.seed = int(seed)
            d.out_rgba_dev = int(ptr)
        self._check(self._L.rt_render_device_batch(self._h, C.byref(params), C.byref(rows), arr, len(frames), C.c_void_p(stream) if stream else None))

    def render_async(self, params, out, slot=0, rgb8=False):
        """rt_render_async: whole frame into device buffer `slot` (0 / 1), device-to-host copy into `out` on the copy stream;
        wait(slot) returns when `out` holds the frame.  out: [H, W, 4] float32 or, rgb8, [H, W, 3] uint8 (PinnedArray: one DMA)."""
        want = (np.uint8, 3) if rgb8 else (np.float32, 4)
        assert out.dtype == want[0] and out.flags.c_contiguous and out.size == params.height * params.width * want[1]
        # the copy stream writes into `out` until wait(slot): the context keeps the array (and through it a PinnedArray's block, which is
        # freed when its last view goes) alive for exactly that long -- a caller may drop its own reference at once
        if not hasattr(self, "_async_out"):
            self._async_out = {}
        self._async_out[int(slot)] = out
        self._check(self._L.rt_render_async(self._h, C.byref(params), int(slot), C.c_void_p(out.ctypes.data), 1 if rgb8 else 0))

    def wait(self, slot=0):
        try:
            self._check(self._L.rt_wait(self._h, int(slot)))
        finally:
            getattr(self, "_async_out", {}).pop(int(slot), None)

    def trace_rays(self, rays, tri_tmin=1e-4, variant="auto"):
        """rt_trace_rays: rays [n, 6] (O, u) through the production traversal kernel of `variant` -> [n, 5] (hit, t, N)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.empty((rays.shape[0], 5), np.float32)
        v = VARIANTS[variant] if isinstance(variant, str) else int(variant)
        self._check(self._L.rt_trace_rays(self._h, rays.ctypes.data_as(C.POINTER(C.c_float)), rays.shape[0], C.c_float(tri_tmin), v,
                                          out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def stats_enable(self, on=True):
        """trav_ms / trav_launches of stats() are measured only while enabled (production frames record no per-launch events)."""
        self._check(self._L.rt_stats_enable(self._h, 1 if on else 0))

    def set_pipelining(self, on=True):
        """rt_ctx_set_pipelining: consecutive render_device calls on one stream into alternating buffers overlap at the frame boundary."""
        self._check(self._L.rt_ctx_set_pipelining(self._h, 1 if on else 0))

    def tonemap_device(self, rgba_ptr, n_pixels, rgb8_ptr, stream=None):
        self._check(self._L.rt_tonemap_device(self._h, C.c_void_p(rgba_ptr), n_pixels, C.c_void_p(rgb8_ptr),
                                              C.c_void_p(stream) if stream else None))

    def count_work(self, params, row_begin=0, row_end=None, detail=False):
        """Traversal work of a frame from the counting instantiation of the kernel: rays, box tests, nodes, triangle
        tests -- what the oracle counts.  detail=True adds the tests the literal divisions decided and the work-stack kernel's
        step counters."""
        row_end = params.height if row_end is None else row_end
        w = Work()
        self._check(self._L.rt_count_work(self._h, C.byref(params), row_begin, row_end, C.byref(w)))
        out = {k: int(getattr(w, k)) for k in ("rays", "box_tests", "nodes", "tri_tests")}
        if not detail:
            return out
        out.update(box_literal=int(w.box_literal), tri_literal=int(w.tri_literal))
        out["steps"] = dict(zip(("iterations", "refill_passes", "refill_rounds", "fetches", "tri_steps", "box_steps", "literal_box_fallbacks", "serial_drains",
                                 "tdiv_blocks", "leaf_push_blocks", "leaf_push2_blocks", "anyhit_stop_steps"),
                                (int(v) for v in w.steps)))
        return out

    def mesh_transform(self, rotation, translation):
        """Device-side `transform` kernel (global_launcher.cu:340-365) on the uploaded mesh + triangle precompute + BVH refit."""
        r = np.ascontiguousarray(rotation, np.float32).reshape(9)
        t = np.ascontiguousarray(translation, np.float32).reshape(3)
        self._check(self._L.rt_mesh_transform(self._h, r.ctypes.data_as(C.POINTER(C.c_float)), t.ctypes.data_as(C.POINTER(C.c_float))))

    def mesh_rebuild(self, n_triangles, mode="reference"):
        """Device-side BVH build over the uploaded triangles and the current device vertices -> (bvh_arr10 [n_nodes, 10], order [n_triangles]).
        mode "reference": TriangleMesh::buildBVH bit for bit; "lbvh": Morton sort + parallel hierarchy, leaves cut by the surface-area heuristic (at most 32 triangles)."""
        arr = np.zeros(((2 * n_triangles + 2), 10), np.float32)
        order = np.zeros(n_triangles, np.int32)
        n = C.c_int32(0)
        self._check(self._L.rt_mesh_rebuild_mode(self._h, BVH_MODES[mode], arr.ctypes.data_as(C.POINTER(C.c_float)), order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)))
        return arr[:n.value].copy(), order

    def build_stats(self):
        s = BuildStats()
        self._check(self._L.rt_mesh_build_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in BuildStats._fields_}

    def mesh_set_normals(self, normals, nidx):
        """Smooth shading: vertex normals + per-triangle (ni, nj, nk) rows in the order of the uploaded indices; None = flat."""
        if normals is None:
            self._check(self._L.rt_mesh_set_normals(self._h, None, 0, None, 3, 0))
            return
        n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        ix = np.ascontiguousarray(nidx, np.int32).reshape(-1, 3)
        self._check(self._L.rt_mesh_set_normals(self._h, n.ctypes.data_as(C.POINTER(C.c_float)), len(n),
                                                ix.ctypes.data_as(C.POINTER(C.c_int32)), 3, len(ix)))

    def render_pose(self, params, pose):
        """One frame with realtime_render.cu's posed camera and per-sample averaging (no accumulation)."""
        out = np.empty((params.height, params.width, 4), np.float32)
        self._check(self._L.rt_render_pose(self._h, C.byref(params), C.byref(pose), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def progressive_reset(self):
        self._check(self._L.rt_progressive_reset(self._h))

    def progressive_frame(self, params, pose):
        """frames++, render with seed WangHash(frames), accumulate; returns (display float4, rgb8)."""
        disp = np.empty((params.height, params.width, 4), np.float32)
        rgb8 = np.empty((params.height, params.width, 3), np.uint8)
        self._check(self._L.rt_progressive_frame(self._h, C.byref(params), C.byref(pose), disp.ctypes.data_as(C.POINTER(C.c_float)),
                                                 rgb8.ctypes.data_as(C.POINTER(C.c_uint8))))
        return disp, rgb8

    def progressive_frames(self):
        n = C.c_int(0)
        self._check(self._L.rt_progressive_frames(self._h, C.byref(n)))
        return n.value

    def synchronize(self):
        self._check(self._L.rt_synchronize(self._h))

    def selfcheck(self):
        """Every device buffer of the context lives on the context's device."""
        self._check(self._L.rt_ctx_selfcheck(self._h))

    def _kat(self, fn, rows, width, owidth, *extra, counts=True):
        a = np.ascontiguousarray(rows, np.float32).reshape(-1, width)
        out = np.zeros((len(a), owidth), np.float32)
        c = KatCounts()
        fp = C.POINTER(C.c_float)
        args = [self._h, a.ctypes.data_as(fp), len(a), *extra, out.ctypes.data_as(fp)] + ([C.byref(c)] if counts else [])
        self._check(fn(*args))
        return out, {k: int(getattr(c, k)) for k, _ in KatCounts._fields_}

    def kat_sphere(self, rows):
        return self._kat(self._L.rt_kat_sphere, rows, 10, 5, counts=False)[0]

    def kat_sqrt(self, x):
        return self._kat(self._L.rt_kat_sqrt, x, 1, 1, counts=False)[0][:, 0]

    def kat_box(self, rows, route):
        out, c = self._kat(self._L.rt_kat_box, rows, 12, 1, int(route))
        return out[:, 0], c

    def kat_triangle(self, rows):
        return self._kat(self._L.rt_kat_triangle, rows, 15, 5)

    def kat_mesh(self, rows, tri_tmin=1e-4, route=0):
        return self._kat(self._L.rt_kat_mesh, rows, 6, 5, C.c_float(tri_tmin), int(route))

    def layout_hash(self):
        """rt_kat_layout_hash -> dict(pairs, fixed_pairs, quads, leaf_boxes) of 64-bit hashes (0 = that layout is not in use)."""
        h = (C.c_uint64 * 4)()
        self._check(self._L.rt_kat_layout_hash(self._h, h))
        return dict(zip(("pairs", "fixed_pairs", "quads", "leaf_boxes"), (int(x) for x in h)))

    def stats_after_render(self, params):
        """Render one frame with `params` and return rt_get_stats of it (which traversal kernel ran: travq_mode)."""
        self.render(params)
        return self.stats()

    def stats(self):
        s = Stats()
        self._check(self._L.rt_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}


class MultiContext:
    """One host process, several devices (rt_multi_*): interleaved 8-row tiles, peer copies to device_ids[0]."""

    def __init__(self, device_ids):
        self._L = load()
        self._h = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        rc = self._L.rt_multi_create(C.byref(self._h), ids, len(device_ids))
        if rc != RT_OK:
            raise RtError(rc, self._L.rt_multi_last_error(None).decode())
        self._keep = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rt_multi_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self._L.rt_multi_last_error(self._h).decode())

    def scene_upload(self, spheres, mesh=None, light=((-10.0, 20.0, 40.0), 3e10), camera=((0.0, 0.0, 55.0), None)):
        arr, n, marr, nm, lt, cam, self._keep = _marshal_scene(spheres, mesh, light, camera)
        if nm <= 1 and not isinstance(mesh, (list, tuple)):
            self._check(self._L.rt_multi_scene_upload(self._h, arr, n, marr if nm else None, C.byref(lt), C.byref(cam)))
        else:
            self._check(self._L.rt_multi_scene_upload_meshes(self._h, arr, n, marr, nm, C.byref(lt), C.byref(cam)))

    def render(self, params):
        out = np.empty((params.height, params.width, 4), np.float32)
        self._check(self._L.rt_render_multi(self._h, C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def render_device(self, params, out_ptr):
        self._check(self._L.rt_render_multi_device(self._h, C.byref(params), C.c_void_p(out_ptr)))

    def render_rgb8(self, params):
        """Every device tonemaps its tiles; the exchange moves the 8-bit image (3 bytes per pixel)."""
        out = np.empty((params.height, params.width, 3), np.uint8)
        self._check(self._L.rt_render_multi_rgb8(self._h, C.byref(params), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def stats(self):
        s = MultiStats()
        self._check(self._L.rt_multi_get_stats(self._h, C.byref(s)))
        n = s.n_devices
        return {"n_devices": n, "device_id": list(s.device_id)[:n], "kernel_ms": list(s.kernel_ms)[:n],
                "gather_ms": s.gather_ms, "frame_ms": s.frame_ms, "rays": int(s.rays), "gather_bytes": int(s.gather_bytes),
                "peer_access": list(s.peer_access)[:n], "submit_ms": s.submit_ms}
