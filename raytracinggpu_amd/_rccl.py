"""ctypes binding of libraytrace_rccl.so (include/raytrace_rccl.h): the RCCL tile gather of the one-process-per-GPU path.

Plumbing for tests and bench.py's `--gather capi` leg.  No fallback transport: a missing library or a failing RCCL call raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libraytrace_rccl.so")
ID_BYTES = 128

# every symbol include/raytrace_rccl.h declares (tests check the .so exports each)
EXPORTS = ["rt_comm_abi_version", "rt_comm_id_create", "rt_comm_create", "rt_comm_destroy", "rt_comm_last_error", "rt_comm_rank",
           "rt_comm_world", "rt_comm_stream", "rt_comm_gather_tiles", "rt_comm_tile_plan", "rt_comm_last_bytes", "rt_comm_sync",
           "rt_comm_set_plan", "rt_comm_last_plan", "rt_comm_choose_plan", "rt_comm_peer_plan"]
PLANS = {"auto": 0, "tile": 1, "coalesced": 2}
COALESCE_BELOW_DEFAULT = 256 * 1024


class Tile(C.Structure):
    _fields_ = [("owner", C.c_int32), ("rows", C.c_int32), ("local_offset", C.c_uint64), ("frame_offset", C.c_uint64), ("bytes", C.c_uint64)]


class Peer(C.Structure):
    _fields_ = [("stage_offset", C.c_uint64), ("bytes", C.c_uint64), ("n_tiles", C.c_int32), ("reserved", C.c_int32)]


class CommError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libraytrace_rccl: status {code}: {msg}")
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    try:                                  # one HIP runtime (and one RCCL) per process: let this library bind to the copies torch mapped
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.rt_comm_abi_version.restype = C.c_int
    L.rt_comm_id_create.argtypes = [C.c_char_p]
    L.rt_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.rt_comm_destroy.argtypes = [C.c_void_p]
    L.rt_comm_last_error.argtypes = [C.c_void_p]
    L.rt_comm_last_error.restype = C.c_char_p
    L.rt_comm_rank.argtypes = [C.c_void_p]
    L.rt_comm_world.argtypes = [C.c_void_p]
    L.rt_comm_stream.argtypes = [C.c_void_p]
    L.rt_comm_stream.restype = C.c_void_p
    L.rt_comm_gather_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.rt_comm_tile_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Tile)]
    L.rt_comm_last_bytes.argtypes = [C.c_void_p]
    L.rt_comm_last_bytes.restype = C.c_uint64
    L.rt_comm_sync.argtypes = [C.c_void_p]
    L.rt_comm_set_plan.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    L.rt_comm_last_plan.argtypes = [C.c_void_p]
    L.rt_comm_choose_plan.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int]
    L.rt_comm_peer_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Peer)]
    _lib = L
    return L


def unique_id():
    """Rank 0: the 128 bytes every rank of the communicator must be given."""
    L = load()
    buf = C.create_string_buffer(ID_BYTES)
    rc = L.rt_comm_id_create(buf)
    if rc != 0:
        raise CommError(rc, L.rt_comm_last_error(None).decode())
    return buf.raw


def tile_plan(W, H, bytes_per_pixel, tile_rows, world, t):
    """Where rt_comm_gather_tiles takes tile t from and puts it (host arithmetic only)."""
    L = load()
    out = Tile()
    rc = L.rt_comm_tile_plan(W, H, bytes_per_pixel, tile_rows, world, t, C.byref(out))
    if rc != 0:
        raise CommError(rc, L.rt_comm_last_error(None).decode())
    return out


def peer_plan(W, H, bytes_per_pixel, tile_rows, world, root, peer):
    """The coalesced plan for one rank: the single message it sends and where the root stages it (host arithmetic only)."""
    L = load()
    out = Peer()
    rc = L.rt_comm_peer_plan(W, H, bytes_per_pixel, tile_rows, world, root, peer, C.byref(out))
    if rc != 0:
        raise CommError(rc, L.rt_comm_last_error(None).decode())
    return out


def choose_plan(plan, W, bytes_per_pixel, tile_rows, coalesce_below=0):
    """What `plan` ("auto" | "tile" | "coalesced") resolves to for a frame: "tile" or "coalesced"."""
    v = load().rt_comm_choose_plan(PLANS[plan], int(coalesce_below), W, bytes_per_pixel, tile_rows)
    return {1: "tile", 2: "coalesced"}[v]


class Comm:
    def __init__(self, device, rank, world, uid):
        L = load()
        if len(uid) != ID_BYTES:
            raise ValueError("a communicator id is %d bytes" % ID_BYTES)
        h = C.c_void_p()
        rc = L.rt_comm_create(C.byref(h), int(device), int(rank), int(world), uid)
        if rc != 0:
            raise CommError(rc, L.rt_comm_last_error(None).decode())
        self._L, self._h = L, h
        self.rank, self.world = int(rank), int(world)

    def _check(self, rc):
        if rc != 0:
            raise CommError(rc, self._L.rt_comm_last_error(self._h).decode())

    @property
    def stream(self):
        """The communicator's hipStream_t (an integer): rt_render_device / rt_tonemap_device take it as their stream."""
        return self._L.rt_comm_stream(self._h)

    def gather_tiles(self, tiles_ptr, W, H, bytes_per_pixel, frame_ptr=None, tile_rows=8, root=0, stream=None):
        self._check(self._L.rt_comm_gather_tiles(self._h, tiles_ptr, W, H, bytes_per_pixel, tile_rows, root, frame_ptr, stream))

    def sync(self):
        self._check(self._L.rt_comm_sync(self._h))

    def set_plan(self, plan="auto", coalesce_below=0):
        """Every rank of the communicator must set the same plan ("auto" | "tile" | "coalesced") and threshold."""
        self._check(self._L.rt_comm_set_plan(self._h, PLANS[plan], int(coalesce_below)))

    @property
    def last_plan(self):
        return {1: "tile", 2: "coalesced"}.get(self._L.rt_comm_last_plan(self._h))

    @property
    def last_bytes(self):
        return int(self._L.rt_comm_last_bytes(self._h))

    def close(self):
        if self._h:
            self._L.rt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
