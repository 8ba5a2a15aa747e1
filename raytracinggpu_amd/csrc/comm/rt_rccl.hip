// rt_rccl.hip -- libraytrace_rccl.so: the tile gather of the one-process-per-GPU render path over RCCL (include/raytrace_rccl.h).
// The root receives every peer tile straight into its place in the frame: an 8-row tile is one contiguous range of the frame, so
// the exchange needs neither a staging buffer nor a de-interleave pass ("direct placement with per-tile recv offsets").
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../../include/raytrace_rccl.h"

struct rt_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint64_t last_bytes = 0;
    std::string err;
};

namespace {

thread_local std::string g_err;

int fail(rt_comm *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (c) c->err = buf;
    return code;
}

#define RC_HIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, RT_COMM_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define RC_NCCL(c, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail(c, RT_COMM_ERR_RCCL, "%s: %s", #call, ncclGetErrorString(r_)); } while (0)

}  // namespace

extern "C" int rt_comm_abi_version(void) { return RT_COMM_ABI_VERSION; }

extern "C" int rt_comm_id_create(unsigned char *id) {
    if (!id) return fail(nullptr, RT_COMM_ERR_INVALID, "id is NULL");
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES is the size of an ncclUniqueId");
    ncclUniqueId u;
    RC_NCCL(nullptr, ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return RT_COMM_OK;
}

extern "C" int rt_comm_create(rt_comm **out, int device, int rank, int world, const unsigned char *id) {
    if (!out) return fail(nullptr, RT_COMM_ERR_INVALID, "comm is NULL");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(nullptr, RT_COMM_ERR_INVALID, "bad rank %d of %d (or id is NULL)", rank, world);
    int n_dev = 0;
    RC_HIP(nullptr, hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return fail(nullptr, RT_COMM_ERR_INVALID, "device %d: the process sees %d", device, n_dev);
    RC_HIP(nullptr, hipSetDevice(device));
    rt_comm *c = new rt_comm;
    c->device = device; c->rank = rank; c->world = world;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(nullptr, RT_COMM_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(e)); }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    const ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, RT_COMM_ERR_RCCL, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, ncclGetErrorString(r));
    }
    *out = c;
    return RT_COMM_OK;
}

extern "C" int rt_comm_destroy(rt_comm *c) {
    if (!c) return RT_COMM_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_COMM_OK;
}

extern "C" const char *rt_comm_last_error(const rt_comm *c) { return c ? c->err.c_str() : g_err.c_str(); }
extern "C" int rt_comm_rank(const rt_comm *c) { return c ? c->rank : -1; }
extern "C" int rt_comm_world(const rt_comm *c) { return c ? c->world : 0; }
extern "C" void *rt_comm_stream(const rt_comm *c) { return c ? static_cast<void *>(c->stream) : nullptr; }
extern "C" uint64_t rt_comm_last_bytes(const rt_comm *c) { return c ? c->last_bytes : 0; }

extern "C" int rt_comm_tile_plan(int W, int H, int bpp, int tile_rows, int world, int t, rt_comm_tile *out) {
    if (!out || W < 1 || H < 1 || bpp < 1 || tile_rows < 1 || world < 1 || t < 0 || (int64_t)t * tile_rows >= H)
        return fail(nullptr, RT_COMM_ERR_INVALID, "bad tile %d of a %dx%d frame in %d-row tiles over %d ranks", t, W, H, tile_rows, world);
    const size_t row_bytes = (size_t)W * (size_t)bpp;
    out->owner = t % world;
    out->rows = std::min(H, (t + 1) * tile_rows) - t * tile_rows;
    out->local_offset = (uint64_t)(t / world) * tile_rows * row_bytes;        // every tile in front of it in the owner's buffer is a full one
    out->frame_offset = (uint64_t)t * tile_rows * row_bytes;
    out->bytes = (uint64_t)out->rows * row_bytes;
    return RT_COMM_OK;
}

extern "C" int rt_comm_gather_tiles(rt_comm *c, const void *tiles_dev, int W, int H, int bpp, int tile_rows, int root, void *frame_dev, void *stream) {
    if (!c) return fail(nullptr, RT_COMM_ERR_INVALID, "comm is NULL");
    if (W < 1 || H < 1 || bpp < 1 || tile_rows < 1 || root < 0 || root >= c->world) return fail(c, RT_COMM_ERR_INVALID, "bad frame %dx%d, %d bytes per pixel, %d-row tiles, root %d", W, H, bpp, tile_rows, root);
    const int n_tiles = (H + tile_rows - 1) / tile_rows;
    if (n_tiles > RT_COMM_MAX_TILES) return fail(c, RT_COMM_ERR_INVALID, "%d tiles: at most %d per frame (use taller tiles)", n_tiles, RT_COMM_MAX_TILES);
    const bool is_root = c->rank == root;
    const bool own_any = c->rank < n_tiles;
    if (own_any && !tiles_dev) return fail(c, RT_COMM_ERR_INVALID, "tiles_dev is NULL");
    if (is_root && !frame_dev) return fail(c, RT_COMM_ERR_INVALID, "frame_dev is NULL on the root");
    RC_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const unsigned char *src = static_cast<const unsigned char *>(tiles_dev);
    unsigned char *dst = static_cast<unsigned char *>(frame_dev);
    auto plan = [&](int t) { rt_comm_tile p; (void)rt_comm_tile_plan(W, H, bpp, tile_rows, c->world, t, &p); return p; };
    c->last_bytes = 0;
    if (is_root)                                                     // the root's own tiles: device to device, same stream
        for (int t = c->rank; t < n_tiles; t += c->world) {
            const rt_comm_tile p = plan(t);
            RC_HIP(c, hipMemcpyAsync(dst + p.frame_offset, src + p.local_offset, p.bytes, hipMemcpyDeviceToDevice, s));
        }
    if (c->world == 1) return RT_COMM_OK;
    // one send per tile on the peers, the matching receive on the root, in tile order on both sides (point-to-point operations
    // between one pair of ranks match in the order they were issued); the group makes them one launch and lets all peers' transfers
    // progress together
    RC_NCCL(c, ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    for (int t = 0; t < n_tiles && r == ncclSuccess; ++t) {
        const rt_comm_tile p = plan(t);
        if (p.owner == root) continue;
        if (is_root) r = ncclRecv(dst + p.frame_offset, p.bytes, ncclUint8, p.owner, c->comm, s);
        else if (p.owner == c->rank) r = ncclSend(src + p.local_offset, p.bytes, ncclUint8, root, c->comm, s);
        else continue;
        c->last_bytes += p.bytes;
    }
    const ncclResult_t rg = ncclGroupEnd();
    if (r != ncclSuccess) return fail(c, RT_COMM_ERR_RCCL, "ncclSend / ncclRecv: %s", ncclGetErrorString(r));
    if (rg != ncclSuccess) return fail(c, RT_COMM_ERR_RCCL, "ncclGroupEnd: %s", ncclGetErrorString(rg));
    return RT_COMM_OK;
}

extern "C" int rt_comm_sync(rt_comm *c) {
    if (!c) return fail(nullptr, RT_COMM_ERR_INVALID, "comm is NULL");
    RC_HIP(c, hipSetDevice(c->device));
    RC_HIP(c, hipStreamSynchronize(c->stream));
    return RT_COMM_OK;
}
