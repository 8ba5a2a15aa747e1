// rt_rccl.hip -- libraytrace_rccl.so: the tile gather of the one-process-per-GPU render path over RCCL (include/raytrace_rccl.h).
// Two plans.  PER_TILE: the root receives every peer tile straight into its place in the frame -- an 8-row tile is one contiguous
// range of the frame, so the exchange needs neither a staging buffer nor a de-interleave pass ("direct placement with
// per-tile recv offsets").  COALESCED: every peer sends its whole dense tile buffer as ONE message into a staging area on the root
// and one kernel puts all tiles (the root's own included) into place -- fewer, bigger messages for small tiles (1920x1080 RGB8 at
// world 8: one message of 783 KB per peer instead of 17 of 46 KB; SURVEY 8e, "a small de-interleave kernel on rank 0").
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
    hipStream_t last_stream = nullptr;       // the stream the last gather was issued on (rt_comm_sync waits for THAT one)
    uint64_t last_bytes = 0;
    int plan = RT_COMM_PLAN_AUTO, last_plan = RT_COMM_PLAN_PER_TILE;
    uint64_t coalesce_below = RT_COMM_COALESCE_BELOW_DEFAULT;   // AUTO: a full tile smaller than this many bytes => the coalesced plan
    void *stage = nullptr;                   // root, coalesced plan: the peers' dense buffers side by side
    size_t stage_bytes = 0;
    hipEvent_t stage_read = nullptr;         // root, coalesced plan: recorded behind the placement kernel that reads `stage`; the next gather -- on whatever stream --
    bool stage_busy = false;                 // waits for it before a receive may overwrite the area (gathers of two frames in flight on two streams)
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

// The coalesced plan's second half: frame byte b belongs to tile t = b / tile_bytes, which rank t % world rendered as the (t / world)-th
// tile of its dense buffer.  One thread per WORD of the frame (16, 4 or 1 bytes: the widest that divides the tile size and the frame
// size; sources are 256-byte aligned).  src[r] = rank r's dense buffer as the root sees it (its own tiles in place, the peers' in the
// staging area).
struct SrcTable { const unsigned char *src[RT_COMM_MAX_WORLD]; };
template <typename WORD>
__global__ __launch_bounds__(256) void place_tiles_kernel(const SrcTable tab, unsigned char *__restrict__ frame, const uint64_t n_words,
                                                          const uint64_t tile_words, const int world) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const uint64_t t = i / tile_words, r = i - t * tile_words;
    const int owner = (int)(t % (uint64_t)world);
    const uint64_t k = t / (uint64_t)world;
    reinterpret_cast<WORD *>(frame)[i] = reinterpret_cast<const WORD *>(tab.src[owner])[k * tile_words + r];
}

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
    if (c->last_stream && c->last_stream != c->stream) (void)hipStreamSynchronize(c->last_stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stage_read) { (void)hipEventSynchronize(c->stage_read); (void)hipEventDestroy(c->stage_read); }
    if (c->stage) (void)hipFree(c->stage);
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

extern "C" int rt_comm_set_plan(rt_comm *c, int plan, uint64_t coalesce_below_bytes) {
    if (!c) return fail(nullptr, RT_COMM_ERR_INVALID, "comm is NULL");
    if (plan != RT_COMM_PLAN_AUTO && plan != RT_COMM_PLAN_PER_TILE && plan != RT_COMM_PLAN_COALESCED) return fail(c, RT_COMM_ERR_INVALID, "unknown plan %d", plan);
    c->plan = plan;
    c->coalesce_below = coalesce_below_bytes ? coalesce_below_bytes : RT_COMM_COALESCE_BELOW_DEFAULT;
    return RT_COMM_OK;
}
extern "C" int rt_comm_last_plan(const rt_comm *c) { return c ? c->last_plan : RT_COMM_ERR_INVALID; }

extern "C" int rt_comm_choose_plan(int plan, uint64_t coalesce_below_bytes, int W, int bpp, int tile_rows) {
    if (plan == RT_COMM_PLAN_PER_TILE || plan == RT_COMM_PLAN_COALESCED) return plan;
    const uint64_t below = coalesce_below_bytes ? coalesce_below_bytes : RT_COMM_COALESCE_BELOW_DEFAULT;
    return (uint64_t)W * (uint64_t)bpp * (uint64_t)tile_rows < below ? RT_COMM_PLAN_COALESCED : RT_COMM_PLAN_PER_TILE;
}

extern "C" int rt_comm_peer_plan(int W, int H, int bpp, int tile_rows, int world, int root, int peer, rt_comm_peer *out) {
    if (!out || W < 1 || H < 1 || bpp < 1 || tile_rows < 1 || world < 1 || world > RT_COMM_MAX_WORLD || root < 0 || root >= world || peer < 0 || peer >= world)
        return fail(nullptr, RT_COMM_ERR_INVALID, "bad peer %d of %d (root %d) for a %dx%d frame in %d-row tiles", peer, world, root, W, H, tile_rows);
    const uint64_t row_bytes = (uint64_t)W * (uint64_t)bpp;
    const int n_tiles = (H + tile_rows - 1) / tile_rows;
    auto dense_bytes = [&](int r) {                                 // rows of rank r's tiles x row_bytes
        uint64_t rows = 0;
        for (int t = r; t < n_tiles; t += world) rows += (uint64_t)(std::min(H, (t + 1) * tile_rows) - t * tile_rows);
        return rows * row_bytes;
    };
    uint64_t off = 0;
    for (int r = 0; r < peer; ++r) if (r != root) off += (dense_bytes(r) + 255) / 256 * 256;   // 256-byte aligned pieces, rank order, the root left out
    out->n_tiles = peer < n_tiles ? (n_tiles - peer + world - 1) / world : 0;
    out->bytes = dense_bytes(peer);
    out->stage_offset = peer == root ? 0 : off;
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
    c->last_stream = s;
    c->err.clear();
    if (c->world > RT_COMM_MAX_WORLD) return fail(c, RT_COMM_ERR_INVALID, "world %d: at most %d ranks", c->world, RT_COMM_MAX_WORLD);
    const int use = c->world == 1 ? RT_COMM_PLAN_PER_TILE : rt_comm_choose_plan(c->plan, c->coalesce_below, W, bpp, tile_rows);
    c->last_plan = use;
    if (use == RT_COMM_PLAN_COALESCED) {
        // every rank decides alike (same arguments, same threshold): one message per peer
        rt_comm_peer me;
        (void)rt_comm_peer_plan(W, H, bpp, tile_rows, c->world, root, c->rank, &me);
        if (is_root) {
            rt_comm_peer last;
            int last_peer = c->world - 1;
            if (last_peer == root) --last_peer;
            (void)rt_comm_peer_plan(W, H, bpp, tile_rows, c->world, root, last_peer, &last);
            const size_t need = (size_t)last.stage_offset + (size_t)((last.bytes + 255) / 256 * 256) + 256;
            if (!c->stage_read) RC_HIP(c, hipEventCreateWithFlags(&c->stage_read, hipEventDisableTiming));
            if (c->stage_bytes < need) {
                RC_HIP(c, hipStreamSynchronize(s));                      // an earlier gather may still read the old staging area
                if (c->stage_busy) { RC_HIP(c, hipEventSynchronize(c->stage_read)); c->stage_busy = false; }   // ... also one issued on another stream
                if (c->stage) (void)hipFree(c->stage);
                c->stage = nullptr; c->stage_bytes = 0;
                RC_HIP(c, hipMalloc(&c->stage, need));
                c->stage_bytes = need;
            }
        }
        // ONE staging area per communicator: this gather's receives must not land in it while the previous gather's placement kernel -- possibly on
        // another stream: bench.py keeps two frames in flight on two streams -- still reads it (RCCL orders only its own kernels)
        if (is_root && c->stage_busy) RC_HIP(c, hipStreamWaitEvent(s, c->stage_read, 0));
        RC_NCCL(c, ncclGroupStart());
        ncclResult_t r = ncclSuccess;
        SrcTable tab{};
        if (is_root) {
            for (int peer = 0; peer < c->world && r == ncclSuccess; ++peer) {
                rt_comm_peer pp;
                (void)rt_comm_peer_plan(W, H, bpp, tile_rows, c->world, root, peer, &pp);
                if (peer == root) { tab.src[peer] = src; continue; }
                tab.src[peer] = static_cast<unsigned char *>(c->stage) + pp.stage_offset;
                if (pp.bytes == 0) continue;                             // more ranks than tiles: this peer holds none
                r = ncclRecv(static_cast<unsigned char *>(c->stage) + pp.stage_offset, pp.bytes, ncclUint8, peer, c->comm, s);
                c->last_bytes += pp.bytes;
            }
        } else if (me.bytes > 0) {
            r = ncclSend(src, me.bytes, ncclUint8, root, c->comm, s);
            c->last_bytes += me.bytes;
        }
        const ncclResult_t rg = ncclGroupEnd();
        if (r != ncclSuccess) return fail(c, RT_COMM_ERR_RCCL, "ncclSend / ncclRecv: %s", ncclGetErrorString(r));
        if (rg != ncclSuccess) return fail(c, RT_COMM_ERR_RCCL, "ncclGroupEnd: %s", ncclGetErrorString(rg));
        if (is_root) {                                                   // all tiles into place, the root's own included: one launch
            const uint64_t tile_bytes = (uint64_t)W * bpp * tile_rows, frame_bytes = (uint64_t)W * bpp * H;
            const uintptr_t align = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst);
            const int ws = (tile_bytes % 16 == 0 && frame_bytes % 16 == 0 && align % 16 == 0) ? 16 : (tile_bytes % 4 == 0 && frame_bytes % 4 == 0 && align % 4 == 0) ? 4 : 1;
            const uint64_t n_words = frame_bytes / ws, tile_words = tile_bytes / ws;
            const dim3 grid((unsigned)((n_words + 255) / 256)), block(256);
            if (ws == 16) hipLaunchKernelGGL(place_tiles_kernel<uint4>, grid, block, 0, s, tab, dst, n_words, tile_words, c->world);
            else if (ws == 4) hipLaunchKernelGGL(place_tiles_kernel<uint32_t>, grid, block, 0, s, tab, dst, n_words, tile_words, c->world);
            else hipLaunchKernelGGL(place_tiles_kernel<unsigned char>, grid, block, 0, s, tab, dst, n_words, tile_words, c->world);
            RC_HIP(c, hipGetLastError());
            RC_HIP(c, hipEventRecord(c->stage_read, s));
            c->stage_busy = true;
        }
        return RT_COMM_OK;
    }
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
    if (c->last_stream && c->last_stream != c->stream) RC_HIP(c, hipStreamSynchronize(c->last_stream));   // a gather issued on a caller's stream
    RC_HIP(c, hipStreamSynchronize(c->stream));
    return RT_COMM_OK;
}
