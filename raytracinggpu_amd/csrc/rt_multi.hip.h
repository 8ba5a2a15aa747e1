// rt_multi.hip.h -- one host process driving several devices (rt_render_multi).
// Included at the end of rt_capi.hip (same translation unit: it launches through launch_render).
//
// The reference renders on the implicit CUDA device 0 (optimized.cu:828-856).  Here the frame is cut into 8-row
// tiles, tile k -> device k mod n (interleaved: the cat sits in the middle rows), the scene is replicated,
// every device renders its tiles into a dense local buffer on its own stream, each peer pushes that buffer over xGMI
// into the root device's staging area (hipMemcpyPeerAsync, one link per peer, no ring), and one kernel on the root
// restores row order.  The one-process-per-GPU path (bench.py, torch.distributed) does the same exchange with one
// RCCL gather instead; results are bitwise the single-device frame either way.

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace rtk {

struct MultiSrc { const float4 *base[RT_MAX_DEVICES]; };

// frame[row][x] <- the dense buffer of the device that rendered `row`
__global__ __launch_bounds__(256) void deinterleave_kernel(const MultiSrc src, int n_dev, int W, int H, int tile_rows, float4 *__restrict__ frame,
                                                           unsigned long long *__restrict__ rays) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float w = 0.f;
    if (i < (int64_t)W * H) {
        const int row = (int)(i / W), x = (int)(i - (int64_t)row * W);
        const int tile = row / tile_rows, k = tile % n_dev;
        const int lrow = (tile / n_dev) * tile_rows + row % tile_rows;
        const float4 v = src.base[k][(size_t)lrow * W + x];
        frame[i] = v;
        w = v.w;
    }
    // rays traced = sum of .w (small exact integers): wave sums, one atomic per wave
    const uint32_t s = wave_sum((uint32_t)w);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(rays, (unsigned long long)s);
}

// the same for the tonemapped image: out[row][x] (3 bytes) <- the dense RGB8 buffer of the device that rendered `row`.
// One thread per 4 output bytes (W * 3 is a multiple of 4 whenever W is; the tail is copied bytewise).
struct MultiSrc8 { const uint8_t *base[RT_MAX_DEVICES]; };
__global__ __launch_bounds__(256) void deinterleave_rgb8_kernel(const MultiSrc8 src, int n_dev, int W, int H, int tile_rows, uint8_t *__restrict__ frame) {
    const int64_t row_bytes = (int64_t)W * 3;
    const int64_t words_per_row = (row_bytes + 3) / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words_per_row * H) return;
    const int row = (int)(i / words_per_row);
    const int64_t b0 = (i - (int64_t)row * words_per_row) * 4;
    const int tile = row / tile_rows, k = tile % n_dev;
    const int lrow = (tile / n_dev) * tile_rows + row % tile_rows;
    const uint8_t *s = src.base[k] + (int64_t)lrow * row_bytes + b0;
    uint8_t *d = frame + (int64_t)row * row_bytes + b0;
    for (int j = 0; j < 4 && b0 + j < row_bytes; ++j) d[j] = s[j];
}

// rays traced by one device = sum of the .w channel of its tiles (small exact integers)
__global__ __launch_bounds__(256) void sum_rays_kernel(const float4 *__restrict__ rgba, int64_t npix, unsigned long long *__restrict__ rays) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = wave_sum(i < npix ? (uint32_t)rgba[i].w : 0u);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(rays, (unsigned long long)s);
}

}  // namespace rtk

// One submit thread per peer device (device 0 is driven by the calling thread): a frame's ~25 runtime calls per device -- launches,
// event records, the peer copy -- are issued by all devices' threads at once instead of one device after the other (with eight
// devices the last one used to start ~300 runtime calls late).  The threads live as long as the multi context.
struct MultiWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, busy = false, quit = false;
    void start() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                std::function<void()> j = std::move(job);
                has_job = false;
                lk.unlock();
                j();
                lk.lock();
                busy = false;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j); has_job = true; busy = true;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return !busy; });
    }
    void stop() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
        th.join();
    }
};

struct rt_multi {
    int n = 0;
    MultiWorker *worker[RT_MAX_DEVICES] = {};   // worker[k] submits device k's work (k >= 1)
    rt_ctx *ctx[RT_MAX_DEVICES] = {};
    DevBuf local[RT_MAX_DEVICES];           // dense tiles of device k (on device k)
    DevBuf local8[RT_MAX_DEVICES], rays_k[RT_MAX_DEVICES];   // RGB8 gather: tonemapped tiles and ray count of device k (on device k)
    int peer_access[RT_MAX_DEVICES] = {};   // 1: device k writes into the root's memory directly (peer access), 0: staged by the runtime, -1: k is the root's device
    DevBuf stage, frame, rays;              // on the root device (device of ctx[0])
    hipEvent_t done[RT_MAX_DEVICES] = {};   // device k's tiles have arrived on the root
    hipEvent_t g0 = nullptr, g1 = nullptr;
    rt_multi_stats stats{};
    std::string err;
};

namespace {

int mfail(rt_multi *m, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (m) m->err = buf;
    return code;
}
#define RT_MHIP(m, call)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return mfail(m, RT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// rgb8: every device tonemaps its tiles (cpu:714-716) and the exchange moves 3 bytes per pixel instead of 16
// (optimized.cu:856 copies the 8-bit image too); the float path stays for parity checks and ray counts per pixel.
int multi_render(rt_multi *m, const rt_params *p, void *out_dev_on_root, void *out_host, bool rgb8 = false) {
    if (!m) return mfail(nullptr, RT_ERR_INVALID, "multi context is NULL");
    if (!p) return mfail(m, RT_ERR_INVALID, "params is NULL");
    if (p->width <= 0 || p->height <= 0) return mfail(m, RT_ERR_INVALID, "width/height must be positive");
    const auto t_begin = std::chrono::steady_clock::now();
    const int n = m->n, W = p->width, H = p->height, R = RT_MULTI_TILE_ROWS;
    const int n_tiles = (H + R - 1) / R;
    rt_ctx *root = m->ctx[0];
    size_t off[RT_MAX_DEVICES + 1] = {0};
    int nrows[RT_MAX_DEVICES] = {0};
    for (int k = 0; k < n; ++k) {
        for (int t = k; t < n_tiles; t += n) nrows[k] += std::min(R, H - t * R);
        off[k + 1] = off[k] + (k == 0 ? 0 : (size_t)nrows[k] * W);          // device 0's tiles are read in place (units: pixels)
    }
    const size_t px_bytes = rgb8 ? 3 : sizeof(float4);
    auto stage_off = [&](int k) { return (off[k] * px_bytes + 15) / 16 * 16 + 16 * (size_t)k; };   // 16-byte aligned pieces
    int rc;
    RT_MHIP(m, hipSetDevice(root->device));
    const size_t frame_bytes = (size_t)W * H * px_bytes;
    if ((rc = ensure(root, m->stage, stage_off(n) + 16)) != RT_OK || (rc = ensure(root, m->rays, 8)) != RT_OK ||
        (!out_dev_on_root && (rc = ensure(root, m->frame, frame_bytes)) != RT_OK)) { m->err = root->err; return rc; }
    for (int k = 0; k < n; ++k)                                             // every device's own stream must exist: no silent fall-back to stream 0
        if (!own_stream(m->ctx[k])) return mfail(m, RT_ERR_HIP, "device %d: the context's stream: %s", m->ctx[k]->device, m->ctx[k]->err.c_str());
    RT_MHIP(m, hipSetDevice(root->device));
    RT_MHIP(m, hipMemsetAsync(m->rays.p, 0, 8, own_stream(root)));
    // 1. every device renders its tiles; peers push them to the root as soon as they are done.  A failure part-way
    //    leaves work in flight on the devices already launched: drain them before returning, so that the caller may
    //    free or reuse its buffers and the next call starts from idle streams.
    auto drain = [&](int launched) {
        for (int j = 0; j < launched; ++j) { (void)hipSetDevice(m->ctx[j]->device); (void)hipStreamSynchronize(own_stream(m->ctx[j])); }
        (void)hipSetDevice(root->device);
    };
    // each device's submission: returns a status, leaves its text in err_k[k] (the submit threads have their own thread-local error)
    int rc_k[RT_MAX_DEVICES] = {};
    std::string err_k[RT_MAX_DEVICES];
    auto submit = [&](int k) -> int {
        rt_ctx *c = m->ctx[k];
        int r;
        hipError_t e = hipSetDevice(c->device);
        if (e != hipSuccess) { err_k[k] = std::string("hipSetDevice: ") + hipGetErrorString(e); return RT_ERR_HIP; }
        if ((r = ensure(c, m->local[k], std::max<size_t>((size_t)nrows[k] * W, 1) * sizeof(float4))) != RT_OK) { err_k[k] = c->err; return r; }
        rt_rows rows{k * R, nrows[k], R, n};
        if ((r = launch_render(c, p, &rows, m->local[k].p, own_stream(c))) != RT_OK) { err_k[k] = c->err; return r; }
        const int64_t npix_k = (int64_t)nrows[k] * W;
        const void *piece = m->local[k].p;
        if (rgb8) {
            if ((r = ensure(c, m->local8[k], (size_t)std::max<int64_t>(npix_k, 1) * 3 + 16)) != RT_OK || (r = ensure(c, m->rays_k[k], 8)) != RT_OK) { err_k[k] = c->err; return r; }
            e = hipMemsetAsync(m->rays_k[k].p, 0, 8, own_stream(c));
            if (e == hipSuccess && npix_k > 0) {
                hipLaunchKernelGGL(rtk::sum_rays_kernel, dim3((unsigned)((npix_k + 255) / 256)), dim3(256), 0, own_stream(c),
                                   static_cast<const float4 *>(m->local[k].p), npix_k, static_cast<unsigned long long *>(m->rays_k[k].p));
                if ((r = launch_tonemap(c, m->local[k].p, npix_k, m->local8[k].p, own_stream(c))) != RT_OK) { err_k[k] = c->err; return r; }
            }
            if (e != hipSuccess) { err_k[k] = std::string("tonemap: ") + hipGetErrorString(e); return RT_ERR_HIP; }
            piece = m->local8[k].p;
        }
        if (k > 0) {
            if (nrows[k] > 0) e = hipMemcpyPeerAsync(static_cast<uint8_t *>(m->stage.p) + stage_off(k), root->device, piece, c->device,
                                                     (size_t)npix_k * px_bytes, own_stream(c));
            if (e == hipSuccess) e = hipEventRecord(m->done[k], own_stream(c));
            if (e != hipSuccess) { err_k[k] = std::string("tile exchange: ") + hipGetErrorString(e); return RT_ERR_HIP; }
        }
        return RT_OK;
    };
    for (int k = 1; k < n; ++k) m->worker[k]->submit([&, k] { rc_k[k] = submit(k); });
    rc_k[0] = submit(0);
    for (int k = 1; k < n; ++k) m->worker[k]->wait();
    m->stats.submit_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    for (int k = 0; k < n; ++k) {
        if (rc_k[k] != RT_OK) {
            drain(n);
            return mfail(m, rc_k[k], "device %d: %s", m->ctx[k]->device, err_k[k].c_str());
        }
    }
    // 2. root: wait for the peers, restore row order
    RT_MHIP(m, hipSetDevice(root->device));
    RT_MHIP(m, hipEventRecord(m->g0, own_stream(root)));
    for (int k = 1; k < n; ++k) RT_MHIP(m, hipStreamWaitEvent(own_stream(root), m->done[k], 0));
    void *frame = out_dev_on_root ? out_dev_on_root : m->frame.p;
    const int64_t npix = (int64_t)W * H;
    if (rgb8) {
        rtk::MultiSrc8 src{};
        src.base[0] = static_cast<const uint8_t *>(m->local8[0].p);
        for (int k = 1; k < n; ++k) src.base[k] = static_cast<const uint8_t *>(m->stage.p) + stage_off(k);
        const int64_t words = ((int64_t)W * 3 + 3) / 4 * H;
        hipLaunchKernelGGL(rtk::deinterleave_rgb8_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, own_stream(root), src, n, W, H, R, static_cast<uint8_t *>(frame));
    } else {
        rtk::MultiSrc src{};
        src.base[0] = static_cast<const float4 *>(m->local[0].p);
        for (int k = 1; k < n; ++k) src.base[k] = reinterpret_cast<const float4 *>(static_cast<const uint8_t *>(m->stage.p) + stage_off(k));
        hipLaunchKernelGGL(rtk::deinterleave_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, own_stream(root), src, n, W, H, R, static_cast<float4 *>(frame),
                           static_cast<unsigned long long *>(m->rays.p));
    }
    RT_MHIP(m, hipGetLastError());
    RT_MHIP(m, hipEventRecord(m->g1, own_stream(root)));
    unsigned long long rays = 0;
    if (!rgb8) RT_MHIP(m, hipMemcpyAsync(&rays, m->rays.p, 8, hipMemcpyDeviceToHost, own_stream(root)));
    if (out_host) RT_MHIP(m, hipMemcpyAsync(out_host, frame, frame_bytes, hipMemcpyDeviceToHost, own_stream(root)));
    RT_MHIP(m, hipStreamSynchronize(own_stream(root)));
    if (rgb8) {                                                       // every device counted its own rays; their streams are drained by now or here
        for (int k = 0; k < n; ++k) {
            unsigned long long rk = 0;
            RT_MHIP(m, hipSetDevice(m->ctx[k]->device));
            RT_MHIP(m, hipMemcpyAsync(&rk, m->rays_k[k].p, 8, hipMemcpyDeviceToHost, own_stream(m->ctx[k])));
            RT_MHIP(m, hipStreamSynchronize(own_stream(m->ctx[k])));
            rays += rk;
        }
        RT_MHIP(m, hipSetDevice(root->device));
    }
    // 3. statistics
    m->stats.n_devices = n;
    m->stats.rays = rays;
    m->stats.gather_bytes = (uint64_t)(off[n] * px_bytes);            // bytes the peers moved into the root device
    for (int k = 0; k < n; ++k) m->stats.peer_access[k] = m->peer_access[k];
    RT_MHIP(m, hipEventElapsedTime(&m->stats.gather_ms, m->g0, m->g1));
    for (int k = 0; k < n; ++k) {
        rt_stats s{};
        if ((rc = rt_get_stats(m->ctx[k], &s)) != RT_OK) { m->err = m->ctx[k]->err; return rc; }
        m->stats.kernel_ms[k] = s.kernel_ms;
        m->stats.device_id[k] = m->ctx[k]->device;
    }
    m->stats.frame_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_multi_create(rt_multi **out, const int *device_ids, int n_devices) {
    if (!out) return mfail(nullptr, RT_ERR_INVALID, "multi out-pointer is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > RT_MAX_DEVICES)
        return mfail(nullptr, RT_ERR_INVALID, "need 1..%d device ids", RT_MAX_DEVICES);
    rt_multi *m = new (std::nothrow) rt_multi();
    if (!m) return mfail(nullptr, RT_ERR_INVALID, "out of host memory");
    m->n = n_devices;
    for (int k = 0; k < n_devices; ++k) {
        const int rc = rt_ctx_create(&m->ctx[k], device_ids[k]);
        if (rc != RT_OK) { rt_multi_destroy(m); return rc; }
        m->ctx[k]->knobs.part_prio = 1;                                  // several contexts in one process: see Knobs::part_prio
    }
    for (int k = 1; k < n_devices; ++k) { m->worker[k] = new MultiWorker(); m->worker[k]->start(); }
    m->peer_access[0] = -1;                                               // the root itself
    hipError_t e = hipSetDevice(m->ctx[0]->device);
    if (e == hipSuccess) e = hipEventCreate(&m->g0);
    if (e == hipSuccess) e = hipEventCreate(&m->g1);
    for (int k = 1; k < n_devices && e == hipSuccess; ++k) {
        e = hipSetDevice(m->ctx[k]->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->done[k], hipEventDisableTiming);
        m->peer_access[k] = -1;                                                            // same device as the root: a local copy
        if (e == hipSuccess && m->ctx[k]->device != m->ctx[0]->device) {
            int can = 0;
            m->peer_access[k] = 0;                                                         // reported in rt_multi_stats: the copy is then staged by the runtime
            if (hipDeviceCanAccessPeer(&can, m->ctx[k]->device, m->ctx[0]->device) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(m->ctx[0]->device, 0);   // direct xGMI writes into the root's staging area
                if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) m->peer_access[k] = 1;
                if (pe != hipSuccess) (void)hipGetLastError();
            }
        }
    }
    if (e != hipSuccess) { const int code = mfail(nullptr, RT_ERR_HIP, "multi context: %s", hipGetErrorString(e)); rt_multi_destroy(m); return code; }
    *out = m;
    return RT_OK;
}

int rt_multi_destroy(rt_multi *m) {
    if (!m) return RT_OK;
    for (int k = 1; k < RT_MAX_DEVICES; ++k) if (m->worker[k]) { m->worker[k]->stop(); delete m->worker[k]; m->worker[k] = nullptr; }
    if (m->ctx[0]) {
        (void)hipSetDevice(m->ctx[0]->device);
        if (m->ctx[0]->stream_) (void)hipStreamSynchronize(m->ctx[0]->stream_);
        m->stage.release(); m->frame.release(); m->rays.release();
        if (m->g0) (void)hipEventDestroy(m->g0);
        if (m->g1) (void)hipEventDestroy(m->g1);
    }
    for (int k = 0; k < m->n; ++k) {
        if (!m->ctx[k]) continue;
        (void)hipSetDevice(m->ctx[k]->device);
        if (m->ctx[k]->stream_) (void)hipStreamSynchronize(m->ctx[k]->stream_);
        m->local[k].release(); m->local8[k].release(); m->rays_k[k].release();
        if (m->done[k]) (void)hipEventDestroy(m->done[k]);
        rt_ctx_destroy(m->ctx[k]);
    }
    delete m;
    return RT_OK;
}

const char *rt_multi_last_error(const rt_multi *m) { return m ? m->err.c_str() : g_last_error.c_str(); }

int rt_multi_scene_upload_meshes(rt_multi *m, const rt_sphere *spheres, int n_spheres, const rt_mesh *meshes, int n_meshes,
                                 const rt_light *light, const rt_camera *camera) {
    if (!m) return mfail(nullptr, RT_ERR_INVALID, "multi context is NULL");
    for (int k = 0; k < m->n; ++k) {
        const int rc = rt_scene_upload_meshes(m->ctx[k], spheres, n_spheres, meshes, n_meshes, light, camera);
        if (rc != RT_OK) { m->err = m->ctx[k]->err; return rc; }
    }
    return RT_OK;
}

int rt_multi_scene_upload(rt_multi *m, const rt_sphere *spheres, int n_spheres, const rt_mesh *mesh,
                          const rt_light *light, const rt_camera *camera) {
    return rt_multi_scene_upload_meshes(m, spheres, n_spheres, mesh, mesh ? 1 : 0, light, camera);
}

int rt_render_multi(rt_multi *m, const rt_params *p, float *out_rgba_host) {
    if (m && !out_rgba_host) return mfail(m, RT_ERR_INVALID, "output pointer is NULL");
    return multi_render(m, p, nullptr, out_rgba_host);
}

int rt_render_multi_rgb8(rt_multi *m, const rt_params *p, uint8_t *out_rgb8_host) {
    if (m && !out_rgb8_host) return mfail(m, RT_ERR_INVALID, "output pointer is NULL");
    return multi_render(m, p, nullptr, out_rgb8_host, true);
}

int rt_render_multi_device(rt_multi *m, const rt_params *p, void *out_rgba_dev_on_root) {
    if (m && !out_rgba_dev_on_root) return mfail(m, RT_ERR_INVALID, "output pointer is NULL");
    return multi_render(m, p, out_rgba_dev_on_root, nullptr);
}

int rt_multi_get_stats(rt_multi *m, rt_multi_stats *stats) {
    if (!m || !stats) return mfail(m, RT_ERR_INVALID, "bad arguments");
    *stats = m->stats;
    return RT_OK;
}

}  // extern "C"
