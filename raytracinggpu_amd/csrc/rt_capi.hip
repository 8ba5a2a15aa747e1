// rt_capi.hip -- implementation of include/raytrace_hip.h (libraytrace_hip.so).
// Host side of the gfx950 render path: context, scene upload (layout conversion
// from the reference's arrays to the kernel's SoA layout), launches, timing.
#include "../../include/raytrace_hip.h"
#include "rt_kernels.hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
};

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    bool have_scene = false, have_kernel_time = false, have_tonemap_time = false;
    rtk::Scene scene{};
    DevBuf node_lo, node_hi, tri, verts, tidx, scratch_rgba, scratch_rgb8, work;
    rt_stats stats{};
    std::string err;
    char name[256] = {0};
};

namespace {

int fail(rt_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
    return code;
}

#define RT_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail(ctx, RT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int ensure(rt_ctx *ctx, DevBuf &b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return RT_OK;
    b.release();
    RT_HIP(ctx, hipMalloc(&b.p, bytes ? bytes : 16));
    b.bytes = bytes ? bytes : 16;
    return RT_OK;
}

int upload(rt_ctx *ctx, DevBuf &b, const void *src, size_t bytes) {
    int rc = ensure(ctx, b, bytes);
    if (rc != RT_OK) return rc;
    if (bytes) RT_HIP(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

// Host-side Vector arithmetic for the triangle precompute (cpu:227-229).  This TU is
// compiled with -ffp-contract=off, so these are the same single roundings as on the device.
struct h3 { float x, y, z; };
inline h3 hsub(h3 a, h3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline h3 hcross(h3 a, h3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// Converts the reference's bvhTreeToArray layout (optimized.cu:512-534) into traversal order.
// The reference pops the right child first (cpu:291-292 push left then right), so the
// pre-order here descends right before left.
int build_threaded(rt_ctx *ctx, const rt_mesh *m, std::vector<float4> &lo, std::vector<float4> &hi) {
    const int n = m->n_nodes;
    lo.assign(n, make_float4(0, 0, 0, 0));
    hi.assign(n, make_float4(0, 0, 0, 0));
    if (n == 0) return RT_OK;
    struct Item { int ref; int out; int stage; };
    std::vector<char> seen(n, 0);
    std::vector<Item> st;
    int emitted = 0;
    auto node = [&](int i) { return m->bvh_arr10 + (size_t)i * 10; };
    st.push_back({0, -1, 0});
    while (!st.empty()) {
        Item &it = st.back();
        const float *a = node(it.ref);
        if (it.stage == 0) {
            if (seen[it.ref]) return fail(ctx, RT_ERR_INVALID, "bvh_arr10: node %d reached twice (not a tree)", it.ref);
            seen[it.ref] = 1;
            it.out = emitted++;
            const int left = (int)a[0], right = (int)a[1];
            const int ts = (int)a[8], te = (int)a[9];
            if (ts < 0 || te < ts || te > m->n_triangles)
                return fail(ctx, RT_ERR_INVALID, "bvh_arr10: node %d has triangle range [%d,%d) outside [0,%d)", it.ref, ts, te, m->n_triangles);
            lo[it.out] = make_float4(a[2], a[3], a[4], 0);
            hi[it.out] = make_float4(a[5], a[6], a[7], 0);
            if (left == -1 || right == -1) {   // leaf (cpu:287 tests `left` only; the builder sets both or none)
                if (left != -1 || right != -1)
                    return fail(ctx, RT_ERR_INVALID, "bvh_arr10: node %d has exactly one child", it.ref);
                lo[it.out].w = __builtin_bit_cast(float, ts);
                hi[it.out].w = __builtin_bit_cast(float, te);
                st.pop_back();
                continue;
            }
            if (left < 0 || left >= n || right < 0 || right >= n)
                return fail(ctx, RT_ERR_INVALID, "bvh_arr10: node %d has a child index out of range", it.ref);
            it.stage = 1;
            st.push_back({right, -1, 0});
        } else if (it.stage == 1) {
            it.stage = 2;
            const int left = (int)a[0];
            st.push_back({left, -1, 0});
        } else {
            lo[it.out].w = __builtin_bit_cast(float, emitted);   // next node on a box miss: past the subtree
            hi[it.out].w = __builtin_bit_cast(float, -1);
            st.pop_back();
        }
    }
    if (emitted != n) return fail(ctx, RT_ERR_INVALID, "bvh_arr10: %d of %d nodes reachable from the root", emitted, n);
    return RT_OK;
}

int check_params(rt_ctx *ctx, const rt_params *p, int &segs) {
    if (!p) return fail(ctx, RT_ERR_INVALID, "params is NULL");
    if (p->width <= 0 || p->height <= 0) return fail(ctx, RT_ERR_INVALID, "width/height must be positive");
    if ((int64_t)p->width * p->height > (int64_t)1 << 31) return fail(ctx, RT_ERR_INVALID, "image too large");
    if (p->num_rays <= 0) return fail(ctx, RT_ERR_INVALID, "num_rays must be >= 1");
    if (p->num_bounce < 0) return fail(ctx, RT_ERR_INVALID, "num_bounce must be >= 0");
    if (p->depth_convention != 0 && p->depth_convention != 1)
        return fail(ctx, RT_ERR_INVALID, "depth_convention must be 0 (cpu_launcher) or 1 (optimized.cu)");
    segs = p->depth_convention == 0 ? p->num_bounce + 1 : p->num_bounce;
    if (segs > RT_MAX_SEGMENTS) return fail(ctx, RT_ERR_INVALID, "more than %d ray segments", RT_MAX_SEGMENTS);
    if (p->variant < RT_VARIANT_AUTO || p->variant > RT_VARIANT_LDS_ALL) return fail(ctx, RT_ERR_INVALID, "unknown variant %d", p->variant);
    return RT_OK;
}

int launch_render(rt_ctx *ctx, const rt_params *p, const rt_rows *rows, void *out_dev, hipStream_t stream,
                  unsigned long long *work_dev = nullptr) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    int segs = 0;
    int rc = check_params(ctx, p, segs);
    if (rc != RT_OK) return rc;
    if (!rows || !out_dev) return fail(ctx, RT_ERR_INVALID, "rows/out is NULL");
    if (rows->n_rows < 0 || rows->row0 < 0 || rows->tile_rows <= 0 || rows->tile_step <= 0)
        return fail(ctx, RT_ERR_INVALID, "bad row specification");
    if (rows->n_rows > 0) {
        const int64_t last = rows->n_rows - 1;
        const int64_t last_row = rows->row0 + (last / rows->tile_rows) * rows->tile_rows * (int64_t)rows->tile_step + (last % rows->tile_rows);
        if (last_row >= p->height) return fail(ctx, RT_ERR_INVALID, "rows reach image row %lld >= height %d", (long long)last_row, p->height);
    }
    int variant = p->variant == RT_VARIANT_AUTO ? RT_VARIANT_GLOBAL : p->variant;
    if (variant != RT_VARIANT_GLOBAL) return fail(ctx, RT_ERR_UNSUPPORTED, "variant %d is not available in this build", variant);

    RT_HIP(ctx, hipSetDevice(ctx->device));
    rtk::Frame fr{};
    fr.W = p->width; fr.H = p->height; fr.spp = p->num_rays; fr.segs = segs;
    fr.sigma = p->sigma; fr.eps = p->eps; fr.tri_tmin = p->tri_tmin;
    // cpu:694 `-W / (2 * tan(alpha/2))`: g++ folds tan of the constant alpha/2 to the correctly rounded
    // binary32 value; binary64 tan narrowed to binary32 reproduces it (DESIGN.md hazard H12).
    fr.z = -(float)p->width / (2 * (float)std::tan((double)(ctx->scene.fov / 2)));
    fr.seed = p->seed;
    fr.row0 = rows->row0; fr.n_rows = rows->n_rows; fr.tile_rows = rows->tile_rows; fr.tile_step = rows->tile_step;
    fr.out = static_cast<float4 *>(out_dev);
    fr.work = work_dev;

    ctx->stats.pixels = (uint64_t)rows->n_rows * p->width;
    ctx->stats.variant = variant;
    ctx->stats.block_threads = rtk::kBlockThreads;
    if (rows->n_rows == 0) { ctx->stats.grid_blocks = 0; ctx->have_kernel_time = false; return RT_OK; }
    dim3 grid((p->width + rtk::kTileW - 1) / rtk::kTileW, (rows->n_rows + rtk::kTileH - 1) / rtk::kTileH);
    const size_t lds = (size_t)(segs > 0 ? segs : 1) * rtk::kBlockThreads * sizeof(float);
    ctx->stats.lds_bytes = (int)lds;
    ctx->stats.grid_blocks = (int)(grid.x * grid.y);
    RT_HIP(ctx, hipEventRecord(ctx->ev_k0, stream));
    if (work_dev) hipLaunchKernelGGL(rtk::render_kernel<true>, grid, dim3(rtk::kBlockThreads), lds, stream, ctx->scene, fr);
    else hipLaunchKernelGGL(rtk::render_kernel<false>, grid, dim3(rtk::kBlockThreads), lds, stream, ctx->scene, fr);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipEventRecord(ctx->ev_k1, stream));
    ctx->have_kernel_time = true;
    return RT_OK;
}

int launch_tonemap(rt_ctx *ctx, const void *rgba_dev, int64_t npix, void *rgb8_dev, hipStream_t stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (npix < 0 || (npix > 0 && (!rgba_dev || !rgb8_dev))) return fail(ctx, RT_ERR_INVALID, "bad tonemap arguments");
    if (npix == 0) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t quads = (npix + 3) / 4;
    RT_HIP(ctx, hipEventRecord(ctx->ev_t0, stream));
    hipLaunchKernelGGL(rtk::tonemap_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const float4 *>(rgba_dev), npix, static_cast<uint8_t *>(rgb8_dev));
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipEventRecord(ctx->ev_t1, stream));
    ctx->have_tonemap_time = true;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(int *count) {
    if (!count) return fail(nullptr, RT_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(nullptr, RT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return RT_OK;
}

int rt_ctx_create(rt_ctx **out, int device_id) {
    if (!out) return fail(nullptr, RT_ERR_INVALID, "ctx out-pointer is NULL");
    *out = nullptr;
    int n = 0;
    int rc = rt_device_count(&n);
    if (rc != RT_OK) return rc;
    if (n == 0) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device_id < 0 || device_id >= n) return fail(nullptr, RT_ERR_INVALID, "device %d out of range [0,%d)", device_id, n);
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "out of host memory");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_k0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_k1);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_t0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_t1);
    if (e != hipSuccess) {
        int code = fail(nullptr, RT_ERR_HIP, "context creation: %s", hipGetErrorString(e));
        rt_ctx_destroy(ctx);
        return code;
    }
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        int code = fail(nullptr, RT_ERR_NO_DEVICE, "device %d is %s; this library contains gfx950 code only", device_id, prop.gcnArchName);
        rt_ctx_destroy(ctx);
        return code;
    }
    *out = ctx;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx *ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->node_lo.release(); ctx->node_hi.release(); ctx->tri.release(); ctx->verts.release(); ctx->tidx.release();
    ctx->scratch_rgba.release(); ctx->scratch_rgb8.release(); ctx->work.release();
    if (ctx->ev_k0) (void)hipEventDestroy(ctx->ev_k0);
    if (ctx->ev_k1) (void)hipEventDestroy(ctx->ev_k1);
    if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
    if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int rt_device_name(const rt_ctx *ctx, char *buf, size_t buflen) {
    if (!ctx || !buf || buflen == 0) return fail(nullptr, RT_ERR_INVALID, "bad arguments");
    snprintf(buf, buflen, "%s", ctx->name);
    return RT_OK;
}

int rt_scene_upload(rt_ctx *ctx, const rt_sphere *spheres, int n_spheres, const rt_mesh *mesh,
                    const rt_light *light, const rt_camera *camera) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (n_spheres < 0 || (n_spheres > 0 && !spheres)) return fail(ctx, RT_ERR_INVALID, "bad sphere array");
    if (!light || !camera) return fail(ctx, RT_ERR_INVALID, "light/camera is NULL");
    const int n_objects = n_spheres + (mesh ? 1 : 0);
    if (n_spheres > RT_MAX_SPHERES || n_objects > 16)
        return fail(ctx, RT_ERR_INVALID, "at most %d objects (reference: Geometry* objects[10])", 16);
    rtk::Scene sc{};
    for (int i = 0; i < n_spheres; ++i) {
        const rt_sphere &s = spheres[i];
        sc.sph[i] = {s.center[0], s.center[1], s.center[2], s.radius, s.albedo[0], s.albedo[1], s.albedo[2],
                     s.mirror ? 1 : 0, s.in_refraction_index, s.out_refraction_index};
    }
    sc.n_spheres = n_spheres;
    sc.n_objects = n_objects;
    sc.mesh_slot = -1;
    sc.Lx = light->position[0]; sc.Ly = light->position[1]; sc.Lz = light->position[2]; sc.intensity = light->intensity;
    sc.camx = camera->position[0]; sc.camy = camera->position[1]; sc.camz = camera->position[2]; sc.fov = camera->fov;

    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_scene = false;
    std::vector<float4> lo, hi, tri, verts;
    std::vector<int4> tidx;
    if (mesh) {
        if (mesh->object_slot < 0 || mesh->object_slot > n_spheres)
            return fail(ctx, RT_ERR_INVALID, "mesh object_slot %d outside [0,%d]", mesh->object_slot, n_spheres);
        if (mesh->n_vertices < 0 || mesh->n_triangles < 0 || mesh->n_nodes < 0 || mesh->index_stride < 3)
            return fail(ctx, RT_ERR_INVALID, "bad mesh sizes");
        if ((mesh->n_vertices && !mesh->vertices) || (mesh->n_triangles && !mesh->indices) || (mesh->n_nodes && !mesh->bvh_arr10))
            return fail(ctx, RT_ERR_INVALID, "mesh array pointer is NULL");
        if (mesh->n_nodes >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "node indices are stored as floats: < 2^24 nodes");
        sc.mesh_slot = mesh->object_slot;
        sc.mar = mesh->albedo[0]; sc.mag = mesh->albedo[1]; sc.mab = mesh->albedo[2];
        int rc = build_threaded(ctx, mesh, lo, hi);
        if (rc != RT_OK) return rc;
        tri.resize((size_t)mesh->n_triangles * 3);
        tidx.resize(mesh->n_triangles);
        for (int t = 0; t < mesh->n_triangles; ++t) {
            const int32_t *ix = mesh->indices + (size_t)t * mesh->index_stride;
            for (int k = 0; k < 3; ++k)
                if (ix[k] < 0 || ix[k] >= mesh->n_vertices)
                    return fail(ctx, RT_ERR_INVALID, "triangle %d references vertex %d outside [0,%d)", t, ix[k], mesh->n_vertices);
            auto V = [&](int i) { return h3{mesh->vertices[3 * (size_t)i], mesh->vertices[3 * (size_t)i + 1], mesh->vertices[3 * (size_t)i + 2]}; };
            const h3 A = V(ix[0]), B = V(ix[1]), C = V(ix[2]);
            const h3 e1 = hsub(B, A), e2 = hsub(C, A), N = hcross(e1, e2);   // cpu:227-229
            tri[3 * (size_t)t + 0] = make_float4(A.x, A.y, A.z, e1.x);
            tri[3 * (size_t)t + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
            tri[3 * (size_t)t + 2] = make_float4(e2.z, N.x, N.y, N.z);
            tidx[t] = make_int4(ix[0], ix[1], ix[2], 0);
        }
        verts.resize(mesh->n_vertices);
        for (int i = 0; i < mesh->n_vertices; ++i)
            verts[i] = make_float4(mesh->vertices[3 * (size_t)i], mesh->vertices[3 * (size_t)i + 1], mesh->vertices[3 * (size_t)i + 2], 0);
        sc.n_nodes = mesh->n_triangles > 0 ? mesh->n_nodes : 0;
        sc.n_tris = mesh->n_triangles;
        sc.n_verts = mesh->n_vertices;
    }
    int rc;
    if ((rc = upload(ctx, ctx->node_lo, lo.data(), lo.size() * sizeof(float4))) != RT_OK) return rc;
    if ((rc = upload(ctx, ctx->node_hi, hi.data(), hi.size() * sizeof(float4))) != RT_OK) return rc;
    if ((rc = upload(ctx, ctx->tri, tri.data(), tri.size() * sizeof(float4))) != RT_OK) return rc;
    if ((rc = upload(ctx, ctx->verts, verts.data(), verts.size() * sizeof(float4))) != RT_OK) return rc;
    if ((rc = upload(ctx, ctx->tidx, tidx.data(), tidx.size() * sizeof(int4))) != RT_OK) return rc;
    sc.node_lo = static_cast<const float4 *>(ctx->node_lo.p);
    sc.node_hi = static_cast<const float4 *>(ctx->node_hi.p);
    sc.tri = static_cast<const float4 *>(ctx->tri.p);
    sc.verts = static_cast<const float4 *>(ctx->verts.p);
    sc.tidx = static_cast<const int4 *>(ctx->tidx.p);
    ctx->scene = sc;
    ctx->have_scene = true;
    return RT_OK;
}

int rt_render_device(rt_ctx *ctx, const rt_params *p, const rt_rows *rows, void *out_rgba_dev, void *stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    return launch_render(ctx, p, rows, out_rgba_dev, stream ? static_cast<hipStream_t>(stream) : ctx->stream);
}

int rt_render(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, float *out_rgba_host) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!p) return fail(ctx, RT_ERR_INVALID, "params is NULL");
    if (row_begin < 0 || row_end < row_begin || row_end > p->height) return fail(ctx, RT_ERR_INVALID, "bad row range [%d,%d)", row_begin, row_end);
    if (!out_rgba_host) return fail(ctx, RT_ERR_INVALID, "output pointer is NULL");
    const int n = row_end - row_begin;
    const size_t bytes = (size_t)n * (p->width > 0 ? p->width : 0) * sizeof(float4);
    int rc = ensure(ctx, ctx->scratch_rgba, bytes);
    if (rc != RT_OK) return rc;
    rt_rows rows{row_begin, n, n > 0 ? n : 1, 1};
    rc = launch_render(ctx, p, &rows, ctx->scratch_rgba.p, ctx->stream);
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipMemcpyAsync(out_rgba_host, ctx->scratch_rgba.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_tonemap_device(rt_ctx *ctx, const void *rgba_dev, int64_t n_pixels, void *rgb8_dev, void *stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    return launch_tonemap(ctx, rgba_dev, n_pixels, rgb8_dev, stream ? static_cast<hipStream_t>(stream) : ctx->stream);
}

int rt_render_rgb8(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, uint8_t *out_rgb8_host) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!p) return fail(ctx, RT_ERR_INVALID, "params is NULL");
    if (row_begin < 0 || row_end < row_begin || row_end > p->height) return fail(ctx, RT_ERR_INVALID, "bad row range [%d,%d)", row_begin, row_end);
    if (!out_rgb8_host) return fail(ctx, RT_ERR_INVALID, "output pointer is NULL");
    const int n = row_end - row_begin;
    const int64_t npix = (int64_t)n * (p->width > 0 ? p->width : 0);
    int rc = ensure(ctx, ctx->scratch_rgba, (size_t)npix * sizeof(float4));
    if (rc != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->scratch_rgb8, (size_t)npix * 3 + 16)) != RT_OK) return rc;
    rt_rows rows{row_begin, n, n > 0 ? n : 1, 1};
    if ((rc = launch_render(ctx, p, &rows, ctx->scratch_rgba.p, ctx->stream)) != RT_OK) return rc;
    if ((rc = launch_tonemap(ctx, ctx->scratch_rgba.p, npix, ctx->scratch_rgb8.p, ctx->stream)) != RT_OK) return rc;
    RT_HIP(ctx, hipMemcpyAsync(out_rgb8_host, ctx->scratch_rgb8.p, (size_t)npix * 3, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_count_work(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, rt_work *out) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!p || !out) return fail(ctx, RT_ERR_INVALID, "params/out is NULL");
    if (row_begin < 0 || row_end < row_begin || row_end > p->height) return fail(ctx, RT_ERR_INVALID, "bad row range [%d,%d)", row_begin, row_end);
    const int n = row_end - row_begin;
    int rc = ensure(ctx, ctx->scratch_rgba, (size_t)n * (p->width > 0 ? p->width : 0) * sizeof(float4));
    if (rc != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->work, 4 * sizeof(unsigned long long))) != RT_OK) return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemsetAsync(ctx->work.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    rt_rows rows{row_begin, n, n > 0 ? n : 1, 1};
    rc = launch_render(ctx, p, &rows, ctx->scratch_rgba.p, ctx->stream, static_cast<unsigned long long *>(ctx->work.p));
    if (rc != RT_OK) return rc;
    unsigned long long h[4];
    RT_HIP(ctx, hipMemcpyAsync(h, ctx->work.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->rays = h[0]; out->box_tests = h[1]; out->nodes = h[2]; out->tri_tests = h[3];
    return RT_OK;
}

int rt_synchronize(rt_ctx *ctx) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_get_stats(rt_ctx *ctx, rt_stats *stats) {
    if (!ctx || !stats) return fail(ctx, RT_ERR_INVALID, "bad arguments");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->stats.kernel_ms = 0.f;
    ctx->stats.tonemap_ms = 0.f;
    if (ctx->have_kernel_time) {
        RT_HIP(ctx, hipEventSynchronize(ctx->ev_k1));
        RT_HIP(ctx, hipEventElapsedTime(&ctx->stats.kernel_ms, ctx->ev_k0, ctx->ev_k1));
    }
    if (ctx->have_tonemap_time) {
        RT_HIP(ctx, hipEventSynchronize(ctx->ev_t1));
        RT_HIP(ctx, hipEventElapsedTime(&ctx->stats.tonemap_ms, ctx->ev_t0, ctx->ev_t1));
    }
    *stats = ctx->stats;
    return RT_OK;
}

}  // extern "C"
