// rt_capi.hip -- implementation of include/raytrace_hip.h (libraytrace_hip.so).
// Host side of the gfx950 render path: context, scene upload (layout conversion
// from the reference's arrays to the kernel's SoA layout), launches, timing.
#include "../../include/raytrace_hip.h"
#include "rt_kernels.hip.h"
#include "rt_persistent.hip.h"
#include "rt_wavefront.hip.h"
#include "rt_travq.hip.h"
#include "rt_path.hip.h"
#include "rt_meshops.hip.h"
#include "rt_qnodes.hip.h"
#include "rt_bvhbuild.hip.h"
#include "rt_lbvh.hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "rt_host_ctx.hip.h"      // rt_ctx, knobs, buffers, errors
#include "rt_host_scene.hip.h"    // scene upload: layouts, fixed-point nodes, forests
#include "rt_host_render.hip.h"   // frames as launches

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
    PhaseClock pc;
    int rc = rt_device_count(&n);
    pc.lap("hipGetDeviceCount (runtime initialisation)");
    if (rc != RT_OK) return rc;
    if (n == 0) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device_id < 0 || device_id >= n) return fail(nullptr, RT_ERR_INVALID, "device %d out of range [0,%d)", device_id, n);
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "out of host memory");
    ctx->device = device_id;
    ctx->knobs = read_knobs();
    { const char *e = getenv("RT_LBVH_HOST_INSTALL"); ctx->lbvh_host_install = (e && *e && atoi(e) != 0) ? 1 : 0; }
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device_id);
    pc.lap("hipSetDevice");
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device_id);
    pc.lap("hipGetDeviceProperties");
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_k0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_k1);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_t0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_t1);
    for (hipEvent_t &ev : ctx->ev_trav) if (e == hipSuccess) e = hipEventCreate(&ev);
    for (hipEvent_t &ev : ctx->ev_adv) if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming);
    for (int k = 0; k < rt_ctx::kSlots; ++k) if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->slot_half[k], hipEventDisableTiming);
    for (int k = 0; k < rt_ctx::kSlots; ++k) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->slot_rendered[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->slot_done[k], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        int code = fail(nullptr, RT_ERR_HIP, "context creation: %s", hipGetErrorString(e));
        rt_ctx_destroy(ctx);
        return code;
    }
    pc.lap("events");
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    ctx->n_cus = prop.multiProcessorCount;
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
    if (ctx->stream_) (void)hipStreamSynchronize(ctx->stream_);
    for (hipStream_t q : ctx->part_stream) if (q) (void)hipStreamSynchronize(q);   // sub-frame chains of frames issued on a caller's stream
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->copy_stream2) { (void)hipStreamSynchronize(ctx->copy_stream2); (void)hipStreamDestroy(ctx->copy_stream2); }
    for (int k = 0; k < rt_ctx::kSlots; ++k) if (ctx->slot_half[k]) (void)hipEventDestroy(ctx->slot_half[k]);
    for (int k = 0; k < rt_ctx::kSlots; ++k) {
        ctx->slot_rgba[k].release(); ctx->slot_rgb8[k].release();
        if (ctx->slot_rendered[k]) (void)hipEventDestroy(ctx->slot_rendered[k]);
        if (ctx->slot_done[k]) (void)hipEventDestroy(ctx->slot_done[k]);
    }
    ctx->node_lo.release(); ctx->node_hi.release(); ctx->nodes2.release(); ctx->nodesq.release(); ctx->nodesb.release(); ctx->nodesh.release(); ctx->tri2leaf.release(); ctx->nodesw.release(); ctx->leaflh.release(); ctx->qdp_parent.release(); ctx->qdp_cnt.release(); ctx->qdp_g.release(); ctx->qdp_ch.release(); ctx->q2thr.release(); ctx->left_dev.release(); ctx->lvl_nodes.release(); ctx->lvl_off.release(); ctx->nrm.release(); ctx->tri.release(); ctx->verts.release(); ctx->tidx.release();
    ctx->scratch_rgba.release(); ctx->scratch_rgb8.release(); ctx->work.release(); ctx->queue.release();
    ctx->wfM.release(); ctx->wfT.release(); ctx->wfLS.release(); ctx->wfSID.release(); ctx->wfSamp.release();
    ctx->wfQR.release(); ctx->accum.release(); ctx->dbgbuf.release(); ctx->batch_dev.release();
    ctx->pathSamp.release(); ctx->pathT.release(); ctx->tidx_up.release();
    for (DevBuf *b : {&ctx->bb_idx, &ctx->bb_cnt, &ctx->bb_pa, &ctx->bb_pb, &ctx->bb_tmp, &ctx->bb_nodes_i, &ctx->bb_nodes_f, &ctx->bb_counter, &ctx->bb_lvl, &ctx->bb_size, &ctx->bb_pre, &ctx->bb_arr, &ctx->lb_pool, &ctx->lb_pool2, &ctx->perm_dev}) b->release();
    for (hipEvent_t &e : ctx->ev_trav) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : ctx->ev_adv) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t &e : ctx->part_ev) if (e) (void)hipEventDestroy(e);
    for (hipStream_t &q : ctx->part_stream) if (q) (void)hipStreamDestroy(q);
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    for (hipEvent_t &e : ctx->pipe.fork2) if (e) (void)hipEventDestroy(e);
    if (ctx->ev_k0) (void)hipEventDestroy(ctx->ev_k0);
    if (ctx->ev_k1) (void)hipEventDestroy(ctx->ev_k1);
    if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
    if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
    if (ctx->stream_) (void)hipStreamDestroy(ctx->stream_);
    delete ctx;
    return RT_OK;
}

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int rt_device_name(const rt_ctx *ctx, char *buf, size_t buflen) {
    if (!ctx || !buf || buflen == 0) return fail(nullptr, RT_ERR_INVALID, "bad arguments");
    snprintf(buf, buflen, "%s", ctx->name);
    return RT_OK;
}

int rt_scene_upload_meshes(rt_ctx *ctx, const rt_sphere *spheres, int n_spheres, const rt_mesh *meshes, int n_meshes,
                           const rt_light *light, const rt_camera *camera) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (n_spheres < 0 || (n_spheres > 0 && !spheres)) return fail(ctx, RT_ERR_INVALID, "bad sphere array");
    if (n_meshes < 0 || (n_meshes > 0 && !meshes)) return fail(ctx, RT_ERR_INVALID, "bad mesh array");
    if (!light || !camera) return fail(ctx, RT_ERR_INVALID, "light/camera is NULL");
    const int n_objects = n_spheres + n_meshes;
    if (n_spheres > RT_MAX_SPHERES || n_objects > RT_MAX_OBJECTS)
        return fail(ctx, RT_ERR_INVALID, "at most %d objects (reference: Geometry* objects[10])", RT_MAX_OBJECTS);
    rtk::Scene sc{};
    // the meshes in object order; the spheres fill, in array order, the positions the meshes leave free (Scene::addObject numbers the objects as they come, cpu:539-542)
    std::vector<int> order(n_meshes);
    for (int k = 0; k < n_meshes; ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return meshes[a].object_slot < meshes[b].object_slot; });
    bool taken[RT_MAX_OBJECTS] = {};
    for (int k = 0; k < n_meshes; ++k) {
        const rt_mesh &m = meshes[order[k]];
        if (m.object_slot < 0 || m.object_slot >= n_objects) return fail(ctx, RT_ERR_INVALID, "mesh object_slot %d outside [0,%d]", m.object_slot, n_objects - 1);
        if (taken[m.object_slot]) return fail(ctx, RT_ERR_INVALID, "two meshes at object_slot %d", m.object_slot);
        taken[m.object_slot] = true;
        sc.mesh[k] = rtk::MeshRec{0, m.object_slot};
        sc.obj_a[m.object_slot] = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, (int)(m.mirror ? 1 : 0)));
        sc.obj_b[m.object_slot] = make_float4(m.albedo[0], m.albedo[1], m.albedo[2], 0.f);
        // Geometry() (cpu:110) gives a mesh the indices 1 / 1; a zero-initialised rt_mesh (ABI 5 callers) says 0 / 0: the same diffuse object, stored as 1 / 1
        const bool unset = m.in_refraction_index == 0.f && m.out_refraction_index == 0.f;
        sc.obj_n[m.object_slot] = make_float2(unset ? 1.f : m.in_refraction_index, unset ? 1.f : m.out_refraction_index);
    }
    sc.n_meshes = n_meshes;
    int pos = 0;
    for (int i = 0; i < n_spheres; ++i) {
        while (pos < n_objects && taken[pos]) ++pos;
        const rt_sphere &s = spheres[i];
        sc.sph[i] = {s.center[0], s.center[1], s.center[2], s.radius, s.radius * s.radius, pos};   // R * R: one binary32 product (-ffp-contract=off), as cpu:513
        sc.obj_a[pos] = make_float4(s.center[0], s.center[1], s.center[2], __builtin_bit_cast(float, (int)(s.mirror ? 1 : 0)));
        sc.obj_b[pos] = make_float4(s.albedo[0], s.albedo[1], s.albedo[2], 0.f);
        sc.obj_n[pos] = make_float2(s.in_refraction_index, s.out_refraction_index);
        ++pos;
    }
    sc.n_spheres = n_spheres;
    sc.n_objects = n_objects;
    sc.mesh_slot = n_meshes > 0 ? sc.mesh[0].obj : -1;
    sc.Lx = light->position[0]; sc.Ly = light->position[1]; sc.Lz = light->position[2]; sc.intensity = light->intensity;
    sc.camx = camera->position[0]; sc.camy = camera->position[1]; sc.camz = camera->position[2]; sc.fov = camera->fov;

    PhaseClock pc;
    std::vector<int> real;                                               // meshes with something to traverse, in object order
    for (int k = 0; k < n_meshes; ++k) if (meshes[order[k]].n_triangles > 0 && meshes[order[k]].n_nodes > 0) real.push_back(order[k]);
    int rc;
    if (real.size() <= 1) {
        // the reference's own scenes: one mesh (or none; a mesh without triangles is an object that is never hit, cpu:322-325)
        const rt_mesh *one = real.empty() ? (n_meshes > 0 ? &meshes[order[0]] : nullptr) : &meshes[real[0]];
        rc = install_scene(ctx, sc, one);
    } else {
        Forest f;
        if ((rc = build_forest(ctx, meshes, real, f)) != RT_OK) return rc;
        // table entry of every mesh -> first triangle in the forest's index array (a mesh without triangles: the next real mesh's)
        std::vector<int> offs(n_meshes + 1, f.tri_off[real.size()]);
        for (int k = n_meshes - 1, r = (int)real.size() - 1; k >= 0; --k) {
            if (r >= 0 && order[k] == real[r]) { offs[k] = f.tri_off[r]; --r; }
            else offs[k] = offs[k + 1];
        }
        rc = install_scene(ctx, sc, &f.m, &offs);
    }
    if (rc == RT_OK) { ctx->n_real_meshes = (int)real.size(); ctx->real_obj = real.empty() ? -1 : meshes[real[0]].object_slot; }
    pc.lap("rt_scene_upload (layouts, hipMalloc, copies)");
    return rc;
}

int rt_scene_upload(rt_ctx *ctx, const rt_sphere *spheres, int n_spheres, const rt_mesh *mesh,
                    const rt_light *light, const rt_camera *camera) {
    return rt_scene_upload_meshes(ctx, spheres, n_spheres, mesh, mesh ? 1 : 0, light, camera);
}

int rt_render_device(rt_ctx *ctx, const rt_params *p, const rt_rows *rows, void *out_rgba_dev, void *stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    const hipStream_t q_ = stream ? static_cast<hipStream_t>(stream) : own_stream(ctx);
    if (!q_) return fail(ctx, RT_ERR_HIP, "the context's stream: %s", ctx->err.c_str());
    return launch_render(ctx, p, rows, out_rgba_dev, q_);
}

int rt_render_device_batch(rt_ctx *ctx, const rt_params *p, const rt_rows *rows, const rt_frame_desc *frames, int n_frames, void *stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    const hipStream_t q_ = stream ? static_cast<hipStream_t>(stream) : own_stream(ctx);
    if (!q_) return fail(ctx, RT_ERR_HIP, "the context's stream: %s", ctx->err.c_str());
    if (!p || !rows || !frames) return fail(ctx, RT_ERR_INVALID, "params / rows / frames is NULL");
    if (n_frames < 1 || n_frames > RT_MAX_BATCH) return fail(ctx, RT_ERR_INVALID, "n_frames %d outside [1,%d]", n_frames, RT_MAX_BATCH);
    if (p->num_rays != 1) return fail(ctx, RT_ERR_UNSUPPORTED, "a batch renders one sample per pixel and frame (num_rays = %d)", p->num_rays);
    if (p->width <= 0 || p->height <= 0) return fail(ctx, RT_ERR_INVALID, "width/height must be positive");
    rtk::Batch bt{};
    bt.n = n_frames;
    const uint8_t *lo = nullptr, *hi = nullptr;
    const size_t bytes = (size_t)std::max(rows->n_rows, 0) * p->width * sizeof(float4);
    for (int k = 0; k < n_frames; ++k) {
        const rt_frame_desc &f = frames[k];
        if (!f.out_rgba_dev) return fail(ctx, RT_ERR_INVALID, "frame %d: output pointer is NULL", k);
        const uint8_t *o = static_cast<const uint8_t *>(f.out_rgba_dev);
        for (int j = 0; j < k; ++j) {
            const uint8_t *oj = static_cast<const uint8_t *>(frames[j].out_rgba_dev);
            if (o < oj + bytes && oj < o + bytes) return fail(ctx, RT_ERR_INVALID, "frames %d and %d render into overlapping buffers", j, k);
        }
        lo = (!lo || o < lo) ? o : lo; hi = (!hi || o + bytes > hi) ? o + bytes : hi;
        // cpu:694 `-W / (2 * tan(alpha/2))` for this frame's camera (evaluated as launch_render_chunk evaluates the uploaded camera's)
        bt.f[k] = rtk::BatchFrame{f.camera.position[0], f.camera.position[1], f.camera.position[2],
                                  -(float)p->width / (2 * (float)std::tan((double)(f.camera.fov / 2))), f.seed, 0, static_cast<float4 *>(f.out_rgba_dev)};
    }
    // one chunk: the batch exists for SMALL shares (a share too big for one chunk fills the chip by itself: render its frames one by one)
    rt_ctx::Pipe &pl = ctx->pipe;
    pl.prev_valid = pl.valid; pl.valid = false;
    pl.call_chunk = 0; pl.call_chunks = 1; pl.open_parts = 0;
    struct ClearBetween { rt_ctx::Pipe &p; ~ClearBetween() { p.between.clear(); p.between_overflow = false; } } clear_between{pl};
    pl.call_lo = lo; pl.call_hi = hi;                                // (the frames' buffers and whatever lies between them: conservative for the pipelining rule)
    return launch_render_chunk(ctx, p, rows, frames[0].out_rgba_dev, q_, nullptr, nullptr, true, true, &bt);
}

int rt_render(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, float *out_rgba_host) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!p) return fail(ctx, RT_ERR_INVALID, "params is NULL");
    if (row_begin < 0 || row_end < row_begin || row_end > p->height) return fail(ctx, RT_ERR_INVALID, "bad row range [%d,%d)", row_begin, row_end);
    if (!out_rgba_host) return fail(ctx, RT_ERR_INVALID, "output pointer is NULL");
    const int n = row_end - row_begin;
    const size_t bytes = (size_t)n * (p->width > 0 ? p->width : 0) * sizeof(float4);
    int rc = ensure(ctx, ctx->scratch_rgba, bytes);
    if (rc != RT_OK) return rc;
    rt_rows rows{row_begin, n, n > 0 ? n : 1, 1};
    rc = launch_render(ctx, p, &rows, ctx->scratch_rgba.p, own_stream(ctx));
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipMemcpyAsync(out_rgba_host, ctx->scratch_rgba.p, bytes, hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    return RT_OK;
}

int rt_ctx_set_pipelining(rt_ctx *ctx, int on) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    ctx->pipe.on = on != 0;
    ctx->pipe.valid = false;
    return RT_OK;
}

int rt_stats_enable(rt_ctx *ctx, int on) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    ctx->stats_on = on != 0;
    return RT_OK;
}

int rt_render_async(rt_ctx *ctx, const rt_params *p, int slot, void *out_host, int rgb8) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!p) return fail(ctx, RT_ERR_INVALID, "params is NULL");
    if (slot < 0 || slot >= rt_ctx::kSlots) return fail(ctx, RT_ERR_INVALID, "slot %d outside [0,%d)", slot, rt_ctx::kSlots);
    if (!out_host) return fail(ctx, RT_ERR_INVALID, "output pointer is NULL");
    if (p->width <= 0 || p->height <= 0) return fail(ctx, RT_ERR_INVALID, "width/height must be positive");
    const int64_t npix = (int64_t)p->width * p->height;
    int rc = ensure(ctx, ctx->slot_rgba[slot], (size_t)npix * sizeof(float4));
    if (rc != RT_OK) return rc;
    if (rgb8 && (rc = ensure(ctx, ctx->slot_rgb8[slot], (size_t)npix * 3 + 16)) != RT_OK) return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    // the slot's previous frame may still be on its way to the host: the kernels that overwrite its device buffer wait for that copy
    if (ctx->slot_pending[slot]) RT_HIP(ctx, hipStreamWaitEvent(own_stream(ctx), ctx->slot_done[slot], 0));
    rt_rows rows{0, p->height, p->height, 1};
    // frames alternate between the two slots: with the sub-frames' chains on their own streams frame k+1 follows frame k chain by chain
    // (launch_render_chunk); what its kernels must not overtake -- the slot's previous copy -- is handed to the chains directly
    const bool pipe_was = ctx->pipe.on;
    ctx->pipe.on = ctx->knobs.async_pipeline != 0;
    ctx->pipe.extra_wait = ctx->slot_pending[slot] ? ctx->slot_done[slot] : nullptr;
    rc = launch_render(ctx, p, &rows, ctx->slot_rgba[slot].p, own_stream(ctx));
    ctx->pipe.on = pipe_was;
    ctx->pipe.extra_wait = nullptr;
    if (rc != RT_OK) return rc;
    if (rgb8 && (rc = launch_tonemap(ctx, ctx->slot_rgba[slot].p, npix, ctx->slot_rgb8[slot].p, own_stream(ctx))) != RT_OK) return rc;
    RT_HIP(ctx, hipEventRecord(ctx->slot_rendered[slot], own_stream(ctx)));
    if ((rc = need_copy_streams(ctx, ctx->knobs.copy_split != 0)) != RT_OK) return rc;
    // the copy runs on its own stream: the next frame's kernels (other slot) do not queue behind it
    RT_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->slot_rendered[slot], 0));
    const size_t bytes = rgb8 ? (size_t)npix * 3 : (size_t)npix * sizeof(float4);
    const uint8_t *src = static_cast<const uint8_t *>(rgb8 ? ctx->slot_rgb8[slot].p : ctx->slot_rgba[slot].p);
    const size_t half = ctx->knobs.copy_split && bytes >= (4u << 20) ? (bytes / 2 + 4095) / 4096 * 4096 : bytes;   // big frames: two halves on two copy streams
    RT_HIP(ctx, hipMemcpyAsync(out_host, src, half, hipMemcpyDeviceToHost, ctx->copy_stream));
    if (half < bytes) {
        RT_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream2, ctx->slot_rendered[slot], 0));
        RT_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(out_host) + half, src + half, bytes - half, hipMemcpyDeviceToHost, ctx->copy_stream2));
        RT_HIP(ctx, hipEventRecord(ctx->slot_half[slot], ctx->copy_stream2));
        RT_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->slot_half[slot], 0));
    }
    RT_HIP(ctx, hipEventRecord(ctx->slot_done[slot], ctx->copy_stream));
    ctx->slot_pending[slot] = true;
    return RT_OK;
}

int rt_wait(rt_ctx *ctx, int slot) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (slot < 0 || slot >= rt_ctx::kSlots) return fail(ctx, RT_ERR_INVALID, "slot %d outside [0,%d)", slot, rt_ctx::kSlots);
    if (!ctx->slot_pending[slot]) return fail(ctx, RT_ERR_INVALID, "slot %d has no frame in flight", slot);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipEventSynchronize(ctx->slot_done[slot]));
    ctx->slot_pending[slot] = false;
    return RT_OK;
}

int rt_tonemap_device(rt_ctx *ctx, const void *rgba_dev, int64_t n_pixels, void *rgb8_dev, void *stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    const hipStream_t q_ = stream ? static_cast<hipStream_t>(stream) : own_stream(ctx);
    if (!q_) return fail(ctx, RT_ERR_HIP, "the context's stream: %s", ctx->err.c_str());
    return launch_tonemap(ctx, rgba_dev, n_pixels, rgb8_dev, q_);
}

int rt_render_rgb8(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, uint8_t *out_rgb8_host) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!p) return fail(ctx, RT_ERR_INVALID, "params is NULL");
    if (row_begin < 0 || row_end < row_begin || row_end > p->height) return fail(ctx, RT_ERR_INVALID, "bad row range [%d,%d)", row_begin, row_end);
    if (!out_rgb8_host) return fail(ctx, RT_ERR_INVALID, "output pointer is NULL");
    const int n = row_end - row_begin;
    const int64_t npix = (int64_t)n * (p->width > 0 ? p->width : 0);
    PhaseClock pc;
    int rc = ensure(ctx, ctx->scratch_rgba, (size_t)npix * sizeof(float4));
    if (rc != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->scratch_rgb8, (size_t)npix * 3 + 16)) != RT_OK) return rc;
    pc.lap("rt_render_rgb8: frame buffers (hipMalloc)");
    rt_rows rows{row_begin, n, n > 0 ? n : 1, 1};
    if ((rc = launch_render(ctx, p, &rows, ctx->scratch_rgba.p, own_stream(ctx))) != RT_OK) return rc;
    pc.lap("rt_render_rgb8: enqueue (path state, code object)");
    if ((rc = launch_tonemap(ctx, ctx->scratch_rgba.p, npix, ctx->scratch_rgb8.p, own_stream(ctx))) != RT_OK) return rc;
    if (pc.on) { RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx))); pc.lap("rt_render_rgb8: kernels (wait)"); }
    RT_HIP(ctx, hipMemcpyAsync(out_rgb8_host, ctx->scratch_rgb8.p, (size_t)npix * 3, hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    pc.lap("rt_render_rgb8: copy to the host");
    return RT_OK;
}

int rt_count_work(rt_ctx *ctx, const rt_params *p, int row_begin, int row_end, rt_work *out) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!p || !out) return fail(ctx, RT_ERR_INVALID, "params/out is NULL");
    if (row_begin < 0 || row_end < row_begin || row_end > p->height) return fail(ctx, RT_ERR_INVALID, "bad row range [%d,%d)", row_begin, row_end);
    const int n = row_end - row_begin;
    int rc = ensure(ctx, ctx->scratch_rgba, (size_t)n * (p->width > 0 ? p->width : 0) * sizeof(float4));
    if (rc != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->work, 24 * sizeof(unsigned long long))) != RT_OK) return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemsetAsync(ctx->work.p, 0, 24 * sizeof(unsigned long long), own_stream(ctx)));
    rt_rows rows{row_begin, n, n > 0 ? n : 1, 1};
    rc = launch_render(ctx, p, &rows, ctx->scratch_rgba.p, own_stream(ctx), static_cast<unsigned long long *>(ctx->work.p));
    if (rc != RT_OK) return rc;
    unsigned long long h[24];
    RT_HIP(ctx, hipMemcpyAsync(h, ctx->work.p, sizeof(h), hipMemcpyDeviceToHost, own_stream(ctx)));
    std::vector<float> fb((size_t)n * (size_t)p->width * 4);
    RT_HIP(ctx, hipMemcpyAsync(fb.data(), ctx->scratch_rgba.p, fb.size() * sizeof(float), hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    double rays = 0;                                  // .w of every pixel = rays traced for it (exact in binary32)
    for (size_t k = 3; k < fb.size(); k += 4) rays += fb[k];
    out->rays = (uint64_t)rays; out->box_tests = h[1]; out->nodes = h[2]; out->tri_tests = h[3];
    out->box_literal = h[5]; out->tri_literal = h[6];
    for (int k = 0; k < 12; ++k) out->steps[k] = h[8 + k];
    // the counting instantiation checks every index that reaches an address (rt_travq.hip.h WQ_CHECK, rt_path.hip.h)
    if (h[4] != 0) return fail(ctx, RT_ERR_INTERNAL, "traversal invariant violated (mask 0x%llx: 1 path, 2 triangle, 4 node, 8 stack, 16 leaf queue, 32 staging)", h[4]);
    return RT_OK;
}

#include "rt_host_mesh.hip.h"     // rt_mesh_set_normals / transform / rebuild (reference tree, LBVH)

int rt_camera_basis(const rt_camera_pose *pose, float bx[3], float by[3], float bz[3]) {
    if (!pose || !bx || !by || !bz) return fail(nullptr, RT_ERR_INVALID, "bad arguments");
    camera_basis(pose->yaw, pose->pitch, bx, by, bz);
    return RT_OK;
}

int rt_render_pose(rt_ctx *ctx, const rt_params *p, const rt_camera_pose *pose, float *out_rgba_host) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!p || !pose || !out_rgba_host) return fail(ctx, RT_ERR_INVALID, "params/pose/out is NULL");
    const size_t bytes = (size_t)(p->height > 0 ? p->height : 0) * (p->width > 0 ? p->width : 0) * sizeof(float4);
    int rc = ensure(ctx, ctx->scratch_rgba, bytes);
    if (rc != RT_OK) return rc;
    rt_rows rows{0, p->height, p->height > 0 ? p->height : 1, 1};
    if ((rc = launch_render(ctx, p, &rows, ctx->scratch_rgba.p, own_stream(ctx), nullptr, pose)) != RT_OK) return rc;
    RT_HIP(ctx, hipMemcpyAsync(out_rgba_host, ctx->scratch_rgba.p, bytes, hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    return RT_OK;
}

int rt_render_pose_device(rt_ctx *ctx, const rt_params *p, const rt_camera_pose *pose, const rt_rows *rows, void *out_rgba_dev, void *stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    const hipStream_t q_ = stream ? static_cast<hipStream_t>(stream) : own_stream(ctx);
    if (!q_) return fail(ctx, RT_ERR_HIP, "the context's stream: %s", ctx->err.c_str());
    if (!pose) return fail(ctx, RT_ERR_INVALID, "pose is NULL");
    return launch_render(ctx, p, rows, out_rgba_dev, q_, nullptr, pose);
}

int rt_progressive_reset(rt_ctx *ctx) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    ctx->prog_frames = 0;                                             // buffer_reset, realtime:1246-1251
    return RT_OK;
}

int rt_progressive_frames(const rt_ctx *ctx, int *frames) {
    if (!ctx || !frames) return fail(nullptr, RT_ERR_INVALID, "bad arguments");
    *frames = ctx->prog_frames;
    return RT_OK;
}

int rt_progressive_frame(rt_ctx *ctx, const rt_params *p, const rt_camera_pose *pose, float *display_rgba_host, uint8_t *rgb8_host) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!p || !pose) return fail(ctx, RT_ERR_INVALID, "params/pose is NULL");
    if (p->width <= 0 || p->height <= 0) return fail(ctx, RT_ERR_INVALID, "width/height must be positive");
    const int64_t npix = (int64_t)p->width * p->height;
    const size_t bytes = (size_t)npix * sizeof(float4);
    int rc;
    if ((rc = ensure(ctx, ctx->scratch_rgba, bytes)) != RT_OK || (rc = ensure(ctx, ctx->accum, 2 * bytes)) != RT_OK ||
        (rc = ensure(ctx, ctx->scratch_rgb8, (size_t)npix * 3 + 16)) != RT_OK)
        return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->prog_frames == 0 || ctx->prog_w != p->width || ctx->prog_h != p->height) {   // realtime:1246-1251 (a new size also resets)
        RT_HIP(ctx, hipMemsetAsync(ctx->accum.p, 0, bytes, own_stream(ctx)));
        ctx->prog_frames = 0; ctx->prog_w = p->width; ctx->prog_h = p->height;
    }
    const int frame_no = ctx->prog_frames + 1;                        // frames++, realtime:1253
    rt_params q = *p;
    uint32_t a = (uint32_t)frame_no;                                  // WangHash(frames), realtime:1190-1197, seeds the frame's RNG
    a = (a ^ 61u) ^ (a >> 16); a = a + (a << 3); a = a ^ (a >> 4); a = a * 0x27d4eb2du; a = a ^ (a >> 15);
    q.seed = a;
    rt_rows rows{0, p->height, p->height, 1};
    if ((rc = launch_render(ctx, &q, &rows, ctx->scratch_rgba.p, own_stream(ctx), nullptr, pose)) != RT_OK) return rc;
    float4 *accum = static_cast<float4 *>(ctx->accum.p), *display = accum + npix;
    hipLaunchKernelGGL(rtk::accumulate_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, own_stream(ctx),
                       static_cast<const float4 *>(ctx->scratch_rgba.p), accum, display, static_cast<uint8_t *>(ctx->scratch_rgb8.p), npix, frame_no);
    RT_HIP(ctx, hipGetLastError());
    ctx->prog_frames = frame_no;
    if (display_rgba_host) RT_HIP(ctx, hipMemcpyAsync(display_rgba_host, display, bytes, hipMemcpyDeviceToHost, own_stream(ctx)));
    if (rgb8_host) RT_HIP(ctx, hipMemcpyAsync(rgb8_host, ctx->scratch_rgb8.p, (size_t)npix * 3, hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    return RT_OK;
}

int rt_host_alloc(void **ptr, size_t bytes) {
    if (!ptr) return fail(nullptr, RT_ERR_INVALID, "ptr is NULL");
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { *ptr = nullptr; return fail(nullptr, RT_ERR_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    return RT_OK;
}

int rt_host_free(void *ptr) {
    if (!ptr) return RT_OK;
    hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) return fail(nullptr, RT_ERR_HIP, "hipHostFree: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_ctx_selfcheck(rt_ctx *ctx) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const DevBuf *bufs[] = {&ctx->node_lo, &ctx->node_hi, &ctx->nodes2, &ctx->nodesq, &ctx->nodesb, &ctx->q2thr, &ctx->tri, &ctx->verts, &ctx->tidx, &ctx->tidx_up, &ctx->nrm,
                            &ctx->scratch_rgba, &ctx->scratch_rgb8, &ctx->work, &ctx->queue, &ctx->wfM, &ctx->wfT, &ctx->wfLS, &ctx->wfSID, &ctx->wfSamp,
                            &ctx->wfQR, &ctx->pathSamp, &ctx->pathT, &ctx->accum, &ctx->left_dev, &ctx->lvl_nodes, &ctx->lvl_off, &ctx->bb_idx, &ctx->bb_cnt, &ctx->bb_pa, &ctx->bb_pb, &ctx->bb_tmp,
                            &ctx->bb_nodes_i, &ctx->bb_nodes_f, &ctx->bb_counter, &ctx->bb_lvl, &ctx->bb_size, &ctx->bb_pre, &ctx->bb_arr, &ctx->lb_pool, &ctx->lb_pool2, &ctx->perm_dev,
                            &ctx->slot_rgba[0], &ctx->slot_rgba[1], &ctx->slot_rgb8[0], &ctx->slot_rgb8[1]};
    for (const DevBuf *b : bufs) {
        if (!b->p) continue;
        hipPointerAttribute_t at{};
        RT_HIP(ctx, hipPointerGetAttributes(&at, b->p));
        if (at.device != ctx->device) return fail(ctx, RT_ERR_INTERNAL, "a buffer of the context of device %d lives on device %d", ctx->device, at.device);
    }
    return RT_OK;
}

int rt_device_alloc(rt_ctx *ctx, void **ptr, size_t bytes) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!ptr) return fail(ctx, RT_ERR_INVALID, "ptr is NULL");
    *ptr = nullptr;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(ptr, bytes ? bytes : 16);
    if (e != hipSuccess) { *ptr = nullptr; return fail(ctx, RT_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    return RT_OK;
}

int rt_device_free(void *ptr) {
    if (!ptr) return RT_OK;
    hipError_t e = hipFree(ptr);
    if (e != hipSuccess) return fail(nullptr, RT_ERR_HIP, "hipFree: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_device_to_host(rt_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (bytes && (!dst_host || !src_dev)) return fail(ctx, RT_ERR_INVALID, "bad copy arguments");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) RT_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    return RT_OK;
}

int rt_synchronize(rt_ctx *ctx) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    return RT_OK;
}

int rt_get_stats(rt_ctx *ctx, rt_stats *stats) {
    if (!ctx || !stats) return fail(ctx, RT_ERR_INVALID, "bad arguments");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->stats.kernel_ms = 0.f;
    ctx->stats.tonemap_ms = 0.f;
    ctx->stats.trav_ms = 0.f;
    ctx->stats.trav_launches = 0;
    ctx->stats.adv_ms = 0.f; ctx->stats.adv_launches = 0; ctx->stats.adv_paths = 0;
    if (ctx->have_kernel_time) {
        RT_HIP(ctx, hipEventSynchronize(ctx->ev_k1));
        RT_HIP(ctx, hipEventElapsedTime(&ctx->stats.kernel_ms, ctx->ev_k0, ctx->ev_k1));
        for (int k = 0; k < ctx->n_trav_events; ++k) {
            float ms = 0.f;
            RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_trav[2 * k], ctx->ev_trav[2 * k + 1]));
            ctx->stats.trav_ms += ms;
        }
        ctx->stats.trav_launches = ctx->n_trav_events;
        for (int k = 0; k < ctx->n_adv_events; ++k) {
            float ms = 0.f;
            RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_adv[2 * k], ctx->ev_adv[2 * k + 1]));
            ctx->stats.adv_ms += ms;
        }
        ctx->stats.adv_launches = ctx->n_adv_events;
        ctx->stats.adv_paths = ctx->adv_paths;
    }
    if (ctx->have_tonemap_time) {
        RT_HIP(ctx, hipEventSynchronize(ctx->ev_t1));
        RT_HIP(ctx, hipEventElapsedTime(&ctx->stats.tonemap_ms, ctx->ev_t0, ctx->ev_t1));
    }
    *stats = ctx->stats;
    return RT_OK;
}

}  // extern "C"

#include "rt_multi.hip.h"
#include "rt_kat.hip.h"
#include "rt_trace.hip.h"
