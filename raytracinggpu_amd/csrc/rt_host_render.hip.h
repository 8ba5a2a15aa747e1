// rt_host_render.hip.h -- host side, part 2 of 4: one frame (or a batch of frames) as launches -- variant choice, launch geometry of the wavefront pipeline, the two
// sub-frames and their streams, pipelining across calls, chunking of big frames, tone mapping.
#pragma once

namespace {

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
    if (p->variant < RT_VARIANT_AUTO || p->variant > RT_VARIANT_PATH) return fail(ctx, RT_ERR_INVALID, "unknown variant %d", p->variant);
    return RT_OK;
}

// wf_travq instantiations: [STATS][R == 32][LDSN][LDSV]
using TravqFn = void (*)(const rtk::Scene, const rtk::Frame, const rtk::WfState, const int, const int, const int, const int);
template <bool S, int R> TravqFn travq_pick(bool ldsn, bool ldsv) {
    return ldsn ? (ldsv ? rtk::wf_travq<S, R, true, true> : rtk::wf_travq<S, R, true, false>) : (ldsv ? rtk::wf_travq<S, R, false, true> : rtk::wf_travq<S, R, false, false>);
}
TravqFn travq_fn(bool stats, int R, bool ldsn, bool ldsv = false, bool qn = false, bool qw = false) {
    if (qw && R == 64 && !ldsn && !ldsv) return stats ? rtk::wf_travq<true, 64, false, false, true, true> : rtk::wf_travq<false, 64, false, false, true, true>;
    if (qn && !stats && R == 64 && !ldsn && !ldsv) return rtk::wf_travq<false, 64, false, false, true>;
    if (stats) return R == 32 ? travq_pick<true, 32>(ldsn, ldsv) : travq_pick<true, 64>(ldsn, ldsv);
    return R == 32 ? travq_pick<false, 32>(ldsn, ldsv) : travq_pick<false, 64>(ldsn, ldsv);
}
size_t travq_carve_bytes(int R, bool qw = false) {
    if (qw) return (size_t)rtk::QCarve<64, rtk::kQwStackCap, rtk::kQwLeafCap, rtk::kQwTris>::kBytes;
    return R == 64 ? (size_t)rtk::QCarve<64, rtk::QStackCap<64>::value, rtk::QLeafCap<64>::value>::kBytes : (size_t)rtk::QCarve<32, rtk::QStackCap<32>::value, rtk::QLeafCap<32>::value>::kBytes;
}
int travq_stack_cap(int R, bool qw = false) { return qw ? rtk::kQwStackCap : R == 64 ? rtk::QStackCap<64>::value : rtk::QStackCap<32>::value; }
int travq_block_threads(int) { return rtk::kQBlock; }

// Camera::rotate(), realtime_render.cu:823-846 (host code there too: float cos/sin/sqrt)
void camera_basis(float yaw, float pitch, float bx[3], float by[3], float bz[3]) {
    h3 x{1, 0, 0}, y{0, 1, 0}, z{0, 0, -1};
    const float cy = cosf(yaw), sy = sinf(yaw);
    x = h3{x.x * cy + z.x * sy, x.y * cy + z.y * sy, x.z * cy + z.z * sy};
    z = hcross(y, x);
    const float cp = cosf(pitch), sp = sinf(pitch);
    y = h3{y.x * cp - z.x * sp, y.y * cp - z.y * sp, y.z * cp - z.z * sp};
    z = hcross(x, y);
    auto norm = [](h3 v) { const float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); return h3{v.x / n, v.y / n, v.z / n}; };
    x = norm(x); y = norm(y); z = norm(z);
    bx[0] = x.x; bx[1] = x.y; bx[2] = x.z; by[0] = y.x; by[1] = y.y; by[2] = y.z; bz[0] = z.x; bz[1] = z.y; bz[2] = z.z;
}

// Traversal-launch geometry of the wavefront pipeline for st.n_paths paths (2 ray slots each): every workgroup owns an equal,
// spatially scrambled share of the ray slots; its waves draw from it on demand.  Fills st.n_groups, log2S, Q, Q_m, slots_per_block.
void wf_geometry(const Knobs &kn, int n_cus, int bpc, int parts, int wpb, bool oversubscribe, rtk::WfState &st, int64_t &tblocks_out) {
    st.n_groups = 2 * st.n_paths / 4;                         // two ray slots per path (continuation + shadow)
    int64_t tblocks = std::max<int64_t>(1, (int64_t)n_cus * bpc / parts);   // all parts co-resident
    // work-stack kernel: more workgroups than fit at once; the dispatcher hands a finished workgroup's CU share to
    // the next one, which evens out the cost differences between the workgroups' shares of the rays
    const int oversub = kn.oversub;                            // default 2, measured: 1.85 -> 1.67 ms/frame (cat, 1080p)
    if (oversubscribe && oversub > 1) {
        // (measured down to one GPU's share of a 1080p frame split over 8: oversubscribing pays at every size;
        // RT_TRAVQ_OVERSUB_MIN = ray slots per wave below which a launch is not oversubscribed, for experiments)
        const int min_slots = kn.oversub_min;
        const int64_t slots_per_wave = (int64_t)st.n_groups * 4 / (tblocks * oversub * wpb);
        if (slots_per_wave >= min_slots) tblocks *= oversub;
    }
    const int min_groups = kn.min_groups * wpb;               // default 16: >= 64 ray slots per wave on average
    int64_t groups_per_block = (st.n_groups + tblocks - 1) / tblocks;
    if (groups_per_block < min_groups) {                      // small launch: fewer, fuller workgroups
        tblocks = (st.n_groups + min_groups - 1) / min_groups;
        if (tblocks < 1) tblocks = 1;
        groups_per_block = (st.n_groups + tblocks - 1) / tblocks;
    }
    // scramble: consecutive group-slots of one workgroup must land on groups spread over the WHOLE sub-frame, so
    // the stride pattern's period S is the largest power of two not above a workgroup's number of groups
    st.log2S = 0;
    while ((2 << st.log2S) <= groups_per_block && st.log2S < 16) ++st.log2S;
    if (kn.log2S >= 0 && kn.log2S < st.log2S) st.log2S = kn.log2S;   // experiment: less scrambling = more coherent rays per workgroup
    const int S = 1 << st.log2S;
    st.Q = (st.n_groups + S - 1) / S; st.Q_m = rtk::wf_div_magic(st.Q);
    const int64_t total_slots = (int64_t)S * st.Q * 4;
    st.slots_per_block = (int)(((total_slots + tblocks - 1) / tblocks + 3) / 4 * 4);
    tblocks_out = tblocks;
}

// One chunk of rows (launch_render below cuts big frames into cache-sized chunks).  rec_begin / rec_end: this chunk opens / closes the
// call's kernel-time bracket (ev_k0 / ev_k1).
// Streams are created when first needed: a context that renders one frame in two sub-frames owns two streams, not eleven.  The runtime
// maps a process's streams onto a handful of hardware queues (four by default); with three contexts' worth of idle streams in one process
// the two ACTIVE streams of a context could land on the same queue and its sub-frames ran one after the other (a 1/8 share of
// 7680x4320 took 2.7 instead of 2.0 ms next to two other contexts).
int need_part_streams(rt_ctx *ctx, int parts, bool chain0 = false) {
    for (int j = chain0 ? 0 : 1; j < parts && j < rt_ctx::kMaxParts; ++j) {
        if (!ctx->part_stream[j]) {
            // the second sub-frame's stream in the HIGH-priority class: the runtime keeps a separate pool of hardware queues per priority, so
            // this stream can never be mapped onto the queue of the caller's (normal-priority) stream, whatever else the process has created
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            const int pr = (ctx->knobs.part_prio && (j & 1)) ? hi : 0;       // (never the LOW class for a chain: the classes do prioritise, and a low chain next to a high one runs after it, not beside it)
            RT_HIP(ctx, hipStreamCreateWithPriority(&ctx->part_stream[j], hipStreamNonBlocking, pr));
        }
        if (!ctx->part_ev[j]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->part_ev[j], hipEventDisableTiming));
    }
    return RT_OK;
}
int need_copy_streams(rt_ctx *ctx, bool second) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const int pr = ctx->knobs.copy_prio > 0 ? lo : ctx->knobs.copy_prio < 0 ? hi : 0;   // (fixed; RT_COPY_PRIO was an environment knob until round 5) 1 = the low-priority class' queue pool, -1 = the high one
    if (!ctx->copy_stream) RT_HIP(ctx, hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, pr));
    if (second && !ctx->copy_stream2) RT_HIP(ctx, hipStreamCreateWithPriority(&ctx->copy_stream2, hipStreamNonBlocking, pr));
    return RT_OK;
}

// RT_VARIANT_AUTO for a scene without a mesh: the lock-step kernel (launch_render_chunk says why)
inline bool auto_is_lockstep(const rt_ctx *ctx, const rt_camera_pose *pose) {
    return ctx->knobs.auto_lockstep != 0 && ctx->have_scene && ctx->scene.mesh_slot < 0 && ctx->scene.nrm == nullptr && pose == nullptr;
}

int launch_render_chunk(rt_ctx *ctx, const rt_params *p, const rt_rows *rows, void *out_dev, hipStream_t stream,
                        unsigned long long *work_dev, const rt_camera_pose *pose, bool rec_begin, bool rec_end, const rtk::Batch *batch = nullptr) {
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
    // LDS budget of the node-staging traversal kernel: all nodes + 16 per-wave carves in one 1024-thread workgroup
    const size_t lds_nodes_bytes = (size_t)ctx->scene.n_nodes * 32 + (rtk::kTravBlockLds / 64) * (size_t)rtk::TravCarve<256, 4>::kBytes + 16;
    const bool lds_fits = ctx->scene.n_nodes > 0 && lds_nodes_bytes <= 160 * 1024;
    int variant = p->variant;
    // measured on MI355X (cat, 1080p): the work-stack traversal (1.67 ms/frame) beats the per-lane stackless walk
    // (2.48 ms/frame; with LDS-staged nodes 2.65), so AUTO is the work-stack variant
    // ... when there is a mesh.  A scene of spheres alone has no traversal to feed and no divergence to sort out: one lane per pixel for the whole frame (the reference's
    // own structure, the lock-step kernel) keeps a path in registers instead of streaming it through HBM once per bounce -- BASELINE config 2, 1920x1080 b 3: 0.198 against
    // 0.220 ms per frame (profiles/round5/ab_spheres_only.txt).  A posed camera exists in the wavefront family only.
    if (variant == RT_VARIANT_AUTO) variant = (auto_is_lockstep(ctx, pose) && !batch) ? RT_VARIANT_LOCKSTEP : RT_VARIANT_WAVEFRONT_QUEUE;
    // BASELINE config 4 / north star: "hot triangle vertices and top BVH levels staged in LDS" = the work-stack traversal kernel
    // with the vertex array (LDS_VERTS), the breadth-first top of the node array (LDS_TOP) or both (LDS_ALL) staged per workgroup
    const int variant_req = variant;
    const bool want_ldsv = variant == RT_VARIANT_LDS_VERTS || variant == RT_VARIANT_LDS_ALL;
    const bool want_ldsn = variant == RT_VARIANT_LDS_TOP || variant == RT_VARIANT_LDS_ALL;
    if (want_ldsv || want_ldsn) variant = RT_VARIANT_WAVEFRONT_QUEUE;
    if (variant == RT_VARIANT_WAVEFRONT_LDS && !lds_fits) {
        if (ctx->scene.n_nodes == 0) variant = RT_VARIANT_WAVEFRONT;      // no mesh: nothing to stage
        else return fail(ctx, RT_ERR_UNSUPPORTED, "%d BVH nodes need %zu bytes of LDS (> 160 KiB)", ctx->scene.n_nodes, lds_nodes_bytes);
    }
    if (variant == RT_VARIANT_WAVEFRONT_QUEUE && (ctx->scene.n_nodes + 2 >= (1 << rtk::kQNodeBits) || !ctx->travq_ok)) {   // entry = node << 10 | slot << 4
        if (want_ldsv || want_ldsn) return fail(ctx, RT_ERR_UNSUPPORTED, "%d BVH nodes: the LDS-staged variants need < 2^22 nodes and leaves below 2^21 triangles", ctx->scene.n_nodes);
        variant = RT_VARIANT_WAVEFRONT;
    }
    if (variant == RT_VARIANT_PATH && ctx->scene.n_nodes + 2 >= (1 << rtk::kPNodeBits)) variant = RT_VARIANT_WAVEFRONT;

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
    fr.out_tile0 = 0; fr.out_tile_step = 1;
    rtk::Scene scn = ctx->scene;                                      // per-launch copy: a pose moves the camera
    fr.cam_mode = 0; fr.inv_n = 1.f;
    const bool wf_family = variant == RT_VARIANT_WAVEFRONT || variant == RT_VARIANT_WAVEFRONT_LDS || variant == RT_VARIANT_WAVEFRONT_QUEUE || variant == RT_VARIANT_PATH;
    if (batch && !(wf_family && variant != RT_VARIANT_PATH))
        return fail(ctx, RT_ERR_UNSUPPORTED, "a batch of frames needs a wavefront variant (auto, wavefront, wavefront_lds, wavefront_queue, lds_*)");
    if (scn.nrm != nullptr && !wf_family)
        return fail(ctx, RT_ERR_UNSUPPORTED, "smooth normals need a wavefront or path variant");
    if (pose) {                                                       // realtime_render.cu's camera
        if (!wf_family) return fail(ctx, RT_ERR_UNSUPPORTED, "a camera pose needs a wavefront or path variant");
        fr.cam_mode = 1;
        camera_basis(pose->yaw, pose->pitch, fr.bx, fr.by, fr.bz);
        scn.camx = pose->position[0]; scn.camy = pose->position[1]; scn.camz = pose->position[2];
        fr.z = -(float)p->width / (2 * (float)std::tan((double)(pose->fov / 2)));   // realtime:1112, evaluated as cpu:694 is here
        fr.inv_n = (float)(1. / p->num_rays);                         // realtime:1131
    }

    ctx->stats.pixels = (uint64_t)rows->n_rows * p->width;
    ctx->stats.travq_mode = -1;
    ctx->stats.variant = (want_ldsv || want_ldsn) ? variant_req : variant;
    if (rows->n_rows == 0) { ctx->stats.grid_blocks = 0; ctx->have_kernel_time = false; return RT_OK; }
    const int nseg = segs > 0 ? segs : 1;
    ctx->n_trav_events = 0; ctx->n_adv_events = 0; ctx->adv_paths = 0;
    if (variant == RT_VARIANT_PATH) {
        // ONE persistent launch per sub-frame and sample chunk (rt_path.hip.h): a wave owns 64 paths (one per lane) from camera ray to framebuffer store
        const Knobs &kn = ctx->knobs;
        constexpr int wpb = rtk::kQBlock / 64;
        const size_t lds = (size_t)wpb * rtk::PCarve::bytes(segs) + 16;
        int nb = 0;                                                   // workgroups per CU the registers and this frame's LDS carve allow
        if (work_dev) RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk::wf_path<true>, rtk::kQBlock, lds));
        else RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk::wf_path<false>, rtk::kQBlock, lds));
        if (nb < 1) return fail(ctx, RT_ERR_UNSUPPORTED, "wf_path does not fit a CU with %zu bytes of LDS per workgroup", lds);
        const int bpc = std::min(kn.path_bpc, nb);
        int parts = std::min(kn.path_parts, (int)rt_ctx::kMaxParts);
        int R = rows->tile_rows, G = rows->tile_step;
        if (G == 1) R = 8;                                            // contiguous rows: any tile height describes them
        const int T = (rows->n_rows + R - 1) / R;                     // local tiles of this call
        if (R % 8 != 0 || work_dev) parts = 1;
        if (parts > T) parts = T > 0 ? T : 1;
        const int tiles_x = (p->width + 7) / 8;
        int qcap = rtk::kPStack;
        if (kn.travq_cap >= 128 && kn.travq_cap < qcap) qcap = kn.travq_cap;   // tests: force the serial drain
        struct PPart { rtk::Frame fr; int n_paths; size_t base; };
        std::vector<PPart> pv(parts);
        size_t np_total = 0;
        for (int j = 0; j < parts; ++j) {
            const int Tj = (T - j + parts - 1) / parts;               // local tiles j, j+parts, ...
            int nrows_j = Tj * R;
            if (Tj > 0 && (T - 1) % parts == j) nrows_j -= T * R - rows->n_rows;   // the last local tile may be partial
            pv[j].fr = fr;
            if (parts > 1 || G == 1) {
                pv[j].fr.row0 = rows->row0 + j * R * G; pv[j].fr.n_rows = nrows_j; pv[j].fr.tile_rows = R; pv[j].fr.tile_step = G * parts;
                pv[j].fr.out_tile0 = j; pv[j].fr.out_tile_step = parts;
            }
            const int64_t n_paths64 = (int64_t)tiles_x * ((pv[j].fr.n_rows + 7) / 8) * 64;
            if (n_paths64 >= ((int64_t)1 << 29)) return fail(ctx, RT_ERR_INVALID, "image too large: %lld pixel slots per sub-frame (limit 2^29)", (long long)n_paths64);
            pv[j].n_paths = (int)n_paths64;
            pv[j].base = np_total;
            np_total += (size_t)n_paths64;
        }
        // samples of a pixel are independent paths; with more than one the per-sample colours are summed in sample order afterwards
        int chunk = 1;
        if (fr.spp > 1) {
            const int64_t biggest = std::max<int64_t>(1, (int64_t)(np_total / parts + 64));
            const int64_t cmax = std::max<int64_t>(1, std::min<int64_t>(fr.spp, std::min<int64_t>((((int64_t)1 << 29) - 1) / biggest, kn.path_samp_bytes / (int64_t)(np_total * 16 + 1))));
            const int64_t chains = (fr.spp + cmax - 1) / cmax;
            chunk = (int)((fr.spp + chains - 1) / chains);
            int rc2;
            if ((rc2 = ensure(ctx, ctx->pathSamp, np_total * 16 * (size_t)chunk)) != RT_OK || (rc2 = ensure(ctx, ctx->pathT, np_total * 16)) != RT_OK) return rc2;
        }
        ctx->stats.lds_bytes = (int)lds;
        ctx->stats.block_threads = rtk::kQBlock;
        ctx->stats.parts = parts;
        if (int rs = need_part_streams(ctx, parts); rs != RT_OK) return rs;
        if (rec_begin) RT_HIP(ctx, hipEventRecord(ctx->ev_k0, stream));
        if (parts > 1) RT_HIP(ctx, hipEventRecord(ctx->fork_ev, stream));
        for (int j = 0; j < parts; ++j) {
            hipStream_t q = j == 0 ? stream : ctx->part_stream[j];
            if (j > 0) RT_HIP(ctx, hipStreamWaitEvent(q, ctx->fork_ev, 0));
            if (pv[j].n_paths > 0) {
                for (int s0 = 0; s0 < fr.spp; s0 += chunk) {
                    rtk::PathState ps{};
                    ps.n_paths = pv[j].n_paths; ps.tiles_x = tiles_x;
                    ps.samp0 = s0; ps.n_samp = std::min(chunk, fr.spp - s0);
                    ps.samp_out = fr.spp > 1 ? static_cast<float4 *>(ctx->pathSamp.p) + pv[j].base * (size_t)chunk : nullptr;
                    const int64_t n_items = (int64_t)ps.n_paths * ps.n_samp;
                    ps.n_groups = (int)(n_items / 4);
                    // every workgroup owns an equal, spatially scrambled share of the items; its waves draw from it on demand; the grid is
                    // oversubscribed so that the dispatcher evens out the cost differences between the shares
                    int64_t tblocks = std::max<int64_t>(1, (int64_t)ctx->n_cus * bpc / parts) * kn.path_oversub;
                    const int min_groups = kn.min_groups * wpb;       // default 16 per wave: >= 64 items per wave on average
                    int64_t groups_per_block = (ps.n_groups + tblocks - 1) / tblocks;
                    if (groups_per_block < min_groups) {              // small launch: fewer, fuller workgroups
                        tblocks = std::max<int64_t>(1, (ps.n_groups + min_groups - 1) / min_groups);
                        groups_per_block = (ps.n_groups + tblocks - 1) / tblocks;
                    }
                    ps.log2S = 0;
                    while ((2 << ps.log2S) <= groups_per_block && ps.log2S < 16) ++ps.log2S;
                    if (kn.log2S >= 0 && kn.log2S < ps.log2S) ps.log2S = kn.log2S;
                    const int S = 1 << ps.log2S;
                    ps.Q = (ps.n_groups + S - 1) / S;
                    const int64_t total_slots = (int64_t)S * ps.Q * 4;
                    ps.slots_per_block = (int)(((total_slots + tblocks - 1) / tblocks + 3) / 4 * 4);
                    if (j == 0 && s0 == 0) ctx->stats.grid_blocks = (int)tblocks;
                    const dim3 tg((unsigned)tblocks), tbd(rtk::kQBlock);
                    if (work_dev) hipLaunchKernelGGL(rtk::wf_path<true>, tg, tbd, lds, q, scn, pv[j].fr, ps, qcap, kn.path_low, kn.path_shade_min);
                    else hipLaunchKernelGGL(rtk::wf_path<false>, tg, tbd, lds, q, scn, pv[j].fr, ps, qcap, kn.path_low, kn.path_shade_min);
                    if (fr.spp > 1)
                        hipLaunchKernelGGL(rtk::path_reduce, dim3((unsigned)((ps.n_paths + 255) / 256)), dim3(256), 0, q, pv[j].fr, ps.n_paths, ps.tiles_x, ps.n_samp,
                                           static_cast<const float4 *>(ps.samp_out), static_cast<float4 *>(ctx->pathT.p) + pv[j].base, s0 == 0 ? 1 : 0, s0 + chunk >= fr.spp ? 1 : 0);
                }
            }
            if (j > 0) RT_HIP(ctx, hipEventRecord(ctx->part_ev[j], q));
        }
        for (int j = 1; j < parts; ++j) RT_HIP(ctx, hipStreamWaitEvent(stream, ctx->part_ev[j], 0));
    } else if (variant == RT_VARIANT_WAVEFRONT || variant == RT_VARIANT_WAVEFRONT_LDS || variant == RT_VARIANT_WAVEFRONT_QUEUE) {
        const bool ldsn = variant == RT_VARIANT_WAVEFRONT_LDS;
        const bool queue = variant == RT_VARIANT_WAVEFRONT_QUEUE;
        const Knobs &kn = ctx->knobs;
        const int qR = kn.travq_R;                                    // ray slots per wave of the work-stack kernel
        // the 4-wide BOX step (RT_TRAVQ_QW): plain launches only; a counting run keeps the binary instantiation (its counters are the reference's) unless RT_TRAVQ_QW_COUNT
        const bool qw = queue && scn.nodesw != nullptr && qR == 64 && !want_ldsv && !want_ldsn && kn.travq_lds == 0 && (work_dev == nullptr || kn.qw_count);
        int qcap = travq_stack_cap(qR, qw);
        if (kn.travq_cap >= 128 && kn.travq_cap < qcap) qcap = kn.travq_cap;   // tests: force the serial drain
        // BVH nodes staged in LDS (breadth-first prefix) by ONE workgroup of qW waves per CU; 0 = nodes through L1/L2
        int qW = kn.travq_lds;
        int q_nlds = 0;
        const bool mesh_here = ctx->scene.mesh_slot >= 0 && ctx->scene.n_nodes > 0;
        const bool ldsv = queue && want_ldsv && mesh_here;
        const bool ldsn_q = queue && mesh_here && (want_ldsn || qW > 0);
        if (ldsn_q || ldsv) {
            const int64_t carve = (int64_t)travq_carve_bytes(qR);
            const int64_t budget = 160 * 1024 - 16 - (ldsv ? (int64_t)ctx->scene.n_verts * 16 : 0);
            if (budget < carve) return fail(ctx, RT_ERR_UNSUPPORTED, "%d vertices need %lld bytes of LDS: no room for a wave next to them (160 KiB per CU)",
                                            ctx->scene.n_verts, (long long)ctx->scene.n_verts * 16);
            if (qW == 0) qW = ldsn_q ? 12 : 16;                       // measured (cat, 1080p): 12 waves + all nodes beats 16 waves + the top levels
            qW = (int)std::min<int64_t>(qW, budget / carve);
            const int64_t room = budget - (int64_t)qW * carve;
            q_nlds = ldsn_q ? (int)(std::min<int64_t>(room / 32, ctx->scene.n_nodes + 1) & ~(int64_t)1) : 0;   // even: sibling pairs stay together
            if (q_nlds < 4) { q_nlds = 0; if (!ldsv) qW = 0; }       // not even the root's children (nodes 2, 3: the pair every ray starts with) fit, or the root is a leaf: plain kernel
        } else {
            qW = 0;
        }
        const bool qlds = queue && qW > 0;                            // ONE workgroup of qW waves per CU
        // (round 4's RT_TRAVQ_TOPLDS -- the ordinary 4-wave launch with the top of the tree staged per workgroup -- lost by 7-20 % and is gone: DESIGN.md section 10)
        if (!qlds) q_nlds = 0;
        const bool qldsn = qlds && q_nlds > 0;
        const int q_low = kn.q_low * (qR == 128 ? 2 : 1);              // refill thresholds of the work-stack kernel (stack entries are sibling pairs)
        const int q_minfree = (kn.q_minfree >= 1 && kn.q_minfree <= qR) ? kn.q_minfree : qR / 4;
        // begin, (trav, advance) x 2*segments per sample; path state SoA in HBM, tile-order path index.
        // The rows are cut into `parts` independent sub-frames (interleaved tiles), each running its own kernel
        // sequence on its own stream: the traversal kernel ends in a latency-bound tail (a few long rays), and
        // the other parts' kernels fill the SIMDs that a tail leaves idle.  (More than 3 concurrent streams fall off
        // a cliff on this runtime: 4 hardware queues per process.)
        int parts = std::min(kn.parts, (int)rt_ctx::kMaxParts);
        int R = rows->tile_rows, G = rows->tile_step;
        if (G == 1) R = 8;                                            // contiguous rows: any tile height describes them
        const int T = (rows->n_rows + R - 1) / R;                     // local tiles of this call
        if (R % 8 != 0 || work_dev || kn.debug_trav != -2) parts = 1;
        // (one sub-frame for SMALL frames was measured in round 6: 512x512 back to back 0.297 -> 0.328 ms at one sample, 0.82 -> 1.03 ms at eight: two stay, profiles/round6/small_frame_parts.txt)
        if (parts > T) parts = T > 0 ? T : 1;
        const int tiles_x = (p->width + 7) / 8;
        const int tb = qlds ? 64 * qW : queue ? travq_block_threads(qR) : ldsn ? rtk::kTravBlockLds : rtk::kTravBlock;
        const int wpb = tb / 64;
        const size_t q_lds = (size_t)wpb * travq_carve_bytes(qR, qw) + 16 + (size_t)q_nlds * 32 + (ldsv ? (size_t)ctx->scene.n_verts * 16 : 0);
        const size_t trav_lds = queue ? q_lds : ldsn ? lds_nodes_bytes : (size_t)(rtk::kTravBlock / 64) * rtk::TravCarve<512, 8>::kBytes + 16;
        if (!ctx->trav_attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rtk::wf_trav<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rtk::wf_trav<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            ctx->trav_attr_set = true;
        }
        const int si = (work_dev ? 1 : 0) + (ldsn ? 2 : 0);
        if (!queue && ctx->trav_blocks_per_cu[si] == 0) {
            int nb = 0;
            if (ldsn) nb = 1;
            else if (work_dev) RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk::wf_trav<true, false>, rtk::kTravBlock, trav_lds));
            else RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk::wf_trav<false, false>, rtk::kTravBlock, trav_lds));
            ctx->trav_blocks_per_cu[si] = nb > 0 ? nb : 1;
        }
        int bpc = ctx->trav_blocks_per_cu[si];                       // blocks per CU
        if (queue) {
            const int qi = (work_dev ? 1 : 0) + (qR == 32 ? 2 : qR == 128 ? 4 : 0);
            if (qlds) {
                bpc = 1;
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(travq_fn(work_dev != nullptr, qR, qldsn, ldsv)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            } else if (qw) {
                int &nbq = ctx->travq_blocks_per_cu_qw[work_dev ? 1 : 0];
                if (nbq == 0) {
                    int nb = 0;
                    RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, travq_fn(work_dev != nullptr, qR, false, false, true, true), tb, trav_lds));
                    nbq = nb > 0 ? nb : 1;
                }
                bpc = std::min(nbq, (kn.bpc5 ? 20 : 16) / (tb / 64));
            } else {
                if (ctx->travq_blocks_per_cu[qi] == 0) {
                    int nb = 0;
                    RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, travq_fn(work_dev != nullptr, qR, false, false), tb, trav_lds));
                    ctx->travq_blocks_per_cu[qi] = nb > 0 ? nb : 1;
                }
                bpc = std::min(ctx->travq_blocks_per_cu[qi], (kn.bpc5 ? 20 : 16) / (tb / 64));    // a fifth workgroup per CU fits but does not pay (measured)
            }
        }
        if (!ldsn && kn.trav_waves >= 1 && kn.trav_waves <= bpc) bpc = kn.trav_waves;
        const bool have_mesh = ctx->scene.mesh_slot >= 0 && ctx->scene.n_nodes > 0;
        int rc2;
#ifdef RT_DEBUG
        const bool dbg_env = kn.debug_trav != -2;
        const int dbg_it = kn.debug_trav;
        if (dbg_env) { rc2 = ensure(ctx, ctx->dbgbuf, 10 * 8 * 65536); if (rc2 != RT_OK) return rc2; }
#endif

        // samples of a pixel are independent paths: a launch chain traces `chunk` of them at once (bigger launches, fewer tails) as
        // long as the chain's state (~130 bytes per item) stays around the size of the Infinity Cache (RT_PATH_SAMP_MB, default 400)
        int chunk = 1;
        if (batch) chunk = batch->n;                                  // the chain's items are (frame, pixel slot) pairs: num_rays == 1 (rt_render_device_batch checks)
        if (fr.spp > 1) {
            const int64_t px_all = (int64_t)tiles_x * ((rows->n_rows + 7) / 8 + parts) * 64;
            const int64_t per_item = 16 + 16 + 64 + 16 + 5 * (int64_t)nseg;
            const int64_t cmax = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(fr.spp, kn.path_samp_bytes / (px_all * per_item)), (((int64_t)1 << 29) - 1) / (px_all / parts + 64)));
            const int64_t chains = (fr.spp + cmax - 1) / cmax;
            chunk = (int)((fr.spp + chains - 1) / chains);            // chains of (almost) equal size: 64 samples at 15 per chain = 4 x 13 + 12
        }
        // per-part geometry.  A batch of an even number of frames is cut by FRAMES, not by tiles: both sub-frames hold every pixel of the call and half of the frames, so they are
        // exactly as long as each other (a 1/8 share of 1080p is 17 tiles: 9 + 8 would leave one chain 12 % longer than the other)
        struct Part { rtk::Frame fr; rtk::WfState st; int64_t tblocks; unsigned pblocks; size_t base; size_t qbase; size_t pxbase; int batch0, batch_n; };
        std::vector<Part> pv(parts);
        size_t np_total = 0, px_total = 0;
        const bool by_frames = batch && parts > 1 && batch->n % parts == 0;
        for (int j = 0; j < parts; ++j) {
            Part &pt = pv[j];
            const int Tj = (T - j + parts - 1) / parts;               // local tiles j, j+parts, ...
            int nrows_j = Tj * R;
            if (Tj > 0 && (T - 1) % parts == j) nrows_j -= T * R - rows->n_rows;   // the last local tile may be partial
            pt.fr = fr;
            pt.batch0 = 0; pt.batch_n = batch ? batch->n : 0;
            int chunk_j = chunk;
            if (by_frames) {
                nrows_j = rows->n_rows;
                pt.fr.row0 = rows->row0; pt.fr.n_rows = nrows_j; pt.fr.tile_rows = R; pt.fr.tile_step = G;
                pt.fr.out_tile0 = 0; pt.fr.out_tile_step = 1;
                chunk_j = batch->n / parts;
                pt.batch0 = j * chunk_j; pt.batch_n = chunk_j;
            } else {
            pt.fr.row0 = rows->row0 + j * R * G; pt.fr.n_rows = nrows_j; pt.fr.tile_rows = R; pt.fr.tile_step = G * parts;
            pt.fr.out_tile0 = j; pt.fr.out_tile_step = parts;
            }
            pt.st = rtk::WfState{};
            pt.st.tiles_x = tiles_x; pt.st.tiles_x_m = rtk::wf_div_magic(tiles_x);
            const int64_t n_px64 = (int64_t)tiles_x * ((nrows_j + 7) / 8) * 64;
            const int64_t n_paths64 = n_px64 * chunk_j;
            // slot arithmetic is 32-bit: ((col << log2S | a) << 2) and 2 * n_paths / 4 must stay below 2^31
            if (n_paths64 >= ((int64_t)1 << 29)) return fail(ctx, RT_ERR_INVALID, "image too large: %lld paths per sub-frame (limit 2^29)", (long long)n_paths64);
            pt.st.n_paths = (int)n_paths64;
            pt.st.n_px = (int)n_px64; pt.st.n_px_m = rtk::wf_div_magic((int)n_px64);
            pt.base = np_total;
            pt.pxbase = px_total;
            np_total += (size_t)n_paths64;
            px_total += (size_t)n_px64;
            int64_t tblocks = 0;
            wf_geometry(kn, ctx->n_cus, bpc, parts, wpb, queue && !qlds, pt.st, tblocks);
            pt.tblocks = tblocks;
            pt.pblocks = (unsigned)((n_paths64 + kn.adv_block - 1) / kn.adv_block);
        }
        const size_t np = np_total;
        if ((rc2 = ensure(ctx, ctx->wfM, 2 * np * 8)) != RT_OK ||
            (rc2 = ensure(ctx, ctx->wfT, (fr.spp > 1 ? px_total : 1) * 16)) != RT_OK || (rc2 = ensure(ctx, ctx->wfSamp, (fr.spp > 1 ? np : 1) * 16)) != RT_OK ||
            (rc2 = ensure(ctx, ctx->wfSID, np * (size_t)nseg)) != RT_OK || (rc2 = ensure(ctx, ctx->wfLS, np * 4 * (size_t)nseg)) != RT_OK)
            return rc2;
        size_t q_slots = 0;                                           // traversal-queue slots of all parts (padding included)
        uint64_t q_sig = 0xcbf29ce484222325ull, layout_sig = 0;
        bool own0 = false, rejoin = false, zeroed = false;
        {
            for (Part &pt : pv) {
                pt.qbase = q_slots;
                q_slots += (size_t)pt.st.slots_per_block * (size_t)pt.tblocks;
                for (uint64_t v : {(uint64_t)pt.st.n_paths, (uint64_t)pt.st.log2S, (uint64_t)pt.st.Q, (uint64_t)pt.st.slots_per_block, (uint64_t)pt.tblocks})
                    q_sig = (q_sig ^ v) * 0x100000001b3ull;
            }
            // chains on their own streams: a chunk whose state layout differs from the previous chunk's must not start while that one's
            // chains are running (its sub-frames' state would overlap theirs), and neither may the queue's zero fill below
            // (not for the chunks of a call when the second sub-frame's stream sits in the high-priority class, RT_PART_PRIO: without a join per
            // chunk the favoured chain runs ahead through all its chunks and the other one finishes alone: 2.06 -> 2.53 ms for half a 3840x2160 frame)
            own0 = (ctx->pipe.on || (ctx->pipe.call_chunks > 1 && !kn.part_prio)) && parts > 1 && !work_dev && kn.debug_trav == -2;
            layout_sig = q_sig;
            for (const Part &pt : pv) for (uint64_t v : {(uint64_t)pt.base, (uint64_t)pt.pxbase, (uint64_t)pt.st.n_px, (uint64_t)fr.spp, (uint64_t)nseg}) layout_sig = (layout_sig ^ v) * 0x100000001b3ull;
            if (ctx->pipe.open_parts > 0 && (!own0 || ctx->pipe.sig != layout_sig)) {
                for (int j = 0; j < ctx->pipe.open_parts; ++j) RT_HIP(ctx, hipStreamWaitEvent(stream, ctx->part_ev[j], 0));   // join the previous chunk (its chains recorded part_ev)
                ctx->pipe.open_parts = 0;
                rejoin = true;
            }
            {
            const size_t had = ctx->wfQR.bytes;
            if ((rc2 = ensure(ctx, ctx->wfQR, q_slots * 32)) != RT_OK) return rc2;
            if (ctx->wfQR.bytes != had || ctx->qf_sig != q_sig || work_dev) {   // padding slots are never written by the kernels: zero once per layout.  (A counting run zeroes too:
                                                                                   // a stale shadow record that passes wq_live costs only a traversal, but the counters would see it)
                RT_HIP(ctx, hipMemsetAsync(ctx->wfQR.p, 0, ctx->wfQR.bytes, stream));
                ctx->qf_sig = q_sig;
                zeroed = true;
            }
            }
        }
        for (Part &pt : pv) {
            rtk::WfState &st = pt.st;
            st.QR = static_cast<float4 *>(ctx->wfQR.p) + 2 * pt.qbase;
            st.init_m = queue ? 0 : 1;                               // wf_trav merges split traversals with atomicMin
            st.M = static_cast<unsigned long long *>(ctx->wfM.p) + 2 * pt.base;
            st.samp_out = fr.spp > 1 ? static_cast<float4 *>(ctx->wfSamp.p) + pt.base : nullptr;
            st.LS = static_cast<float *>(ctx->wfLS.p) + pt.base * (size_t)nseg;   // LS[d * n_paths + i] inside the part's block
            st.SID = static_cast<unsigned char *>(ctx->wfSID.p) + pt.base * (size_t)nseg;
            st.batch = nullptr; st.n_batch = 0;
            st.anyhit = (kn.anyhit && (work_dev == nullptr || qw)) ? 1 : 0;   // a run that counts the REFERENCE's work (the float-pair counting instantiation) traces every shadow ray to the end
        }
        if (batch) {
            // the frames' descriptors live in device memory, one copy PER SUB-FRAME, written by a one-wave kernel at the head of that sub-frame's own chain (below): a chain is
            // ordered behind the previous chain of its stream, so the copy is never rewritten under a running kernel, and nothing has to wait on the caller's stream -- a batch
            // takes the relaxed start of rt_ctx_set_pipelining like a frame does
            if ((rc2 = ensure(ctx, ctx->batch_dev, rt_ctx::kMaxParts * rtk::kMaxBatch * sizeof(rtk::BatchFrame))) != RT_OK) return rc2;
            for (int j = 0; j < parts; ++j) {
                pv[j].st.batch = static_cast<rtk::BatchFrame *>(ctx->batch_dev.p) + j * rtk::kMaxBatch;
                pv[j].st.n_batch = pv[j].batch_n;
            }
        }
        ctx->stats.lds_bytes = (int)trav_lds;
        ctx->stats.block_threads = tb;
        ctx->stats.grid_blocks = (int)pv[0].tblocks;
        ctx->stats.parts = parts;
        ctx->stats.travq_mode = (queue && have_mesh) ? (qw ? 2 : (scn.nodesh != nullptr && !work_dev && qR == 64 && !qldsn && !ldsv) ? 1 : 0) : -1;
        if (int rs = need_part_streams(ctx, parts, own0); rs != RT_OK) return rs;
        if (rec_begin) RT_HIP(ctx, hipEventRecord(ctx->ev_k0, stream));
        // Where the chains start.  Chain 0 on the caller's stream, the others forked from it and joined back at the end (one chunk, no
        // pipelining); or every chain on a stream of its own (own0): forked once per CALL, joined once per call, chunk after chunk
        // following per sub-frame without a join -- a sub-frame's next chunk re-uses exactly its own state -- unless the layout changes.
        // With rt_ctx_set_pipelining the same holds across calls: a frame into a buffer the previous frame did not use starts behind
        // what was on the caller's stream when the PREVIOUS call was made (everything that could read or write this frame's buffer is
        // older than that), so its sub-frames follow the previous frame's sub-frames one by one and no stream idles at a frame boundary.
        rt_ctx::Pipe &pl = ctx->pipe;
        hipEvent_t start_ev = ctx->fork_ev;
        bool fork = parts > 1;
        if (own0) {
            if (!pl.fork2[0]) { RT_HIP(ctx, hipEventCreateWithFlags(&pl.fork2[0], hipEventDisableTiming)); RT_HIP(ctx, hipEventCreateWithFlags(&pl.fork2[1], hipEventDisableTiming)); }
            if (pl.call_chunk == 0) {
                const bool disjoint = pl.call_hi <= pl.out_lo || pl.out_hi <= pl.call_lo;
                bool hazard = pl.between_overflow;                       // a library call younger than the previous render call touches this frame's buffer (or: too many to tell)
                for (const rt_ctx::Pipe::Range &r : pl.between) if (r.stream == stream && r.lo < pl.call_hi && pl.call_lo < r.hi) hazard = true;
#ifdef RT_DEBUG
                if (pl.on && pl.prev_valid && pl.stream == stream && hazard)
                    return fail(ctx, RT_ERR_INVALID, "pipelining rule broken: work submitted to this stream after the previous render call (rt_tonemap_device) touches the buffer "
                                                     "this frame renders into; with rt_ctx_set_pipelining the frame would not wait for it (raytrace_hip.h)");
#endif
                const bool relaxed = pl.on && pl.prev_valid && pl.stream == stream && pl.sig == layout_sig && disjoint && !zeroed && !hazard;
                pl.cur ^= 1;
                RT_HIP(ctx, hipEventRecord(pl.fork2[pl.cur], stream));
                start_ev = pl.fork2[relaxed ? pl.cur ^ 1 : pl.cur];
            } else if (rejoin || zeroed || pl.open_parts == 0) {
                RT_HIP(ctx, hipEventRecord(ctx->fork_ev, stream));
            } else {
                fork = false;                                            // the chains go on where the previous chunk left them
            }
        } else if (fork) {
            RT_HIP(ctx, hipEventRecord(ctx->fork_ev, stream));
        }
        // Every sub-frame's chain starts behind the fork; then the chains of ONE sample chunk are issued for all sub-frames before the next chunk's.
        // (Rounds 2-4 issued all chunks of sub-frame 0 first: with hundreds of chains the host was still feeding stream 0 while stream 1 sat empty, the
        // sub-frames ran one after the other instead of side by side, and a 256-sample 1080p frame cost 1.26 ms per sample against 0.94 at 32 samples --
        // tools/spp_slope.py, profiles/round5/spp_slope.txt.)
        for (int j = 0; j < parts; ++j) {
            hipStream_t q = (j == 0 && !own0) ? stream : ctx->part_stream[j];
            if (fork && (j > 0 || own0)) RT_HIP(ctx, hipStreamWaitEvent(q, start_ev, 0));
            if (own0 && pl.call_chunk == 0 && pl.extra_wait) RT_HIP(ctx, hipStreamWaitEvent(q, pl.extra_wait, 0));
        }
        for (int s = 0; s < fr.spp; s += chunk) {
            for (int j = 0; j < parts; ++j) {
                Part &pt = pv[j];
                hipStream_t q = (j == 0 && !own0) ? stream : ctx->part_stream[j];
                if (pt.st.n_paths == 0) continue;
                pt.st.samp0 = s;
                pt.st.epoch = 0;
                pt.st.nonce = (int)(++ctx->chain_nonce[j] & (unsigned)rtk::PQ_NONCE_MASK);
                if (batch) {                                          // this sub-frame's frames, at the head of its chain
                    rtk::Batch bj{};
                    bj.n = pt.batch_n;
                    for (int k = 0; k < pt.batch_n; ++k) bj.f[k] = batch->f[pt.batch0 + k];
                    hipLaunchKernelGGL(rtk::batch_store_kernel, dim3(1), dim3(64), 0, q, bj, const_cast<rtk::BatchFrame *>(pt.st.batch));
                }
                if (work_dev) hipLaunchKernelGGL((rtk::wf_advance<true, true>), dim3(pt.pblocks), dim3(kn.adv_block), 0, q, scn, pt.fr, pt.st);
                else hipLaunchKernelGGL((rtk::wf_advance<false, true>), dim3(pt.pblocks), dim3(kn.adv_block), 0, q, scn, pt.fr, pt.st);
                for (int it = 0; it < (segs > 0 ? segs + 1 : 0); ++it) {
                    pt.st.epoch = it;
                    if (have_mesh) {
#ifdef RT_DEBUG
                        pt.st.dbg = (dbg_env && it == dbg_it) ? static_cast<unsigned long long *>(ctx->dbgbuf.p) : nullptr;
#endif
                        const bool timed = ctx->stats_on && j == 0 && s + chunk >= fr.spp;   // on request (rt_stats_enable): time part 0's traversal launches of the last chain
                        if (timed) RT_HIP(ctx, hipEventRecord(ctx->ev_trav[2 * it], q));
                        const dim3 tg((unsigned)pt.tblocks), tbd(tb);
                        if (queue) {
                            hipLaunchKernelGGL(travq_fn(work_dev != nullptr, qR, qldsn, ldsv, scn.nodesh != nullptr, qw), tg, tbd, trav_lds, q, scn, pt.fr, pt.st, qcap, q_nlds, q_low, q_minfree);
                        } else if (ldsn) {
                            if (work_dev) hipLaunchKernelGGL((rtk::wf_trav<true, true>), tg, tbd, trav_lds, q, scn, pt.fr, pt.st);
                            else hipLaunchKernelGGL((rtk::wf_trav<false, true>), tg, tbd, trav_lds, q, scn, pt.fr, pt.st);
                        } else {
                            if (work_dev) hipLaunchKernelGGL((rtk::wf_trav<true, false>), tg, tbd, trav_lds, q, scn, pt.fr, pt.st);
                            else hipLaunchKernelGGL((rtk::wf_trav<false, false>), tg, tbd, trav_lds, q, scn, pt.fr, pt.st);
                        }
                        if (timed) { RT_HIP(ctx, hipEventRecord(ctx->ev_trav[2 * it + 1], q)); ctx->n_trav_events = it + 1; }
                        pt.st.dbg = nullptr;
                    }
                    const bool timed_adv = ctx->stats_on && j == 0 && s + chunk >= fr.spp;
                    if (timed_adv) RT_HIP(ctx, hipEventRecord(ctx->ev_adv[2 * it], q));
                    if (work_dev) hipLaunchKernelGGL((rtk::wf_advance<true, false>), dim3(pt.pblocks), dim3(kn.adv_block), 0, q, scn, pt.fr, pt.st);
                    else hipLaunchKernelGGL((rtk::wf_advance<false, false>), dim3(pt.pblocks), dim3(kn.adv_block), 0, q, scn, pt.fr, pt.st);
                    if (timed_adv) { RT_HIP(ctx, hipEventRecord(ctx->ev_adv[2 * it + 1], q)); ctx->n_adv_events = it + 1; ctx->adv_paths = pt.st.n_paths; }
                }
                if (fr.spp > 1)                                       // the chain's samples, added in sample order (cpu:711), into the running sum / the frame
                    hipLaunchKernelGGL(rtk::path_reduce, dim3((unsigned)((pt.st.n_px + 255) / 256)), dim3(256), 0, q, pt.fr, pt.st.n_px, tiles_x, std::min(chunk, fr.spp - s),
                                       static_cast<const float4 *>(pt.st.samp_out), static_cast<float4 *>(ctx->wfT.p) + pt.pxbase, s == 0 ? 1 : 0, s + chunk >= fr.spp ? 1 : 0);
            }
        }
        for (int j = 0; j < parts; ++j) {
            hipStream_t q = (j == 0 && !own0) ? stream : ctx->part_stream[j];
            if (j > 0 || own0) { RT_HIP(ctx, hipEventRecord(ctx->part_ev[j], q)); }
        }
        if (!own0) {
            for (int j = 1; j < parts; ++j) RT_HIP(ctx, hipStreamWaitEvent(stream, ctx->part_ev[j], 0));
        } else {
            pl.sig = layout_sig;
            pl.open_parts = parts;
            if (pl.call_chunk + 1 >= pl.call_chunks) {                   // the call's last chunk: its result is complete behind this join
                for (int j = 0; j < parts; ++j) RT_HIP(ctx, hipStreamWaitEvent(stream, ctx->part_ev[j], 0));
                pl.open_parts = 0;
                pl.valid = true; pl.stream = stream; pl.out_lo = pl.call_lo; pl.out_hi = pl.call_hi;
            }
        }
#ifdef RT_DEBUG
        if (dbg_env) {       // tools/dbg_travq.py: per-wave records of one traversal launch
            std::vector<unsigned long long> h(10 * (size_t)65536);
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(h.data(), ctx->dbgbuf.p, h.size() * 8, hipMemcpyDeviceToHost);
            FILE *f = fopen("gpurun_out/trav_dbg.bin", "wb");
            if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        }
#endif
    } else if (variant == RT_VARIANT_LOCKSTEP) {
        dim3 grid((p->width + rtk::kTileW - 1) / rtk::kTileW, (rows->n_rows + rtk::kTileH - 1) / rtk::kTileH);
        const size_t lds = (size_t)nseg * rtk::kBlockThreads * sizeof(float);
        ctx->stats.lds_bytes = (int)lds;
        ctx->stats.block_threads = rtk::kBlockThreads;
        ctx->stats.grid_blocks = (int)(grid.x * grid.y);
        if (rec_begin) RT_HIP(ctx, hipEventRecord(ctx->ev_k0, stream));
        if (work_dev) hipLaunchKernelGGL(rtk::render_kernel<true>, grid, dim3(rtk::kBlockThreads), lds, stream, scn, fr);
        else hipLaunchKernelGGL(rtk::render_kernel<false>, grid, dim3(rtk::kBlockThreads), lds, stream, scn, fr);
    } else {
        // persistent lanes: as many workgroups as are co-resident, pixels drawn from a global queue
        rtk::PFrame pf{};
        pf.f = fr;
        pf.tiles_x = (p->width + 7) / 8;
        const int64_t slots = (int64_t)pf.tiles_x * ((rows->n_rows + 7) / 8) * 64;
        if (slots >= ((int64_t)1 << 32) - 65536) return fail(ctx, RT_ERR_INVALID, "image too large for the pixel queue");
        pf.n_slots = (unsigned int)slots;
        int rc2 = ensure(ctx, ctx->queue, sizeof(unsigned int));
        if (rc2 != RT_OK) return rc2;
        pf.queue = static_cast<unsigned int *>(ctx->queue.p);
        const size_t lds = (size_t)nseg * rtk::kPBlock * sizeof(float);
        const int si = work_dev ? 1 : 0;
        if (ctx->persist_blocks_per_cu[si] == 0) {
            int nb = 0;
            if (work_dev) RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk::render_persistent<true>, rtk::kPBlock, (size_t)RT_MAX_SEGMENTS * rtk::kPBlock * sizeof(float)));
            else RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk::render_persistent<false>, rtk::kPBlock, (size_t)RT_MAX_SEGMENTS * rtk::kPBlock * sizeof(float)));
            ctx->persist_blocks_per_cu[si] = nb > 0 ? nb : 1;
        }
        int64_t blocks = (int64_t)ctx->n_cus * ctx->persist_blocks_per_cu[si];
        const int64_t useful = (slots + rtk::kPBlock - 1) / rtk::kPBlock;
        if (blocks > useful) blocks = useful;
        ctx->stats.lds_bytes = (int)lds;
        ctx->stats.block_threads = rtk::kPBlock;
        ctx->stats.grid_blocks = (int)blocks;
        RT_HIP(ctx, hipMemsetAsync(pf.queue, 0, sizeof(unsigned int), stream));
        if (rec_begin) RT_HIP(ctx, hipEventRecord(ctx->ev_k0, stream));
        if (work_dev) hipLaunchKernelGGL(rtk::render_persistent<true>, dim3((unsigned)blocks), dim3(rtk::kPBlock), lds, stream, scn, pf);
        else hipLaunchKernelGGL(rtk::render_persistent<false>, dim3((unsigned)blocks), dim3(rtk::kPBlock), lds, stream, scn, pf);
    }
    RT_HIP(ctx, hipGetLastError());
    if (rec_end) { RT_HIP(ctx, hipEventRecord(ctx->ev_k1, stream)); ctx->have_kernel_time = true; }
    return RT_OK;
}

// The wavefront pipeline streams ~150 bytes of path state per pixel and launch through the memory system.  While a (sub-)frame's state
// fits the 256 MB Infinity Cache the uniform kernel runs at the rate the headline 1080p frame shows; a 3840x2160 frame (1 GB of
// state) does not, and ran 8 % slower per ray.  So a call is cut into sequential chunks of about RT_CHUNK_MPX million pixels
// (default 2.3: a 1080p frame is ONE chunk) of whole tiles; every chunk is the same pipeline on the same streams and buffers.
int launch_render(rt_ctx *ctx, const rt_params *p, const rt_rows *rows, void *out_dev, hipStream_t stream,
                  unsigned long long *work_dev = nullptr, const rt_camera_pose *pose = nullptr) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    const int v = p ? p->variant : 0;
    // (the work-stack pipeline and its LDS-staged variants; the per-lane-walk variants with one big workgroup per CU lose more to the
    // smaller launches than the cache gives back: wavefront_lds 7.9 -> 9.1 ms at 3840x2160)
    const bool wf = (v == RT_VARIANT_AUTO && !auto_is_lockstep(ctx, pose)) || v == RT_VARIANT_WAVEFRONT_QUEUE || v == RT_VARIANT_LDS_VERTS || v == RT_VARIANT_LDS_TOP || v == RT_VARIANT_LDS_ALL;
    const int64_t chunk_px = (int64_t)(ctx->knobs.chunk_mpx * 1e6);
    rt_ctx::Pipe &pl = ctx->pipe;
    pl.prev_valid = pl.valid; pl.valid = false;                          // every asynchronous user of the path state comes through here
    pl.call_chunk = 0; pl.call_chunks = 1; pl.open_parts = 0;
    struct ClearBetween { rt_ctx::Pipe &p; ~ClearBetween() { p.between.clear(); p.between_overflow = false; } } clear_between{pl};   // the ranges describe the gap BEFORE this call: consumed by it
    pl.call_lo = static_cast<const uint8_t *>(out_dev);
    pl.call_hi = pl.call_lo + ((p && rows && p->width > 0 && rows->n_rows > 0) ? (size_t)rows->n_rows * p->width * sizeof(float4) : 0);
    if (!p || !rows || !wf || chunk_px <= 0 || p->width <= 0 || rows->tile_rows <= 0 || (int64_t)rows->n_rows * p->width <= chunk_px * 5 / 4) {
        return launch_render_chunk(ctx, p, rows, out_dev, stream, work_dev, pose, true, true);
    }
    // rows per chunk: whole tiles (and whole 8-row wave tiles for contiguous rows), two sub-frames' worth at least
    // (contiguous rows: tile_rows only says how the caller described them -- rt_render passes one tile of n_rows -- and every chunk is
    // re-described below; only interleaved tiles must be cut at tile boundaries)
    int unit = rows->tile_step == 1 ? 16 : rows->tile_rows * 2;
    if (rows->tile_step != 1 && unit % rows->tile_rows != 0) unit *= rows->tile_rows;
    const int64_t n_chunks = ((int64_t)rows->n_rows * p->width + chunk_px - 1) / chunk_px;
    int per = (int)(((int64_t)rows->n_rows + n_chunks - 1) / n_chunks);
    per = (per + unit - 1) / unit * unit;
    uint64_t pixels = 0;
    pl.call_chunks = (rows->n_rows + per - 1) / per;
    for (int a = 0; a < rows->n_rows; a += per, ++pl.call_chunk) {
        const int nr = std::min(per, rows->n_rows - a);
        rt_rows rc{rows->row0 + (a / rows->tile_rows) * rows->tile_rows * rows->tile_step, nr, rows->tile_rows, rows->tile_step};
        if (rows->tile_step == 1) { rc.row0 = rows->row0 + a; rc.tile_rows = nr; }      // contiguous rows: one tile of any height describes them
        const int r = launch_render_chunk(ctx, p, &rc, static_cast<uint8_t *>(out_dev) + (size_t)a * p->width * sizeof(float4), stream, work_dev, pose,
                                          a == 0, a + per >= rows->n_rows);
        if (r != RT_OK) {                                                 // chains of earlier chunks may be running on their own streams: wait for them
            for (hipStream_t q : ctx->part_stream) if (q) (void)hipStreamSynchronize(q);
            pl.valid = false; pl.call_chunk = 0; pl.call_chunks = 1; pl.open_parts = 0;
            return r;
        }
        pixels += ctx->stats.pixels;
    }
    pl.call_chunk = 0; pl.call_chunks = 1;
    ctx->stats.pixels = pixels;
    return RT_OK;
}

int launch_tonemap(rt_ctx *ctx, const void *rgba_dev, int64_t npix, void *rgb8_dev, hipStream_t stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (npix < 0 || (npix > 0 && (!rgba_dev || !rgb8_dev))) return fail(ctx, RT_ERR_INVALID, "bad tonemap arguments");
    if (npix == 0) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t quads = (npix + 3) / 4;
    if (ctx->pipe.on && ctx->pipe.between.size() >= 64) ctx->pipe.between_overflow = true;   // more ranges than are kept: the next render call takes the full fork
    if (ctx->pipe.on && ctx->pipe.between.size() < 64) {              // (see Pipe::between)
        const uint8_t *a = static_cast<const uint8_t *>(rgba_dev), *b = static_cast<const uint8_t *>(rgb8_dev);
        ctx->pipe.between.push_back({a, a + (size_t)npix * sizeof(float4), stream});
        ctx->pipe.between.push_back({b, b + (size_t)npix * 3, stream});
    }
    RT_HIP(ctx, hipEventRecord(ctx->ev_t0, stream));
    hipLaunchKernelGGL(rtk::tonemap_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const float4 *>(rgba_dev), npix, static_cast<uint8_t *>(rgb8_dev));
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipEventRecord(ctx->ev_t1, stream));
    ctx->have_tonemap_time = true;
    return RT_OK;
}

}  // namespace
