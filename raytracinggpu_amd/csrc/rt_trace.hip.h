// rt_trace.hip.h -- rt_trace_rays: a batch of explicit rays through the PRODUCTION traversal launches.
// Included at the end of rt_capi.hip (same translation unit: it uses the launch geometry of launch_render).
//
// TriangleMesh::intersect (cpu_launcher.cpp:238-313; optimized.cu:220-285) is callable with ANY ray; in the render path rays reach
// the traversal kernels only through the uniform kernels (camera rays, bounce and shadow rays), which never produce a zero or
// denormal direction component, an origin inside the mesh on purpose, or the other corner cases the reference's own vectors cover.
// rt_trace_rays writes the caller's rays into the traversal queue exactly as wf_emit_ray does (root-box test, cpu:279, by the
// same slab_filtered; slot-order records) and runs the same kernel instantiations with the same launch geometry a frame uses:
// wf_travq (work stack: BOX / TRI steps, refill, leaf queue, serial drain under RT_TRAVQ_CAP), wf_trav (per-lane stackless walk
// with work splitting) or wf_path (the fused kernel, explicit-ray items).  out[i] = (hit, t, N.xyz) with N normalised as cpu:308.
#pragma once

namespace rtk {

// one lane per ray slot pair: rays r < n are the caller's, the rest of the 2 n_paths slots carry no ray
__global__ __launch_bounds__(256) void trace_emit_kernel(const Scene sc, const WfState st, const float *__restrict__ rays, int n) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 2 * st.n_paths) return;
    const int q = wf_ray_to_slot(st, r);
    bool need = false;
    f3 O = mk(0, 0, 0), u = mk(0, 0, 1);
    if (r < n) {
        const float *p = rays + 6 * (size_t)r;
        O = mk(p[0], p[1], p[2]); u = mk(p[3], p[4], p[5]);
        if (sc.mesh_slot >= 0 && sc.n_nodes > 0) need = slab_filtered(sc.root_lo, sc.root_hi, O, u, ray_inv(u));   // wf_emit_ray's root-box test
    }
    st.M[r] = WF_NOHIT;
    st.QR[2 * (size_t)q] = make_float4(O.x, O.y, O.z, u.x);
    st.QR[2 * (size_t)q + 1] = make_float4(u.y, u.z, __int_as_float(need ? PQ_TRAV : 0), 0.f);
}

__global__ __launch_bounds__(256) void trace_close_kernel(const Scene sc, const unsigned long long *__restrict__ M, int n, float *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const unsigned long long m = M[r];
    float *o = out + 5 * (size_t)r;
    if (m == WF_NOHIT) { o[0] = 0.f; o[1] = 1e9f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; return; }   // t = INF narrowed (cpu:283)
    const float4 q2 = sc.tri[3 * (size_t)(unsigned int)m + 2];
    const f3 N = normalize(mk(q2.y, q2.z, q2.w));                     // cpu:308
    o[0] = 1.f; o[1] = __uint_as_float((unsigned int)(m >> 32)); o[2] = N.x; o[3] = N.y; o[4] = N.z;
}

}  // namespace rtk

extern "C" int rt_trace_rays(rt_ctx *ctx, const float *rays, int n, float tri_tmin, int variant, float *out) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    if (n < 0 || (n > 0 && (!rays || !out))) return fail(ctx, RT_ERR_INVALID, "bad ray batch");
    if (n >= (1 << 28)) return fail(ctx, RT_ERR_INVALID, "at most 2^28 rays per call");
    if (variant == RT_VARIANT_AUTO) variant = RT_VARIANT_WAVEFRONT_QUEUE;
    if (variant != RT_VARIANT_WAVEFRONT_QUEUE && variant != RT_VARIANT_WAVEFRONT && variant != RT_VARIANT_PATH)
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_trace_rays runs the traversal of variant wavefront_queue (wf_travq), wavefront (wf_trav) or path (wf_path)");
    if (n == 0) return RT_OK;
    const rtk::Scene &sc = ctx->scene;
    if (variant == RT_VARIANT_WAVEFRONT_QUEUE && (sc.n_nodes + 2 >= (1 << rtk::kQNodeBits) || !ctx->travq_ok)) variant = RT_VARIANT_WAVEFRONT;
    if (variant == RT_VARIANT_PATH && sc.n_nodes + 2 >= (1 << rtk::kPNodeBits)) variant = RT_VARIANT_WAVEFRONT;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_OWN_STREAM(ctx);
    hipStream_t q = own_stream(ctx);
    const Knobs &kn = ctx->knobs;
    const bool have_mesh = sc.mesh_slot >= 0 && sc.n_nodes > 0;
    DevBuf din, dout;
    int rc;
    auto done = [&](int code) { din.release(); dout.release(); return code; };
    if ((rc = upload(ctx, din, rays, (size_t)n * 6 * sizeof(float))) != RT_OK || (rc = ensure(ctx, dout, (size_t)n * 5 * sizeof(float))) != RT_OK) return done(rc);
    rtk::Frame fr{};
    fr.tri_tmin = tri_tmin; fr.segs = 1; fr.spp = 1; fr.W = 1; fr.H = 1; fr.n_rows = 1; fr.tile_rows = 1; fr.tile_step = 1; fr.out_tile_step = 1;
    unsigned long long *M = nullptr;
    if (variant == RT_VARIANT_PATH) {
        // the fused kernel: items = the rays, in wf_path's own launch geometry (launch_render, RT_VARIANT_PATH)
        constexpr int wpb = rtk::kQBlock / 64;
        const size_t lds = (size_t)wpb * rtk::PCarve::bytes(1) + 16;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtk::wf_path<false>, rtk::kQBlock, lds) != hipSuccess || nb < 1) return done(fail(ctx, RT_ERR_UNSUPPORTED, "wf_path does not fit a CU"));
        const int bpc = std::min(kn.path_bpc, nb);
        rtk::PathState ps{};
        ps.n_paths = (n + 63) / 64 * 64; ps.tiles_x = 1; ps.samp0 = 0; ps.n_samp = 1; ps.samp_out = nullptr;
        ps.n_groups = ps.n_paths / 4;
        if ((rc = ensure(ctx, ctx->wfM, (size_t)ps.n_paths * 8)) != RT_OK) return done(rc);
        M = static_cast<unsigned long long *>(ctx->wfM.p);
        ps.ext_rays = static_cast<const float *>(din.p); ps.ext_out = M; ps.n_ext = n;
        int64_t tblocks = std::max<int64_t>(1, (int64_t)ctx->n_cus * bpc) * kn.path_oversub;
        const int min_groups = kn.min_groups * wpb;
        int64_t groups_per_block = (ps.n_groups + tblocks - 1) / tblocks;
        if (groups_per_block < min_groups) { tblocks = std::max<int64_t>(1, (ps.n_groups + min_groups - 1) / min_groups); groups_per_block = (ps.n_groups + tblocks - 1) / tblocks; }
        ps.log2S = 0;
        while ((2 << ps.log2S) <= groups_per_block && ps.log2S < 16) ++ps.log2S;
        const int S = 1 << ps.log2S;
        ps.Q = (ps.n_groups + S - 1) / S;
        ps.slots_per_block = (int)((((int64_t)S * ps.Q * 4 + tblocks - 1) / tblocks + 3) / 4 * 4);
        int qcap = rtk::kPStack;
        if (kn.travq_cap >= 128 && kn.travq_cap < qcap) qcap = kn.travq_cap;
        RT_HIP(ctx, hipMemsetAsync(M, 0xff, (size_t)ps.n_paths * 8, q));      // rays the kernel never reaches (none) would read as no hit
        hipLaunchKernelGGL(rtk::wf_path<false>, dim3((unsigned)tblocks), dim3(rtk::kQBlock), lds, q, sc, fr, ps, qcap, kn.path_low, kn.path_shade_min);
    } else {
        const bool queue = variant == RT_VARIANT_WAVEFRONT_QUEUE;
        const int qR = kn.travq_R;
        const bool qw = queue && sc.nodesw != nullptr && qR == 64;    // the 4-wide BOX step, as a frame would run it (RT_TRAVQ_QW)
        int qcap = travq_stack_cap(qR, qw);
        if (kn.travq_cap >= 128 && kn.travq_cap < qcap) qcap = kn.travq_cap;   // tests: force the serial drain
        const int tb = queue ? travq_block_threads(qR) : rtk::kTravBlock;
        const int wpb = tb / 64;
        const size_t trav_lds = queue ? (size_t)wpb * travq_carve_bytes(qR, qw) + 16 : (size_t)(rtk::kTravBlock / 64) * rtk::TravCarve<512, 8>::kBytes + 16;
        int bpc = 0;
        if (queue) {
            RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, travq_fn(false, qR, false, false, qw, qw), tb, trav_lds));
            bpc = std::min(bpc > 0 ? bpc : 1, (kn.bpc5 ? 20 : 16) / (tb / 64));
        } else {
            RT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, rtk::wf_trav<false, false>, rtk::kTravBlock, trav_lds));
            if (bpc < 1) bpc = 1;
        }
        rtk::WfState st{};
        st.n_paths = ((n + 1) / 2 + 1) / 2 * 2;                       // 2 n_paths ray slots >= n, a multiple of 4
        st.n_px = st.n_paths; st.tiles_x = 1;
        int64_t tblocks = 0;
        wf_geometry(kn, ctx->n_cus, bpc, 1, wpb, queue, st, tblocks);
        const size_t q_slots = (size_t)st.slots_per_block * (size_t)tblocks;
        if ((rc = ensure(ctx, ctx->wfM, 2 * (size_t)st.n_paths * 8)) != RT_OK || (rc = ensure(ctx, ctx->wfQR, q_slots * 32)) != RT_OK) return done(rc);
        RT_HIP(ctx, hipMemsetAsync(ctx->wfQR.p, 0, q_slots * 32, q));    // padding slots carry no ray
        ctx->qf_sig = 0;                                               // the render path zeroes its own layout again
        st.QR = static_cast<float4 *>(ctx->wfQR.p);
        st.M = M = static_cast<unsigned long long *>(ctx->wfM.p);
        st.init_m = queue ? 0 : 1;
        st.epoch = 0; st.nonce = 0;
        hipLaunchKernelGGL(rtk::trace_emit_kernel, dim3((unsigned)((2 * st.n_paths + 255) / 256)), dim3(256), 0, q, sc, st, static_cast<const float *>(din.p), n);
        if (have_mesh) {
            if (queue) hipLaunchKernelGGL(travq_fn(false, qR, false, false, sc.nodesh != nullptr, qw), dim3((unsigned)tblocks), dim3(tb), trav_lds, q, sc, fr, st, qcap, 0, kn.q_low * (qR == 128 ? 2 : 1),
                                          (kn.q_minfree >= 1 && kn.q_minfree <= qR) ? kn.q_minfree : qR / 4);
            else hipLaunchKernelGGL((rtk::wf_trav<false, false>), dim3((unsigned)tblocks), dim3(tb), trav_lds, q, sc, fr, st);
        }
    }
    hipLaunchKernelGGL(rtk::trace_close_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, sc, M, n, static_cast<float *>(dout.p));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, (size_t)n * 5 * sizeof(float), hipMemcpyDeviceToHost, q);
    if (e == hipSuccess) e = hipStreamSynchronize(q);
    if (e != hipSuccess) return done(fail(ctx, RT_ERR_HIP, "rt_trace_rays: %s", hipGetErrorString(e)));
    return done(RT_OK);
}
