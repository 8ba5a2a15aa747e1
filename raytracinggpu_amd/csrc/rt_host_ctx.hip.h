// rt_host_ctx.hip.h -- host side of libraytrace_hip.so, part 1 of 4 (textually included by rt_capi.hip, one translation unit): the context (rt_ctx), its knobs
// (environment variables read once per context), device buffers, error reporting, the stream a call runs on when the caller passes none.
#pragma once

namespace {

thread_local std::string g_last_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
};

}  // namespace

// Tuning / test knobs: environment variables read ONCE, when the context is created (rt_ctx_create).  Defaults are the
// measured optimum on MI355X; tests create a context under a modified environment to reach the rare code paths.
struct Knobs {
    int travq_R = 64;          // RT_TRAVQ_R: ray slots per wave of the work-stack kernel (32 | 64).  (128 -- two slots per lane, two sibling pairs per lane and BOX step -- measured 19 % slower
                               // per frame: 10 waves per CU instead of 16, profiles/round4/ab_128_rays_per_wave.txt; its instantiations left the library in round 5, the kernel source still
                               // carries the two-bank form behind R > 64)
    int travq_cap = 0;         // RT_TRAVQ_CAP: stack capacity (>= 128; tests force the serial drain); 0 = the carve's capacity
    int travq_lds = 0;         // RT_TRAVQ_LDS: waves of the ONE workgroup per CU that stages the top of the BVH in LDS; 0 = nodes through L1/L2
    int q_low = 48;            // RT_TRAVQ_LOW: refill while the stack holds fewer sibling pairs than this (measured 1.19 / 1.21 / 1.25 ms per frame for 48 / 64 / 96)
    int q16 = -1;              // RT_TRAVQ_Q16: the BOX step reads 16-bit fixed-point sibling pairs (32 bytes: two loads instead of four; rt_qnodes.hip.h) when the tree allows it.
                               // -1 (default) = for trees of at least kQ16AutoNodes nodes, 0 = never, 1 = always.  Bit-exact either way; on the cat (2 019 nodes, L1-resident) it measures
                               // +-0 once every leaf decision is exact (profiles/round4/ab_fixed_point_pairs.txt), on 55 000 / 533 000 nodes -8 % / -18 % per frame (big_mesh_bench.txt)
    int qw = -1;               // RT_TRAVQ_QW: the BOX step is four boxes wide (fixed-point quads: the children of both nodes of a sibling pair in 64 bytes, every other level of the tree
                               // skipped; leaves flagged and decided as the fixed-point pairs decide them; rt_travq.hip.h, QW).  -1 (default) = 1 = on where the tree allows the format (boxes nest, leaves of
                               // at most 127 triangles, fewer than 2^21 nodes), 0 = off.  Bit-exact either way; cat 1920x1080: 0.934 -> 0.861 ms per frame (profiles/round5/ab_wide_nodes.txt)
    int quad_sel = 1;          // RT_TRAVQ_QSEL=0: the quads of the 4-wide step take every other level of the tree (A/B; default: the four nodes a surface-area DP picks, rt_qnodes.hip.h)
    int anyhit = 1;            // RT_TRAVQ_ANYHIT=0: shadow rays are traced to the end like every other ray (A/B, cross-check; default: the fixed-point traversal kernels stop a shadow ray at the
                               // first accepted triangle that certainly shades, rt_wavefront.hip.h wf_anyhit_bound).  Bit-exact either way
    int auto_lockstep = 1;     // RT_AUTO_LOCKSTEP=0: RT_VARIANT_AUTO stays the wavefront pipeline for scenes without a mesh (A/B; default: the lock-step kernel renders them)
    int qw_count = 0;          // RT_TRAVQ_QW_COUNT=1: rt_count_work runs the 4-wide kernel's counting instantiation (its own step counters; the box / node counts then describe
                               // THAT kernel, not the reference's traversal)
    float lbvh_ct = 0.f;       // RT_LBVH_CT: cost of a triangle test relative to a box test in the LBVH's leaf cut (0 = kLbvhCt)
    int q_minfree = 0;         // RT_TRAVQ_MINFREE: ... and at least this many slots are free (0 = R / 4)
    int parts = 2;             // RT_PARTS: concurrent sub-frames of the wavefront pipeline
    int bpc5 = 0;              // (fixed; RT_TRAVQ_BPC5 was an environment knob until round 5) allow a fifth workgroup per CU
    int trav_waves = 0;        // RT_TRAV_WAVES: cap on traversal workgroups per CU
    int oversub = 2;           // RT_TRAVQ_OVERSUB: grid oversubscription of the work-stack kernel
    int oversub_min = 0;       // (fixed; RT_TRAVQ_OVERSUB_MIN was an environment knob until round 5)
    int min_groups = 16;       // RT_TRAV_MIN_GROUPS: ray groups per wave below which a launch uses fewer workgroups
    int log2S = -1;            // RT_TRAV_LOG2S: cap on the scramble period (experiment)
    int path_low = 96;         // (fixed; RT_PATH_LOW was an environment knob until round 5) wf_path runs a SHADE step only while the stack holds fewer sibling pairs than this
    int path_shade_min = 32;   // (fixed; RT_PATH_SHADE_MIN was an environment knob until round 5) ... and at least this many of the wave's 64 paths are ready (or nothing else is left to do)
    int path_oversub = 2;      // (fixed; RT_PATH_OVERSUB was an environment knob until round 5) grid oversubscription of wf_path
    int path_bpc = 4;          // (fixed; RT_PATH_BPC was an environment knob until round 5) workgroups (4 waves) per CU
    int path_parts = 1;        // (fixed; RT_PATH_PARTS was an environment knob until round 5) concurrent sub-frames (launches on separate streams)
    long long path_samp_bytes = 400ll << 20; // RT_PATH_SAMP_MB: state of the samples traced together (frames with num_rays > 1; ~130 B per sample and pixel slot).
                                             // Measured: a chain is fastest while its state stays near the 256 MB Infinity Cache -- 512x512, 64 samples: 23.5 / 8.8 / 8.0 /
                                             // 8.8 ms for 30 / 192 / 400 / 4096 MB; 1920x1080 (277 MB per sample): one sample per chain is best (71.5 vs 76.9 ms at 13)
    double chunk_mpx = 2.3;    // RT_CHUNK_MPX: pixels (millions) of one sequential chunk of a big frame in the wavefront pipeline; 0 = never cut
    int part_prio = 0;         // RT_PART_PRIO=1: the second sub-frame's stream in the high-priority class, which has its own pool of hardware queues.  A process
                               // that holds SEVERAL contexts (rt_multi does this itself; bench.py with N > 1) should set it: with more streams than the runtime has
                               // hardware queues (four) two active streams may share one and a context's sub-frames then run one after the other (1/8 of
                               // 7680x4320: 2.7 instead of 2.0 ms).  Off by default: a lone context is 1 % faster with both sub-frames at equal priority.
    int adv_block = 64;        // RT_ADV_BLOCK: threads per workgroup of wf_advance (64 / 128 / 256).  One-wave workgroups slip into the wave slots the traversal
                               // kernel of the other sub-frame frees one by one: 0.970 -> 0.957 ms per frame (128: 0.963; profiles/round3/ab_advance_block.log)
    int copy_prio = 1;         // (fixed; RT_COPY_PRIO was an environment knob until round 5) the copy streams of rt_render_async in the low-priority class (1, default), the normal one (0) or the high one (-1).  The
                               // runtime keeps a pool of hardware queues per class; in the normal class the copy stream can share a queue with one of the
                               // sub-frame streams and the copy then waits behind kernels (pipelined float4 frames 1.72 instead of 1.19 ms)
    int async_pipeline = 1;    // RT_ASYNC_PIPELINE=0: rt_render_async joins the sub-frames of frame k before frame k+1 starts (as rt_render_device does without rt_ctx_set_pipelining)
    int copy_split = 0;        // (fixed; RT_COPY_SPLIT=1 was an environment knob until round 5) rt_render_async sends the two halves of a big frame through two copy streams (measured SLOWER: 1.62 vs 1.48 ms per
                               // pipelined 1080p float4 frame -- one DMA already runs at the rate the PCIe link gives, two share it and add an event hop)
    int debug_trav = -2;       // RT_DEBUG_TRAV: traversal launch whose per-wave records are dumped (-DRT_DEBUG builds only)
};

static Knobs read_knobs() {
    Knobs k;
    auto geti = [](const char *name, int &out) { const char *e = getenv(name); if (e && *e) { out = atoi(e); return true; } return false; };
    int v;
    // --- knobs a caller or a test may set: which (bit-identical) kernel form runs, how a call is cut up.  INTEGRATION.md section 4 lists them.
    if (geti("RT_TRAVQ_R", v) && (v == 32 || v == 64)) k.travq_R = v;
    if (geti("RT_TRAVQ_CAP", v) && v >= 128) k.travq_cap = v;
    if (geti("RT_TRAVQ_LDS", v) && v >= 1 && v <= 16) k.travq_lds = v;
    if (geti("RT_TRAVQ_Q16", v) && v >= -1 && v <= 1) k.q16 = v;
    if (geti("RT_TRAVQ_QW", v) && v >= -1 && v <= 1) k.qw = v;
    if (geti("RT_TRAVQ_QW_COUNT", v)) k.qw_count = v != 0;
    if (geti("RT_AUTO_LOCKSTEP", v)) k.auto_lockstep = v != 0;
    if (geti("RT_TRAVQ_QSEL", v)) k.quad_sel = v != 0;
    if (geti("RT_TRAVQ_ANYHIT", v)) k.anyhit = v != 0;
    if (geti("RT_PARTS", v) && v >= 1 && v <= 8) k.parts = v;
    if (geti("RT_PART_PRIO", v)) k.part_prio = v != 0;
    { const char *e = getenv("RT_CHUNK_MPX"); if (e && *e) { const double d = atof(e); if (d >= 0 && d < 1e4) k.chunk_mpx = d; } }
    if (geti("RT_ASYNC_PIPELINE", v)) k.async_pipeline = v != 0;
    if (geti("RT_PATH_SAMP_MB", v) && v >= 1) k.path_samp_bytes = (long long)v << 20;
    // --- launch-geometry knobs of the A/B tools (tools/ab_variants.sh, share_*.py): honoured only under RT_EXPERIMENT=1, so that a stray variable in a
    //     caller's environment cannot move a product frame off its measured optimum
    if (geti("RT_EXPERIMENT", v) && v != 0) {
        if (geti("RT_TRAVQ_LOW", v) && v >= 32 && v <= 320) k.q_low = v;
        if (geti("RT_TRAVQ_MINFREE", v) && v >= 1 && v <= 64) k.q_minfree = v;
        if (const char *e = std::getenv("RT_LBVH_CT")) { const float f = (float)std::atof(e); if (f > 0.f && f < 100.f) k.lbvh_ct = f; }
        if (geti("RT_TRAV_WAVES", v) && v >= 1) k.trav_waves = v;
        if (geti("RT_TRAVQ_OVERSUB", v) && v >= 1 && v <= 16) k.oversub = v;
        if (geti("RT_TRAV_MIN_GROUPS", v) && v >= 4) k.min_groups = v;
        if (geti("RT_TRAV_LOG2S", v) && v >= 0) k.log2S = v;
        if (geti("RT_ADV_BLOCK", v) && (v == 64 || v == 128 || v == 256)) k.adv_block = v;
    }
#ifdef RT_DEBUG
    if (geti("RT_DEBUG_TRAV", v)) k.debug_trav = v;
#endif
    return k;
}

struct rt_ctx {
    int device = 0;
    Knobs knobs;
    hipStream_t stream_ = nullptr;                                   // the context's own stream: created when first needed (own_stream)
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    bool have_scene = false, have_kernel_time = false, have_tonemap_time = false;
    rtk::Scene scene{};
    DevBuf nrm;                                                     // smooth shading: 3 normals per triangle, visit order
    std::vector<int> tri_perm;                                       // visit order -> triangle index in the uploaded (BVH-order) arrays
    std::vector<int> up_indices;                                     // vertex indices of the uploaded triangles, 3 per triangle
    int n_up_tris = 0;
    DevBuf tidx_up;                                                  // the same on the device (int4 per triangle)
    DevBuf bb_idx, bb_cnt, bb_pa, bb_pb, bb_tmp, bb_nodes_i, bb_nodes_f, bb_counter, bb_lvl, bb_size, bb_pre, bb_arr;   // device BVH build scratch
    DevBuf left_dev, lvl_nodes, lvl_off;                             // tree topology for the device-side refit
    DevBuf lb_pool, lb_pool2, perm_dev;                              // LBVH builder / layout scratch (rt_lbvh.hip.h), carved pools; visit rank -> uploaded index on the device
    rtk::LbvhArgs lb_args{};                                         // the builder's arrays of the last LBVH build (valid until the next one)
    bool host_mesh_stale = false;                                    // tri_perm / up_indices describe an older layout: the device copies (perm_dev, tidx_up) are current
    int lbvh_host_install = 0;                                       // RT_LBVH_HOST_INSTALL=1: re-lay an LBVH tree out on the host, as the reference-mode rebuild does (tests compare the two)
    rt_build_stats build{};                                          // what the last rt_mesh_rebuild_mode did
    int n_levels = 0;
    DevBuf nodesh, tri2leaf;                                        // 16-bit fixed-point sibling pairs and the triangle -> leaf table (rt_qnodes.hip.h)
    DevBuf leaflh;                                                  // (lo, hi) of each triangle's leaf, by triangle (the flagged-leaf check of the fixed-point kernels)
    DevBuf nodesw, qdp_parent, qdp_cnt, qdp_g, qdp_ch;              // 4-wide fixed-point nodes (RT_TRAVQ_QW) and the scratch of the DP that picks which four nodes a quad holds (rt_qnodes.hip.h)
    int travq_blocks_per_cu_qw[2] = {0, 0};                         // [STATS]
    unsigned chain_nonce[8] = {};                                   // launch chains started so far, PER SUB-FRAME (WfState::nonce): every part owns its own region of the ray queue, so each
                                                                    // region must cycle through all four values (one context-wide counter gave a part only two of them with two parts: ADVICE round 4)
    int q16_leaf_shift = 0;                                         // where a leaf's triangle count sits in its payload word (rtk::q16_leaf_shift), 0 = leaves too large
    bool q16_topo_ok = false;                                       // the tree's shape allows them (leaf sizes, node count, boxes nest)
    bool qw_topo_ok = false;                                        // ... and no leaf is empty: places 0 and 2 of a quad must hold a node (the pairs cope with an empty leaf)
    int real_obj = -1;                                              // object position of the (first) mesh with triangles, -1 = none
    int n_real_meshes = 0;                                          // meshes WITH triangles in the scene: with more than one the tree in use is a forest (build_forest) and the per-mesh operations are refused
    DevBuf node_lo, node_hi, nodes2, nodesq, nodesb, q2thr, tri, verts, tidx, scratch_rgba, scratch_rgb8, work, queue;
    int n_cus = 0;
    DevBuf wfM, wfT, wfLS, wfSID, wfSamp;                     // wavefront path state (HBM); wfSamp / wfT: per-sample colours and their running sum (num_rays > 1)
    DevBuf wfQR;                                                    // traversal queue in slot order: the rays (32 B each)
    DevBuf pathSamp, pathT;                                         // wf_path with num_rays > 1: per-sample colours, running sum
    DevBuf dbgbuf;                                                  // -DRT_DEBUG builds: per-wave traversal records
    DevBuf batch_dev;                                               // rt_render_device_batch: the frames' descriptors, one copy per sub-frame
    DevBuf accum;                                                   // progressive mode: sum of the frames so far (float4 per pixel)
    int prog_frames = 0, prog_w = 0, prog_h = 0;
    uint64_t qf_sig = 0;                                            // layout the queue flags were last zeroed for
    int trav_blocks_per_cu[4] = {0, 0, 0, 0};
    int travq_blocks_per_cu[6] = {0, 0, 0, 0, 0, 0};   // [STATS + 2 * (R == 32) + 4 * (R == 128)]
    static constexpr int kMaxParts = 8;
    hipStream_t part_stream[kMaxParts] = {};
    hipEvent_t part_ev[kMaxParts] = {};
    hipEvent_t fork_ev = nullptr;
    // Chains on streams of their own (launch_render_chunk): consecutive chunks of one call -- and, with rt_ctx_set_pipelining, consecutive
    // frames into different buffers -- follow each other per sub-frame without a join in between.
    struct Pipe {
        bool on = false;                   // rt_ctx_set_pipelining: frames of consecutive calls may overlap
        hipEvent_t fork2[2] = {};          // the caller's stream at the start of this call / of the previous one
        int cur = 0;
        bool valid = false, prev_valid = false;   // the call before this one ended with its chains on their own streams (joined into `stream`)
        int open_parts = 0;                // chains of an earlier chunk of THIS call that have not been joined into the caller's stream
        hipStream_t stream = nullptr;      // ... and was issued on this stream,
        uint64_t sig = 0;                  // ... with this state layout (sub-frames, sizes, offsets) in its last chunk,
        const uint8_t *out_lo = nullptr, *out_hi = nullptr;   // ... into this output range
        int call_chunk = 0, call_chunks = 1;                  // position of the chunk being issued in its call (launch_render)
        const uint8_t *call_lo = nullptr, *call_hi = nullptr; // output range of the call being issued
        hipEvent_t extra_wait = nullptr;   // this call's chains also wait for this event (rt_render_async: the slot's previous copy)
        // What the library ITSELF put on the caller's stream since the previous render call and what that work touches (rt_tonemap_device:
        // reads a frame, writes an image).  A frame that starts behind the PREVIOUS call does not wait for it, so a frame whose output overlaps
        // one of these ranges must not take the relaxed start: the product falls back to the full fork, a -DRT_DEBUG build refuses the call
        // with RT_ERR_INVALID so that the caller learns its sequence breaks the rule of rt_ctx_set_pipelining.  (Work the caller submits
        // through HIP directly is invisible to the library: no run-time check can cover it.)
        struct Range { const uint8_t *lo, *hi; hipStream_t stream; };
        std::vector<Range> between;
        bool between_overflow = false;     // more than 64 ranges came in between two render calls: treated as a hazard
    } pipe;
    bool trav_attr_set = false;
    bool stats_on = false;                                          // rt_stats_enable: bracket the traversal launches with timing events (production frames record none)
    static constexpr int kSlots = 2;                                // rt_render_async: double-buffered device frames, one copy stream
    DevBuf slot_rgba[kSlots], slot_rgb8[kSlots];
    hipEvent_t slot_rendered[kSlots] = {}, slot_done[kSlots] = {};
    bool slot_pending[kSlots] = {false, false};
    hipStream_t copy_stream = nullptr, copy_stream2 = nullptr;      // two copy streams: the halves of a frame go out through two DMA engines
    hipEvent_t slot_half[kSlots] = {};
    bool travq_ok = true;                                           // the uploaded tree fits wf_travq's entry formats (leaf sizes, triangle offsets)
    static constexpr int kMaxTravEvents = 2 * RT_MAX_SEGMENTS;
    hipEvent_t ev_trav[2 * kMaxTravEvents] = {};
    hipEvent_t ev_adv[2 * kMaxTravEvents] = {};                     // ... and of the uniform kernel's launches (rt_stats_enable)
    int n_trav_events = 0, n_adv_events = 0, adv_paths = 0;
    int persist_blocks_per_cu[2] = {0, 0};   // [STATS]
    rt_stats stats{};
    std::string err;
    char name[256] = {0};
};

namespace {

// The stream a call runs on when the caller passes none.  Created on first use: a context driven on the caller's streams (bench.py,
// rt_render_device with a stream) owns no stream of its own -- idle streams still take part in the runtime's mapping of streams onto
// its few hardware queues.
hipStream_t own_stream(rt_ctx *ctx) {
    if (!ctx->stream_) {
        (void)hipSetDevice(ctx->device);
        const hipError_t e = hipStreamCreateWithFlags(&ctx->stream_, hipStreamNonBlocking);
        if (e != hipSuccess) { ctx->stream_ = nullptr; (void)hipGetLastError(); ctx->err = std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e); g_last_error = ctx->err; }
    }
    return ctx->stream_;   // nullptr: the entry points fail with RT_ERR_HIP (RT_OWN_STREAM) rather than fall back to the legacy default stream
}

// RT_TIMING=1: host-side wall time of the library's start-up phases on stderr (rt_launcher --timing 1 sets it): where a short program's time goes
// -- runtime initialisation, the first launch's code-object load, uploads, the frame itself, the copy back.
struct PhaseClock {
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseClock() : on([] { const char *e = getenv("RT_TIMING"); return e && *e && atoi(e) != 0; }()), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "timing: %-44s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

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

// Entry points that run on the context's own stream: its creation must have succeeded (a failure used to fall back to
// stream 0 without a word).
#define RT_OWN_STREAM(ctx)                                                                              \
    do {                                                                                                \
        if (!own_stream(ctx)) return fail(ctx, RT_ERR_HIP, "the context's stream: %s", (ctx)->err.c_str()); \
    } while (0)

// Every allocation happens on the context's device, whatever the calling thread's current device is.
int ensure(rt_ctx *ctx, DevBuf &b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    b.release();
    RT_HIP(ctx, hipMalloc(&b.p, bytes ? bytes : 16));
    b.bytes = bytes ? bytes : 16;
    return RT_OK;
}

// Host -> device, complete on return.  (hipMemcpy = the NULL stream.  Round 6 tried the context's own stream instead, to spare a C++ program one hardware queue: the ~27 ms the
// FIRST copy of a process costs -- the runtime creating a queue and its staging -- just moved to that stream, and the headline frame read 1 % slower:
// profiles/round6/launcher_timing.txt.)
int upload(rt_ctx *ctx, DevBuf &b, const void *src, size_t bytes) {
    int rc = ensure(ctx, b, bytes);
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) RT_HIP(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}


}  // namespace
