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

// Host-side Vector arithmetic for the triangle precompute (cpu:227-229).  This TU is
// compiled with -ffp-contract=off, so these are the same single roundings as on the device.
struct h3 { float x, y, z; };
inline h3 hsub(h3 a, h3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline h3 hcross(h3 a, h3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// Converts the reference's bvhTreeToArray layout (optimized.cu:512-534) into traversal order.
// The reference pops the right child first (cpu:291-292 push left then right), so the
// pre-order here descends right before left.
int build_threaded(rt_ctx *ctx, const rt_mesh *m, std::vector<float4> &lo, std::vector<float4> &hi, std::vector<int> &perm,
                   std::vector<int> &left_of) {
    const int n = m->n_nodes;
    perm.clear();
    left_of.assign(n, -1);                                        // internal nodes: traversal-order index of the LEFT child
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
                // triangles are re-stored in VISIT order (leaves as the traversal reaches them, ascending inside a
                // leaf, cpu:295), so a triangle's index is its rank in the reference's scan: the strict '<' of
                // cpu:301 keeps, among equal t, the smallest index -- which is what lets sub-ranges of one ray
                // be traversed independently and merged by min over (t, index)
                const int first = (int)perm.size();
                for (int q = ts; q < te; ++q) perm.push_back(q);
                lo[it.out].w = __builtin_bit_cast(float, first);
                hi[it.out].w = __builtin_bit_cast(float, (int)perm.size());
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
            left_of[it.out] = emitted;                               // the left subtree starts right behind the right one
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

// The part of rt_scene_upload after validation of the sphere / light / camera arguments: layout conversion of the mesh (the
// reference's arrays -> traversal-order nodes, visit-order triangle records, breadth-first sibling pairs, refit levels) and
// the uploads.  `sc` carries the spheres, light and camera; rt_mesh_rebuild re-enters here with the tree it built on the device.
constexpr int kQ16AutoNodes = 16384;                                 // RT_TRAVQ_Q16 = -1: from this many nodes on (the node array no longer sits in the L1s)
// (Re)derive the 16-bit fixed-point sibling pairs and the triangle -> leaf table from the breadth-first arrays on the device (rt_qnodes.hip.h), on stream q
// (the upload passes the null stream, as its copies do: creating the context's own stream here would change which hardware queues the sub-frame streams
// get later, profiles/round3/ab_hw_queues_parts.log), joined before returning.  ctx->scene must be final (root box, node arrays); trees the format does not fit keep scene.nodesh = nullptr.
int requantize(rt_ctx *ctx, hipStream_t q) {
    rtk::Scene &sc = ctx->scene;
    sc.nodesh = nullptr; sc.tri2leaf = nullptr; sc.nodesw = nullptr; sc.leaflh = nullptr;
    // (wherever the format fits: with flagged leaves the 4-wide step beats the fixed-point pairs on every tree measured -- 2 019 nodes -8 %, 32 889 -9 %, 358 503 -12 %: profiles/round5/ab_wide_nodes.txt)
    const bool want_qw = ctx->knobs.qw != 0 && ctx->qw_topo_ok && ctx->q16_leaf_shift == 24 && sc.n_nodes + 2 < (1 << 21);   // the quad's payload word: leaves of <= 127 triangles, child << 10 positive; no empty leaf
    if (!(ctx->knobs.q16 == 1 || (ctx->knobs.q16 < 0 && sc.n_nodes >= kQ16AutoNodes) || want_qw) || !ctx->q16_topo_ok || ctx->q16_leaf_shift == 0 || !ctx->travq_ok || !sc.fast_box || sc.mesh_slot < 0 || sc.n_nodes < 3 || sc.n_tris <= 0) return RT_OK;
    int rc;
    if ((rc = ensure(ctx, ctx->nodesh, ((size_t)sc.n_nodes + 2) * 16)) != RT_OK || (rc = ensure(ctx, ctx->tri2leaf, (size_t)sc.n_tris * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->leaflh, (size_t)sc.n_tris * 32)) != RT_OK) return rc;
    RT_HIP(ctx, hipMemsetAsync(ctx->tri2leaf.p, 0, (size_t)sc.n_tris * sizeof(int), q));
    const rtk::QGrid g = rtk::q16_grid(sc.root_lo, sc.root_hi);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemsetAsync(ctx->nodesh.p, 0, 32, q));            // nodes 0 (padding) and 1 (the root: tested when a ray is emitted)
    if (want_qw && (rc = ensure(ctx, ctx->nodesw, ((size_t)sc.n_nodes + 4) * 32)) != RT_OK) return rc;
    hipLaunchKernelGGL(rtk::qnodes_kernel, dim3((unsigned)((sc.n_nodes + 255) / 256)), dim3(256), 0, q, sc.nodesq, sc.nodesb, sc.n_nodes, g,
                       static_cast<uint4 *>(ctx->nodesh.p), static_cast<int *>(ctx->tri2leaf.p), sc.n_tris, rtk::kQLeafShift, ctx->q16_leaf_shift);
    hipLaunchKernelGGL(rtk::leaflh_kernel, dim3((unsigned)((sc.n_tris + 255) / 256)), dim3(256), 0, q, sc.nodesq, static_cast<const int *>(ctx->tri2leaf.p), sc.n_tris, sc.n_nodes,
                       static_cast<float4 *>(ctx->leaflh.p));
    if (want_qw) {
        const bool dp = ctx->knobs.quad_sel != 0;
        if (dp) {
            const size_t nn = (size_t)sc.n_nodes + 2;
            if ((rc = ensure(ctx, ctx->qdp_parent, nn * 4)) != RT_OK || (rc = ensure(ctx, ctx->qdp_cnt, nn * 4)) != RT_OK || (rc = ensure(ctx, ctx->qdp_g, nn * 16)) != RT_OK ||
                (rc = ensure(ctx, ctx->qdp_ch, nn * 4)) != RT_OK) return rc;
            rtk::QdpArgs a{};
            a.nodesh = static_cast<const uint4 *>(ctx->nodesh.p); a.n_bfs = sc.n_nodes; a.node_shift = rtk::kQNodeShift;
            a.sx = g.sx; a.sy = g.sy; a.sz = g.sz;
            a.parent = static_cast<int *>(ctx->qdp_parent.p); a.cnt = static_cast<int *>(ctx->qdp_cnt.p);
            a.g = static_cast<float4 *>(ctx->qdp_g.p); a.ch = static_cast<uchar4 *>(ctx->qdp_ch.p);
            RT_HIP(ctx, hipMemsetAsync(ctx->qdp_ch.p, 0, nn * 4, q));
            const dim3 grid((unsigned)((sc.n_nodes + 255) / 256));
            hipLaunchKernelGGL(rtk::qdp_init_kernel, grid, dim3(256), 0, q, a);
            hipLaunchKernelGGL(rtk::qdp_up_kernel, grid, dim3(256), 0, q, a);
        }
        hipLaunchKernelGGL(rtk::qquads_kernel, dim3((unsigned)((sc.n_nodes / 2 + 1 + 255) / 256)), dim3(256), 0, q, static_cast<const uint4 *>(ctx->nodesh.p), sc.n_nodes,
                           rtk::kQNodeShift, ctx->q16_leaf_shift, dp ? static_cast<const int *>(ctx->qdp_parent.p) : nullptr, dp ? static_cast<const uchar4 *>(ctx->qdp_ch.p) : nullptr,
                           static_cast<uint4 *>(ctx->nodesw.p));
    }
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipStreamSynchronize(q));
    // the DP's scratch (28 bytes per node) is needed while this function runs only: big trees give it back (a cat-sized one keeps it for the next refit)
    if ((size_t)sc.n_nodes * 28 > (16u << 20)) { ctx->qdp_parent.release(); ctx->qdp_cnt.release(); ctx->qdp_g.release(); ctx->qdp_ch.release(); }
    sc.nodesh = static_cast<const uint4 *>(ctx->nodesh.p); sc.tri2leaf = static_cast<const int *>(ctx->tri2leaf.p); sc.leaflh = static_cast<const float4 *>(ctx->leaflh.p);
    if (want_qw) sc.nodesw = static_cast<const uint4 *>(ctx->nodesw.p);
    sc.qgx = g.gx; sc.qgy = g.gy; sc.qgz = g.gz; sc.qsx = g.sx; sc.qsy = g.sy; sc.qsz = g.sz; sc.qleaf_shift = ctx->q16_leaf_shift;
    return RT_OK;
}

// the triangle ranges of the scene's mesh table when at most ONE mesh has triangles (object position real_obj): a mesh without triangles is an empty range at its place in the order
void mesh_table_single(rtk::Scene &sc, int real_obj) {
    for (int k = 0; k < sc.n_meshes; ++k) sc.mesh[k].tri_begin = sc.mesh[k].obj <= real_obj ? 0 : sc.n_tris;
}

// sc: spheres (with their object ids), light, camera and the mesh table (object ids, materials; sc.mesh_slot = the first mesh object's position or -1) filled in by the caller.
// mesh: the geometry to traverse -- one TriangleMesh as uploaded, or the forest build_forest made of several (tri_offsets[k] = first triangle of table entry k in mesh->indices,
// n_meshes + 1 entries) -- or nullptr.
int install_scene(rt_ctx *ctx, rtk::Scene sc, const rt_mesh *mesh, const std::vector<int> *tri_offsets = nullptr) {
    PhaseClock pc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->stream_) RT_HIP(ctx, hipStreamSynchronize(ctx->stream_));   // (renders issued on a caller's stream are the caller's to order)
    ctx->have_scene = false;
    ctx->host_mesh_stale = false;                                     // what follows rewrites tri_perm / up_indices
    ctx->tri_perm.clear();
    std::vector<float4> lo, hi, tri, verts;
    std::vector<int4> tidx;
    std::vector<int> left_of;
    if (mesh) {
        if (mesh->object_slot < 0 || mesh->object_slot >= RT_MAX_OBJECTS)   // (validated against the scene's objects by rt_scene_upload_meshes)
            return fail(ctx, RT_ERR_INVALID, "mesh object_slot %d outside [0,%d)", mesh->object_slot, RT_MAX_OBJECTS);
        if (mesh->n_vertices < 0 || mesh->n_triangles < 0 || mesh->n_nodes < 0 || mesh->index_stride < 3)
            return fail(ctx, RT_ERR_INVALID, "bad mesh sizes");
        if ((mesh->n_vertices && !mesh->vertices) || (mesh->n_triangles && !mesh->indices) || (mesh->n_nodes && !mesh->bvh_arr10))
            return fail(ctx, RT_ERR_INVALID, "mesh array pointer is NULL");
        if (mesh->n_nodes >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "node indices are stored as floats: < 2^24 nodes");
        std::vector<int> perm;
        int rc = build_threaded(ctx, mesh, lo, hi, perm, left_of);
        if (rc != RT_OK) return rc;
        if (perm.size() >= ((size_t)1 << 31)) return fail(ctx, RT_ERR_INVALID, "too many leaf triangles");
        for (int t = 0; t < mesh->n_triangles; ++t) {
            const int32_t *ix = mesh->indices + (size_t)t * mesh->index_stride;
            for (int k = 0; k < 3; ++k)
                if (ix[k] < 0 || ix[k] >= mesh->n_vertices)
                    return fail(ctx, RT_ERR_INVALID, "triangle %d references vertex %d outside [0,%d)", t, ix[k], mesh->n_vertices);
        }
        const int n_int = (int)perm.size();
        ctx->tri_perm = perm;
        ctx->up_indices.resize((size_t)mesh->n_triangles * 3);              // the mesh as uploaded (BVH order): rt_mesh_rebuild starts from it
        std::vector<int4> tup(mesh->n_triangles);
        for (int t = 0; t < mesh->n_triangles; ++t) {
            const int32_t *ix = mesh->indices + (size_t)t * mesh->index_stride;
            for (int k = 0; k < 3; ++k) ctx->up_indices[3 * (size_t)t + k] = ix[k];
            tup[t] = make_int4(ix[0], ix[1], ix[2], 0);
        }
        ctx->n_up_tris = mesh->n_triangles;
        if (int rcu = upload(ctx, ctx->tidx_up, tup.data(), tup.size() * sizeof(int4)); rcu != RT_OK) return rcu;
        tri.resize((size_t)n_int * 3);
        tidx.resize(n_int);
        for (int t = 0; t < n_int; ++t) {
            const int32_t *ix = mesh->indices + (size_t)perm[t] * mesh->index_stride;
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
        sc.n_nodes = n_int > 0 ? mesh->n_nodes : 0;
        sc.n_tris = n_int;
        if (sc.n_nodes > 0) { sc.root_lo = lo[0]; sc.root_hi = hi[0]; }
        sc.n_verts = mesh->n_vertices;
        if (tri_offsets) {
            // the forest is laid out so that the traversal reaches the meshes in object order (build_forest): the visit-order triangle array is mesh after mesh
            int cur = 0;
            for (int k = 0; k < sc.n_meshes; ++k) sc.mesh[k].tri_begin = -1;
            for (int t = 0; t < n_int; ++t) {
                while (cur + 1 < sc.n_meshes && perm[t] >= (*tri_offsets)[cur + 1]) ++cur;
                if (perm[t] < (*tri_offsets)[cur]) return fail(ctx, RT_ERR_INTERNAL, "forest layout: triangle %d of an earlier mesh is visited after a later mesh's", perm[t]);
                if (sc.mesh[cur].tri_begin < 0) sc.mesh[cur].tri_begin = t;
            }
            int next = n_int;
            for (int k = sc.n_meshes - 1; k >= 0; --k) { if (sc.mesh[k].tri_begin < 0) sc.mesh[k].tri_begin = next; next = sc.mesh[k].tri_begin; }
        } else {
            mesh_table_single(sc, mesh->object_slot);
        }
    } else {
        mesh_table_single(sc, -1);
    }
    int rc;
    pc.lap("  scene: traversal order, triangle records (host) + the FIRST hipMalloc / copy of the process (runtime: stream = hardware queue, staging)");
    if ((rc = upload(ctx, ctx->node_lo, lo.data(), lo.size() * sizeof(float4))) != RT_OK) return rc;
    pc.lap("  scene: one more hipMalloc + copy");
    if ((rc = upload(ctx, ctx->node_hi, hi.data(), hi.size() * sizeof(float4))) != RT_OK) return rc;
    std::vector<float4> inter(lo.size() * 2);
    for (size_t k = 0; k < lo.size(); ++k) { inter[2 * k] = lo[k]; inter[2 * k + 1] = hi[k]; }
    if ((rc = upload(ctx, ctx->nodes2, inter.data(), inter.size() * sizeof(float4))) != RT_OK) return rc;
    {   // work-stack layout: breadth-first order (the top of the tree is a prefix: LDS staging), children adjacent
        const size_t n = lo.size();
        std::vector<int> order;                                      // order[k] = traversal-order index of breadth-first node k
        std::vector<int> bfs_of(n, -1);
        order.reserve(n);
        if (n) { order.push_back(0); bfs_of[0] = 0; }
        for (size_t k = 0; k < order.size(); ++k) {
            const int x = order[k];
            if (left_of[x] >= 0) {                                    // internal: right child x + 1, left child left_of[x]
                bfs_of[x + 1] = (int)order.size(); order.push_back(x + 1);
                bfs_of[left_of[x]] = (int)order.size(); order.push_back(left_of[x]);
            }
        }
        // index 0 is padding, the root is node 1, so that every sibling pair (2m, 2m + 1) is one aligned 64-byte line
        std::vector<float4> q(2 * (order.size() + 1), make_float4(0, 0, 0, 0));
        std::vector<int> q2t(order.size() + 1, 0);
        for (size_t k = 0; k < order.size(); ++k) {
            const int x = order[k];
            q[2 * (k + 1)] = lo[x]; q[2 * (k + 1) + 1] = hi[x];
            if (left_of[x] >= 0) q[2 * (k + 1)].w = __builtin_bit_cast(float, bfs_of[x + 1] + 1);
            q2t[k + 1] = x;
        }
        if ((rc = upload(ctx, ctx->nodesq, q.data(), q.size() * sizeof(float4))) != RT_OK) return rc;
        // wf_travq's form of the same array (rt_travq.hip.h): box as centre / half extent, payload and kind pre-shifted the way stack
        // and leaf-queue entries carry them; and the scene-wide quantities its box filter needs
        std::vector<float4> qb(q.size(), make_float4(0, 0, 0, 0));
        float bm[3] = {0.f, 0.f, 0.f};
        bool fast = true, travq_ok = true;
        for (size_t k = 0; k < order.size(); ++k) {
            const int x = order[k];
            const float4 l = lo[x], h = hi[x];
            float4 cb = make_float4(rtk::box_centre(l.x, h.x), rtk::box_centre(l.y, h.y), rtk::box_centre(l.z, h.z), 0.f);
            float4 hb = make_float4(rtk::box_half(l.x, h.x), rtk::box_half(l.y, h.y), rtk::box_half(l.z, h.z), 0.f);
            const float v[6] = {l.x, l.y, l.z, h.x, h.y, h.z};
            for (int a = 0; a < 3; ++a) {
                if (!(v[a] <= v[a + 3]) || !(std::fabs(v[a]) < 1e8f) || !(std::fabs(v[a + 3]) < 1e8f)) fast = false;   // also false for NaN
                bm[a] = std::max(bm[a], std::max(std::fabs(v[a]), std::fabs(v[a + 3])));
            }
            if (left_of[x] >= 0) {
                cb.w = __builtin_bit_cast(float, (uint32_t)(bfs_of[x + 1] + 1) << rtk::kQNodeShift);
                hb.w = __builtin_bit_cast(float, (int)0x80000000);
            } else {
                const int first = __builtin_bit_cast(int, l.w), cnt = __builtin_bit_cast(int, h.w) - first;
                if (cnt >= rtk::kQMaxLeaf) travq_ok = false;
                cb.w = __builtin_bit_cast(float, first);
                hb.w = __builtin_bit_cast(float, cnt > 0 && cnt < rtk::kQMaxLeaf ? cnt << rtk::kQLeafShift : 0);
            }
            qb[2 * (k + 1)] = cb; qb[2 * (k + 1) + 1] = hb;
        }
        if ((rc = upload(ctx, ctx->nodesb, qb.data(), qb.size() * sizeof(float4))) != RT_OK) return rc;
        {   // may this tree use the 16-bit fixed-point pairs (rt_qnodes.hip.h)?  Leaf sizes and counts fit the payload word, and every box nests inside its parent's
            bool topo = order.size() >= 3, empty_leaf = false;
            int max_leaf = 0;
            for (size_t x = 0; topo && x < n; ++x) {
                if (left_of[x] < 0) {
                    const int cnt = __builtin_bit_cast(int, hi[x].w) - __builtin_bit_cast(int, lo[x].w);
                    max_leaf = std::max(max_leaf, cnt);
                    if (cnt <= 0) empty_leaf = true;                  // the quads' places 0 and 2 must hold a node; the fixed-point and the float pairs cope with an empty leaf
                    continue;
                }
                for (const int c : {(int)x + 1, left_of[x]}) {
                    const float4 cl = lo[c], ch = hi[c], pl = lo[x], ph = hi[x];
                    if (!(cl.x >= pl.x && cl.y >= pl.y && cl.z >= pl.z && ch.x <= ph.x && ch.y <= ph.y && ch.z <= ph.z)) topo = false;   // also false for NaN
                }
            }
            ctx->q16_topo_ok = topo;
            ctx->qw_topo_ok = topo && !empty_leaf;
            ctx->q16_leaf_shift = rtk::q16_leaf_shift(max_leaf, (long long)(tri.size() / 3));
        }
        sc.bmx = bm[0]; sc.bmy = bm[1]; sc.bmz = bm[2];
        sc.fast_box = fast ? 1 : 0;
        ctx->travq_ok = travq_ok && (uint64_t)tri.size() * 16 < ((uint64_t)1 << 32);   // 32-bit byte offsets into the triangle records
        if ((rc = upload(ctx, ctx->q2thr, q2t.data(), q2t.size() * sizeof(int))) != RT_OK) return rc;
        // levels of the tree (pre-order indices sorted by depth) for the device-side refit (rt_mesh_transform)
        std::vector<int> depth(n, 0), lvl_off, lvl_nodes(n);
        int maxd = 0;
        for (size_t x = 0; x < n; ++x)
            if (left_of[x] >= 0) { depth[x + 1] = depth[left_of[x]] = depth[x] + 1; maxd = std::max(maxd, depth[x] + 1); }
        lvl_off.assign(maxd + 2, 0);
        for (size_t x = 0; x < n; ++x) lvl_off[depth[x] + 1]++;
        for (int d = 0; d <= maxd; ++d) lvl_off[d + 1] += lvl_off[d];
        std::vector<int> fill(lvl_off.begin(), lvl_off.end() - 1);
        for (size_t x = 0; x < n; ++x) lvl_nodes[fill[depth[x]]++] = (int)x;
        ctx->n_levels = n ? maxd + 1 : 0;
        if ((rc = upload(ctx, ctx->left_dev, left_of.data(), left_of.size() * sizeof(int))) != RT_OK) return rc;
        if ((rc = upload(ctx, ctx->lvl_nodes, lvl_nodes.data(), lvl_nodes.size() * sizeof(int))) != RT_OK) return rc;
        if ((rc = upload(ctx, ctx->lvl_off, lvl_off.data(), lvl_off.size() * sizeof(int))) != RT_OK) return rc;
    }
    if ((rc = upload(ctx, ctx->tri, tri.data(), tri.size() * sizeof(float4))) != RT_OK) return rc;
    if ((rc = upload(ctx, ctx->verts, verts.data(), verts.size() * sizeof(float4))) != RT_OK) return rc;
    if ((rc = upload(ctx, ctx->tidx, tidx.data(), tidx.size() * sizeof(int4))) != RT_OK) return rc;
    sc.node_lo = static_cast<const float4 *>(ctx->node_lo.p);
    sc.node_hi = static_cast<const float4 *>(ctx->node_hi.p);
    sc.nodes = static_cast<const float4 *>(ctx->nodes2.p);
    sc.nodesq = static_cast<const float4 *>(ctx->nodesq.p);
    sc.nodesb = static_cast<const float4 *>(ctx->nodesb.p);
    sc.q2thr = static_cast<const int *>(ctx->q2thr.p);
    sc.tri = static_cast<const float4 *>(ctx->tri.p);
    sc.verts = static_cast<const float4 *>(ctx->verts.p);
    sc.tidx = static_cast<const int4 *>(ctx->tidx.p);
    ctx->scene = sc;
    ctx->have_scene = true;
    pc.lap("  scene: layouts (host) + the other hipMallocs and copies");
    rc = requantize(ctx, nullptr);
    pc.lap("  scene: fixed-point nodes on the device (first kernel launch: code object load)");
    return rc;
}

// tri_perm / up_indices (host copies of the mesh's orders) after a device-side install: fetched when a host-side path needs them
int refresh_host_mesh(rt_ctx *ctx) {
    if (!ctx->host_mesh_stale) return RT_OK;
    const size_t nt = (size_t)ctx->n_up_tris;
    std::vector<int4> up(nt);
    ctx->tri_perm.resize(nt);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemcpy(up.data(), ctx->tidx_up.p, nt * sizeof(int4), hipMemcpyDeviceToHost));
    RT_HIP(ctx, hipMemcpy(ctx->tri_perm.data(), ctx->perm_dev.p, nt * sizeof(int), hipMemcpyDeviceToHost));
    ctx->up_indices.resize(nt * 3);
    for (size_t t = 0; t < nt; ++t) { ctx->up_indices[3 * t] = up[t].x; ctx->up_indices[3 * t + 1] = up[t].y; ctx->up_indices[3 * t + 2] = up[t].z; }
    ctx->host_mesh_stale = false;
    return RT_OK;
}

// Several TriangleMesh objects in one scene (cpu:538-564): ONE tree for the traversal kernels.  Every mesh keeps the tree its own buildBVH made; the roots hang below synthetic
// internal nodes whose boxes are the unions of their children (exact: min / max of floats).  What this preserves:
//   * a mesh's triangles are tested iff the reference's own walk of that mesh reaches their leaf: the reference enters a mesh iff its root box is hit (cpu:279) and the
//     synthetic nodes above a root are entered whenever any root below them is -- BoundingBox::intersect is monotone along nested boxes: per axis the two plane parameters of
//     the larger box bracket the smaller box's (one rounding each of a monotone expression), an axis with u = 0 constrains neither box or both alike (the origin lies strictly inside
//     both intervals or the smaller box is missed), and a NaN on the first axis makes the smaller box a miss already;
//   * the synthetic tree is shaped so that the traversal order (right child first, cpu:291-292) reaches the meshes in OBJECT order, hence the visit-order triangle array holds them
//     mesh after mesh and min over (t, triangle index) = min over (t, object position, scan rank): the winner of the reference's loop over the objects with its strict '<' (cpu:554).
// real[k]: index into `meshes` of the k-th mesh with triangles, in object order.  Fills the combined arrays and f.m (which points into them).
struct Forest {
    std::vector<float> verts, arr;
    std::vector<int32_t> idx;
    std::vector<int> tri_off;                                           // per real mesh: first triangle in idx (+ the total at the end)
    rt_mesh m{};
};
int build_forest(rt_ctx *ctx, const rt_mesh *meshes, const std::vector<int> &real, Forest &f) {
    const int K = (int)real.size();
    std::vector<int> voff(K + 1, 0), noff(K + 1, 0);
    f.tri_off.assign(K + 1, 0);
    int64_t nv = 0, nt = 0, nn = K - 1;                                 // K - 1 synthetic nodes come first (node 0 = the forest's root)
    for (int k = 0; k < K; ++k) {
        const rt_mesh &m = meshes[real[k]];
        if (m.n_vertices < 0 || m.n_triangles < 0 || m.n_nodes < 0 || m.index_stride < 3) return fail(ctx, RT_ERR_INVALID, "mesh %d: bad sizes", real[k]);
        if ((m.n_vertices && !m.vertices) || (m.n_triangles && !m.indices) || (m.n_nodes && !m.bvh_arr10)) return fail(ctx, RT_ERR_INVALID, "mesh %d: array pointer is NULL", real[k]);
        voff[k] = (int)nv; f.tri_off[k] = (int)nt; noff[k] = (int)nn;
        nv += m.n_vertices; nt += m.n_triangles; nn += m.n_nodes;
        if (nv >= ((int64_t)1 << 31) || nt >= ((int64_t)1 << 31) || nn >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "the meshes together are too large (2^31 vertices / triangles, 2^24 nodes)");
    }
    voff[K] = (int)nv; f.tri_off[K] = (int)nt; noff[K] = (int)nn;
    f.verts.resize((size_t)nv * 3); f.idx.resize((size_t)nt * 3); f.arr.assign((size_t)nn * 10, 0.f);
    for (int k = 0; k < K; ++k) {
        const rt_mesh &m = meshes[real[k]];
        std::copy(m.vertices, m.vertices + (size_t)m.n_vertices * 3, f.verts.begin() + (size_t)voff[k] * 3);
        for (int t = 0; t < m.n_triangles; ++t)
            for (int c = 0; c < 3; ++c) {
                const int32_t v = m.indices[(size_t)t * m.index_stride + c];
                if (v < 0 || v >= m.n_vertices) return fail(ctx, RT_ERR_INVALID, "mesh %d: triangle %d references vertex %d outside [0,%d)", real[k], t, v, m.n_vertices);
                f.idx[3 * ((size_t)f.tri_off[k] + t) + c] = v + voff[k];
            }
        for (int n = 0; n < m.n_nodes; ++n) {
            const float *a = m.bvh_arr10 + (size_t)n * 10;
            float *o = f.arr.data() + ((size_t)noff[k] + n) * 10;
            const int l = (int)a[0], r = (int)a[1];
            if ((l != -1 && (l < 0 || l >= m.n_nodes)) || (r != -1 && (r < 0 || r >= m.n_nodes))) return fail(ctx, RT_ERR_INVALID, "mesh %d: bvh_arr10 node %d has a child index out of range", real[k], n);
            const int ts = (int)a[8], te = (int)a[9];
            if (ts < 0 || te < ts || te > m.n_triangles) return fail(ctx, RT_ERR_INVALID, "mesh %d: bvh_arr10 node %d has triangle range [%d,%d) outside [0,%d)", real[k], n, ts, te, m.n_triangles);
            o[0] = l == -1 ? -1.f : (float)(l + noff[k]); o[1] = r == -1 ? -1.f : (float)(r + noff[k]);
            for (int c = 2; c < 8; ++c) o[c] = a[c];
            o[8] = (float)(ts + f.tri_off[k]); o[9] = (float)(te + f.tri_off[k]);
        }
    }
    // the synthetic nodes: meshes [a, b) below node `self`; the RIGHT child holds the first half (visited first)
    int next_syn = 1;
    struct Job { int a, b, self; };
    std::vector<Job> jobs{{0, K, 0}};
    std::vector<Job> post;
    while (!jobs.empty()) {
        const Job j = jobs.back(); jobs.pop_back();
        post.push_back(j);
        const int mid = j.a + (j.b - j.a + 1) / 2;
        auto child = [&](int a, int b) { if (b - a == 1) return noff[a]; const int id = next_syn++; jobs.push_back({a, b, id}); return id; };
        float *o = f.arr.data() + (size_t)j.self * 10;
        o[1] = (float)child(j.a, mid);                                  // right = the earlier meshes
        o[0] = (float)child(mid, j.b);
        o[8] = (float)f.tri_off[j.a]; o[9] = (float)f.tri_off[j.b];
    }
    for (size_t q = post.size(); q-- > 0;) {                            // children before parents: a synthetic node's index is larger than its parent's
        float *o = f.arr.data() + (size_t)post[q].self * 10;
        const float *l = f.arr.data() + (size_t)(int)o[0] * 10, *r = f.arr.data() + (size_t)(int)o[1] * 10;
        for (int c = 0; c < 3; ++c) { o[2 + c] = std::min(l[2 + c], r[2 + c]); o[5 + c] = std::max(l[5 + c], r[5 + c]); }
    }
    f.m = rt_mesh{};
    f.m.vertices = f.verts.data(); f.m.n_vertices = (int)nv; f.m.indices = f.idx.data(); f.m.index_stride = 3; f.m.n_triangles = (int)nt;
    f.m.bvh_arr10 = f.arr.data(); f.m.n_nodes = (int)nn;
    f.m.object_slot = meshes[real[0]].object_slot;
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

int rt_mesh_set_normals(rt_ctx *ctx, const float *normals_xyz, int n_normals, const int32_t *nidx, int index_stride, int n_triangles) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    if (!normals_xyz || !nidx) { ctx->scene.nrm = nullptr; return RT_OK; }          // back to flat shading
    if (ctx->n_real_meshes > 1) return fail(ctx, RT_ERR_UNSUPPORTED, "the scene holds %d meshes: smooth normals are set for ONE TriangleMesh", ctx->n_real_meshes);
    if (int rr = refresh_host_mesh(ctx); rr != RT_OK) return rr;
    if (ctx->scene.mesh_slot < 0) return fail(ctx, RT_ERR_INVALID, "the scene has no mesh");
    if (n_normals <= 0 || index_stride < 3) return fail(ctx, RT_ERR_INVALID, "bad normal array sizes");
    std::vector<float4> nr(ctx->tri_perm.size() * 3);
    for (size_t t = 0; t < ctx->tri_perm.size(); ++t) {
        const int src = ctx->tri_perm[t];
        if (src < 0 || src >= n_triangles) return fail(ctx, RT_ERR_INVALID, "n_triangles %d does not cover the uploaded mesh", n_triangles);
        for (int k = 0; k < 3; ++k) {
            const int ni = nidx[(size_t)src * index_stride + k];
            if (ni < 0 || ni >= n_normals) return fail(ctx, RT_ERR_INVALID, "triangle %d references normal %d outside [0,%d)", src, ni, n_normals);
            nr[3 * t + k] = make_float4(normals_xyz[3 * (size_t)ni], normals_xyz[3 * (size_t)ni + 1], normals_xyz[3 * (size_t)ni + 2], 0.f);
        }
    }
    int rc = upload(ctx, ctx->nrm, nr.data(), nr.size() * sizeof(float4));
    if (rc != RT_OK) return rc;
    ctx->scene.nrm = static_cast<const float4 *>(ctx->nrm.p);
    return RT_OK;
}

int rt_mesh_transform(rt_ctx *ctx, const float rotation[9], const float translation[3]) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!rotation || !translation) return fail(ctx, RT_ERR_INVALID, "rotation/translation is NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    rtk::Scene &sc = ctx->scene;
    if (sc.mesh_slot < 0 || sc.n_nodes <= 0 || sc.n_verts <= 0) return RT_OK;     // no mesh: nothing to move
    RT_HIP(ctx, hipSetDevice(ctx->device));
    rtk::Mat3 m;
    for (int k = 0; k < 9; ++k) m.r[k] = rotation[k];
    for (int k = 0; k < 3; ++k) m.t[k] = translation[k];
    hipLaunchKernelGGL(rtk::transform_kernel, dim3((unsigned)((sc.n_verts + 255) / 256)), dim3(256), 0, own_stream(ctx),
                       static_cast<float4 *>(ctx->verts.p), sc.n_verts, m);
    if (sc.nrm != nullptr)      // the reference's kernel rotates the normals and ADDS the translation to them as well (global_launcher.cu:357-363)
        hipLaunchKernelGGL(rtk::transform_kernel, dim3((unsigned)((3 * sc.n_tris + 255) / 256)), dim3(256), 0, own_stream(ctx),
                           static_cast<float4 *>(ctx->nrm.p), 3 * sc.n_tris, m);
    hipLaunchKernelGGL(rtk::retri_kernel, dim3((unsigned)((sc.n_tris + 255) / 256)), dim3(256), 0, own_stream(ctx),
                       static_cast<const int4 *>(ctx->tidx.p), static_cast<const float4 *>(ctx->verts.p), static_cast<float4 *>(ctx->tri.p), sc.n_tris);
    rtk::RefitArgs a{};
    a.node_lo = static_cast<float4 *>(ctx->node_lo.p); a.node_hi = static_cast<float4 *>(ctx->node_hi.p);
    a.nodes2 = static_cast<float4 *>(ctx->nodes2.p); a.nodesq = static_cast<float4 *>(ctx->nodesq.p); a.nodesb = static_cast<float4 *>(ctx->nodesb.p);
    a.q2thr = static_cast<const int *>(ctx->q2thr.p); a.left_of = static_cast<const int *>(ctx->left_dev.p);
    a.lvl_nodes = static_cast<const int *>(ctx->lvl_nodes.p); a.lvl_off = static_cast<const int *>(ctx->lvl_off.p);
    a.tidx = static_cast<const int4 *>(ctx->tidx.p); a.verts = static_cast<const float4 *>(ctx->verts.p);
    a.n_nodes = sc.n_nodes; a.n_levels = ctx->n_levels;
    hipLaunchKernelGGL(rtk::refit_kernel, dim3(1), dim3(1024), 0, own_stream(ctx), a);
    RT_HIP(ctx, hipGetLastError());
    // the root box travels as a kernel argument (uniform root-box pre-test): fetch the refitted one
    float4 root[2];
    RT_HIP(ctx, hipMemcpyAsync(&root[0], ctx->node_lo.p, sizeof(float4), hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipMemcpyAsync(&root[1], ctx->node_hi.p, sizeof(float4), hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    sc.root_lo = root[0]; sc.root_hi = root[1];
    // the refitted root box contains every node's (unions, bottom-up): it bounds the magnitudes wf_travq's box filter needs
    const float rv[6] = {root[0].x, root[0].y, root[0].z, root[1].x, root[1].y, root[1].z};
    bool fast = true;
    float bm[3];
    for (int a = 0; a < 3; ++a) {
        if (!(rv[a] <= rv[a + 3]) || !(std::fabs(rv[a]) < 1e8f) || !(std::fabs(rv[a + 3]) < 1e8f)) fast = false;
        bm[a] = std::max(std::fabs(rv[a]), std::fabs(rv[a + 3]));
    }
    sc.bmx = bm[0]; sc.bmy = bm[1]; sc.bmz = bm[2];
    sc.fast_box = fast ? 1 : 0;
    return requantize(ctx, own_stream(ctx));                                          // the fixed-point pairs follow the refitted boxes (same topology: q16_topo_ok stands; unions nest)
}

// TriangleMesh::buildBVH on the device, bit for bit (rt_bvhbuild.hip.h): leaves the flat tree in ctx->bb_arr and the triangle order in ctx->bb_idx
static int rebuild_reference_tree(rt_ctx *ctx, const int nt, int &n_nodes_out) {
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cap = 2 * (size_t)nt + 2;                                          // nodes: every split makes two
    int rc;
    if ((rc = ensure(ctx, ctx->bb_idx, nt * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_cnt, nt * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_pa, nt * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_pb, nt * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_tmp, nt * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_nodes_i, 4 * cap * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_nodes_f, 2 * cap * sizeof(float4))) != RT_OK || (rc = ensure(ctx, ctx->bb_counter, 2 * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_lvl, (cap + 1) * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_size, cap * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_pre, cap * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_arr, cap * 10 * sizeof(float))) != RT_OK)
        return rc;
    rtk::BuildArgs a{};
    a.verts = static_cast<const float4 *>(ctx->verts.p); a.tidx_up = static_cast<const int4 *>(ctx->tidx_up.p);
    a.idx = static_cast<int *>(ctx->bb_idx.p); a.cnt = static_cast<int *>(ctx->bb_cnt.p);
    a.ptr_a = static_cast<int *>(ctx->bb_pa.p); a.ptr_b = static_cast<int *>(ctx->bb_pb.p); a.tmp = static_cast<int *>(ctx->bb_tmp.p);
    int *ni = static_cast<int *>(ctx->bb_nodes_i.p);
    a.n_start = ni; a.n_end = ni + cap; a.n_left = ni + 2 * cap; a.n_right = ni + 3 * cap;
    a.n_mn = static_cast<float4 *>(ctx->bb_nodes_f.p); a.n_mx = a.n_mn + cap;
    a.counter = static_cast<int *>(ctx->bb_counter.p); a.n_tris = nt; a.cap = (int)cap;
    hipStream_t q = own_stream(ctx);
    // root = node 0 over all triangles (buildBVH(&bvh, 0, T), cpu:684); the permutation starts as the identity
    hipLaunchKernelGGL(rtk::iota_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, q, a.idx, nt);
    const int root_range[2] = {0, nt}, one[2] = {1, 0};              // counter[0] = nodes allocated, counter[1] = a split was refused for lack of capacity
    RT_HIP(ctx, hipMemcpyAsync(a.n_start, &root_range[0], sizeof(int), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipMemcpyAsync(a.n_end, &root_range[1], sizeof(int), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipMemcpyAsync(a.counter, one, 2 * sizeof(int), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipStreamSynchronize(q));                                           // the three sources above live on this stack frame
    std::vector<int> lvl_first{0};
    int first = 0, count = 1;
    while (count > 0) {                                                             // one launch per level, one workgroup per node
        hipLaunchKernelGGL(rtk::bvh_level_kernel, dim3((unsigned)count), dim3(rtk::kBuildThreads), 0, q, a, first);
        RT_HIP(ctx, hipGetLastError());
        int tot[2] = {0, 0};
        RT_HIP(ctx, hipMemcpyAsync(tot, a.counter, 2 * sizeof(int), hipMemcpyDeviceToHost, q));
        RT_HIP(ctx, hipStreamSynchronize(q));
        const int total = tot[0];
        // the kernel refuses a split that would pass the arrays' capacity (it cannot for a tree over nt triangles); the scene in use is untouched so far
        if (tot[1] != 0 || (size_t)total > cap) return fail(ctx, RT_ERR_INTERNAL, "BVH build needed more than %zu nodes for %d triangles (scene unchanged)", cap, nt);
        first += count;
        lvl_first.push_back(first);
        count = total - first;
    }
    const int n_nodes = first, n_levels = (int)lvl_first.size() - 1;
    n_nodes_out = n_nodes;
    if (n_nodes >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "node indices are stored as floats: < 2^24 nodes");
    RT_HIP(ctx, hipMemcpyAsync(ctx->bb_lvl.p, lvl_first.data(), lvl_first.size() * sizeof(int), hipMemcpyHostToDevice, q));
    hipLaunchKernelGGL(rtk::bvh_flatten_kernel, dim3(1), dim3(1024), 0, q, a, static_cast<const int *>(ctx->bb_lvl.p), n_levels,
                       static_cast<int *>(ctx->bb_size.p), static_cast<int *>(ctx->bb_pre.p), static_cast<float *>(ctx->bb_arr.p));
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

// The LBVH builder (rt_lbvh.hip.h): Morton sort + parallel hierarchy emission, leaves cut by the surface-area heuristic (at most kLbvhLeaf = 32 triangles); same outputs
static int rebuild_lbvh_tree(rt_ctx *ctx, const int nt, int &n_nodes_out) {
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t q = own_stream(ctx);
    const size_t n = (size_t)nt, nc = 2 * n - 1;
    int rc;
    DevBuf &B = ctx->lb_pool;
    // one pool, carved: keys (2 x 8n), vals (2 x 4n), 6 int arrays of n, flags, boxes (4 x 16n), alive + index (2 x 4 (2n)), bounds / stats
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_keys = carve(8 * n), o_keys2 = carve(8 * n), o_vals = carve(4 * n), o_vals2 = carve(4 * n);
    const size_t o_left = carve(4 * n), o_right = carve(4 * n), o_parent = carve(4 * n), o_first = carve(4 * n), o_last = carve(4 * n), o_lparent = carve(4 * n), o_flag = carve(4 * n), o_cost = carve(4 * n), o_leafify = carve(4 * n);
    const size_t o_ilo = carve(16 * n), o_ihi = carve(16 * n), o_llo = carve(16 * n), o_lhi = carve(16 * n);
    const size_t o_alive = carve(4 * nc), o_index = carve(4 * nc), o_small = carve(64);
    size_t sort_tmp = 0, scan_tmp = 0;
    {
        unsigned long long *k0 = nullptr; int *v0 = nullptr;
        if (rocprim::radix_sort_pairs(nullptr, sort_tmp, k0, k0, v0, v0, n, 0, 63, q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::radix_sort_pairs (size query) failed");
        if (rocprim::exclusive_scan(nullptr, scan_tmp, v0, v0, 0, nc, rocprim::plus<int>(), q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::exclusive_scan (size query) failed");
    }
    const size_t o_tmp = carve(std::max(sort_tmp, scan_tmp) + 256);
    if ((rc = ensure(ctx, B, off)) != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->bb_idx, n * sizeof(int))) != RT_OK) return rc;
    uint8_t *base = static_cast<uint8_t *>(B.p);
    rtk::LbvhArgs a{};
    // the leaf cut's triangle cost: kLbvhCt (wf_travq's step times on 64-byte pairs) for small trees; 1.0 for trees that will use the 32-byte fixed-point pairs, where a box test is
    // cheaper still but a triangle's 48-byte gather is not (swept on 524 288 / 2 M triangles: Ct 1.0 / 1.6 / 2.5 / 4 / 8 = 2.77 / 2.84 / 2.99 / 3.04 / 3.06 and 8.21 / 8.40 / 8.72 / 8.73 / 8.80 ms per frame)
    a.ct = ctx->knobs.lbvh_ct > 0.f ? ctx->knobs.lbvh_ct : (nt >= kQ16AutoNodes ? 1.0f : rtk::kLbvhCt); a.cb = rtk::kLbvhCb;
    a.verts = static_cast<const float4 *>(ctx->verts.p); a.tidx_up = static_cast<const int4 *>(ctx->tidx_up.p); a.n = nt;
    a.bounds = reinterpret_cast<unsigned int *>(base + o_small); a.stats = reinterpret_cast<int *>(base + o_small + 32);
    a.keys = reinterpret_cast<unsigned long long *>(base + o_keys); a.vals = reinterpret_cast<int *>(base + o_vals);
    a.left = reinterpret_cast<int *>(base + o_left); a.right = reinterpret_cast<int *>(base + o_right); a.parent = reinterpret_cast<int *>(base + o_parent);
    a.first = reinterpret_cast<int *>(base + o_first); a.last = reinterpret_cast<int *>(base + o_last); a.leaf_parent = reinterpret_cast<int *>(base + o_lparent);
    a.flag = reinterpret_cast<int *>(base + o_flag);
    a.cost = reinterpret_cast<float *>(base + o_cost); a.leafify = reinterpret_cast<int *>(base + o_leafify);
    a.ibox_lo = reinterpret_cast<float4 *>(base + o_ilo); a.ibox_hi = reinterpret_cast<float4 *>(base + o_ihi);
    a.lbox_lo = reinterpret_cast<float4 *>(base + o_llo); a.lbox_hi = reinterpret_cast<float4 *>(base + o_lhi);
    a.alive = reinterpret_cast<int *>(base + o_alive); a.index = reinterpret_cast<int *>(base + o_index);
    const unsigned int binit[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    int zero4[4] = {0, 0, 0, 0};
    RT_HIP(ctx, hipMemcpyAsync(a.bounds, binit, sizeof(binit), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipMemcpyAsync(a.stats, zero4, sizeof(zero4), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipStreamSynchronize(q));                                        // (the two sources live on this stack frame)
    const dim3 gt((unsigned)((n + 255) / 256)), gc((unsigned)((nc + 255) / 256)), blk(256);
    hipLaunchKernelGGL(rtk::lbvh_bounds_kernel, gt, blk, 0, q, a);
    hipLaunchKernelGGL(rtk::lbvh_morton_kernel, gt, blk, 0, q, a);
    {   // (code, triangle) pairs by code; the sorted arrays become a.keys / a.vals
        unsigned long long *k2 = reinterpret_cast<unsigned long long *>(base + o_keys2);
        int *v2 = reinterpret_cast<int *>(base + o_vals2);
        size_t tmp = sort_tmp;
        if (rocprim::radix_sort_pairs(base + o_tmp, tmp, a.keys, k2, a.vals, v2, n, 0, 63, q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::radix_sort_pairs failed");
        a.keys = k2; a.vals = v2;
    }
    hipLaunchKernelGGL(rtk::lbvh_hierarchy_kernel, gt, blk, 0, q, a);
    hipLaunchKernelGGL(rtk::lbvh_boxes_kernel, gt, blk, 0, q, a);          // boxes + the leaf-or-subtree decision, bottom-up
    hipLaunchKernelGGL(rtk::lbvh_alive_kernel, gc, blk, 0, q, a);
    {
        size_t tmp = scan_tmp;
        if (rocprim::exclusive_scan(base + o_tmp, tmp, a.alive, a.index, 0, nc, rocprim::plus<int>(), q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::exclusive_scan failed");
    }
    RT_HIP(ctx, hipGetLastError());
    int last2[2] = {0, 0};                                                       // n_alive = index[last] + alive[last]
    RT_HIP(ctx, hipMemcpyAsync(&last2[0], a.index + (nc - 1), sizeof(int), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(&last2[1], a.alive + (nc - 1), sizeof(int), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipStreamSynchronize(q));
    const int n_nodes = last2[0] + last2[1];
    if (n_nodes < 1 || (size_t)n_nodes > nc) return fail(ctx, RT_ERR_INTERNAL, "LBVH build: %d nodes for %d triangles (scene unchanged)", n_nodes, nt);
    if (n_nodes >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "node indices are stored as floats: < 2^24 nodes");
    if ((rc = ensure(ctx, ctx->bb_arr, (size_t)n_nodes * 10 * sizeof(float))) != RT_OK) return rc;
    a.arr10 = static_cast<float *>(ctx->bb_arr.p);
    hipLaunchKernelGGL(rtk::lbvh_emit_kernel, gc, blk, 0, q, a);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(ctx->bb_idx.p, a.vals, n * sizeof(int), hipMemcpyDeviceToDevice, q));   // the triangle order, where the shared tail expects it
    int st[4] = {0, 0, 0, 0};
    RT_HIP(ctx, hipMemcpyAsync(st, a.stats, sizeof(st), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipStreamSynchronize(q));
    ctx->build.n_leaves = st[0]; ctx->build.max_leaf_tris = st[1]; ctx->build.max_depth = st[2];
    ctx->lb_args = a;
    n_nodes_out = n_nodes;
    return RT_OK;
}

// The render kernels' formats from the LBVH builder's arrays, on the device (rt_lbvh.hip.h, second half): what install_scene does on the
// host for an uploaded tree.  `old`: the scene in use (spheres, light, camera, albedo, mesh slot carry over).
static int install_lbvh_device(rt_ctx *ctx, const rtk::Scene &old, const int n_nodes) {
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t q = own_stream(ctx);
    const rtk::LbvhArgs &a = ctx->lb_args;
    const size_t n = (size_t)a.n, N = (size_t)n_nodes, nc = 2 * n - 1;
    int rc;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_flag = carve(4 * (n + 1)), o_scan = carve(4 * (n + 1)), o_X = carve(4 * N), o_bfs = carve(4 * N), o_key = carve(8 * N), o_key2 = carve(8 * N),
                 o_val = carve(4 * N), o_val2 = carve(4 * N), o_hist = carve(4 * 80), o_upnew = carve(16 * n);
    size_t sort_tmp = 0, scan_tmp = 0;
    {
        unsigned long long *k0 = nullptr; int *v0 = nullptr;
        if (rocprim::radix_sort_pairs(nullptr, sort_tmp, k0, k0, v0, v0, N, 0, 64, q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::radix_sort_pairs (size query) failed");
        if (rocprim::exclusive_scan(nullptr, scan_tmp, v0, v0, 0, n + 1, rocprim::plus<int>(), q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::exclusive_scan (size query) failed");
    }
    const size_t o_tmp = carve(std::max(sort_tmp, scan_tmp) + 256);
    if ((rc = ensure(ctx, ctx->lb_pool2, off)) != RT_OK) return rc;
    // the scene in use stays untouched until every allocation has succeeded
    DevBuf *outs[] = {&ctx->node_lo, &ctx->node_hi, &ctx->nodes2, &ctx->nodesq, &ctx->nodesb, &ctx->q2thr, &ctx->left_dev, &ctx->lvl_nodes, &ctx->lvl_off, &ctx->tri, &ctx->tidx, &ctx->perm_dev};
    const size_t need[] = {N * 16, N * 16, 2 * N * 16, 2 * (N + 1) * 16, 2 * (N + 1) * 16, (N + 1) * 4, N * 4, N * 4, 80 * 4, 3 * n * 16, n * 16, n * 4};
    ctx->have_scene = false;                                                     // (a failure from here on leaves the context without a scene, as the host path does)
    for (size_t k = 0; k < sizeof(need) / sizeof(need[0]); ++k) if ((rc = ensure(ctx, *outs[k], need[k])) != RT_OK) return rc;
    uint8_t *base = static_cast<uint8_t *>(ctx->lb_pool2.p);
    int *flag = reinterpret_cast<int *>(base + o_flag);
    rtk::LbvhLayout y{};
    y.n_nodes = n_nodes;
    y.lscan = reinterpret_cast<int *>(base + o_scan); y.X = reinterpret_cast<int *>(base + o_X); y.bfs = reinterpret_cast<int *>(base + o_bfs);
    y.bkey = reinterpret_cast<unsigned long long *>(base + o_key); y.bval = reinterpret_cast<int *>(base + o_val);
    y.dhist = reinterpret_cast<int *>(base + o_hist);
    y.node_lo = static_cast<float4 *>(ctx->node_lo.p); y.node_hi = static_cast<float4 *>(ctx->node_hi.p); y.nodes2 = static_cast<float4 *>(ctx->nodes2.p);
    y.nodesq = static_cast<float4 *>(ctx->nodesq.p); y.nodesb = static_cast<float4 *>(ctx->nodesb.p); y.q2thr = static_cast<int *>(ctx->q2thr.p);
    y.left_of = static_cast<int *>(ctx->left_dev.p); y.lvl_nodes = static_cast<int *>(ctx->lvl_nodes.p);
    y.tidx_visit = static_cast<int4 *>(ctx->tidx.p); y.tidx_up_new = reinterpret_cast<int4 *>(base + o_upnew); y.perm = static_cast<int *>(ctx->perm_dev.p);
    RT_HIP(ctx, hipMemsetAsync(flag, 0, 4 * (n + 1), q));
    RT_HIP(ctx, hipMemsetAsync(y.dhist, 0, 4 * 80, q));
    RT_HIP(ctx, hipMemsetAsync(y.nodesq, 0, 32, q));                              // entry 0 of the breadth-first arrays is padding
    RT_HIP(ctx, hipMemsetAsync(y.nodesb, 0, 32, q));
    RT_HIP(ctx, hipMemsetAsync(y.q2thr, 0, 4, q));
    const dim3 gt((unsigned)((n + 255) / 256)), gc((unsigned)((nc + 255) / 256)), gn((unsigned)((N + 255) / 256)), blk(256);
    hipLaunchKernelGGL(rtk::lbvh_leafflag_kernel, gc, blk, 0, q, a, flag);
    { size_t tmp = scan_tmp; if (rocprim::exclusive_scan(base + o_tmp, tmp, flag, y.lscan, 0, n + 1, rocprim::plus<int>(), q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::exclusive_scan failed"); }
    hipLaunchKernelGGL(rtk::lbvh_walk_kernel, gc, blk, 0, q, a, y);
    unsigned long long *key2 = reinterpret_cast<unsigned long long *>(base + o_key2);
    int *val2 = reinterpret_cast<int *>(base + o_val2);
    { size_t tmp = sort_tmp; if (rocprim::radix_sort_pairs(base + o_tmp, tmp, y.bkey, key2, y.bval, val2, N, 0, 64, q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::radix_sort_pairs failed"); }
    hipLaunchKernelGGL(rtk::lbvh_rank_kernel, gn, blk, 0, q, y, val2);
    const int4 *up_old = static_cast<const int4 *>(ctx->tidx_up.p);
    hipLaunchKernelGGL(rtk::lbvh_layout_kernel, gc, blk, 0, q, a, y, up_old);
    hipLaunchKernelGGL(rtk::lbvh_reorder_kernel, gt, blk, 0, q, a, up_old, y.tidx_up_new);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(ctx->tidx_up.p, y.tidx_up_new, n * sizeof(int4), hipMemcpyDeviceToDevice, q));   // the sorted order is the new uploaded order
    hipLaunchKernelGGL(rtk::retri_kernel, gt, blk, 0, q, static_cast<const int4 *>(ctx->tidx.p), static_cast<const float4 *>(ctx->verts.p), static_cast<float4 *>(ctx->tri.p), (int)n);
    RT_HIP(ctx, hipGetLastError());
    int hist[65];
    float4 root[2];
    RT_HIP(ctx, hipMemcpyAsync(hist, y.dhist, sizeof(hist), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(&root[0], ctx->node_lo.p, sizeof(float4), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(&root[1], ctx->node_hi.p, sizeof(float4), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipStreamSynchronize(q));
    const int maxd = hist[64];
    if (maxd > 58) return fail(ctx, RT_ERR_INTERNAL, "LBVH layout: depth %d exceeds the 58 path bits of the breadth-first sort key", maxd);
    std::vector<int> lvl_off(maxd + 2, 0);
    for (int d = 0; d <= maxd; ++d) lvl_off[d + 1] = lvl_off[d] + hist[d];
    if (lvl_off[maxd + 1] != n_nodes) return fail(ctx, RT_ERR_INTERNAL, "LBVH layout: %d nodes in the depth histogram, %d in the tree", lvl_off[maxd + 1], n_nodes);
    if ((rc = upload(ctx, ctx->lvl_off, lvl_off.data(), lvl_off.size() * sizeof(int))) != RT_OK) return rc;
    ctx->n_levels = maxd + 1;
    rtk::Scene sc = old;
    sc.nrm = nullptr;
    sc.n_nodes = n_nodes; sc.n_tris = (int)n;
    mesh_table_single(sc, ctx->real_obj);
    sc.root_lo = root[0]; sc.root_hi = root[1];
    bool fast = true;
    const float rv[6] = {root[0].x, root[0].y, root[0].z, root[1].x, root[1].y, root[1].z};
    float bm[3];
    for (int k = 0; k < 3; ++k) {                                                // every box nests inside the root's (unions, bottom-up), min <= max by construction
        if (!(rv[k] <= rv[k + 3]) || !(std::fabs(rv[k]) < 1e8f) || !(std::fabs(rv[k + 3]) < 1e8f)) fast = false;
        bm[k] = std::max(std::fabs(rv[k]), std::fabs(rv[k + 3]));
    }
    sc.bmx = bm[0]; sc.bmy = bm[1]; sc.bmz = bm[2]; sc.fast_box = fast ? 1 : 0;
    sc.node_lo = static_cast<const float4 *>(ctx->node_lo.p); sc.node_hi = static_cast<const float4 *>(ctx->node_hi.p);
    sc.nodes = static_cast<const float4 *>(ctx->nodes2.p); sc.nodesq = static_cast<const float4 *>(ctx->nodesq.p); sc.nodesb = static_cast<const float4 *>(ctx->nodesb.p);
    sc.q2thr = static_cast<const int *>(ctx->q2thr.p); sc.tri = static_cast<const float4 *>(ctx->tri.p);
    sc.verts = static_cast<const float4 *>(ctx->verts.p); sc.tidx = static_cast<const int4 *>(ctx->tidx.p);
    ctx->travq_ok = n_nodes + 2 < (1 << rtk::kQNodeBits) && (uint64_t)n * 48 < ((uint64_t)1 << 32);   // leaves hold at most kLbvhLeaf triangles
    ctx->scene = sc;
    ctx->host_mesh_stale = true;                                                 // tri_perm / up_indices: on the device now (perm_dev, tidx_up)
    ctx->have_scene = true;
    ctx->q16_topo_ok = true;                                                     // boxes are unions, bottom-up: they nest
    ctx->qw_topo_ok = true;                                                      // (an LBVH leaf holds at least one triangle)
    ctx->q16_leaf_shift = rtk::q16_leaf_shift(rtk::kLbvhLeaf, n);                // leaves of at most kLbvhLeaf triangles
    return requantize(ctx, q);
}

int rt_mesh_rebuild_mode(rt_ctx *ctx, int mode, float *bvh_arr10_out, int32_t *tri_order_out, int32_t *n_nodes_out) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (mode != RT_BVH_REFERENCE && mode != RT_BVH_LBVH) return fail(ctx, RT_ERR_INVALID, "unknown BVH mode %d", mode);
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    if (n_nodes_out) *n_nodes_out = 0;
    const rtk::Scene old = ctx->scene;
    const int nt = ctx->n_up_tris, nv = old.n_verts;
    if (ctx->n_real_meshes > 1) return fail(ctx, RT_ERR_UNSUPPORTED, "the scene holds %d meshes: a rebuild works on ONE TriangleMesh (upload the rebuilt meshes again)", ctx->n_real_meshes);
    if (old.mesh_slot < 0 || ctx->real_obj < 0 || nt <= 0 || nv <= 0) return RT_OK;  // no mesh: nothing to build
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t q = own_stream(ctx);
    int rc;
    int n_nodes = 0;
    ctx->build = rt_build_stats{};
    hipEvent_t e0 = ctx->ev_t0, e1 = ctx->ev_t1;                                    // (the tone-mapping events are free here: nothing else runs on the stream)
    RT_HIP(ctx, hipEventRecord(e0, q));
    // a mesh of a single leaf's worth of triangles is a single leaf in either mode (cpu:217: fewer than five triangles are never split)
    if (mode == RT_BVH_LBVH && nt > 4) rc = rebuild_lbvh_tree(ctx, nt, n_nodes);
    else { mode = RT_BVH_REFERENCE; rc = rebuild_reference_tree(ctx, nt, n_nodes); }
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipEventRecord(e1, q));
    RT_HIP(ctx, hipEventSynchronize(e1));
    RT_HIP(ctx, hipEventElapsedTime(&ctx->build.device_build_ms, e0, e1));
    ctx->have_tonemap_time = false;                                                 // (the borrowed events no longer bracket a tone mapping)
    ctx->build.mode = mode; ctx->build.n_nodes = n_nodes; ctx->build.n_triangles = nt;
    const auto t_install = std::chrono::steady_clock::now();
    int *const order_dev = static_cast<int *>(ctx->bb_idx.p);
    if (mode == RT_BVH_LBVH && old.nrm == nullptr && !ctx->lbvh_host_install && ctx->build.max_depth <= 56) {
        // the kernels' formats straight from the builder's arrays; the flat tree and the order travel to the host only if the caller asks
        if ((rc = install_lbvh_device(ctx, old, n_nodes)) != RT_OK) return rc;
        if (bvh_arr10_out) RT_HIP(ctx, hipMemcpyAsync(bvh_arr10_out, ctx->bb_arr.p, (size_t)n_nodes * 10 * sizeof(float), hipMemcpyDeviceToHost, q));
        if (tri_order_out) RT_HIP(ctx, hipMemcpyAsync(tri_order_out, order_dev, (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, q));
        RT_HIP(ctx, hipStreamSynchronize(q));
        if (n_nodes_out) *n_nodes_out = n_nodes;
        ctx->build.install_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_install).count();
        ctx->build.install_on_device = 1;
        return RT_OK;
    }
    if ((rc = refresh_host_mesh(ctx)) != RT_OK) return rc;                          // the host path below starts from up_indices / tri_perm
    // The tree is built.  The O(n) re-layout for the kernels (traversal order, visit-order triangle records, sibling pairs, refit
    // levels) reuses the upload path on the host: ~30 bytes per triangle over PCIe each way.
    std::vector<float> arr((size_t)n_nodes * 10);
    std::vector<int> order(nt);
    std::vector<float4> hv(nv);
    RT_HIP(ctx, hipMemcpyAsync(arr.data(), ctx->bb_arr.p, arr.size() * sizeof(float), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(order.data(), order_dev, order.size() * sizeof(int), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(hv.data(), ctx->verts.p, hv.size() * sizeof(float4), hipMemcpyDeviceToHost, q));
    std::vector<float4> old_nrm;
    if (old.nrm != nullptr) {
        old_nrm.resize((size_t)old.n_tris * 3);
        RT_HIP(ctx, hipMemcpyAsync(old_nrm.data(), ctx->nrm.p, old_nrm.size() * sizeof(float4), hipMemcpyDeviceToHost, q));
    }
    RT_HIP(ctx, hipStreamSynchronize(q));
    std::vector<float> vx((size_t)nv * 3);
    for (int i = 0; i < nv; ++i) { vx[3 * (size_t)i] = hv[i].x; vx[3 * (size_t)i + 1] = hv[i].y; vx[3 * (size_t)i + 2] = hv[i].z; }
    std::vector<int32_t> ix((size_t)nt * 3);
    for (int t = 0; t < nt; ++t) for (int k = 0; k < 3; ++k) ix[3 * (size_t)t + k] = ctx->up_indices[3 * (size_t)order[t] + k];
    const std::vector<int> old_perm = ctx->tri_perm;                               // old visit order -> old uploaded order
    rt_mesh m{};
    m.vertices = vx.data(); m.n_vertices = nv; m.indices = ix.data(); m.index_stride = 3; m.n_triangles = nt;
    m.bvh_arr10 = arr.data(); m.n_nodes = n_nodes;
    m.object_slot = ctx->real_obj;                                                  // (albedo and material stay in the scene's mesh table, which `sc` carries over)
    rtk::Scene sc = old;
    sc.n_nodes = sc.n_tris = sc.n_verts = 0; sc.nrm = nullptr;
    if ((rc = install_scene(ctx, sc, &m)) != RT_OK) return rc;
    if (!old_nrm.empty()) {                                                        // smooth normals travel with their triangles
        std::vector<int> old_visit_of(nt, -1);
        for (size_t t = 0; t < old_perm.size(); ++t) old_visit_of[old_perm[t]] = (int)t;
        std::vector<float4> nn(ctx->tri_perm.size() * 3);
        for (size_t t = 0; t < ctx->tri_perm.size(); ++t) {
            const int ov = old_visit_of[order[ctx->tri_perm[t]]];
            for (int k = 0; k < 3; ++k) nn[3 * t + k] = ov >= 0 ? old_nrm[3 * (size_t)ov + k] : make_float4(0, 0, 0, 0);
        }
        if ((rc = upload(ctx, ctx->nrm, nn.data(), nn.size() * sizeof(float4))) != RT_OK) return rc;
        ctx->scene.nrm = static_cast<const float4 *>(ctx->nrm.p);
    }
    if (bvh_arr10_out) memcpy(bvh_arr10_out, arr.data(), arr.size() * sizeof(float));
    if (tri_order_out) memcpy(tri_order_out, order.data(), order.size() * sizeof(int));
    if (n_nodes_out) *n_nodes_out = n_nodes;
    ctx->build.install_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_install).count();
    return RT_OK;
}

int rt_mesh_rebuild(rt_ctx *ctx, float *bvh_arr10_out, int32_t *tri_order_out, int32_t *n_nodes_out) {
    return rt_mesh_rebuild_mode(ctx, RT_BVH_REFERENCE, bvh_arr10_out, tri_order_out, n_nodes_out);
}

int rt_mesh_build_stats(const rt_ctx *ctx, rt_build_stats *out) {
    if (!ctx || !out) return fail(nullptr, RT_ERR_INVALID, "bad arguments");
    *out = ctx->build;
    return RT_OK;
}


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
