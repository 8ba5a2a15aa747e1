// rt_travq.hip.h -- wf_travq: BVH traversal as a wave-level work stack of (ray, node) pairs.
//
// Replaces the walk of TriangleMesh::intersect (cpu_launcher.cpp:277-311; optimized.cu:245-285) inside the
// wavefront pipeline of rt_wavefront.hip.h (same path state, same wf_advance kernel).
//
// The reference pushes EVERY child whose box is hit (no distance pruning), so the set of nodes and
// triangles a ray visits does not depend on the order they are visited in, and the nearest hit is the minimum
// over the visited triangles of (t, visit rank) -- the strict '<' of cpu:301 keeps the earliest of equal t, and
// triangles are stored in visit order, so rank == triangle index.  Hence a ray's traversal is not a walk but a
// bag of independent box tests: a wave keeps R rays resident in LDS and one LIFO stack of 32-bit entries
// (ray slot << 26 | c), each standing for the SIBLING PAIR of nodes c, c + 1 (the reference tests both children of a hit
// node, cpu:288-289; in the breadth-first node array they share one 64-byte line).  A BOX step pops up to 64 entries,
// every lane tests the two boxes of its pair (BoundingBox::intersect, cpu:146-157, through the error-bounded filter of
// rt_kernels.hip.h: one table read, four loads in flight together, one counter update) and pushes one entry per hit
// internal child or appends a hit leaf's (first, count) to the wave's leaf queue.  A TRI step takes leaf entries worth up to 128 triangles, hands every lane two of them
// (prefix sum + mark/ballot expansion) and merges accepted hits with a 64-bit LDS min on bits(t) << 32 | index
// (moller_trumbore, cpu:226-236).  Per-slot counters of outstanding entries tell when a ray is finished; finished
// slots are refilled from the workgroup's share of the traversal queue (slot order: flag and record arrive in one
// round trip, 64 slots at a time, and wait in an LDS staging area).  Lanes carry no per-ray state, so there are no
// dependent node-to-node load chains, no stragglers (a long ray is spread over the lanes), and lane occupancy is
// that of the stack, not of the slowest ray.
//
// LDS is bounded for any tree: when the stack cannot take the pushes of a full BOX step, the wave drains the popped
// pairs by walking their subtrees serially with the stackless (skip-pointer) node array instead.
#pragma once
#include "rt_wavefront.hip.h"

namespace rtk {

#ifndef RT_TRAVQ_BLOCK
#define RT_TRAVQ_BLOCK 256
#endif
constexpr int kQBlock = RT_TRAVQ_BLOCK;      // 4 waves per workgroup share one ray-slot cursor (128 / 512 measured: no better)
#ifndef RT_TRAVQ_KP
#define RT_TRAVQ_KP 1                        // sibling pairs per lane and BOX step
#endif
constexpr int kQLeafCap = 256 * RT_TRAVQ_KP;  // leaf-queue entries per wave: < 64 before a BOX step, which appends up to 128 per pair
constexpr int kQNodeBits = 26;               // entry = slot << 26 | node
constexpr unsigned kQNodeMask = (1u << kQNodeBits) - 1u;

// stack capacity: sized so that four waves' carves (+ the cursor) fill 32 KiB (R = 64: 5 workgroups per CU) or less;
// a fuller stack is drained serially (see above), which the cat never needs
template <int R> struct QStackCap { static constexpr int value = 652; };

template <int R, int SCAP, int LCAP> struct QCarve {
    static constexpr int kTabA = 0;                       // float4[R]: (1/u.xyz by v_rcp_f32, c0 | +inf if the filter must not decide)
    static constexpr int kTabC = kTabA + 16 * R;          // float4[R]: (O.xyz, u.x)
    static constexpr int kTabD = kTabC + 16 * R;          // float2[R]: (u.y, u.z)
    static constexpr int kBest = kTabD + 8 * R;           // u64[R]: nearest accepted hit
    static constexpr int kPend = kBest + 8 * R;           // int[R]: outstanding stack + leaf-queue entries
    static constexpr int kMarks = kPend + 4 * R;          // u8[128]: TRI-step expansion marks (all zero between steps)
    static constexpr int kStack = kMarks + 128;           // u32[SCAP]
    static constexpr int kLeaf = kStack + 4 * SCAP;       // uint2[LCAP]: (first triangle, slot | count << 8)
    static constexpr int kStage = kLeaf + 8 * LCAP;       // u8[64]: lanes whose registers hold a fetched ray record that has no slot yet
    static constexpr int kBytes = kStage + 64;
    static_assert(kBytes % 16 == 0 && kLeaf % 8 == 0 && kStack % 4 == 0, "the next wave's float4 tables start at kBytes");
};

// wave64 inclusive prefix sum by DPP (row_shr 1,2,4,8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31).
// Must be called with all 64 lanes active.
__device__ __forceinline__ unsigned int wave_incl_scan(unsigned int x) {
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ int lanes_below(unsigned long long m) {   // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}

// BoundingBox::intersect (cpu:146-157) through the fused filter (see RayBox in rt_wavefront.hip.h): returns whether
// the filter decided; `hit` is then the reference's result.  A = (r.xyz, c0) with r = v_rcp_f32(u), C = (O.xyz, -);
// c0 = +inf for rays the filter must not decide (0 / denormal / inf / NaN components), which makes every comparison
// below false.
__device__ __forceinline__ bool qbox_filter(const float4 lo, const float4 hi, const float4 A, const float4 C, bool &hit) {
    const float ox = C.x * A.x, oy = C.y * A.y, oz = C.z * A.z;       // RayBox::ox.. (single roundings, as in ray_box)
    const float ax = fmaf(lo.x, A.x, -ox), bx = fmaf(hi.x, A.x, -ox);
    const float ay = fmaf(lo.y, A.y, -oy), by = fmaf(hi.y, A.y, -oy);
    const float az = fmaf(lo.z, A.z, -oz), bz = fmaf(hi.z, A.z, -oz);
    const float tn = vmax3(vmin(ax, bx), vmin(ay, by), vmin(az, bz));
    const float tf = vmin3(vmax(ax, bx), vmax(ay, by), vmax(az, bz));
    const float M = vmax(vmax3abs(ax, bx, ay), vmax3abs(by, az, bz));
    const float d = tf - tn;
    const float band = fmaf(M, 2.f * kRel, A.w);
    hit = d > band;
    return M < kBig && (hit || d < -band);
}

// moller_trumbore (cpu:226-236) + the acceptance test of the leaf loop (cpu:301): beta/gamma through the filter,
// undecided lanes by the literal divisions, t always by the exact division.
// `how` reports the route (callers that do not look at it pay nothing): 0 filter rejected, 1 filter accepted, 2 literal divisions.
__device__ __forceinline__ bool qtri_test(const float4 q0, const float4 q1, const float4 q2, const f3 Oo, const f3 uo,
                                          const float tri_tmin, float &t_out, int &how) {
    const f3 A = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
    const f3 N = mk(q2.y, q2.z, q2.w);
    const float det = dot(uo, N);
    const f3 AO = A - Oo;
    const f3 c = cross(AO, uo);
    const float bn = dot(e2, c);
    const float gn = -dot(e1, c);
    const float rd = __builtin_amdgcn_rcpf(det);
    const float b = bn * rd, g = gn * rd;
    const float eb = fmaf(fabsf(b), kRel, kAbs), eg = fmaf(fabsf(g), kRel, kAbs);
    const float sum = b + g;
    const float es = fmaf(fabsf(sum), 0x1p-22f, eb + eg);
    const bool trust = fabsf(det) > kTiny;      // also false for det == 0 and NaN
    // cpu:232-235 accept iff 0 <= beta <= 1, 0 <= gamma <= 1 and fl(beta + gamma) <= 1.  For finite values the two "<= 1" follow from
    // the rest (gamma >= 0 => fl(beta + gamma) >= beta: rounding is monotone and beta is representable), so three comparisons
    // decide; whatever they leave open -- including beta or gamma above 1 next to an inconclusive sum -- takes the literal tests.
    const bool reject = trust && (b < -eb || g < -eg || sum > 1.f + es);
    bool ok = trust && b >= eb && g >= eg && sum <= 1.f - es;
    how = ok ? 1 : 0;
    if (!reject && !ok && det != 0) {           // undecided: the literal tests (rare)
        const float beta = bn / det;
        const float gamma = gn / det;
        ok = (0 <= beta && beta <= 1) && (0 <= gamma && gamma <= 1) && (beta + gamma <= 1);
        how = 2;
    }
    if (!ok) return false;
    const float t = dot(AO, N) / det;
    t_out = t;
    return t > (tri_tmin > 0.f ? tri_tmin : 0.f) && t < 1e9f;   // cpu:235 (t > 0) and cpu:301 (t > 1e-4); 1e9f = INF narrowed (cpu:283)
}
__device__ __forceinline__ bool qtri_test(const float4 q0, const float4 q1, const float4 q2, const f3 Oo, const f3 uo,
                                          const float tri_tmin, float &t_out) {
    int how;
    return qtri_test(q0, q1, q2, Oo, uo, tri_tmin, t_out, how);
}

// LDSN: the first n_lds nodes (breadth-first order: the top of the tree) are staged in LDS behind the waves' carves and
// read from there; the launch then uses ONE workgroup per CU (blockDim.x = 64 x waves, up to 1024).
// LDSV: the vertex array is staged in LDS by a cooperative copy (different-versions/optimized_vertices-in-shared.cu:681-686)
// and a TRI step reads a triangle as its three vertex indices (global, 16 B) + three LDS vertices, forming e1, e2 and
// N = e1 x e2 with the operations of moller_trumbore (cpu:227-229) -- the same single roundings rt_scene_upload applies
// when it precomputes the 48-byte triangle records the other variants read.  Also ONE workgroup per CU.
template <bool STATS, int R, bool LDSN, bool LDSV>
__global__ __launch_bounds__((LDSN || LDSV) ? 1024 : kQBlock, (LDSN || LDSV || RT_TRAVQ_KP > 1 || kQBlock != 256) ? 4 : 5) void wf_travq(const Scene sc, const Frame fr, const WfState st, const int cap, const int n_lds, const int kLow, const int kMinFree) {
    // kLow: refill while the stack holds fewer entries (sibling pairs) than this (default 96); kMinFree: ... and at least this
    // many slots are free, or the stack is short (default R / 4)
    constexpr int SCAP = QStackCap<R>::value, LCAP = kQLeafCap;
    using Carve = QCarve<R, SCAP, LCAP>;
    static_assert(R <= 64 && (R & (R - 1)) == 0, "ray slots are owned by lanes");
    extern __shared__ __attribute__((aligned(16))) unsigned char travq_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wib = tid >> 6;
    const int wpb = (LDSN || LDSV) ? (int)(blockDim.x >> 6) : kQBlock / 64;
    unsigned char *const wl = travq_smem + wib * Carve::kBytes;
    int *const blk_cur = reinterpret_cast<int *>(travq_smem + wpb * Carve::kBytes);
    float4 *const lnodes = reinterpret_cast<float4 *>(travq_smem + wpb * Carve::kBytes + 16);
    float4 *const lverts = lnodes + (LDSN ? 2 * n_lds : 0);        // LDSV: every vertex (x, y, z, -)
    float4 *const tabA = reinterpret_cast<float4 *>(wl + Carve::kTabA);
    float4 *const tabC = reinterpret_cast<float4 *>(wl + Carve::kTabC);
    float2 *const tabD = reinterpret_cast<float2 *>(wl + Carve::kTabD);
    unsigned long long *const best = reinterpret_cast<unsigned long long *>(wl + Carve::kBest);
    int *const pend = reinterpret_cast<int *>(wl + Carve::kPend);
    unsigned char *const marks = wl + Carve::kMarks;
    unsigned int *const stack = reinterpret_cast<unsigned int *>(wl + Carve::kStack);
    uint2 *const leafq = reinterpret_cast<uint2 *>(wl + Carve::kLeaf);
    unsigned char *const sidx = wl + Carve::kStage;
    if (tid == 0) *blk_cur = 0;
    marks[lane] = 0; marks[lane + 64] = 0;
    if (lane < R) pend[lane] = 0;
    if (LDSN) for (int k = tid; k < 2 * n_lds; k += (int)blockDim.x) lnodes[k] = sc.nodesq[k];
    if (LDSV) for (int k = tid; k < sc.n_verts; k += (int)blockDim.x) lverts[k] = sc.verts[k];
    __syncthreads();

    const float4 *const nodes = sc.nodesq;        // breadth-first order from index 1 (0 is padding): lo.w = first child (even; the other one is next to it) | first triangle (leaf); hi.w = -1 | end
    const size_t blk_base = (size_t)blockIdx.x * (size_t)st.slots_per_block;
    const int blk_n = st.slots_per_block;
    int stage_n = 0, stage_used = 0;              // wave-uniform: staged records and how many of them have been given a slot
    float4 sp0 = make_float4(0, 0, 0, 0);         // staged record of this lane (registers; sidx[k] = lane of the k-th staged record)
    float2 sp1 = make_float2(0, 0);
    int spf = 0;
    const int root_hiw = __float_as_int(sc.root_hi.w);
    int path = -1;                                // lane r < R owns ray slot r: the path index of the ray in it
    int top = 0;                                  // wave-uniform: entries on the stack
    unsigned int lhead = 0, ltail = 0;            // wave-uniform: leaf-queue cursors (monotonic)
    bool drained = false;
    Work wk;
    // optional per-wave record (-DRT_DEBUG builds with RT_DEBUG_TRAV set; tools/dbg_travq.py): st.dbg[16 * wave + k]
#ifdef RT_DEBUG
    const bool dbg_on = st.dbg != nullptr;
#else
    constexpr bool dbg_on = false;
#endif
    const unsigned long long dbg_t0 = dbg_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long dbg_tdrain = 0ull, cy_srv = 0, cy_tri = 0, cy_box = 0, stamp = dbg_on ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned int d_box = 0, d_boxl = 0, d_tri = 0, d_tril = 0, d_rounds = 0, d_rays = 0, d_serial = 0, d_idle = 0, d_fetch = 0, d_maxtop = 0;
#define WQ_STAMP(acc) do { if (dbg_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - stamp; stamp = t_; } } while (0)
    // Invariants of every index that reaches a global or LDS address.  The counting (STATS) instantiation checks them and raises a
    // bit in fr.work[4] instead of faulting (rt_count_work then fails with RT_ERR_INTERNAL); the product instantiation
    // carries no checks.  Bits: 1 path index, 2 triangle index, 4 node index, 8 stack height, 16 leaf queue, 32 staging index.
#define WQ_CHECK(cond, bit, fixup) do { if (STATS && !(cond)) { atomicOr(&fr.work[4], (unsigned long long)(bit)); fixup; } } while (0)

    // the sibling nodes c, c + 1 of a stack entry (c is even: the pair is one 64-byte line): LDS for the staged top of the
    // tree (n_lds is even), L1/L2 otherwise
    auto load_pair = [&](int c, float4 &lo0, float4 &hi0, float4 &lo1, float4 &hi1) {
        if (LDSN) {
            const bool inl = c < n_lds;
            if (inl) { lo0 = lnodes[2 * c]; hi0 = lnodes[2 * c + 1]; lo1 = lnodes[2 * c + 2]; hi1 = lnodes[2 * c + 3]; }
            if (__ballot(!inl) != 0ull) { if (!inl) { lo0 = nodes[2 * c]; hi0 = nodes[2 * c + 1]; lo1 = nodes[2 * c + 2]; hi1 = nodes[2 * c + 3]; } }
        } else {
            lo0 = nodes[2 * c]; hi0 = nodes[2 * c + 1]; lo1 = nodes[2 * c + 2]; hi1 = nodes[2 * c + 3];
        }
    };
    // stack nearly full: walk the subtree of one popped entry serially with the stackless (skip-pointer) node array
    auto drain_serial = [&](int o, int node) {
        const float4 A = tabA[o], C = tabC[o];
        const float2 D = tabD[o];
        const f3 O = mk(C.x, C.y, C.z), u = mk(C.w, D.x, D.y);
        const int xt = sc.q2thr[node];                                // the same node in the stackless array
        const float4 h0 = sc.nodes[2 * xt + 1];
        const int end = __float_as_int(h0.w) >= 0 ? xt + 1 : __float_as_int(sc.nodes[2 * xt].w);
        for (int x = xt; x < end;) {
            const float4 lo = sc.nodes[2 * x], hi = sc.nodes[2 * x + 1];
            const int hiw = __float_as_int(hi.w), low = __float_as_int(lo.w);
            bool hit;
            if (!qbox_filter(lo, hi, A, C, hit)) hit = slab(lo, hi, O, u);
            if (STATS) { wk.box++; if (hit) wk.nodes++; }
            if (hit && hiw >= 0) {
                if (STATS) wk.tris += (uint32_t)(hiw - low);
                for (int i = low; i < hiw; ++i) {
                    const float4 *tp = sc.tri + 3 * (size_t)i;
                    float t;
                    if (qtri_test(tp[0], tp[1], tp[2], O, u, fr.tri_tmin, t))
                        atomicMin(&best[o], (unsigned long long)__float_as_uint(t) << 32 | (unsigned int)i);
                }
            }
            x = (hit || hiw >= 0) ? x + 1 : low;
        }
    };

    for (;;) {
        // wave-uniform by construction; say so (the loop-carried values then live in SGPRs and the branches are scalar)
        top = __builtin_amdgcn_readfirstlane(top);
        lhead = (unsigned int)__builtin_amdgcn_readfirstlane((int)lhead);
        ltail = (unsigned int)__builtin_amdgcn_readfirstlane((int)ltail);
        drained = __builtin_amdgcn_readfirstlane((int)drained) != 0;
        stage_n = __builtin_amdgcn_readfirstlane(stage_n);
        stage_used = __builtin_amdgcn_readfirstlane(stage_used);
        // =============================== retire + refill ===============================
        if (top < kLow) {
            if (lane < R && path >= 0) {
                if (pend[lane] == 0) {
                    st.M[path] = best[lane];          // always: the emitter does not initialise M (WF_NOHIT = no triangle accepted)
                    path = -1;
                }
            }
            // refill free slots.  The workgroup owns a spatially scrambled, contiguous share of the traversal queue; its
            // waves take 64 slots at a time through an LDS cursor, flag and record in ONE round trip, and park the rays that
            // need traversal in the LDS staging area, from where free slots are filled.
            for (int round = 0; round < 6 && top < kLow; ++round) {
                const unsigned long long freem = __ballot(lane < R && path < 0);
                const int n_free = __popcll(freem);
                if (n_free == 0 || (n_free < kMinFree && top >= 64)) break;
                if (stage_used >= stage_n) {
                    if (drained) break;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(blk_cur, 64);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + 64 >= blk_n) { drained = true; if (dbg_on && !dbg_tdrain) dbg_tdrain = __builtin_amdgcn_s_memrealtime(); }
                    spf = 0;
                    if (base + lane < blk_n) {
                        const size_t q = (size_t)blk_base + (size_t)(base + lane);
                        const float4 qb = st.QR[2 * q + 1];             // the slot's 32-byte record: flag and ray in one round trip
                        sp0 = st.QR[2 * q]; sp1 = make_float2(qb.x, qb.y); spf = __float_as_int(qb.z);
                    }
                    const unsigned long long am = __ballot(spf != 0);
                    if (spf != 0) sidx[lanes_below(am)] = (unsigned char)lane;
                    __builtin_amdgcn_wave_barrier();
                    stage_n = __popcll(am);
                    stage_used = 0;
                    if (dbg_on) d_fetch++;
                    if (stage_n == 0) continue;
                }
                const int take = n_free < stage_n - stage_used ? n_free : stage_n - stage_used;
                const int rank = lanes_below(freem);
                const bool got = lane < R && path < 0 && rank < take;
                WQ_CHECK(!got || stage_used + rank < 64, 32, (void)0);
                const int src = got ? (int)sidx[(stage_used + rank) & 63] : lane;   // the staged record lives in that lane's registers
                const float4 r0 = make_float4(__shfl(sp0.x, src, 64), __shfl(sp0.y, src, 64), __shfl(sp0.z, src, 64), __shfl(sp0.w, src, 64));
                const float2 r1 = make_float2(__shfl(sp1.x, src, 64), __shfl(sp1.y, src, 64));
                const int rf = __shfl(spf, src, 64);
                if (got) {
                    const f3 O = mk(r0.x, r0.y, r0.z), u = mk(r0.w, r1.x, r1.y);
                    const RayBox rb = ray_box(O, u);
                    tabA[lane] = make_float4(rb.rx, rb.ry, rb.rz, rb.safe ? rb.c0 : __builtin_inff());
                    tabC[lane] = r0;
                    tabD[lane] = r1;
                    best[lane] = WF_NOHIT;
                    path = rf - 1;
                    WQ_CHECK(path >= 0 && path < 2 * st.n_paths, 1, path = 0);
                }
                stage_used += take;
                // the root box was tested when the ray was emitted (wf_emit_ray): start with what is below it
                const unsigned long long gm = __ballot(got);
                if (dbg_on) { d_rounds++; d_rays += (unsigned int)__popcll(gm); }
                if (root_hiw < 0) {
                    if (got) {
                        stack[top + lanes_below(gm)] = (unsigned int)lane << kQNodeBits | 2u;   // the root (node 1) has the children 2, 3
                        pend[lane] = 1;
                    }
                    top += __popcll(gm);
                } else {                           // the root is a leaf
                    const int first = __float_as_int(sc.root_lo.w), cnt = root_hiw - first;
                    if (cnt > 0) {
                        if (got) {
                            leafq[(ltail + (unsigned int)lanes_below(gm)) & (LCAP - 1)] = make_uint2((unsigned int)first, (unsigned int)lane | (unsigned int)cnt << 8);
                            pend[lane] = 1;
                            if (STATS) wk.tris += (uint32_t)cnt;
                        }
                        ltail += (unsigned int)__popcll(gm);
                    } else if (got) {
                        pend[lane] = 0;
                    }
                    break;                         // at most R leaf entries per pass: the TRI steps below drain them
                }
            }
        }
        WQ_STAMP(cy_srv);
        // =============================== TRI step: two triangles per lane ===============================
        const unsigned int lcount = ltail - lhead;
        if (lcount >= 64u || (top == 0 && lcount > 0u)) {
            const unsigned int m = lcount < 64u ? lcount : 64u;
            uint2 E = make_uint2(0u, 0u);
            if ((unsigned int)lane < m) E = leafq[(lhead + (unsigned int)lane) & (LCAP - 1)];
            const unsigned int c = E.y >> 8;                         // >= 1 for queued entries, 0 beyond them
            const unsigned int incl = wave_incl_scan(c);
            const unsigned int P = incl - c;                         // position of this entry's first triangle
            const bool part = c > 0u && P < 128u;
            const unsigned int all = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
            const unsigned int total = all < 128u ? all : 128u;
            if (part) marks[P] = 1;
            __builtin_amdgcn_wave_barrier();
            const unsigned int mk0 = marks[lane], mk1 = marks[lane + 64];
            const unsigned long long B0 = __ballot(mk0 != 0u), B1 = __ballot(mk1 != 0u);
            __builtin_amdgcn_wave_barrier();
            if (part) marks[P] = 0;
            // entry whose triangle range covers position lane (j0) and position lane + 64 (j1)
            const int j0 = lanes_below(B0) + (int)((B0 >> lane) & 1ull) - 1;
            const int j1 = __popcll(B0) + lanes_below(B1) + (int)((B1 >> lane) & 1ull) - 1;
            const unsigned int f0 = (unsigned int)__shfl((int)E.x, j0, 64), y0 = (unsigned int)__shfl((int)E.y, j0, 64), P0 = (unsigned int)__shfl((int)P, j0, 64);
            const unsigned int f1 = (unsigned int)__shfl((int)E.x, j1, 64), y1 = (unsigned int)__shfl((int)E.y, j1, 64), P1 = (unsigned int)__shfl((int)P, j1, 64);
            const bool t0 = (unsigned int)lane < total, t1 = (unsigned int)lane + 64u < total;
            const int o0 = t0 ? (int)(y0 & 0xffu) : 0, o1 = t1 ? (int)(y1 & 0xffu) : 0;
            int i0 = t0 ? (int)(f0 + ((unsigned int)lane - P0)) : 0, i1 = t1 ? (int)(f1 + ((unsigned int)lane + 64u - P1)) : 0;
            WQ_CHECK(i0 >= 0 && i0 < sc.n_tris && i1 >= 0 && i1 < sc.n_tris, 2, (i0 = 0, i1 = 0));
            WQ_CHECK(ltail - lhead <= (unsigned int)LCAP, 16, (void)0);
            float4 a0, a1, a2, b0, b1, b2;
            if (LDSV) {
                const int4 ia = sc.tidx[i0], ib = sc.tidx[i1];
                auto record = [&](const int4 ix, float4 &q0, float4 &q1, float4 &q2) {
                    const float4 va = lverts[ix.x], vb = lverts[ix.y], vc = lverts[ix.z];
                    const f3 A = mk(va.x, va.y, va.z);
                    const f3 e1 = mk(vb.x, vb.y, vb.z) - A, e2 = mk(vc.x, vc.y, vc.z) - A, N = cross(e1, e2);   // cpu:227-229
                    q0 = make_float4(A.x, A.y, A.z, e1.x); q1 = make_float4(e1.y, e1.z, e2.x, e2.y); q2 = make_float4(e2.z, N.x, N.y, N.z);
                };
                record(ia, a0, a1, a2);
                record(ib, b0, b1, b2);
            } else {
                const float4 *tp0 = sc.tri + 3 * (size_t)i0, *tp1 = sc.tri + 3 * (size_t)i1;
                a0 = tp0[0]; a1 = tp0[1]; a2 = tp0[2];
                b0 = tp1[0]; b1 = tp1[1]; b2 = tp1[2];
            }
            const float4 C0 = tabC[o0], C1 = tabC[o1];
            const float2 D0 = tabD[o0], D1 = tabD[o1];
            float ta, tb;
            int how0, how1;
            const bool ok0 = qtri_test(a0, a1, a2, mk(C0.x, C0.y, C0.z), mk(C0.w, D0.x, D0.y), fr.tri_tmin, ta, how0) && t0;
            const bool ok1 = qtri_test(b0, b1, b2, mk(C1.x, C1.y, C1.z), mk(C1.w, D1.x, D1.y), fr.tri_tmin, tb, how1) && t1;
            if (STATS) wk.lit_tri += ((t0 && how0 == 2) ? 1u : 0u) + ((t1 && how1 == 2) ? 1u : 0u);
            if (ok0) atomicMin(&best[o0], (unsigned long long)__float_as_uint(ta) << 32 | (unsigned int)i0);
            if (ok1) atomicMin(&best[o1], (unsigned long long)__float_as_uint(tb) << 32 | (unsigned int)i1);
            const bool full = part && P + c <= 128u;
            if (part && !full) {                                     // at most one entry straddles position 127: keep its rest
                const unsigned int took = 128u - P;
                leafq[(lhead + (unsigned int)lane) & (LCAP - 1)] = make_uint2(E.x + took, (E.y & 0xffu) | (c - took) << 8);
            }
            lhead += (unsigned int)__popcll(__ballot(full));
            if (full) atomicAdd(&pend[E.y & 0xffu], -1);            // after the mins above (LDS operations stay in order)
            if (dbg_on) { d_tri++; d_tril += total; }
            WQ_STAMP(cy_tri);
            continue;
        }
        if (top == 0) {
            if (drained && stage_used >= stage_n && __ballot(path >= 0) == 0ull) break;   // every wave gets here: each step consumes entries
            if (dbg_on) d_idle++;
            continue;
        }
        // =============================== BOX step: KP sibling pairs (2 KP boxes) per lane ===============================
        constexpr int KP = RT_TRAVQ_KP;
#if defined(RT_DEBUG) && defined(RT_PAD_SALU)     // sensitivity experiment (tools/pad_experiment.sh): 32 extra scalar / vector issue slots, 4 extra LDS reads per step
        asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
                     "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
#endif
#if defined(RT_DEBUG) && defined(RT_PAD_VALU)
        { int pad_ = lane; for (int k_ = 0; k_ < 32; ++k_) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pad_)); }
#endif
#if defined(RT_DEBUG) && defined(RT_PAD_LDS)
        { unsigned int pl_ = marks[lane]; pl_ += marks[lane + 64]; pl_ += marks[lane]; pl_ += marks[lane + 64]; asm volatile("" :: "v"(pl_)); }
#endif
        const int n = top < 64 * KP ? top : 64 * KP;
        if (cap - top < 64 * KP) {                                    // no room for up to 128 KP pushes: serial drain of 64 entries
            const int nd = top < 64 ? top : 64;
            const bool actd = lane < nd;
            const unsigned int ed = actd ? stack[top - 1 - lane] : 0u;
            top -= nd;
            if (actd) { const int od = (int)(ed >> kQNodeBits), cd = (int)(ed & kQNodeMask); drain_serial(od, cd); drain_serial(od, cd + 1); atomicAdd(&pend[od], -1); }
            if (dbg_on) d_serial++;
            WQ_STAMP(cy_box);
            continue;
        }
        bool act[KP]; unsigned int e[KP]; int o[KP], c[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            act[k] = lane + 64 * k < n;
            e[k] = act[k] ? stack[top - 1 - lane - 64 * k] : 0u;      // slot << 26 | c: the sibling nodes c, c + 1 (one 64-byte line)
            o[k] = (int)(e[k] >> kQNodeBits);
            c[k] = (int)(e[k] & kQNodeMask);
        }
        top -= n;
#pragma unroll
        for (int k = 0; k < KP; ++k) WQ_CHECK(!act[k] || (c[k] >= 2 && c[k] + 1 <= sc.n_nodes), 4, c[k] = 0);
        float4 A[KP], C[KP], lo0[KP], hi0[KP], lo1[KP], hi1[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {                                // all loads first: they are in flight together
            A[k] = tabA[o[k]]; C[k] = tabC[o[k]];                      // siblings belong to one ray: one table read for both
            load_pair(c[k], lo0[k], hi0[k], lo1[k], hi1[k]);
        }
#if defined(RT_DEBUG) && defined(RT_PAD_VMEM)     // sensitivity experiment: the four 16-byte loads of the first pair once more (L1 hits: address / tag pipeline only)
        {
            const float4 *pp = nodes + 2 * c[0];
            typedef float pad_v4f __attribute__((ext_vector_type(4)));
            pad_v4f x0, x1, x2, x3;
            asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n"
                         "global_load_dwordx4 %3, %4, off offset:48\n s_waitcnt vmcnt(0)" : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(pp) : "memory");
            asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        }
#endif
        bool hit0[KP], hit1[KP], und = false;
        bool dec0[KP], dec1[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            dec0[k] = qbox_filter(lo0[k], hi0[k], A[k], C[k], hit0[k]);
            dec1[k] = qbox_filter(lo1[k], hi1[k], A[k], C[k], hit1[k]);
            und = und || (act[k] && !(dec0[k] && dec1[k]));
            if (STATS) wk.lit_box += act[k] ? (dec0[k] ? 0u : 1u) + (dec1[k] ? 0u : 1u) : 0u;
        }
        // literal arithmetic for undecided lanes behind a wave-uniform branch (six IEEE divisions, almost never needed)
        if (__builtin_expect(__ballot(und) != 0ull, 0)) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (act[k] && !(dec0[k] && dec1[k])) {
                    const float2 D = tabD[o[k]];
                    const f3 O = mk(C[k].x, C[k].y, C[k].z), u = mk(C[k].w, D.x, D.y);
                    if (!dec0[k]) hit0[k] = slab(lo0[k], hi0[k], O, u);
                    if (!dec1[k]) hit1[k] = slab(lo1[k], hi1[k], O, u);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int hiw0 = __float_as_int(hi0[k].w), low0 = __float_as_int(lo0[k].w), cnt0 = hiw0 - low0;
            const int hiw1 = __float_as_int(hi1[k].w), low1 = __float_as_int(lo1[k].w), cnt1 = hiw1 - low1;
            const bool h0 = hit0[k] && act[k], h1 = hit1[k] && act[k];
            const bool hI0 = h0 && hiw0 < 0, hI1 = h1 && hiw1 < 0;
            const bool hL0 = h0 && hiw0 >= 0 && cnt0 > 0, hL1 = h1 && hiw1 >= 0 && cnt1 > 0;
            if (STATS) {
                wk.box += act[k] ? 2u : 0u; wk.nodes += (h0 ? 1u : 0u) + (h1 ? 1u : 0u);
                wk.tris += ((h0 && hiw0 >= 0) ? (uint32_t)cnt0 : 0u) + ((h1 && hiw1 >= 0) ? (uint32_t)cnt1 : 0u);
            }
            const unsigned long long mI0 = __ballot(hI0), mI1 = __ballot(hI1), mL0 = __ballot(hL0), mL1 = __ballot(hL1);
            const int nI0 = __popcll(mI0), nL0 = __popcll(mL0);
            const unsigned int sbits = e[k] & ~kQNodeMask;
            if (hI0) stack[top + lanes_below(mI0)] = sbits | (unsigned int)low0;        // a hit internal node pushes ITS pair of children
            if (hI1) stack[top + nI0 + lanes_below(mI1)] = sbits | (unsigned int)low1;
            top += nI0 + __popcll(mI1);
            if (hL0) leafq[(ltail + (unsigned int)lanes_below(mL0)) & (LCAP - 1)] = make_uint2((unsigned int)low0, (unsigned int)o[k] | (unsigned int)cnt0 << 8);
            if (hL1) leafq[(ltail + (unsigned int)(nL0 + lanes_below(mL1))) & (LCAP - 1)] = make_uint2((unsigned int)low1, (unsigned int)o[k] | (unsigned int)cnt1 << 8);
            ltail += (unsigned int)(nL0 + __popcll(mL1));
            // outstanding entries of the ray: this pair is gone (-1), every pushed pair and leaf entry counts +1: one LDS add per lane
            const int delta = (hI0 ? 1 : 0) + (hI1 ? 1 : 0) + (hL0 ? 1 : 0) + (hL1 ? 1 : 0) - (act[k] ? 1 : 0);
            if (delta != 0) atomicAdd(&pend[o[k]], delta);
        }
        WQ_CHECK(top >= 0 && top <= cap && top <= SCAP, 8, (void)0);
        if (dbg_on) { d_box++; d_boxl += 2u * (unsigned int)n; if ((unsigned int)top > d_maxtop) d_maxtop = (unsigned int)top; }
        WQ_STAMP(cy_box);
    }
#undef WQ_STAMP
#undef WQ_CHECK
    if (dbg_on && lane == 0) {
        unsigned long long *d = st.dbg + 16 * (size_t)((blockIdx.x * blockDim.x + tid) >> 6);
        d[0] = dbg_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = d_box; d[3] = d_boxl; d[4] = d_tri; d[5] = d_tril;
        d[6] = d_rounds; d[7] = d_rays; d[8] = d_serial; d[9] = cy_srv; d[10] = cy_tri; d[11] = cy_box; d[12] = dbg_tdrain; d[13] = d_idle;
        d[14] = d_fetch; d[15] = d_maxtop;
    }
    wf_flush_work<STATS>(fr, wk);
}

}  // namespace rtk
