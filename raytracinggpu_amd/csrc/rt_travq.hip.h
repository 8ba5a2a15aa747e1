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
template <int R> struct QPairs { static constexpr int value = R > 64 ? 2 : RT_TRAVQ_KP; };   // sibling pairs per lane and BOX step: two where a wave keeps 128 rays resident
template <int R> struct QLeafCap { static constexpr int value = 256 * QPairs<R>::value; };       // leaf-queue entries per wave: < 64 before a BOX step, which appends up to 128 per pair
constexpr int kQLeafCap = QLeafCap<64>::value;
// Stack entry (32 bits) = node << 11 | slot << 4: the sibling pair of nodes (node, node + 1) of the ray in slot `slot` (up to 128 slots per
// wave).  Both fields are stored the way they are used: entry & 0x7f0 is the byte offset of the slot's row in the four per-slot tables
// (16-byte rows), (entry >> 6) & ~31 the byte offset of the pair in the node array (32 bytes per node).  A leaf-queue entry is (first triangle,
// count << 11 | slot << 4).  Decoding costs two full-rate instructions per field (and, shift) instead of the shift-and-add forms
// that issue at half rate on gfx950 (tools/ubench/issue_table: v_lshlrev_b32, v_lshl_add_u32, v_and_or_b32 ... take twice the
// issue time of v_and_b32 / v_lshrrev_b32 / v_add_u32 / v_fma_f32).
constexpr int kQNodeShift = 10, kQNodeBits = 22;   // stack entry = node << 10 | slot << 4: the node of an entry is EVEN (a sibling pair), so its lowest bit may share bit 10 with the seventh slot bit (128 resident rays)
constexpr float kQ16FaceCells = 2.f;                // how far outside its real face a fixed-point face can sit, in cells (the quantiser of rt_qnodes.hip.h guarantees it)
constexpr int kQLeafShift = 11;                    // leaf-queue entry / a leaf's kind word = triangle count << 11 | slot << 4 (| flag): counts may be odd, they start above the slot bits
constexpr unsigned int kQSlotMask = 0x7f0u;
constexpr int kQMaxLeaf = 1 << 20;           // triangles per leaf: count << 11 must stay a positive int (the sign says "internal")

// stack capacity: sized so that four waves' carves (+ the cursor) fill 36 KiB (R = 64: 4 workgroups per CU) or less;
// a fuller stack is drained serially (see above), which the cat never needs
#ifndef RT_TRAVQ_SCAP
#define RT_TRAVQ_SCAP 652
#endif
template <int R> struct QStackCap { static constexpr int value = R > 64 ? 800 : RT_TRAVQ_SCAP; };   // (R = 128: twice the rays, two pairs per lane and step)
// 4-wide BOX step (QW, below): a step may append up to 256 stack entries and 256 leaf entries, but it knows how many before it writes one (the
// masks are scalar registers): the carve stays the pair kernel's, and a step that would not fit walks its popped pairs serially instead
#ifndef RT_TRAVQ_SCAP_QW
#define RT_TRAVQ_SCAP_QW RT_TRAVQ_SCAP
#endif
constexpr int kQwStackCap = RT_TRAVQ_SCAP_QW, kQwLeafCap = QLeafCap<64>::value;
// payload word of a quad's child (qquads_kernel): > 0 internal = first child << kQNodeShift (a stack entry without its slot bits; trees below 2^21 nodes),
// < 0 leaf = 1 << 31 | triangle count << 24 | first triangle (leaves of at most 127 triangles, 2^24 triangles), 0 = nothing there
constexpr int kQwLeafShift = 24;
#ifndef RT_TRAVQ_QW_TRIS
#define RT_TRAVQ_QW_TRIS 2
#endif
constexpr int kQwTris = RT_TRAVQ_QW_TRIS;      // triangles per lane and TRI step of the 4-wide kernel

// Per-wave LDS carve.  Four tables of 16-byte rows indexed by ray slot, so that ONE address register (wave base + slot * 16)
// reaches everything a step needs about a ray through the instructions' immediate offsets:
//   A = (1/u.xyz by v_rcp_f32, c0 | +inf if the filter must not decide)      BOX step
//   O = (fl(O.x / u.x) .., int: outstanding stack + leaf-queue entries)      BOX step (the counter shares the row: no address arithmetic)
//   C = (O.xyz, u.x)                                                         TRI step, literal box test
//   D = (u.y, u.z, u64: nearest accepted hit)                                TRI step
template <int R, int SCAP, int LCAP, int NT = 2> struct QCarve {
    // the leaf ring FIRST: its pushes are ds_write2_b32, whose immediate offsets reach 1 KB only -- at the carve's base the ring's address is index * 8 + the wave's base and no
    // vector add is left (the tables and the stack are reached through 16-bit offsets wherever they lie)
    static constexpr int kLeaf = 0;                       // uint2[LCAP]: (first triangle, count << 11 | slot << 4 | flag); QW: (payload word, slot << 4 | flag)
    static constexpr int kTabA = kLeaf + 8 * LCAP;
    static constexpr int kTabO = kTabA + 16 * R;
    static constexpr int kTabC = kTabO + 16 * R;
    static constexpr int kTabD = kTabC + 16 * R;
    static constexpr int kMarks = kTabD + 16 * R;         // u8[64 NT]: TRI-step expansion marks (all zero between steps)
    static constexpr int kStack = kMarks + 64 * NT;       // u32[SCAP]
    static constexpr int kStage = kStack + 4 * SCAP;      // u8[64]: lanes whose registers hold a fetched ray record that has no slot yet
    static constexpr int kBytes = kStage + 64;
    static_assert(kBytes % 16 == 0 && kTabA % 16 == 0 && kStack % 4 == 0, "the next wave's ring and float4 tables start at kBytes");
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
__device__ __forceinline__ int lanes_below2(unsigned long long m0, unsigned long long m1) {   // ... of m0 plus those of m1: v_mbcnt accumulates
    const unsigned int a = __builtin_amdgcn_mbcnt_hi((unsigned int)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m0, 0u));
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m1, a));
}
// number of the four lane masks that contain this lane, minus one if `act` contains it: the masks stay scalar registers and enter
// as carry-ins (five half-rate instructions; written out in C++ the compiler materialises every mask as 0 / 1 in a vector
// register and compares it back into a mask for the ballots)
__device__ __forceinline__ int lane_count4_minus(unsigned long long m0, unsigned long long m1, unsigned long long m2, unsigned long long m3, unsigned long long act) {
    int d;
    asm("v_cndmask_b32_e64 %0, 0, -1, %5\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %1\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, 0, %0, %3\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %4"
        : "=&v"(d) : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(act) : "vcc");
    return d;
}

// x | 1 in the lanes of m (bit 0 of x clear): the mask enters as the carry-in of one add
__device__ __forceinline__ unsigned int or_bit0(unsigned int x, unsigned long long m) {
    unsigned int r;
    asm("v_addc_co_u32_e64 %0, vcc, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m) : "vcc");
    return r;
}

// the same over eight masks (the 4-wide BOX step: four internal and four leaf masks)
__device__ __forceinline__ int lane_count8_minus(const unsigned long long (&a)[4], const unsigned long long (&b)[4], unsigned long long act) {
    int d;
    asm("v_cndmask_b32_e64 %0, 0, -1, %9\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %1\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, 0, %0, %3\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %4\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %5\n\t"
        "v_addc_co_u32_e64 %0, vcc, 0, %0, %6\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %7\n\tv_addc_co_u32_e64 %0, vcc, 0, %0, %8"
        : "=&v"(d) : "s"(a[0]), "s"(a[1]), "s"(a[2]), "s"(a[3]), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]), "s"(act) : "vcc");
    return d;
}

// BoundingBox::intersect (cpu:146-157) through the fused filter (see RayBox in rt_wavefront.hip.h): returns whether
// the filter decided; `hit` is then the reference's result.  A = (r.xyz, c0) with r = v_rcp_f32(u), C = (O.xyz, -);
// c0 = +inf for rays the filter must not decide (0 / denormal / inf / NaN components), which makes every comparison
// below false.  (wf_path and the known-answer entry use this form; wf_travq the centre / half-extent form below.)
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

// ---- the box filter of wf_travq: centre / half-extent form -----------------------------------------------------------------
// The slab test through min / max costs 6 fma + 6 min/max + max3 + min3 per box, and min / max / compares issue at HALF the rate
// of fma / mul / add on gfx950.  With the box stored as centre c and half extent h >= 0 the near and far planes of an axis are
//     k = fma(c, r, -o),   near = fma(-h, |r|, k),   far = fma(h, |r|, k)            (r = v_rcp_f32(u), o = fl(O * r))
// -- nine full-rate instructions and no min / max per axis pair.  Error against the reference's q = RN(RN(bound - O) / u), for
// the plane bound = c* -/+ h* (c* = (lo + hi) / 2, h* = (hi - lo) / 2 exactly; c = RN(c*), h = RN(h*), formed in binary64 by
// the upload): with R = 1 / u,
//     near~ = R (1 + e_r) [ (c* - O + c* e_c - O e_o)(1 + e_1) - h* (1 + e_h) ] (1 + e_2),   |e_r| <= 2^-23, the others <= 2^-24
//  => |near~ - q| <= (2^-23 + 2^-24 + 2 * 2^-24) |q|  +  2^-24 |R| (|c* - O| + |c*| + |O| + |h*|)  (+ second order, + denormal slack)
//                 <= 1.25 * 2^-22 |q|  +  2^-24 |r| (2 |O| + 3 B)                    B >= |lo|, |hi| of every node on that axis.
// max3 / min3 are 1-Lipschitz and the reference's per-axis min / max IS the near / far plane (RN is monotone, h >= 0), so
// tn~, tf~ carry the same bound with |q| read off the computed values; band = kRel (|tn~| + |tf~|) + c0 with kRel = 2^-21
// (1.6 x the relative part) and c0 = 1.125 * 2^-23 max_k |r_k| (2 |O_k| + 3 B_k) + 2e-35 (2 x the absolute part + slack)
// covers both ends, the rounding of d and of the band itself.  Decide only when |d| > band; c0 = +inf (rays with a 0 / denormal /
// huge / NaN component, scenes whose boxes are not finite, ordered and below 1e8 in magnitude: Scene::fast_box) makes both
// comparisons false, and so does any NaN.  Within those guards no intermediate overflows: |c r|, |h r| < 1e8 * 1e30.
struct RayBoxC { float rx, ry, rz, ox, oy, oz, c0; };
__device__ __forceinline__ RayBoxC ray_box_c(f3 O, f3 u, f3 bm, bool fast_box) {   // bm: per axis max |bound| over the boxes the ray will meet
    RayBoxC b;
    b.rx = __builtin_amdgcn_rcpf(u.x); b.ry = __builtin_amdgcn_rcpf(u.y); b.rz = __builtin_amdgcn_rcpf(u.z);
    b.ox = O.x * b.rx; b.oy = O.y * b.ry; b.oz = O.z * b.rz;
    const float omax = vmax3abs(b.ox, b.oy, b.oz);
    const float wx = fabsf(b.rx) * fmaf(3.f, bm.x, 2.f * fabsf(O.x)), wy = fabsf(b.ry) * fmaf(3.f, bm.y, 2.f * fabsf(O.y)), wz = fabsf(b.rz) * fmaf(3.f, bm.z, 2.f * fabsf(O.z));
    const float w = vmax3(wx, wy, wz);
    const float umin = fminf(fminf(fabsf(u.x), fabsf(u.y)), fabsf(u.z)), umax = fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fabsf(u.z));
    const bool safe = fast_box && umin > kTiny && umax < kBig && omax < kBig && w < kBig;      // false for 0, denormal, inf, NaN anywhere
    b.c0 = safe ? fmaf(w, 0x1.2p-23f, 2e-35f) : __builtin_inff();
    return b;
}
// n0 = (c.xyz, -), n1 = (h.xyz, -), A = (r.xyz, c0), Oo = (o.xyz, -): `hit` / `miss` are the reference's result when one of them is set
__device__ __forceinline__ void cbox_filter(const float4 n0, const float4 n1, const float4 A, const float4 Oo, bool &hit, bool &miss) {
    const float kx = fmaf(n0.x, A.x, -Oo.x), ky = fmaf(n0.y, A.y, -Oo.y), kz = fmaf(n0.z, A.z, -Oo.z);
    const float tn = vmax3(fmaf(-n1.x, fabsf(A.x), kx), fmaf(-n1.y, fabsf(A.y), ky), fmaf(-n1.z, fabsf(A.z), kz));
    const float tf = vmin3(fmaf(n1.x, fabsf(A.x), kx), fmaf(n1.y, fabsf(A.y), ky), fmaf(n1.z, fabsf(A.z), kz));
    const float d = tf - tn;
    const float band = fmaf(fabsf(tf), kRel, fmaf(fabsf(tn), kRel, A.w));
    hit = d > band;
    miss = d < -band;
}
// the same planes, returning d = far - near and the band instead of the two decisions (the fixed-point BOX step compares d with two thresholds)
__device__ __forceinline__ void cbox_dband(const float4 n0, const float4 n1, const float4 A, const float4 Oo, float &d, float &band) {
    const float kx = fmaf(n0.x, A.x, -Oo.x), ky = fmaf(n0.y, A.y, -Oo.y), kz = fmaf(n0.z, A.z, -Oo.z);
    const float tn = vmax3(fmaf(-n1.x, fabsf(A.x), kx), fmaf(-n1.y, fabsf(A.y), ky), fmaf(-n1.z, fabsf(A.z), kz));
    const float tf = vmin3(fmaf(n1.x, fabsf(A.x), kx), fmaf(n1.y, fabsf(A.y), ky), fmaf(n1.z, fabsf(A.z), kz));
    d = tf - tn;
    band = fmaf(fabsf(tf), kRel, fmaf(fabsf(tn), kRel, A.w));
}
// centre / half extent of one axis as the upload and the device-side refit form them (binary64 sum / difference: exact; one rounding)
__host__ __device__ inline float box_centre(float lo, float hi) { return (float)(((double)lo + (double)hi) * 0.5); }
__host__ __device__ inline float box_half(float lo, float hi) { return (float)(((double)hi - (double)lo) * 0.5); }

// moller_trumbore (cpu:226-236) + the acceptance test of the leaf loop (cpu:301): beta/gamma through the filter,
// undecided lanes by the literal divisions, t always by the exact division.
// `how` reports the route (callers that do not look at it pay nothing): 0 filter rejected, 1 filter accepted, 2 literal divisions;
// + 4 when the barycentrics were accepted and t was computed.
//
// Filter.  b = bn * rcp(det), g = gn * rcp(det) differ from the reference's beta' = RN(bn / det), gamma' by at most 2^-22 of their
// magnitude (+ denormal slack) whenever rd = rcp(det) is a NORMAL number (one v_cmp_class: excludes det = 0, denormal or beyond
// 8.5e37, inf, NaN at either end).  cpu:232-235 accept iff 0 <= beta' <= 1, 0 <= gamma' <= 1 and fl(beta' + gamma') <= 1; for
// finite values the two "<= 1" follow from the rest.  With E = 2^-18 and s = fl(b + g):
//   accept  iff  min(b, g) >= E  and  s <= 1 - 3E     (then 0 < beta', gamma' and beta' + gamma' <= s + 2^-24 + 2^-21 < 1)
//   reject  iff  min(b, g) < -E  or   s > 1 + 3E      (a negative value stays negative under a relative error; and if the reference
//                                                      accepted, beta' + gamma' <= 1 + 2^-24 would give s < 1 + 2^-20)
// everything else -- including every NaN: v_min drops one, but s keeps it and then neither line holds, or the surviving operand is
// itself < -E, which the reference rejects too -- takes the literal expression.  Constant thresholds instead of the earlier
// per-value bands: 7 half-rate + 1 full-rate instruction instead of 12 + 7; the undecided band is 2^-16 wide instead of 2^-20,
// still one test in ~10^4.
// `valid` = false: the lane holds no test (padding of a TRI step); it takes neither the literal nor the division branch.
__device__ __forceinline__ bool qtri_test(const float4 q0, const float4 q1, const float4 q2, const f3 Oo, const f3 uo,
                                          const float tri_tmin, float &t_out, int &how, const bool valid = true) {
    const f3 A = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
    const f3 N = mk(q2.y, q2.z, q2.w);
    const float det = dot(uo, N);
    const f3 AO = A - Oo;
    const f3 c = cross(AO, uo);
    const float bn = dot(e2, c);
    const float gn = -dot(e1, c);
    const float rd = __builtin_amdgcn_rcpf(det);
    const float b = bn * rd, g = gn * rd;
    constexpr float kE = 0x1p-18f;
    const float sum = b + g;
    const float mbg = vmin(b, g);
    const bool trust = __builtin_amdgcn_classf(rd, 0x108);      // rd is +-normal
    const bool reject = trust && (mbg < -kE || sum > 1.f + 3.f * kE);
    bool ok = trust && mbg >= kE && sum <= 1.f - 3.f * kE && valid;
    how = ok ? 1 : 0;
    if (!reject && !ok && valid) {              // undecided: the literal tests (rare)
        asm volatile("rt_mark_trilit_begin_%=:" ::);
        how = 2;
        if (det != 0) {                         // cpu:230
            const float beta = bn / det;
            const float gamma = gn / det;
            ok = (0 <= beta && beta <= 1) && (0 <= gamma && gamma <= 1) && (beta + gamma <= 1);
        }
        asm volatile("rt_mark_trilit_end_%=:" ::);
    }
    if (!ok) return false;
    how |= 4;                                                       // the lane reaches the division (callers that do not look at `how` pay nothing)
    // (code-object markers: tools/static_counts.py prices this block by how often a step enters it; the operands tie the labels to
    // the computation -- no instruction is emitted for them)
    float dd = det;
    asm volatile("rt_mark_tdiv_begin_%=:" : "+v"(dd));
    float t = dot(AO, N) / dd;
    asm volatile("rt_mark_tdiv_end_%=:" : "+v"(t));
    t_out = t;
    return t > (tri_tmin > 0.f ? tri_tmin : 0.f) && t < 1e9f;       // cpu:235 (t > 0) and cpu:301 (t > 1e-4); 1e9f = INF narrowed (cpu:283)
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
//
// Step counters of the counting (STATS) instantiation, fr.work[8..15] (bench.py prices the vector-issue roofline with them and the
// static per-step instruction counts of the production code object, tools/static_counts.py): loop iterations, refill passes
// entered, refill rounds, queue fetches, TRI steps, BOX steps, literal-box fall-backs taken, serial drains; then the conditionally
// executed blocks of the steps, counted per entry: the t-division block of a triangle test (some lane accepted the barycentrics), the
// first and the second leaf-queue push of a BOX step.
//
// QW (round 5; implies QN): the BOX step is FOUR boxes wide.  An entry still names the sibling pair (c, c + 1), but the step reads the pair's QUAD from
// sc.nodesw -- the children of c and the children of c + 1 (a leaf child of the pair stands for itself), four 16-byte fixed-point records = the same four
// loads a step of the float pairs issues -- tests the four boxes and pushes one entry per hit internal grandchild: every other level of the tree is
// never tested, a ray's chain of dependent steps is half as long and a frame runs half as many BOX steps.  Exactness: the fixed-point boxes contain the
// real ones and the reference's test is monotone along nested boxes (rt_qnodes.hip.h), so the leaves the reference reaches are exactly the leaves whose
// OWN box its test hits.  Leaves are decided as the fixed-point pairs decide them: a leaf hit by more than the boxes' enlargement (one threshold per RAY: its band + four
// cells along its steepest axis) is hit by the reference; a leaf in between is queued with a flag, and a triangle ACCEPTED in a flagged leaf counts only if the
// reference's own test of the leaf's real box (slab_filtered on its (lo, hi) record, behind a vote: a few steps in a hundred) says hit.  (The first versions tested the
// real box of EVERY leaf entry in the TRI step, two gathers and 26 vector instructions per step: 8 % of the cat's frame and a third of a frame of 524 288 triangles,
// profiles/round5/ab_wide_nodes.txt.)  Work counters of this instantiation differ from the oracle's by construction (no test of the skipped level, a superset of
// internal nodes entered): the counter tests use the binary instantiation, this one is held to frames and ray counts.
//
// ANY-HIT (round 6; the fixed-point instantiations QN / QW).  A shadow ray's record carries a bound (PQ_ANYHIT, wf_anyhit_bound): an ACCEPTED triangle with t at or below
// it settles cpu:615 whatever else the ray would hit.  The TRI step that accepts such a triangle marks the slot dead by writing -inf over the ray's band (row A, .w): every
// later BOX step then excludes all boxes of the ray's remaining entries with the comparison it makes anyway (d < -band = +inf), nothing is pushed, the entries drain and
// the slot retires with what it has -- a hit at or below the bound, which is all the closing launch looks at.  Leaf entries of a dead slot that are already queued
// shrink to one position (their first triangle).  The bound lives in a register of the lane that owns the slot.  The float-pair instantiation (the counting one: its work counters are the oracle's) never stops early; frames are bit-identical either way.
template <bool STATS, int R, bool LDSN, bool LDSV, bool QN = false, bool QW = false>
__global__ __launch_bounds__((LDSN || LDSV) ? 1024 : kQBlock, (LDSN || LDSV || kQBlock != 256 || QPairs<R>::value > 1 || (QW && kQwTris > 2)) ? 4 : 5) void wf_travq(const Scene sc, const Frame fr, const WfState st, const int cap, const int n_lds, const int kLow, const int kMinFree) {
    // kLow: refill while the stack holds fewer entries (sibling pairs) than this (default 48); kMinFree: ... and at least this
    // many slots are free, or the stack is short (default R / 4)
    static_assert(!QW || (QN && R == 64 && !LDSN && !LDSV), "the 4-wide step reads fixed-point quads through L1 / L2, one ray slot per lane");
    constexpr int SCAP = QW ? kQwStackCap : QStackCap<R>::value, LCAP = QW ? kQwLeafCap : QLeafCap<R>::value;
    constexpr int NB = R > 64 ? 2 : 1;                      // ray slots per lane ("banks"): lane l owns slots l and, with 128 resident rays, l + 64
    constexpr int NT = QW ? kQwTris : 2;                    // triangles per lane and TRI step
    using Carve = QCarve<R, SCAP, LCAP, NT>;
    static_assert(R <= 128 && (R & (R - 1)) == 0, "ray slots are owned by lanes: one per lane, or two");
    extern __shared__ __attribute__((aligned(16))) unsigned char travq_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: the carve's base address lives in a scalar register
    const int wpb = (int)(blockDim.x >> 6);
    unsigned char *const wl = travq_smem + wib * Carve::kBytes;
    int *const blk_cur = reinterpret_cast<int *>(travq_smem + wpb * Carve::kBytes);
    float4 *const lnodes = reinterpret_cast<float4 *>(travq_smem + wpb * Carve::kBytes + 16);
    float4 *const lverts = lnodes + (LDSN ? 2 * n_lds : 0);        // LDSV: every vertex (x, y, z, -)
    unsigned char *const marks = wl + Carve::kMarks;
    unsigned int *const stack = reinterpret_cast<unsigned int *>(wl + Carve::kStack);
    uint2 *const leafq = reinterpret_cast<uint2 *>(wl + Carve::kLeaf);
    unsigned char *const sidx = wl + Carve::kStage;
    // row `sb` (= slot * 16, the form entries carry it in) of the four per-slot tables
    auto rowA = [&](unsigned int sb) -> float4 & { return *reinterpret_cast<float4 *>(wl + Carve::kTabA + sb); };
    auto rowO = [&](unsigned int sb) -> float4 & { return *reinterpret_cast<float4 *>(wl + Carve::kTabO + sb); };
    auto rowC = [&](unsigned int sb) -> float4 & { return *reinterpret_cast<float4 *>(wl + Carve::kTabC + sb); };
    auto rowD = [&](unsigned int sb) -> float4 & { return *reinterpret_cast<float4 *>(wl + Carve::kTabD + sb); };
    auto pend = [&](unsigned int sb) -> int * { return reinterpret_cast<int *>(wl + Carve::kTabO + sb + 12); };
    auto best = [&](unsigned int sb) -> unsigned long long * { return reinterpret_cast<unsigned long long *>(wl + Carve::kTabD + sb + 8); };
    auto bandw = [&](unsigned int sb) -> float & { return *reinterpret_cast<float *>(wl + Carve::kTabA + sb + 12); };      // row A's .w: the ray's band; -inf = the slot is dead (any-hit)
    const unsigned int my_sb0 = (unsigned int)lane << 4;           // lane l owns ray slot l (bank 0) and l + 64 (bank 1, R = 128): row offsets
    if (tid == 0) *blk_cur = 0;
    for (int k = 0; k < NT; ++k) marks[lane + 64 * k] = 0;
    for (int b = 0; b < NB; ++b) if (lane + 64 * b < R) *pend(my_sb0 + 1024u * b) = 0;
    if (LDSN) for (int k = tid; k < 2 * n_lds; k += (int)blockDim.x) lnodes[k] = sc.nodesb[k];
    if (LDSV) for (int k = tid; k < sc.n_verts; k += (int)blockDim.x) lverts[k] = sc.verts[k];
    __syncthreads();

    // breadth-first order from index 1 (0 is padding), 32 bytes per node: (centre.xyz, payload) (half extent.xyz, kind);
    // internal node: payload = first child << 10 (even; the other child is next to it), kind < 0;  leaf: payload = first triangle,
    // kind = triangle count << 11 (0 for an empty leaf)
    const unsigned char *const nodes = reinterpret_cast<const unsigned char *>(sc.nodesb);
    const size_t blk_base = (size_t)blockIdx.x * (size_t)st.slots_per_block;
    const int blk_n = st.slots_per_block;
    int stage_n = 0, stage_used = 0;              // wave-uniform: staged records and how many of them have been given a slot
    float4 sp0 = make_float4(0, 0, 0, 0);         // staged record of this lane (registers; sidx[k] = lane of the k-th staged record)
    float2 sp1 = make_float2(0, 0);
    int spf = 0;
    float spb = -__builtin_inff();                // ... and its any-hit bound
    const int root_hiw = __float_as_int(sc.root_hi.w);
    int path_[NB];                                // the path index of the ray in each of the lane's slots, -1 = free
    for (int b = 0; b < NB; ++b) path_[b] = -1;
    float bound_[NB];                             // ... and its any-hit bound (-inf: none; only shadow rays carry one)
    for (int b = 0; b < NB; ++b) bound_[b] = -__builtin_inff();
    int top = 0;                                  // wave-uniform: entries on the stack
    unsigned int lhead = 0, ltail = 0;            // wave-uniform: leaf-queue cursors (monotonic)
    bool drained = false;
    Work wk;
    unsigned int n_iter = 0, n_refill = 0, n_round = 0, n_fetch = 0, n_tri = 0, n_box = 0, n_lit = 0, n_serial = 0;   // STATS: step counters (wave-uniform)
    unsigned int n_stop = 0;                                        // STATS: TRI steps in which some lane stopped its ray (any-hit)
    unsigned int n_tdiv = 0, n_lpush = 0, n_lpush2 = 0;             // STATS: conditionally executed blocks of the steps (entered when any lane needs them)
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
    // code-object markers: labels, not instructions (tools/static_counts.py counts the instructions between them)
#ifdef RT_NO_MARKS
#define WQ_MARK(name) do { } while (0)
#else
#define WQ_MARK(name) asm volatile("rt_mark_" name "_%=:" ::)
#endif

    // the sibling nodes of a stack entry at byte offset `off` (the pair is one 64-byte line): LDS for the staged top of the
    // tree (n_lds is even), L1/L2 otherwise
    auto load_pair = [&](unsigned int off, float4 &c0, float4 &h0, float4 &c1, float4 &h1) {
        const float4 *gp = reinterpret_cast<const float4 *>(nodes + off);
        if (LDSN) {
            const bool inl = off < (unsigned int)n_lds * 32u;
            const float4 *lp = reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(lnodes) + off);
            if (inl) { c0 = lp[0]; h0 = lp[1]; c1 = lp[2]; h1 = lp[3]; }
            if (__ballot(!inl) != 0ull) { if (!inl) { c0 = gp[0]; h0 = gp[1]; c1 = gp[2]; h1 = gp[3]; } }
        } else {
            c0 = gp[0]; h0 = gp[1]; c1 = gp[2]; h1 = gp[3];
        }
    };
    // stack nearly full: walk the subtree of one popped entry serially with the stackless (skip-pointer) node array
    auto drain_serial = [&](unsigned int sb, int node) {
        const float4 C = rowC(sb), D = rowD(sb);
        const f3 O = mk(C.x, C.y, C.z), u = mk(C.w, D.x, D.y);
        const RayInv ri = ray_inv(u);
        const int xt = sc.q2thr[node];                                // the same node in the stackless array
        const float4 h0 = sc.nodes[2 * xt + 1];
        const int end = __float_as_int(h0.w) >= 0 ? xt + 1 : __float_as_int(sc.nodes[2 * xt].w);
        for (int x = xt; x < end;) {
            const float4 lo = sc.nodes[2 * x], hi = sc.nodes[2 * x + 1];
            const int hiw = __float_as_int(hi.w), low = __float_as_int(lo.w);
            const bool hit = slab_filtered(lo, hi, O, u, ri);
            if (STATS) { wk.box++; if (hit) wk.nodes++; }
            if (hit && hiw >= 0) {
                if (STATS) wk.tris += (uint32_t)(hiw - low);
                for (int i = low; i < hiw; ++i) {
                    const float4 *tp = sc.tri + 3 * (size_t)i;
                    float t;
                    if (qtri_test(tp[0], tp[1], tp[2], O, u, fr.tri_tmin, t))
                        atomicMin(best(sb), (unsigned long long)__float_as_uint(t) << 32 | (unsigned int)i);
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
        stage_n = __builtin_amdgcn_readfirstlane(stage_n);
        stage_used = __builtin_amdgcn_readfirstlane(stage_used);
        if (STATS) n_iter++;
        WQ_MARK("head");
        // =============================== retire + refill ===============================
        if (top < kLow) {
            if (STATS) n_refill++;
            WQ_MARK("refill_begin");
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const unsigned int sbk = my_sb0 + 1024u * b;
                if (lane + 64 * b < R && path_[b] >= 0) {
                    if (*pend(sbk) == 0) {
                        st.M[path_[b]] = *best(sbk);  // always: the emitter does not initialise M (WF_NOHIT = no triangle accepted)
                        path_[b] = -1;
                        if (QN) bound_[b] = -__builtin_inff();
                    }
                }
            }
            // refill free slots.  The workgroup owns a spatially scrambled, contiguous share of the traversal queue; its
            // waves take 64 slots at a time through an LDS cursor, flag and record in ONE round trip, and park the rays that
            // need traversal in registers (the LDS staging index says which lane holds the k-th of them), from where free slots are filled.
            for (int round = 0; round < 6 && top < kLow; ++round) {
                unsigned long long freem_[NB];
                int n_free = 0;
#pragma unroll
                for (int b = 0; b < NB; ++b) { freem_[b] = __ballot(lane + 64 * b < R && path_[b] < 0); n_free += __popcll(freem_[b]); }
                if (n_free == 0 || (n_free < kMinFree && top >= 64)) break;
                if (stage_used >= stage_n) {
                    if (drained) break;
                    WQ_MARK("fetch_begin");
                    int base = 0;
                    if (lane == 0) base = atomicAdd(blk_cur, 64);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + 64 >= blk_n) { drained = true; if (dbg_on && !dbg_tdrain) dbg_tdrain = __builtin_amdgcn_s_memrealtime(); }
                    spf = 0;
                    if (base + lane < blk_n) {
                        const size_t q = (size_t)blk_base + (size_t)(base + lane);
                        const float4 qb = st.QR[2 * q + 1];             // the slot's 32-byte record: flag and ray in one round trip
                        sp0 = st.QR[2 * q]; sp1 = make_float2(qb.x, qb.y);
                        spf = wq_live(__float_as_int(qb.z), st.epoch, st.nonce) ? wf_slot_to_path(st, (int)q) + 1 : 0;   // ray + 1 if the slot's record is live for this launch
                        if (QN) spb = (__float_as_int(qb.z) & PQ_ANYHIT) ? qb.w : -__builtin_inff();
                    }
                    const unsigned long long am = __ballot(spf != 0);
                    if (spf != 0) sidx[lanes_below(am)] = (unsigned char)lane;
                    __builtin_amdgcn_wave_barrier();
                    stage_n = __popcll(am);
                    stage_used = 0;
                    if (STATS) n_fetch++;
                    if (dbg_on) d_fetch++;
                    WQ_MARK("fetch_end");
                    if (stage_n == 0) continue;
                }
                if (STATS) n_round++;                                  // hand-off rounds: staged rays go to free slots
                WQ_MARK("round_begin");
                // the root box was tested when the ray was emitted (wf_emit_ray): start with what is below it
                const int first = __float_as_int(sc.root_lo.w), cnt = root_hiw - first;
                const bool work = root_hiw < 0 || cnt > 0;
                bool leaf_root_done = false;
#pragma unroll
                for (int b = 0; b < NB; ++b) {                         // the lane's slots, one bank after the other
                    const int avail = stage_n - stage_used;
                    const int nf = __popcll(freem_[b]);
                    if (avail <= 0 || nf == 0) continue;
                    const unsigned int sbk = my_sb0 + 1024u * b;
                    const int take = nf < avail ? nf : avail;
                    const int rank = lanes_below(freem_[b]);
                    const bool got = __builtin_amdgcn_inverse_ballot_w64(freem_[b]) && rank < take;
                    WQ_CHECK(!got || stage_used + rank < 64, 32, (void)0);
                    const int src = got ? (int)sidx[(stage_used + rank) & 63] : lane;   // the staged record lives in that lane's registers
                    const float4 r0 = make_float4(__shfl(sp0.x, src, 64), __shfl(sp0.y, src, 64), __shfl(sp0.z, src, 64), __shfl(sp0.w, src, 64));
                    const float2 r1 = make_float2(__shfl(sp1.x, src, 64), __shfl(sp1.y, src, 64));
                    const int rf = __shfl(spf, src, 64);
                    const float rb_any = QN ? __shfl(spb, src, 64) : 0.f;
                    if (got) {
                        const f3 O = mk(r0.x, r0.y, r0.z), u = mk(r0.w, r1.x, r1.y);
                        const RayBoxC rb = ray_box_c(O, u, mk(sc.bmx, sc.bmy, sc.bmz), sc.fast_box != 0);
                        if (QN) {     // the same planes in grid units: k = fma(cq, r s, -(O - g) r)
                            const float qrx = rb.rx * sc.qsx, qry = rb.ry * sc.qsy, qrz = rb.rz * sc.qsz;
                            const float qox = (O.x - sc.qgx) * rb.rx, qoy = (O.y - sc.qgy) * rb.ry, qoz = (O.z - sc.qgz) * rb.rz;
                            float aw = 4.f * rb.c0;
                            if (QW) {
                                // ONE band per ray instead of one per box: every plane value of a box on the 16-bit grid is bounded by |o'| + (cq + hq) |r'| <= |o'| + 2^17 |r'|,
                                // so B = 2 kRel Tm (1 + 2^-10) + 4 c0 with Tm = the largest such bound over the axes is at least cbox_dband's band for ANY box: d < -B excludes.
                                // (About an eighth of a cell along the ray's steepest axis; the boxes carry up to two cells of slack per face.)
                                const float tm = vmax3(fmaf(0x1p17f, fabsf(qrx), fabsf(qox)), fmaf(0x1p17f, fabsf(qry), fabsf(qoy)), fmaf(0x1p17f, fabsf(qrz), fabsf(qoz)));
                                aw = fmaf(tm, 2.f * kRel * (1.f + 0x1p-10f), aw);
                            }
                            rowA(sbk) = make_float4(qrx, qry, qrz, aw);
                            rowO(sbk) = make_float4(qox, qoy, qoz, __int_as_float(work ? 1 : 0));
                        } else {
                        rowA(sbk) = make_float4(rb.rx, rb.ry, rb.rz, rb.c0);
                        rowO(sbk) = make_float4(rb.ox, rb.oy, rb.oz, __int_as_float(work ? 1 : 0));      // .w: one outstanding entry
                        }
                        rowC(sbk) = r0;
                        if (QN) bound_[b] = rb_any;
                        rowD(sbk) = make_float4(r1.x, r1.y, __uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu));   // .zw: WF_NOHIT
                        path_[b] = rf - 1;
                        WQ_CHECK(path_[b] >= 0 && path_[b] < 2 * st.n_paths, 1, path_[b] = 0);
                    }
                    stage_used += take;
                    bool pushes = got;
                    if (QN && root_hiw < 0) {     // a ray with a zero / denormal / huge component: the box test is not monotone for it, so it never meets the fixed-point pairs
                        const bool serial = got && !(rowA(sbk).w < __builtin_inff());
                        if (__builtin_expect(__ballot(serial) != 0ull, 0)) {
                            if (serial) { *pend(sbk) = 0; drain_serial(sbk, 2); drain_serial(sbk, 3); }   // the root's children and everything below them, literally; the slot retires at the next pass
                        }
                        pushes = got && !serial;
                    }
                    const unsigned long long gm = __ballot(pushes);
                    if (dbg_on) { d_rounds++; d_rays += (unsigned int)__popcll(gm); }
                    if (root_hiw < 0) {
                        if (pushes) stack[top + lanes_below(gm)] = 2u << kQNodeShift | sbk;   // the root (node 1) has the children 2, 3
                        top += __popcll(gm);
                    } else {                           // the root is a leaf
                        if (cnt > 0) {
                            if (got) {
                                leafq[(ltail + (unsigned int)lanes_below(gm)) & (LCAP - 1)] = QW ? make_uint2(0x80000000u | (unsigned int)cnt << kQwLeafShift | (unsigned int)first, sbk)
                                                                                                 : make_uint2((unsigned int)first, (unsigned int)cnt << kQLeafShift | sbk);
                                if (STATS) wk.tris += (uint32_t)cnt;
                            }
                            ltail += (unsigned int)__popcll(gm);
                        }
                        leaf_root_done = true;
                    }
                }
                if (leaf_root_done) break;             // at most R leaf entries per pass: the TRI steps below drain them
                WQ_MARK("round_end");
            }
        }
        WQ_STAMP(cy_srv);
        WQ_MARK("refill_end");
        // =============================== TRI step: NT triangles per lane ===============================
        // NT = 2 (128 triangles per step).  -DRT_TRAVQ_QW_TRIS=3 / 4 give the 4-wide kernel 192 / 256: with 256 the 64 entries a step reads are the 64 it consumes and TRI steps fall
        // from 736 K to 425 K per frame -- measured SLOWER twice (profiles/round5/ab_wide_nodes.txt): nine / twelve triangle records per lane want 101 / 124 registers (68 spills at a
        // 64-register limit), the launch owns the register file and the other sub-frame's wf_advance no longer fits beside it: the launch alone 0.173 -> 0.167 ms, the frame 0.866 -> 0.882 / 0.953 ms.
        const unsigned int lcount = ltail - lhead;
        // a TRI step as soon as 32 leaf entries wait (about the 128 triangles a step takes: the cat's leaves hold 3.9 on average).  Until round 6 it waited for 64: the same frame
        // time without any-hit (0.848 vs 0.844 ms), but a shadow ray stops the sooner its leaves are tested -- with any-hit 0.830 (64), 0.825 (48), 0.822 (32), 0.829 (24), 0.845 (16)
#ifndef RT_TRAVQ_TRI_MIN
#define RT_TRAVQ_TRI_MIN 32
#endif
        if (lcount >= (unsigned int)RT_TRAVQ_TRI_MIN || (top == 0 && lcount > 0u)) {
            if (STATS) n_tri++;
            WQ_MARK("tri_begin");
            constexpr unsigned int LIM = 64u * NT;                   // triangles a step takes
            const unsigned int m = lcount < 64u ? lcount : 64u;
            uint2 E = make_uint2(0u, 0u);
            if ((unsigned int)lane < m) E = leafq[(lhead + (unsigned int)lane) & (LCAP - 1)];
            unsigned int c = QW ? (E.x >> kQwLeafShift) & 0x7fu : E.y >> kQLeafShift;               // >= 1 for queued entries, 0 beyond them (QW: the entry is (payload word, slot << 4 | flag))
            if (QN) {              // any-hit: the entry of a dead slot (band = -inf, nothing else writes that value: bands are positive, +inf or -- a ray with a NaN -- NaN) shrinks to ONE
                                   // position -- the expansion below wants every entry to own one; what that position accepts joins the slot's minimum, which stays at or below the bound
                const unsigned int band_bits = *reinterpret_cast<const unsigned int *>(wl + Carve::kTabA + (E.y & kQSlotMask) + 12);
                c = band_bits == 0xff800000u ? (c < 1u ? c : 1u) : c;
            }
            if (QW) E.x &= (1u << kQwLeafShift) - 1u;
            const unsigned int incl = wave_incl_scan(c);
            const unsigned int P = incl - c;                         // position of this entry's first triangle
            const bool part = c > 0u && P < LIM;
            const unsigned int all = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
            const unsigned int total = all < LIM ? all : LIM;
            if (part) marks[P] = 1;
            __builtin_amdgcn_wave_barrier();
            unsigned int mk_[NT];
            unsigned long long B_[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k) { mk_[k] = marks[lane + 64 * k]; B_[k] = __ballot(mk_[k] != 0u); }
            __builtin_amdgcn_wave_barrier();
            if (part) marks[P] = 0;
            // entry whose triangle range covers position lane + 64 k: marks at or below the position, minus one (a mark byte is the lane's own bit of B_[k]);
            // ds_bpermute takes the source lane as a byte address
            int j_[NT], i_[NT];
            unsigned int y_[NT], o_[NT];
            bool t_[NT];
            int before = 0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                j_[k] = (before + lanes_below(B_[k]) + (int)mk_[k] - 1) << 2;
                before += __popcll(B_[k]);
                const unsigned int fk = (unsigned int)__builtin_amdgcn_ds_bpermute(j_[k], (int)E.x), Pk = (unsigned int)__builtin_amdgcn_ds_bpermute(j_[k], (int)P);
                y_[k] = (unsigned int)__builtin_amdgcn_ds_bpermute(j_[k], (int)E.y);
                t_[k] = (unsigned int)lane + 64u * k < total;
                o_[k] = y_[k] & kQSlotMask;                           // the owner's table row (positions beyond `total` read some entry's row: harmless)
                i_[k] = t_[k] ? (int)(fk + ((unsigned int)lane + 64u * k - Pk)) : 0;
                WQ_CHECK(i_[k] >= 0 && i_[k] < sc.n_tris, 2, i_[k] = 0);
            }
            WQ_CHECK(ltail - lhead <= (unsigned int)LCAP, 16, (void)0);
            float4 q0_[NT], q1_[NT], q2_[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (LDSV) {
                    const int4 ix = sc.tidx[i_[k]];
                    const float4 va = lverts[ix.x], vb = lverts[ix.y], vc = lverts[ix.z];
                    const f3 A = mk(va.x, va.y, va.z);
                    const f3 e1 = mk(vb.x, vb.y, vb.z) - A, e2 = mk(vc.x, vc.y, vc.z) - A, N = cross(e1, e2);   // cpu:227-229
                    q0_[k] = make_float4(A.x, A.y, A.z, e1.x); q1_[k] = make_float4(e1.y, e1.z, e2.x, e2.y); q2_[k] = make_float4(e2.z, N.x, N.y, N.z);
                } else {
                    // 32-bit byte offsets: the loads take the scalar base + vector offset form (n_tris * 48 < 2^32 is checked by the upload)
                    const float4 *tp = reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(sc.tri) + (unsigned int)i_[k] * 48u);
                    q0_[k] = tp[0]; q1_[k] = tp[1]; q2_[k] = tp[2];
                }
            }
            float tt_[NT];
            bool ok_[NT];
            float4 C_[NT]; float2 D_[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k) { C_[k] = rowC(o_[k]); D_[k] = *reinterpret_cast<const float2 *>(&rowD(o_[k])); }
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                int how;
                ok_[k] = qtri_test(q0_[k], q1_[k], q2_[k], mk(C_[k].x, C_[k].y, C_[k].z), mk(C_[k].w, D_[k].x, D_[k].y), fr.tri_tmin, tt_[k], how, t_[k]);
                if (STATS) {
                    wk.lit_tri += (t_[k] && (how & 3) == 2) ? 1u : 0u;
                    n_tdiv += __ballot((how & 4) != 0) != 0ull ? 1u : 0u;   // division blocks some lane entered
                }
            }
            if (QN) {              // a triangle accepted in a flagged leaf counts only if the reference's test of the leaf's real box says hit (rare: behind a vote)
                bool ch_[NT], any = false;
#pragma unroll
                for (int k = 0; k < NT; ++k) { ch_[k] = ok_[k] && (y_[k] & 1u) != 0u; any = any || ch_[k]; }
                if (__builtin_expect(__ballot(any) != 0ull, 0)) {
                    if (STATS) n_lit++;                              // (the counter of the float pairs' literal-box blocks: they never run in this instantiation)
                    if (any) {                                       // one region for the lane's triangles: the table lookups, then the boxes, of all of them in flight together
                        WQ_MARK("lflag_begin");
                        float4 lo_[NT], hi_[NT];
#pragma unroll
                        for (int k = 0; k < NT; ++k) { const size_t x = 2 * (size_t)(ch_[k] ? i_[k] : 0); lo_[k] = sc.leaflh[x]; hi_[k] = sc.leaflh[x + 1]; }   // the leaf's (lo, hi), kept per triangle: one hop
#pragma unroll
                        for (int k = 0; k < NT; ++k) {
                            const f3 Or = mk(C_[k].x, C_[k].y, C_[k].z), ur = mk(C_[k].w, D_[k].x, D_[k].y);
                            const bool hitk = slab_filtered(lo_[k], hi_[k], Or, ur, ray_inv(ur));
                            if (ch_[k]) ok_[k] = hitk;
                        }
                        WQ_MARK("lflag_end");
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NT; ++k) if (ok_[k]) atomicMin(best(o_[k]), (unsigned long long)__float_as_uint(tt_[k]) << 32 | (unsigned int)i_[k]);
            if (QN) {              // any-hit: every lane looks at ITS slot's nearest accepted hit (after the mins above: LDS operations stay in order); at or below the ray's bound ends the ray
                WQ_MARK("stop_begin");   // (false for a ray without a bound, -inf, and for a slot without a hit: the high word of WF_NOHIT is a NaN)
                bool any = false;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const unsigned int sbk = my_sb0 + 1024u * b;
                    const bool stop = __uint_as_float(reinterpret_cast<const unsigned int *>(best(sbk))[1]) <= bound_[b];   // (a lane without a ray in the slot holds -inf)
                    if (stop) bandw(sbk) = -__builtin_inff();
                    any = any || stop;
                }
                if (STATS) n_stop += __ballot(any) != 0ull ? 1u : 0u;
                WQ_MARK("stop_end");
            }
            const bool full = part && P + c <= LIM;
            if (part && !full) {                                     // at most one entry straddles position LIM - 1: keep its rest
                const unsigned int took = LIM - P;
                if (QW) leafq[(lhead + (unsigned int)lane) & (LCAP - 1)] = make_uint2(0x80000000u | (c - took) << kQwLeafShift | (E.x + took), E.y & (kQSlotMask | 1u));
                else leafq[(lhead + (unsigned int)lane) & (LCAP - 1)] = make_uint2(E.x + took, (E.y & (kQSlotMask | 1u)) | (c - took) << kQLeafShift);
            }
            lhead += (unsigned int)__popcll(__ballot(full));
            if (full) atomicAdd(pend(E.y & kQSlotMask), -1);        // after the mins above (LDS operations stay in order)
            if (dbg_on) { d_tri++; d_tril += total; }
            WQ_STAMP(cy_tri);
            WQ_MARK("tri_end");
            continue;
        }
        if (top == 0) {
            bool busy = false;
#pragma unroll
            for (int b = 0; b < NB; ++b) busy = busy || path_[b] >= 0;
            if (drained && stage_used >= stage_n && __ballot(busy) == 0ull) break;   // every wave gets here: each step consumes entries
            if (dbg_on) d_idle++;
            continue;
        }
        // =============================== BOX step: one sibling pair (2 boxes) per lane ===============================
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
        constexpr int KP = QPairs<R>::value;                        // sibling pairs per lane and step: the loads of all of them are in flight together
        const int n = top < 64 * KP ? top : 64 * KP;
        if (!QW && cap - top < 64 * KP) {                                  // no room for up to 128 pushes: serial drain of 64 entries (QW counts what a step found before it writes)
            if (STATS) n_serial++;
            const int nd = top < 64 ? top : 64;
            const bool actd = lane < nd;
            const unsigned int ed = actd ? stack[top - 1 - lane] : 0u;
            top -= nd;
            if (actd) { const unsigned int sd = ed & kQSlotMask; const int cd = (int)((ed >> kQNodeShift) & ~1u); drain_serial(sd, cd); drain_serial(sd, cd + 1); atomicAdd(pend(sd), -1); }
            if (dbg_on) d_serial++;
            WQ_STAMP(cy_box);
            continue;
        }
        if (STATS) n_box++;
        if (QW) {
            // =============================== 4-wide BOX step: the quad of one sibling pair (4 boxes) per lane ===============================
            WQ_MARK("boxw_begin");
            const bool act = lane < n;
            const unsigned int e = act ? stack[top - 1 - lane] : 0u;   // node << 10 | slot << 4, as ever: the pair (c, c + 1)
            const unsigned int sb = e & kQSlotMask;
            unsigned int off = (e >> (kQNodeShift - 5)) & ~63u;         // the pair's quad: 64 bytes at 32 * c
            WQ_CHECK(!act || (((e >> kQNodeShift) & ~1u) >= 2u && (int)((e >> kQNodeShift) & ~1u) + 1 <= sc.n_nodes), 4, off = 0u);
            top -= n;
            const float4 A = rowA(sb), Oo = rowO(sb);
            const uint4 *qp = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(sc.nodesw) + off);
            const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
            const unsigned long long mact = __ballot(act);
            unsigned long long mI[4], mL[4], mF[4];
            const float W = fmaf(2.f * kQ16FaceCells * (1.f + 0x1p-12f), vmax3abs(A.x, A.y, A.z), A.w);   // band + the fixed-point boxes' enlargement (two cells per face, rt_qnodes.hip.h) in the ray's parameter: a hit by more than this is a hit of the REAL box
            unsigned int p_[4];
            auto child = [&](const uint4 q, const int j) {
                const float4 cq = make_float4((float)(q.x & 0xffffu), (float)(q.x >> 16), (float)(q.y & 0xffffu), 0.f);
                const float4 hq = make_float4((float)(q.y >> 16), (float)(q.z & 0xffffu), (float)(q.z >> 16), 0.f);
                const float kx = fmaf(cq.x, A.x, -Oo.x), ky = fmaf(cq.y, A.y, -Oo.y), kz = fmaf(cq.z, A.z, -Oo.z);
                const float tn = vmax3(fmaf(-hq.x, fabsf(A.x), kx), fmaf(-hq.y, fabsf(A.y), ky), fmaf(-hq.z, fabsf(A.z), kz));
                const float tf = vmin3(fmaf(hq.x, fabsf(A.x), kx), fmaf(hq.y, fabsf(A.y), ky), fmaf(hq.z, fabsf(A.z), kz));
                const float d = tf - tn;
                const unsigned long long g = mact & ~__ballot(d < -A.w);         // not excluded: the ray's own band (hand-off) covers every box; a NaN is not excluded either
                p_[j] = q.w;                                                     // > 0 internal (the entry), < 0 leaf (count, first triangle), 0 nothing
                mL[j] = g & __ballot((int)q.w < 0);
                mF[j] = mL[j] & ~__ballot(d > W);                                // leaves hit by less than the enlargement: flagged (their real box is tested when a triangle is accepted)
                mI[j] = (j & 1) ? g & __ballot((int)q.w > 0) : g & ~mL[j];      // places 0 and 2 always hold a node (qquads_kernel): only 1 and 3 can be empty
                if (STATS) {
                    wk.box += act ? 1u : 0u;
                    if (__builtin_amdgcn_inverse_ballot_w64(mI[j] | mL[j])) wk.nodes++;
                    if (__builtin_amdgcn_inverse_ballot_w64(mL[j])) wk.tris += (q.w >> kQwLeafShift) & 0x7fu;
                }
            };
            child(q0, 0); child(q1, 1); child(q2, 2); child(q3, 3);
            const int nI = __popcll(mI[0]) + __popcll(mI[1]) + __popcll(mI[2]) + __popcll(mI[3]);
            const int nL = __popcll(mL[0]) + __popcll(mL[1]) + __popcll(mL[2]) + __popcll(mL[3]);
            if (__builtin_expect(top + nI > cap || (int)(ltail - lhead) + nL > LCAP, 0)) {   // no room for what this step found: its pairs are walked serially instead
                if (STATS) n_serial++;
                if (act) { const int cd = (int)((e >> kQNodeShift) & ~1u); drain_serial(sb, cd); drain_serial(sb, cd + 1); atomicAdd(pend(sb), -1); }
                if (dbg_on) d_serial++;
                WQ_STAMP(cy_box);
                continue;
            }
            // pushes, child by child (the stack is a bag): position = entries before this child's + set lanes below
            // (STATS: n_lpush2 / n_lpush count the internal / leaf push blocks entered -- a block runs only if some lane pushes that child)
            int base = top;
            WQ_MARK("ipushw_begin");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mI[j] != 0ull) {                  // (a block without the test measured the same: profiles/round5/ab_wide_nodes.txt)
                    if (STATS) n_lpush2++;
                    const int pos = base + lanes_below(mI[j]);
                    if (__builtin_amdgcn_inverse_ballot_w64(mI[j])) stack[pos] = p_[j] | sb;
                    base += __popcll(mI[j]);
                }
            }
            top = base;
            unsigned int lb = ltail;
            WQ_MARK("lpushw_begin");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mL[j] != 0ull) {
                    if (STATS) n_lpush++;
                    const unsigned int pos = lb + (unsigned int)lanes_below(mL[j]);
                    if (__builtin_amdgcn_inverse_ballot_w64(mL[j])) { unsigned int *const lq = reinterpret_cast<unsigned int *>(leafq + (pos & (LCAP - 1))); lq[0] = p_[j]; lq[1] = or_bit0(sb, mF[j]); }
                    lb += (unsigned int)__popcll(mL[j]);
                }
            }
            ltail = lb;
            WQ_MARK("lpushw_end");
            const int delta = lane_count8_minus(mI, mL, mact);
            if (delta != 0) atomicAdd(pend(sb), delta);
            WQ_CHECK(top >= 0 && top <= cap && top <= SCAP, 8, (void)0);
            WQ_CHECK(ltail - lhead <= (unsigned int)LCAP, 16, (void)0);
            if (dbg_on) { d_box++; d_boxl += 4u * (unsigned int)n; if ((unsigned int)top > d_maxtop) d_maxtop = (unsigned int)top; }
            WQ_STAMP(cy_box);
            WQ_MARK("boxw_end");
            continue;
        }
        WQ_MARK("box_begin");
        bool act_[KP]; unsigned int e_[KP], sb_[KP], off_[KP];
        float4 A_[KP], Oo_[KP], c0_[KP], h0_[KP], c1_[KP], h1_[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            act_[k] = lane + 64 * k < n;
            e_[k] = act_[k] ? stack[top - 1 - lane - 64 * k] : 0u;   // node << 10 | slot << 4: the sibling nodes (one 64-byte line) of the ray in that slot
            sb_[k] = e_[k] & kQSlotMask;                             // the slot's table row
            off_[k] = QN ? (e_[k] >> (kQNodeShift - 4)) & ~31u : (e_[k] >> (kQNodeShift - 5)) & ~63u;   // the pair's byte offset in the node array (QN: 32 bytes per pair)
            WQ_CHECK(!act_[k] || (((e_[k] >> kQNodeShift) & ~1u) >= 2u && (int)((e_[k] >> kQNodeShift) & ~1u) + 1 <= sc.n_nodes), 4, off_[k] = 0u);
        }
        top -= n;
#pragma unroll
        for (int k = 0; k < KP; ++k) {                                // all loads first: they are in flight together
            A_[k] = rowA(sb_[k]); Oo_[k] = rowO(sb_[k]);              // siblings belong to one ray: one table read for both
            if (QN) {
                const uint4 *qp = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(sc.nodesh) + off_[k]);
                const uint4 q0 = qp[0], q1 = qp[1];
                c0_[k] = make_float4((float)(q0.x & 0xffffu), (float)(q0.x >> 16), (float)(q0.y & 0xffffu), __uint_as_float(q0.w));
                h0_[k] = make_float4((float)(q0.y >> 16), (float)(q0.z & 0xffffu), (float)(q0.z >> 16), 0.f);
                c1_[k] = make_float4((float)(q1.x & 0xffffu), (float)(q1.x >> 16), (float)(q1.y & 0xffffu), __uint_as_float(q1.w));
                h1_[k] = make_float4((float)(q1.y >> 16), (float)(q1.z & 0xffffu), (float)(q1.z >> 16), 0.f);
            } else {
            load_pair(off_[k], c0_[k], h0_[k], c1_[k], h1_[k]);
            }
        }
#if defined(RT_DEBUG) && defined(RT_PAD_VMEM)     // sensitivity experiment: the four 16-byte loads of the pair once more (L1 hits: address / tag pipeline only)
        {
            const unsigned char *pp = nodes + off_[0];
            typedef float pad_v4f __attribute__((ext_vector_type(4)));
            pad_v4f x0, x1, x2, x3;
            asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n"
                         "global_load_dwordx4 %3, %4, off offset:48\n s_waitcnt vmcnt(0)" : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(pp) : "memory");
            asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        }
#endif
#if defined(RT_DEBUG) && defined(RT_PAD_VMEM_N)   // what a vector load costs: N extra loads of W dwords each (RT_PAD_VMEM_N x RT_PAD_VMEM_W), same lines as the step's own
        {
            const unsigned char *pp = nodes + off_[0];
            typedef float pad_v4f __attribute__((ext_vector_type(4)));
#if RT_PAD_VMEM_W == 4
#define RT_PAD_LD "global_load_dwordx4"
            pad_v4f x0, x1, x2, x3;
#elif RT_PAD_VMEM_W == 2
#define RT_PAD_LD "global_load_dwordx2"
            typedef float pad_v2f __attribute__((ext_vector_type(2)));
            pad_v2f x0, x1, x2, x3;
#else
#define RT_PAD_LD "global_load_dword"
            float x0, x1, x2, x3;
#endif
            asm volatile(RT_PAD_LD " %0, %1, off" : "=&v"(x0) : "v"(pp) : "memory");
            if (RT_PAD_VMEM_N >= 2) asm volatile(RT_PAD_LD " %0, %1, off offset:16" : "=&v"(x1) : "v"(pp) : "memory");
            if (RT_PAD_VMEM_N >= 3) asm volatile(RT_PAD_LD " %0, %1, off offset:32" : "=&v"(x2) : "v"(pp) : "memory");
            if (RT_PAD_VMEM_N >= 4) asm volatile(RT_PAD_LD " %0, %1, off offset:48" : "=&v"(x3) : "v"(pp) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(x0));
            if (RT_PAD_VMEM_N >= 2) asm volatile("" :: "v"(x1));
            if (RT_PAD_VMEM_N >= 3) asm volatile("" :: "v"(x2));
            if (RT_PAD_VMEM_N >= 4) asm volatile("" :: "v"(x3));
#undef RT_PAD_LD
        }
#endif
#if defined(RT_DEBUG) && defined(RT_PAD_VMEM2)    // ... the same pair from the OTHER breadth-first copy (lo / hi form): four more loads to lines the step does not otherwise touch
        {
            const unsigned char *pp = reinterpret_cast<const unsigned char *>(sc.nodesq) + off_[0];
            typedef float pad_v4f __attribute__((ext_vector_type(4)));
            pad_v4f x0, x1, x2, x3;
            asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:16\n global_load_dwordx4 %2, %4, off offset:32\n"
                         "global_load_dwordx4 %3, %4, off offset:48\n s_waitcnt vmcnt(0)" : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(pp) : "memory");
            asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        }
#endif
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const bool act = act_[k];
            const unsigned int sb = sb_[k], off = off_[k];
            const float4 A = A_[k], Oo = Oo_[k], c0 = c0_[k], h0 = h0_[k], c1 = c1_[k], h1 = h1_[k];
        // From here on every predicate of the step is a 64-bit LANE MASK in scalar registers: a ballot of a plain comparison is the
            // comparison's result register, masks combine on the scalar unit, and __builtin_amdgcn_inverse_ballot_w64 turns one into the
            // execution mask of an `if` (or the selector of a v_cndmask) for free.  Written with bools the compiler materialises each
            // combined predicate as 0 / 1 in a vector register and compares it back into a mask for its ballot: ten half-rate
            // instructions per step.
            bool hit0 = false, miss0 = false, hit1 = false, miss1 = false;
            if (!QN) {
            cbox_filter(c0, h0, A, Oo, hit0, miss0);
            cbox_filter(c1, h1, A, Oo, hit1, miss1);
            }
            const unsigned long long mact = __ballot(act);
            unsigned long long mflag0 = 0ull, mflag1 = 0ull;   // QN: leaves hit by less than the fixed-point box's enlargement (their real box is tested when a triangle is accepted)
            if (QN) {
                // the fixed-point box contains the real one and no face is further than 2 cells from the real face (rt_qnodes.hip.h): in the
                // ray's parameter that is 2 max|r'| per face, 4 for d = far - near.  `miss` (d < -band) therefore holds for the real box too, and so
                // does a hit by more than band + 4 max|r'|; what lies between enters internal nodes (a superset of the reference's visits: harmless,
                // its box test is monotone along a path of nested boxes) and flags leaves
                const float slack = 2.f * kQ16FaceCells * (1.f + 0x1p-12f) * vmax3abs(A.x, A.y, A.z);
                float d0, d1, band0, band1;
                cbox_dband(c0, h0, A, Oo, d0, band0);
                cbox_dband(c1, h1, A, Oo, d1, band1);
                miss0 = d0 < -band0; miss1 = d1 < -band1;
                hit0 = !miss0; hit1 = !miss1;
                mflag0 = __ballot(!(d0 > band0 + slack)); mflag1 = __ballot(!(d1 > band1 + slack));
            }
            unsigned long long mh0 = __ballot(hit0), mh1 = __ballot(hit1);
            const unsigned long long md0 = mh0 | __ballot(miss0), md1 = mh1 | __ballot(miss1);
            const unsigned long long und = mact & ~(md0 & md1);
            // literal arithmetic for undecided lanes behind a wave-uniform branch (six IEEE divisions per box, almost never needed)
            if (!QN && __builtin_expect(und != 0ull, 0)) {
                if (STATS) n_lit++;
                bool l0 = false, l1 = false;
                const bool u0 = __builtin_amdgcn_inverse_ballot_w64(mact & ~md0), u1 = __builtin_amdgcn_inverse_ballot_w64(mact & ~md1);
                if (STATS) wk.lit_box += (u0 ? 1u : 0u) + (u1 ? 1u : 0u);
                if (u0 || u1) {
                    const float4 C = rowC(sb), D = rowD(sb);
                    const f3 O = mk(C.x, C.y, C.z), u = mk(C.w, D.x, D.y);
                    const float4 *bp = sc.nodesq + (off >> 4);            // the same pair as (lo, hi): the breadth-first array of the boxes themselves
                    if (u0) l0 = slab(bp[0], bp[1], O, u);
                    if (u1) l1 = slab(bp[2], bp[3], O, u);
                }
                mh0 = (mh0 & md0) | __ballot(l0);
                mh1 = (mh1 & md1) | __ballot(l1);
            }
            {
                int k0 = __float_as_int(h0.w), k1 = __float_as_int(h1.w);               // kind: < 0 internal, > 0 leaf (count << 11), 0 empty leaf
                unsigned int p0 = __float_as_uint(c0.w), p1 = __float_as_uint(c1.w);   // payload: first child << 10 | first triangle
                if (QN) {   // one word: internal = first child << (kQNodeShift - 1) (bit 31 clear); leaf = 1 << 31 | count << S | first triangle, S = sc.qleaf_shift
                    const unsigned int S = (unsigned int)sc.qleaf_shift, fm = (1u << S) - 1u;
                    k0 = (int)p0 < 0 ? (int)(((p0 & 0x7fffffffu) >> S) << kQLeafShift) : -1; k1 = (int)p1 < 0 ? (int)(((p1 & 0x7fffffffu) >> S) << kQLeafShift) : -1;
                    p0 = (int)p0 < 0 ? p0 & fm : p0 << 1; p1 = (int)p1 < 0 ? p1 & fm : p1 << 1;
                }
                const unsigned long long g0 = mh0 & mact, g1 = mh1 & mact;
                const unsigned long long mI0 = g0 & __ballot(k0 < 0), mI1 = g1 & __ballot(k1 < 0), mL0 = g0 & __ballot(k0 > 0), mL1 = g1 & __ballot(k1 > 0);
                if (STATS) {
                    const bool b0 = __builtin_amdgcn_inverse_ballot_w64(g0), b1 = __builtin_amdgcn_inverse_ballot_w64(g1);
                    wk.box += act ? 2u : 0u; wk.nodes += (b0 ? 1u : 0u) + (b1 ? 1u : 0u);
                    wk.tris += ((b0 && k0 > 0) ? (uint32_t)(k0 >> kQLeafShift) : 0u) + ((b1 && k1 > 0) ? (uint32_t)(k1 >> kQLeafShift) : 0u);
                }
                // a hit internal node pushes ITS pair of children; the order of entries on the stack does not matter (the traversal is a bag),
                // so a lane's one or two entries go next to each other: one prefix count over both masks, one address
                const int oI = lanes_below2(mI0, mI1);
                unsigned int *const sp = stack + top + oI;
                const bool sI0 = __builtin_amdgcn_inverse_ballot_w64(mI0), sL0 = __builtin_amdgcn_inverse_ballot_w64(mL0);
                if (__builtin_amdgcn_inverse_ballot_w64(mI0 | mI1)) sp[0] = (sI0 ? p0 : p1) | sb;
                if (__builtin_amdgcn_inverse_ballot_w64(mI0 & mI1)) sp[1] = p1 | sb;
                top += __popcll(mI0) + __popcll(mI1);
                const unsigned int oL = ltail + (unsigned int)lanes_below2(mL0, mL1);
                if (STATS) { n_lpush += (mL0 | mL1) != 0ull ? 1u : 0u; n_lpush2 += (mL0 & mL1) != 0ull ? 1u : 0u; }
                WQ_MARK("lpush_begin");
                if (QN) {
                    const unsigned int f0 = __builtin_amdgcn_inverse_ballot_w64(mflag0) ? 1u : 0u, f1 = __builtin_amdgcn_inverse_ballot_w64(mflag1) ? 1u : 0u;
                    if (__builtin_amdgcn_inverse_ballot_w64(mL0 | mL1)) leafq[oL & (LCAP - 1)] = make_uint2(sL0 ? p0 : p1, (unsigned int)(sL0 ? k0 : k1) | sb | (sL0 ? f0 : f1));
                    WQ_MARK("lpush2_begin");
                    if (__builtin_amdgcn_inverse_ballot_w64(mL0 & mL1)) leafq[(oL + 1u) & (LCAP - 1)] = make_uint2(p1, (unsigned int)k1 | sb | f1);
                } else {
                if (__builtin_amdgcn_inverse_ballot_w64(mL0 | mL1)) leafq[oL & (LCAP - 1)] = make_uint2(sL0 ? p0 : p1, (unsigned int)(sL0 ? k0 : k1) | sb);
                WQ_MARK("lpush2_begin");
                if (__builtin_amdgcn_inverse_ballot_w64(mL0 & mL1)) leafq[(oL + 1u) & (LCAP - 1)] = make_uint2(p1, (unsigned int)k1 | sb);
                }
                WQ_MARK("lpush_end");
                ltail += (unsigned int)(__popcll(mL0) + __popcll(mL1));
                // outstanding entries of the ray: this pair is gone (-1), every pushed pair and leaf entry counts +1: one LDS add per lane
                const int delta = lane_count4_minus(mI0, mI1, mL0, mL1, mact);
                if (delta != 0) atomicAdd(pend(sb), delta);
            }
        }
        WQ_CHECK(top >= 0 && top <= cap && top <= SCAP, 8, (void)0);
        if (dbg_on) { d_box++; d_boxl += 2u * (unsigned int)n; if ((unsigned int)top > d_maxtop) d_maxtop = (unsigned int)top; }
        WQ_STAMP(cy_box);
        WQ_MARK("box_end");
    }
#undef WQ_STAMP
#undef WQ_CHECK
#undef WQ_MARK
    if (dbg_on && lane == 0) {
        unsigned long long *d = st.dbg + 16 * (size_t)((blockIdx.x * blockDim.x + tid) >> 6);
        d[0] = dbg_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = d_box; d[3] = d_boxl; d[4] = d_tri; d[5] = d_tril;
        d[6] = d_rounds; d[7] = d_rays; d[8] = d_serial; d[9] = cy_srv; d[10] = cy_tri; d[11] = cy_box; d[12] = dbg_tdrain; d[13] = d_idle;
        d[14] = d_fetch; d[15] = d_maxtop;
    }
    if (STATS && lane == 0) {
        const unsigned int v[12] = {n_iter, n_refill, n_round, n_fetch, n_tri, n_box, n_lit, n_serial, n_tdiv, n_lpush, n_lpush2, n_stop};
        for (int k = 0; k < 12; ++k) if (v[k]) atomicAdd(&fr.work[8 + k], (unsigned long long)v[k]);
    }
    wf_flush_work<STATS>(fr, wk);
}

}  // namespace rtk
