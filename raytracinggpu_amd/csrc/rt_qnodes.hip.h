// rt_qnodes.hip.h -- the sibling pairs of the breadth-first node array once more, in 16-bit fixed point (round 4).
//
// wf_travq's BOX step is bound by the NUMBER of vector-memory instructions it issues (profiles/round4/ab_load_cost_model.txt): a pair of
// (centre, half extent, payload, kind) nodes is 64 bytes = four 16-byte loads per lane.  Here a node is 16 bytes -- centre and half
// extent of each axis as unsigned 16-bit numbers on a grid laid over the root box, and one payload word -- so a pair is TWO loads.
// Every face is rounded OUTWARDS: the fixed-point box contains the real one, and no face is further than TWO cells from the real one (kQ16FaceCells in rt_travq.hip.h):
//     cq = round((c - g) / s),   hq = the least integer with hq s >= h + |c - (g + cq s)|          (binary64; c, h = exact centre / half extent of lo, hi)
// -- a face sits at most one cell (the ceiling) + 2 |c - (g + cq s)| <= one more cell outside.  (Rounds 4-5 added a spare cell to hq against the rounding of the quotient; the
// comparison below makes the containment hold without it, and the kernels' flag threshold shrank from six cells to four: profiles/round5/ab_wide_nodes.txt.)
// The kernel (wf_travq<.., QN = true>) uses the fixed-point box in two ways, both conservative with respect to BoundingBox::intersect
// (cpu_launcher.cpp:146-157) on the real box:
//   * a box the ray misses even enlarged is missed by the reference (no entry pushed, no triangle tested);
//   * a LEAF the ray hits by more than the enlargement is hit by the reference (its triangles are tested); a leaf in between is
//     tested too, but carries a flag, and a triangle accepted in it counts only if the reference's own test of the leaf's REAL box
//     (slab_filtered on the (lo, hi) copy, found through tri2leaf) says hit;
//   * an INTERNAL node that is not excluded is entered.  That visits a superset of the nodes the reference visits, which changes no
//     result: for a ray without zero / denormal / huge components and a tree whose boxes nest (child inside parent), the reference's
//     test is monotone -- every bound (b - O) / u is a monotone function of b under round-to-nearest, so a child's interval per axis
//     lies inside its parent's and "hit child => hit parent" holds in the COMPUTED values -- hence the leaves the reference reaches
//     are exactly the leaves whose own box it hits, and those are decided exactly as above.  Rays outside that class never use the
//     fixed-point pairs (they are walked serially with the literal test at hand-off), trees that do not nest keep the 64-byte pairs.
// Payload word: internal node = first child << (kQNodeShift - 1) (bit 31 clear: the kernel shifts it once more for the stack entry); leaf =
// 1 << 31 | count << S | first triangle with S = 24 (leaves of at most 127 triangles, 2^24 triangles) or S = 20 (2 047 and 2^20): q16_leaf_shift.
#pragma once
#include <hip/hip_runtime.h>

namespace rtk {

struct QGrid { float gx, gy, gz, sx, sy, sz; };

// S for a tree whose largest leaf holds max_leaf triangles, 0 = the payload word cannot hold its leaves
__host__ inline int q16_leaf_shift(int max_leaf, long long n_tris) {
    if (max_leaf <= 127 && n_tris <= (1ll << 24)) return 24;
    if (max_leaf <= 2047 && n_tris <= (1ll << 20)) return 20;
    return 0;
}
constexpr double kQ16Cells = 65000.0;           // cells per axis over the root box (the rest of the 16-bit range is slack for the outward rounding)

__host__ inline QGrid q16_grid(float4 root_lo, float4 root_hi) {
    QGrid g;
    g.gx = root_lo.x; g.gy = root_lo.y; g.gz = root_lo.z;
    const double ex = (double)root_hi.x - (double)root_lo.x, ey = (double)root_hi.y - (double)root_lo.y, ez = (double)root_hi.z - (double)root_lo.z;
    g.sx = (float)(ex / kQ16Cells > 1e-30 ? ex / kQ16Cells : 1e-30);
    g.sy = (float)(ey / kQ16Cells > 1e-30 ? ey / kQ16Cells : 1e-30);
    g.sz = (float)(ez / kQ16Cells > 1e-30 ? ez / kQ16Cells : 1e-30);
    return g;
}

__device__ __forceinline__ void q16_axis(float lo, float hi, float g, float s, unsigned int &cq, unsigned int &hq) {
    const double c = ((double)lo + (double)hi) * 0.5, h = ((double)hi - (double)lo) * 0.5;
    double ci = floor((c - (double)g) / (double)s + 0.5);
    ci = ci < 0.0 ? 0.0 : (ci > 65535.0 ? 65535.0 : ci);
    const double cw = (double)g + ci * (double)s;
    const double need = (h + fabs(c - cw)) * (1.0 + 0x1p-50);            // (the sum may have rounded down by half an ulp)
    double hi_ = ceil(need / (double)s);
    if (hi_ * (double)s < need) hi_ += 1.0;                              // the quotient rounded down across an integer: hi_ s is exact (16 x 24 bits)
    // 0 is a legitimate value (a flat box whose centre sits on a grid point: an axis-aligned leaf on the root's minimum face): it becomes ONE cell, which is
    // still within kQ16FaceCells of the real face and makes the box thinner than any ray's accept threshold on that axis, so the leaf is always flagged and the
    // reference's own test of its zero-thickness box (never a hit: strict '>', cpu:156) decides.  Only NaN / out-of-range becomes the widest box (never excluded).
    hi_ = hi_ > 65535.0 ? 65535.0 : (hi_ >= 1.0 ? hi_ : (hi_ == 0.0 ? 1.0 : 65535.0));
    cq = (unsigned int)ci; hq = (unsigned int)hi_;
}

// one thread per breadth-first node b in [1, n_bfs]: nodesq[2b], [2b+1] = (lo, hi) of the node, nodesb[2b].w / [2b+1].w = payload / kind as wf_travq carries them
__global__ __launch_bounds__(256) void qnodes_kernel(const float4 *__restrict__ nodesq, const float4 *__restrict__ nodesb, int n_bfs, QGrid g,
                                                     uint4 *__restrict__ nodesh, int *__restrict__ tri2leaf, int n_tris, int leaf_kind_shift, int leaf_shift) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (b > n_bfs) return;
    const float4 lo = nodesq[2 * (size_t)b], hi = nodesq[2 * (size_t)b + 1];
    const unsigned int payload = __float_as_uint(nodesb[2 * (size_t)b].w);
    const int kind = __float_as_int(nodesb[2 * (size_t)b + 1].w);
    unsigned int cx, cy, cz, hx, hy, hz;
    q16_axis(lo.x, hi.x, g.gx, g.sx, cx, hx);
    q16_axis(lo.y, hi.y, g.gy, g.sy, cy, hy);
    q16_axis(lo.z, hi.z, g.gz, g.sz, cz, hz);
    unsigned int pay;
    if (kind < 0) pay = payload >> 1;                                            // first child << (kQNodeShift - 1): the child index is even
    else {
        const int cnt = kind >> leaf_kind_shift, first = (int)payload;
        pay = 0x80000000u | (unsigned int)cnt << leaf_shift | (unsigned int)first;
        for (int t = 0; t < cnt; ++t) if (first + t < n_tris) tri2leaf[first + t] = b;
    }
    nodesh[b] = make_uint4(cx | cy << 16, cz | hx << 16, hy | hz << 16, pay);
}

// the real box of every triangle's leaf, by triangle: the flagged-leaf check of a TRI step reads it without going through tri2leaf first
__global__ __launch_bounds__(256) void leaflh_kernel(const float4 *__restrict__ nodesq, const int *__restrict__ tri2leaf, int n_tris, int n_bfs, float4 *__restrict__ leaflh) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    const int lf = tri2leaf[i];
    const bool ok = lf >= 1 && lf <= n_bfs;
    leaflh[2 * (size_t)i] = ok ? nodesq[2 * (size_t)lf] : make_float4(0, 0, 0, 0);
    leaflh[2 * (size_t)i + 1] = ok ? nodesq[2 * (size_t)lf + 1] : make_float4(0, 0, 0, 0);
}

// WHICH four nodes a quad holds.  Any cut of at most four nodes of the subtree below a sibling pair's parent P is exact (the boxes nest; only the leaves' own boxes decide what the
// reference reaches); "the children of c and of c + 1" is one choice.  The three kernels below pick, for every P, the cut that minimises the expected number of stack entries below
// P under the surface-area model -- an internal node y in a cut costs area(y) + the best cost below y -- by a bottom-up DP over
//     g(x, k) = the least cost of covering x's subtree with at most k nodes          (k = 1, 2, 3;  a leaf costs nothing;  tot(x) = min over k1 + k2 = 4 of g(l, k1) + g(r, k2))
// one thread per leaf climbing towards the root, the SECOND arrival at a node computing it.  The result is a function of the tree alone (tests hash the quads of two uploads):
// the first arrival wrote its values, fenced (release at agent scope) and then announced itself with the atomic add; the second one sees the count, fences, and reads both
// children's values with agent-scope atomic loads, i.e. from L2 where the fence put them -- the usual bottom-up reduction (Karras 2012).  (And every recorded choice
// describes a valid cut whatever the numbers were: the DP can only cost optimality, never a result.)  On the cat the model says 7.05 -> 6.69 entries per root hit; measured: BOX steps -3.3 %, frame -1.0 %
// (profiles/round5/ab_wide_nodes.txt).  Areas are taken from the fixed-point half extents (a model needs no more).
struct QdpArgs {
    const uint4 *nodesh; int n_bfs, node_shift;
    float sx, sy, sz;
    int *parent, *cnt;          // [n_bfs + 2]
    float4 *g;                  // (g1, g2, g3, tot)
    uchar4 *ch;                 // (first child's share of g2's split or 0 = the node itself, the same for g3, the first child's share of tot's four, -)
};
__device__ __forceinline__ int qdp_first_child(uint4 rec, int node_shift) { return (int)rec.w > 0 ? (int)((rec.w << 1) >> node_shift) : 0; }

__global__ __launch_bounds__(256) void qdp_init_kernel(const QdpArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (b > a.n_bfs) return;
    a.cnt[b] = 0;
    if (b == 1) a.parent[1] = 0;
    const int fc = qdp_first_child(a.nodesh[b], a.node_shift);
    if (fc > 0 && fc + 1 <= a.n_bfs) { a.parent[fc] = b; a.parent[fc + 1] = b; }
}
__global__ __launch_bounds__(256) void qdp_up_kernel(const QdpArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (b > a.n_bfs || qdp_first_child(a.nodesh[b], a.node_shift) > 0) return;      // leaves start the climb
    a.g[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld = [&](int x) {                                                            // a child's values: written by another workgroup, read past the L1
        const float *p = reinterpret_cast<const float *>(a.g + x);
        return make_float4(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0.f);
    };
    int cur = a.parent[b];
    for (int guard = 0; cur > 0 && cur <= a.n_bfs && guard <= a.n_bfs; ++guard) {    // (a climb is at most the tree's depth long; the range test and the guard only bound a corrupted tree)
        __threadfence();                                                              // my values are visible before I announce myself
        if (atomicAdd(&a.cnt[cur], 1) == 0) return;                                   // first to arrive: the sibling's thread goes on from here
        __threadfence();
        const uint4 rec = a.nodesh[cur];
        const int l = qdp_first_child(rec, a.node_shift), r = l + 1;
        if (l <= 0 || r > a.n_bfs) return;
        const float4 gl = ld(l), gr = ld(r);
        const float ex = (float)(rec.y >> 16) * a.sx, ey = (float)(rec.z & 0xffffu) * a.sy, ez = (float)(rec.z >> 16) * a.sz;
        const float area = ex * ey + ey * ez + ex * ez;
        float tot = gl.x + gr.z; unsigned char ct = 1;                                // k1 + k2 = 4: (1, 3), (2, 2), (3, 1)
        if (gl.y + gr.y < tot) { tot = gl.y + gr.y; ct = 2; }
        if (gl.z + gr.x < tot) { tot = gl.z + gr.x; ct = 3; }
        const float g1 = area + tot;
        float g2 = g1; unsigned char c2 = 0;
        if (gl.x + gr.x < g2) { g2 = gl.x + gr.x; c2 = 1; }
        float g3 = g1; unsigned char c3 = 0;
        if (gl.x + gr.y < g3) { g3 = gl.x + gr.y; c3 = 1; }
        if (gl.y + gr.x < g3) { g3 = gl.y + gr.x; c3 = 2; }
        a.g[cur] = make_float4(g1, g2, g3, tot);
        a.ch[cur] = make_uchar4(c2, c3, ct, 0);
        cur = a.parent[cur];
    }
}

// The 4-wide nodes of wf_travq<.., QW> (rt_travq.hip.h): one thread per sibling pair c = 2, 4, .. of the breadth-first array.  The quad of the pair (c, c + 1), at uint4 index
// 2 c, holds the nodesh records of up to four nodes that cut the subtree below the pair's parent: the DP's choice (parent / ch given), or the children of c and of c + 1 (a LEAF of
// the pair stands for itself next to an empty place).  Internal nodes come first, leaves last (a BOX step runs one push block per place and kind that some lane needs); places 0 and 2
// are never empty.  Index 0 (what an idle lane's zero entry addresses) is four empty places.
__global__ __launch_bounds__(256) void qquads_kernel(const uint4 *__restrict__ nodesh, int n_bfs, int node_shift, int leaf_shift, const int *__restrict__ parent,
                                                     const uchar4 *__restrict__ ch, uint4 *__restrict__ nodesw) {
    const int c = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (c > n_bfs) return;
    uint4 *out = nodesw + 2 * (size_t)c;
    const uint4 none = make_uint4(0u, 0u, 0u, 0u);
    if (c == 0) { out[0] = out[1] = out[2] = out[3] = none; return; }
    // a nodesh record with the payload word the 4-wide step wants: internal = first child << node_shift (> 0: the tree has fewer than 2^21 nodes), leaf = 1 << 31 |
    // count << 24 | first triangle (count <= 127, first < 2^24), nothing = 0
    auto conv = [&](uint4 r) {
        if ((int)r.w >= 0) { r.w = r.w << 1; return r; }
        const unsigned int cnt = (r.w & 0x7fffffffu) >> leaf_shift, first = r.w & ((1u << leaf_shift) - 1u);
        if (cnt == 0u) return none;
        r.w = 0x80000000u | cnt << 24 | first;
        return r;
    };
    int cut[4], nc = 0;
    if (parent && ch && c + 1 <= n_bfs) {
        const int P = parent[c] >= 1 && parent[c] <= n_bfs ? parent[c] : 1;
        const int k1 = ch[P].z >= 1 && ch[P].z <= 3 ? ch[P].z : 2;
        int sx[8], sk[8], sp = 0;
        sx[sp] = c + 1; sk[sp++] = 4 - k1; sx[sp] = c; sk[sp++] = k1;                 // the first child's part comes out first
        while (sp > 0 && nc < 4) {
            const int x = sx[--sp], k = sk[sp];
            const int fc = x <= n_bfs ? qdp_first_child(nodesh[x], node_shift) : 0;
            const int c1 = (fc > 0 && k >= 2) ? (k == 2 ? ch[x].x : ch[x].y) : 0;
            if (x < 1 || x > n_bfs) continue;
            if (c1 <= 0 || c1 >= k || sp + 2 > 8) { cut[nc++] = x; continue; }
            sx[sp] = fc + 1; sk[sp++] = k - c1; sx[sp] = fc; sk[sp++] = c1;
        }
    } else {
        for (int s = 0; s < 2; ++s) {
            const int x = c + s;
            const int fc = x <= n_bfs ? qdp_first_child(nodesh[x], node_shift) : 0;
            if (fc > 0 && fc + 1 <= n_bfs) { cut[nc++] = fc; cut[nc++] = fc + 1; } else if (x <= n_bfs) cut[nc++] = x;
        }
    }
    // internal nodes first; then two nodes sit at places 0 and 2, three at 0, 1, 2
    int ord[4], no = 0;
    for (int j = 0; j < nc; ++j) if (qdp_first_child(nodesh[cut[j]], node_shift) > 0) ord[no++] = cut[j];
    for (int j = 0; j < nc; ++j) if (qdp_first_child(nodesh[cut[j]], node_shift) <= 0) ord[no++] = cut[j];
    uint4 recs[4] = {none, none, none, none};
    for (int j = 0; j < no; ++j) recs[j] = conv(nodesh[ord[j]]);
    // empty leaves (count 0) convert to nothing: keep the real ones in front, then spread
    uint4 real[4]; int nr = 0;
    for (int j = 0; j < no; ++j) if (recs[j].w != 0u) real[nr++] = recs[j];
    out[0] = nr > 0 ? real[0] : none;
    out[1] = nr > 2 ? real[1] : none;
    out[2] = nr > 2 ? real[2] : (nr > 1 ? real[1] : none);
    out[3] = nr > 3 ? real[3] : none;
}

}  // namespace rtk
