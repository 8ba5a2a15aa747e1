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
    hi_ = hi_ > 65535.0 ? 65535.0 : (hi_ >= 1.0 ? hi_ : 65535.0);       // NaN or an out-of-range box: the widest box (never excluded)
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

// The 4-wide nodes of wf_travq<.., QW> (rt_travq.hip.h): one thread per sibling pair c = 2, 4, .. of the breadth-first array.  The quad of the pair (c, c + 1), at
// uint4 index 2 c, is the nodesh records of the children of c and of c + 1 -- the boxes a ray meets two levels below the pair's parent -- where a LEAF of the pair
// stands for itself next to an empty place.  Index 0 (what an idle lane's zero entry addresses) is four empty places.
// sel (optional): for the pair at uint4 index 2 c, sel[c / 2] = the breadth-first indices of the up to four nodes its quad holds instead (0 = empty place; places 0 and 2 are never
// empty) -- ANY cut of the subtree below the pair's parent is exact; the host picks the one a surface-area model likes best (install_scene in rt_capi.hip).
__global__ __launch_bounds__(256) void qquads_kernel(const uint4 *__restrict__ nodesh, int n_bfs, int node_shift, int leaf_shift, const int4 *__restrict__ sel, uint4 *__restrict__ nodesw) {
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
    if (sel) {
        const int4 q = sel[c / 2];
        const int ids[4] = {q.x, q.y, q.z, q.w};
        if (q.x > 0 && q.z > 0) {
            for (int j = 0; j < 4; ++j) out[j] = (ids[j] > 0 && ids[j] <= n_bfs) ? conv(nodesh[ids[j]]) : none;
            return;
        }
    }
    for (int s = 0; s < 2; ++s) {
        const int x = c + s;
        const uint4 rec = x <= n_bfs ? nodesh[x] : make_uint4(0u, 0u, 0u, 0x80000000u);
        if ((int)rec.w > 0) {                                                   // internal: its two children
            const int p = (int)((rec.w << 1) >> node_shift);
            out[2 * s] = conv(nodesh[p]); out[2 * s + 1] = conv(nodesh[p + 1]);
        } else {
            out[2 * s] = conv(rec); out[2 * s + 1] = none;
        }
    }
}

}  // namespace rtk
