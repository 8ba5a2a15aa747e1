// rt_lbvh.hip.h -- device-side LBVH builder ("... and a GPU LBVH build").  Included by rt_capi.hip.
//
// The reference's builder (TriangleMesh::buildBVH, cpu_launcher.cpp:190-224; its device twin is ONE thread recursing,
// global_launcher.cu:298-331 launched <<<1,1>>> from :848-881) splits at the spatial midpoint and stops when a side would be empty or
// hold a single triangle (cpu:217) -- on a mesh much larger than the cat that leaves leaves of hundreds of triangles (2 M triangles:
// 357 triangle tests per ray).  rt_mesh_rebuild reproduces that tree bit for bit (rt_bvhbuild.hip.h); this file builds a DIFFERENT,
// better tree for callers who ask for it (rt_mesh_rebuild_mode(RT_BVH_LBVH)), in parallel:
//
//   1. bounds of the triangle centroids (one atomic min / max per wave and axis on order-preserving integer keys)
//   2. 63-bit Morton code of every centroid (21 bits per axis), radix sort of (code, triangle) pairs (rocPRIM)
//   3. the binary radix tree over the sorted codes, one thread per internal node (Karras 2012: direction, range by doubling +
//      binary search, split by binary search on the common prefix; equal codes fall back to the index, so the tree is always proper)
//   4. boxes bottom-up, one thread per triangle, the second thread to arrive at a node merges its children (min / max of the vertex
//      coordinates: the same values compute_bbox (cpu:180-188) folds for the node's range, in a tree order) -- and decides, by the
//      surface-area heuristic, whether the node is cheaper as ONE leaf than as its subtree: the reference's traversal never prunes by
//      distance, so a ray pays for every box it pierces and every triangle of every leaf it enters, which is exactly the
//      cost the SAH models: cost(leaf) = Ct * triangles, cost(inner) = 2 Cb + (A_l cost_l + A_r cost_r) / A, with Ct / Cb = 1.6 from the
//      measured step times of wf_travq (128 triangle tests ~2 400 cycles, 128 box tests ~1 500).  Leaves hold at most kLbvhLeaf
//      triangles; ranges of up to kLbvhMinLeaf are always leaves.
//   5. a node survives iff no ancestor became a leaf (one walk up per node)
//   6. the surviving nodes, numbered by a prefix sum (root = 0), written in the reference's own flat layout -- 10 floats per node
//      [left, right, mn.xyz, mx.xyz, triangle_start, triangle_end), bvhTreeToArray (optimized.cu:512-534) -- with the triangle ranges
//      referring to the sorted order, which is returned beside it.
//
// The output is therefore exactly what rt_scene_upload accepts (a8) and what the oracle can be handed (or_mesh_set_bvh): every kernel
// variant runs on it unchanged, and parity stays "HIP == oracle on the same tree".
#pragma once
#include "rt_travq.hip.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace rtk {

constexpr int kLbvhLeaf = 32;            // triangles per leaf at most (the SAH decides below that)
constexpr int kLbvhMinLeaf = 2;          // ranges this small are never split further
constexpr float kLbvhCt = 1.6f, kLbvhCb = 1.0f;   // cost of one triangle test / one box test (wf_travq step times)
constexpr int kLbvhLeafBit = 1 << 30;    // child reference: leaf j = j | kLbvhLeafBit, internal i = i

struct LbvhArgs {
    const float4 *verts;                 // current vertex positions
    const int4 *tidx_up;                 // vertex indices of uploaded triangle t
    int n;                               // triangles
    unsigned int *bounds;                // [6] ordered-int keys: min xyz, max xyz of the centroids
    unsigned long long *keys;            // [n] Morton codes (sorted by the host call in between)
    int *vals;                           // [n] triangle of sorted position k
    int *left, *right, *parent;          // [n - 1] internal nodes (children as references, see kLbvhLeafBit)
    int *first, *last;                   // [n - 1] range of sorted positions an internal node covers
    int *leaf_parent;                    // [n] parent (internal node) of primitive leaf k
    int *flag;                           // [n - 1] arrivals during the bottom-up pass
    float4 *ibox_lo, *ibox_hi;           // [n - 1] boxes of the internal nodes
    float4 *lbox_lo, *lbox_hi;           // [n] boxes of the primitive leaves (one triangle each)
    float *cost;                         // [n - 1] SAH cost of the subtree as decided (leaf or inner)
    int *leafify;                        // [n - 1] the node is cheaper as one leaf
    int *alive;                          // [2 n - 1] candidate c survives: c < n - 1 internal node c, else primitive leaf c - (n - 1)
    int *index;                          // [2 n - 1] exclusive prefix sum of alive: the node's number in the output
    float *arr10;                        // [n_alive * 10] output
    int *stats;                          // [4]: leaves, largest leaf, deepest leaf (filled by lbvh_emit_kernel)
    float ct, cb;                        // cost of one triangle test / one box test in the leaf cut (kLbvhCt, kLbvhCb unless RT_LBVH_CT says otherwise)
};

// float <-> unsigned key with the same order (for atomicMin / atomicMax on floats)
__device__ __forceinline__ unsigned int lbvh_fkey(float f) { const unsigned int b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__host__ __device__ inline float lbvh_fkey_inv(unsigned int k) {
    const unsigned int b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, b);
}

__device__ __forceinline__ f3 lbvh_centroid(const LbvhArgs &a, int t) {
    const int4 ix = a.tidx_up[t];
    const float4 A = a.verts[ix.x], B = a.verts[ix.y], C = a.verts[ix.z];
    return mk((A.x + B.x + C.x) * (1.f / 3.f), (A.y + B.y + C.y) * (1.f / 3.f), (A.z + B.z + C.z) * (1.f / 3.f));   // quality only: any rounding will do
}

__global__ __launch_bounds__(256) void lbvh_bounds_kernel(const LbvhArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f3 c = mk(0, 0, 0);
    const bool on = t < a.n;
    if (on) c = lbvh_centroid(a, t);
    const float v[3] = {c.x, c.y, c.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float mn = on ? v[k] : 3.0e38f, mx = on ? v[k] : -3.0e38f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&a.bounds[k], lbvh_fkey(mn)); atomicMax(&a.bounds[3 + k], lbvh_fkey(mx)); }
    }
}

__device__ __forceinline__ unsigned long long lbvh_spread21(unsigned long long x) {   // 21 bits -> every third bit
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void lbvh_morton_kernel(const LbvhArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    const f3 c = lbvh_centroid(a, t);
    const float v[3] = {c.x, c.y, c.z};
    unsigned long long q[3];
    // ONE scale for the three axes (the largest extent): cubic cells.  Scaling every axis to its own extent makes the code split a flat
    // mesh across its thin axis as often as along the others -- boxes that overlap (displaced grid of 524 288 triangles: 63.8 box tests per
    // ray and a 5.1 ms frame; with cubic cells 43.5 and 3.55 ms): the thin axis' top bits are then equal for all triangles and the radix
    // tree simply has no split there.
    float ext = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, lbvh_fkey_inv(a.bounds[3 + k]) - lbvh_fkey_inv(a.bounds[k]));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lo = lbvh_fkey_inv(a.bounds[k]);
        float u = ext > 0.f ? (v[k] - lo) / ext : 0.f;
        u = fminf(fmaxf(u, 0.f), 1.f);
        q[k] = (unsigned long long)(u * 2097151.f);                                   // 2^21 - 1 cells
    }
    a.keys[t] = lbvh_spread21(q[0]) << 2 | lbvh_spread21(q[1]) << 1 | lbvh_spread21(q[2]);
    a.vals[t] = t;
}

// length of the common prefix of sorted positions i and j (Karras 2012, section 4); equal codes are told apart by the positions
__device__ __forceinline__ int lbvh_delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned int)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ __launch_bounds__(256) void lbvh_hierarchy_kernel(const LbvhArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.n;
    if (i >= n - 1) return;
    const unsigned long long *keys = a.keys;
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int lc = lo == gamma ? (gamma | kLbvhLeafBit) : gamma;
    const int rc = hi == gamma + 1 ? ((gamma + 1) | kLbvhLeafBit) : gamma + 1;
    a.left[i] = lc; a.right[i] = rc; a.first[i] = lo; a.last[i] = hi;
    if (lc & kLbvhLeafBit) a.leaf_parent[gamma] = i; else a.parent[gamma] = i;
    if (rc & kLbvhLeafBit) a.leaf_parent[gamma + 1] = i; else a.parent[gamma + 1] = i;
    if (i == 0) a.parent[0] = -1;
    a.flag[i] = 0;
}

// which nodes survive: those without an ancestor that became a leaf
__global__ __launch_bounds__(256) void lbvh_alive_kernel(const LbvhArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.n;
    if (c >= 2 * n - 1) return;
    int alive = 1;
    int guard = 0;
    for (int p = c < n - 1 ? a.parent[c] : a.leaf_parent[c - (n - 1)]; p >= 0 && guard < 4096; p = a.parent[p], ++guard)
        if (a.leafify[p]) { alive = 0; break; }
    a.alive[c] = alive;
}

// boxes bottom-up: one thread per triangle (sorted position k); the second arrival at an internal node merges its children
__global__ __launch_bounds__(256) void lbvh_boxes_kernel(const LbvhArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const int4 ix = a.tidx_up[a.vals[k]];
    const float4 A = a.verts[ix.x], B = a.verts[ix.y], C = a.verts[ix.z];
    float4 lo = make_float4(fminf(fminf(A.x, B.x), C.x), fminf(fminf(A.y, B.y), C.y), fminf(fminf(A.z, B.z), C.z), 0.f);
    float4 hi = make_float4(fmaxf(fmaxf(A.x, B.x), C.x), fmaxf(fmaxf(A.y, B.y), C.y), fmaxf(fmaxf(A.z, B.z), C.z), 0.f);
    a.lbox_lo[k] = lo; a.lbox_hi[k] = hi;
    int cur = a.leaf_parent[k];
    for (int guard = 0; cur >= 0 && guard < 4096; ++guard) {              // (depth <= 64 + 32: the guard only bounds a corrupted tree)
        __threadfence();                                                 // my child box is visible before I announce myself
        if (atomicAdd(&a.flag[cur], 1) == 0) return;                     // first to arrive: the sibling's thread goes on from here
        __threadfence();
        const int lc = a.left[cur], rc = a.right[cur];
        const float4 l0 = (lc & kLbvhLeafBit) ? a.lbox_lo[lc & ~kLbvhLeafBit] : a.ibox_lo[lc], l1 = (lc & kLbvhLeafBit) ? a.lbox_hi[lc & ~kLbvhLeafBit] : a.ibox_hi[lc];
        const float4 r0 = (rc & kLbvhLeafBit) ? a.lbox_lo[rc & ~kLbvhLeafBit] : a.ibox_lo[rc], r1 = (rc & kLbvhLeafBit) ? a.lbox_hi[rc & ~kLbvhLeafBit] : a.ibox_hi[rc];
        lo = make_float4(fminf(l0.x, r0.x), fminf(l0.y, r0.y), fminf(l0.z, r0.z), 0.f);
        hi = make_float4(fmaxf(l1.x, r1.x), fmaxf(l1.y, r1.y), fmaxf(l1.z, r1.z), 0.f);
        a.ibox_lo[cur] = lo; a.ibox_hi[cur] = hi;
        // leaf or subtree?  (surface-area heuristic over the hierarchy as it is: the optimal cut)
        auto area = [](const float4 p0, const float4 p1) { const float dx = p1.x - p0.x, dy = p1.y - p0.y, dz = p1.z - p0.z; return dx * dy + dy * dz + dz * dx; };
        const int cnt = a.last[cur] - a.first[cur] + 1;
        const float A = area(lo, hi);
        const float cl = (lc & kLbvhLeafBit) ? a.ct : a.cost[lc], cr = (rc & kLbvhLeafBit) ? a.ct : a.cost[rc];
        const float wl = A > 0.f ? area(l0, l1) / A : 1.f, wr = A > 0.f ? area(r0, r1) / A : 1.f;
        const float inner = 2.f * a.cb + wl * cl + wr * cr, leaf = a.ct * (float)cnt;
        const bool as_leaf = cnt <= kLbvhMinLeaf || (cnt <= kLbvhLeaf && leaf <= inner);
        a.cost[cur] = as_leaf ? leaf : inner;
        a.leafify[cur] = as_leaf ? 1 : 0;
        cur = a.parent[cur];
    }
}

// the surviving nodes in the reference's flat layout (optimized.cu:512-534); a node's number is the prefix sum of `alive`
__global__ __launch_bounds__(256) void lbvh_emit_kernel(const LbvhArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.n;
    if (c >= 2 * n - 1 || !a.alive[c]) return;
    float *o = a.arr10 + (size_t)a.index[c] * 10;
    auto ref_index = [&](int r) { return (float)a.index[(r & kLbvhLeafBit) ? (n - 1) + (r & ~kLbvhLeafBit) : r]; };
    float4 lo, hi;
    int s, e, lc = -1, rc = -1;
    if (c < n - 1) {
        lo = a.ibox_lo[c]; hi = a.ibox_hi[c]; s = a.first[c]; e = a.last[c] + 1;
        if (!a.leafify[c]) { lc = a.left[c]; rc = a.right[c]; }          // else: this node is a leaf, what hangs below it is gone
    } else {
        const int k = c - (n - 1);
        lo = a.lbox_lo[k]; hi = a.lbox_hi[k]; s = k; e = k + 1;
    }
    o[0] = lc == -1 ? -1.f : ref_index(lc);
    o[1] = rc == -1 ? -1.f : ref_index(rc);
    o[2] = lo.x; o[3] = lo.y; o[4] = lo.z; o[5] = hi.x; o[6] = hi.y; o[7] = hi.z;
    o[8] = (float)s; o[9] = (float)e;
    if (lc == -1) {                                                      // leaf statistics
        int depth = 0;
        for (int p = c < n - 1 ? a.parent[c] : a.leaf_parent[c - (n - 1)]; p >= 0 && depth < 4096; p = a.parent[p]) ++depth;
        atomicAdd(&a.stats[0], 1); atomicMax(&a.stats[1], e - s); atomicMax(&a.stats[2], depth);
    }
}

// ---- the kernels' own formats, straight from the builder's arrays (no trip through the host) --------------------------------------------
// rt_scene_upload converts a flat tree into what the render kernels read -- nodes in traversal order (pre-order, right child first) with
// skip pointers, the same nodes breadth-first as sibling pairs (boxes, and centre / half extent), triangles as 48-byte records in VISIT
// order, the levels for the device-side refit -- on the host (install_scene).  For a tree that was just built on the device that is a
// read-back, a re-layout and an upload of everything: 0.39 s for 2 M triangles, against 21 ms for the build itself.  The same formats
// follow from the builder's arrays by closed forms:
//   * a subtree of L leaves has 2 L - 1 nodes, and L = the number of leaf STARTS inside the node's range (a prefix sum over the sorted
//     positions), so a node's size needs no bottom-up pass;
//   * traversal order visits the right child first (cpu:291-292 push left, then right): x(right) = x(parent) + 1,
//     x(left) = x(parent) + 1 + size(right); summed over a node's ancestors (one walk up per node, which also yields depth and the path);
//   * a left child's range precedes its sibling's, so leaves are visited by DESCENDING range: the leaf covering [s, e) of n sorted
//     positions holds visit ranks [n - e, n - e + (e - s));
//   * breadth-first order with the right child first = sort by (depth, path bits with right = 0, left = 1): siblings come out adjacent and
//     every pair even-aligned behind the padding entry 0 and the root at 1.  (Ordering the pairs by their parent's traversal index instead
//     -- subtrees contiguous -- was measured on 2 M triangles: no faster, 17.2 ms either way; breadth-first keeps the top of the tree a
//     prefix, as the host path has it.)
struct LbvhLayout {
    int n_nodes;
    int *lscan;                          // [n + 1] exclusive prefix sum of the leaf-start flags
    int *X, *bfs;                        // [n_nodes] traversal-order / breadth-first (1-based) index of output node j
    unsigned long long *bkey; int *bval; // [n_nodes] sort key (depth << 58 | path) and the node it belongs to
    int *dhist;                          // [65] nodes per depth; [64] = deepest level
    float4 *node_lo, *node_hi, *nodes2, *nodesq, *nodesb;
    int *q2thr, *left_of, *lvl_nodes;
    int4 *tidx_visit, *tidx_up_new;
    int *perm;                           // [n] visit rank -> triangle's position in the new uploaded order (= sorted position)
};

__device__ __forceinline__ void lbvh_range(const LbvhArgs &a, int c, int &s, int &e) {
    if (c < a.n - 1) { s = a.first[c]; e = a.last[c] + 1; } else { s = c - (a.n - 1); e = s + 1; }
}
__device__ __forceinline__ int lbvh_cand(const LbvhArgs &a, int ref) { return (ref & kLbvhLeafBit) ? (a.n - 1) + (ref & ~kLbvhLeafBit) : ref; }
__device__ __forceinline__ bool lbvh_is_leaf(const LbvhArgs &a, int c) { return c >= a.n - 1 || a.leafify[c] != 0; }

__global__ __launch_bounds__(256) void lbvh_leafflag_kernel(const LbvhArgs a, int *flag) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * a.n - 1 || !a.alive[c] || !lbvh_is_leaf(a, c)) return;
    int s, e; lbvh_range(a, c, s, e);
    flag[s] = 1;
}

__global__ __launch_bounds__(256) void lbvh_walk_kernel(const LbvhArgs a, const LbvhLayout y) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * a.n - 1 || !a.alive[c]) return;
    int x = 0, depth = 0;
    unsigned long long key = 0ull;
    int cur = c;
    for (int p = cur < a.n - 1 ? a.parent[cur] : a.leaf_parent[cur - (a.n - 1)]; p >= 0 && depth < 4096; p = a.parent[p]) {
        const int lc = lbvh_cand(a, a.left[p]), rc = lbvh_cand(a, a.right[p]);
        const bool is_left = lc == cur;
        if (is_left) { int s, e; lbvh_range(a, rc, s, e); x += 2 * (y.lscan[e] - y.lscan[s]) - 1; }   // the right sibling's whole subtree comes first
        x += 1;
        if (depth < 58 && is_left) key |= 1ull << depth;
        depth++;
        cur = p;
    }
    const int j = a.index[c];
    y.X[j] = x;
    y.bkey[j] = (unsigned long long)(depth < 63 ? depth : 63) << 58 | key;
    y.bval[j] = j;
    atomicAdd(&y.dhist[depth < 63 ? depth : 63], 1);
    atomicMax(&y.dhist[64], depth);
}

__global__ __launch_bounds__(256) void lbvh_rank_kernel(const LbvhLayout y, const int *sorted_val) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= y.n_nodes) return;
    const int j = sorted_val[r];
    y.bfs[j] = r + 1;                                                    // index 0 of the breadth-first arrays is padding, the root is 1
    y.lvl_nodes[r] = y.X[j];                                             // nodes level by level (the refit walks them bottom-up)
}

__global__ __launch_bounds__(256) void lbvh_layout_kernel(const LbvhArgs a, const LbvhLayout y, const int4 *__restrict__ tidx_up_old) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.n;
    if (c >= 2 * n - 1 || !a.alive[c]) return;
    const int j = a.index[c], x = y.X[j], b = y.bfs[j];
    int s, e; lbvh_range(a, c, s, e);
    float4 lo, hi;
    if (c < n - 1) { lo = a.ibox_lo[c]; hi = a.ibox_hi[c]; } else { lo = a.lbox_lo[c - (n - 1)]; hi = a.lbox_hi[c - (n - 1)]; }
    float4 cb = make_float4(box_centre(lo.x, hi.x), box_centre(lo.y, hi.y), box_centre(lo.z, hi.z), 0.f);
    float4 hb = make_float4(box_half(lo.x, hi.x), box_half(lo.y, hi.y), box_half(lo.z, hi.z), 0.f);
    float4 ql = lo, qh = hi;
    if (!lbvh_is_leaf(a, c)) {
        const int jl = a.index[lbvh_cand(a, a.left[c])], jr = a.index[lbvh_cand(a, a.right[c])];
        const int size = 2 * (y.lscan[e] - y.lscan[s]) - 1;
        lo.w = __int_as_float(x + size); hi.w = __int_as_float(-1);     // next node on a box miss: past the subtree
        y.left_of[x] = y.X[jl];
        ql.w = __int_as_float(y.bfs[jr]); qh.w = __int_as_float(-1);     // first child of the pair: the right one (visited first)
        cb.w = __uint_as_float((unsigned int)y.bfs[jr] << kQNodeShift); hb.w = __int_as_float((int)0x80000000);
    } else {
        const int fv = n - e, cnt = e - s;                               // visit ranks of the leaf's triangles
        lo.w = __int_as_float(fv); hi.w = __int_as_float(fv + cnt);
        y.left_of[x] = -1;
        ql.w = lo.w; qh.w = hi.w;
        cb.w = __int_as_float(fv); hb.w = __int_as_float(cnt << kQLeafShift);
        for (int k = s; k < e; ++k) {
            y.tidx_visit[fv + (k - s)] = tidx_up_old[a.vals[k]];
            y.perm[fv + (k - s)] = k;
        }
    }
    y.node_lo[x] = lo; y.node_hi[x] = hi;
    y.nodes2[2 * x] = lo; y.nodes2[2 * x + 1] = hi;
    y.nodesq[2 * b] = ql; y.nodesq[2 * b + 1] = qh;
    y.nodesb[2 * b] = cb; y.nodesb[2 * b + 1] = hb;
    y.q2thr[b] = x;
}

__global__ __launch_bounds__(256) void lbvh_reorder_kernel(const LbvhArgs a, const int4 *__restrict__ tidx_up_old, int4 *__restrict__ tidx_up_new) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < a.n) tidx_up_new[k] = tidx_up_old[a.vals[k]];
}

}  // namespace rtk
