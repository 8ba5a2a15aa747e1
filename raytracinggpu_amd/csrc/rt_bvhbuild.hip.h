// rt_bvhbuild.hip.h -- device-side BVH build.  Included by rt_capi.hip.
//
// The reference builds its tree with TriangleMesh::buildBVH (cpu_launcher.cpp:190-224; device twin: the one-thread
// recursive `__device__ buildBVH` of global_launcher.cu:298-331 launched from KernelInit <<<1,1>>>, :848-881): top-down,
// box of the range (compute_bbox, cpu:180-188), longest axis (ties: x, then y), split at the midpoint, IN-PLACE partition of
// `indices` by centroid < split, stop if one side would be empty / a single triangle, or fewer than five triangles remain
// (cpu:217; the partition has already happened by then).
//
// Here the same tree is built level by level, one workgroup per node, and comes out bit for bit: boxes, node numbering of
// bvhTreeToArray (optimized.cu:512-534) and the order the triangles are left in.  Two things need care:
//
//  * compute_bbox folds std::min / std::max over the vertices in order: among equal values (+0 / -0) the FIRST one stays.
//    The parallel reduction therefore carries (value, position) pairs and prefers the smaller position on ties.
//  * the partition loop `if (cen < split) { swap(indices[i], indices[pivot]); ++pivot; }` is not stable for the elements
//    that stay right of the pivot: they form a FIFO -- every "less" element that arrives while the block [pivot, i) is not
//    empty takes the block's front element's place and sends that element to the back.  With E[j] the j-th element ever
//    appended to that FIFO (a new element, or the recycled front E[number of earlier recycles]) the final right part is
//    E[D .. ), D = number of recycles; the chains E[j] -> E[earlier] are resolved by pointer jumping.  "Less" elements keep
//    their order.
#pragma once
#include "rt_kernels.hip.h"

namespace rtk {

struct BuildArgs {
    const float4 *verts;       // current vertex positions (x, y, z, -)
    const int4 *tidx_up;       // vertex indices of triangle t, t in the order the mesh was uploaded
    int *idx;                  // [n_tris] the permutation being built: position -> uploaded triangle
    int *cnt, *ptr_a, *ptr_b, *tmp;   // [n_tris] scratch
    int *n_start, *n_end, *n_left, *n_right;   // [2 n_tris] nodes in allocation order (a level's nodes are contiguous)
    float4 *n_mn, *n_mx;
    int *counter;              // nodes allocated so far
    int n_tris;
    int cap;                   // node capacity of the n_* arrays: a split that would pass it is refused (the node stays a leaf, counter[1] is raised)
};

struct VP { float v; int p; };
__device__ __forceinline__ VP vp_min(VP a, VP b) { return (b.v < a.v || (b.v == a.v && b.p < a.p)) ? b : a; }   // std::min fold: first of equals stays
__device__ __forceinline__ VP vp_max(VP a, VP b) { return (a.v < b.v || (b.v == a.v && b.p < a.p)) ? b : a; }   // std::max fold: first of equals stays

constexpr int kBuildThreads = 256;

__global__ __launch_bounds__(kBuildThreads) void bvh_level_kernel(const BuildArgs a, const int first_node) {
    __shared__ float red_v[6][kBuildThreads];
    __shared__ int red_p[6][kBuildThreads];
    __shared__ int sh_i[8];
    __shared__ float sh_f[8];
    __shared__ int wave_tot[kBuildThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int node = first_node + blockIdx.x;
    const int s = a.n_start[node], e = a.n_end[node], n = e - s;
    // ---- 1. compute_bbox (cpu:180-188) ----
    VP mn[3], mx[3];
    for (int c = 0; c < 3; ++c) { mn[c].v = 1e9f; mn[c].p = -1; mx[c].v = -1e9f; mx[c].p = -1; }   // BoundingBox(): INF narrowed (cpu:135); position -1 = the initial value
    for (int k = tid; k < n; k += kBuildThreads) {
        const int4 ix = a.tidx_up[a.idx[s + k]];
        const int vi[3] = {ix.x, ix.y, ix.z};
        for (int j = 0; j < 3; ++j) {
            const float4 v = a.verts[vi[j]];
            const float c3[3] = {v.x, v.y, v.z};
            for (int c = 0; c < 3; ++c) {
                VP cur; cur.v = c3[c]; cur.p = 3 * k + j;
                mn[c] = vp_min(mn[c], cur); mx[c] = vp_max(mx[c], cur);
            }
        }
    }
    for (int c = 0; c < 3; ++c) { red_v[c][tid] = mn[c].v; red_p[c][tid] = mn[c].p; red_v[3 + c][tid] = mx[c].v; red_p[3 + c][tid] = mx[c].p; }
    __syncthreads();
    for (int w = kBuildThreads / 2; w > 0; w >>= 1) {
        if (tid < w) {
            for (int c = 0; c < 6; ++c) {
                VP x; x.v = red_v[c][tid]; x.p = red_p[c][tid];
                VP y; y.v = red_v[c][tid + w]; y.p = red_p[c][tid + w];
                const VP r = c < 3 ? vp_min(x, y) : vp_max(x, y);
                red_v[c][tid] = r.v; red_p[c][tid] = r.p;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float m0 = red_v[0][0], m1 = red_v[1][0], m2 = red_v[2][0], x0 = red_v[3][0], x1 = red_v[4][0], x2 = red_v[5][0];
        a.n_mn[node] = make_float4(m0, m1, m2, 0.f); a.n_mx[node] = make_float4(x0, x1, x2, 0.f);
        const float d0 = x0 - m0, d1 = x1 - m1, d2 = x2 - m2;             // diag = bb.mx - bb.mn (cpu:199)
        int axis = 2;
        if (d0 >= d1 && d0 >= d2) axis = 0;                                // cpu:200-205
        else if (d1 >= d0 && d1 >= d2) axis = 1;
        const float lo = axis == 0 ? m0 : axis == 1 ? m1 : m2, hi = axis == 0 ? x0 : axis == 1 ? x1 : x2;
        sh_i[0] = axis;
        sh_f[0] = (lo + hi) / 2;                                           // cpu:206
        sh_i[1] = n;                                                       // first "not less" position, min-reduced below
    }
    __syncthreads();
    const int axis = sh_i[0];
    const float split = sh_f[0];
    // ---- 2. classify (cpu:209-215) and count the "less" elements before every position ----
    int running = 0;                                                       // "less" elements in the chunks already scanned
    for (int base = 0; base < n; base += kBuildThreads) {
        const int k = base + tid;
        bool less = false;
        if (k < n) {
            const int4 ix = a.tidx_up[a.idx[s + k]];
            const float4 va = a.verts[ix.x], vb = a.verts[ix.y], vc = a.verts[ix.z];
            const float ca = axis == 0 ? va.x : axis == 1 ? va.y : va.z, cb = axis == 0 ? vb.x : axis == 1 ? vb.y : vb.z, cc = axis == 0 ? vc.x : axis == 1 ? vc.y : vc.z;
            const float cen = (ca + cb + cc) / 3;                         // centroid coordinate, cpu:211
            less = cen < split;
        }
        const unsigned long long m = __ballot(less);
        const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int before = running;
        for (int w = 0; w < wv; ++w) before += wave_tot[w];
        int chunk_total = 0;
        for (int w = 0; w < kBuildThreads / 64; ++w) chunk_total += wave_tot[w];
        if (k < n) {
            const int c = before + in_wave;                                // "less" elements at positions < k
            a.cnt[s + k] = less ? ~c : c;                                  // sign bit = less
            if (!less) atomicMin(&sh_i[1], k);
        }
        running += chunk_total;
        __syncthreads();
    }
    const int total_less = running, first_g = sh_i[1];
    // ---- 3. the right part as a FIFO: ptr[k] = k for an element that stays right, else the position whose element it recycles ----
    int *pa = a.ptr_a + s, *pb = a.ptr_b + s;
    for (int k = first_g + tid; k < n; k += kBuildThreads) {
        const int c = a.cnt[s + k];
        pa[k] = c < 0 ? ~c : k;                                            // ~c = number of "less" before k >= first_g, and < k
    }
    __syncthreads();
    for (int span = 1; span < n; span <<= 1) {                             // pointer jumping: after r rounds chains of 2^r hops are resolved
        for (int k = first_g + tid; k < n; k += kBuildThreads) pb[k] = pa[pa[k]];
        __syncthreads();
        int *t = pa; pa = pb; pb = t;
    }
    // ---- 4. scatter: "less" elements in order, then the FIFO's final content E[D ..) ----
    const int D = total_less - first_g;                                    // recycles = "less" elements that met a non-empty block
    for (int k = tid; k < n; k += kBuildThreads) {
        const int c = a.cnt[s + k];
        if (c < 0) a.tmp[s + ~c] = a.idx[s + k];
        if (k >= first_g + D) a.tmp[s + total_less + (k - first_g - D)] = a.idx[s + pa[k]];
    }
    __syncthreads();
    for (int k = tid; k < n; k += kBuildThreads) a.idx[s + k] = a.tmp[s + k];
    // ---- 5. stop rule (cpu:217) or two children ----
    if (tid == 0) {
        const int pivot = s + total_less;
        int l = -1, r = -1;
        if (!(pivot <= s || pivot >= e - 1 || e - s < 5)) {
            l = atomicAdd(a.counter, 2); r = l + 1;
            if (r >= a.cap) { a.counter[1] = 1; l = -1; r = -1; }          // cannot happen for a tree of n triangles (<= 2 n - 1 nodes): refuse rather than write past the arrays
            else { a.n_start[l] = s; a.n_end[l] = pivot; a.n_start[r] = pivot; a.n_end[r] = e; }
        }
        a.n_left[node] = l; a.n_right[node] = r;
    }
}

// Numbering of bvhTreeToArray (optimized.cu:512-534): pre-order, left subtree first; then the float[10] records.
// One workgroup; lvl_first[L] .. lvl_first[L + 1] are the nodes (allocation order) of level L.
__global__ __launch_bounds__(1024) void bvh_flatten_kernel(const BuildArgs a, const int *__restrict__ lvl_first, const int n_levels, int *__restrict__ size,
                                                           int *__restrict__ pre, float *__restrict__ arr10) {
    for (int L = n_levels - 1; L >= 0; --L) {                              // subtree sizes, bottom-up
        for (int x = lvl_first[L] + (int)threadIdx.x; x < lvl_first[L + 1]; x += (int)blockDim.x)
            size[x] = a.n_left[x] < 0 ? 1 : 1 + size[a.n_left[x]] + size[a.n_right[x]];
        __syncthreads();
    }
    if (threadIdx.x == 0) pre[0] = 0;
    __syncthreads();
    for (int L = 0; L < n_levels; ++L) {                                   // pre-order numbers, top-down
        for (int x = lvl_first[L] + (int)threadIdx.x; x < lvl_first[L + 1]; x += (int)blockDim.x) {
            if (a.n_left[x] >= 0) { pre[a.n_left[x]] = pre[x] + 1; pre[a.n_right[x]] = pre[x] + 1 + size[a.n_left[x]]; }
        }
        __syncthreads();
    }
    const int n_nodes = lvl_first[n_levels];
    for (int x = (int)threadIdx.x; x < n_nodes; x += (int)blockDim.x) {
        float *o = arr10 + 10 * (size_t)pre[x];
        const float4 mn = a.n_mn[x], mx = a.n_mx[x];
        o[0] = a.n_left[x] < 0 ? -1.f : (float)pre[a.n_left[x]];
        o[1] = a.n_right[x] < 0 ? -1.f : (float)pre[a.n_right[x]];
        o[2] = mn.x; o[3] = mn.y; o[4] = mn.z; o[5] = mx.x; o[6] = mx.y; o[7] = mx.z;
        o[8] = (float)a.n_start[x]; o[9] = (float)a.n_end[x];
    }
}

__global__ __launch_bounds__(256) void iota_kernel(int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

}  // namespace rtk
