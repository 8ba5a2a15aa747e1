// rt_meshops.hip.h -- device-side mesh transform + BVH refit.  Included by rt_capi.hip.
//
// transform_kernel  = the `transform` kernel of global_launcher.cu:340-365 (realtime_render.cu:1151-1166 launches the
//                     same): v' = R v (row-major 3x3, products summed left to right) then += translation.
// retri_kernel      = the per-triangle precompute of rt_scene_upload on the device: (A, e1, e2, N) with the operations
//                     of moller_trumbore (cpu:227-229), triangles in visit order.
// refit_kernel      = every node's box recomputed for the EXISTING tree and triangle order: a leaf's box is
//                     compute_bbox of its triangles (cpu:180-188: INF-initialised std::min / std::max), an internal
//                     node's box the union of its children's, which is compute_bbox of its whole range.  One workgroup
//                     walks the levels bottom-up (the tree has a few thousand nodes; this is a step before the hot path).
// The reference never refits (its transform variant has no BVH and buildBVH runs once on the host); rebuilding on the
// host after a transform remains possible through rt_scene_upload.
#pragma once
#include "rt_travq.hip.h"

namespace rtk {

struct Mat3 { float r[9]; float t[3]; };

__global__ __launch_bounds__(256) void transform_kernel(float4 *__restrict__ verts, int nv, const Mat3 m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const float4 v = verts[i];
    float x = m.r[0] * v.x + m.r[1] * v.y + m.r[2] * v.z;
    float y = m.r[3] * v.x + m.r[4] * v.y + m.r[5] * v.z;
    float z = m.r[6] * v.x + m.r[7] * v.y + m.r[8] * v.z;
    x += m.t[0]; y += m.t[1]; z += m.t[2];
    verts[i] = make_float4(x, y, z, 0.f);
}

__global__ __launch_bounds__(256) void retri_kernel(const int4 *__restrict__ tidx, const float4 *__restrict__ verts, float4 *__restrict__ tri, int nt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const int4 ix = tidx[t];
    const float4 a = verts[ix.x], b = verts[ix.y], c = verts[ix.z];
    const f3 A = mk(a.x, a.y, a.z), B = mk(b.x, b.y, b.z), C = mk(c.x, c.y, c.z);
    const f3 e1 = B - A, e2 = C - A, N = cross(e1, e2);               // cpu:227-229
    tri[3 * (size_t)t + 0] = make_float4(A.x, A.y, A.z, e1.x);
    tri[3 * (size_t)t + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    tri[3 * (size_t)t + 2] = make_float4(e2.z, N.x, N.y, N.z);
}

struct RefitArgs {
    float4 *node_lo, *node_hi, *nodes2, *nodesq, *nodesb;   // the node layouts (pre-order SoA, pre-order interleaved, breadth-first as boxes and as centre / half extent)
    const int *q2thr, *left_of, *lvl_nodes, *lvl_off;
    const int4 *tidx;
    const float4 *verts;
    int n_nodes, n_levels;
};

__global__ __launch_bounds__(1024) void refit_kernel(const RefitArgs a) {
    for (int L = a.n_levels - 1; L >= 0; --L) {
        for (int k = a.lvl_off[L] + (int)threadIdx.x; k < a.lvl_off[L + 1]; k += (int)blockDim.x) {
            const int x = a.lvl_nodes[k];
            float4 lo = a.node_lo[x], hi = a.node_hi[x];
            const int hiw = __float_as_int(hi.w), low = __float_as_int(lo.w);
            f3 mn = mk(1e9f, 1e9f, 1e9f), mx = mk(-1e9f, -1e9f, -1e9f);          // BoundingBox(), cpu:135 (INF narrowed)
            if (hiw >= 0) {                                            // leaf: compute_bbox over triangles [low, hiw)
                for (int t = low; t < hiw; ++t) {
                    const int4 ix = a.tidx[t];
                    const int vi[3] = {ix.x, ix.y, ix.z};
                    for (int j = 0; j < 3; ++j) {
                        const float4 v = a.verts[vi[j]];
                        mn.x = v.x < mn.x ? v.x : mn.x; mn.y = v.y < mn.y ? v.y : mn.y; mn.z = v.z < mn.z ? v.z : mn.z;   // std::min(mn, v)
                        mx.x = mx.x < v.x ? v.x : mx.x; mx.y = mx.y < v.y ? v.y : mx.y; mx.z = mx.z < v.z ? v.z : mx.z;   // std::max(mx, v)
                    }
                }
            } else {                                                   // internal: children x + 1 and left_of[x] (one level down: done)
                const int c[2] = {x + 1, a.left_of[x]};
                for (int j = 0; j < 2; ++j) {
                    const float4 cl = a.node_lo[c[j]], ch = a.node_hi[c[j]];
                    mn.x = cl.x < mn.x ? cl.x : mn.x; mn.y = cl.y < mn.y ? cl.y : mn.y; mn.z = cl.z < mn.z ? cl.z : mn.z;
                    mx.x = mx.x < ch.x ? ch.x : mx.x; mx.y = mx.y < ch.y ? ch.y : mx.y; mx.z = mx.z < ch.z ? ch.z : mx.z;
                }
            }
            lo.x = mn.x; lo.y = mn.y; lo.z = mn.z; hi.x = mx.x; hi.y = mx.y; hi.z = mx.z;
            a.node_lo[x] = lo; a.node_hi[x] = hi;
        }
        __threadfence_block();
        __syncthreads();
    }
    // the other two layouts carry the same boxes (their .w fields keep their own meaning)
    for (int x = (int)threadIdx.x; x < a.n_nodes; x += (int)blockDim.x) {
        const float4 lo = a.node_lo[x], hi = a.node_hi[x];
        float4 l2 = a.nodes2[2 * x], h2 = a.nodes2[2 * x + 1];
        l2.x = lo.x; l2.y = lo.y; l2.z = lo.z; h2.x = hi.x; h2.y = hi.y; h2.z = hi.z;
        a.nodes2[2 * x] = l2; a.nodes2[2 * x + 1] = h2;
    }
    for (int k = 1 + (int)threadIdx.x; k <= a.n_nodes; k += (int)blockDim.x) {   // breadth-first array: node k lives at index k (0 is padding)
        const int x = a.q2thr[k];
        const float4 lo = a.node_lo[x], hi = a.node_hi[x];
        float4 lq = a.nodesq[2 * k], hq = a.nodesq[2 * k + 1];
        lq.x = lo.x; lq.y = lo.y; lq.z = lo.z; hq.x = hi.x; hq.y = hi.y; hq.z = hi.z;
        a.nodesq[2 * k] = lq; a.nodesq[2 * k + 1] = hq;
        float4 cb = a.nodesb[2 * k], hb = a.nodesb[2 * k + 1];        // wf_travq's form of the same box (their .w fields keep payload and kind)
        cb.x = box_centre(lo.x, hi.x); cb.y = box_centre(lo.y, hi.y); cb.z = box_centre(lo.z, hi.z);
        hb.x = box_half(lo.x, hi.x); hb.y = box_half(lo.y, hi.y); hb.z = box_half(lo.z, hi.z);
        a.nodesb[2 * k] = cb; a.nodesb[2 * k + 1] = hb;
    }
}

}  // namespace rtk
