// rt_kernels.hip.h -- gfx950 render kernels of libraytrace_hip.so.
//
// One lane per pixel, one wave64 per 8x8 pixel tile, four waves (32x8 pixels) per
// workgroup.  Replaces KernelLaunch (optimized.cu:670-772) / the pixel loop of
// cpu_launcher.cpp:693-718.  Written for CDNA4 only; compiled with
// -ffp-contract=off and IEEE-correct f32 divide/sqrt so that every arithmetic
// operation is the same single rounding the reference performs (DESIGN.md
// "Numerics").  No MFMA: there is no dense contraction on this path.
//
// Data layout (built by rt_scene_upload: install_scene, rt_host_scene.hip.h):
//   node_lo[n], node_hi[n]  float4 SoA, nodes in TRAVERSAL order (pre-order,
//        right child first = the order cpu_launcher.cpp:284-293 pops them).
//        lo = (mn.x, mn.y, mn.z, bits(next-on-miss | tri_start))
//        hi = (mx.x, mx.y, mx.z, bits(-1 internal | tri_end leaf))
//        Because the reference never prunes by distance the visit set
//        is a pure function of the ray, so traversal needs no stack: on a box hit
//        go to node+1, on a miss of an internal node jump past its subtree.
//   tri[3*t .. 3*t+2]       float4 x3 per triangle in the reference's BVH order:
//        (A.xyz, e1.x) (e1.yz, e2.xy) (e2.z, N.xyz), e1=B-A, e2=C-A, N=e1 x e2
//        computed on the host with the same IEEE operations cpu:227-229 performs.
//   verts[nv] float4, tidx[nt] int4   (LDS-staged variants)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_sincos.h"
#include "rt_div.h"

namespace rtk {

constexpr int kMaxSpheres = 16;
constexpr int kMaxMeshes = 16;       // RT_MAX_OBJECTS: every object of a scene may be a mesh
constexpr int kMaxSegments = 16;
constexpr int kBlockThreads = 256;   // 4 waves: 32 x 8 pixels
constexpr int kTileW = 32, kTileH = 8;

struct Sphere {           // the geometry of rt_sphere, cpu:505-511 (its material sits in the per-object tables of Scene)
    float cx, cy, cz, R;
    float R2;             // R * R as the reference evaluates it in cpu:513 (one binary32 product), computed once by rt_scene_upload
    int obj;              // position in Scene::objects (cpu:541): decides exact ties (strict '<', cpu:554) and names the object in the path records
};

// A TriangleMesh of the scene: where its triangles sit in the visit-order arrays.  The meshes are stored one after the other in OBJECT order, so the
// 64-bit minimum over bits(t) << 32 | triangle index that the traversal kernels form is the minimum over (t, object position, scan rank): what the
// reference's loop over the objects keeps (cpu:549-558) with its strict '<'.
struct MeshRec {
    int tri_begin;        // first triangle (visit order) of this mesh; the next record's tri_begin (or n_tris) ends it
    int obj;              // position in Scene::objects
};

struct Scene {
    Sphere sph[kMaxSpheres];
    int n_spheres;
    int n_objects;        // spheres + meshes
    int mesh_slot;        // object index of the FIRST mesh with triangles, -1 = none ("the scene has a mesh to traverse")
    int n_meshes;         // every TriangleMesh of the scene in object order, mesh[0 .. n_meshes): one without triangles (a missing OBJ, cpu:322-325) holds
                          // its position in Scene::objects and an empty triangle range
    MeshRec mesh[kMaxMeshes];
    // Geometry's fields (cpu:106-118) by OBJECT id, sphere or mesh alike -- Scene::getColor reads objects[id]->mirror / the indices / the albedo of whichever
    // object was hit (cpu:573-606, 624, 642): one dynamically indexed read each, no search
    float4 obj_a[kMaxSpheres];   // (sphere centre xyz | 0 for a mesh, mirror as an int's bits)
    float4 obj_b[kMaxSpheres];   // (albedo rgb, -)
    float2 obj_n[kMaxSpheres];   // (in_refraction_index, out_refraction_index)
    float Lx, Ly, Lz, intensity;
    float camx, camy, camz, fov;
    const float4 *node_lo, *node_hi;
    const float4 *nodes;      // the same nodes interleaved: nodes[2i] = lo, nodes[2i+1] = hi (one address per visit)
    const float4 *nodesq;     // interleaved, for the work-stack traversal, in BREADTH-FIRST order from index 1 (index 0 is padding, so
                              // sibling pairs are aligned 64-byte lines): lo.w of an internal node = its first child (even), the second
                              // one is stored next to it (the root is node 1, its children 2 and 3)
    const int *q2thr;         // nodesq index -> index of the same node in `nodes`
    const float4 *nodesb;     // the same breadth-first array in the form wf_travq reads (rt_travq.hip.h): (centre.xyz, payload) (half extent.xyz, kind)
    float bmx, bmy, bmz;      // per axis: max |bound| over every node box (the absolute term of wf_travq's box filter)
    const uint4 *nodesh;      // the sibling pairs once more in 16-bit fixed point, 32 bytes per pair (rt_travq.hip.h, QN): per child (centre.xyz, half extent.xyz) on the
                              // grid below, rounded OUTWARDS, and one payload word (rt_qnodes.hip.h); nullptr = not available for this tree
    const int *tri2leaf;      // triangle (visit order) -> breadth-first index of its leaf (the exact box of a flagged leaf: rt_qnodes.hip.h)
    const float4 *leaflh;     // (lo, hi) of the leaf that holds triangle i (visit order) at [2 i], [2 i + 1]: what the check of a triangle accepted in a FLAGGED leaf reads in one hop (rt_travq.hip.h)
    const uint4 *nodesw;      // 4-wide fixed-point nodes (rt_travq.hip.h, QW): for the sibling pair (c, c + 1) the 64 bytes at byte offset 32 c hold the nodesh records of up to
                              // four nodes that cut the subtree below the pair's parent -- the cut a surface-area DP picks (rt_qnodes.hip.h; RT_TRAVQ_QSEL=0: the children of c
                              // and of c + 1), internal nodes first, a free place = 0; nullptr = not in use
    float qgx, qgy, qgz;      // grid origin (the root box's lower corner) and cell size per axis
    float qsx, qsy, qsz;
    int qleaf_shift;          // a leaf's payload word in nodesh = 1 << 31 | count << qleaf_shift | first triangle (20 or 24: rt_qnodes.hip.h)
    int fast_box;             // every node box is finite, ordered (lo <= hi) and below 1e8 in magnitude: the centre / half-extent filter may decide
    const float4 *nrm;        // smooth shading (wavefront variants): 3 vertex normals per triangle, visit order; nullptr = flat
    const float4 *tri;
    const float4 *verts;
    const int4 *tidx;
    int n_nodes, n_tris, n_verts;
    float4 root_lo, root_hi;   // node 0 again, as kernel arguments (SGPRs) for the uniform root-box pre-test
};

struct Frame {
    int W, H, spp, segs;
    float sigma, eps, tri_tmin, z;
    uint32_t seed;
    int row0, n_rows, tile_rows, tile_step;
    float4 *out;
    unsigned long long *work;   // STATS kernels only: [24] {rays, box_tests, nodes, tri_tests, invariant mask, literal box tests, literal triangle tests, -,
                                // wf_travq step counters [8..19]: loop iterations, refill passes, refill rounds, queue fetches, TRI steps, BOX steps, literal-box fall-backs,
                                // serial drains, t-division blocks, first / second leaf pushes, TRI steps that stopped a shadow ray (any-hit)}
    int out_tile0, out_tile_step;   // local row r is stored at output row ((r / tile_rows) * out_tile_step + out_tile0) * tile_rows + r % tile_rows
    // cam_mode 1 = realtime_render.cu's camera and sample averaging (KernelLaunch realtime:1100-1134; wavefront variants only):
    // u_center = C + bz * z + bx * X + by * Y, every sample weighted by inv_n = (float)(1. / num_rays) as it is added
    int cam_mode;
    float bx[3], by[3], bz[3];
    float inv_n;
};
__device__ __forceinline__ size_t out_index(const Frame &fr, int lrow, int px) {
    const int orow = ((lrow / fr.tile_rows) * fr.out_tile_step + fr.out_tile0) * fr.tile_rows + lrow % fr.tile_rows;
    return (size_t)orow * fr.W + px;
}

// per-lane traversal work counters (STATS instantiation only; SURVEY 8d accounting)
struct Work { uint32_t box = 0, nodes = 0, tris = 0, rays = 0, lit_box = 0, lit_tri = 0; };   // lit_*: tests decided by the literal divisions (filter undecided)

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(float s, f3 b) { return mk(s * b.x, s * b.y, s * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// IEEE-754 correctly rounded binary32 square root.  NOT __fsqrt_rn: ROCm 7.2 maps that to the native
// (1-ulp) v_sqrt_f32; sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt gets the fix-up sequence.
// The compiler's sequence is: scale arguments below 2^-96 up, v_sqrt_f32 (1 ulp), pick among s - 1 ulp, s, s + 1 ulp by the signs of the
// residuals fma(-s', s, x), scale back, pass +-0 / +inf through -- 17 instructions of which only the middle eight matter for an argument in
// [2^-96, inf): there no scaling happens, the residuals are normal numbers and the result is what the full sequence returns (it IS that
// sequence minus the parts that do nothing for such an argument).  Everything else -- zero, tiny, negative, inf, NaN -- takes the builtin
// behind a wave-uniform branch.
__device__ __forceinline__ float rt_sqrtf(float x) {
    const bool plain = (__float_as_uint(x) - 0x0f800000u) < (0x7f800000u - 0x0f800000u);     // 2^-96 <= x < inf (false for negatives and NaN)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    float r = (0.f >= rd) ? sd : s;
    r = (0.f < ru) ? su : r;
    if (__builtin_expect(__ballot(!plain) != 0ull, 0)) {
        if (!plain) r = __builtin_sqrtf(x);
    }
    return r;
}
__device__ __forceinline__ float norm2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// cpu:58-63: three divisions by sqrt(norm2).  The three quotients share one reciprocal (rt_div.h: the compiler's own correctly rounded
// sequence with its denominator part done once); lanes with a component or a norm outside [2^-60, 2^60] -- zero, denormal, overflowed,
// NaN -- take the literal divisions behind a wave-uniform branch.
__device__ __forceinline__ f3 normalize(f3 a, float &n_out) {         // n_out: the norm the components were divided by
    const float n = rt_sqrtf(norm2(a));
    n_out = n;
    float mn, mx;
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(mn) : "v"(a.x), "v"(a.y), "v"(a.z));
    const bool fast = mn >= kDivLo && n <= kDivHi;                     // n >= every |component| >= 2^-60 then; false for NaN
    (void)mx;
    const float r1 = div_refine(n, __builtin_amdgcn_rcpf(n));
    f3 q = mk(div_by(a.x, n, r1), div_by(a.y, n, r1), div_by(a.z, n, r1));
    if (__builtin_expect(__ballot(!fast) != 0ull, 0)) {
        if (!fast) q = mk(a.x / n, a.y / n, a.z / n);
    }
    return q;
}
__device__ __forceinline__ f3 normalize(f3 a) { float n; return normalize(a, n); }
// the same for a vector whose THIRD listed component is the literal +0 (T1 of cpu:634-636: (-Ny, Nx, 0) or (-Nz, 0, Nx)): +0 / n = +0
// for every n the fast range admits, so only two quotients are formed; (p, q) are the other two components
__device__ __forceinline__ void normalize_pq0(float p, float q, float &op, float &oq, float &ozero) {
    const float n = rt_sqrtf(p * p + q * q + 0.f * 0.f);
    const bool fast = fminf(fabsf(p), fabsf(q)) >= kDivLo && n <= kDivHi;
    const float r1 = div_refine(n, __builtin_amdgcn_rcpf(n));
    op = div_by(p, n, r1); oq = div_by(q, n, r1); ozero = 0.f;
    if (__builtin_expect(__ballot(!fast) != 0ull, 0)) {
        if (!fast) { op = p / n; oq = q / n; ozero = 0.f / n; }
    }
}

// counter RNG, DESIGN.md "RNG" (same definition as the oracle's or_uniform)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float uniform01(uint32_t hs, uint32_t depth, uint32_t dim) {
    uint32_t h = mix32(hs ^ ((depth * 4U + dim) * 0x85EBCA77U));
    return (float)((h >> 8) + 1U) * 0x1p-24f;
}

// BoundingBox::intersect, cpu:146-157, literal (division by u, swap, min_element/max_element order)
__device__ __forceinline__ bool slab(float4 lo, float4 hi, f3 O, f3 u) {
    float t0x = (lo.x - O.x) / u.x, t0y = (lo.y - O.y) / u.y, t0z = (lo.z - O.z) / u.z;
    float t1x = (hi.x - O.x) / u.x, t1y = (hi.y - O.y) / u.y, t1z = (hi.z - O.z) / u.z;
    float tmp;
    if (t0x > t1x) { tmp = t0x; t0x = t1x; t1x = tmp; }
    if (t0y > t1y) { tmp = t0y; t0y = t1y; t1y = tmp; }
    if (t0z > t1z) { tmp = t0z; t0z = t1z; t1z = tmp; }
    float mn = t1x; if (t1y < mn) mn = t1y; if (t1z < mn) mn = t1z;
    float mx = t0x; if (mx < t0y) mx = t0y; if (mx < t0z) mx = t0z;
    return mn > mx;
}

// ---- error-bounded filters (DESIGN.md "Filtered exact arithmetic") ------------------------------------
// The reference decides box and barycentric tests on correctly rounded quotients a/b.  A correctly
// rounded f32 division costs ~12 instructions on gfx950, and a box test needs six.  The filters below
// evaluate the same quotients as a * rcp(b) (v_rcp_f32: <= 1 ulp), bound the distance to the exactly
// rounded quotient by kRel*|q| + kAbs, and decide the comparison only when it cannot depend on that
// distance; otherwise (and for any non-finite, huge or denormal-range operand) the literal code runs.
// The decision is therefore bit-identical to the literal code by construction, not by sampling.
//   |a*rcp(b) - RN(a/b)| <= (2^-23 [rcp] + 2^-24 [mul] + 2^-24 [RN]) |a/b| = 2^-22 |a/b|;  kRel = 2^-21.
constexpr float kRel = 0x1p-21f;
constexpr float kAbs = 1e-35f;     // covers the absolute error of results in the denormal range
constexpr float kBig = 1e30f;      // beyond this the product may overflow where the quotient does not
constexpr float kTiny = 1e-30f;    // divisors below this are not trusted to v_rcp_f32

struct RayInv { float x, y, z; bool safe; };
__device__ __forceinline__ RayInv ray_inv(f3 u) {
    RayInv r;
    r.x = __builtin_amdgcn_rcpf(u.x); r.y = __builtin_amdgcn_rcpf(u.y); r.z = __builtin_amdgcn_rcpf(u.z);
    const float m = fminf(fminf(fabsf(u.x), fabsf(u.y)), fabsf(u.z));
    r.safe = m > kTiny && fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fabsf(u.z)) < kBig;   // false for 0, denormal, inf, NaN
    return r;
}

// BoundingBox::intersect through the filter.  For finite quotients the literal swap/min_element/max_element
// sequence (cpu:153-156) equals min over axes of max(t0,t1) > max over axes of min(t0,t1).
// `decided` reports whether the filter settled the test (false: the literal code ran).
__device__ __forceinline__ bool slab_filtered(float4 lo, float4 hi, f3 O, f3 u, const RayInv &r, bool &decided) {
    decided = true;
    if (r.safe) {
        const float ax = (lo.x - O.x) * r.x, bx = (hi.x - O.x) * r.x;
        const float ay = (lo.y - O.y) * r.y, by = (hi.y - O.y) * r.y;
        const float az = (lo.z - O.z) * r.z, bz = (hi.z - O.z) * r.z;
        const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        const float M = fmaxf(fmaxf(fmaxf(fabsf(ax), fabsf(bx)), fmaxf(fabsf(ay), fabsf(by))), fmaxf(fabsf(az), fabsf(bz)));
        const float d = tf - tn;
        // every quotient is within kRel*M+kAbs of its exact value and min/max are 1-Lipschitz, so both tn and tf
        // are; d > 2*band  =>  exact (min > max) is true, d < -2*band => false.  (r.safe => no NaN can occur.)
        const float band = 2.f * (M * kRel + kAbs);
        if (M < kBig) {
            if (d > band) return true;
            if (d < -band) return false;
        }
    }
    decided = false;
    return slab(lo, hi, O, u);
}
__device__ __forceinline__ bool slab_filtered(float4 lo, float4 hi, f3 O, f3 u, const RayInv &r) {
    bool decided;
    return slab_filtered(lo, hi, O, u, r, decided);
}

// Sphere::intersect, cpu:512-527 (the normal, cpu:524-525, is evaluated by the caller for the winning object only), in two
// parts: what depends on the ray's origin only -- shared by the shadow ray and the bounce ray that leave one hit point -- and
// the rest.  cpu:516-517 evaluate dot(u, C - O) next to dot(u, O - C) = d: every difference, product and sum of the one is the
// exact negation of the other's (round-to-nearest is symmetric), so b = -d bit for bit -- except for the SIGN of a zero result
// (x - x is +0 either way round), which can only reach b when d == 0: then, and only then, b is evaluated literally.
struct SphereOrigin { f3 OC; float c; };
__device__ __forceinline__ SphereOrigin sphere_origin(const Sphere &s, f3 O) {
    SphereOrigin so;
    so.OC = O - mk(s.cx, s.cy, s.cz);
    so.c = norm2(so.OC) - s.R2;                                // cpu:513: (O - C).norm2() - R * R
    return so;
}
__device__ __forceinline__ bool sphere_dir(const Sphere &s, const SphereOrigin &so, f3 O, f3 u, float &t) {
    const float d = dot(u, so.OC);
    const float delta = d * d - so.c;                          // cpu:513
    if (delta < 0) return false;
    const float sq = rt_sqrtf(delta);
    float b = -d;
    if (__builtin_expect(__ballot(d == 0.f) != 0ull, 0)) {     // wave-uniform: the literal dot product only where a zero's sign could differ
        if (d == 0.f) b = dot(u, mk(s.cx, s.cy, s.cz) - O);
    }
    const float t1 = b - sq, t2 = b + sq;                      // cpu:516-517
    if (t2 < 0) return false;
    t = t1 < 0 ? t2 : t1;
    return true;
}
__device__ __forceinline__ bool sphere_test(const Sphere &s, f3 O, f3 u, float &t) { return sphere_dir(s, sphere_origin(s, O), O, u, t); }

// TriangleMesh::intersect, cpu:277-311.  Returns true iff some triangle was accepted
// (SURVEY H4); t/Nraw are the nearest accepted t and its unnormalised e1 x e2.
template <bool STATS>
__device__ __forceinline__ bool mesh_intersect(const Scene &sc, f3 O, f3 u, float tri_tmin, float &t_out, f3 &N_out, int &tri_out, Work &wk) {
    float t_min = 1e9f;   // INF (1e9+9) narrowed to float, cpu:283
    bool any = false;
    int tri_best = 0;     // the winning triangle (visit order): with several meshes in the tree it names the mesh -- the walk reaches the meshes in object order, so the strict '<'
                          // below keeps, among equal t, the earliest object and inside it the earliest triangle of the reference's scan
    f3 Nb = mk(0, 0, 0);
    int node = 0;
    const int n_nodes = sc.n_nodes;
    const RayInv ri = ray_inv(u);
    while (node < n_nodes) {
        const float4 lo = sc.node_lo[node];
        const float4 hi = sc.node_hi[node];
        const int hiw = __float_as_int(hi.w);
        const int low = __float_as_int(lo.w);
        if (STATS) wk.box++;
        if (slab_filtered(lo, hi, O, u, ri)) {
            if (STATS) wk.nodes++;
            if (hiw >= 0) {   // leaf: triangles [low, hiw), ascending (cpu:295)
                if (STATS) wk.tris += (uint32_t)(hiw - low);
                for (int i = low; i < hiw; ++i) {
                    const float4 q0 = sc.tri[3 * i + 0], q1 = sc.tri[3 * i + 1], q2 = sc.tri[3 * i + 2];
                    const f3 A = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
                    const f3 N = mk(q2.y, q2.z, q2.w);
                    // moller_trumbore, cpu:226-236
                    const float det = dot(u, N);
                    if (det == 0) continue;
                    const f3 AO = A - O;
                    const f3 c = cross(AO, u);
                    const float bn = dot(e2, c);
                    const float gn = -dot(e1, c);
                    bool pass = false;
                    if (fabsf(det) > kTiny) {   // filter on beta, gamma, beta+gamma (same bound as the box filter)
                        const float rd = __builtin_amdgcn_rcpf(det);
                        const float b = bn * rd, g = gn * rd;
                        const float eb = fabsf(b) * kRel + kAbs, eg = fabsf(g) * kRel + kAbs;
                        if (b < -eb || b > 1.f + eb || g < -eg || g > 1.f + eg) continue;        // certainly outside [0,1]
                        const float sum = b + g;
                        const float es = eb + eg + fabsf(sum) * 0x1p-22f;
                        if (sum > 1.f + es) continue;                                             // certainly beta+gamma > 1
                        pass = b >= eb && b <= 1.f - eb && g >= eg && g <= 1.f - eg && sum <= 1.f - es;
                    }
                    if (!pass) {   // undecided (or NaN/inf/tiny det): the literal tests
                        const float beta = bn / det;
                        const float gamma = gn / det;
                        if (!(0 <= beta && beta <= 1) || !(0 <= gamma && gamma <= 1)) continue;
                        if (!(beta + gamma <= 1)) continue;
                    }
                    const float t = dot(AO, N) / det;   // exact: t is compared and returned
                    if (!(t > 0)) continue;
                    if (t > tri_tmin && t < t_min) { t_min = t; Nb = N; any = true; tri_best = i; }   // cpu:301
                }
            }
            node = node + 1;
        } else {
            node = (hiw < 0) ? low : node + 1;
        }
    }
    t_out = t_min;
    N_out = Nb;
    tri_out = tri_best;
    return any;
}

// Scene::intersect_all, cpu:545-564.  Objects are visited in insertion order with a
// strict '<' (exact-tie behaviour); the winner's normal is evaluated once at the end
// (it is a pure function of the winner, so this is bit-identical to cpu:524-525/308).
// the mesh that holds triangle `tri` (visit order): wave-uniform loop over the scene's meshes, no iteration for the usual single mesh
__device__ __forceinline__ int mesh_of_tri(const Scene &sc, int tri) {
    int m = 0;
    for (int k = 1; k < sc.n_meshes; ++k) m = (tri >= sc.mesh[k].tri_begin) ? k : m;
    return m;
}
__device__ __forceinline__ int mesh_obj_of_tri(const Scene &sc, int tri) {   // ... and its position in Scene::objects
    const int mw = mesh_of_tri(sc, tri);
    int mobj = sc.mesh[0].obj;
    for (int k = 1; k < sc.n_meshes; ++k) mobj = (k == mw) ? sc.mesh[k].obj : mobj;
    return mobj;
}
// Scene::intersect_all's running minimum with the strict '<' of cpu:554 is the lexicographic minimum over (t, position in Scene::objects).  Given the spheres' own winner
// (t_s, object id; -1 = none) and the meshes' (t_m, object id of the winning triangle's mesh): does the mesh win?  A tie goes to whichever comes first.
__device__ __forceinline__ bool mesh_beats_sphere(float t_s, int obj_s, float t_m, int obj_m) { return (obj_s > obj_m) ? !(t_s < t_m) : (t_m < t_s); }

template <bool STATS>
__device__ __forceinline__ bool intersect_all(const Scene &sc, f3 O, f3 u, float tri_tmin, f3 &P, f3 &N, int &objectId, Work &wk) {
    float t_min = 1e9f;
    int id_min = -1;
    int sph_min = -1;
    f3 Nmesh = mk(0, 0, 0);
    // the spheres in insertion order (strict '<': the earliest of equal t), then the mesh(es): the lexicographic minimum over (t, position) either way round
    for (int k = 0; k < sc.n_spheres; ++k) {
        const Sphere &s = sc.sph[k];
        float t;
        if (!sphere_test(s, O, u, t)) continue;
        if (t < t_min) { t_min = t; id_min = s.obj; sph_min = k; }
    }
    if (sc.mesh_slot >= 0 && sc.n_nodes > 0) {
        float t; f3 Nr; int tri;
        if (mesh_intersect<STATS>(sc, O, u, tri_tmin, t, Nr, tri, wk)) {
            const int mobj = mesh_obj_of_tri(sc, tri);
            if (mesh_beats_sphere(t_min, id_min, t, mobj)) { t_min = t; id_min = mobj; sph_min = -1; Nmesh = Nr; }
        }
    }
    P = O + t_min * u;   // cpu:560 (also on a miss)
    objectId = id_min;
    if (id_min < 0) { N = mk(0, 0, 0); return false; }
    if (sph_min >= 0) {
        const Sphere &s = sc.sph[sph_min];
        N = normalize(O + t_min * u - mk(s.cx, s.cy, s.cz));   // cpu:524-525
    } else {
        N = normalize(Nmesh);                                  // cpu:308
    }
    return true;
}

struct Material { float ar, ag, ab; int mirror; float n_in, n_out; };
// Geometry's fields (cpu:106-118) of object `obj`, sphere or mesh: Scene::getColor reads them for whichever object was hit (cpu:573-606).  (Callers use a
// part of the record: the loads of the rest are dropped.)
__device__ __forceinline__ Material material_of(const Scene &sc, int obj) {
    Material m;
    const float4 a = sc.obj_a[obj], b = sc.obj_b[obj];
    const float2 n = sc.obj_n[obj];
    m.ar = b.x; m.ag = b.y; m.ab = b.z; m.mirror = __float_as_int(a.w); m.n_in = n.x; m.n_out = n.y;
    return m;
}
__device__ __forceinline__ f3 sphere_centre_of(const Scene &sc, int obj) { const float4 a = sc.obj_a[obj]; return mk(a.x, a.y, a.z); }

// Scene::getColor, cpu:566-648, made iterative: the path is walked front to back recording for each
// diffuse segment the scalar l (cpu:623) and the object id, then folded back to front exactly as the
// recursion returns (color = direct + albedo (.) child, cpu:642-644).  Mirror/refraction segments return
// the child unchanged (cpu:579,594,601) and a miss returns black (cpu:571).
// lstack: per-lane LDS column, lstack[d * kBlockThreads].
template <bool STATS>
__device__ __forceinline__ f3 get_color(const Scene &sc, const Frame &fr, f3 O, f3 u, uint32_t hs, float *lstack, float &rays, Work &wk) {
    const float PI_F = (float)3.14159265358979323846;
    const double PI_D = 3.14159265358979323846;
    const f3 L = mk(sc.Lx, sc.Ly, sc.Lz);
    float refr = 1.f;             // Ray::refraction_index, cpu:100
    uint64_t ids = 0;             // 4 bits of object id per segment
    uint32_t diffuse_mask = 0;
    int nseg = 0;
    for (int d = 0; d < fr.segs; ++d) {
        f3 P, N; int id;
        rays += 1.f;
        if (!intersect_all<STATS>(sc, O, u, fr.tri_tmin, P, N, id, wk)) break;
        nseg = d + 1;
        const Material m = material_of(sc, id);
        if (m.mirror) {                                             // cpu:573-579
            O = P + fr.eps * N;
            u = u - (2 * dot(u, N)) * N;
        } else if (m.n_in != m.n_out) {                             // cpu:580-604
            float ratio;
            const bool out2in = refr == m.n_out;
            if (out2in) ratio = m.n_out / m.n_in;
            else { ratio = m.n_in / m.n_out; N = -N; }
            const float un = dot(u, N);
            if (((out2in && refr > m.n_in) || (!out2in && refr > m.n_out)) && (ratio * ratio) * (1 - un * un) > 1) {
                O = P + fr.eps * N;
                u = u - (2 * un) * N;
            } else {
                O = P - fr.eps * N;
                const f3 Nc = (-rt_sqrtf(1 - (ratio * ratio) * (1 - un * un))) * N;
                const f3 Tc = ratio * (u - un * N);
                u = Nc + Tc;
                refr = out2in ? m.n_in : m.n_out;
            }
        } else {                                                    // cpu:605-645
            const f3 Pa = P + fr.eps * N;
            const f3 toL = L - Pa;
            const f3 sdir = normalize(toL);   // = toL / sqrt(norm2(toL))           // NORMED_VEC, cpu:30,614
            f3 Pp, Np; int ids_;
            rays += 1.f;
            (void)intersect_all<STATS>(sc, Pa, sdir, fr.tri_tmin, Pp, Np, ids_, wk);
            float l = 0.f;
            if (!(norm2(Pp - Pa) <= norm2(L - Pa))) {               // cpu:615
                const f3 wl = normalize(L - P);
                const float dn = dot(N, wl);
                const float mx = (dn < 0.f) ? 0.f : dn;             // std::max(dn, 0.f)
                l = (float)((double)sc.intensity / (4 * PI_D * (double)norm2(L - P)) * (double)mx);   // cpu:623
            }
            lstack[d * kBlockThreads] = l;
            ids |= (uint64_t)(id & 15) << (4 * d);
            diffuse_mask |= 1u << d;
            const float r1 = uniform01(hs, (uint32_t)d, 0);         // cpu:628-629
            const float r2 = uniform01(hs, (uint32_t)d, 1);
            double sn, cs;
            rt_sincos_2pi(2 * PI_D * (double)r1, sn, cs);
            const float s1 = rt_sqrtf(1 - r2);
            const float x = (float)(cs * (double)s1);               // cpu:630
            const float y = (float)(sn * (double)s1);               // cpu:631
            const float zz = rt_sqrtf(r2);                        // cpu:632
            // T1 = normalize((-Ny, Nx, 0)) if Nx != 0 && Ny != 0 else normalize((-Nz, 0, Nx)) (cpu:634-638): two quotients, the third component is +0 / n
            const bool t1a = N.y != 0 && N.x != 0;
            float t1p, t1q, t1z;
            normalize_pq0(t1a ? -N.y : -N.z, N.x, t1p, t1q, t1z);
            const f3 T1 = t1a ? mk(t1p, t1q, t1z) : mk(t1p, t1z, t1q);
            const f3 T2 = cross(N, T1);
            u = x * T1 + y * T2 + zz * N;                           // cpu:641
            O = Pa;
            refr = 1.f;                                             // Ray(P_adjusted, random_direction), cpu:642
        }
    }
    f3 ans = mk(0, 0, 0);
    for (int d = nseg - 1; d >= 0; --d) {
        if (diffuse_mask & (1u << d)) {
            const Material m = material_of(sc, (int)((ids >> (4 * d)) & 15));
            const float l = lstack[d * kBlockThreads];
            const f3 alb = mk(m.ar, m.ag, m.ab);
            const f3 direct = (l * alb) / PI_F;                     // cpu:624
            ans = direct + alb * ans;                               // cpu:642,644
        }
    }
    return ans;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

template <bool STATS>
__global__ __launch_bounds__(kBlockThreads) void render_kernel(const Scene sc, const Frame fr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int px = blockIdx.x * kTileW + wave * 8 + (lane & 7);
    const int lrow = blockIdx.y * kTileH + (lane >> 3);
    const int row = fr.row0 + (lrow / fr.tile_rows) * fr.tile_rows * fr.tile_step + (lrow % fr.tile_rows);
    const bool active = px < fr.W && lrow < fr.n_rows && row < fr.H;
    Work wk;
    float rays = 0.f;
    if (active) {
    float *lstack = smem + tid;

    // cpu:699: +0.5/-0.5 are double literals, narrowed by the Vector constructor
    const f3 uc = mk((float)((double)((float)px - (float)fr.W / 2) + 0.5),
                     (float)((double)((float)fr.H / 2 - (float)row) - 0.5), fr.z);
    const f3 C = mk(sc.camx, sc.camy, sc.camz);
    const uint32_t pixel = (uint32_t)row * (uint32_t)fr.W + (uint32_t)px;
    const uint32_t hp = mix32(pixel ^ mix32(fr.seed));
    f3 total = mk(0, 0, 0);
    for (int s = 0; s < fr.spp; ++s) {
        const uint32_t hs = mix32(hp ^ ((uint32_t)s * 0x9E3779B1U));
        f3 uu = uc;
        if (fr.sigma != 0.f) {   // cpu:705-707; with sigma == 0 the jitter is exactly +-0 and uc is unchanged
            const float r1 = uniform01(hs, 0, 2), r2 = uniform01(hs, 0, 3);
            const float bm = fr.sigma * rt_sqrtf(-2 * logf(r1));
            double sn, cs;
            rt_sincos_2pi(2 * 3.14159265358979323846 * (double)r2, sn, cs);
            uu = uc + mk((float)((double)bm * cs), (float)((double)bm * sn), 0.f);
        }
        const f3 u = normalize(uu);
        const f3 col = get_color<STATS>(sc, fr, C, u, hs, lstack, rays, wk);
        total = total + col;
    }
    const f3 avg = total / (float)fr.spp;                           // cpu:713
    fr.out[out_index(fr, lrow, px)] = make_float4(avg.x, avg.y, avg.z, rays);
    }
    if (STATS) {   // one atomic per wave and counter
        const uint32_t r = wave_sum((uint32_t)rays), b = wave_sum(wk.box), n = wave_sum(wk.nodes), t = wave_sum(wk.tris);
        if (lane == 0) {
            atomicAdd(&fr.work[0], (unsigned long long)r); atomicAdd(&fr.work[1], (unsigned long long)b);
            atomicAdd(&fr.work[2], (unsigned long long)n); atomicAdd(&fr.work[3], (unsigned long long)t);
        }
    }
}

// cpu:714-716: std::min(std::pow(c, 1./2.2), 255.) -> unsigned char.  4 pixels per lane,
// 12 output bytes written as three dwords.
__device__ __forceinline__ uint32_t tone1(float c) {
    double v = pow((double)c, 1. / 2.2);
    if (255. < v) v = 255.;
    if (!(v == v)) return 0u;
    return (uint32_t)(int)v;
}
__global__ __launch_bounds__(256) void tonemap_kernel(const float4 *__restrict__ rgba, int64_t npix, uint8_t *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p0 = q * 4;
    if (p0 >= npix) return;
    uint32_t b[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float4 c = (p0 + k < npix) ? rgba[p0 + k] : make_float4(0, 0, 0, 0);
        b[3 * k + 0] = tone1(c.x); b[3 * k + 1] = tone1(c.y); b[3 * k + 2] = tone1(c.z);
    }
    if (p0 + 4 <= npix) {
        uint32_t *o = reinterpret_cast<uint32_t *>(out + 3 * p0);   // 12*q bytes: 4-byte aligned
        o[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        o[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        o[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
    } else {
        for (int64_t k = 0; p0 + k < npix; ++k)
            for (int ch = 0; ch < 3; ++ch) out[3 * (p0 + k) + ch] = (uint8_t)b[3 * k + ch];
    }
}

// Progressive accumulation, realtime_render.cu:1136-1147: accumbuffer += frame; display = accumbuffer / framenumber
// (cutil_math: a * (1.0f / s)); 8-bit image = (unsigned char)min(powf(c, 1 / 2.2f), 255.).  .w carries the ray counts.
__global__ __launch_bounds__(256) void accumulate_kernel(const float4 *__restrict__ frame, float4 *__restrict__ accum, float4 *__restrict__ display,
                                                         uint8_t *__restrict__ rgb8, int64_t npix, int framenumber) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float4 f = frame[i];
    float4 a = accum[i];
    a.x += f.x; a.y += f.y; a.z += f.z; a.w += f.w;
    accum[i] = a;
    const float inv = 1.0f / (float)framenumber;
    const float4 d = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w);
    display[i] = d;
    const float c[3] = {d.x, d.y, d.z};
    for (int k = 0; k < 3; ++k) {
        double v = (double)powf(c[k], 1 / 2.2f);
        if (!(v < 255.)) v = 255.;
        rgb8[3 * i + k] = (uint8_t)v;
    }
}

}  // namespace rtk
