// rt_wavefront.hip.h -- wavefront render pipeline for gfx950 (default variant).
//
// The per-pixel render of cpu_launcher.cpp:693-718 / KernelLaunch (optimized.cu:670-772) split by
// divergence behaviour instead of by pixel:
//
//   wf_advance<FIRST>  one lane per pixel, uniform: camera ray of sample s (cpu:699-709), its ray/sphere tests
//              (cpu:512-527) and the mesh's root-box test (cpu:279); path record initialised
//   wf_travq   (rt_travq.hip.h; the pipeline's traversal kernel by default) the BVH traversal as a per-wave work stack of
//              (ray slot, sibling pair) entries: BoundingBox::intersect cpu:146-157, moller_trumbore cpu:226-236, traversal
//              cpu:277-311
//   wf_trav    (variants wavefront / wavefront_lds) persistent lanes, one RAY (or a sub-range of one ray's traversal) per lane:
//              stackless BVH walk.  Two micro-ops (BOX, TRI) chosen by wave vote; lanes refill from the wave's own, spatially
//              scrambled, share of the rays; when that runs out busy lanes hand parts of their traversal
//              to idle lanes (see "work splitting" below).
//   wf_advance one lane per pixel, uniform: closes the query (Scene::intersect_all cpu:545-564), then
//              Scene::getColor's branch for it -- material / shadow ray (cpu:573-614) or direct light +
//              cosine bounce (cpu:615-642) -- writes the next ray with its sphere and root-box tests, or
//              folds the finished path (cpu:642-644), accumulates the sample (cpu:711) and, after the last
//              sample, stores the pixel as one float4 (cpu:713)
//
// A shadow ray and the continuation (bounce / mirror / refraction) ray that leave the same hit point do not
// depend on each other, so both are traced by the SAME traversal launch: per sample the sequence is
//     advance<FIRST>, (trav, advance) x (segments + 1)
// with launch j tracing the shadow rays of segment j-1 and the continuation rays of segment j.  The path records (16 bytes)
// live in HBM as float4 indexed by a TILE-ORDER path index (8x8 pixel tiles: a wave's 64 consecutive paths are one
// tile, every record access is a fully coalesced 1 KiB wave transaction); the rays themselves live only in the traversal queue
// (slot order), see WfState.
//
// Work splitting.  The reference never prunes by distance, so which nodes and triangles a ray
// visits does not depend on what it has hit so far, and the nearest hit is  min over visited triangles of
// (t, visit rank)  -- the strict '<' of cpu:301 keeps the earliest of equal t.  In the stackless layout the
// traversal of node range [a,b) that starts at a is self-contained whenever a is a node the walk is certain
// to reach: the node after any subtree (skip(X)) is such a node.  So a lane at node X owning [X,b) can keep
// [X,skip(X)) and give [skip(X),b) to an idle lane; a lane inside a leaf can give away everything after the
// leaf, or half of the leaf's triangles.  Triangles are stored in visit order, so rank == triangle index,
// and sub-results meet in one 64-bit atomic min on (bits(t) << 32 | index).  Results stay bit-identical.
#pragma once
#include "rt_kernels.hip.h"

namespace rtk {

// path record flags (F.x; rt_path.hip.h keeps the same record in LDS)
constexpr int PF_ALIVE = 1, PF_HASX = 2, PF_HASY = 4;       // path being traced; a shadow ray (X) of segment d-1's hit / a continuation ray (Y) of segment d is in flight
constexpr int PF_MESHX = 8, PF_MESHY = 16;                  // that ray passed the mesh's root box: its traversal result M[] is meaningful
constexpr int PF_DEPTH_SHIFT = 5, PF_DEPTH_MASK = 31;       // segment index d of the continuation ray in flight (or of the next one), 0..16
constexpr int PF_RAYS_SHIFT = 10, PF_RAYS_MASK = 63;        // rays traced so far for this sample (<= 2 * 16 + 1)
constexpr int PF_WINS_SHIFT = 16;                           // 10 bits: the Y ray's nearest sphere before / after the mesh slot (object id + 1)
constexpr int PF_XSPHERE = 1 << 26;                         // the X ray is blocked by a sphere already (cpu:615 true whatever the mesh says)
constexpr unsigned long long WF_NOHIT = ~0ull;
// the wavefront pipeline keeps the path's flag word in its continuation (Y) slot's queue record; bits 0..15 as above, then:
constexpr int PQ_WIN_SHIFT = 16;                            // 5 bits: object id + 1 of the Y ray's nearest sphere (0 = none); its t is the record's last word
constexpr int PQ_ANYHIT = 1 << 21;                          // (X slot only) the record's last word is the shadow ray's ANY-HIT bound: see wf_anyhit_bound
constexpr int PQ_XSPHERE = 1 << 22;                         // PF_XSPHERE of this record
constexpr int PQ_REFR_SHIFT = 23;                           // 6 bits: Ray::refraction_index of the Y ray: 0 = 1.0, else (object id + 1) << 1 | (0: that object's n_in, 1: its n_out)
constexpr int PQ_TRAV = 1 << 29;                            // (either slot) the record's ray passed the mesh's root box: the traversal launch whose number (WfState::epoch) equals the
                                                            // record's depth field picks it up.  A shadow ray that misses the root box is not written at all: what its slot still
                                                            // holds is an older launch's record (other depth or another chain's number), or the zero of the layout's memset
constexpr int PQ_NONCE_SHIFT = 30, PQ_NONCE_MASK = 3;        // (either slot) the launch chain's number mod 4, so that wf_advance<FIRST> need not clear the shadow slots of the previous chain
constexpr int PQ_LIVE_MASK = (int)((unsigned)PQ_TRAV | ((unsigned)PF_DEPTH_MASK << PF_DEPTH_SHIFT) | ((unsigned)PQ_NONCE_MASK << PQ_NONCE_SHIFT));
// the record is live for traversal launch `epoch` of the chain numbered `nonce`: one masked compare.  A stale shadow record that passes (written four chains ago at the
// same depth into a slot nobody wrote since) costs one wasted traversal and nothing else: wf_advance reads a shadow ray's result only if ITS OWN flag word says the
// ray went to the mesh (PF_MESHX)
__device__ __forceinline__ bool wq_live(int w0, int epoch, int nonce) {
    return (w0 & PQ_LIVE_MASK) == (int)((unsigned)PQ_TRAV | (unsigned)epoch << PF_DEPTH_SHIFT | (unsigned)nonce << PQ_NONCE_SHIFT);
}

// ANY-HIT bound of a shadow ray.  cpu:615 asks whether |P' - Pa|^2 <= |L - Pa|^2 for P' = Pa + t_min u, the NEAREST hit of the shadow ray (Pa = P_adjusted, u = (L - Pa) /
// |L - Pa|), and uses nothing else of that hit.  Every rounding on the left is monotone in t (fl(t u_k), fl(Pa_k + .), fl(. - Pa_k), the squares, the sums: the magnitude of
// each component grows with t, see the comment at the close of the X query in wf_advance_path), so the comparison holds for the nearest hit iff it holds for SOME accepted
// hit: a traversal that has accepted one triangle whose t certainly passes may stop -- the frame cannot tell.  "Certainly": with nl = fl(sqrt(fl|L - Pa|^2)) (the value
// normalize() divides by), g_k = fl(fl(Pa_k + fl(t u_k)) - Pa_k) obeys |g_k| <= t |u_k| (1 + 3 * 2^-24) + 2^-24 (1 + 2^-23) |Pa_k|, |u| <= 1 + 2^-22, so the left side's
// square root is at most t (1 + 2^-20) + 2^-22 |Pa|_1, while the right side's is at least nl (1 - 2^-24).  Any t <= nl (1 - 2^-15) - 2^-20 |Pa|_1 is therefore inside with
// a margin of more than 2^-16 nl (the bound's own two roundings are 2^-24 each); hits between this bound and the light (a band 3e-5 of the distance wide) simply do not stop
// the traversal, whose complete result then decides as before.  -inf = never (a light within 1e-12 of the surface or so far that the squared distance overflows; with a NaN
// anywhere the bound is a NaN or -inf: every comparison with it is false).
__device__ __forceinline__ float wf_anyhit_bound(f3 Pa, float nl) {
    const float b = fmaf(nl, 1.f - 0x1p-15f, -0x1p-20f * ((fabsf(Pa.x) + fabsf(Pa.y)) + fabsf(Pa.z)));
    return (nl > 1e-12f && nl < 1e30f) ? b : -__builtin_inff();      // (above 1e-12 no square that matters to the budget is denormal; a finite nl is below 1.9e19 anyway: beyond it |L - Pa|^2 is +inf)
}

// A BATCH of frames in one launch chain (rt_render_device_batch): the items of a chain are (frame f, pixel slot) pairs -- the machinery that traces several samples of a pixel
// as parallel items (n_paths = n_px x items per pixel), with a camera, a seed and an output buffer PER FRAME instead of per-sample colours to reduce.  A rank that renders a
// small share of a frame (1/8 of 1920x1080 = 0.26 Mpixel) fills the chip with K frames' worth of paths per launch instead of K chains on K streams.  Only wf_advance reads it.
// The descriptors live in device memory (WfState::batch: one copy per sub-frame, written by batch_store_kernel at the head of that sub-frame's own chain) and are read with
// a wave-uniform index: as kernel arguments a dynamically indexed array is copied to scratch by the compiler (2 KB per lane, measured).
constexpr int kMaxBatch = 16;
struct BatchFrame { float camx, camy, camz, z; uint32_t seed; int pad; float4 *out; };
struct Batch { int n; int pad; BatchFrame f[kMaxBatch]; };
__global__ void batch_store_kernel(const Batch bt, BatchFrame *__restrict__ dst) {
    const int k = threadIdx.x;
    if (k < kMaxBatch) dst[k] = bt.f[k < bt.n ? k : 0];               // (constant trip structure: one lane per descriptor)
}

// Path state of the wavefront pipeline, in HBM.  A ray lives in ONE place: its 32-byte record in the traversal queue (slot order;
// the four rays of a group are one 128-byte line), which the uniform kernel writes when it emits the ray, the traversal kernel
// reads, and the next uniform launch reads back to compute the hit point.  The record's two spare words hold the PATH's state as
// well (flag word and the nearest sphere's t: round 4; rounds 2-3 kept a 16-byte path record beside the queue), so a path that is
// alive costs one 32-byte read and one 32-byte write of its Y slot, the X slot's record when a shadow ray leaves, 8 bytes of
// traversal result per ray and 5 bytes per shaded segment (round 1: ~300 B per path and launch, rounds 2-3: 154).
struct WfState {
    unsigned long long *M;   // [2 n_paths] traversal result by ray (Y rays at [0, n_paths), X rays at [n_paths, 2 n_paths)): bits(t) << 32 | triangle index (visit order); WF_NOHIT if none
    float4 *samp_out;     // frames with more than one sample: [n_paths] (colour of the item's sample, rays traced); path_reduce adds them in sample order
    float *LS;            // l (cpu:623) of every diffuse segment, written when the segment is shaded and zeroed if its shadow ray is blocked: LS[d * n_paths + i]
    unsigned char *SID;   // object id of the surface shaded at segment d, 0xff if it was not diffuse: SID[d * n_paths + i]
    int n_paths;          // items of this launch chain: n_px * (samples traced together); item i = sample (samp0 + i / n_px) of pixel slot i % n_px
    int n_px;             // pixel slots of the sub-frame: tiles_x * tiles_y * 64
    unsigned int n_px_m, tiles_x_m, Q_m;   // floor(2^32 / d) for the three divisors the uniform kernel divides by (wf_div)
    int samp0;            // first sample of the chain
    int tiles_x;
    // traversal scheduling: ray-slot q in [0, slots) maps to ray 4*g + (q & 3), g = ((q>>2) & (S-1)) * Q + ((q>>2) >> log2S)
    int log2S, Q, n_groups;   // n_groups = 2 n_paths / 4 (ray groups);  S * Q >= n_groups
    int slots_per_block;      // multiple of 4: ray slots owned by one workgroup, handed to its waves on demand
    int nonce;                // number of the launch chain mod 4 (from a counter of the context): part of every live record's flag word
    int epoch;                // index of the traversal launch inside its chain (0 after wf_advance<FIRST>): a queue record is live iff its flag word carries PQ_TRAV and this number
    int init_m;               // the traversal kernel merges partial results with atomicMin (wf_trav's work splitting): emitters initialise M
    unsigned long long *dbg;  // optional per-wave debug record (-DRT_DEBUG)
    int anyhit;               // shadow rays carry their any-hit bound (PQ_ANYHIT): the fixed-point instantiations of wf_travq stop a shadow ray at the first accepted triangle that certainly
                              // shades; a shadow ray that a sphere shades already is not traced through the mesh; a shadow ray whose segment has the direct term +0 either way is not traced
    const BatchFrame *batch;  // rt_render_device_batch: n_batch frame descriptors in device memory (item i belongs to frame i / n_px); nullptr / 0 = the launch's own camera, seed, output
    int n_batch;
    // traversal queue: the rays in TRAVERSAL-SLOT order, so that the slots a traversal workgroup owns are contiguous and one
    // round trip brings flag and record
    float4 *QR;               // [2 slots] record of slot q: QR[2q] = (O.xyz, u.x), QR[2q+1] = (u.y, u.z, bits(W0), W1).  Y slot (ray r < n_paths): W0 = the path's
                              // flag word (PF_* | PQ_*; 0 = no path), W1 = t of the Y ray's nearest sphere; X slot: W0 = PQ_TRAV | depth | chain number (| PQ_ANYHIT: W1 = the any-hit bound), written only when the ray needs traversal
};

// n / d for 0 <= n < 2^32 with m = floor(2^32 / d) from the host: the estimate mulhi(n, m) is the quotient or one below it
// (n * m / 2^32 > n / d - 1), so one correction makes it exact -- 5 vector instructions instead of the ~22 of an integer division.
__device__ __forceinline__ int wf_div(int n, int d, unsigned int m) {
    unsigned int q = __umulhi((unsigned int)n, m);
    q += ((unsigned int)n - q * (unsigned int)d >= (unsigned int)d) ? 1u : 0u;
    return (int)q;
}
__host__ __device__ inline unsigned int wf_div_magic(int d) { return d <= 1 ? 0xffffffffu : (unsigned int)(0x100000000ull / (unsigned long long)d); }

__device__ __forceinline__ void wf_decode(const WfState &st, const Frame &fr, int slot, int &px, int &lrow, bool &valid) {
    const int tile = slot >> 6, p = slot & 63;
    const int ty = wf_div(tile, st.tiles_x, st.tiles_x_m);
    px = (tile - ty * st.tiles_x) * 8 + (p & 7);
    lrow = ty * 8 + (p >> 3);
    valid = px < fr.W && lrow < fr.n_rows;
}

// Scene::intersect_all's running minimum over the SPHERES alone, for the two rays that leave one point (origins equal bit for bit: the origin part of every sphere test
// is shared): (t, object id) of the nearest sphere with the strict '<' of cpu:554 (the earliest of equal t).  The meshes join when the traversal is back: a triangle at tm replaces the sphere iff tm < t, or tm == t and the
// triangle's mesh comes before the sphere in Scene::objects -- the lexicographic minimum over (t, position) IS what the reference's loop keeps.
struct SphereNear { float t; int obj; };
__device__ __forceinline__ SphereNear spheres_near1(const Scene &sc, f3 O, f3 u) {   // ... for one ray (wf_path)
    SphereNear h; h.t = 1e9f; h.obj = -1;
    for (int k = 0; k < sc.n_spheres; ++k) {
        float t;
        if (sphere_test(sc.sph[k], O, u, t) && t < h.t) { h.t = t; h.obj = sc.sph[k].obj; }
    }
    return h;
}
__device__ __forceinline__ void spheres_near2(const Scene &sc, f3 O, f3 uy, bool on_y, f3 ux, bool on_x, SphereNear &hy, SphereNear &hx) {
    hy.t = 1e9f; hy.obj = -1;
    hx = hy;
    for (int k = 0; k < sc.n_spheres; ++k) {
        const SphereOrigin so = sphere_origin(sc.sph[k], O);
        float t;
        if (on_y && sphere_dir(sc.sph[k], so, O, uy, t)) { if (t < hy.t) { hy.t = t; hy.obj = sc.sph[k].obj; } }
        if (on_x && sphere_dir(sc.sph[k], so, O, ux, t)) { if (t < hx.t) { hx.t = t; hx.obj = sc.sph[k].obj; } }
    }
}
// inverse of wf_slot_to_path: the traversal slot of ray r
__device__ __forceinline__ int wf_ray_to_slot(const WfState &st, int r) {
    const int g = r >> 2;
    const int a = wf_div(g, st.Q, st.Q_m), col = g - a * st.Q;
    return ((col << st.log2S | a) << 2) | (r & 3);
}

// The mesh's root-box test of an emitted ray (cpu:279; wave-uniform node data from kernel arguments): whether the ray needs traversal.
template <bool STATS>
__device__ __forceinline__ bool wf_root_test(const Scene &sc, const WfState &st, int r, f3 O, f3 u, Work &wk) {
    bool need = false;
    if (sc.mesh_slot >= 0 && sc.n_nodes > 0) {
        if (STATS) wk.box++;
        if (slab_filtered(sc.root_lo, sc.root_hi, O, u, ray_inv(u))) {
            if (STATS) wk.nodes++;
            need = true;
            if (st.init_m) st.M[r] = WF_NOHIT;
        }
    }
    return need;
}

template <bool STATS>
__device__ __forceinline__ void wf_flush_work(const Frame &fr, Work &wk) {
    if (STATS) {
        const uint32_t b = wave_sum(wk.box), n = wave_sum(wk.nodes), t = wave_sum(wk.tris), lb = wave_sum(wk.lit_box), lt = wave_sum(wk.lit_tri);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&fr.work[1], (unsigned long long)b);
            atomicAdd(&fr.work[2], (unsigned long long)n); atomicAdd(&fr.work[3], (unsigned long long)t);
            if (lb) atomicAdd(&fr.work[5], (unsigned long long)lb);
            if (lt) atomicAdd(&fr.work[6], (unsigned long long)lt);
        }
    }
}

// ---- wf_trav: persistent stackless traversal -----------------------------------------------------------
// Hot loop = BOX steps only.  A lane that hits a leaf appends (first, count) to its private LDS leaf list and
// keeps walking; nothing else happens in the loop but one wave vote.  When fewer than kBoxMin lanes can
// still walk, a SERVICE pass runs: leaf lists are expanded into the wave's LDS ring of (owner lane, triangle)
// entries, the ring is consumed 64 entries at a time by full-occupancy TRI steps (ray data from the owners'
// LDS copies, results merged by a 64-bit LDS min on bits(t) << 32 | triangle), finished tasks retire, idle
// lanes refill from the wave's scrambled share of the rays or take a split from a busy lane.
constexpr int kTravBlock = 512;             // nodes from HBM/L2: 512-thread blocks (8 waves share one slot pool), 3 per CU
constexpr int kTravBlockLds = 1024;         // nodes staged in LDS: ONE 1024-thread block per CU shares the copy
#ifndef RT_BOXMIN
#define RT_BOXMIN 36
#endif
constexpr int kBoxMin = RT_BOXMIN;                 // service when fewer lanes than this can take a BOX step

// per-wave LDS carve: ray0[64] ray1[64] (float4) | best[64] (u64) | ring[QCAP] (u32) | leaf[LEAFCAP][64] (u32)
template <int QCAP, int LEAFCAP> struct TravCarve {
    static constexpr int kRay = 0, kBest = 2048, kRing = 2560, kLeaf = kRing + 4 * QCAP, kBytes = kLeaf + 256 * LEAFCAP;
};

__device__ __forceinline__ int wf_slot_to_path(const WfState &st, int q) {
    const int gs = q >> 2;
    const int col = gs >> st.log2S;                  // >= Q for the padding slots of the last waves
    const int g = (gs & ((1 << st.log2S) - 1)) * st.Q + col;
    return (col < st.Q && g < st.n_groups) ? 4 * g + (q & 3) : -1;
}

// single-instruction min/max (the operands are results of arithmetic, never signalling NaNs; IEEE-mode
// v_min/v_max return the other operand for a quiet NaN, which the callers exclude beforehand)
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3abs(float a, float b, float c) { float r; asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Per-ray constants of the fused box filter.  q~ = fma(bound, r, -fl(O*r)) differs from the reference's
// RN((bound - O)/u) by at most 2^-21 |q| + 2^-24 |O*r| (+ denormal slack): r carries 2^-23, fl(O*r) 2^-24 of
// |O*r|, the fma 2^-24, the reference's own subtraction and division 2^-24 each.
struct RayBox { float rx, ry, rz, ox, oy, oz, c0; bool safe; };
__device__ __forceinline__ RayBox ray_box(f3 O, f3 u) {
    RayBox b;
    b.rx = __builtin_amdgcn_rcpf(u.x); b.ry = __builtin_amdgcn_rcpf(u.y); b.rz = __builtin_amdgcn_rcpf(u.z);
    b.ox = O.x * b.rx; b.oy = O.y * b.ry; b.oz = O.z * b.rz;
    const float omax = vmax3abs(b.ox, b.oy, b.oz);
    b.c0 = 2.f * (omax * 0x1p-24f + kAbs);
    const float umin = fminf(fminf(fabsf(u.x), fabsf(u.y)), fabsf(u.z)), umax = fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fabsf(u.z));
    b.safe = umin > kTiny && umax < kBig && omax < kBig;      // false for 0, denormal, inf, NaN anywhere
    return b;
}

template <bool STATS, bool LDSN>
__global__ __launch_bounds__(LDSN ? kTravBlockLds : kTravBlock) void wf_trav(const Scene sc, const Frame fr, const WfState st) {
    constexpr int kQCap = LDSN ? 256 : 512;         // ring entries per wave (power of two)
    constexpr int kLeafCap = LDSN ? 4 : 8;          // leaf-list entries per lane
    using Carve = TravCarve<kQCap, kLeafCap>;
    extern __shared__ __attribute__((aligned(16))) unsigned char trav_smem[];
    unsigned char *const smem = trav_smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wib = tid >> 6;                       // wave in block
    const int wave = (blockIdx.x * blockDim.x + tid) >> 6;
    // LDS: [LDSN: every node, 2 x float4 each] then one carve per wave
    const float4 *nodes = sc.nodes;
    unsigned char *wl = smem + wib * Carve::kBytes;
    if (LDSN) {
        float4 *ln = reinterpret_cast<float4 *>(smem);
        for (int k = tid; k < 2 * sc.n_nodes; k += blockDim.x) ln[k] = sc.nodes[k];
        __syncthreads();
        nodes = ln;
        wl += (size_t)sc.n_nodes * 32;
    }
    // the workgroup's slot cursor: waves take slots with one LDS atomic per refill, which evens out the cost
    // variance between the waves of a workgroup
    int *const blk_cur = reinterpret_cast<int *>(smem + (LDSN ? (size_t)sc.n_nodes * 32 : 0) + (blockDim.x >> 6) * Carve::kBytes);
    if (tid == 0) *blk_cur = 0;
    __syncthreads();
    float4 *const lray0 = reinterpret_cast<float4 *>(wl + Carve::kRay);
    float4 *const lray1 = lray0 + 64;
    unsigned long long *const lbest = reinterpret_cast<unsigned long long *>(wl + Carve::kBest);
    unsigned int *const q = reinterpret_cast<unsigned int *>(wl + Carve::kRing);
    unsigned int *const leaf = reinterpret_cast<unsigned int *>(wl + Carve::kLeaf) + lane;     // leaf[k * 64]
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    // lane state: a task = traversal of node range [node, nend) of ray `ray`
    int ray = -1;                       // path index, -1 = idle
    f3 O = mk(0, 0, 0), u = mk(0, 0, 1);
    RayBox rb = ray_box(O, u);
    int node = 0, nend = 0, sk = 0;
    int nl = 0;                         // entries in the lane's leaf list
    int pend_first = 0, pend_cnt = 0;   // triangle range currently being queued
    unsigned int last_pos = 0;          // ring position after this task's last queued entry
    bool shared = false;                // this ray's traversal was split: merge with a global atomic
    // wave-uniform: the wave's own ray slots and its ring cursors (monotonic positions)
    const int blk_base = blockIdx.x * st.slots_per_block;
    const int blk_n = st.slots_per_block;
    bool drained = false;
    unsigned int qhead = 0, qtail = 0;
    Work wk;
#ifdef RT_DEBUG
    const bool dbg_on = st.dbg != nullptr;
#else
    constexpr bool dbg_on = false;
#endif
    const unsigned long long dbg_t0 = dbg_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned int dbg_steps = 0, dbg_lanes = 0, dbg_splits = 0;
    unsigned long long cy_box = 0, cy_tri = 0, cy_exp = 0, cy_ref = 0, stamp = 0;
#define WF_STAMP(acc) do { if (dbg_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - stamp; stamp = t_; } } while (0)
    if (dbg_on) stamp = __builtin_amdgcn_s_memtime();
    bool boxable = false;
    for (;;) {
        int nB = __popcll(__ballot(boxable));
        if (nB < kBoxMin) {
            // =============================== SERVICE ===============================
            for (;;) {
                WF_STAMP(cy_box);
                // ---- (1) expand leaf lists into the ring, one entry per lane and round ----
                for (;;) {
                    if (pend_cnt == 0 && nl > 0) {
                        nl--;
                        const unsigned int e = leaf[nl * 64];
                        pend_first = (int)(e & 0x3ffffffu); pend_cnt = (int)(e >> 26);
                    }
                    const unsigned long long pm = __ballot(pend_cnt > 0);
                    if (!pm || (unsigned int)kQCap - (qtail - qhead) < 64u) break;
                    if (pend_cnt > 0) {
                        const unsigned int pos = qtail + (unsigned int)__popcll(pm & lane_lt);
                        q[pos & (kQCap - 1)] = (unsigned int)lane << 26 | (unsigned int)pend_first;
                        pend_first++; pend_cnt--;
                        last_pos = pos + 1u;
                    }
                    qtail += (unsigned int)__popcll(pm);
                }
                WF_STAMP(cy_exp);
                // ---- (2) TRI steps: 64 queued (owner, triangle) pairs, one per lane ----
                const bool walkable = ray >= 0 && node < nend && nl < kLeafCap && pend_cnt == 0;
                const int nW = __popcll(__ballot(walkable));
                for (;;) {
                    const unsigned int count = qtail - qhead;
                    if (count < 64u && !(count > 0u && nW < kBoxMin)) break;
                    const unsigned int n = count < 64u ? count : 64u;
                    dbg_steps++; dbg_lanes += n;
                    if ((unsigned int)lane < n) {
                        const unsigned int e = q[(qhead + (unsigned int)lane) & (kQCap - 1)];
                        const int o = (int)(e >> 26);
                        const int i = (int)(e & 0x3ffffffu);
                        const float4 *tp = sc.tri + 3 * i;
                        const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                        const float4 r0 = lray0[o], r1 = lray1[o];
                        const f3 Oo = mk(r0.x, r0.y, r0.z), uo = mk(r0.w, r1.x, r1.y);
                        const f3 A = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
                        const f3 N = mk(q2.y, q2.z, q2.w);
                        const float det = dot(uo, N);               // moller_trumbore, cpu:226-236
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
                        const bool reject = trust && (b < -eb || b > 1.f + eb || g < -eg || g > 1.f + eg || sum > 1.f + es);
                        bool ok = trust && b >= eb && b <= 1.f - eb && g >= eg && g <= 1.f - eg && sum <= 1.f - es;
                        if (!reject && !ok && det != 0) {           // undecided: the literal tests (rare)
                            const float beta = bn / det;
                            const float gamma = gn / det;
                            ok = (0 <= beta && beta <= 1) && (0 <= gamma && gamma <= 1) && (beta + gamma <= 1);
                        }
                        if (ok) {
                            const float t = dot(AO, N) / det;
                            if (t > 0 && t > fr.tri_tmin && t < 1e9f)    // cpu:235,301; the min is the strict '<' scan
                                atomicMin(&lbest[o], (unsigned long long)__float_as_uint(t) << 32 | (unsigned int)i);
                        }
                    }
                    qhead += n;
                }
                WF_STAMP(cy_tri);
                // ---- (3) retire tasks whose nodes are walked and whose queued triangles have all been tested ----
                if (ray >= 0 && node >= nend && nl == 0 && pend_cnt == 0 && (int)(qhead - last_pos) >= 0) {
                    const unsigned long long key = lbest[lane];
                    if (key != WF_NOHIT) {
                        if (shared) atomicMin(&st.M[ray], key);
                        else st.M[ray] = key;
                    }
                    ray = -1;
                }
                // ---- (4) refill idle lanes from the wave's slots, or split ----
                const unsigned long long idle = __ballot(ray < 0);
                const int n_idle = __popcll(idle);
                if (n_idle && !drained) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(blk_cur, n_idle);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + n_idle >= blk_n) drained = true;
                    if (ray < 0) {
                        const int qo = base + __popcll(idle & lane_lt);
                        const float4 rq = qo < blk_n ? st.QR[2 * ((size_t)blk_base + qo) + 1] : make_float4(0, 0, 0, 0);
                        const int rf = wq_live(__float_as_int(rq.z), st.epoch, st.nonce) ? wf_slot_to_path(st, blk_base + qo) + 1 : 0;   // ray + 1 if the slot's ray needs traversal
                        if (rf != 0) {
                            const int path = rf - 1;
                            {
                                const float4 r0 = st.QR[2 * ((size_t)blk_base + qo)];
                                const float4 r1 = make_float4(rq.x, rq.y, 0.f, 0.f);
                                O = mk(r0.x, r0.y, r0.z); u = mk(r0.w, r1.x, r1.y);
                                rb = ray_box(O, u);
                                lray0[lane] = r0; lray1[lane] = r1; lbest[lane] = WF_NOHIT;
                                // the root box was tested when the ray was emitted; start below it
                                const int rw = __float_as_int(sc.root_hi.w);
                                node = rw < 0 ? 1 : sc.n_nodes; nend = sc.n_nodes; sk = 0; nl = 0;
                                pend_first = rw < 0 ? 0 : __float_as_int(sc.root_lo.w);
                                pend_cnt = rw < 0 ? 0 : rw - pend_first;
                                if (STATS && rw >= 0) wk.tris += (uint32_t)pend_cnt;
                                last_pos = qhead; shared = false;
                                ray = path;
                            }
                        }
                    }
                } else if (n_idle >= 4 && n_idle < 64) {
                    // a busy lane below a hit internal node X keeps [node, skip(X)) and gives [skip(X), nend) away
                    const bool d_skip = ray >= 0 && sk > node && sk < nend;
                    const unsigned long long donors = __ballot(d_skip);
                    if (donors) {
                        const int n_pairs = min(n_idle, __popcll(donors));
                        const int my_idle_rank = __popcll(idle & lane_lt);
                        const int my_donor_rank = __popcll(donors & lane_lt);
                        int partner = -1;
                        if (ray < 0 && my_idle_rank < n_pairs) {   // lane id of the my_idle_rank-th donor
                            unsigned long long m = donors;
                            for (int k = 0; k < my_idle_rank; ++k) m &= m - 1;
                            partner = __ffsll((long long)m) - 1;
                        }
                        const bool is_donor = ((donors >> lane) & 1ull) && my_donor_rank < n_pairs;
                        int g_node = 0, g_nend = 0;
                        if (is_donor) { g_node = sk; g_nend = nend; nend = sk; shared = true; }
                        const int src = partner >= 0 ? partner : lane;
                        const int r_ray = __shfl(ray, src, 64);
                        const float r_ox = __shfl(O.x, src, 64), r_oy = __shfl(O.y, src, 64), r_oz = __shfl(O.z, src, 64);
                        const float r_ux = __shfl(u.x, src, 64), r_uy = __shfl(u.y, src, 64), r_uz = __shfl(u.z, src, 64);
                        const int r_node = __shfl(g_node, src, 64), r_nend = __shfl(g_nend, src, 64);
                        if (partner >= 0) {
                            ray = r_ray; O = mk(r_ox, r_oy, r_oz); u = mk(r_ux, r_uy, r_uz); rb = ray_box(O, u);
                            lray0[lane] = make_float4(O.x, O.y, O.z, u.x); lray1[lane] = make_float4(u.y, u.z, 0, 0);
                            lbest[lane] = WF_NOHIT;
                            node = r_node; nend = r_nend; sk = 0; nl = 0; pend_cnt = 0;
                            last_pos = qhead; shared = true;
                        }
                        dbg_splits += n_pairs;
                    }
                }
                WF_STAMP(cy_ref);
                // ---- done servicing? ----
                boxable = ray >= 0 && node < nend && nl < kLeafCap && pend_cnt == 0;
                nB = __popcll(__ballot(boxable));
                if (nB >= kBoxMin) break;
                const bool more = (qtail != qhead) || __ballot(pend_cnt > 0 || nl > 0) != 0ull || (!drained && __ballot(ray < 0) != 0ull);
                if (more) continue;                                  // queue to drain / lists to expand / rays to fetch
                if (nB > 0) break;                                   // run with the lanes we have
                if (__ballot(ray >= 0) == 0ull) goto finished;       // nothing left anywhere
                // busy lanes that are neither walkable nor retired can only be waiting for the ring, which is empty,
                // so the next pass retires them
            }
        }
        // =============================== BOX step ===============================
        dbg_steps++; dbg_lanes += nB;
        if (boxable) {
            const float4 *np = nodes + 2 * node;
            const float4 lo = np[0], hi = np[1];
            const int hiw = __float_as_int(hi.w);
            const int low = __float_as_int(lo.w);
            if (STATS) wk.box++;
            // BoundingBox::intersect (cpu:146-157) through the fused filter
            const float ax = fmaf(lo.x, rb.rx, -rb.ox), bx = fmaf(hi.x, rb.rx, -rb.ox);
            const float ay = fmaf(lo.y, rb.ry, -rb.oy), by = fmaf(hi.y, rb.ry, -rb.oy);
            const float az = fmaf(lo.z, rb.rz, -rb.oz), bz = fmaf(hi.z, rb.rz, -rb.oz);
            const float tn = vmax3(vmin(ax, bx), vmin(ay, by), vmin(az, bz));
            const float tf = vmin3(vmax(ax, bx), vmax(ay, by), vmax(az, bz));
            const float M = vmax(vmax3abs(ax, bx, ay), vmax3abs(by, az, bz));
            const float d = tf - tn;
            const float band = fmaf(M, 2.f * kRel, rb.c0);
            bool hit = d > band;
            const bool decided = rb.safe && M < kBig && (hit || d < -band);
            // literal arithmetic for undecided lanes: behind a wave-uniform branch so that the six IEEE divisions are
            // skipped, not if-converted, when (as almost always) every lane decided
            if (__builtin_expect(__ballot(!decided) != 0ull, 0)) {
                if (!decided) hit = slab(lo, hi, O, u);
            }
            int next = node + 1;
            if (hit) {
                if (STATS) wk.nodes++;
                if (hiw >= 0) {                                // leaf: triangles [low, hiw)
                    const int cnt = hiw - low;
                    if (STATS) wk.tris += (uint32_t)cnt;
                    if (cnt <= 63) { leaf[nl * 64] = (unsigned int)cnt << 26 | (unsigned int)low; nl++; }
                    else { pend_first = low; pend_cnt = cnt; }
                } else {
                    sk = low;                                  // everything from skip(node) on can be given away
                }
            } else if (hiw < 0) {
                next = low;
            }
            node = next;
            boxable = next < nend && nl < kLeafCap && pend_cnt == 0;
        }
    }
finished:
    if (dbg_on && lane == 0) {
        st.dbg[6 * wave + 0] = dbg_t0; st.dbg[6 * wave + 1] = __builtin_amdgcn_s_memrealtime();
        st.dbg[6 * wave + 2] = dbg_steps; st.dbg[6 * wave + 3] = dbg_lanes; st.dbg[6 * wave + 4] = dbg_splits; st.dbg[6 * wave + 5] = 0;
        unsigned long long *d2 = st.dbg + 6 * 65536 + 4 * wave;
        d2[0] = cy_box; d2[1] = cy_exp; d2[2] = cy_tri; d2[3] = cy_ref;
    }
#undef WF_STAMP
    wf_flush_work<STATS>(fr, wk);
}

// ---- wf_advance: close the queries, shade, emit the next rays -----------------------------------------------
// FIRST: the launch that opens the chain's samples -- camera rays (cpu:699-709) instead of closing queries.
// The samples of a pixel are independent paths (the reference's loop cpu:701-712 carries nothing but the sum): a chain traces
// several of them at once as items, each writes its colour, and path_reduce adds the colours in sample order.
// code-object markers (labels, not instructions): tools/static_counts.py cuts the production instantiation into regions at them -- what a path pays for, region by region
#ifdef RT_NO_MARKS
#define ADV_MARK(name) do { } while (0)
#else
#define ADV_MARK(name) asm volatile("rt_mark_adv_" name "_%=:" ::)
#endif
template <bool STATS, bool FIRST>
__device__ __forceinline__ void wf_advance_path(const Scene &sc, const Frame &fr, const WfState &st, const int i, Work &wk) {
    const float4 kDead = make_float4(0, 0, 0, 0);                     // second half of a queue record without a ray (and, in a Y slot, without a path)
    const int rx = st.n_paths + i;                                    // ray index of this path's shadow ray
    const int qy = wf_ray_to_slot(st, i), qx = wf_ray_to_slot(st, rx);
    const float4 y1 = FIRST ? kDead : st.QR[2 * (size_t)qy + 1];      // (u.y, u.z, flag word, t of the nearest sphere) of the continuation ray in flight
    const int F = __float_as_int(y1.z);
    if (!FIRST && !(F & PF_ALIVE)) return;                            // finished (or padding): its queue flags are already 0
    const float PI_F = (float)3.14159265358979323846;
    const double PI_D = 3.14159265358979323846;
    const f3 L = mk(sc.Lx, sc.Ly, sc.Lz);
    int refr_code = FIRST ? 0 : (F >> PQ_REFR_SHIFT) & 63;            // Ray::refraction_index = 1 (cpu:100)
    int d = 0, nrays = 0;
    bool emitY = false, emitX = false, finished = false;
    float x_bound = -__builtin_inff();                                // any-hit bound of the shadow ray (wf_anyhit_bound)
    bool x_moot = false;                                              // the shadow ray's answer cannot reach the pixel (direct term +0 either way)
    f3 Oy = mk(0, 0, 0), uy = mk(0, 0, 1), Ox = mk(0, 0, 0), ux = mk(0, 0, 1);
    int px, lrow; bool valid;
    int s_rel = 0;
    if (st.n_paths != st.n_px) s_rel = wf_div(i, st.n_px, st.n_px_m);
    // a batch: item i belongs to frame s_rel (a wave's 64 items share it: n_px is a multiple of 64), whose camera, seed and output replace the launch's -- from scalar registers
    float camx = sc.camx, camy = sc.camy, camz = sc.camz, fr_z = fr.z;
    uint32_t fr_seed = fr.seed;
    float4 *fr_out = fr.out;
    int samp = st.samp0 + s_rel;
    bool in_batch = true;
    if (st.n_batch > 0) {
        const int f = __builtin_amdgcn_readfirstlane(s_rel);
        in_batch = f < st.n_batch;
        const BatchFrame b = st.batch[f & (kMaxBatch - 1)];             // wave-uniform address: scalar loads
        camx = b.camx; camy = b.camy; camz = b.camz;
        fr_z = b.z; fr_seed = b.seed; fr_out = b.out;
        samp = 0;
    }
    wf_decode(st, fr, i - s_rel * st.n_px, px, lrow, valid);
    valid = valid && samp < fr.spp && in_batch;                       // the last chain of a frame may be short of samples
    const int row = fr.row0 + (lrow / fr.tile_rows) * fr.tile_rows * fr.tile_step + (lrow % fr.tile_rows);

    if (FIRST) {
        if (!valid) { st.QR[2 * (size_t)qy + 1] = kDead; return; }       // (the X slot of a pixel outside the frame is never written: zero from the layout's memset)
        if (fr.segs <= 0) {
            finished = true;                                          // optimized.cu convention with num_bounce 0: black
        } else {
            // cpu:699: +0.5/-0.5 are double literals, narrowed by the Vector constructor
            const f3 uc = mk((float)((double)((float)px - (float)fr.W / 2) + 0.5),
                             (float)((double)((float)fr.H / 2 - (float)row) - 0.5), fr_z);
            f3 ucm = uc;
            if (fr.cam_mode == 1) {   // realtime:1115: cam.C + cam.bz * z + cam.bx * X + cam.by * Y (the position is part of the direction there)
                const f3 Cc = mk(camx, camy, camz), Bx = mk(fr.bx[0], fr.bx[1], fr.bx[2]), By = mk(fr.by[0], fr.by[1], fr.by[2]), Bz = mk(fr.bz[0], fr.bz[1], fr.bz[2]);
                const f3 a = Cc + mk(Bz.x * fr_z, Bz.y * fr_z, Bz.z * fr_z);
                const f3 b = a + mk(Bx.x * uc.x, Bx.y * uc.x, Bx.z * uc.x);
                ucm = b + mk(By.x * uc.y, By.y * uc.y, By.z * uc.y);
            }
            f3 uu = ucm;
            if (fr.sigma != 0.f) {   // cpu:705-707; with sigma == 0 the jitter is exactly +-0
                const uint32_t hp = mix32(((uint32_t)row * (uint32_t)fr.W + (uint32_t)px) ^ mix32(fr_seed));
                const uint32_t hs = mix32(hp ^ ((uint32_t)samp * 0x9E3779B1U));
                const float r1 = uniform01(hs, 0, 2), r2 = uniform01(hs, 0, 3);
                const float bm = fr.sigma * rt_sqrtf(-2 * logf(r1));
                double sn, cs;
                rt_sincos_2pi(2 * 3.14159265358979323846 * (double)r2, sn, cs);
                uu = ucm + mk((float)((double)bm * cs), (float)((double)bm * sn), 0.f);
            }
            Oy = mk(camx, camy, camz);
            uy = normalize(uu);
            emitY = true;                                             // continuation ray of segment 0
            nrays = 1;
        }
    } else {
        ADV_MARK("closex_begin");
        d = (F >> PF_DEPTH_SHIFT) & PF_DEPTH_MASK;                    // segment of the continuation ray in flight
        nrays = (F >> PF_RAYS_SHIFT) & PF_RAYS_MASK;
        // ---- (1) the shadow ray of segment d-1's hit came back: direct light or not (cpu:615) ----
        // cpu:615 compares |P' - P_adj|^2, P' = P_adj + t_min u, with |L - P_adj|^2; it is monotone in t_min (every rounding involved is),
        // so it holds iff it holds for the nearest sphere (decided when the ray was emitted: PF_XSPHERE) or for the nearest triangle
        if (F & PF_HASX) {
            bool shadowed = (F & PQ_XSPHERE) != 0;
            if (!shadowed && (F & PF_MESHX)) {
                const unsigned long long m = st.M[rx];
                if (m != WF_NOHIT) {
                    const float4 x0 = st.QR[2 * (size_t)qx], x1 = st.QR[2 * (size_t)qx + 1];
                    const f3 Oxr = mk(x0.x, x0.y, x0.z), uxr = mk(x0.w, x1.x, x1.y);
                    const f3 Pp = Oxr + __uint_as_float((unsigned int)(m >> 32)) * uxr;   // cpu:560 (Ox is P_adjusted)
                    shadowed = norm2(Pp - Oxr) <= norm2(L - Oxr);
                }
            }
            if (shadowed) st.LS[(size_t)(d - 1) * st.n_paths + i] = 0.f;               // the l stored when the segment was shaded does not count
        }
        ADV_MARK("closex_end");
        // ---- (2) the continuation ray of segment d came back: Scene::getColor's branch for its hit (cpu:570-614) ----
        if (F & PF_HASY) {
            ADV_MARK("closey_begin");
            const float4 r0 = st.QR[2 * (size_t)qy];
            f3 O = mk(r0.x, r0.y, r0.z), u = mk(r0.w, y1.x, y1.y);
            int sid = 0xff;                                           // object id if the hit is diffuse
            // Scene::intersect_all's running minimum (strict '<' in object order, cpu:554) = the lexicographic minimum over (t, position in Scene::objects):
            // the spheres' own winner was decided at emission (spheres_near2), the meshes' by the traversal; between the two a tie goes to the earlier object
            float t_min = y1.w;
            int win = ((F >> PQ_WIN_SHIFT) & 31) - 1, tri_win = -1;
            if (F & PF_MESHY) {
                const unsigned long long m = st.M[i];
                if (m != WF_NOHIT) {
                    const float tm = __uint_as_float((unsigned int)(m >> 32));
                    const int mobj = mesh_obj_of_tri(sc, (int)(unsigned int)m);   // the meshes' own winner: minimum over (t, object position, scan rank) by the order the triangles are stored in
                    if (mesh_beats_sphere(t_min, win, tm, mobj)) { t_min = tm; win = mobj; tri_win = (int)(unsigned int)m; }   // a tie goes to whichever comes first in Scene::objects (no sphere: win = -1, 1e9 > tm)
                }
            }
            if (win >= 0) {                                           // a miss is black (cpu:571): nothing to emit
                const f3 P = O + t_min * u;                           // cpu:560
                f3 N;
                if (tri_win >= 0 && sc.nrm != nullptr) {              // get_smooth_normal, realtime_render.cu:221-245
                    const float4 q0 = sc.tri[3 * tri_win], q1 = sc.tri[3 * tri_win + 1], q2 = sc.tri[3 * tri_win + 2];
                    const f3 A = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x), Nt = mk(q2.y, q2.z, q2.w);
                    const float beta = dot(e2, cross(A - O, u)) / dot(u, Nt);
                    const float gamma = -dot(e1, cross(A - O, u)) / dot(u, Nt);
                    const float alpha = 1 - beta - gamma;
                    const float4 na = sc.nrm[3 * tri_win], nb = sc.nrm[3 * tri_win + 1], nc = sc.nrm[3 * tri_win + 2];
                    N = normalize((alpha * mk(na.x, na.y, na.z) + beta * mk(nb.x, nb.y, nb.z)) + gamma * mk(nc.x, nc.y, nc.z));
                } else if (tri_win >= 0) {
                    const float4 q2 = sc.tri[3 * tri_win + 2];
                    N = normalize(mk(q2.y, q2.z, q2.w));              // cpu:308
                } else {
                    N = normalize(P - sphere_centre_of(sc, win));     // cpu:524-525
                }
                const Material m = material_of(sc, win);
                ADV_MARK("closey_end");
                bool cont = false;                                    // a continuation ray of segment d+1 was built in (O,u)
                if (m.mirror) {                                       // cpu:573-579
                    O = P + fr.eps * N;
                    u = u - (2 * dot(u, N)) * N;
                    cont = true;
                } else if (m.n_in != m.n_out) {                       // cpu:580-604
                    float ratio;
                    float refr = 1.f;                                 // the ray's index: 1, or the n_in / n_out of the surface it last crossed
                    if (refr_code != 0) { const Material mr = material_of(sc, (refr_code >> 1) - 1); refr = (refr_code & 1) ? mr.n_out : mr.n_in; }
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
                        refr_code = (win + 1) << 1 | (out2in ? 0 : 1);    // refr = out2in ? m.n_in : m.n_out
                    }
                    cont = true;
                } else {                                              // cpu:605-642: diffuse
                    ADV_MARK("diffuse_begin");
                    const f3 Pa = P + fr.eps * N;
                    const f3 toL = L - Pa;
                    float nl;
                    Ox = Pa; ux = normalize(toL, nl);   // = toL / nl, nl = sqrt(norm2(toL))     // NORMED_VEC, cpu:614: the shadow ray of segment d
                    x_bound = wf_anyhit_bound(Pa, nl);
                    nrays += 1;
                    // the segment's direct term if the light turns out to be visible (cpu:620-623); kept until the shadow ray is back
                    const f3 wl = normalize(L - P);
                    const float dn = dot(N, wl);
                    const float mx = (dn < 0.f) ? 0.f : dn;
                    const float lvis = (float)((double)sc.intensity / (4 * PI_D * (double)norm2(L - P)) * (double)mx);
                    st.LS[(size_t)d * st.n_paths + i] = lvis;
                    // A surface that faces away from the light (mx = 0) has the direct term +0 whether the light is visible or not: shaded, cpu:616 stores the literal 0; lit,
                    // cpu:623 computes l = +0 -- the same 32 bits (a NaN or a -0 from a degenerate light is not +0 and keeps its ray).  The reference still calls intersect_all
                    // for it (the ray is counted), but nothing of its answer can reach the pixel: with any-hit on the ray is neither set up nor traced.  These are the shadow
                    // rays that start on the mesh's far side and run through its whole body: 4.7 % of a headline frame's traversal work.
                    // (The path still spends a launch on the ray -- PF_HASX without PF_MESHX, closed as a ray that missed the mesh -- so that it finishes in the launch it always
                    // finished in: folding a launch early costs the uniform kernel more than the traversal saves.)
                    emitX = true;
                    x_moot = st.anyhit && __float_as_uint(lvis) == 0u;
                    sid = win;
                    ADV_MARK("diffuse_end");
                    if (d + 1 < fr.segs) {                            // the bounce ray (cpu:627-642): needs r1, r2 and N only
                        ADV_MARK("bounce_begin");
                        const uint32_t hp = mix32(((uint32_t)row * (uint32_t)fr.W + (uint32_t)px) ^ mix32(fr_seed));
                        const uint32_t hs = mix32(hp ^ ((uint32_t)samp * 0x9E3779B1U));
                        const float r1u = uniform01(hs, (uint32_t)d, 0);
                        const float r2u = uniform01(hs, (uint32_t)d, 1);
                        double sn, cs;
                        rt_sincos_2pi(2 * PI_D * (double)r1u, sn, cs);
                        const float s1f = rt_sqrtf(1 - r2u);
                        const float x = (float)(cs * (double)s1f);
                        const float y = (float)(sn * (double)s1f);
                        const float zz = rt_sqrtf(r2u);
                        // T1 = normalize((-Ny, Nx, 0)) if Nx != 0 && Ny != 0 else normalize((-Nz, 0, Nx)) (cpu:634-638): two quotients, the third component is +0 / n
                        const bool t1a = N.y != 0 && N.x != 0;
                        float t1p, t1q, t1z;
                        normalize_pq0(t1a ? -N.y : -N.z, N.x, t1p, t1q, t1z);
                        const f3 T1 = t1a ? mk(t1p, t1q, t1z) : mk(t1p, t1z, t1q);
                        const f3 T2 = cross(N, T1);
                        u = x * T1 + y * T2 + zz * N;
                        O = Pa;
                        refr_code = 0;                                // Ray(P_adjusted, random_direction): index 1
                        cont = true;
                        ADV_MARK("bounce_end");
                    }
                }
                if (cont && d + 1 < fr.segs) {
                    emitY = true; Oy = O; uy = u;                     // continuation ray of segment d+1
                    nrays += 1;
                }
            }
            st.SID[(size_t)d * st.n_paths + i] = (unsigned char)sid;
            d = d + 1;
        }
        finished = !(emitX || emitY);
    }

    if (finished) {   // nothing in flight: fold the path back to front (cpu:642-644), accumulate the sample (cpu:711)
        ADV_MARK("fold_begin");
        f3 ans = mk(0, 0, 0);
        const int nseg = d < fr.segs ? d : fr.segs;
        for (int k = nseg - 1; k >= 0; --k) {
            const int sid = st.SID[(size_t)k * st.n_paths + i];
            if (sid != 0xff) {
                const Material m = material_of(sc, sid);
                const float l = st.LS[(size_t)k * st.n_paths + i];
                const f3 alb = mk(m.ar, m.ag, m.ab);
                ans = (l * alb) / PI_F + alb * ans;
            }
        }
        if (st.samp_out != nullptr) {                                 // more than one sample per pixel: path_reduce sums in sample order
            st.samp_out[i] = make_float4(ans.x, ans.y, ans.z, (float)nrays);
        } else {                                                      // one sample: T = 0 + ans, out = T / n (cpu:711-713 + the framebuffer store)
            float4 t = make_float4(0, 0, 0, 0);
            if (fr.cam_mode == 1) { t.x += ans.x * fr.inv_n; t.y += ans.y * fr.inv_n; t.z += ans.z * fr.inv_n; }   // realtime:1131
            else { t.x += ans.x; t.y += ans.y; t.z += ans.z; }
            t.w += (float)nrays;
            const float n = fr.cam_mode == 1 ? 1.f : (float)fr.spp;
            fr_out[out_index(fr, lrow, px)] = make_float4(t.x / n, t.y / n, t.z / n, t.w);
        }
        st.QR[2 * (size_t)qy + 1] = kDead;                            // the path is over; its X slot keeps a record of an older launch, which no later one takes for its own
        ADV_MARK("fold_end");
        return;
    }

    // ---- (3) emission: sphere tests (cpu:512-527), root-box test (cpu:279), queue records ----
    int flags = PF_ALIVE | (d << PF_DEPTH_SHIFT) | (nrays << PF_RAYS_SHIFT) | (refr_code << PQ_REFR_SHIFT) | (int)((unsigned)st.nonce << PQ_NONCE_SHIFT);
    SphereNear h, hx;
    ADV_MARK("spheres_begin");
    spheres_near2(sc, emitX ? Ox : Oy, uy, emitY, ux, emitX && !x_moot, h, hx);   // a shadow ray and a bounce ray leave the same point (Oy == Ox == P_adjusted)
    ADV_MARK("spheres_end");
    float t_sph = 0.f;
    if (emitX) {
        const float tS = hx.t;                                        // only the value of the shadow ray's nearest hit matters
        const f3 Pp = Ox + tS * ux;                                   // cpu:560
        flags |= PF_HASX;
        if (!x_moot && norm2(Pp - Ox) <= norm2(L - Ox)) flags |= PQ_XSPHERE;      // cpu:615 holds for the sphere already: whatever the mesh says, the segment is shaded (the comparison is monotone in t)
        // ... so with any-hit on that ray is not traced through the mesh at all (intersect_all does: a run that counts the reference's work has any-hit off)
        if (!x_moot && (!(flags & PQ_XSPHERE) || !st.anyhit) && wf_root_test<STATS>(sc, st, rx, Ox, ux, wk)) {   // only then does anybody read the record: the traversal, and this kernel if the mesh is hit
            flags |= PF_MESHX;
            st.QR[2 * (size_t)qx] = make_float4(Ox.x, Ox.y, Ox.z, ux.x);
            st.QR[2 * (size_t)qx + 1] = make_float4(ux.y, ux.z, __int_as_float((int)((unsigned)(PQ_TRAV | (st.anyhit ? PQ_ANYHIT : 0) | (d << PF_DEPTH_SHIFT)) | (unsigned)st.nonce << PQ_NONCE_SHIFT)), x_bound);
        }
    }
    if (emitY) {
        t_sph = h.t;
        flags |= PF_HASY | (((h.obj + 1) & 31) << PQ_WIN_SHIFT);
        if (wf_root_test<STATS>(sc, st, i, Oy, uy, wk)) flags |= PF_MESHY | PQ_TRAV;
    }
    st.QR[2 * (size_t)qy] = make_float4(Oy.x, Oy.y, Oy.z, uy.x);    // whole sectors also when no continuation ray leaves (then nobody reads this half)
    st.QR[2 * (size_t)qy + 1] = make_float4(emitY ? uy.y : 0.f, emitY ? uy.z : 0.f, __int_as_float(flags), t_sph);   // the path's state travels with its Y slot
}

template <bool STATS, bool FIRST>
__global__ __launch_bounds__(256, 8) void wf_advance(const Scene sc, const Frame fr, const WfState st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Work wk;
    if (i < st.n_paths) wf_advance_path<STATS, FIRST>(sc, fr, st, i, wk);
    wf_flush_work<STATS>(fr, wk);                                     // every lane of the wave arrives here (wave-level sums)
}

}  // namespace rtk
