// rt_path.hip.h -- wf_path: the whole per-pixel render in ONE persistent launch (variant RT_VARIANT_PATH, the default).
//
// Replaces KernelLaunch (optimized.cu:670-772) / the pixel loop of cpu_launcher.cpp:693-718 including Scene::getColor
// (cpu:566-648), Scene::intersect_all (cpu:545-564) and TriangleMesh::intersect (cpu:238-313).
//
// The wavefront pipeline (rt_wavefront.hip.h) alternates a traversal kernel with a uniform shading kernel and streams
// every path's state through HBM between them: per frame 11 dependent launches per sub-frame and ~3 GB of path-state
// traffic for 33 MB of image.  Here a WAVE owns 64 paths from the camera ray to the framebuffer store and nothing but
// the result leaves the CU:
//
//   * ray slots 0..63 hold the paths' continuation rays (Y), slots 64..127 their shadow rays (X); the two rays that leave
//     a hit point do not depend on each other and are traced together, exactly as launch j of the wavefront pipeline does;
//   * traversal is the work stack of rt_travq.hip.h: one LIFO of (ray slot, sibling pair) entries per wave, BOX steps of
//     64 pairs, TRI steps of 128 triangles, per-slot counters of outstanding entries -- lanes carry no per-ray state;
//   * when a path's two rays have no entries left the path is READY; once enough paths are ready (or nothing else is
//     left to do) the wave runs one SHADE step: lane p closes path p's queries (intersect_all's strict '<' replay),
//     runs getColor's branch for the hit, folds finished paths into the framebuffer and starts new ones from the
//     workgroup's share of the pixels, then emits the path's continuation ray and its shadow ray (ray/sphere tests,
//     root-box test, filter constants, stack push);
//   * per-path state (flags, object ids, sphere hits, the l of every diffuse segment) lives in the wave's LDS carve.
//
// HBM traffic: the scene (cache resident), 16 B per pixel of output.  No inter-launch tails, no kernel boundaries: a
// 1/8 share of a frame costs 1/8 of the time plus one ray lifetime, which is what row-tile scaling over 8 GPUs needs.
// Samples of a pixel are independent paths (item = sample * n_paths + pixel slot); with more than one sample the paths
// write per-sample colours and path_reduce adds them in sample order, so the sum is bit-identical to the serial loop
// of cpu:701-713.
#pragma once
#include "rt_travq.hip.h"

namespace rtk {

constexpr int kPP = 64;                       // paths per wave: one per lane in a SHADE step
constexpr int kPR = 2 * kPP;                  // ray slots per wave: slot p = continuation ray of path p, slot kPP + p = its shadow ray
constexpr int kPStack = 896;                  // stack entries (sibling pairs) per wave; fuller -> serial drain, as in wf_travq
constexpr int kPLeafCap = 256;
constexpr int kPNodeBits = 25;                // stack entry = slot << 25 | node (7 bits of slot)
constexpr unsigned kPNodeMask = (1u << kPNodeBits) - 1u;

// (path record flags PF_*: rt_wavefront.hip.h)

struct PathState {
    int n_paths;          // pixel slots of the (sub-)frame in tile order: tiles_x * tiles_y * 64
    int tiles_x;
    int samp0, n_samp;    // this launch traces samples [samp0, samp0 + n_samp) of every pixel; item = (s - samp0) * n_paths + slot
    int log2S, Q, n_groups;   // scrambled static shares: item slot q -> item 4 g + (q & 3), g = ((q >> 2) & (S - 1)) * Q + ((q >> 2) >> log2S)
    int slots_per_block;      // item slots owned by one workgroup (multiple of 4)
    float4 *samp_out;     // [n_samp][n_paths] (colour of the sample, rays traced) when the frame has more than one sample; else nullptr
    // rt_trace_rays: the items are n_ext EXPLICIT rays (6 floats each: O, u) instead of camera rays; a path is one ray, and what leaves
    // the CU is its traversal result ext_out[item] = bits(t) << 32 | triangle (WF_NOHIT if none) instead of a colour
    const float *ext_rays;
    unsigned long long *ext_out;
    int n_ext;
};

struct PCarve {
    static constexpr int kTabA = 0;                            // float4[128]: (1/u by v_rcp_f32, filter constant | +inf)
    static constexpr int kTabC = kTabA + 16 * kPR;             // float4[128]: (O.xyz, u.x)
    static constexpr int kTabD = kTabC + 16 * kPR;             // float2[128]: (u.y, u.z)
    static constexpr int kBest = kTabD + 8 * kPR;              // u64[128]: nearest accepted hit of the slot's ray
    static constexpr int kPend = kBest + 8 * kPR;              // int[128]: outstanding stack + leaf-queue entries
    static constexpr int kPF = kPend + 4 * kPR;                // int4[64]: (flags, diffuse mask, object ids lo, hi)
    static constexpr int kPS = kPF + 16 * kPP;                 // float4[64]: (tA, tB of the Y ray's spheres; nearest sphere t of the X ray; refraction index)
    static constexpr int kPL = kPS + 16 * kPP;                 // float[64]: unshadowed l of the segment whose shadow ray is in flight (cpu:623)
    static constexpr int kPI = kPL + 4 * kPP;                  // int[64]: item index of the path
    static constexpr int kMarks = kPI + 4 * kPP;               // u8[128]
    static constexpr int kStack = kMarks + 128;                // u32[kPStack]
    static constexpr int kLeaf = kStack + 4 * kPStack;         // uint2[kPLeafCap]
    static constexpr int kLS = kLeaf + 8 * kPLeafCap;          // float[segs][64]: l of every diffuse segment (dynamic: the launch knows segs)
    static_assert(kLeaf % 8 == 0 && kLS % 16 == 0 && kStack % 4 == 0, "alignment of the carve");
    static constexpr int bytes(int segs) { return kLS + 4 * kPP * (segs > 0 ? segs : 1); }   // multiple of 16
};

__device__ __forceinline__ int path_slot_to_item(const PathState &ps, int q) {
    const int gs = q >> 2;
    const int col = gs >> ps.log2S;
    const int g = (gs & ((1 << ps.log2S) - 1)) * ps.Q + col;
    return (col < ps.Q && g < ps.n_groups) ? 4 * g + (q & 3) : -1;
}

template <bool STATS>
__global__ __launch_bounds__(kQBlock, 2) void wf_path(const Scene sc, const Frame fr, const PathState ps, const int cap, const int kLow, const int kShadeMin) {
    constexpr int P = kPP, SCAP = kPStack, LCAP = kPLeafCap;
    extern __shared__ __attribute__((aligned(16))) unsigned char path_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wib = tid >> 6;
    constexpr int wpb = kQBlock / 64;
    const int carve_bytes = PCarve::bytes(fr.segs);
    unsigned char *const wl = path_smem + wib * carve_bytes;
    int *const blk_cur = reinterpret_cast<int *>(path_smem + wpb * carve_bytes);
    float4 *const tabA = reinterpret_cast<float4 *>(wl + PCarve::kTabA);
    float4 *const tabC = reinterpret_cast<float4 *>(wl + PCarve::kTabC);
    float2 *const tabD = reinterpret_cast<float2 *>(wl + PCarve::kTabD);
    unsigned long long *const best = reinterpret_cast<unsigned long long *>(wl + PCarve::kBest);
    int *const pend = reinterpret_cast<int *>(wl + PCarve::kPend);
    int4 *const pF = reinterpret_cast<int4 *>(wl + PCarve::kPF);
    float4 *const pS = reinterpret_cast<float4 *>(wl + PCarve::kPS);
    float *const pL = reinterpret_cast<float *>(wl + PCarve::kPL);
    int *const pI = reinterpret_cast<int *>(wl + PCarve::kPI);
    unsigned char *const marks = wl + PCarve::kMarks;
    unsigned int *const stack = reinterpret_cast<unsigned int *>(wl + PCarve::kStack);
    uint2 *const leafq = reinterpret_cast<uint2 *>(wl + PCarve::kLeaf);
    float *const lsq = reinterpret_cast<float *>(wl + PCarve::kLS);       // lsq[d * P + p]
    if (tid == 0) *blk_cur = 0;
    marks[lane] = 0; marks[lane + 64] = 0;
    pend[lane] = 0; pend[P + lane] = 0;
    pF[lane] = make_int4(0, 0, 0, 0);
    __syncthreads();

    const float4 *const nodes = sc.nodesq;
    const size_t blk_base = (size_t)blockIdx.x * (size_t)ps.slots_per_block;
    const int blk_n = ps.slots_per_block;
    const int root_hiw = __float_as_int(sc.root_hi.w);
    const bool have_mesh = sc.mesh_slot >= 0 && sc.n_nodes > 0;
    int top = 0;
    unsigned int lhead = 0, ltail = 0;
    bool drained = false;
    Work wk;
#define PQ_CHECK(cond, bit, fixup) do { if (STATS && !(cond)) { atomicOr(&fr.work[4], (unsigned long long)(bit)); fixup; } } while (0)

    // stack nearly full: walk the subtree of one popped entry serially with the stackless (skip-pointer) node array
    auto drain_serial = [&](int o, int node) {
        const float4 A = tabA[o], C = tabC[o];
        const float2 D = tabD[o];
        const f3 O = mk(C.x, C.y, C.z), u = mk(C.w, D.x, D.y);
        const int xt = sc.q2thr[node];
        const float4 h0 = sc.nodes[2 * xt + 1];
        const int end = __float_as_int(h0.w) >= 0 ? xt + 1 : __float_as_int(sc.nodes[2 * xt].w);
        for (int x = xt; x < end;) {
            const float4 lo = sc.nodes[2 * x], hi = sc.nodes[2 * x + 1];
            const int hiw = __float_as_int(hi.w), low = __float_as_int(lo.w);
            bool hit;
            if (!qbox_filter(lo, hi, A, C, hit)) { hit = slab(lo, hi, O, u); if (STATS) wk.lit_box++; }
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
        top = __builtin_amdgcn_readfirstlane(top);
        lhead = (unsigned int)__builtin_amdgcn_readfirstlane((int)lhead);
        ltail = (unsigned int)__builtin_amdgcn_readfirstlane((int)ltail);
        drained = __builtin_amdgcn_readfirstlane((int)drained) != 0;
        // =============================== SHADE: ready paths ===============================
        if (top < kLow) {
            int4 F = pF[lane];
            bool alive = (F.x & PF_ALIVE) != 0;
            const bool ready = alive ? (pend[lane] == 0 && pend[P + lane] == 0) : !drained;
            const unsigned long long rm = __ballot(ready);
            const int n_ready = __popcll(rm);
            const bool idle = top == 0 && ltail == lhead;
            if (n_ready == 0 && idle) {
                if (__ballot(alive) == 0ull) break;        // no ray in flight, no path alive, no pixel left: every wave gets here
                // (alive paths without entries are ready; this point is not reachable with alive paths)
            }
            if (n_ready >= kShadeMin || (idle && n_ready > 0)) {
                const float PI_F = (float)3.14159265358979323846;
                const double PI_D = 3.14159265358979323846;
                const f3 L = mk(sc.Lx, sc.Ly, sc.Lz);
                float4 S = make_float4(0, 0, 0, 1.f);
                int item = -1;
                bool emitY = false, emitX = false;
                f3 Oy = mk(0, 0, 0), uy = mk(0, 0, 1), Ox = mk(0, 0, 0), ux = mk(0, 0, 1);
                int px = 0, lrow = 0, samp = ps.samp0, slot_i = 0;
                auto decode_item = [&](int it) {
                    int srel = 0;
                    if (ps.n_samp > 1) srel = it / ps.n_paths;
                    slot_i = it - srel * ps.n_paths;
                    samp = ps.samp0 + srel;
                    const int tile = slot_i >> 6, p = slot_i & 63;
                    px = (tile % ps.tiles_x) * 8 + (p & 7);
                    lrow = (tile / ps.tiles_x) * 8 + (p >> 3);
                };
                auto image_row = [&]() { return fr.row0 + (lrow / fr.tile_rows) * fr.tile_rows * fr.tile_step + (lrow % fr.tile_rows); };
                // the sample is finished: its colour `ans` and ray count leave the CU (cpu:711-713)
                auto write_result = [&](f3 ans, float rays) {
                    if (ps.samp_out != nullptr) {
                        ps.samp_out[item] = make_float4(ans.x, ans.y, ans.z, rays);
                    } else {                                           // one sample per pixel: T = 0 + ans, out = T / n
                        float tx = 0.f, ty = 0.f, tz = 0.f;
                        if (fr.cam_mode == 1) { tx += ans.x * fr.inv_n; ty += ans.y * fr.inv_n; tz += ans.z * fr.inv_n; }   // realtime:1131
                        else { tx += ans.x; ty += ans.y; tz += ans.z; }
                        const float n = fr.cam_mode == 1 ? 1.f : (float)fr.spp;
                        fr.out[out_index(fr, lrow, px)] = make_float4(tx / n, ty / n, tz / n, rays);
                    }
                };
                if (ready && alive && ps.ext_rays != nullptr) {               // rt_trace_rays: the ray's traversal result is the path's result
                    item = pI[lane];
                    PQ_CHECK(item >= 0 && item < ps.n_ext, 1, item = 0);
                    ps.ext_out[item] = (F.x & PF_MESHY) ? best[lane] : WF_NOHIT;
                    alive = false;
                    F = make_int4(0, 0, 0, 0);
                } else if (ready && alive) {
                    S = pS[lane];
                    item = pI[lane];
                    decode_item(item);
                    int d = (F.x >> PF_DEPTH_SHIFT) & PF_DEPTH_MASK;          // segment of the continuation ray in flight
                    int nrays = (F.x >> PF_RAYS_SHIFT) & PF_RAYS_MASK;
                    float refr = S.w;
                    // ---- (1) the shadow ray of segment d-1's hit came back: direct light or not (cpu:615) ----
                    if (F.x & PF_HASX) {
                        const float4 x0 = tabC[P + lane];
                        const float2 x1 = tabD[P + lane];
                        const f3 Oxr = mk(x0.x, x0.y, x0.z), uxr = mk(x0.w, x1.x, x1.y);
                        float t_min = S.z;                                    // nearest sphere (the value of intersect_all's running minimum)
                        if (F.x & PF_MESHX) {
                            const unsigned long long m = best[P + lane];
                            if (m != WF_NOHIT) { const float tm = __uint_as_float((unsigned int)(m >> 32)); if (tm < t_min) t_min = tm; }
                        }
                        const f3 Pp = Oxr + t_min * uxr;                      // cpu:560 (Ox is P_adjusted)
                        const bool lit = !(norm2(Pp - Oxr) <= norm2(L - Oxr));   // cpu:615
                        lsq[(d - 1) * P + lane] = lit ? pL[lane] : 0.f;
                    }
                    // ---- (2) the continuation ray of segment d came back: Scene::getColor's branch for its hit (cpu:570-614) ----
                    if (F.x & PF_HASY) {
                        const float4 r0 = tabC[lane];
                        const float2 r1 = tabD[lane];
                        f3 O = mk(r0.x, r0.y, r0.z), u = mk(r0.w, r1.x, r1.y);
                        // Scene::intersect_all's running minimum (strict '<' in object order, cpu:554) = the lexicographic minimum over (t, position in Scene::objects): the spheres'
                        // own winner was decided at emission, the meshes' by the traversal; between the two a tie goes to the earlier object (rt_kernels.hip.h mesh_beats_sphere)
                        float t_min = S.x;
                        int win = ((F.x >> PF_WINS_SHIFT) & 31) - 1, tri_win = -1;
                        if (F.x & PF_MESHY) {
                            const unsigned long long m = best[lane];
                            if (m != WF_NOHIT) {
                                const float tm = __uint_as_float((unsigned int)(m >> 32));
                                const int mobj = mesh_obj_of_tri(sc, (int)(unsigned int)m);
                                if (mesh_beats_sphere(t_min, win, tm, mobj)) { t_min = tm; win = mobj; tri_win = (int)(unsigned int)m; }
                            }
                        }
                        if (win >= 0) {                                       // a miss is black (cpu:571): nothing to emit
                            const f3 Pt = O + t_min * u;                      // cpu:560
                            f3 N;
                            if (tri_win >= 0 && sc.nrm != nullptr) {          // get_smooth_normal, realtime_render.cu:221-245
                                PQ_CHECK(tri_win >= 0 && tri_win < sc.n_tris, 2, tri_win = 0);
                                const float4 q0 = sc.tri[3 * tri_win], q1 = sc.tri[3 * tri_win + 1], q2 = sc.tri[3 * tri_win + 2];
                                const f3 A = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x), Nt = mk(q2.y, q2.z, q2.w);
                                const float beta = dot(e2, cross(A - O, u)) / dot(u, Nt);
                                const float gamma = -dot(e1, cross(A - O, u)) / dot(u, Nt);
                                const float alpha = 1 - beta - gamma;
                                const float4 na = sc.nrm[3 * tri_win], nb = sc.nrm[3 * tri_win + 1], nc = sc.nrm[3 * tri_win + 2];
                                N = normalize((alpha * mk(na.x, na.y, na.z) + beta * mk(nb.x, nb.y, nb.z)) + gamma * mk(nc.x, nc.y, nc.z));
                            } else if (tri_win >= 0) {
                                PQ_CHECK(tri_win >= 0 && tri_win < sc.n_tris, 2, tri_win = 0);
                                const float4 q2 = sc.tri[3 * tri_win + 2];
                                N = normalize(mk(q2.y, q2.z, q2.w));          // cpu:308
                            } else {
                                N = normalize(Pt - sphere_centre_of(sc, win));  // cpu:524-525
                            }
                            const Material m = material_of(sc, win);
                            bool cont = false;                                // a continuation ray of segment d+1 was built in (O,u)
                            if (m.mirror) {                                   // cpu:573-579
                                O = Pt + fr.eps * N;
                                u = u - (2 * dot(u, N)) * N;
                                cont = true;
                            } else if (m.n_in != m.n_out) {                   // cpu:580-604
                                float ratio;
                                const bool out2in = refr == m.n_out;
                                if (out2in) ratio = m.n_out / m.n_in;
                                else { ratio = m.n_in / m.n_out; N = -N; }
                                const float un = dot(u, N);
                                if (((out2in && refr > m.n_in) || (!out2in && refr > m.n_out)) && (ratio * ratio) * (1 - un * un) > 1) {
                                    O = Pt + fr.eps * N;
                                    u = u - (2 * un) * N;
                                } else {
                                    O = Pt - fr.eps * N;
                                    const f3 Nc = (-rt_sqrtf(1 - (ratio * ratio) * (1 - un * un))) * N;
                                    const f3 Tc = ratio * (u - un * N);
                                    u = Nc + Tc;
                                    refr = out2in ? m.n_in : m.n_out;
                                }
                                cont = true;
                            } else {                                          // cpu:605-642: diffuse
                                const f3 Pa = Pt + fr.eps * N;
                                const f3 toL = L - Pa;
                                Ox = Pa; ux = normalize(toL);   // = toL / sqrt(norm2(toL))     // NORMED_VEC, cpu:614: the shadow ray of segment d
                                emitX = true;
                                nrays += 1;
                                // its direct term if the light turns out to be visible (cpu:620-623), kept until the shadow ray is back
                                const f3 wl = normalize(L - Pt);
                                const float dn = dot(N, wl);
                                const float mx = (dn < 0.f) ? 0.f : dn;
                                pL[lane] = (float)((double)sc.intensity / (4 * PI_D * (double)norm2(L - Pt)) * (double)mx);
                                const uint64_t ids = ((uint64_t)(uint32_t)F.w << 32 | (uint32_t)F.z) | (uint64_t)(win & 15) << (4 * d);
                                F.z = (int)(uint32_t)ids; F.w = (int)(uint32_t)(ids >> 32);
                                F.y |= 1 << d;
                                if (d + 1 < fr.segs) {                        // the bounce ray (cpu:627-642): needs r1, r2 and N only
                                    const int row = image_row();
                                    const uint32_t hp = mix32(((uint32_t)row * (uint32_t)fr.W + (uint32_t)px) ^ mix32(fr.seed));
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
                                    refr = 1.f;                               // Ray(P_adjusted, random_direction): index 1
                                    cont = true;
                                }
                            }
                            if (cont && d + 1 < fr.segs) {
                                emitY = true; Oy = O; uy = u;                 // continuation ray of segment d+1
                                nrays += 1;
                            }
                        }
                        d = d + 1;
                    }
                    S.w = refr;
                    if (emitX || emitY) {
                        F.x = PF_ALIVE | (emitX ? PF_HASX : 0) | (emitY ? PF_HASY : 0) | (d << PF_DEPTH_SHIFT) | (nrays << PF_RAYS_SHIFT);
                    } else {   // nothing in flight: fold the path back to front (cpu:642-644) and hand the sample over (cpu:711)
                        f3 ans = mk(0, 0, 0);
                        const int nseg = d < fr.segs ? d : fr.segs;
                        const uint64_t ids = (uint64_t)(uint32_t)F.w << 32 | (uint32_t)F.z;
                        for (int k = nseg - 1; k >= 0; --k) {
                            if (F.y & (1 << k)) {
                                const Material m = material_of(sc, (int)((ids >> (4 * k)) & 15));
                                const float l = lsq[k * P + lane];
                                const f3 alb = mk(m.ar, m.ag, m.ab);
                                ans = (l * alb) / PI_F + alb * ans;
                            }
                        }
                        write_result(ans, (float)nrays);
                        alive = false;
                        F = make_int4(0, 0, 0, 0);
                    }
                }
                // ---- (3) free path slots take the next pixels of the workgroup's share: camera ray (cpu:699-709) ----
                const bool want = ready && !alive && !drained;
                const unsigned long long wm = __ballot(want);
                if (wm != 0ull) {
                    const int n_new = __popcll(wm);
                    int base = 0;
                    if (lane == 0) base = atomicAdd(blk_cur, n_new);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + n_new >= blk_n) drained = true;
                    if (want) {
                        const int qo = base + lanes_below(wm);
                        item = qo < blk_n ? path_slot_to_item(ps, (int)blk_base + qo) : -1;
                        if (item >= 0 && ps.ext_rays != nullptr) {            // rt_trace_rays: item = ray index, the ray as given (not normalised)
                            if (item < ps.n_ext) {
                                const float *er = ps.ext_rays + 6 * (size_t)item;
                                Oy = mk(er[0], er[1], er[2]);
                                uy = mk(er[3], er[4], er[5]);
                                emitY = true;
                                alive = true;
                                F = make_int4(PF_ALIVE | PF_HASY | (1 << PF_RAYS_SHIFT), 0, 0, 0);
                                S = make_float4(0, 0, 0, 1.f);
                            }
                        } else if (item >= 0) {
                            decode_item(item);
                            const bool valid = px < fr.W && lrow < fr.n_rows;
                            if (valid && fr.segs <= 0) {                       // optimized.cu convention with num_bounce 0: black
                                write_result(mk(0, 0, 0), 0.f);
                            } else if (valid) {
                                const int row = image_row();
                                // cpu:699: +0.5/-0.5 are double literals, narrowed by the Vector constructor
                                const f3 uc = mk((float)((double)((float)px - (float)fr.W / 2) + 0.5),
                                                 (float)((double)((float)fr.H / 2 - (float)row) - 0.5), fr.z);
                                f3 ucm = uc;
                                if (fr.cam_mode == 1) {   // realtime:1115: cam.C + cam.bz * z + cam.bx * X + cam.by * Y
                                    const f3 Cc = mk(sc.camx, sc.camy, sc.camz), Bx = mk(fr.bx[0], fr.bx[1], fr.bx[2]), By = mk(fr.by[0], fr.by[1], fr.by[2]), Bz = mk(fr.bz[0], fr.bz[1], fr.bz[2]);
                                    const f3 a = Cc + mk(Bz.x * fr.z, Bz.y * fr.z, Bz.z * fr.z);
                                    const f3 b = a + mk(Bx.x * uc.x, Bx.y * uc.x, Bx.z * uc.x);
                                    ucm = b + mk(By.x * uc.y, By.y * uc.y, By.z * uc.y);
                                }
                                f3 uu = ucm;
                                if (fr.sigma != 0.f) {   // cpu:705-707; with sigma == 0 the jitter is exactly +-0
                                    const uint32_t hp = mix32(((uint32_t)row * (uint32_t)fr.W + (uint32_t)px) ^ mix32(fr.seed));
                                    const uint32_t hs = mix32(hp ^ ((uint32_t)samp * 0x9E3779B1U));
                                    const float r1 = uniform01(hs, 0, 2), r2 = uniform01(hs, 0, 3);
                                    const float bm = fr.sigma * rt_sqrtf(-2 * logf(r1));
                                    double sn, cs;
                                    rt_sincos_2pi(2 * 3.14159265358979323846 * (double)r2, sn, cs);
                                    uu = ucm + mk((float)((double)bm * cs), (float)((double)bm * sn), 0.f);
                                }
                                Oy = mk(sc.camx, sc.camy, sc.camz);
                                uy = normalize(uu);
                                emitY = true;
                                alive = true;
                                F = make_int4(PF_ALIVE | PF_HASY | (1 << PF_RAYS_SHIFT), 0, 0, 0);   // segment 0, one ray
                                S = make_float4(0, 0, 0, 1.f);                // Ray::refraction_index = 1 (cpu:100)
                            }
                        }
                    }
                }
                // ---- (4) emission: the continuation ray of path p into slot p, then its shadow ray into slot 64 + p ----
                bool needY = false, needX = false;
#pragma unroll 1
                for (int k = 0; k < 2; ++k) {
                    const bool on = k == 0 ? emitY : emitX;
                    const f3 O = k == 0 ? Oy : Ox, u = k == 0 ? uy : ux;
                    const int slot = k * P + lane;
                    bool need = false;
                    if (on) {
                        const SphereNear h = spheres_near1(sc, O, u);                   // Sphere::intersect x n (cpu:512-527)
                        tabC[slot] = make_float4(O.x, O.y, O.z, u.x);
                        tabD[slot] = make_float2(u.y, u.z);
                        if (k == 0) {
                            S.x = h.t;
                            F.x = (F.x & ~(1023 << PF_WINS_SHIFT)) | (((h.obj + 1) & 31) << PF_WINS_SHIFT);
                        } else {
                            S.z = h.t;                                                   // only the value of the shadow ray's nearest hit matters
                        }
                        if (have_mesh) {                                                 // root-box test (cpu:279)
                            if (STATS) wk.box++;
                            if (slab_filtered(sc.root_lo, sc.root_hi, O, u, ray_inv(u))) {
                                if (STATS) wk.nodes++;
                                need = true;
                                const RayBox rb = ray_box(O, u);
                                tabA[slot] = make_float4(rb.rx, rb.ry, rb.rz, rb.safe ? rb.c0 : __builtin_inff());
                                best[slot] = WF_NOHIT;
                            }
                        }
                    }
                    const unsigned long long nm = __ballot(need);
                    if (root_hiw < 0) {                                                  // the root (node 1) has the children 2, 3
                        if (need) { stack[top + lanes_below(nm)] = (unsigned int)slot << kPNodeBits | 2u; pend[slot] = 1; }
                        top += __popcll(nm);
                    } else {                                                             // the root is a leaf
                        const int first = __float_as_int(sc.root_lo.w), cnt = root_hiw - first;
                        if (cnt > 0) {
                            if (need) {
                                leafq[(ltail + (unsigned int)lanes_below(nm)) & (LCAP - 1)] = make_uint2((unsigned int)first, (unsigned int)slot | (unsigned int)cnt << 8);
                                pend[slot] = 1;
                                if (STATS) wk.tris += (uint32_t)cnt;
                            }
                            ltail += (unsigned int)__popcll(nm);
                        }
                    }
                    if (k == 0) needY = need; else needX = need;
                }
                if (ready) {
                    F.x = (F.x & ~(PF_MESHY | PF_MESHX)) | (needY ? PF_MESHY : 0) | (needX ? PF_MESHX : 0);
                    pF[lane] = F;
                    pS[lane] = S;
                    pI[lane] = item;
                }
                __builtin_amdgcn_wave_barrier();
                continue;                                                                // re-evaluate: more paths may be ready, the stack has entries now
            }
        }
        // =============================== TRI step: two triangles per lane ===============================
        const unsigned int lcount = ltail - lhead;
        if (lcount >= 64u || (top == 0 && lcount > 0u)) {
            const unsigned int m = lcount < 64u ? lcount : 64u;
            uint2 E = make_uint2(0u, 0u);
            if ((unsigned int)lane < m) E = leafq[(lhead + (unsigned int)lane) & (LCAP - 1)];
            const unsigned int c = E.y >> 8;
            const unsigned int incl = wave_incl_scan(c);
            const unsigned int Pq = incl - c;
            const bool part = c > 0u && Pq < 128u;
            const unsigned int all = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
            const unsigned int total = all < 128u ? all : 128u;
            if (part) marks[Pq] = 1;
            __builtin_amdgcn_wave_barrier();
            const unsigned int mk0 = marks[lane], mk1 = marks[lane + 64];
            const unsigned long long B0 = __ballot(mk0 != 0u), B1 = __ballot(mk1 != 0u);
            __builtin_amdgcn_wave_barrier();
            if (part) marks[Pq] = 0;
            const int j0 = lanes_below(B0) + (int)((B0 >> lane) & 1ull) - 1;
            const int j1 = __popcll(B0) + lanes_below(B1) + (int)((B1 >> lane) & 1ull) - 1;
            const unsigned int f0 = (unsigned int)__shfl((int)E.x, j0, 64), y0 = (unsigned int)__shfl((int)E.y, j0, 64), P0 = (unsigned int)__shfl((int)Pq, j0, 64);
            const unsigned int f1 = (unsigned int)__shfl((int)E.x, j1, 64), y1 = (unsigned int)__shfl((int)E.y, j1, 64), P1 = (unsigned int)__shfl((int)Pq, j1, 64);
            const bool t0 = (unsigned int)lane < total, t1 = (unsigned int)lane + 64u < total;
            const int o0 = t0 ? (int)(y0 & 0xffu) : 0, o1 = t1 ? (int)(y1 & 0xffu) : 0;
            int i0 = t0 ? (int)(f0 + ((unsigned int)lane - P0)) : 0, i1 = t1 ? (int)(f1 + ((unsigned int)lane + 64u - P1)) : 0;
            PQ_CHECK(i0 >= 0 && i0 < sc.n_tris && i1 >= 0 && i1 < sc.n_tris, 2, (i0 = 0, i1 = 0));
            PQ_CHECK(ltail - lhead <= (unsigned int)LCAP, 16, (void)0);
            const float4 *tp0 = sc.tri + 3 * (size_t)i0, *tp1 = sc.tri + 3 * (size_t)i1;
            const float4 a0 = tp0[0], a1 = tp0[1], a2 = tp0[2];
            const float4 b0 = tp1[0], b1 = tp1[1], b2 = tp1[2];
            const float4 C0 = tabC[o0], C1 = tabC[o1];
            const float2 D0 = tabD[o0], D1 = tabD[o1];
            float ta, tb;
            int how0, how1;
            const bool ok0 = qtri_test(a0, a1, a2, mk(C0.x, C0.y, C0.z), mk(C0.w, D0.x, D0.y), fr.tri_tmin, ta, how0) && t0;
            const bool ok1 = qtri_test(b0, b1, b2, mk(C1.x, C1.y, C1.z), mk(C1.w, D1.x, D1.y), fr.tri_tmin, tb, how1) && t1;
            if (STATS) wk.lit_tri += ((t0 && (how0 & 3) == 2) ? 1u : 0u) + ((t1 && (how1 & 3) == 2) ? 1u : 0u);
            if (ok0) atomicMin(&best[o0], (unsigned long long)__float_as_uint(ta) << 32 | (unsigned int)i0);
            if (ok1) atomicMin(&best[o1], (unsigned long long)__float_as_uint(tb) << 32 | (unsigned int)i1);
            const bool full = part && Pq + c <= 128u;
            if (part && !full) {
                const unsigned int took = 128u - Pq;
                leafq[(lhead + (unsigned int)lane) & (LCAP - 1)] = make_uint2(E.x + took, (E.y & 0xffu) | (c - took) << 8);
            }
            lhead += (unsigned int)__popcll(__ballot(full));
            if (full) atomicAdd(&pend[E.y & 0xffu], -1);            // after the mins above (LDS operations stay in order)
            continue;
        }
        if (top == 0) continue;                                      // the SHADE section decides: more paths, or the end
        // =============================== BOX step: one sibling pair (two boxes) per lane ===============================
        const int n = top < 64 ? top : 64;
        if (cap - top < 64) {                                        // no room for up to 128 pushes: serial drain of the popped entries
            const bool actd = lane < n;
            const unsigned int ed = actd ? stack[top - 1 - lane] : 0u;
            top -= n;
            if (actd) { const int od = (int)(ed >> kPNodeBits), cd = (int)(ed & kPNodeMask); drain_serial(od, cd); drain_serial(od, cd + 1); atomicAdd(&pend[od], -1); }
            continue;
        }
        const bool act = lane < n;
        const unsigned int e = act ? stack[top - 1 - lane] : 0u;     // slot << 26 | c: the sibling nodes c, c + 1 (one 64-byte line)
        const int o = (int)(e >> kPNodeBits);
        int c = (int)(e & kPNodeMask);
        top -= n;
        PQ_CHECK(!act || (c >= 2 && c + 1 <= sc.n_nodes), 4, c = 0);
        const float4 A = tabA[o], C = tabC[o];
        const float4 lo0 = nodes[2 * c], hi0 = nodes[2 * c + 1], lo1 = nodes[2 * c + 2], hi1 = nodes[2 * c + 3];
        bool hit0, hit1;
        const bool dec0 = qbox_filter(lo0, hi0, A, C, hit0);
        const bool dec1 = qbox_filter(lo1, hi1, A, C, hit1);
        const bool und = act && !(dec0 && dec1);
        if (STATS) wk.lit_box += act ? (dec0 ? 0u : 1u) + (dec1 ? 0u : 1u) : 0u;
        if (__builtin_expect(__ballot(und) != 0ull, 0)) {            // literal arithmetic for undecided lanes (six IEEE divisions, almost never needed)
            if (und) {
                const float2 D = tabD[o];
                const f3 O = mk(C.x, C.y, C.z), u = mk(C.w, D.x, D.y);
                if (!dec0) hit0 = slab(lo0, hi0, O, u);
                if (!dec1) hit1 = slab(lo1, hi1, O, u);
            }
        }
        {
            const int hiw0 = __float_as_int(hi0.w), low0 = __float_as_int(lo0.w), cnt0 = hiw0 - low0;
            const int hiw1 = __float_as_int(hi1.w), low1 = __float_as_int(lo1.w), cnt1 = hiw1 - low1;
            const bool h0 = hit0 && act, h1 = hit1 && act;
            const bool hI0 = h0 && hiw0 < 0, hI1 = h1 && hiw1 < 0;
            const bool hL0 = h0 && hiw0 >= 0 && cnt0 > 0, hL1 = h1 && hiw1 >= 0 && cnt1 > 0;
            if (STATS) {
                wk.box += act ? 2u : 0u; wk.nodes += (h0 ? 1u : 0u) + (h1 ? 1u : 0u);
                wk.tris += ((h0 && hiw0 >= 0) ? (uint32_t)cnt0 : 0u) + ((h1 && hiw1 >= 0) ? (uint32_t)cnt1 : 0u);
            }
            const unsigned long long mI0 = __ballot(hI0), mI1 = __ballot(hI1), mL0 = __ballot(hL0), mL1 = __ballot(hL1);
            const int nI0 = __popcll(mI0), nL0 = __popcll(mL0);
            const unsigned int sbits = e & ~kPNodeMask;
            if (hI0) stack[top + lanes_below(mI0)] = sbits | (unsigned int)low0;        // a hit internal node pushes ITS pair of children
            if (hI1) stack[top + nI0 + lanes_below(mI1)] = sbits | (unsigned int)low1;
            top += nI0 + __popcll(mI1);
            if (hL0) leafq[(ltail + (unsigned int)lanes_below(mL0)) & (LCAP - 1)] = make_uint2((unsigned int)low0, (unsigned int)o | (unsigned int)cnt0 << 8);
            if (hL1) leafq[(ltail + (unsigned int)(nL0 + lanes_below(mL1))) & (LCAP - 1)] = make_uint2((unsigned int)low1, (unsigned int)o | (unsigned int)cnt1 << 8);
            ltail += (unsigned int)(nL0 + __popcll(mL1));
            const int delta = (hI0 ? 1 : 0) + (hI1 ? 1 : 0) + (hL0 ? 1 : 0) + (hL1 ? 1 : 0) - (act ? 1 : 0);
            if (delta != 0) atomicAdd(&pend[o], delta);
        }
        PQ_CHECK(top >= 0 && top <= cap && top <= SCAP, 8, (void)0);
    }
#undef PQ_CHECK
    wf_flush_work<STATS>(fr, wk);
}

// Sum of the samples of every pixel in sample order (cpu:701-713: color_avg += color; color_avg /= num_rays), from the
// per-sample colours wf_path wrote.  A frame's samples may come in several chunks (launches): T carries the running sum.
__global__ __launch_bounds__(256) void path_reduce(const Frame fr, const int n_px, const int tiles_x, const int n_samp, const float4 *__restrict__ samp_out,
                                                   float4 *__restrict__ T, int first, int last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px) return;
    const int tile = i >> 6, p = i & 63;
    const int px = (tile % tiles_x) * 8 + (p & 7), lrow = (tile / tiles_x) * 8 + (p >> 3);
    if (!(px < fr.W && lrow < fr.n_rows)) return;
    float4 t = first ? make_float4(0, 0, 0, 0) : T[i];
    for (int s = 0; s < n_samp; ++s) {
        const float4 a = samp_out[(size_t)s * n_px + i];
        if (fr.cam_mode == 1) { t.x += a.x * fr.inv_n; t.y += a.y * fr.inv_n; t.z += a.z * fr.inv_n; }   // realtime:1131
        else { t.x += a.x; t.y += a.y; t.z += a.z; }
        t.w += a.w;
    }
    if (last) {
        const float n = fr.cam_mode == 1 ? 1.f : (float)fr.spp;
        fr.out[out_index(fr, lrow, px)] = make_float4(t.x / n, t.y / n, t.z / n, t.w);
    } else {
        T[i] = t;
    }
}

}  // namespace rtk
