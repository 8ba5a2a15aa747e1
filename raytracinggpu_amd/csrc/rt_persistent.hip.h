// rt_persistent.hip.h -- persistent-lane render kernel for gfx950 (the fast path).
//
// Why: with one lane bound to one pixel for a whole lock-step ray query, a wave waits for its slowest
// lane: measured VALU lane utilisation of the lock-step kernel on the cat scene is 15 % (half the
// rays miss the mesh's root box after one test, the others walk ~40 nodes).  Here every lane is still
// one pixel at a time, but it runs its own program counter through six micro-ops
//
//     NEXT  -> START -> BOX* -> TRI* -> ... -> HIT | SHADE -> START ... -> NEXT
//
//   NEXT   close the previous path (back-to-front fold, cpu:642-644), write the pixel when its samples
//          are done (float4 store), pull a new pixel from the global queue, build the camera ray (cpu:699-709)
//   START  begin a nearest-hit query: the spheres that precede the mesh in Scene::objects (cpu:549-558)
//   BOX    one BoundingBox::intersect of the stackless traversal (cpu:146-157, 284-293)
//   TRI    one moller_trumbore of the current leaf (cpu:226-236, 295-305)
//   HIT    end of a path-segment query: remaining spheres, P/N (cpu:560-562), material branch (cpu:573-611),
//          emits the shadow ray or the reflected/refracted ray
//   SHADE  end of a shadow query: direct light (cpu:615-625), cosine-weighted bounce ray (cpu:627-642)
//
// and the wave executes, each iteration, the micro-op most lanes are waiting for (one __ballot per
// phase, s_bcnt1, scalar branch).  Lanes never wait for another pixel's ray to finish, only for their own
// next micro-op to be scheduled.  All arithmetic is the literal / filtered-exact arithmetic of
// rt_kernels.hip.h, so results are bit-identical to the lock-step kernel and to the CPU oracle.
#pragma once
#include "rt_kernels.hip.h"

namespace rtk {

enum Phase : int { PH_NEXT = 0, PH_START = 1, PH_BOX = 2, PH_TRI = 3, PH_HIT = 4, PH_SHADE = 5, PH_DONE = 6 };

struct PFrame {
    Frame f;
    unsigned int *queue;      // global pixel-slot counter, zeroed before the launch
    int tiles_x;              // ceil(W / 8)
    unsigned int n_slots;     // tiles_x * ceil(n_rows / 8) * 64
};

constexpr int kPBlock = 256;

// every sphere of Scene::sph in insertion order with the strict '<' of cpu:554 (the earliest of equal t); win = its position in Scene::objects.  The mesh(es) join after the
// traversal: intersect_all's running minimum is the lexicographic minimum over (t, position), whichever way round it is formed (mesh_beats_sphere)
__device__ __forceinline__ void spheres_all(const Scene &sc, f3 O, f3 u, float &t_min, int &win) {
    for (int k = 0; k < sc.n_spheres; ++k) {
        const Sphere &s = sc.sph[k];
        const f3 C = mk(s.cx, s.cy, s.cz);
        const f3 OC = O - C;
        const float d = dot(u, OC);
        const float delta = d * d - (norm2(OC) - s.R * s.R);       // cpu:513
        if (delta < 0) continue;
        const float sq = rt_sqrtf(delta);
        const float b = dot(u, C - O);
        const float t1 = b - sq, t2 = b + sq;                      // cpu:516-517
        if (t2 < 0) continue;
        const float t = t1 < 0 ? t2 : t1;
        if (t < t_min) { t_min = t; win = s.obj; }
    }
}

template <bool STATS>
__global__ __launch_bounds__(kPBlock) void render_persistent(const Scene sc, const PFrame pf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Frame &fr = pf.f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    float *lstack = smem + tid;                 // lstack[d * kPBlock]
    const float PI_F = (float)3.14159265358979323846;
    const double PI_D = 3.14159265358979323846;
    const f3 L = mk(sc.Lx, sc.Ly, sc.Lz);
    const f3 Cam = mk(sc.camx, sc.camy, sc.camz);
    const int n_nodes = sc.n_nodes;
    const int mesh_slot = sc.mesh_slot;

    // ---- lane state ----
    int phase = PH_NEXT;
    bool have_pixel = false, path_open = false, shadow = false;
    int px = 0, lrow = 0, samp = 0, d = 0;
    uint32_t hp = 0, hs = 0;
    f3 O = mk(0, 0, 0), u = mk(0, 0, 1);
    RayInv ri = ray_inv(u);
    float t_min = 1e9f; int win = -1;
    int node = 0, ti = 0, te = 0; float tm = 1e9f; f3 Nb = mk(0, 0, 0); bool many = false; int tbest = 0;   // tbest: the winning triangle (names its mesh: rt_kernels.hip.h mesh_obj_of_tri)
    f3 Ps = mk(0, 0, 0), Ns = mk(0, 0, 0); int sid = 0;             // surface being shaded (shadow query in flight)
    float refr = 1.f; uint64_t ids = 0; uint32_t dmask = 0;
    f3 total = mk(0, 0, 0); float rays = 0.f;
    Work wk;

    for (;;) {
        // ---- scheduler: run the micro-op most lanes are waiting for ----
        const int nNext = __popcll(__ballot(phase == PH_NEXT));
        const int nStart = __popcll(__ballot(phase == PH_START));
        const int nBox = __popcll(__ballot(phase == PH_BOX));
        const int nTri = __popcll(__ballot(phase == PH_TRI));
        const int nHit = __popcll(__ballot(phase == PH_HIT));
        const int nShade = __popcll(__ballot(phase == PH_SHADE));
        int sel = PH_BOX, best = nBox;
        if (nTri > best) { best = nTri; sel = PH_TRI; }
        if (nStart > best) { best = nStart; sel = PH_START; }
        if (nHit > best) { best = nHit; sel = PH_HIT; }
        if (nShade > best) { best = nShade; sel = PH_SHADE; }
        if (nNext > best) { best = nNext; sel = PH_NEXT; }
        if (best == 0) break;                    // every lane is PH_DONE

        if (sel == PH_BOX) {
            if (phase == PH_BOX) {
                const float4 lo = sc.node_lo[node];
                const float4 hi = sc.node_hi[node];
                const int hiw = __float_as_int(hi.w);
                const int low = __float_as_int(lo.w);
                if (STATS) wk.box++;
                if (slab_filtered(lo, hi, O, u, ri)) {
                    if (STATS) wk.nodes++;
                    if (hiw >= 0 && low < hiw) {          // leaf with triangles [low, hiw)
                        if (STATS) wk.tris += (uint32_t)(hiw - low);
                        ti = low; te = hiw; phase = PH_TRI;
                    }
                    node = node + 1;
                } else {
                    node = (hiw < 0) ? low : node + 1;
                }
                if (phase == PH_BOX && node >= n_nodes) phase = shadow ? PH_SHADE : PH_HIT;
            }
        } else if (sel == PH_TRI) {
            if (phase == PH_TRI) {
                const int i = ti;
                const float4 q0 = sc.tri[3 * i + 0], q1 = sc.tri[3 * i + 1], q2 = sc.tri[3 * i + 2];
                const f3 A = mk(q0.x, q0.y, q0.z), e1 = mk(q0.w, q1.x, q1.y), e2 = mk(q1.z, q1.w, q2.x);
                const f3 N = mk(q2.y, q2.z, q2.w);
                const float det = dot(u, N);               // moller_trumbore, cpu:226-236
                if (det != 0) {
                    const f3 AO = A - O;
                    const f3 c = cross(AO, u);
                    const float bn = dot(e2, c);
                    const float gn = -dot(e1, c);
                    bool reject = false, pass = false;
                    if (fabsf(det) > kTiny) {
                        const float rd = __builtin_amdgcn_rcpf(det);
                        const float b = bn * rd, g = gn * rd;
                        const float eb = fabsf(b) * kRel + kAbs, eg = fabsf(g) * kRel + kAbs;
                        const float sum = b + g;
                        const float es = eb + eg + fabsf(sum) * 0x1p-22f;
                        reject = b < -eb || b > 1.f + eb || g < -eg || g > 1.f + eg || sum > 1.f + es;
                        pass = b >= eb && b <= 1.f - eb && g >= eg && g <= 1.f - eg && sum <= 1.f - es;
                    }
                    if (!reject) {
                        bool ok = pass;
                        if (!pass) {   // undecided: the literal tests on correctly rounded quotients
                            const float beta = bn / det;
                            const float gamma = gn / det;
                            ok = (0 <= beta && beta <= 1) && (0 <= gamma && gamma <= 1) && (beta + gamma <= 1);
                        }
                        if (ok) {
                            const float t = dot(AO, N) / det;
                            if (t > 0 && t > fr.tri_tmin && t < tm) { tm = t; Nb = N; many = true; tbest = i; }   // cpu:235,301
                        }
                    }
                }
                ti = i + 1;
                if (ti >= te) phase = (node >= n_nodes) ? (shadow ? PH_SHADE : PH_HIT) : PH_BOX;
            }
        } else if (sel == PH_START) {
            if (phase == PH_START) {
                rays += 1.f;
                if (STATS) wk.rays++;
                t_min = 1e9f; win = -1;                                  // cpu:546-547
                spheres_all(sc, O, u, t_min, win);
                tm = 1e9f; many = false;                                 // cpu:283
                node = 0;
                ri = ray_inv(u);
                phase = (mesh_slot >= 0 && n_nodes > 0) ? PH_BOX : (shadow ? PH_SHADE : PH_HIT);
            }
        } else if (sel == PH_HIT) {
            if (phase == PH_HIT) {
                bool mesh_won = false;
                if (many) { const int mobj = mesh_obj_of_tri(sc, tbest); if (mesh_beats_sphere(t_min, win, tm, mobj)) { t_min = tm; win = mobj; mesh_won = true; } }   // the meshes' turn in cpu:549-558
                if (win < 0) {
                    phase = PH_NEXT;                                     // miss: black (cpu:571)
                } else {
                    const f3 P = O + t_min * u;                          // cpu:560
                    f3 N;
                    if (mesh_won) N = normalize(Nb);                     // cpu:308
                    else {
                        N = normalize(P - sphere_centre_of(sc, win));    // cpu:524-525
                    }
                    const Material m = material_of(sc, win);
                    bool next_segment = true;
                    if (m.mirror) {                                      // cpu:573-579
                        O = P + fr.eps * N;
                        u = u - (2 * dot(u, N)) * N;
                    } else if (m.n_in != m.n_out) {                      // cpu:580-604
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
                    } else {                                             // cpu:605-614: shadow ray
                        Ps = P; Ns = N; sid = win;
                        const f3 Pa = P + fr.eps * N;
                        const f3 toL = L - Pa;
                        u = normalize(toL);   // = toL / sqrt(norm2(toL))                  // NORMED_VEC
                        O = Pa;
                        shadow = true;
                        next_segment = false;
                        phase = PH_START;
                    }
                    if (next_segment) {
                        d = d + 1;
                        phase = (d < fr.segs) ? PH_START : PH_NEXT;
                    }
                }
            }
        } else if (sel == PH_SHADE) {
            if (phase == PH_SHADE) {
                if (many && tm < t_min) t_min = tm;                      // (a shadow ray needs the nearest hit's value only: ties do not matter)
                const f3 Pp = O + t_min * u;                             // cpu:560 (O is P_adjusted)
                float l = 0.f;
                if (!(norm2(Pp - O) <= norm2(L - O))) {                  // cpu:615
                    const f3 wl = normalize(L - Ps);
                    const float dn = dot(Ns, wl);
                    const float mx = (dn < 0.f) ? 0.f : dn;
                    l = (float)((double)sc.intensity / (4 * PI_D * (double)norm2(L - Ps)) * (double)mx);   // cpu:623
                }
                lstack[d * kPBlock] = l;
                ids |= (uint64_t)(sid & 15) << (4 * d);
                dmask |= 1u << d;
                shadow = false;
                if (d + 1 < fr.segs) {                                   // the bounce ray (cpu:627-642)
                    const float r1 = uniform01(hs, (uint32_t)d, 0);
                    const float r2 = uniform01(hs, (uint32_t)d, 1);
                    double sn, cs;
                    rt_sincos_2pi(2 * PI_D * (double)r1, sn, cs);
                    const float s1 = rt_sqrtf(1 - r2);
                    const float x = (float)(cs * (double)s1);
                    const float y = (float)(sn * (double)s1);
                    const float zz = rt_sqrtf(r2);
                    // T1 = normalize((-Ny, Nx, 0)) if Nx != 0 && Ny != 0 else normalize((-Nz, 0, Nx)) (cpu:634-638): two quotients, the third component is +0 / n
                    const bool t1a = Ns.y != 0 && Ns.x != 0;
                    float t1p, t1q, t1z;
                    normalize_pq0(t1a ? -Ns.y : -Ns.z, Ns.x, t1p, t1q, t1z);
                    const f3 T1 = t1a ? mk(t1p, t1q, t1z) : mk(t1p, t1z, t1q);
                    const f3 T2 = cross(Ns, T1);
                    u = x * T1 + y * T2 + zz * Ns;
                    refr = 1.f;                                          // O stays P_adjusted
                    d = d + 1;
                    phase = PH_START;
                } else {
                    d = d + 1;
                    phase = PH_NEXT;
                }
            }
        } else {   // PH_NEXT
            if (phase == PH_NEXT) {
                if (path_open) {   // fold the finished path back to front (cpu:642-644) and accumulate (cpu:711)
                    f3 ans = mk(0, 0, 0);
                    const int nseg = d < fr.segs ? d : fr.segs;
                    for (int k = nseg - 1; k >= 0; --k) {
                        if (dmask & (1u << k)) {
                            const Material m = material_of(sc, (int)((ids >> (4 * k)) & 15));
                            const float l = lstack[k * kPBlock];
                            const f3 alb = mk(m.ar, m.ag, m.ab);
                            ans = (l * alb) / PI_F + alb * ans;
                        }
                    }
                    total = total + ans;
                    samp = samp + 1;
                    path_open = false;
                }
            }
            // pixels: lanes whose pixel is complete (or that have none yet) draw the next slots together
            const bool want = phase == PH_NEXT && (!have_pixel || samp >= fr.spp);
            if (phase == PH_NEXT && have_pixel && samp >= fr.spp) {
                const f3 avg = total / (float)fr.spp;                    // cpu:713
                fr.out[out_index(fr, lrow, px)] = make_float4(avg.x, avg.y, avg.z, rays);
                have_pixel = false;
            }
            const unsigned long long wm = __ballot(want);
            if (wm) {
                unsigned int base = 0;
                const int leader = __ffsll((long long)wm) - 1;
                if (lane == leader) base = atomicAdd(pf.queue, (unsigned int)__popcll(wm));
                base = __shfl(base, leader, 64);
                if (want) {
                    const unsigned int slot = base + (unsigned int)__popcll(wm & ((1ull << lane) - 1ull));
                    if (slot >= pf.n_slots) {
                        phase = PH_DONE;
                    } else {
                        const unsigned int tile = slot >> 6, p = slot & 63u;
                        px = (int)(tile % (unsigned)pf.tiles_x) * 8 + (int)(p & 7u);
                        lrow = (int)(tile / (unsigned)pf.tiles_x) * 8 + (int)(p >> 3);
                        if (px < fr.W && lrow < fr.n_rows) {
                            have_pixel = true;
                            samp = 0; total = mk(0, 0, 0); rays = 0.f;
                            const int row = fr.row0 + (lrow / fr.tile_rows) * fr.tile_rows * fr.tile_step + (lrow % fr.tile_rows);
                            hp = mix32(((uint32_t)row * (uint32_t)fr.W + (uint32_t)px) ^ mix32(fr.seed));
                        }   // else: a slot of an edge tile outside the image; draw again next time
                    }
                }
            }
            if (phase == PH_NEXT && have_pixel && samp < fr.spp) {       // camera ray of sample `samp` (cpu:699-709)
                const int row = fr.row0 + (lrow / fr.tile_rows) * fr.tile_rows * fr.tile_step + (lrow % fr.tile_rows);
                const f3 uc = mk((float)((double)((float)px - (float)fr.W / 2) + 0.5),
                                 (float)((double)((float)fr.H / 2 - (float)row) - 0.5), fr.z);
                hs = mix32(hp ^ ((uint32_t)samp * 0x9E3779B1U));
                f3 uu = uc;
                if (fr.sigma != 0.f) {
                    const float r1 = uniform01(hs, 0, 2), r2 = uniform01(hs, 0, 3);
                    const float bm = fr.sigma * rt_sqrtf(-2 * logf(r1));
                    double sn, cs;
                    rt_sincos_2pi(2 * PI_D * (double)r2, sn, cs);
                    uu = uc + mk((float)((double)bm * cs), (float)((double)bm * sn), 0.f);
                }
                u = normalize(uu);
                O = Cam;
                d = 0; refr = 1.f; ids = 0; dmask = 0; shadow = false;
                path_open = true;
                phase = (fr.segs > 0) ? PH_START : PH_NEXT;
            }
        }
    }
    if (STATS) {
        const uint32_t r = wave_sum(wk.rays), b = wave_sum(wk.box), n = wave_sum(wk.nodes), t = wave_sum(wk.tris);
        if (lane == 0) {
            atomicAdd(&fr.work[0], (unsigned long long)r); atomicAdd(&fr.work[1], (unsigned long long)b);
            atomicAdd(&fr.work[2], (unsigned long long)n); atomicAdd(&fr.work[3], (unsigned long long)t);
        }
    }
}

}  // namespace rtk
