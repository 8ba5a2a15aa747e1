// rt_kat.hip.h -- known-answer entry points: the DEVICE primitives of the render path fed with explicit inputs.
// Included by rt_capi.hip (same library: the functions called here are the ones the render kernels inline).
//
// The reference's own Sphere::intersect (cpu_launcher.cpp:512-527), BoundingBox::intersect (cpu:146-157),
// moller_trumbore (cpu:226-236) and TriangleMesh::intersect (cpu:238-313) produced tests/golden/kat.npz; these
// kernels run the same inputs through sphere_test, slab / slab_filtered / qbox_filter, qtri_test and the two mesh
// walks, one lane per row, and report how often the error-bounded filters decided and how often the literal
// divisions ran -- so that the rare branches (zero direction components, 0/0 slabs, rays through edges and
// vertices, |det| <= 1e-30) are shown to be reached on the device, not only through images.
#pragma once
#include "rt_travq.hip.h"

namespace rtk {

__device__ __forceinline__ void kat_count(unsigned long long *cnt, bool decided, bool active) {
    const unsigned long long md = __ballot(active && decided), ml = __ballot(active && !decided);
    if ((threadIdx.x & 63) == 0) {
        if (md) atomicAdd(&cnt[0], (unsigned long long)__popcll(md));
        if (ml) atomicAdd(&cnt[1], (unsigned long long)__popcll(ml));
    }
}

// in: C[3] R O[3] u[3]; out: hit t N[3]  (N = normalize(O + t u - C), cpu:524-525)
// rt_sqrtf (rt_kernels.hip.h): one value per lane; waves hold plain and special arguments mixed, so both of its routes run
__global__ __launch_bounds__(256) void kat_sqrt_kernel(const float *__restrict__ in, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rt_sqrtf(in[i]);
}

__global__ __launch_bounds__(256) void kat_sphere_kernel(const float *__restrict__ in, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = in + 10 * (size_t)i;
    Sphere s{};
    s.cx = r[0]; s.cy = r[1]; s.cz = r[2]; s.R = r[3]; s.R2 = r[3] * r[3];
    const f3 O = mk(r[4], r[5], r[6]), u = mk(r[7], r[8], r[9]);
    float t = 0.f;
    const bool hit = sphere_test(s, O, u, t);
    f3 N = mk(0, 0, 0);
    if (hit) N = normalize(O + t * u - mk(s.cx, s.cy, s.cz));
    float *o = out + 5 * (size_t)i;
    o[0] = hit ? 1.f : 0.f; o[1] = t; o[2] = N.x; o[3] = N.y; o[4] = N.z;
}

// in: mn[3] mx[3] O[3] u[3]; out: hit.  route 0: literal slab; 1: slab_filtered (RayInv: root-box pre-test, stackless
// walks); 2: qbox_filter with the per-ray table entry of wf_path, literal slab when undecided; 3: wf_travq's centre / half-extent
// filter (cbox_filter: the box as rt_scene_upload stores it, the per-ray constants of the refill), literal slab when undecided
__global__ __launch_bounds__(256) void kat_box_kernel(const float *__restrict__ in, int n, int route, float *__restrict__ out, unsigned long long *cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    bool hit = false, decided = false;
    if (active) {
        const float *r = in + 12 * (size_t)i;
        const float4 lo = make_float4(r[0], r[1], r[2], 0), hi = make_float4(r[3], r[4], r[5], 0);
        const f3 O = mk(r[6], r[7], r[8]), u = mk(r[9], r[10], r[11]);
        if (route == 0) {
            hit = slab(lo, hi, O, u);
        } else if (route == 1) {
            hit = slab_filtered(lo, hi, O, u, ray_inv(u), decided);
        } else if (route == 2) {
            const RayBox rb = ray_box(O, u);
            const float4 A = make_float4(rb.rx, rb.ry, rb.rz, rb.safe ? rb.c0 : __builtin_inff());
            const float4 C = make_float4(O.x, O.y, O.z, u.x);
            decided = qbox_filter(lo, hi, A, C, hit);
            if (!decided) hit = slab(lo, hi, O, u);
        } else {
            // the scene-wide quantities of install_scene, for this one box: magnitudes, and whether the fast filter may run at all
            const float v[6] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z};
            bool fast = true;
            for (int a = 0; a < 3; ++a) fast = fast && v[a] <= v[a + 3] && fabsf(v[a]) < 1e8f && fabsf(v[a + 3]) < 1e8f;
            const f3 bm = mk(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y)), fmaxf(fabsf(lo.z), fabsf(hi.z)));
            const RayBoxC rb = ray_box_c(O, u, bm, fast);
            const float4 n0 = make_float4(box_centre(lo.x, hi.x), box_centre(lo.y, hi.y), box_centre(lo.z, hi.z), 0.f);
            const float4 n1 = make_float4(box_half(lo.x, hi.x), box_half(lo.y, hi.y), box_half(lo.z, hi.z), 0.f);
            bool miss;
            cbox_filter(n0, n1, make_float4(rb.rx, rb.ry, rb.rz, rb.c0), make_float4(rb.ox, rb.oy, rb.oz, 0.f), hit, miss);
            decided = hit || miss;
            if (!decided) hit = slab(lo, hi, O, u);
        }
        out[i] = hit ? 1.f : 0.f;
    }
    kat_count(cnt, decided, active);
}

// in: A[3] B[3] C[3] O[3] u[3]; out: hit t N[3]  (N = e1 x e2, unnormalised, always written: cpu:229)
__global__ __launch_bounds__(256) void kat_tri_kernel(const float *__restrict__ in, int n, float *__restrict__ out, unsigned long long *cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    int how = 0;
    if (active) {
        const float *r = in + 15 * (size_t)i;
        const f3 A = mk(r[0], r[1], r[2]), B = mk(r[3], r[4], r[5]), C = mk(r[6], r[7], r[8]);
        const f3 O = mk(r[9], r[10], r[11]), u = mk(r[12], r[13], r[14]);
        const f3 e1 = B - A, e2 = C - A, N = cross(e1, e2);               // the triangle record of rt_scene_upload / retri_kernel
        const float4 q0 = make_float4(A.x, A.y, A.z, e1.x), q1 = make_float4(e1.y, e1.z, e2.x, e2.y), q2 = make_float4(e2.z, N.x, N.y, N.z);
        float t = 0.f;
        // moller_trumbore alone accepts t > 0 (cpu:235); the leaf loop's t > 1e-4 and t < t_min belong to the traversal
        const bool hit = qtri_test(q0, q1, q2, O, u, 0.f, t, how);
        float *o = out + 5 * (size_t)i;
        o[0] = hit ? 1.f : 0.f; o[1] = t; o[2] = N.x; o[3] = N.y; o[4] = N.z;
    }
    kat_count(cnt, (how & 3) != 2, active);
}

// in: O[3] u[3]; out: hit t N[3] (N normalised, cpu:308) against the uploaded mesh.  route 0: the work-stack kernels'
// primitives (root box by slab_filtered as wf_emit_ray does, then qbox_filter / slab and qtri_test over the pre-order
// skip-pointer array, nearest = min over (bits(t) << 32 | visit rank)); route 1: mesh_intersect (stackless walk of the
// lock-step and persistent kernels).  cnt[0..1]: box tests decided / literal, cnt[2..3]: triangle tests decided / literal.
__global__ __launch_bounds__(256) void kat_mesh_kernel(const Scene sc, const float *__restrict__ in, int n, float tri_tmin, int route,
                                                       float *__restrict__ out, unsigned long long *cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    unsigned int bd = 0, bl = 0, td = 0, tl = 0;
    if (active) {
        const float *r = in + 6 * (size_t)i;
        const f3 O = mk(r[0], r[1], r[2]), u = mk(r[3], r[4], r[5]);
        bool hit = false; float t = 1e9f; f3 N = mk(0, 0, 0);
        if (route == 1) {
            Work wk;
            f3 Nr;
            int tri_; hit = mesh_intersect<false>(sc, O, u, tri_tmin, t, Nr, tri_, wk);
            if (hit) N = normalize(Nr);
        } else if (sc.n_nodes > 0) {
            bool dec;
            const bool root = slab_filtered(sc.root_lo, sc.root_hi, O, u, ray_inv(u), dec);
            if (dec) bd++; else bl++;
            unsigned long long best = WF_NOHIT;
            if (root) {
                const RayBox rb = ray_box(O, u);
                const float4 A = make_float4(rb.rx, rb.ry, rb.rz, rb.safe ? rb.c0 : __builtin_inff());
                const float4 C = make_float4(O.x, O.y, O.z, u.x);
                const int root_hiw = __float_as_int(sc.root_hi.w);
                auto leaf = [&](int first, int end) {
                    for (int k = first; k < end; ++k) {
                        const float4 *tp = sc.tri + 3 * (size_t)k;
                        float tt; int how;
                        if (qtri_test(tp[0], tp[1], tp[2], O, u, tri_tmin, tt, how)) {
                            const unsigned long long key = (unsigned long long)__float_as_uint(tt) << 32 | (unsigned int)k;
                            if (key < best) best = key;
                        }
                        if ((how & 3) == 2) tl++; else td++;
                    }
                };
                if (root_hiw >= 0) {
                    leaf(__float_as_int(sc.root_lo.w), root_hiw);
                } else {
                    for (int x = 1; x < sc.n_nodes;) {
                        const float4 lo = sc.nodes[2 * x], hi = sc.nodes[2 * x + 1];
                        const int hiw = __float_as_int(hi.w), low = __float_as_int(lo.w);
                        bool h;
                        if (qbox_filter(lo, hi, A, C, h)) bd++; else { h = slab(lo, hi, O, u); bl++; }
                        if (h && hiw >= 0) leaf(low, hiw);
                        x = (h || hiw >= 0) ? x + 1 : low;
                    }
                }
            }
            if (best != WF_NOHIT) {
                hit = true;
                t = __uint_as_float((unsigned int)(best >> 32));
                const float4 q2 = sc.tri[3 * (size_t)(unsigned int)best + 2];
                N = normalize(mk(q2.y, q2.z, q2.w));
            }
        }
        float *o = out + 5 * (size_t)i;
        o[0] = hit ? 1.f : 0.f; o[1] = t; o[2] = N.x; o[3] = N.y; o[4] = N.z;
    }
    const uint32_t s0 = wave_sum(bd), s1 = wave_sum(bl), s2 = wave_sum(td), s3 = wave_sum(tl);
    if ((threadIdx.x & 63) == 0) {
        if (s0) atomicAdd(&cnt[0], (unsigned long long)s0);
        if (s1) atomicAdd(&cnt[1], (unsigned long long)s1);
        if (s2) atomicAdd(&cnt[2], (unsigned long long)s2);
        if (s3) atomicAdd(&cnt[3], (unsigned long long)s3);
    }
}

}  // namespace rtk

namespace {

// rows of `width` floats in, `owidth` floats out, up to 4 counters back
template <typename Launch>
int kat_run(rt_ctx *ctx, const float *in, int n, int width, float *out, int owidth, rt_kat_counts *counts, Launch launch) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (n < 0 || (n > 0 && (!in || !out))) return fail(ctx, RT_ERR_INVALID, "bad KAT arguments");
    if (counts) *counts = rt_kat_counts{};
    if (n == 0) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_OWN_STREAM(ctx);
    DevBuf din, dout, dcnt;
    int rc;
    if ((rc = upload(ctx, din, in, (size_t)n * width * sizeof(float))) != RT_OK || (rc = ensure(ctx, dout, (size_t)n * owidth * sizeof(float))) != RT_OK ||
        (rc = ensure(ctx, dcnt, 4 * sizeof(unsigned long long))) != RT_OK) { din.release(); dout.release(); dcnt.release(); return rc; }
    hipError_t e = hipMemsetAsync(dcnt.p, 0, 4 * sizeof(unsigned long long), own_stream(ctx));
    if (e == hipSuccess) {
        launch(static_cast<const float *>(din.p), static_cast<float *>(dout.p), static_cast<unsigned long long *>(dcnt.p), dim3((unsigned)((n + 255) / 256)), dim3(256));
        e = hipGetLastError();
    }
    unsigned long long h[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, (size_t)n * owidth * sizeof(float), hipMemcpyDeviceToHost, own_stream(ctx));
    if (e == hipSuccess) e = hipMemcpyAsync(h, dcnt.p, sizeof(h), hipMemcpyDeviceToHost, own_stream(ctx));
    if (e == hipSuccess) e = hipStreamSynchronize(own_stream(ctx));
    din.release(); dout.release(); dcnt.release();
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "KAT launch: %s", hipGetErrorString(e));
    if (counts) { counts->n = (uint64_t)n; counts->box_decided = h[0]; counts->box_literal = h[1]; counts->tri_decided = h[2]; counts->tri_literal = h[3]; }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_kat_sqrt(rt_ctx *ctx, const float *in, int n, float *out) {
    return kat_run(ctx, in, n, 1, out, 1, nullptr, [&](const float *di, float *dres, unsigned long long *, dim3 g, dim3 b) {
        hipLaunchKernelGGL(rtk::kat_sqrt_kernel, g, b, 0, own_stream(ctx), di, n, dres);
    });
}

int rt_kat_sphere(rt_ctx *ctx, const float *in, int n, float *out) {
    return kat_run(ctx, in, n, 10, out, 5, nullptr, [&](const float *di, float *dres, unsigned long long *, dim3 g, dim3 b) {
        hipLaunchKernelGGL(rtk::kat_sphere_kernel, g, b, 0, own_stream(ctx), di, n, dres);
    });
}

int rt_kat_box(rt_ctx *ctx, const float *in, int n, int route, float *out, rt_kat_counts *counts) {
    if (route < 0 || route > 3) return fail(ctx, RT_ERR_INVALID, "route must be 0 (literal), 1 (slab_filtered), 2 (qbox_filter) or 3 (cbox_filter)");
    return kat_run(ctx, in, n, 12, out, 1, counts, [&](const float *di, float *dres, unsigned long long *dc, dim3 g, dim3 b) {
        hipLaunchKernelGGL(rtk::kat_box_kernel, g, b, 0, own_stream(ctx), di, n, route, dres, dc);
    });
}

int rt_kat_triangle(rt_ctx *ctx, const float *in, int n, float *out, rt_kat_counts *counts) {
    const int rc = kat_run(ctx, in, n, 15, out, 5, counts, [&](const float *di, float *dres, unsigned long long *dc, dim3 g, dim3 b) {
        hipLaunchKernelGGL(rtk::kat_tri_kernel, g, b, 0, own_stream(ctx), di, n, dres, dc);
    });
    if (rc == RT_OK && counts) { counts->tri_decided = counts->box_decided; counts->tri_literal = counts->box_literal; counts->box_decided = counts->box_literal = 0; }
    return rc;
}

int rt_kat_mesh(rt_ctx *ctx, const float *in, int n, float tri_tmin, int route, float *out, rt_kat_counts *counts) {
    if (ctx && !ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    if (route < 0 || route > 1) return fail(ctx, RT_ERR_INVALID, "route must be 0 (work-stack primitives) or 1 (stackless mesh_intersect)");
    return kat_run(ctx, in, n, 6, out, 5, counts, [&](const float *di, float *dres, unsigned long long *dc, dim3 g, dim3 b) {
        hipLaunchKernelGGL(rtk::kat_mesh_kernel, g, b, 0, own_stream(ctx), ctx->scene, di, n, tri_tmin, route, dres, dc);
    });
}

// FNV-1a over the device arrays the traversal kernels read: out[0] the float sibling pairs (nodesb), [1] the 16-bit fixed-point pairs (nodesh), [2] the 4-wide quads
// (nodesw: WHICH four nodes a quad holds is the surface-area DP's choice, rt_qnodes.hip.h), [3] the leaf boxes by triangle (leaflh); 0 = the array is not in use.
int rt_kat_layout_hash(rt_ctx *ctx, uint64_t out[4]) {
    if (!ctx || !out) return fail(ctx, RT_ERR_INVALID, "bad arguments");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    RT_OWN_STREAM(ctx);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const rtk::Scene &sc = ctx->scene;
    const void *src[4] = {sc.nodesb, sc.nodesh, sc.nodesw, sc.leaflh};
    const size_t bytes[4] = {2 * ((size_t)sc.n_nodes + 1) * 16, ((size_t)sc.n_nodes + 1) * 16, ((size_t)(sc.n_nodes & ~1) + 2) * 32 /* quads of the pairs c = 0, 2, .. <= n_nodes */, (size_t)sc.n_tris * 32};
    std::vector<unsigned char> h;
    for (int k = 0; k < 4; ++k) {
        out[k] = 0;
        if (!src[k] || sc.n_nodes <= 0) continue;
        h.resize(bytes[k]);
        RT_HIP(ctx, hipMemcpyAsync(h.data(), src[k], bytes[k], hipMemcpyDeviceToHost, own_stream(ctx)));
        RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
        uint64_t x = 0xcbf29ce484222325ull;
        for (unsigned char c : h) x = (x ^ c) * 0x100000001b3ull;
        out[k] = x ? x : 1;
    }
    return RT_OK;
}

}  // extern "C"
