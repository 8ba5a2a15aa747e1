// unpack_rate.hip -- issue time of the instructions that could unpack 16-bit node fields (round 4: compressed BVH nodes), per SIMD, by waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 64, kIters = 2000;

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND> __global__ void k(float *out, unsigned long long *cyc) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    unsigned int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {0.999f, 0.998f}, p5 = {1e-3f, 2e-3f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        if (KIND == 0) { REP8(asm volatile("v_cvt_f32_u32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %4, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %5, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %6, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %7, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if (KIND == 1) { REP8(asm volatile("v_fma_mix_f32 %0, %8, %1, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %9, %3, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %9, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %10, %5, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %10, %6, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %11, %7, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %11, %0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if (KIND == 2) { REP8(asm volatile("v_cvt_f32_u32 %0, %8\n v_cvt_f32_u32 %1, %9\n v_cvt_f32_u32 %2, %10\n v_cvt_f32_u32 %3, %11\n v_cvt_f32_u32 %4, %8\n v_cvt_f32_u32 %5, %9\n v_cvt_f32_u32 %6, %10\n v_cvt_f32_u32 %7, %11" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if (KIND == 3) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %9\n v_cvt_f32_ubyte2 %2, %10\n v_cvt_f32_ubyte3 %3, %11\n v_cvt_f32_ubyte0 %4, %8\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %10\n v_cvt_f32_ubyte3 %7, %11" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if (KIND == 4) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1\n v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (KIND == 5) { REP8(asm volatile("v_cvt_f32_f16_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_f16_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %2, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_f16_sdwa %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %4, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_f16_sdwa %5, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %6, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_f16_sdwa %7, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if (KIND == 6) { REP8(asm volatile("v_and_b32 %0, 0xffff, %8\n v_lshrrev_b32 %1, 16, %8\n v_and_b32 %2, 0xffff, %9\n v_lshrrev_b32 %3, 16, %9\n v_and_b32 %4, 0xffff, %10\n v_lshrrev_b32 %5, 16, %10\n v_and_b32 %6, 0xffff, %11\n v_lshrrev_b32 %7, 16, %11" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
        if (KIND == 7) { REP8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4), "v"(p5));) }
        if (KIND == 8) { REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0 * 0.f + 0.999f), "v"(a1 * 0.f + 1e-3f));) }
        if (KIND == 9) { REP8(asm volatile("v_pk_mul_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n v_pk_mul_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4), "v"(p5));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    a0 += p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(i0 + i1 + i2 + i3);
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND> void run(const char *name, float *out, unsigned long long *cyc, int cus) {
    for (int wps : {1, 2, 4, 8}) {                       // waves per SIMD: one block of 256 * wps threads per CU
        const int threads = 256 * wps > 1024 ? 1024 : 256 * wps, blocks = cus * (256 * wps / threads);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * threads / 64);
        CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
        const double insts = (double)kIters * kUnroll;                      // per wave
        // s_memtime ticks at a constant 100 MHz on gfx950; wall time gives the SIMD rate
        const double wave_insts_per_simd = insts * wps;
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_wave_inst_per_simd\": %.4f, \"memtime_ticks_per_wave\": %.0f}\n",
               name, wps, ms, ms * 1e6 / wave_insts_per_simd, avg);
    }
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * 8 * sizeof(float))); CHECK(hipMalloc(&cyc, 256 * 64 * 8));
    const int cus = p.multiProcessorCount;
    run<0>("v_cvt_f32_u32_sdwa WORD_0/1", out, cyc, cus);
    run<2>("v_cvt_f32_u32", out, cyc, cus);
    run<5>("v_cvt_f32_f16_sdwa WORD_0/1", out, cyc, cus);
    run<3>("v_cvt_f32_ubyte0..3", out, cyc, cus);
    run<1>("v_fma_mix_f32 (f16 lo/hi x f32 + f32)", out, cyc, cus);
    run<4>("v_perm_b32", out, cyc, cus);
    run<6>("v_and 0xffff / v_lshrrev 16", out, cyc, cus);
    run<8>("v_fma_f32", out, cyc, cus);
    run<7>("v_pk_fma_f32", out, cyc, cus);
    run<9>("v_pk_mul_f32 / v_pk_add_f32", out, cyc, cus);
    return 0;
}
