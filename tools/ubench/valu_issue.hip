// valu_issue.hip -- how many cycles one SIMD of gfx950 needs per wave64 vector instruction, by instruction kind and by waves per SIMD.
// Calibrates the denominator of bench.py's `roofline.bound = "valu_issue"`.  Build: hipcc -O3 --offload-arch=gfx950.
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
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 1) { REP8(asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n v_min_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_min_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 2) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc\n v_cmp_gt_f32 vcc, %5, %6\n v_cndmask_b32 %4, %4, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");) }
        if (KIND == 3) { REP8(asm volatile("v_and_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_lshl_add_u32 %2, %2, 1, %3\n v_add_u32 %3, %3, %0\n v_and_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_lshl_add_u32 %2, %2, 1, %3\n v_add_u32 %3, %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (KIND == 4) { REP8(asm volatile("v_mbcnt_lo_u32_b32 %0, exec_lo, %0\n v_mbcnt_hi_u32_b32 %1, exec_hi, %1\n v_mbcnt_lo_u32_b32 %2, exec_lo, %2\n v_mbcnt_hi_u32_b32 %3, exec_hi, %3\n v_mbcnt_lo_u32_b32 %0, exec_lo, %0\n v_mbcnt_hi_u32_b32 %1, exec_hi, %1\n v_mbcnt_lo_u32_b32 %2, exec_lo, %2\n v_mbcnt_hi_u32_b32 %3, exec_hi, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
        if (KIND == 5) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 6) { REP8(asm volatile("v_max3_f32 %0, |%0|, |%1|, |%2|\n v_max3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_min3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_min3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 7) { REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_mul_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_sub_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_add_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 8) { REP8(asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[20:21]\n s_add_i32 s24, s24, s25\n s_bcnt1_i32_b64 s25, s[20:21]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[20:21]\n s_add_i32 s24, s24, s25\n s_bcnt1_i32_b64 s25, s[20:21]" ::: "s20", "s21", "s22", "s23", "s24", "s25", "scc");) }
        if (KIND == 9) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_fma_f32 %1, %1, %1, %1\n s_or_b64 s[22:23], s[22:23], s[20:21]\n v_fma_f32 %2, %2, %2, %2\n s_add_i32 s24, s24, s25\n v_fma_f32 %3, %3, %3, %3\n s_bcnt1_i32_b64 s25, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20", "s21", "s22", "s23", "s24", "s25", "scc");) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
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
    run<0>("v_fma_f32", out, cyc, cus);
    run<7>("v_mul/add/sub_f32", out, cyc, cus);
    run<1>("v_min_f32", out, cyc, cus);
    run<6>("v_min3/max3_f32", out, cyc, cus);
    run<2>("v_cmp+v_cndmask", out, cyc, cus);
    run<3>("int and/or/lshl_add/add", out, cyc, cus);
    run<4>("v_mbcnt", out, cyc, cus);
    run<5>("v_rcp_f32", out, cyc, cus);
    run<8>("salu (4 kinds)", out, cyc, cus);
    run<9>("valu+salu interleaved 1:1", out, cyc, cus);
    return 0;
}
