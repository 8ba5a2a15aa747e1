// l1_rate.hip -- how many cycles the vector L1 of one CU needs per wave64 global_load_dwordx4 that HITS it, by access shape:
// (a) 64 lanes x 16 B contiguous (16 cache lines of 64 B), (b) every lane in its own 64-byte line (64 lines), (c) the work-stack
// kernels' shape: four loads of one lane to the four 16-byte pieces of ITS line (4 instructions x 64 lines).
// The table stays L1 resident (16 KiB per workgroup region, one workgroup per CU; 4 or 8 or 16 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 400;

template <int SHAPE> __global__ void k(const f4 *tab, float *out, int lines) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const f4 *base = tab + (size_t)blockIdx.x * lines * 4;              // this workgroup's region: `lines` 64-byte lines
    f4 acc = {0, 0, 0, 0};
    unsigned int h = lane * 2654435761u + w * 40503u;
    for (int it = 0; it < kIters; ++it) {
        h = h * 1664525u + 1013904223u;
        if (SHAPE == 0) {                                               // contiguous: lane L reads 16 B at (start + L): 16 lines per instruction
            const unsigned int start = ((h >> 8) % (unsigned)(lines / 16)) * 64u;   // wave-uniform-ish start per lane differs... use lane-independent part
            const unsigned int s = __builtin_amdgcn_readfirstlane(start);
            for (int j = 0; j < 4; ++j) { const f4 v = base[(s + lane + 64u * j) % (unsigned)(lines * 4)]; acc += v; }
        } else if (SHAPE == 1) {                                        // scattered: lane L reads piece 0 of a random line: 64 lines per instruction
            for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; const f4 v = base[((h >> 8) % (unsigned)lines) * 4]; acc += v; }
        } else {                                                        // the kernels' shape: the four pieces of ONE random line per lane
            const unsigned int line = (h >> 8) % (unsigned)lines;
            for (int j = 0; j < 4; ++j) { const f4 v = base[line * 4 + j]; acc += v; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int SHAPE> void run(const char *name, const f4 *tab, float *out, int cus, int lines) {
    for (int waves : {4, 8, 16}) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<SHAPE>, dim3(cus), dim3(64 * waves), 0, 0, tab, out, lines);
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(cus), dim3(64 * waves), 0, 0, tab, out, lines);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double loads = (double)kIters * 4 * waves;               // wave-instructions per CU
        printf("{\"shape\": \"%s\", \"waves_per_cu\": %d, \"lines_resident\": %d, \"ns_per_wave_load_per_cu\": %.2f}\n", name, waves, lines, ms / 3 * 1e6 / loads);
    }
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    for (int lines : {256, 2048}) {                                      // 16 KiB (L1 resident) and 128 KiB (L2) per workgroup
        f4 *tab; float *out;
        CHECK(hipMalloc(&tab, (size_t)cus * lines * 64)); CHECK(hipMemset(tab, 0, (size_t)cus * lines * 64));
        CHECK(hipMalloc(&out, (size_t)cus * 1024 * 4));
        run<0>("contiguous: 16 lines per instruction", tab, out, cus, lines);
        run<1>("scattered: 64 lines per instruction", tab, out, cus, lines);
        run<2>("four pieces of one line per lane (4 instructions, 64 lines each)", tab, out, cus, lines);
        CHECK(hipFree(tab)); CHECK(hipFree(out));
    }
    return 0;
}
