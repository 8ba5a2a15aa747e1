// launch_gap.hip -- what a kernel boundary costs on one stream: N dependent launches of (a) an empty kernel, (b) a kernel that dirties
// `mb` MB (every launch ends with the write-back of the XCDs' L2s and the next starts behind it), timed with HIP events.
// build: hipcc -O2 --offload-arch=gfx950 -o launch_gap launch_gap.hip ; run: ./launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}
__global__ void k_write(float4 *p, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_float4(v, v, v, v);
}
__global__ void k_spin(long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 2000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("{\"kind\": \"empty kernel, 1 workgroup\", \"us_per_dependent_launch\": %.3f}\n", ms * 1e3 / N);
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(4096), dim3(256), 0, s);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("{\"kind\": \"empty kernel, 4096 workgroups of 256\", \"us_per_dependent_launch\": %.3f}\n", ms * 1e3 / N);
    // a fixed-length kernel (every workgroup spins a fixed number of clock ticks): launch-to-launch time minus the spin = the boundary
    for (int wgs : {256, 1024, 4096}) {
        const long long cyc = 2000;   // a short fixed-length kernel (about 1 us of clock64 ticks)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a, s));
            for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, s, cyc);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("{\"kind\": \"short spin kernel, %d workgroups\", \"us_per_dependent_launch\": %.3f}\n", wgs, ms * 1e3 / 500);
    }
    for (int mb : {16, 64, 256}) {
        const size_t n = (size_t)mb * 1024 * 1024 / 16;
        float4 *p = nullptr;
        CK(hipMalloc(&p, n * 16));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a, s));
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_write, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, (float)i);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("{\"kind\": \"kernel writing %d MB\", \"us_per_dependent_launch\": %.3f, \"us_at_5TBps\": %.3f}\n", mb, ms * 1e3 / 200, mb * 1.048576 / 5.0);
        CK(hipFree(p));
    }
    return 0;
}
