// hip_init.hip -- what a process pays the ROCm runtime before its first kernel runs: an EMPTY program's share of rt_launcher's wall time (tools/launcher_timing.py).
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/hip_init tools/ubench/hip_init.hip     run on the GPU box; prints one JSON line
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(int *p) { if (threadIdx.x == 0) *p = 1; }
static double ms(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main() {
    auto t = std::chrono::steady_clock::now();
    int n = 0;
    hipGetDeviceCount(&n);
    const double t_count = ms(t); t = std::chrono::steady_clock::now();
    hipSetDevice(0);
    int *p = nullptr;
    hipMalloc(&p, 4);
    const double t_malloc = ms(t); t = std::chrono::steady_clock::now();
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const double t_stream = ms(t); t = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, p);
    hipStreamSynchronize(s);
    const double t_launch = ms(t); t = std::chrono::steady_clock::now();
    int h = 0; hipMemcpy(&h, p, 4, hipMemcpyDeviceToHost);
    const double t_copy = ms(t);
    printf("{\"devices\": %d, \"hipGetDeviceCount_ms\": %.2f, \"first_hipMalloc_ms\": %.2f, \"stream_create_ms\": %.2f, \"first_launch_and_sync_ms\": %.2f, \"copy_ms\": %.2f, \"total_ms\": %.2f, \"ok\": %d}\n",
           n, t_count, t_malloc, t_stream, t_launch, t_copy, t_count + t_malloc + t_stream + t_launch + t_copy, h);
    return 0;
}
