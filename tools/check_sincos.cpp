// Build container: exhaustive comparison of rt_sincos_2pi (raytracinggpu_amd/csrc/rt_sincos.h, host build, no contraction) with
// glibc's sin / cos over every argument the renderer can produce: x = 2 * PI * r1, r1 = k * 2^-24, k = 1 .. 2^24 (cpu:628-631).
//   g++ -O2 -ffp-contract=off -o /tmp/check_sincos tools/check_sincos.cpp && /tmp/check_sincos
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../raytracinggpu_amd/csrc/rt_sincos.h"

static int64_t ulps(double a, double b) {
    int64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8);
    if (x < 0) x = INT64_MIN - x; if (y < 0) y = INT64_MIN - y;
    return x > y ? x - y : y - x;
}

int main() {
    const double PI = 3.14159265358979323846;
    long long dbl_diff = 0, max_ulp = 0, flt_diff = 0, n = 0;
    uint32_t h = 12345u;
    for (uint32_t k = 1; k <= (1u << 24); ++k) {
        const float r1 = (float)k * 0x1p-24f;
        const double x = 2 * PI * (double)r1;
        double s0 = sin(x), c0 = cos(x), s1, c1;
        rtk::rt_sincos_2pi(x, s1, c1);
        const int64_t us = ulps(s0, s1), uc = ulps(c0, c1);
        if (us) dbl_diff++; if (uc) dbl_diff++;
        if (us > max_ulp) max_ulp = us; if (uc > max_ulp) max_ulp = uc;
        h = h * 1664525u + 1013904223u;                                   // s1f = sqrt(1 - r2) of some other uniform r2 (cpu:630)
        const float r2 = (float)((h >> 8) + 1u) * 0x1p-24f;
        const float s1f = sqrtf(1 - r2);
        if ((float)(c0 * (double)s1f) != (float)(c1 * (double)s1f)) flt_diff++;
        if ((float)(s0 * (double)s1f) != (float)(s1 * (double)s1f)) flt_diff++;
        n += 2;
    }
    printf("%lld values: binary64 results differing from glibc %lld (max %lld ulp), binary32 products (cpu:630-631) differing %lld\n", n, dbl_diff, max_ulp, flt_diff);
    return max_ulp > 1 ? 1 : 0;
}
