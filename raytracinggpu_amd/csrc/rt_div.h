// rt_div.h -- correctly rounded binary32 quotients of several numerators by ONE denominator through one reciprocal.
//
// Vector::normalize (cpu_launcher.cpp:58-63) divides three components by the same sqrt(norm2); the renderer normalises four to
// five vectors per path and launch.  A correctly rounded division costs ~11 instructions on gfx950 (v_div_scale x2, v_rcp, five
// fma / mul, v_div_fmas, v_div_fixup; the scale / fixup / rcp ones at half or quarter rate) and the compiler cannot share any of
// it between the three: v_div_scale looks at numerator AND denominator.  Here the denominator's part -- r = rcp(d), one Newton
// step r1 = r + r (1 - d r) -- is done once and each quotient is the remaining five full-rate instructions of the SAME sequence
// the compiler emits (AMDGPU's IEEE-accurate fdiv expansion):
//       q = n r1;   q1 = q + r1 (n - d q);   result = q1 + r1 (n - d q1)
// which is that expansion with the scale factors equal to 1 and the fix-up a no-op -- true whenever v_div_scale would not scale and
// v_div_fixup would not intervene: d and every |n| in [2^-60, 2^60] (no denormal, zero, inf or NaN anywhere, quotient in
// [2^-120, 2^120]).  Outside that range (a zero component, a denormal, an overflowed norm) the caller takes the literal division
// behind a wave-uniform branch.  Bit-exactness: tools/check_div.cpp compares the sequence with the compiler's `/` on the host
// (fma = fmaf, the reciprocal perturbed by -1 / 0 / +1 ulp: v_rcp_f32 is accurate to 1 ulp) over 10^8 operand pairs, and the
// render tests compare every frame with the oracle bit for bit.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RT_DIV_HD __host__ __device__ __forceinline__
#else
#define RT_DIV_HD inline
#endif

namespace rtk {

constexpr float kDivLo = 0x1p-60f, kDivHi = 0x1p60f;

// r = an approximation of 1 / d within 1 ulp (v_rcp_f32); returns the refined reciprocal the quotients share
RT_DIV_HD float div_refine(float d, float r) {
    const float e = fmaf(-d, r, 1.0f);
    return fmaf(e, r, r);
}
RT_DIV_HD float div_by(float n, float d, float r1) {
    const float q = n * r1;
    const float e2 = fmaf(-d, q, n);
    const float q1 = fmaf(e2, r1, q);
    const float e3 = fmaf(-d, q1, n);
    return fmaf(e3, r1, q1);
}
// the range in which the shared sequence IS the correctly rounded quotient (see above); false for NaN
RT_DIV_HD bool div_in_range(float x) { return fabsf(x) >= kDivLo && fabsf(x) <= kDivHi; }

}  // namespace rtk
