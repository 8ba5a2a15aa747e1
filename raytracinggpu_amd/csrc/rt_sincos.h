// rt_sincos.h -- binary64 sin and cos of x = 2 * PI * r1, r1 = k * 2^-24 a binary32 uniform in (0, 1]: the bounce direction
// of Scene::getColor (cpu_launcher.cpp:630-631: `cos(2*PI*r1)`, `sin(2*PI*r1)` with the double literal PI).
//
// The reference calls glibc's cos / sin (correctly rounded in all but astronomically rare cases) on x in (0, 2 pi].  The ROCm
// device library's sincos is a general routine (huge arguments, inf / nan, ~100 binary64 instructions).  This one does what
// the argument range allows: quadrant n = rint(x * 2/pi) in 0..4, Cody-Waite reduction y = x - n * pi/2 with pi/2 in three
// 33-bit pieces (the products n * piece are exact), a second / third piece only when the first subtraction cancelled, then
// the classic minimax kernels for |y| <= pi/4 (error < 1 ulp).  Plain +, -, * in the written order (the translation unit is
// compiled without contraction), so the host build of this header computes bit for bit what the device does:
// tools/check_sincos.cpp compares it with glibc over all 2^24 arguments.
//
// Effect on parity: a binary64 result 1 ulp from glibc's changes the binary32 product (float)(cos * sqrt(1 - r2)) with
// probability ~2^-29; the test suite's bound (L-inf <= 1e-4, > 99.9 % of the channels bit-identical) is unaffected.
#pragma once

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtk {

RT_HD double rt_ksin(double x, double y) {          // sin(x + y), |x| <= pi/4, y the tail of x
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

RT_HD double rt_kcos(double x, double y) {          // cos(x + y), |x| <= pi/4
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double ax = x < 0 ? -x : x;
    if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
    const double qx = ax > 0.78125 ? 0.28125 : (double)(float)(ax * 0.25);   // x/4 with a short significand, so that 1 - qx is exact (the classic code clears the low word)
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

// sn = sin(x), cs = cos(x) for 0 < x <= 2 pi (+ a rounding)
RT_HD void rt_sincos_2pi(double x, double &sn, double &cs) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;    // first 33 bits of pi/2, pi/2 - pio2_1
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;    // second 33 bits, pi/2 - (pio2_1 + pio2_2)
    const double pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;    // third 33 bits, the rest
    const int n = (int)(x * invpio2 + 0.5);
    const double fn = (double)n;
    double r = x - fn * pio2_1;                      // exact: fn * pio2_1 has at most 36 bits
    double w = fn * pio2_1t;
    double y0 = r - w;
    // cancellation: |y0| far below |x| means the 33 + 53 bits used so far are not enough
    const double ay = y0 < 0 ? -y0 : y0;
    if (ay < x * 1.52587890625e-05) {               // 2^-16
        const double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        y0 = r - w;
        const double ay2 = y0 < 0 ? -y0 : y0;
        if (ay2 < x * 1.7763568394002505e-15) {      // 2^-49
            const double t2 = r;
            w = fn * pio2_3;
            r = t2 - w;
            w = fn * pio2_3t - ((t2 - r) - w);
            y0 = r - w;
        }
    }
    const double y1 = (r - y0) - w;
    const double s = rt_ksin(y0, y1), c = rt_kcos(y0, y1);
    switch (n & 3) {
        case 0: sn = s; cs = c; break;
        case 1: sn = c; cs = -s; break;
        case 2: sn = -s; cs = -c; break;
        default: sn = -c; cs = s; break;
    }
}

}  // namespace rtk
