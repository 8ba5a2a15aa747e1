"""The any-hit bound of a shadow ray (raytracinggpu_amd/csrc/rt_wavefront.hip.h: wf_anyhit_bound), restated in numpy binary32 and checked by brute force on the CPU.

cpu_launcher.cpp:615 compares |P' - Pa|^2 with |L - Pa|^2 for P' = Pa + t u, u = (L - Pa) / |L - Pa|, every operation one binary32 rounding.  The kernels rely on two facts:
  (1) the left side is monotone in t (so the comparison holds for the nearest hit iff it holds for some accepted hit), and
  (2) every t at or below  b = fma(nl, 1 - 2^-15, -2^-20 (|Pa.x| + |Pa.y| + |Pa.z|))  passes it.
Neither is a property of this repo's code alone -- they are claims about IEEE arithmetic -- so they are tested here without a GPU, on tens of millions of samples that
include the adversarial ones: t exactly at the bound, origins a thousand units from the axes' origin with the light a hair away, lights far away, axis-parallel rays."""
import numpy as np

f32 = np.float32


def _norm2(v):
    return (v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]      # Vector::norm2: ((x*x + y*y) + z*z), each a binary32 rounding


def _setup(rng, n):
    scale_p = f32(10.0) ** rng.uniform(-2, 3.2, (n, 1)).astype(f32)                  # |Pa| from 0.01 to ~1600 (the walls' centres sit at 1000)
    pa = (rng.standard_normal((n, 3)).astype(f32) * scale_p).astype(f32)
    dist = f32(10.0) ** rng.uniform(-3, 4, (n, 1)).astype(f32)                        # the light 0.001 ... 10 000 away
    dirs = rng.standard_normal((n, 3)).astype(f32)
    dirs[: n // 8, rng.integers(0, 3)] = 0                                            # some rays in a coordinate plane
    dirs[n // 8: n // 6] = np.eye(3, dtype=f32)[rng.integers(0, 3, n // 6 - n // 8)]  # some along an axis
    dirs /= np.maximum(np.sqrt((dirs.astype(np.float64) ** 2).sum(-1, keepdims=True)), 1e-30).astype(f32)
    light = (pa + dirs * dist).astype(f32)
    to_l = (light - pa).astype(f32)                                                    # cpu:613
    d2 = _norm2(to_l).astype(f32)
    nl = np.sqrt(d2).astype(f32)                                                       # numpy's binary32 sqrt is correctly rounded, as rt_sqrtf is
    ok = nl > 0
    u = (to_l / np.where(ok, nl, 1)[:, None]).astype(f32)                              # NORMED_VEC: three divisions by the norm
    absum = ((np.abs(pa[:, 0]) + np.abs(pa[:, 1])).astype(f32) + np.abs(pa[:, 2])).astype(f32)
    b = (nl.astype(np.float64) * np.float64(f32(1) - f32(2.0 ** -15)) + (-(f32(2.0 ** -20) * absum).astype(f32)).astype(np.float64)).astype(f32)   # fmaf: one rounding
    b = np.where((nl > f32(1e-12)) & (nl < f32(1e30)), b, -np.inf).astype(f32)
    return pa, u, d2, b, ok


def _lhs(pa, u, t):
    p = (pa + (t[:, None] * u).astype(f32)).astype(f32)                                # cpu:560  P = O + t u
    return _norm2((p - pa).astype(f32)).astype(f32)                                    # cpu:615  (P' - P_adjusted).norm2()


def test_every_t_at_or_below_the_bound_passes_the_comparison():
    rng = np.random.default_rng(615)
    worst = np.inf
    for _ in range(12):
        pa, u, d2, b, ok = _setup(rng, 2_000_000)
        use = ok & (b > 0)
        pa, u, d2, b = pa[use], u[use], d2[use], b[use]
        for t in (b, np.nextafter(b, f32(0)), (b * f32(0.999)).astype(f32), (b * rng.uniform(0, 1, b.shape).astype(f32)).astype(f32)):
            lhs = _lhs(pa, u, t.astype(f32))
            assert (lhs <= d2).all()
        worst = min(worst, float(((np.sqrt(d2.astype(np.float64)) - np.sqrt(_lhs(pa, u, b).astype(np.float64))) / np.sqrt(d2.astype(np.float64))).min()))
    assert worst > 2.0 ** -17                                                          # the margin the header promises (2^-16 of the distance), not a near miss


def test_the_comparison_is_monotone_in_t():
    rng = np.random.default_rng(616)
    for _ in range(6):
        pa, u, d2, b, ok = _setup(rng, 2_000_000)
        pa, u = pa[ok], u[ok]
        t1 = (f32(10.0) ** rng.uniform(-4, 4, len(pa)).astype(f32)).astype(f32)
        for t2 in (np.nextafter(t1, f32(np.inf)), (t1 * f32(1.0000002)).astype(f32), (t1 * rng.uniform(1, 3, len(pa)).astype(f32)).astype(f32)):
            assert (_lhs(pa, u, t1) <= _lhs(pa, u, t2.astype(f32))).all()
        assert (_lhs(pa, u, np.zeros(len(pa), f32)) == 0).all()                        # t = +0: P' = Pa exactly
