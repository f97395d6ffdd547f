"""Arithmetic identities the kernels rely on where they replace a reference expression by a cheaper one, checked on the host in f64
(numpy's sqrt and division are correctly rounded, like the engine's): not about any device, only about the mathematics."""
import numpy as np


def test_square_comparison_equals_sqrt_comparison_for_float32_operands():
    """amp_scalefac_bands (lamejs_amd/csrc/k_quant.h): for Float32 d >= 0 and M > 1, d < sqrt(M) [f64, correctly rounded] <=> d * d < M [exact in f64]."""
    rng = np.random.default_rng(3)
    M = np.concatenate([rng.uniform(1.0, 4.0, 400000), 10.0 ** rng.uniform(0, 12, 400000)]).astype(np.float32)
    M = M[M > 1.0]
    root = np.sqrt(M.astype(np.float64))
    near = root.astype(np.float32)                                   # the Float32 values around the root: where the two comparisons could part
    for d in (near, np.nextafter(near, np.float32(0)), np.nextafter(near, np.float32(np.inf)), np.nextafter(np.nextafter(near, np.float32(0)), np.float32(0)),
              (rng.uniform(0, 2, len(M)) * root).astype(np.float32)):
        d64 = d.astype(np.float64)
        assert np.array_equal(d64 < root, d64 * d64 < M.astype(np.float64))
    # perfect squares: d * d == M exactly
    d = rng.integers(2, 4000, 100000).astype(np.float32) / np.float32(8)
    M2 = (d.astype(np.float64) ** 2).astype(np.float32)
    ok = M2.astype(np.float64) == d.astype(np.float64) ** 2
    d64, M64 = d[ok].astype(np.float64), M2[ok].astype(np.float64)
    assert np.array_equal(d64 < np.sqrt(M64), d64 * d64 < M64)


def test_division_through_reciprocal_is_the_division():
    """calc_noise (lhip_math.h div_by_f32): q0 = a * rb, r = fma(-b, q0, a), q = fma(r, rb, q0) with rb = RN(1 / b) is RN(a / b) for a Float32 b.
    numpy has no fma: the residual is formed with exact integer arithmetic on the operands' significands instead."""
    from fractions import Fraction
    rng = np.random.default_rng(4)
    a = np.concatenate([10.0 ** rng.uniform(-30, 20, 3000), rng.uniform(0, 4, 1000)])
    b = np.concatenate([10.0 ** rng.uniform(-40, 30, 3000), rng.uniform(0.5, 2, 1000)]).astype(np.float32).astype(np.float64)
    rb = 1.0 / b
    q0 = a * rb
    for ai, bi, rbi, q0i in zip(a, b, rb, q0):
        r = float(Fraction(ai) - Fraction(bi) * Fraction(q0i))       # exact residual; representable (the proof in lhip_math.h), so float() is exact
        assert Fraction(r) == Fraction(ai) - Fraction(bi) * Fraction(q0i)
        q = float(Fraction(q0i) + Fraction(r) * Fraction(rbi))       # one rounding of the exact sum: what the second fma does
        assert q == ai / bi, (ai, bi)
