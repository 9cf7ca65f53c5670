"""The integer-exponent power of the Phong highlight (`powi_dd`, csrc/rtk_shading.h) against exact arithmetic.

The reference raises the highlight's cosine to `material.shininess` with `f64::powf`
(src/scene/lighting.rs:143-147), i.e. the C library's pow, which returns the correctly rounded power
in all but a few cases per thousand.  For integer-valued exponents the kernels square and multiply in
double-double arithmetic instead; this file restates that sequence operation by operation (every fused
multiply-add evaluated exactly and rounded once, as the hardware does) and checks that its result is
the correctly rounded power.
"""
import math
import random
from fractions import Fraction


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def powi_dd(x, n):
    """x ** n for an integer n >= 1: left-to-right binary powering on an unevaluated sum h + l."""
    top = n.bit_length() - 1
    h, l = x, 0.0
    for b in range(top - 1, -1, -1):
        p = h * h
        e = fma(h, h, -p)
        e = fma(h + h, l, e)
        h, l = p, e
        m = x if (n >> b) & 1 else 1.0
        p = h * m
        e = fma(h, m, -p)
        e = fma(l, m, e)
        h, l = p, e
    return h + l


def exact_power_rounded(x, n):
    return float(Fraction(x) ** n)  # Fraction -> float rounds to nearest even


def test_powi_dd_is_the_correctly_rounded_power():
    rng = random.Random(20250)
    worst = 0
    cases = 0
    for n in (1, 2, 3, 5, 10, 31, 50, 100, 127, 128, 200, 255, 300, 1000, 4095):
        for _ in range(120):
            # cosines of the highlight: mostly close to 1, some small
            x = 1.0 - 10.0 ** rng.uniform(-9.0, -0.3) if rng.random() < 0.8 else rng.uniform(0.0, 1.0)
            want = exact_power_rounded(x, n)
            if want < 2.0 ** -900:  # below that the error terms go subnormal; no such highlight reaches a pixel
                continue
            got = powi_dd(x, n)
            cases += 1
            if got != want:
                worst = max(worst, abs(got - want) / math.ulp(want))
    assert cases > 1000
    assert worst == 0, f"off by {worst} ulp"


def test_powi_dd_edge_values():
    assert powi_dd(0.0, 200) == 0.0
    assert powi_dd(1.0, 200) == 1.0
    assert powi_dd(0.5, 1) == 0.5
    assert powi_dd(0.5, 200) == 2.0 ** -200
    x = 1.0 + 2.0 ** -52  # a cosine a rounding above one
    assert powi_dd(x, 200) == exact_power_rounded(x, 200)
    assert powi_dd(1e-3, 200) == 0.0  # underflows like the library's pow
