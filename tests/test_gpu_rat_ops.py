"""GPU parity of the device's canonical rational forms (scalar.hip.h: fma_canon, div_canon -- binary gcds, exact
quotients and remainders through fp64 reciprocals) against the oracle's two-operation restatement of
src/com/rational.cpp:273-397, on operands chosen to stress every divider: common factors of every size up to 2^30,
powers of two and of five (the `appro` denominators), values at the appro thresholds, sums that cancel."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IMAX = 0x7FFFFFFF
MUL, DIV, ADD = 0, 1, 2


def canon(n, d):
    if n == 0:
        return (0, 1)
    g = math.gcd(abs(n), d)
    n, d = n // g, d // g
    if abs(n) >= IMAX or d >= IMAX:
        return None
    return (n, d)


def draw(rng, n):
    """Canonical (num, den) pairs in size classes, many sharing large factors in their denominators."""
    factors = [1, 2, 3, 5, 7, 64, 625, 15625, 1 << 20, 1000000, 65537, 999983, 1 << 29, (1 << 30) - 35, 715827883, 1073741827]
    caps = [4, 100, 10000, 1000000, 1 << 24, (IMAX >> 2) + 3, IMAX - 2]
    out = []
    while len(out) < n:
        f = factors[int(rng.integers(len(factors)))]
        cap = caps[int(rng.integers(len(caps)))]
        d = f * int(rng.integers(1, max(2, min(cap, (IMAX - 2) // f) + 1)))
        num = int(rng.integers(-cap, cap + 1))
        if rng.random() < 0.15:
            d = 1
        if rng.random() < 0.05:
            num = int(rng.choice([IMAX - 2, -(IMAX - 2), (IMAX >> 2), (IMAX >> 2) - 1, 1, -1, 0]))
        c = canon(num, d)
        if c is not None:
            out.append(c)
    return out


def test_canonical_forms_equal_the_reference_operations(ctx, port):
    from xpoly_amd._capi import lib
    rng = np.random.default_rng(20261003)
    n = 60000
    a, k, e = draw(rng, n), draw(rng, n), draw(rng, n)
    # a share of sums that cancel exactly or nearly: a = -(k * e) +- small
    for i in range(0, n, 17):
        p = port.rat_op(MUL, k[i], e[i])
        if p[1] > 0 and abs(p[0]) < IMAX - 4 and p[1] < IMAX:
            c = canon(-p[0] + int(rng.integers(-1, 2)), p[1])
            if c is not None:
                a[i] = c
    A, K, E = (np.array(v, dtype=np.int32) for v in (a, k, e))
    fma = np.empty_like(A)
    div = np.empty_like(A)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    ctx.check(lib().xpg_test_canon_ops_rat32(ctx._h, C.c_int(n), p(A), p(K), p(E), p(fma), p(div)), "xpg_test_canon_ops_rat32")
    appro0 = port.appro_count()
    bad = 0
    for i in range(n):
        want = port.rat_op(ADD, a[i], port.rat_op(MUL, k[i], e[i]))
        if tuple(fma[i]) != want:
            bad += 1
            assert bad < 5, "fma_canon(%s, %s, %s) = %s, reference %s" % (a[i], k[i], e[i], tuple(fma[i]), want)
        if k[i][0] != 0:
            wd = port.rat_op(DIV, a[i], k[i])
            assert tuple(div[i]) == wd, "div_canon(%s, %s) = %s, reference %s" % (a[i], k[i], tuple(div[i]), wd)
    assert bad == 0
    assert port.appro_count() - appro0 > 1000          # the operands do reach the float32 rescue


def test_canonical_forms_refuse_other_operands(ctx):
    from xpoly_amd._capi import lib
    A = np.array([[2, 4]], dtype=np.int32)              # not in lowest terms
    K = np.array([[1, 1]], dtype=np.int32)
    out = np.empty_like(A)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    assert lib().xpg_test_canon_ops_rat32(ctx._h, C.c_int(1), p(A), p(K), p(K), p(out), None) != 0


def test_generic_forms_equal_the_reference_operations(ctx, port):
    """mul / add / div on arbitrary (num, den) pairs -- small and large numerators, denominators that are zero,
    negative or not coprime (what MIP's substituted nodes and unreduced inputs hold) -- through the device's generic
    forms (32-bit gcd and fp64 quotients when both sides of a result fit 32 bits, 64-bit binary gcd otherwise)."""
    from xpoly_amd._capi import lib
    rng = np.random.default_rng(77)
    n = 60000
    caps = [1, 3, 10, 1000, 1000000, 0x3FFFFFFF, 0x7FFFFFFF]

    def draw_any():
        out = np.empty((n, 2), dtype=np.int64)
        for i in range(n):
            cn, cd = caps[int(rng.integers(7))], caps[int(rng.integers(7))]
            num, den = int(rng.integers(-cn, cn + 1)), int(rng.integers(-cd, cd + 1))
            r = rng.random()
            if r < 0.25:
                den = 1
            elif r < 0.34:
                den = 0
            out[i] = (num, den)
        return out.astype(np.int32)
    A, B = draw_any(), draw_any()
    outs = [np.empty_like(A) for _ in range(3)]
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    ctx.check(lib().xpg_test_any_ops_rat32(ctx._h, C.c_int(n), p(A), p(B), p(outs[0]), p(outs[1]), p(outs[2])), "xpg_test_any_ops_rat32")
    for i in range(n):
        a, b = (int(A[i, 0]), int(A[i, 1])), (int(B[i, 0]), int(B[i, 1]))
        for op, got in ((MUL, outs[0]), (ADD, outs[1]), (DIV, outs[2])):
            want = port.rat_op(op, a, b)
            assert tuple(got[i]) == want, "op %d on %s, %s: device %s, reference %s" % (op, a, b, tuple(got[i]), want)
