// Device-side scalar semantics of xpoly's Float and Rational for gfx950.
//
// The solver must replay the reference's arithmetic exactly:
//   * Float   = one fp64 (src/com/flty.h:45-62); '*' then '+' are two roundings
//               (flty.cpp:97-116), so this translation unit is compiled with
//               -ffp-contract=off and carries the pragma below: the compiler must
//               never contract a Float product and sum (the only v_fma_f64 in these
//               kernels are the explicit ones of the integer quotients, DivFp below).
//   * Rational = int32/int32 with int64 intermediates, gcd reduction and the
//               float32 "appro" rescue (src/com/rational.cpp:163-397).
// Both are written as overloads on two POD types so that every kernel is a
// single template over the scalar.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define XPG_HD __host__ __device__ __forceinline__

// Environment switches (host side). xpg_env: the switches the product documents -- always read.
// xpg_hook: test hooks (fault injection, forced routes, debug prints) and the lab's A/B knobs -- they exist only in the
// -DXPG_TEST_HOOKS build (xpoly_amd/libxpoly_amd_hooks.so: the tests that need one load it, tools/lab runs on it); the
// product library does not look at them.
#include <stdlib.h>
inline const char * xpg_env(const char * name) { return getenv(name); }
inline const char * xpg_hook(const char * name)
{
#ifdef XPG_TEST_HOOKS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// diagnostic builds (-DXPG_TRACE): every committed pivot of workgroup 0 is printed (entering, leaving, row)
#if defined(XPG_TRACE) && defined(__HIP_DEVICE_COMPILE__)
#define XPG_TRACE_PIVOT(tag_, e_, l_, r_) do { if (blockIdx.x == 0 && blockIdx.y == 0) printf("%s: enter %d leave %d row %d\n", tag_, (int)(e_), (int)(l_), (int)(r_)); } while (0)
#else
#define XPG_TRACE_PIVOT(tag_, e_, l_, r_) do { } while (0)
#endif

namespace xpg {

// ---- fp64 --------------------------------------------------------------------
struct F64 {
    double v;
    XPG_HD F64() : v(0.0) {}
    XPG_HD explicit F64(double d) : v(d) {}
    static XPG_HD F64 from_int(int i) { return F64((double)i); }
};

// flty.cpp:41-58: not of opposite sign and |a|,|b| within 1e-17 of each other.
XPG_HD bool eq(F64 a, F64 b)
{
    const double tiny = 0.00000000000000001;
    double x = a.v, y = b.v;
    if ((x > 0 && y < 0) || (x < 0 && y > 0)) return false;
    x = x < 0 ? -x : x;
    y = y < 0 ? -y : y;
    if ((x == 0.0 && y <= tiny) || (y == 0.0 && x <= tiny)) return true;
    return (x > y ? x - y : y - x) <= tiny;
}
XPG_HD bool ne(F64 a, F64 b) { return !eq(a, b); }
XPG_HD bool lt(F64 a, F64 b) { return a.v < b.v; }
XPG_HD bool gt(F64 a, F64 b) { return a.v > b.v; }
XPG_HD bool le(F64 a, F64 b) { return a.v < b.v || eq(a, b); }
XPG_HD bool ge(F64 a, F64 b) { return a.v > b.v || eq(a, b); }
XPG_HD F64 mul(F64 a, F64 b) { return F64(a.v * b.v); }
XPG_HD F64 div(F64 a, F64 b) { return F64(a.v / b.v); }
XPG_HD F64 add(F64 a, F64 b) { return F64(a.v + b.v); }
XPG_HD F64 neg(F64 a) { return F64(-a.v); }
XPG_HD F64 sub(F64 a, F64 b) { return F64(a.v - b.v); }
XPG_HD void reduce(F64 &) {}
XPG_HD bool is_int(F64 a)   // flty.cpp:182-201
{
    const double tiny = 0.00000000000000001;
    double x = a.v < 0 ? -a.v : a.v;
    long long t = (long long)x;
    if ((x - (double)t) < tiny) return true;
    return ((double)(t + 1) - x) < tiny;
}
XPG_HD int to_int(F64 a) { return (int)a.v; }
// strict weak order used by the parallel arg-min: value first, -0 == +0
XPG_HD bool same_value(F64 a, F64 b) { return a.v == b.v; }

// ---- int32 rational ------------------------------------------------------------
struct R32 {
    int32_t num, den;
    XPG_HD R32() : num(0), den(1) {}
    XPG_HD R32(int32_t n, int32_t d) : num(n), den(d) {}
    static XPG_HD R32 from_int(int i) { return R32(i, 1); }
};

XPG_HD long long gcd64(long long x, long long y)   // rational.cpp:142-158
{
    if (x < 0) x = -x;
    if (y < 0) y = -y;
    if (x > y) { long long t = x; x = y; y = t; }
    // Fast path: once both operands fit in 32 bits the 64-bit remainder
    // (emulated on gfx950) is replaced by the 32-bit one; results are equal.
    while (x != 0) {
        if (y <= 0xFFFFFFFFLL) {
            uint32_t a = (uint32_t)x, b = (uint32_t)y;
            while (a != 0) { uint32_t t = a; a = b % a; b = t; }
            return (long long)b;
        }
        long long t = x; x = y % x; y = t;
    }
    return y;
}

XPG_HD void reduce64(long long & n, long long & d)   // rational.cpp:163-185
{
    if (n == 0) { d = 1; return; }
    long long g = gcd64(n, d);
    if (g != 1) { n /= g; d /= g; }
    if (d < 0) { d = -d; n = -n; }
}

// rational.cpp:189-226. float32 divide/multiply are IEEE (hipcc keeps
// -fhip-fp32-correctly-rounded-divide-sqrt on), the cast truncates like C.
XPG_HD void appro64(long long & n, long long & d)
{
    float q = (float)n / (float)d;
    if (q < 100.0) { q = q * 1000000.0f; n = (int)q; d = 1000000; }
    else if (q < 1000.0) { q = q * 100000.0f; n = (int)q; d = 100000; }
    else if (q < 100000.0) { q = q * 10000.0f; n = (int)q; d = 10000; }
    else if (q < 1000000.0) { q = q * 1000.0f; n = (int)q; d = 1000; }
    else if (q < 10000000.0) { q = q * 100.0f; n = (int)q; d = 100; }
    else if (q < 100000000.0) { q = q * 10.0f; n = (int)q; d = 10; }
    else if (q < 2147483647.0) { n = (int)q; d = 1; }
    else { n = 0; d = 1; }
    reduce64(n, d);
}

XPG_HD R32 squeeze(long long n, long long d)   // rational.cpp:285-309, :336-360, :373-396
{
    const long long imax = 0x7fffFFFFLL;
    if (n == d) return R32(1, 1);
    if (n == -d) return R32(-1, 1);
    if (d < 0) { n = -n; d = -d; }
    reduce64(n, d);
    long long mag = n >= 0 ? n : -n;
    if (mag >= (imax >> 2) || d >= (imax >> 2)) {
        // the reference reduces a second time here (a no-op on the value)
        if (mag >= imax || d >= imax) appro64(mag, d);
    }
    return R32((int32_t)(n < 0 ? -mag : mag), (int32_t)d);
}

XPG_HD R32 mul(R32 a, R32 b)   // rational.cpp:273-310
{
    long long n = (long long)a.num * (long long)b.num;
    if (n == 0) return R32(0, 1);
    return squeeze(n, (long long)a.den * (long long)b.den);
}
XPG_HD R32 div(R32 a, R32 b)   // rational.cpp:312-361
{
    if (a.num == 0) return R32(0, 1);
    if (a.num == a.den) return b.num < 0 ? R32(-b.den, -b.num) : R32(b.den, b.num);
    return squeeze((long long)a.num * (long long)b.den, (long long)a.den * (long long)b.num);
}
XPG_HD R32 add(R32 a, R32 b)   // rational.cpp:363-397
{
    long long n = (long long)a.num * (long long)b.den + (long long)a.den * (long long)b.num;
    if (n == 0) return R32(0, 1);
    return squeeze(n, (long long)a.den * (long long)b.den);
}
XPG_HD R32 neg(R32 a) { return R32(-a.num, a.den); }

// ---- a + k*e for CANONICAL operands ------------------------------------------------------------------
// "Canonical" = in lowest terms with den > 0, below the appro threshold: what squeeze() returns, and a fixed
// point of it -- so every tableau cell is canonical once it has been through one operation, and from the start
// when the input is (xpg::canonical below; the sweep takes this path only then).
// fma_canon(a, k, e) == add(a, mul(k, e)) bit for bit (tests/cxx/fma_canon_fuzz.cpp), with four 32-bit Euclid
// loops in place of two 64-bit ones and their 64-bit divisions:
//   * k*e: gcd(k.num, k.den) = gcd(e.num, e.den) = 1, so cancelling gcd(|k.num|, e.den) and gcd(|e.num|, k.den)
//     crosswise leaves the product in lowest terms -- the pair reduce64 (rational.cpp:163-185) arrives at;
//   * a + p with g = gcd(a.den, p.den), A = a.den/g, P = p.den/g: n' = a.num*P + p.num*A is coprime to A and to P,
//     so gcd(n', A*P*g) = gcd(n' mod g, g);
//   * the zero cases: k*0 = 0/1 (rational.cpp:276-281) and a + 0/1 = squeeze(a.num, a.den) = a.
// The tail of squeeze (second-reduce and appro thresholds, rational.cpp:294-309) is applied to the same
// lowest-terms pair the reference applies it to.
// Binary gcd (the value is the Euclidean one; the step counts of the lanes of a wave lie closer together than
// Euclid's). Both operands are kept odd, so a step is min, |x - y|, test, count-trailing-zeros, shift: five VALU
// instructions on the device (v_sad_u32 spelled out -- the compiler's max - min form and its register copy made it
// seven) against ~30 for a 32-bit remainder on gfx950.
XPG_HD uint32_t gcd32(uint32_t x, uint32_t y)
{
    if (x == 0) return y;
    if (y == 0) return x;
    const int sh = __builtin_ctz(x | y);
    x >>= __builtin_ctz(x);
    y >>= __builtin_ctz(y);
    for (;;) {
        uint32_t d;
#ifdef __HIP_DEVICE_COMPILE__
        asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(x), "v"(y));
#else
        d = x < y ? y - x : x - y;
#endif
        x = x < y ? x : y;
        if (d == 0) break;
        y = d >> __builtin_ctz(d);
    }
    return x << sh;
}
// Division without a divide (gfx950 has none: a 32-bit quotient costs ~30 instructions, a 64-bit one far more, and
// the canonical forms below would spend most of theirs there).
//  * x / g where g divides x: shift out g's factors of two and multiply by the inverse of its odd part modulo
//    2^32 (2^64) -- Newton's y <- y * (2 - o * y) doubles the correct low bits, (3 * o) ^ 2 starts with five.
//  * x mod g for a 64-bit x < 2^63: two quotient estimates through fp64 (plain IEEE operations, the same on the
//    host), each followed by an integer remainder; the first leaves |r| < 2^13 * g, the second at most one g off.
struct ExactDiv32 {
    uint32_t odd, inv; int sh;
    XPG_HD explicit ExactDiv32(uint32_t g)
    {
        sh = __builtin_ctz(g);
        odd = g >> sh;
        uint32_t y = (3u * odd) ^ 2u;
        y *= 2u - odd * y; y *= 2u - odd * y; y *= 2u - odd * y;
        inv = y;
    }
    XPG_HD uint32_t operator()(uint32_t x) const { return (x >> sh) * inv; }
    XPG_HD unsigned long long wide(unsigned long long x) const          // a 64-bit multiple of g
    {
        unsigned long long y = (unsigned long long)inv;                  // correct to 32 bits ...
        y *= 2ull - (unsigned long long)odd * y;                         // ... to 64
        return (x >> sh) * y;
    }
};
XPG_HD uint32_t mod_u64_u32(unsigned long long x, uint32_t g)
{
    const double inv = 1.0 / (double)g;
    const unsigned long long q = (unsigned long long)((double)x * inv);
    long long r = (long long)(x - q * (unsigned long long)g);            // exact modulo 2^64; |r| < 2^13 * g
    const double qf = (double)r * inv;
    long long q2 = (long long)qf;
    if ((double)q2 > qf) q2 -= 1;                                        // floor
    r -= q2 * (long long)g;
    if (r < 0) r += (long long)g;
    if (r >= (long long)g) r -= (long long)g;
    if ((unsigned long long)r >= (unsigned long long)g) return (uint32_t)(x % g);   // never (kept as the definition)
    return (uint32_t)r;
}
// The same quotients and remainders through fp64 (the canonical forms below use these; the modular-inverse forms
// above stay for the 64-bit generic operations and as the cross-check of tests/cxx/fma_canon_fuzz.cpp). gfx950's
// 32-bit integer multiplies are quarter rate and an inverse costs seven of them; fp64 FMAs are full rate:
//  * r = 1/g from v_rcp_f64 and two Newton steps (relative error ~2^-52 whatever the seed's; the host divides);
//  * x / g for a multiple x < 2^32 of g: trunc(x * r + 1/2) -- the product is within 2^-19 of the integer;
//  * floor and remainder of an integer |x| < 2^52: q = floor(x * r) is at most one off, x - q * g is exact in
//    an FMA, two conditional corrections;
//  * x mod g for x < 2^63: reduce the high word first (t < g), then t * 2^32 (exact) through one more quotient
//    estimate, whose error (< 2^12 units) leaves an exact integer remainder below 2^44; add the low word, reduce;
//  * x / g for a multiple x < 2^63 of g: floor-divide the high word, the rest (r1 * 2^32 + lo) / g is below 2^32
//    and exact, so the rounded product finds it.
// Every result is an exact integer independent of the reciprocal's last bits, so host and device agree.
XPG_HD double rcp_int(double g)
{
#ifdef __HIP_DEVICE_COMPILE__
    double r = __builtin_amdgcn_rcp(g);
    r = __builtin_fma(__builtin_fma(-g, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-g, r, 1.0), r, r);
    return r;
#else
    return 1.0 / g;
#endif
}
struct DivFp {
    double g, r;
    XPG_HD explicit DivFp(uint32_t gi) : g((double)gi), r(rcp_int((double)gi)) {}
    XPG_HD uint32_t operator()(uint32_t x) const { return (uint32_t)__builtin_fma((double)x, r, 0.5); }
    XPG_HD double floor_rem(double x, double & q) const
    {
        double qq = __builtin_floor(x * r);
        double m = __builtin_fma(-qq, g, x);
        if (m < 0.0) { m += g; qq -= 1.0; }
        if (m >= g) { m -= g; qq += 1.0; }
        q = qq;
        return m;
    }
    XPG_HD unsigned long long wide(unsigned long long x) const
    {
        const double hi = (double)(uint32_t)(x >> 32), lo = (double)(uint32_t)x;
        double q1;
        const double r1 = floor_rem(hi, q1);
        const uint32_t q0 = (uint32_t)__builtin_fma(__builtin_fma(r1, 4294967296.0, lo), r, 0.5);
        return ((unsigned long long)(uint32_t)q1 << 32) | q0;
    }
    XPG_HD uint32_t mod(unsigned long long x) const
    {
        const double hi = (double)(uint32_t)(x >> 32), lo = (double)(uint32_t)x;
        double q;
        const double xh = floor_rem(hi, q) * 4294967296.0;
        const double m = __builtin_fma(-__builtin_floor(xh * r), g, xh) + lo;
        return (uint32_t)floor_rem(m, q);
    }
};
XPG_HD bool canonical(R32 a)
{
    if (a.den <= 0) return false;
    if (a.num == 0) return a.den == 1;
    const uint32_t mag = a.num < 0 ? (uint32_t)(-(long long)a.num) : (uint32_t)a.num;
    return gcd32(mag, (uint32_t)a.den) == 1 && mag < 0x7fffFFFFu && a.den < 0x7fffFFFF;
}
// appro64 (rational.cpp:189-226) for n, d > 0: the same float32 operations, then the lowest terms of the pair they
// leave -- m < 2^31 over 10^k, k <= 6 -- which its reduce64 reaches with a Euclid loop. The denominator is
// 2^k 5^k: the common twos are a count of trailing zeros, and a multiple of five is recognised and divided in one
// multiplication by 5^-1 mod 2^32 (m * 0xCCCCCCCD <= 0x33333333 exactly when 5 divides m, and the product is m / 5).
XPG_HD void appro_lowest(long long & n, long long & d)
{
    float q = (float)n / (float)d;
    uint32_t m, t5; int k;
    if (q < 100.0) { q = q * 1000000.0f; m = (uint32_t)(int)q; k = 6; t5 = 15625u; }
    else if (q < 1000.0) { q = q * 100000.0f; m = (uint32_t)(int)q; k = 5; t5 = 3125u; }
    else if (q < 100000.0) { q = q * 10000.0f; m = (uint32_t)(int)q; k = 4; t5 = 625u; }
    else if (q < 1000000.0) { q = q * 1000.0f; m = (uint32_t)(int)q; k = 3; t5 = 125u; }
    else if (q < 10000000.0) { q = q * 100.0f; m = (uint32_t)(int)q; k = 2; t5 = 25u; }
    else if (q < 100000000.0) { q = q * 10.0f; m = (uint32_t)(int)q; k = 1; t5 = 5u; }
    else if (q < 2147483647.0) { m = (uint32_t)(int)q; k = 0; t5 = 1u; }
    else { m = 0; k = 0; t5 = 1u; }
    if (m == 0) { n = 0; d = 1; return; }
    const int z = __builtin_ctz(m);
    const int z2 = z < k ? z : k;
    m >>= z2;
    for (int i = 0; i < k; i++) {
        const uint32_t fifth = m * 0xCCCCCCCDu;
        if (fifth > 0x33333333u) break;
        m = fifth; t5 *= 0xCCCCCCCDu;
    }
    n = (long long)m; d = (long long)(t5 << (k - z2));
}
XPG_HD R32 squeeze_lowest(long long n, long long d)            // squeeze() for a pair already in lowest terms, d > 0
{
    const long long imax = 0x7fffFFFFLL;
    long long mag = n >= 0 ? n : -n;
    if (mag >= (imax >> 2) || d >= (imax >> 2)) {
        if (mag >= imax || d >= imax) appro_lowest(mag, d);
    }
    return R32((int32_t)(n < 0 ? -mag : mag), (int32_t)d);
}
XPG_HD R32 mul_canon(R32 a, R32 b)                             // == mul(a, b) for canonical a, b
{
    if (a.num == 0 || b.num == 0) return R32(0, 1);
    const uint32_t an = a.num < 0 ? (uint32_t)(-(long long)a.num) : (uint32_t)a.num;
    const uint32_t bn = b.num < 0 ? (uint32_t)(-(long long)b.num) : (uint32_t)b.num;
    const DivFp by1(gcd32(an, (uint32_t)b.den)), by2(gcd32(bn, (uint32_t)a.den));
    const long long mag = (long long)by1(an) * (long long)by2(bn);
    const long long den = (long long)by2((uint32_t)a.den) * (long long)by1((uint32_t)b.den);
    return squeeze_lowest(((a.num < 0) != (b.num < 0)) ? -mag : mag, den);
}
XPG_HD R32 div_canon(R32 a, R32 b)                             // == div(a, b) for canonical a, b with b != 0
{
    if (a.num == 0) return R32(0, 1);
    const R32 r = b.num < 0 ? R32(-b.den, -b.num) : R32(b.den, b.num);     // canonical: |b.num| < 2^31 - 1
    if (a.num == a.den) return r;
    return mul_canon(a, r);                                    // the lowest terms of (a.num * b.den) / (a.den * b.num)
}
// a + p for canonical non-zero a, p: with g = gcd(a.den, p.den), A = a.den / g, P = p.den / g the numerator
// n' = a.num * P + p.num * A is coprime to A and to P, so gcd(n', A * P * g) = gcd(n' mod g, g).
XPG_HD R32 add_lowest(R32 a, R32 p)
{
    const uint32_t g = gcd32((uint32_t)a.den, (uint32_t)p.den);
    const DivFp by(g);
    const uint32_t A = by((uint32_t)a.den), P = by((uint32_t)p.den);
    const long long n = (long long)a.num * (long long)P + (long long)p.num * (long long)A;
    if (n == 0) return R32(0, 1);
    unsigned long long nm = n < 0 ? (unsigned long long)(-n) : (unsigned long long)n;
    unsigned long long d = (unsigned long long)A * (unsigned long long)(uint32_t)p.den;
    if (g != 1) {
        const uint32_t h = gcd32(by.mod(nm), g);
        if (h != 1) {                                          // h divides g, and g divides p.den
            const DivFp byh(h);
            nm = byh.wide(nm); d = (unsigned long long)A * (unsigned long long)byh((uint32_t)p.den);
        }
    }
    return squeeze_lowest(n < 0 ? -(long long)nm : (long long)nm, (long long)d);
}
XPG_HD R32 add_canon(R32 a, R32 p)                             // == add(a, p) for canonical a, p
{
    if (p.num == 0) return a;
    if (a.num == 0) return p;
    return add_lowest(a, p);
}
XPG_HD R32 fma_canon(R32 a, R32 k, R32 e)
{
    if (k.num == 0 || e.num == 0) return a;
    const uint32_t kn = k.num < 0 ? (uint32_t)(-(long long)k.num) : (uint32_t)k.num;
    const uint32_t en = e.num < 0 ? (uint32_t)(-(long long)e.num) : (uint32_t)e.num;
    const DivFp by1(gcd32(kn, (uint32_t)e.den)), by2(gcd32(en, (uint32_t)k.den));
    const long long pmag = (long long)by1(kn) * (long long)by2(en);
    const long long pden = (long long)by2((uint32_t)k.den) * (long long)by1((uint32_t)e.den);
    const R32 p = squeeze_lowest(((k.num < 0) != (e.num < 0)) ? -pmag : pmag, pden);
    if (p.num == 0) return a;                                  // (appro can return 0/1)
    if (a.num == 0) return p;                                  // 0/1 + p = squeeze(p.num, p.den) = p
    return add_lowest(a, p);
}
// ---- the GENERIC operations without a divide -----------------------------------------------------------------
// mul / div / add above are the literal restatement (64-bit Euclid with remainders, two 64-bit quotients) and stay
// the definition. These compute the same results for EVERY pair of operands whose products stay below 2^63 in
// magnitude -- any numerators, any denominators above INT_MIN, zero and negative ones included (a problem whose
// cells are not canonical, e.g. the den = 0 values MIP's equality substitution produces, runs on them) -- with a
// binary gcd on 64 bits and the exact quotients by modular inverse. tests/cxx/fma_canon_fuzz.cpp compares them
// with the literal forms on arbitrary operands.
XPG_HD unsigned long long gcd_u64(unsigned long long x, unsigned long long y)      // x > 0; gcd(x, 0) = x
{
    if (y == 0) return x;
    const int sh = __builtin_ctzll(x | y);
    x >>= __builtin_ctzll(x);
    do {
        y >>= __builtin_ctzll(y);
        const unsigned long long lo = x < y ? x : y, hi = x < y ? y : x;
        x = lo; y = hi - lo;
    } while (y != 0);
    return x << sh;
}
XPG_HD unsigned long long exact_div_u64(unsigned long long x, unsigned long long g)    // g divides x, g > 0
{
    const int sh = __builtin_ctzll(g);
    const unsigned long long o = g >> sh;
    unsigned long long y = (3ull * o) ^ 2ull;
    y *= 2ull - o * y; y *= 2ull - o * y; y *= 2ull - o * y; y *= 2ull - o * y;
    return (x >> sh) * y;
}
XPG_HD R32 squeeze_any(long long n, long long d)               // == squeeze(n, d)
{
    const long long imax = 0x7fffFFFFLL;
    if (n == d) return R32(1, 1);
    if (n == -d) return R32(-1, 1);
    if (d < 0) { n = -n; d = -d; }
    const bool minus = n < 0;
    unsigned long long un = minus ? 0ull - (unsigned long long)n : (unsigned long long)n, ud = (unsigned long long)d;
    if (un == 0) ud = 1;                                       // reduce64: n == 0 -> d = 1
    else if (((un | ud) >> 32) == 0) {                         // both below 2^32 (small data: MIP nodes, dependence systems): the 32-bit forms
        const uint32_t g = gcd32((uint32_t)un, (uint32_t)ud);  // (ud == 0: g = un, the pair becomes 1 / 0 as with gcd_u64)
        if (g != 1) { const DivFp by(g); un = by((uint32_t)un); ud = by((uint32_t)ud); }
    } else {
        const unsigned long long g = gcd_u64(un, ud);
        if (g != 1) {
            if ((g >> 32) == 0 && (un >> 63) == 0 && (ud >> 63) == 0) { const DivFp by((uint32_t)g); un = by.wide(un); ud = by.wide(ud); }
            else { un = exact_div_u64(un, g); ud = exact_div_u64(ud, g); }
        }
    }
    long long mag = (long long)un, dd = (long long)ud;
    if (mag >= (imax >> 2) || dd >= (imax >> 2)) {
        if (mag >= imax || dd >= imax) appro64(mag, dd);
    }
    return R32((int32_t)(minus ? -mag : mag), (int32_t)dd);
}
XPG_HD R32 mul_any_fast(R32 a, R32 b)
{
    const long long n = (long long)a.num * (long long)b.num;
    if (n == 0) return R32(0, 1);
    return squeeze_any(n, (long long)a.den * (long long)b.den);
}
XPG_HD R32 div_any_fast(R32 a, R32 b)
{
    if (a.num == 0) return R32(0, 1);
    if (a.num == a.den) return b.num < 0 ? R32(-b.den, -b.num) : R32(b.den, b.num);
    return squeeze_any((long long)a.num * (long long)b.den, (long long)a.den * (long long)b.num);
}
XPG_HD R32 add_any_fast(R32 a, R32 b)
{
    const long long n = (long long)a.num * (long long)b.den + (long long)a.den * (long long)b.num;
    if (n == 0) return R32(0, 1);
    return squeeze_any(n, (long long)a.den * (long long)b.den);
}
XPG_HD R32 sub(R32 a, R32 b) { return add(a, neg(b)); }
XPG_HD bool eq(R32 a, R32 b) { return a.num == b.num && a.den == b.den; }   // rational.h:80-83
XPG_HD bool ne(R32 a, R32 b) { return a.num != b.num || a.den != b.den; }
XPG_HD bool lt(R32 a, R32 b) { return (long long)a.num * b.den <  (long long)a.den * b.num; }
XPG_HD bool le(R32 a, R32 b) { return (long long)a.num * b.den <= (long long)a.den * b.num; }
XPG_HD bool gt(R32 a, R32 b) { return (long long)a.num * b.den >  (long long)a.den * b.num; }
XPG_HD bool ge(R32 a, R32 b) { return (long long)a.num * b.den >= (long long)a.den * b.num; }
XPG_HD void reduce(R32 & a)   // rational.cpp:76-98, :125-139
{
    if (a.num == 0) { a.den = 1; return; }
    int32_t x = a.num < 0 ? -a.num : a.num;
    int32_t y = a.den < 0 ? -a.den : a.den;
    if (x > y) { int32_t t = x; x = y; y = t; }
    while (x != 0) { int32_t t = x; x = y % x; y = t; }
    if (y != 1) { a.num /= y; a.den /= y; }
    if (a.den < 0) { a.den = -a.den; a.num = -a.num; }
}
XPG_HD bool is_int(R32 a) { return a.den == 1; }
XPG_HD int to_int(R32 a) { return a.num / a.den; }
// gt() is not a strict weak order on arbitrary (num,den) pairs but is on the
// den > 0 values the solver produces; ties are "neither a>b nor b>a".
XPG_HD bool same_value(R32 a, R32 b) { return !gt(a, b) && !gt(b, a); }

template <class S> struct is_f64 { static const bool value = false; };
template <> struct is_f64<F64> { static const bool value = true; };

template <class S> XPG_HD S zero() { return S::from_int(0); }
template <class S> XPG_HD S one() { return S::from_int(1); }
template <class S> XPG_HD S minus_one() { return S::from_int(-1); }

// Matrix::mul / mulOfRow / mulOfColumn scaling rule (matt.h:1331-1368):
// multiplier "== 1" leaves the cell, "== 0" stores zero, else cell * x.
enum ScaleMode { SCALE_KEEP = 0, SCALE_ZERO = 1, SCALE_MUL = 2 };
template <class S> XPG_HD int scale_mode(S x)
{
    if (eq(x, one<S>())) return SCALE_KEEP;
    if (eq(x, zero<S>())) return SCALE_ZERO;
    return SCALE_MUL;
}
template <class S> XPG_HD S scaled(S cell, S x, int mode)
{
    return mode == SCALE_KEEP ? cell : (mode == SCALE_ZERO ? zero<S>() : mul(cell, x));
}

} // namespace xpg
