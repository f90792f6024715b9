// CPU check of xpg::fma_canon (xpoly_amd/csrc/scalar.hip.h): the fused a + k*e of the rational sweep must equal
// add(a, mul(k, e)) -- the reference's two operations (src/com/rational.cpp:273-310, :363-397) -- bit for bit
// on canonical operands, including the magnitudes that trigger the float32 `appro` rescue.
// Build: hipcc -O2 -ffp-contract=off -o tests/cxx/fma_canon_fuzz tests/cxx/fma_canon_fuzz.cpp   (host code only)
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "../../xpoly_amd/csrc/scalar.hip.h"

using namespace xpg;

static uint64_t s = 88172645463325252ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

// a canonical rational whose size class is drawn first (small integers ... near the appro threshold)
static R32 draw()
{
    static const int64_t caps[] = { 4, 10, 100, 10000, 1000000, 0x1fffffff, 0x7ffffffe };
    for (;;) {
        const int64_t cn = caps[rnd() % 7], cd = caps[rnd() % 7];
        int64_t n = (int64_t)(rnd() % (uint64_t)(2 * cn + 1)) - cn, d = 1 + (int64_t)(rnd() % (uint64_t)cd);
        if (rnd() % 5 == 0) d = 1;
        R32 r((int32_t)n, (int32_t)d);
        reduce(r);
        if (canonical(r)) return r;
    }
}

// the divide-free helpers on their own: remainders of 64-bit values below 2^63 (random, and at the edges of the
// fp64 estimates), exact quotients of 32- and 64-bit multiples
static int check_division_helpers(long n)
{
    static const uint32_t gs[] = { 1u, 2u, 3u, 5u, 7u, 10u, 1000000u, 0x7ffffffeu, 0x7fffffffu, 0x40000000u, 0x3fffffffu, 65537u, 0xfffeu };
    for (long it = 0; it < n; it++) {
        uint32_t g = (it % 7 == 0) ? gs[rnd() % (sizeof(gs) / sizeof(gs[0]))] : (uint32_t)(rnd() % 0x7fffffffu) + 1u;
        unsigned long long x = rnd() >> (1 + rnd() % 40);
        if (it % 11 == 0) x = 0x7fffffffffffffffull - rnd() % 1000;
        if (it % 13 == 0) x = (unsigned long long)g * (rnd() % 0xffffffffull) + (it % 3 == 0 ? g - 1 : 0);
        if (x >> 63) x >>= 1;
        if (mod_u64_u32(x, g) != (uint32_t)(x % g)) { printf("MISMATCH mod_u64_u32 x=%llu g=%u\n", x, g); return 1; }
        const uint32_t q32 = (uint32_t)(rnd() % (0xffffffffull / g + 1));
        const ExactDiv32 by(g);
        if (by(q32 * g) != q32) { printf("MISMATCH ExactDiv32 q=%u g=%u\n", q32, g); return 1; }
        const unsigned long long q64 = rnd() % (0xffffffffffffffffull / g);
        if (by.wide(q64 * g) != q64) { printf("MISMATCH ExactDiv32::wide q=%llu g=%u\n", q64, g); return 1; }
        // the fp64 forms the canonical operations use (g < 2^31 there; dividends below 2^63)
        const DivFp fp(g);
        if (fp(q32 * g) != q32) { printf("MISMATCH DivFp q=%u g=%u\n", q32, g); return 1; }
        if (fp.mod(x) != (uint32_t)(x % g)) { printf("MISMATCH DivFp::mod x=%llu g=%u\n", x, g); return 1; }
        const unsigned long long q63 = rnd() % (0x7fffffffffffffffull / g + 1);
        if (fp.wide(q63 * g) != q63) { printf("MISMATCH DivFp::wide q=%llu g=%u\n", q63, g); return 1; }
        const unsigned long long top = (0x7fffffffffffffffull / g) * g;      // the largest multiple below 2^63
        if (fp.wide(top) != top / g || fp.mod(top + (g > 1 ? g - 1 : 0)) != (g > 1 ? g - 1 : 0)) { printf("MISMATCH DivFp at the top, g=%u\n", g); return 1; }
    }
    return 0;
}

// the divide-free GENERIC operations against the literal ones on arbitrary operands: small and large numerators,
// denominators that are zero, negative, not coprime -- everything the flat ABI lets through (INT_MIN excepted)
static R32 draw_any()
{
    static const int64_t caps[] = { 1, 3, 10, 1000, 1000000, 0x3fffffff, 0x7fffffff };
    const int64_t cn = caps[rnd() % 7], cd = caps[rnd() % 7];
    int64_t n = (int64_t)(rnd() % (uint64_t)(2 * cn + 1)) - cn, d = (int64_t)(rnd() % (uint64_t)(2 * cd + 1)) - cd;
    if (rnd() % 4 == 0) d = 1;
    if (rnd() % 11 == 0) d = 0;
    return R32((int32_t)n, (int32_t)d);
}
static int check_generic_forms(long n)
{
    for (long it = 0; it < n; it++) {
        const R32 a = draw_any(), b = draw_any();
        const R32 m1 = mul(a, b), m2 = mul_any_fast(a, b), s1 = add(a, b), s2 = add_any_fast(a, b), d1 = div(a, b), d2 = div_any_fast(a, b);
        if (m1.num != m2.num || m1.den != m2.den || s1.num != s2.num || s1.den != s2.den || d1.num != d2.num || d1.den != d2.den) {
            printf("MISMATCH generic a=%d/%d b=%d/%d mul %d/%d vs %d/%d add %d/%d vs %d/%d div %d/%d vs %d/%d\n", a.num, a.den, b.num, b.den,
                   m1.num, m1.den, m2.num, m2.den, s1.num, s1.den, s2.num, s2.den, d1.num, d1.den, d2.num, d2.den);
            return 1;
        }
    }
    return 0;
}

int main(int argc, char ** argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 3000000;
    if (check_division_helpers(N)) return 1;
    if (check_generic_forms(N)) return 1;
    long appro_like = 0, zeros = 0;
    for (long it = 0; it < N; it++) {
        const R32 a = draw(), k = draw(), e = draw();
        const R32 want = add(a, mul(k, e));
        const R32 got = fma_canon(a, k, e);
        if (want.num != got.num || want.den != got.den) {
            printf("MISMATCH a=%d/%d k=%d/%d e=%d/%d want=%d/%d got=%d/%d\n", a.num, a.den, k.num, k.den, e.num, e.den,
                   want.num, want.den, got.num, got.den);
            return 1;
        }
        const R32 m1 = mul(k, e), m2 = mul_canon(k, e), a1 = add(a, k), a2 = add_canon(a, k);
        if (m1.num != m2.num || m1.den != m2.den || a1.num != a2.num || a1.den != a2.den) {
            printf("MISMATCH mul/add_canon a=%d/%d k=%d/%d e=%d/%d\n", a.num, a.den, k.num, k.den, e.num, e.den);
            return 1;
        }
        if (k.num != 0) {
            const R32 d1 = div(a, k), d2 = div_canon(a, k);
            if (d1.num != d2.num || d1.den != d2.den || !canonical(d2)) {
                printf("MISMATCH div_canon a=%d/%d k=%d/%d want=%d/%d got=%d/%d\n", a.num, a.den, k.num, k.den, d1.num, d1.den, d2.num, d2.den);
                return 1;
            }
        }
        if (!canonical(m2) || !canonical(a2)) { printf("mul/add_canon result not canonical\n"); return 1; }
        if (!canonical(got)) { printf("result not canonical: %d/%d\n", got.num, got.den); return 1; }
        if (want.den == 1000000 || want.den == 100000 || want.den == 10000 || want.den == 1000) appro_like++;
        if (want.num == 0) zeros++;
    }
    printf("fma_canon == add(mul) on %ld canonical triples (%ld results with an appro denominator, %ld zeros)\n", N, appro_like, zeros);
    return 0;
}
