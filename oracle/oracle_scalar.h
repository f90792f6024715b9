// TEST INFRASTRUCTURE -- CPU restatement ("port") of xpoly's two scalar types.
// Never included by the product (xpoly_amd/); see oracle/README.md.
//
// Each function cites the reference lines (relative to /root/reference/src/com)
// whose behaviour it restates. The restatement is deliberately bug-compatible:
// tolerant Float '==', field-wise Rational '==', the float32 'appro' rescue,
// silent int64->int32 truncation, denominators of zero flowing through.
#ifndef XPOLY_ORACLE_SCALAR_H
#define XPOLY_ORACLE_SCALAR_H

#include <stdint.h>

namespace orc {

// ---------------------------------------------------------------------------
// F64 -- xcom::Float (flty.h:47-98): one IEEE double, PRECISION_TYPE = double.
// ---------------------------------------------------------------------------
struct F64 {
    double v;
    F64() : v(0.0) {}
    F64(int i) : v((double)i) {}
    explicit F64(double d) : v(d) {}
};

static const double kTiny = 0.00000000000000001; // INFINITESIMAL, flty.h:46

// flty.cpp:41-58 -- equal iff not of strictly opposite sign and the
// magnitudes differ by at most 1e-17.
inline bool eq(F64 a, F64 b)
{
    double x = a.v, y = b.v;
    if ((x > 0 && y < 0) || (x < 0 && y > 0)) return false;
    if (x < 0) x = -x;
    if (y < 0) y = -y;
    if ((x == 0.0 && y <= kTiny) || (y == 0.0 && x <= kTiny)) return true;
    return (x > y ? x - y : y - x) <= kTiny;
}
inline bool ne(F64 a, F64 b) { return !eq(a, b); }             // flty.h:103
inline bool lt(F64 a, F64 b) { return a.v < b.v; }              // flty.cpp:61-67
inline bool gt(F64 a, F64 b) { return a.v > b.v; }              // flty.cpp:79-85
inline bool le(F64 a, F64 b) { return a.v < b.v || eq(a, b); }  // flty.cpp:70-76
inline bool ge(F64 a, F64 b) { return a.v > b.v || eq(a, b); }  // flty.cpp:88-94
inline F64 mul(F64 a, F64 b) { return F64(a.v * b.v); }         // flty.cpp:97-101
inline F64 div(F64 a, F64 b) { return F64(a.v / b.v); }         // flty.cpp:104-108
inline F64 add(F64 a, F64 b) { return F64(a.v + b.v); }         // flty.cpp:111-115
inline F64 sub(F64 a, F64 b) { return F64(a.v - b.v); }         // flty.cpp:119-123
inline F64 neg(F64 a) { return F64(-a.v); }                     // flty.cpp:127-131
inline void reduce(F64 &) {}                                    // flty.h:95
inline int to_int(F64 a) { return (int)a.v; }                   // flty.h:85-88
inline bool int_cast_defined(F64) { return true; }

// flty.cpp:182-201
inline bool is_int(F64 a)
{
    double x = a.v < 0 ? -a.v : a.v;
    long long t = (long long)x;
    if ((x - (double)t) < kTiny) return true;
    return ((double)(t + 1) - x) < kTiny;
}

// ---------------------------------------------------------------------------
// R32 -- xcom::Rational (rational.h:39-71): int32 numerator / int32 denominator.
// ---------------------------------------------------------------------------
struct R32 {
    int32_t num, den;
    R32() : num(0), den(1) {}                 // rational.cpp:44-48
    R32(int n) : num(n), den(1) {}            // rational.cpp:60-64 (den defaults to 1)
    R32(int n, int d) : num(n), den(d) {}
};

static const long long kIntMax = 0x7fffFFFFLL;

// rational.cpp:142-158 -- Euclid on magnitudes; gcd(x,0) = |x|.
inline long long gcd_ll(long long x, long long y)
{
    if (x < 0) x = -x;
    if (y < 0) y = -y;
    if (x > y) { long long t = x; x = y; y = t; }
    while (x != 0) { long long t = x; x = y % x; y = t; }
    return y;
}

struct Counters { long long reduce_calls, appro_calls, pivots; };
inline Counters & counters() { static Counters c = {0, 0, 0}; return c; }

// rational.cpp:163-185
inline void reduce_ll(long long & n, long long & d)
{
    counters().reduce_calls++;
    if (n == 0) { d = 1; return; }
    long long g = gcd_ll(n, d);
    if (g != 1) { n /= g; d /= g; }
    if (d < 0) { d = -d; n = -n; }
}

// rational.cpp:189-226 -- lossy re-quantisation through *float32*.
// The ladder skips from 1e3 straight to 1e5, as the source does.
inline void appro(long long & n, long long & d)
{
    counters().appro_calls++;
    float q = (float)n / (float)d;
    if (q < 100.0) { q = q * 1000000; n = (int)q; d = 1000000; }
    else if (q < 1000.0) { q = q * 100000; n = (int)q; d = 100000; }
    else if (q < 100000.0) { q = q * 10000; n = (int)q; d = 10000; }
    else if (q < 1000000.0) { q = q * 1000; n = (int)q; d = 1000; }
    else if (q < 10000000.0) { q = q * 100; n = (int)q; d = 100; }
    else if (q < 100000000.0) { q = q * 10; n = (int)q; d = 10; }
    else if (q < 2147483647.0) { n = (int)q; d = 1; }
    else { n = 0; d = 1; }   // release build: ASSERT compiled out (ltype.h:136-139)
    reduce_ll(n, d);
}

// Common tail of '*', '/', '+' (rational.cpp:285-309, :336-360, :373-396):
// given the raw int64 fraction, normalise sign, reduce, and squeeze into int32.
inline R32 squeeze(long long n, long long d)
{
    if (n == d) return R32(1, 1);
    if (n == -d) return R32(-1, 1);
    if (d < 0) { n = -n; d = -d; }
    reduce_ll(n, d);
    long long mag = n >= 0 ? n : -n;
    if (mag >= (kIntMax >> 2) || d >= (kIntMax >> 2)) {
        reduce_ll(mag, d);
        if (mag >= kIntMax || d >= kIntMax) appro(mag, d);
    }
    R32 r;
    r.num = (int32_t)(n < 0 ? -mag : mag);
    r.den = (int32_t)d;
    return r;
}

// rational.cpp:273-310
inline R32 mul(R32 a, R32 b)
{
    long long n = (long long)a.num * (long long)b.num;
    if (n == 0) return R32(0, 1);
    return squeeze(n, (long long)a.den * (long long)b.den);
}

// rational.cpp:312-361 -- note the unreduced-reciprocal shortcut when a == x/x.
inline R32 div(R32 a, R32 b)
{
    if (a.num == 0) return R32(0, 1);
    if (a.num == a.den) return b.num < 0 ? R32(-b.den, -b.num) : R32(b.den, b.num);
    return squeeze((long long)a.num * (long long)b.den,
                   (long long)a.den * (long long)b.num);
}

// rational.cpp:363-397
inline R32 add(R32 a, R32 b)
{
    long long n = (long long)a.num * (long long)b.den +
                  (long long)a.den * (long long)b.num;
    if (n == 0) return R32(0, 1);
    return squeeze(n, (long long)a.den * (long long)b.den);
}

inline R32 neg(R32 a) { return R32(-a.num, a.den); }   // rational.h:100-105
inline R32 sub(R32 a, R32 b) { return add(a, neg(b)); } // rational.h:96-97

// rational.h:80-83 -- field-wise, no cross-multiplication.
inline bool eq(R32 a, R32 b) { return a.num == b.num && a.den == b.den; }
inline bool ne(R32 a, R32 b) { return a.num != b.num || a.den != b.den; }
// rational.cpp:229-271 -- int64 cross products, denominators' signs not inspected.
inline bool lt(R32 a, R32 b) { return (long long)a.num * b.den <  (long long)a.den * b.num; }
inline bool le(R32 a, R32 b) { return (long long)a.num * b.den <= (long long)a.den * b.num; }
inline bool gt(R32 a, R32 b) { return (long long)a.num * b.den >  (long long)a.den * b.num; }
inline bool ge(R32 a, R32 b) { return (long long)a.num * b.den >= (long long)a.den * b.num; }

// rational.cpp:76-98 with the int32 gcd of :125-139
inline void reduce(R32 & a)
{
    if (a.num == 0) { a.den = 1; return; }
    int32_t x = a.num < 0 ? -a.num : a.num;
    int32_t y = a.den < 0 ? -a.den : a.den;
    if (x > y) { int32_t t = x; x = y; y = t; }
    while (x != 0) { int32_t t = x; x = y % x; y = t; }
    if (y != 1) { a.num /= y; a.den /= y; }
    if (a.den < 0) { a.den = -a.den; a.num = -a.num; }
}
inline bool is_int(R32 a) { return a.den == 1; }        // rational.h:60
inline int to_int(R32 a) { return a.num / a.den; }      // rational.h:59
inline bool int_cast_defined(R32 a) { return a.den != 0; }
inline R32 rabs(R32 a)                                    // rational.cpp:101-112
{
    return R32(a.num < 0 ? -a.num : a.num, a.den < 0 ? -a.den : a.den);
}

} // namespace orc
#endif
