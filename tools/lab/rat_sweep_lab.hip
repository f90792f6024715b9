// A/B bench of the rational sweep's per-cell arithmetic on exported states of the cfg-4 LP (tools/lab/rat_export.py).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I xpoly_amd/csrc -o /tmp/rat_sweep_lab tools/lab/rat_sweep_lab.hip
// /tmp/rat_sweep_lab tools/lab/_data/pivot16.bin
#include "scalar.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
using namespace xpg;

// ---- policies -----------------------------------------------------------------------------------------------
struct GcdOld { static XPG_HD uint32_t gcd(uint32_t x, uint32_t y) { return gcd32(x, y); } };
struct GcdNew {                                             // 5 VALU per step: min, |x - y|, test, ctz, shift
    static XPG_HD uint32_t gcd(uint32_t x, uint32_t y)
    {
        if (x == 0) return y;
        if (y == 0) return x;
        const int sh = __builtin_ctz(x | y);
        x >>= __builtin_ctz(x); y >>= __builtin_ctz(y);
        for (;;) {
            const uint32_t lo = x < y ? x : y;
#ifdef __HIP_DEVICE_COMPILE__
            const uint32_t d = __usad(x, y, 0u);
#else
            const uint32_t d = x < y ? y - x : x - y;
#endif
            x = lo;
            if (d == 0) break;
            y = d >> __builtin_ctz(d);
        }
        return x << sh;
    }
};
struct GcdAsm {                                            // the same loop with v_sad_u32 spelled out
    static XPG_HD uint32_t gcd(uint32_t x, uint32_t y)
    {
#ifdef __HIP_DEVICE_COMPILE__
        if (x == 0) return y;
        if (y == 0) return x;
        const int sh = __builtin_ctz(x | y);
        x >>= __builtin_ctz(x); y >>= __builtin_ctz(y);
        for (;;) {
            uint32_t d;
            asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(x), "v"(y));
            x = x < y ? x : y;
            if (d == 0) break;
            y = d >> __builtin_ctz(d);
        }
        return x << sh;
#else
        return GcdNew::gcd(x, y);
#endif
    }
};
struct DivOld {
    ExactDiv32 by; uint32_t g;
    XPG_HD explicit DivOld(uint32_t gi) : by(gi), g(gi) {}
    XPG_HD uint32_t q(uint32_t x) const { return by(x); }
    XPG_HD unsigned long long wide(unsigned long long x) const { return by.wide(x); }
    XPG_HD uint32_t mod(unsigned long long x) const { return mod_u64_u32(x, g); }
};
XPG_HD double lab_rcp_int(double g)                            // 1/g for an integer 1 <= g < 2^32, relative error ~2^-52
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
struct LabDivFp {
    double g, r;
    XPG_HD explicit LabDivFp(uint32_t gi) : g((double)gi), r(lab_rcp_int((double)gi)) {}
    XPG_HD uint32_t q(uint32_t x) const { return (uint32_t)__builtin_fma((double)x, r, 0.5); }     // g divides x
    XPG_HD double floor_rem(double x, double & qo) const       // x an integer, |x| < 2^52: x mod g in [0, g), qo = floor(x / g)
    {
        double qq = __builtin_floor(x * r);
        double m = __builtin_fma(-qq, g, x);
        if (m < 0.0) { m += g; qq -= 1.0; }
        if (m >= g) { m -= g; qq += 1.0; }
        qo = qq;
        return m;
    }
    XPG_HD unsigned long long wide(unsigned long long x) const  // g divides x < 2^63
    {
        const double hi = (double)(uint32_t)(x >> 32), lo = (double)(uint32_t)x;
        double q1;
        const double r1 = floor_rem(hi, q1);
        const double xl = __builtin_fma(r1, 4294967296.0, lo);
        const uint32_t q0 = (uint32_t)__builtin_fma(xl, r, 0.5);
        return ((unsigned long long)(uint32_t)q1 << 32) | q0;
    }
    XPG_HD uint32_t mod(unsigned long long x) const             // x < 2^63
    {
        const double hi = (double)(uint32_t)(x >> 32), lo = (double)(uint32_t)x;
        double qd;
        const double t = floor_rem(hi, qd);                    // < g < 2^31
        const double xh = t * 4294967296.0;                    // exact, < 2^63
        const double qh = __builtin_floor(xh * r);
        const double m = __builtin_fma(-qh, g, xh) + lo;       // an integer of magnitude < 2^13 g + 2^32
        return (uint32_t)floor_rem(m, qd);
    }
};

template <class G, class D> XPG_HD R32 add_lowest_t(R32 a, R32 p)
{
    const uint32_t g = G::gcd((uint32_t)a.den, (uint32_t)p.den);
    const D by(g);
    const uint32_t A = by.q((uint32_t)a.den), P = by.q((uint32_t)p.den);
    const long long n = (long long)a.num * (long long)P + (long long)p.num * (long long)A;
    if (n == 0) return R32(0, 1);
    unsigned long long nm = n < 0 ? (unsigned long long)(-n) : (unsigned long long)n;
    unsigned long long d = (unsigned long long)A * (unsigned long long)(uint32_t)p.den;
    if (g != 1) {
        const uint32_t h = G::gcd(by.mod(nm), g);
        if (h != 1) { const D byh(h); nm = byh.wide(nm); d = (unsigned long long)A * (unsigned long long)byh.q((uint32_t)p.den); }
    }
    return squeeze_lowest(n < 0 ? -(long long)nm : (long long)nm, (long long)d);
}
template <class G, class D> XPG_HD R32 fma_canon_t(R32 a, R32 k, R32 e)
{
    if (k.num == 0 || e.num == 0) return a;
    const uint32_t kn = k.num < 0 ? (uint32_t)(-(long long)k.num) : (uint32_t)k.num;
    const uint32_t en = e.num < 0 ? (uint32_t)(-(long long)e.num) : (uint32_t)e.num;
    const D by1(G::gcd(kn, (uint32_t)e.den)), by2(G::gcd(en, (uint32_t)k.den));
    const long long pmag = (long long)by1.q(kn) * (long long)by2.q(en);
    const long long pden = (long long)by2.q((uint32_t)k.den) * (long long)by1.q((uint32_t)e.den);
    const R32 p = squeeze_lowest(((k.num < 0) != (e.num < 0)) ? -pmag : pmag, pden);
    if (p.num == 0) return a;
    if (a.num == 0) return p;
    return add_lowest_t<G, D>(a, p);
}
struct FmaProduct { static XPG_HD R32 f(R32 a, R32 k, R32 e) { return fma_canon(a, k, e); } };
template <class G, class D> struct FmaT { static XPG_HD R32 f(R32 a, R32 k, R32 e) { return fma_canon_t<G, D>(a, k, e); } };

// ---- kernels ------------------------------------------------------------------------------------------------
// the product's shape: one row x 256 columns per workgroup, rows as the fast grid index
template <class F, int EARLY> __global__ __launch_bounds__(256)
void k_sweep(R32 * __restrict__ tab, int m, int W, int ld, const R32 * __restrict__ rowbuf, const R32 * __restrict__ colbuf, int r)
{
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= W) return;
    const int i = blockIdx.x;
    R32 * p = tab + (size_t)i * ld + j;
    R32 a, k;
    if (EARLY) { a = *p; k = colbuf[i]; }
    const R32 e = rowbuf[j];
    if (e.num == 0) { if (i == r) *p = e; return; }
    if (!EARLY) { a = *p; k = colbuf[i]; }
    *p = (i == r) ? e : F::f(a, k, e);
}

static std::vector<char> slurp(const char * fn)
{
    FILE * f = fopen(fn, "rb"); if (!f) { perror(fn); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> b(n); if (fread(b.data(), 1, n, f) != (size_t)n) exit(2); fclose(f); return b;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)

int main(int argc, char ** argv)
{
    for (int f = 1; f < argc; f++) {
        std::vector<char> b = slurp(argv[f]);
        const int * h = (const int *)b.data();
        const int m = h[0], W = h[1], r = h[2];
        const size_t cells = (size_t)m * W;
        const R32 * T = (const R32 *)(b.data() + 16), * E = T + cells, * K = E + W, * U = K + m;
        const int ld = W;
        R32 * dT, * d0, * dE, * dK;
        CK(hipMalloc(&dT, cells * 8)); CK(hipMalloc(&d0, cells * 8)); CK(hipMalloc(&dE, W * 8)); CK(hipMalloc(&dK, m * 8));
        CK(hipMemcpy(d0, T, cells * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dE, E, W * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dK, K, m * 8, hipMemcpyHostToDevice));
        std::vector<R32> out(cells);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        printf("%s: %d x %d, pivot row %d\n", argv[f], m, W, r);
        size_t lds_bytes = 0;
        auto run = [&](const char * name, auto kern) {
            float best = 1e9f, sum = 0;
            const int reps = 12;
            for (int it = 0; it < reps; it++) {
                CK(hipMemcpy(dT, d0, cells * 8, hipMemcpyDeviceToDevice));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(kern, dim3(m, (W + 255) / 256), dim3(256), lds_bytes, 0, dT, m, W, ld, dE, dK, r);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            CK(hipMemcpy(out.data(), dT, cells * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t c = 0; c < cells; c++) bad += (out[c].num != U[c].num || out[c].den != U[c].den);
            printf("  %-44s best %7.2f us  mean %7.2f us  cells differing from the oracle: %zu\n", name, best * 1e3f, sum / (reps - 2) * 1e3f, bad);
        };
        run("product fma_canon", k_sweep<FmaProduct, 0>);
        run("product fma_canon, loads first", k_sweep<FmaProduct, 1>);
        run("old gcd, old div (template)", k_sweep<FmaT<GcdOld, DivOld>, 0>);
        run("new gcd, old div", k_sweep<FmaT<GcdNew, DivOld>, 0>);
        run("old gcd, fp64 div", k_sweep<FmaT<GcdOld, LabDivFp>, 0>);
        run("new gcd, fp64 div", k_sweep<FmaT<GcdNew, LabDivFp>, 0>);
        run("new gcd, fp64 div, loads first", k_sweep<FmaT<GcdNew, LabDivFp>, 1>);
        run("asm gcd, old div", k_sweep<FmaT<GcdAsm, DivOld>, 0>);
        run("asm gcd, fp64 div", k_sweep<FmaT<GcdAsm, LabDivFp>, 0>);
        for (int kb : {24, 32, 48, 64}) {                        // dynamic LDS as an occupancy limiter: 160 KB per CU
            char nm[96]; snprintf(nm, 96, "asm gcd, fp64 div, %d workgroups per CU", 160 / kb);
            lds_bytes = (size_t)kb * 1024;
            CK(hipFuncSetAttribute((const void *)k_sweep<FmaT<GcdAsm, LabDivFp>, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            run(nm, k_sweep<FmaT<GcdAsm, LabDivFp>, 0>);
        }
        lds_bytes = 0;
        hipFree(dT); hipFree(d0); hipFree(dE); hipFree(dK);
    }
    return 0;
}
