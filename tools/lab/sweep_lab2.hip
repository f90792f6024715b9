// Lab bench, round 3: the blocked sweep OUTSIDE the Infinity Cache.
// Times variants of the 16-stage body at 4096 x 8192 (268 MB), 4096 x 12289 (403 MB, ld 12304), 8192 x 8192
// (537 MB) with HIP events, outside the library. Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/_build/sweep_lab2 tools/lab/sweep_lab2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// order: 0 natural, 1 reversed (both axes)
template <int ROWS, int U, int NB, bool NT, int WAVES> __global__ __launch_bounds__(256, WAVES)
void k_sweep(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
             const double * __restrict__ K, int rev, int oddscalar)
{
    int bx, by;
    {
        const int gx = (int)gridDim.x, gy = (int)gridDim.y;
        int lid = (int)blockIdx.y * gx + (int)blockIdx.x;
        if ((rev & 1) && !(rev & 16)) lid = gx * gy - 1 - lid;
        if (rev & 16) {                                         // hybrid: strips below F = 8*floor(S/8) static (strip s on XCD s % 8), the S - F leftover strips dealt over the XCDs by row block
            const int S = (W + 511) / 512, c = lid & 7, k = lid >> 3;
            const int nown = S >> 3, L = S & 7, per = 8 * nown + L;
            const int super = k / per, rem = k - super * per;
            int yy;
            if (rem < 8 * nown) { yy = rem / nown; bx = c + 8 * (rem - yy * nown); }
            else { const int q = rem - 8 * nown; yy = (c - q) & 7; bx = 8 * nown + q; }
            by = 8 * super + yy;
            const int R = m / ROWS;
            if (by >= R) return;
            if (rev & 1) { by = R - 1 - by; }
        } else
        if (rev & 8) {                                          // XCD-static: strip s belongs to XCD s % 8 (gx is padded to a multiple of 8)
            const int S = (W + 511) / 512, c = lid & 7, k = lid >> 3;
            const int nc = (S - c + 7) / 8;                     // strips of this XCD
            if (nc <= 0 || k >= nc * gy) return;
            by = k / nc; bx = c + 8 * (k % nc);
        } else
        if (rev & 2) { bx = lid / gy; by = lid % gy; if ((rev & 4) && (bx & 1)) by = gy - 1 - by; }
        else { by = lid / gx; bx = lid % gx; }
    }
    const int j = bx * 512 + threadIdx.x * 2;
    const int i0 = by * ROWS;
    if (j >= W) return;
    if (oddscalar && j + 1 >= W) {                              // round 2's odd last column: run-time scalar loop
        for (int i = i0; i < i0 + ROWS; i++) {
            double * p = tab + (size_t)i * ld + j;
            double ax = p[0];
            for (int s = 0; s < NB; s++) {
                const double k = K[(size_t)i * NB + s];
                const double ex = E[(size_t)s * ld + j];
                ax = ax + k * ex;
            }
            p[0] = ax;
        }
        return;
    }
    v2d a[U], b[U];
    auto load = [&](v2d (&d)[U], const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const v2d * q = reinterpret_cast<const v2d *>(p + (size_t)u * ld);
            d[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
    };
    double * base = tab + (size_t)i0 * ld + j;
    load(a, base);
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
    auto apply = [&](v2d (&d)[U], double * p, int row0) {
#pragma unroll
        for (int s = 0; s < NB; s++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double k = K[(size_t)(row0 + u) * NB + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                d[u].x = d[u].x + p0; d[u].y = d[u].y + p1;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            v2d * q = reinterpret_cast<v2d *>(p + (size_t)u * ld);
            if (NT) __builtin_nontemporal_store(d[u], q); else *q = d[u];
        }
    };
#pragma unroll 1
    for (int i = i0; i < i0 + ROWS; i += 2 * U) {
        load(b, base + (size_t)U * ld);
        apply(a, base, i);
        if (i + 2 * U < i0 + ROWS) load(a, base + (size_t)2 * U * ld);
        apply(b, base + (size_t)U * ld, i + U);
        base += (size_t)2 * U * ld;
    }
}

// in-place "copy": the same loads and stores, one multiply
template <int ROWS, int U, bool NT> __global__ __launch_bounds__(256)
void k_copy(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
            const double * __restrict__ K, int rev, int)
{
    const int bx = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int by = rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
    const int j = bx * 512 + threadIdx.x * 2;
    const int i0 = by * ROWS;
    if (j >= W) return;
    double * base = tab + (size_t)i0 * ld + j;
    for (int i = 0; i < ROWS; i += U) {
        v2d a[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const v2d * q = reinterpret_cast<const v2d *>(base + (size_t)(i + u) * ld); a[u] = NT ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
        for (int u = 0; u < U; u++) { v2d * q = reinterpret_cast<v2d *>(base + (size_t)(i + u) * ld); v2d x = a[u] * 1.0000001; if (NT) __builtin_nontemporal_store(x, q); else *q = x; }
    }
}

typedef void (*kern_t)(double *, int, int, int, const double *, const double *, int, int);

int main(int argc, char ** argv)
{
    struct Shape { int m, W, ld; } shapes[] = { {4096, 8192, 0}, {4096, 12289, 0}, {8192, 8192, 0}, {4096, 12288, 0}, {4096, 16385, 0},
        {4096, 12289, 12320}, {4096, 12289, 12352}, {4096, 12289, 12416}, {4096, 12289, 12544}, {4096, 12289, 12800}, {4096, 8193, 8208}, {4096, 8193, 8704},
        {4096, 8192, 8208}, {4096, 8192, 8704}, {4096, 16385, 16896}, {1024, 3073, 3088}, {2048, 6145, 6160},
        {4096, 12290, 12304}, {4096, 12304, 12304}, {4096, 12352, 12352}, {4096, 12544, 12544}, {4096, 12800, 12800}, {4096, 12288, 12304}, {4096, 11776, 11776}, {4096, 13312, 13312} };
    const size_t maxcells = (size_t)8192 * 16900;
    double *tab, *E, *K;
    CK(hipMalloc(&tab, maxcells * 8)); CK(hipMalloc(&E, (size_t)32 * 16900 * 8)); CK(hipMalloc(&K, (size_t)8192 * 32 * 8));
    {
        std::vector<double> h(maxcells);
        unsigned long long x = 88172645463325252ull;
        auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 - 0.5; };
        for (auto & v : h) v = rnd();
        CK(hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < (size_t)32 * 16900; i++) h[i] = rnd();
        CK(hipMemcpy(E, h.data(), (size_t)32 * 16900 * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < (size_t)8192 * 32; i++) h[i] = rnd() * 1e-3;
        CK(hipMemcpy(K, h.data(), (size_t)8192 * 32 * 8, hipMemcpyHostToDevice));
    }
    struct V { const char * name; kern_t f; int rows; int nb; int odd; int serp; int xm; } vs[] = {
        {"copy <16,4>", k_copy<16, 4, false>, 16, 1, 0, 0},
        {"copy <16,4> serpentine", k_copy<16, 4, false>, 16, 1, 0, 1},
        {"copy <32,8>", k_copy<32, 8, false>, 32, 1, 0, 0},
        {"copy <32,8> nt", k_copy<32, 8, true>, 32, 1, 0, 0},
        {"copy <32,8> nt serpentine", k_copy<32, 8, true>, 32, 1, 0, 1},
        {"r2 product shape <16,4> odd scalar", k_sweep<16, 4, 16, false, 1>, 16, 16, 1, 0},
        {"<16,4> pad", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 0},
        {"<16,4> pad serpentine", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 1},
        {"<16,4> pad nt", k_sweep<16, 4, 16, true, 1>, 16, 16, 0, 0},
        {"<16,4> pad nt serpentine", k_sweep<16, 4, 16, true, 1>, 16, 16, 0, 1},
        {"<16,4> pad x-major", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 0, 2},
        {"<16,4> pad x-major serpentine", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 1, 2},
        {"<16,4> pad x-major boustrophedon serp", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 1, 6},
        {"<32,4> pad x-major serpentine", k_sweep<32, 4, 16, false, 1>, 32, 16, 0, 1, 2},
        {"<8,4> pad x-major serpentine", k_sweep<8, 4, 16, false, 1>, 8, 16, 0, 1, 2},
        {"<8,2> pad x-major serpentine", k_sweep<8, 2, 16, false, 1>, 8, 16, 0, 1, 2},
        {"<16,2> pad x-major serpentine", k_sweep<16, 2, 16, false, 1>, 16, 16, 0, 1, 2},
        {"NB=32 <16,2> pad x-major serpentine", k_sweep<16, 2, 32, false, 1>, 16, 32, 0, 1, 2},
        {"NB=24 <16,4> pad x-major serpentine", k_sweep<16, 4, 24, false, 1>, 16, 24, 0, 1, 2},
        {"<16,4> pad xcd-static", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 0, 8},
        {"<16,4> pad xcd-static serpentine", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 1, 8},
        {"<32,4> pad xcd-static serpentine", k_sweep<32, 4, 16, false, 1>, 32, 16, 0, 1, 8},
        {"<8,4> pad xcd-static serpentine", k_sweep<8, 4, 16, false, 1>, 8, 16, 0, 1, 8},
        {"<16,2> pad xcd-static serpentine", k_sweep<16, 2, 16, false, 1>, 16, 16, 0, 1, 8},
        {"NB=32 <16,2> pad xcd-static serpentine", k_sweep<16, 2, 32, false, 1>, 16, 32, 0, 1, 8},
        {"NB=24 <16,4> pad xcd-static serpentine", k_sweep<16, 4, 24, false, 1>, 16, 24, 0, 1, 8},
        {"<16,4> pad hybrid", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 0, 16},
        {"<16,4> pad hybrid serpentine", k_sweep<16, 4, 16, false, 1>, 16, 16, 0, 1, 16},
        {"<32,4> pad hybrid serpentine", k_sweep<32, 4, 16, false, 1>, 32, 16, 0, 1, 16},
        {"<8,4> pad hybrid serpentine", k_sweep<8, 4, 16, false, 1>, 8, 16, 0, 1, 16},
        {"<16,2> pad hybrid serpentine", k_sweep<16, 2, 16, false, 1>, 16, 16, 0, 1, 16},
        {"NB=32 <16,2> pad hybrid serpentine", k_sweep<16, 2, 32, false, 1>, 16, 32, 0, 1, 16},
        {"NB=24 <16,4> pad hybrid serpentine", k_sweep<16, 4, 24, false, 1>, 16, 24, 0, 1, 16},
        {"<32,4> pad", k_sweep<32, 4, 16, false, 1>, 32, 16, 0, 0},
        {"<32,4> pad serpentine", k_sweep<32, 4, 16, false, 1>, 32, 16, 0, 1},
        {"<64,4> pad serpentine", k_sweep<64, 4, 16, false, 1>, 64, 16, 0, 1},
        {"<16,8> pad", k_sweep<16, 8, 16, false, 1>, 16, 16, 0, 0},
        {"<16,8> pad serpentine", k_sweep<16, 8, 16, false, 1>, 16, 16, 0, 1},
        {"<32,8> pad serpentine", k_sweep<32, 8, 16, false, 1>, 32, 16, 0, 1},
        {"<16,2> pad serpentine", k_sweep<16, 2, 16, false, 1>, 16, 16, 0, 1},
        {"<8,2> pad serpentine", k_sweep<8, 2, 16, false, 1>, 8, 16, 0, 1},
        {"<8,4> pad serpentine", k_sweep<8, 4, 16, false, 1>, 8, 16, 0, 1},
        {"NB=32 <16,2> pad", k_sweep<16, 2, 32, false, 1>, 16, 32, 0, 0},
        {"NB=32 <16,2> pad serpentine", k_sweep<16, 2, 32, false, 1>, 16, 32, 0, 1},
        {"NB=32 <16,4> pad serpentine", k_sweep<16, 4, 32, false, 1>, 16, 32, 0, 1},
        {"NB=32 <32,4> pad serpentine", k_sweep<32, 4, 32, false, 1>, 32, 32, 0, 1},
        {"NB=32 <32,2> pad serpentine", k_sweep<32, 2, 32, false, 1>, 32, 32, 0, 1},
        {"NB=24 <16,4> pad serpentine", k_sweep<16, 4, 24, false, 1>, 16, 24, 0, 1},
        {"NB=8 <32,8> pad serpentine", k_sweep<32, 8, 8, false, 1>, 32, 8, 0, 1},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (argc > 2 && !strcmp(argv[1], "ldscan")) {                // ldscan W [m]: the product shape over leading dimensions W.. in steps of 16
        const int W = atoi(argv[2]), m = argc > 3 ? atoi(argv[3]) : 4096;
        const double bytes = 2.0 * m * W * 8;
        for (int ld = (W + 15) / 16 * 16; ld <= W + 1100; ld += (ld - W < 160 ? 16 : 64)) {
            const int S = (W + 511) / 512, R = m / 16;
            dim3 g(8 * ((R + 7) / 8) * (8 * (S / 8) + (S & 7)));
            int flip = 0;
            for (int w = 0; w < 6; w++) hipLaunchKernelGGL((k_sweep<16, 4, 16, false, 1>), g, dim3(256), 0, 0, tab, m, W, ld, E, K, (flip ^= 1) | 16, 0);
            CK(hipDeviceSynchronize());
            const int reps = 30;
            CK(hipEventRecord(e0, 0));
            for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k_sweep<16, 4, 16, false, 1>), g, dim3(256), 0, 0, tab, m, W, ld, E, K, (flip ^= 1) | 16, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / reps;
            printf("W %d ld %d (+%d, %d B mod 4096 = %d)  %8.2f us  frac %.3f\n", W, ld, ld - W, ld * 8, (ld * 8) % 4096, us, bytes / us / 1e6 / 8.0);
        }
        return 0;
    }
    const int only = argc > 1 ? atoi(argv[1]) : -1; const int from = argc > 3 ? atoi(argv[3]) : 0;
    for (int si = 0; si < (int)(sizeof shapes / sizeof shapes[0]); si++) {
        if ((only >= 0 && si != only) || si < from) continue;
        const int m = shapes[si].m, W = shapes[si].W, ld = shapes[si].ld ? shapes[si].ld : (W + 15) / 16 * 16;
        const double bytes = 2.0 * m * W * 8;
        printf("== tableau %d x %d (ld %d): %.1f MB, %.1f MB per launch\n", m, W, ld, (double)m * ld * 8 / 1e6, bytes / 1e6);
        for (auto & v : vs) {
            if (argc > 2 && !strstr(argv[2], "all") && !(strstr(v.name, "copy <16,4>") || strstr(v.name, "<16,4> pad") || strstr(v.name, "<32,4> pad") || strstr(v.name, "<64,4>") || strstr(v.name, "x-major") || strstr(v.name, "xcd") || strstr(v.name, "hybrid"))) continue;
            dim3 g((W + 511) / 512, m / v.rows); if (v.xm & 8) g.x = (g.x + 7) / 8 * 8;
            if (v.xm & 16) { const int S = g.x, R = g.y; g.x = 8 * ((R + 7) / 8) * (8 * (S / 8) + (S & 7)); g.y = 1; }
            int flip = 0;
            for (int w = 0; w < 6; w++) { hipLaunchKernelGGL(v.f, g, dim3(256), 0, 0, tab, m, W, ld, E, K, (v.serp ? (flip ^= 1) : 0) | v.xm, v.odd); }
            CK(hipDeviceSynchronize());
            const int reps = 30;
            CK(hipEventRecord(e0, 0));
            for (int w = 0; w < reps; w++) hipLaunchKernelGGL(v.f, g, dim3(256), 0, 0, tab, m, W, ld, E, K, (v.serp ? (flip ^= 1) : 0) | v.xm, v.odd);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / reps;
            printf("  %-40s %8.2f us/launch  %6.2f TB/s  frac %.3f   per pivot %6.2f us\n", v.name, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0, us / v.nb);
        }
    }
    return 0;
}
