// Lab bench, round 5: can a pass apply 32 staged pivots at (nearly) the price of 24? Variants of the blocked sweep's body at
// 4096 x 8192 (268 MB), natural tile order with alternating direction, HIP events, outside the library:
//   pair   -- the product's shape: a thread owns a column PAIR (16-byte accesses), NB register pairs of e_s
//   single -- a thread owns ONE column (8-byte accesses, a wave still covers 512 contiguous bytes): half the e registers
//   twoph  -- pairs, but the 2 * NH stages in two phases over the same in-register rows: e_s reloaded (from the L2) in between
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/_build/sweep_lab3 tools/lab/sweep_lab3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tile_of(int rev, int & bx, int & by)
{
    const int gx = (int)gridDim.x, gy = (int)gridDim.y;
    int lid = (int)blockIdx.y * gx + (int)blockIdx.x;
    if (rev) lid = gx * gy - 1 - lid;
    by = lid / gx; bx = lid % gx;
}

template <int ROWS, int U, int NB, int WAVES> __global__ __launch_bounds__(256, WAVES)
void k_pair(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    int bx, by; tile_of(rev, bx, by);
    const int j = bx * 512 + threadIdx.x * 2, i0 = by * ROWS;
    if (j >= W) return;
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
    double * base = tab + (size_t)i0 * ld + j;
    v2d a[U], b[U];
    auto load = [&](v2d (&d)[U], const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = *reinterpret_cast<const v2d *>(p + (size_t)u * ld);
    };
    auto apply = [&](v2d (&d)[U], double * p, int row0) {
#pragma unroll
        for (int s = 0; s < NB; s++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double k = K[(size_t)(row0 + u) * 32 + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                d[u].x = d[u].x + p0; d[u].y = d[u].y + p1;
            }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<v2d *>(p + (size_t)u * ld) = d[u];
    };
    load(a, base);
#pragma unroll 1
    for (int i = i0; i < i0 + ROWS; i += 2 * U) {
        load(b, base + (size_t)U * ld);
        apply(a, base, i);
        if (i + 2 * U < i0 + ROWS) load(a, base + (size_t)2 * U * ld);
        apply(b, base + (size_t)U * ld, i + U);
        base += (size_t)2 * U * ld;
    }
}

template <int ROWS, int U, int NB, int WAVES> __global__ __launch_bounds__(256, WAVES)
void k_single(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    int bx, by; tile_of(rev, bx, by);
    const int j = bx * 256 + threadIdx.x, i0 = by * ROWS;
    if (j >= W) return;
    double e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = E[(size_t)s * ld + j];
    double * base = tab + (size_t)i0 * ld + j;
    double a[U], b[U];
    auto load = [&](double (&d)[U], const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = p[(size_t)u * ld];
    };
    auto apply = [&](double (&d)[U], double * p, int row0) {
#pragma unroll
        for (int s = 0; s < NB; s++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double k = K[(size_t)(row0 + u) * 32 + s];
                const double p0 = k * e[s];
                d[u] = d[u] + p0;
            }
#pragma unroll
        for (int u = 0; u < U; u++) p[(size_t)u * ld] = d[u];
    };
    load(a, base);
#pragma unroll 1
    for (int i = i0; i < i0 + ROWS; i += 2 * U) {
        load(b, base + (size_t)U * ld);
        apply(a, base, i);
        if (i + 2 * U < i0 + ROWS) load(a, base + (size_t)2 * U * ld);
        apply(b, base + (size_t)U * ld, i + U);
        base += (size_t)2 * U * ld;
    }
}

// 2 * NH stages in two phases: all ROWS rows of the tile in registers, e_s for one half at a time
template <int ROWS, int NH, int WAVES> __global__ __launch_bounds__(256, WAVES)
void k_twoph(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    int bx, by; tile_of(rev, bx, by);
    const int j = bx * 512 + threadIdx.x * 2, i0 = by * ROWS;
    if (j >= W) return;
    double * base = tab + (size_t)i0 * ld + j;
    v2d a[ROWS];
#pragma unroll
    for (int u = 0; u < ROWS; u++) a[u] = *reinterpret_cast<const v2d *>(base + (size_t)u * ld);
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        v2d e[NH];
#pragma unroll
        for (int s = 0; s < NH; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)(h * NH + s) * ld + j);
#pragma unroll
        for (int u = 0; u < ROWS; u++)
#pragma unroll
            for (int s = 0; s < NH; s++) {
                const double k = K[(size_t)(i0 + u) * 32 + h * NH + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                a[u].x = a[u].x + p0; a[u].y = a[u].y + p1;
            }
    }
#pragma unroll
    for (int u = 0; u < ROWS; u++) *reinterpret_cast<v2d *>(base + (size_t)u * ld) = a[u];
}

// KMODE 0: the row's k_s through the scalar cache (the product's); 1: no load at all (a ceiling: k from a kernel argument's
// bits and the stage number); 2: the tile's ROWS x NB block of K staged in LDS once, read back as broadcasts
template <int ROWS, int U, int NB, int KMODE> __global__ __launch_bounds__(256)
void k_pairk(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    __shared__ double ks[KMODE == 2 ? ROWS * NB : 1];
    int bx, by; tile_of(rev, bx, by);
    const int j = bx * 512 + threadIdx.x * 2, i0 = by * ROWS;
    if (KMODE == 2) {
        for (int q = threadIdx.x; q < ROWS * NB; q += 256) ks[q] = K[(size_t)(i0 + q / NB) * 32 + q % NB];
        __syncthreads();
    }
    if (j >= W) return;
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
    double * base = tab + (size_t)i0 * ld + j;
    v2d a[U], b[U];
    double kc[U][8];                                        // KMODE 1: sixteen k loaded ONCE (distinct per row, so that no product is shared)
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int q = 0; q < 8; q++) kc[u][q] = K[(size_t)(rev * 64 + u) * 32 + q];
    auto load = [&](v2d (&d)[U], const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = *reinterpret_cast<const v2d *>(p + (size_t)u * ld);
    };
    auto apply = [&](v2d (&d)[U], double * p, int row0) {
#pragma unroll
        for (int s = 0; s < NB; s++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                double k;
                if (KMODE == 0) k = K[(size_t)(row0 + u) * 32 + s];
                else if (KMODE == 1) k = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, kc[u][s & 7]) ^ ((unsigned long long)(row0 & 4) << 61));   // (one scalar xor per k: differs from row pair to row pair)
                else k = ks[(row0 - i0 + u) * NB + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                d[u].x = d[u].x + p0; d[u].y = d[u].y + p1;
            }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<v2d *>(p + (size_t)u * ld) = d[u];
    };
    load(a, base);
#pragma unroll 1
    for (int i = i0; i < i0 + ROWS; i += 2 * U) {
        load(b, base + (size_t)U * ld);
        apply(a, base, i);
        if (i + 2 * U < i0 + ROWS) load(a, base + (size_t)2 * U * ld);
        apply(b, base + (size_t)U * ld, i + U);
        base += (size_t)2 * U * ld;
    }
}

// a ring of D row groups in flight: the rows of group g + D - 1 are requested before the arithmetic of group g (the
// product's pass is D = 2); the row loop fully unrolled so that the ring's indices are static
template <int ROWS, int U, int NB, int D> __global__ __launch_bounds__(256)
void k_ring(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    constexpr int NG = ROWS / U;
    int bx, by; tile_of(rev, bx, by);
    const int j = bx * 512 + threadIdx.x * 2, i0 = by * ROWS;
    if (j >= W) return;
    double * base = tab + (size_t)i0 * ld + j;
    v2d r[D][U];
#pragma unroll
    for (int g = 0; g < D - 1; g++)
#pragma unroll
        for (int u = 0; u < U; u++) r[g][u] = *reinterpret_cast<const v2d *>(base + (size_t)(g * U + u) * ld);
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g + D - 1 < NG) {
#pragma unroll
            for (int u = 0; u < U; u++) r[(g + D - 1) % D][u] = *reinterpret_cast<const v2d *>(base + (size_t)((g + D - 1) * U + u) * ld);
        }
#pragma unroll
        for (int s = 0; s < NB; s++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double k = K[(size_t)(i0 + g * U + u) * 32 + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                r[g % D][u].x = r[g % D][u].x + p0; r[g % D][u].y = r[g % D][u].y + p1;
            }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<v2d *>(base + (size_t)(g * U + u) * ld) = r[g % D][u];
    }
}

__device__ unsigned long long g_clk[4];                    // one workgroup's (s_memtime, 100 MHz wall clock) at its start and end
// the k_s of a row pair in groups of G stages through two sets of scalar registers: the loads of group g + 1 are requested
// before the arithmetic of group g (the compiler, left alone, requests a group where its registers were last used and
// waits for it on the spot)
template <int ROWS, int NB, int G> __global__ __launch_bounds__(256)
void k_pipe(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    constexpr int U = 2, NG = NB / G;
    int bx, by; tile_of(rev, bx, by);
    const int j = bx * 512 + threadIdx.x * 2, i0 = by * ROWS;
    if (j >= W) return;
    const bool probe = blockIdx.x == 7 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0;
    if (probe) { g_clk[0] = __builtin_readcyclecounter(); g_clk[1] = wall_clock64(); }
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
    double * base = tab + (size_t)i0 * ld + j;
    v2d a[U], b[U];
    auto load = [&](v2d (&d)[U], const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = *reinterpret_cast<const v2d *>(p + (size_t)u * ld);
    };
    double kq[2][U][G];
    auto kload = [&](int set, int row0, int g) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < G; q++) kq[set][u][q] = K[(size_t)(row0 + u) * 32 + g * G + q];
    };
    // group 0 of (row0) is in set 0 on entry; leaves group 0 of (next_row0) in set 0
    auto apply = [&](v2d (&d)[U], double * p, int row0, int next_row0) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
            // scalar loads return out of order: the only wait there is is "all of them". Asking for group g's registers HERE
            // puts that wait in front of the requests of group g + 1, which then have the arithmetic of group g to arrive in
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int q = 0; q < G; q += 4)
                    asm volatile("" :: "s"(kq[g & 1][u][q]), "s"(kq[g & 1][u][q + 1]), "s"(kq[g & 1][u][q + 2]), "s"(kq[g & 1][u][q + 3]));
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) kload((g + 1) & 1, row0, g + 1); else kload((g + 1) & 1, next_row0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; q++)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const double k = kq[g & 1][u][q];
                    const double p0 = k * e[g * G + q].x, p1 = k * e[g * G + q].y;
                    d[u].x = d[u].x + p0; d[u].y = d[u].y + p1;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<v2d *>(p + (size_t)u * ld) = d[u];
    };
    static_assert(NG % 2 == 0, "an even number of groups: a row pair starts and ends in set 0");
    load(a, base);
    kload(0, i0, 0);
#pragma unroll 1
    for (int i = i0; i < i0 + ROWS; i += 2 * U) {
        load(b, base + (size_t)U * ld);
        apply(a, base, i, i + U);
        if (i + 2 * U < i0 + ROWS) load(a, base + (size_t)2 * U * ld);
        apply(b, base + (size_t)U * ld, i + U, i + 2 * U < m ? i + 2 * U : i);
        base += (size_t)2 * U * ld;
    }
    if (probe) { g_clk[2] = __builtin_readcyclecounter(); g_clk[3] = wall_clock64(); }
}

// `pipe` with ONE column per thread: half the e registers (more waves per SIMD), U rows share a group of scalar k_s
template <int ROWS, int U, int NB, int G> __global__ __launch_bounds__(256)
void k_pipe1(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    constexpr int NG = NB / G;
    int bx, by; tile_of(rev, bx, by);
    const int j = bx * 256 + threadIdx.x, i0 = by * ROWS;
    if (j >= W) return;
    double e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = E[(size_t)s * ld + j];
    double * base = tab + (size_t)i0 * ld + j;
    double a[U], b[U];
    auto load = [&](double (&d)[U], const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = p[(size_t)u * ld];
    };
    double kq[2][U][G];
    auto kload = [&](int set, int row0, int g) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < G; q++) kq[set][u][q] = K[(size_t)(row0 + u) * 32 + g * G + q];
    };
    auto apply = [&](double (&d)[U], double * p, int row0, int next_row0) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int q = 0; q < G; q += 4)
                    asm volatile("" :: "s"(kq[g & 1][u][q]), "s"(kq[g & 1][u][q + 1]), "s"(kq[g & 1][u][q + 2]), "s"(kq[g & 1][u][q + 3]));
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) kload((g + 1) & 1, row0, g + 1); else kload((g + 1) & 1, next_row0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; q++)
#pragma unroll
                for (int u = 0; u < U; u++) { const double p0 = kq[g & 1][u][q] * e[g * G + q]; d[u] = d[u] + p0; }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) p[(size_t)u * ld] = d[u];
    };
    static_assert(NG % 2 == 0, "an even number of groups");
    load(a, base);
    kload(0, i0, 0);
#pragma unroll 1
    for (int i = i0; i < i0 + ROWS; i += 2 * U) {
        load(b, base + (size_t)U * ld);
        apply(a, base, i, i + U);
        if (i + 2 * U < i0 + ROWS) load(a, base + (size_t)2 * U * ld);
        apply(b, base + (size_t)U * ld, i + U, i + 2 * U < m ? i + 2 * U : i);
        base += (size_t)2 * U * ld;
    }
}

// `pipe` as a RESIDENT workgroup: one workgroup per (column strip, chunk of rows), as many as the device seats at once; the 32
// e_s pairs are loaded once per workgroup instead of once per 16 rows (as many load instructions as the rows themselves), and no
// workgroup is dispatched behind another. gridDim.x = strips, gridDim.y = chunks; rows of chunk c: [c * rpc, (c + 1) * rpc)
template <int NB, int G> __global__ __launch_bounds__(256)
void k_resident(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    constexpr int U = 2, NG = NB / G;
    const int bx = blockIdx.x, c = blockIdx.y;
    const int rpc = ((m + (int)gridDim.y - 1) / (int)gridDim.y + 3) / 4 * 4;          // rows per chunk, a multiple of 2 U
    const int j = bx * 512 + threadIdx.x * 2;
    int i0 = c * rpc, iend = min(i0 + rpc, m);
    if (j >= W || i0 >= iend) return;
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
    // alternate passes walk the chunk in opposite directions (what the Infinity Cache still holds of it is read first)
    const long long step = rev ? -(long long)ld : (long long)ld;
    const int dir = rev ? -1 : 1;
    int i = rev ? iend - 1 : i0;                            // first row; rows i, i + dir, ...
    const int nrows = iend - i0;                            // a multiple of 4 except in the last chunk (m a multiple of 4 here)
    double * base = tab + (size_t)i * ld + j;
    v2d a[U], b[U];
    auto load = [&](v2d (&d)[U], const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = *reinterpret_cast<const v2d *>(p + u * step);
    };
    double kq[2][U][G];
    auto kload = [&](int set, int row0, int g) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < G; q++) kq[set][u][q] = K[(size_t)(row0 + u * dir) * 32 + g * G + q];
    };
    auto apply = [&](v2d (&d)[U], double * p, int row0, int next_row0) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int q = 0; q < G; q += 4)
                    asm volatile("" :: "s"(kq[g & 1][u][q]), "s"(kq[g & 1][u][q + 1]), "s"(kq[g & 1][u][q + 2]), "s"(kq[g & 1][u][q + 3]));
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) kload((g + 1) & 1, row0, g + 1); else kload((g + 1) & 1, next_row0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; q++)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const double k = kq[g & 1][u][q];
                    const double p0 = k * e[g * G + q].x, p1 = k * e[g * G + q].y;
                    d[u].x = d[u].x + p0; d[u].y = d[u].y + p1;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<v2d *>(p + u * step) = d[u];
    };
    load(a, base);
    kload(0, i, 0);
#pragma unroll 1
    for (int n = 0; n < nrows; n += 2 * U) {
        load(b, base + U * step);
        apply(a, base, i, i + U * dir);
        const bool more = n + 2 * U < nrows;
        if (more) load(a, base + 2 * U * step);
        apply(b, base + U * step, i + U * dir, more ? i + 2 * U * dir : i);
        base += 2 * U * step; i += 2 * U * dir;
    }
}

// `resident` with THREE row groups in flight (the rows two groups ahead are requested before a group's arithmetic)
template <int NB, int G> __global__ __launch_bounds__(256)
void k_resident3(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    constexpr int U = 2, NG = NB / G;
    const int bx = blockIdx.x, c = blockIdx.y;
    const int rpc = ((m + (int)gridDim.y - 1) / (int)gridDim.y + 3) / 4 * 4;
    const int j = bx * 512 + threadIdx.x * 2;
    const int i0 = c * rpc, iend = min(i0 + rpc, m);
    if (j >= W || i0 >= iend) return;
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
    const int ngroups = (iend - i0) / U;
    double * base = tab + (size_t)i0 * ld + j;
    v2d a[U], b[U], c3[U];
    auto load = [&](v2d (&d)[U], int g) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = *reinterpret_cast<const v2d *>(base + (size_t)(g * U + u) * ld);
    };
    double kq[2][U][G];
    auto kload = [&](int set, int row0, int g) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < G; q++) kq[set][u][q] = K[(size_t)(row0 + u) * 32 + g * G + q];
    };
    auto apply = [&](v2d (&d)[U], int grp) {
        const int row0 = i0 + grp * U, next_row0 = grp + 1 < ngroups ? row0 + U : i0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int q = 0; q < G; q += 4)
                    asm volatile("" :: "s"(kq[g & 1][u][q]), "s"(kq[g & 1][u][q + 1]), "s"(kq[g & 1][u][q + 2]), "s"(kq[g & 1][u][q + 3]));
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) kload((g + 1) & 1, row0, g + 1); else kload((g + 1) & 1, next_row0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; q++)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const double k = kq[g & 1][u][q];
                    const double p0 = k * e[g * G + q].x, p1 = k * e[g * G + q].y;
                    d[u].x = d[u].x + p0; d[u].y = d[u].y + p1;
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<v2d *>(base + (size_t)(grp * U + u) * ld) = d[u];
    };
    load(a, 0);
    if (ngroups > 1) load(b, 1);
    kload(0, i0, 0);
#pragma unroll 1
    for (int n = 0; n < ngroups; n += 3) {
        if (n + 2 < ngroups) load(c3, n + 2);
        apply(a, n);
        if (n + 1 >= ngroups) break;
        if (n + 3 < ngroups) load(a, n + 3);
        apply(b, n + 1);
        if (n + 2 >= ngroups) break;
        if (n + 4 < ngroups) load(b, n + 4);
        apply(c3, n + 2);
    }
}

typedef void (*kern_t)(double *, int, int, int, const double *, const double *, int);
// round 6: more bytes in flight without more registers -- the tableau rows of a workgroup come through an LDS ring filled by
// LDS-DMA loads (global_load_lds_dwordx4: no VGPR holds a row before it is needed), D row groups of U rows ahead of the arithmetic.
// At 3 waves per SIMD the register version keeps 24 KB per CU in flight (U = 2: one group ahead); the ring adds D * U KB per wave.
template <int ROWS, int U, int NB, int D> __global__ __launch_bounds__(256)
void k_ldsring(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E, const double * __restrict__ K, int rev)
{
    __shared__ __attribute__((aligned(16))) double ring[D][U][512];
    int bx, by; tile_of(rev, bx, by);
    const int j = bx * 512 + threadIdx.x * 2, i0 = by * ROWS;
    if (j >= W) return;
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
    double * base = tab + (size_t)i0 * ld + j;
    const int wb = (int)(threadIdx.x & ~63u) * 2;                  // this wave's 128 doubles of a ring row
    constexpr int NG = ROWS / U;
    auto issue = [&](int slot, const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(p + (size_t)u * ld),
                                             (void __attribute__((address_space(3))) *)(&ring[slot][u][wb]), 16, 0, 0);
    };
#pragma unroll
    for (int g = 0; g < D && g < NG; g++) issue(g, base + (size_t)g * U * ld);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int slot = g % D;
        // the group's U loads are the oldest outstanding LDS-DMA loads: everything younger may stay in flight. Stores are counted
        // by vmcnt as well (U per group behind the loads): (min(D, NG - g) - 1) groups of loads + the stores issued since
        constexpr int dummy = 0; (void)dummy;
        const int younger_loads = ((NG - g) < D ? (NG - g) : D) - 1;
        // stores of the groups g - 1 .. : all issued BEFORE the loads of group g + D - 1?  order of issue per iteration: [wait][read LDS][issue loads g + D][arith][stores g]
        // outstanding at the wait of group g: loads g .. g + younger, stores of groups <= g - 1 (older than the loads issued in iterations >= g - 1 ... interleaved)
        // conservative and simple: wait until at most (younger_loads * U) + 0 remain -- stores older than those loads are then complete too
        if (younger_loads * U == 0) __builtin_amdgcn_s_waitcnt(0x0f70 | 0);         // vmcnt(0)   (gfx9 encoding: vmcnt[3:0] | expcnt | lgkmcnt; high bits 15:14 = vmcnt[5:4])
        else if (younger_loads * U == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | 1);
        else if (younger_loads * U == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | 2);
        else if (younger_loads * U == 3) __builtin_amdgcn_s_waitcnt(0x0f70 | 3);
        else if (younger_loads * U == 4) __builtin_amdgcn_s_waitcnt(0x0f70 | 4);
        else if (younger_loads * U == 6) __builtin_amdgcn_s_waitcnt(0x0f70 | 6);
        else if (younger_loads * U == 8) __builtin_amdgcn_s_waitcnt(0x0f70 | 8);
        else if (younger_loads * U == 10) __builtin_amdgcn_s_waitcnt(0x0f70 | 10);
        else if (younger_loads * U == 12) __builtin_amdgcn_s_waitcnt(0x0f70 | 12);
        else if (younger_loads * U == 14) __builtin_amdgcn_s_waitcnt(0x0f70 | 14);
        else __builtin_amdgcn_s_waitcnt(0x0f70 | 0);
        v2d d[U];
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = *reinterpret_cast<const v2d *>(&ring[slot][u][threadIdx.x * 2]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (g + D < NG) issue(slot, base + (size_t)(g + D) * U * ld);
#pragma unroll
        for (int s = 0; s < NB; s++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double k = K[(size_t)(i0 + g * U + u) * 32 + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                d[u].x = d[u].x + p0; d[u].y = d[u].y + p1;
            }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<v2d *>(base + (size_t)(g * U + u) * ld) = d[u];
    }
}

int main(int argc, char ** argv)
{
    const char * only = argc > 1 ? argv[1] : nullptr;            // run the variants whose name contains this
    const int m = getenv("LAB_M") ? atoi(getenv("LAB_M")) : 4096, W = getenv("LAB_W") ? atoi(getenv("LAB_W")) : 8192, ld = getenv("LAB_LD") ? atoi(getenv("LAB_LD")) : W;
    printf("tableau %d x %d (ld %d): %.0f MB\n", m, W, ld, (double)m * ld * 8 / 1e6);
    double *tab, *E, *K;
    CK(hipMalloc(&tab, (size_t)m * ld * 8)); CK(hipMalloc(&E, (size_t)32 * ld * 8)); CK(hipMalloc(&K, (size_t)m * 32 * 8));
    {
        std::vector<double> h((size_t)m * ld);
        unsigned long long x = 88172645463325252ull;
        auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 - 0.5; };
        for (auto & v : h) v = rnd();
        CK(hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < (size_t)32 * ld; i++) h[i] = rnd();
        CK(hipMemcpy(E, h.data(), (size_t)32 * ld * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < (size_t)m * 32; i++) h[i] = rnd() * 1e-3;
        CK(hipMemcpy(K, h.data(), (size_t)m * 32 * 8, hipMemcpyHostToDevice));
    }
    struct V { const char * name; kern_t f; int rows, cols, nb; } vs[] = {
        {"pair   NB=24 <16,2> (the product's)", k_pair<16, 2, 24, 1>, 16, 512, 24},
        {"pair   NB=32 <16,2>", k_pair<16, 2, 32, 1>, 16, 512, 32},
        {"ldsring NB=32 <16,2> D=2", k_ldsring<16, 2, 32, 2>, 16, 512, 32},
        {"ldsring NB=32 <16,2> D=4", k_ldsring<16, 2, 32, 4>, 16, 512, 32},
        {"ldsring NB=32 <16,1> D=8", k_ldsring<16, 1, 32, 8>, 16, 512, 32},
        {"ldsring NB=32 <32,2> D=4", k_ldsring<32, 2, 32, 4>, 32, 512, 32},
        {"ldsring NB=32 <32,2> D=6", k_ldsring<32, 2, 32, 6>, 32, 512, 32},
        {"ldsring NB=16 <16,2> D=4", k_ldsring<16, 2, 16, 4>, 16, 512, 16},
        {"ring   NB=32 U=2 D=2 (unrolled product)", k_ring<16, 2, 32, 2>, 16, 512, 32},
        {"ring   NB=32 U=2 D=3", k_ring<16, 2, 32, 3>, 16, 512, 32},
        {"ring   NB=32 U=2 D=4", k_ring<16, 2, 32, 4>, 16, 512, 32},
        {"ring   NB=32 U=1 D=4", k_ring<16, 1, 32, 4>, 16, 512, 32},
        {"ring   NB=32 U=1 D=6", k_ring<16, 1, 32, 6>, 16, 512, 32},
        {"ring   NB=32 U=2 D=3 32 rows", k_ring<32, 2, 32, 3>, 32, 512, 32},
        {"ring   NB=24 U=2 D=3", k_ring<16, 2, 24, 3>, 16, 512, 24},
        {"ring   NB=24 U=2 D=4", k_ring<16, 2, 24, 4>, 16, 512, 24},
        {"ring   NB=24 U=1 D=3", k_ring<16, 1, 24, 3>, 16, 512, 24},
        {"pipe   NB=32 groups of 8", k_pipe<16, 32, 8>, 16, 512, 32},
        {"resident NB=32, 48 chunks (768 wgs)", k_resident<32, 8>, -48, 512, 32},
        {"resident3 NB=32, 48 chunks (768 wgs)", k_resident3<32, 8>, -48, 512, 32},
        {"resident3 NB=32, 43 chunks of 96 rows", k_resident3<32, 8>, -43, 512, 32},
        {"resident3 NB=24, 64 chunks (1024 wgs)", k_resident3<24, 4>, -64, 512, 24},
        {"resident NB=32, 32 chunks (512 wgs)", k_resident<32, 8>, -32, 512, 32},
        {"resident NB=32, 64 chunks (1024 wgs)", k_resident<32, 8>, -64, 512, 32},
        {"resident NB=32, 96 chunks (1536 wgs)", k_resident<32, 8>, -96, 512, 32},
        {"resident NB=24, 64 chunks (1024 wgs)", k_resident<24, 4>, -64, 512, 24},
        {"pipe   NB=32 groups of 8, 32 rows", k_pipe<32, 32, 8>, 32, 512, 32},
        {"pipe1  NB=32 U=4 G=4 (one column)", k_pipe1<16, 4, 32, 4>, 16, 256, 32},
        {"pipe1  NB=32 U=4 G=4 32 rows", k_pipe1<32, 4, 32, 4>, 32, 256, 32},
        {"pipe1  NB=32 U=2 G=8", k_pipe1<16, 2, 32, 8>, 16, 256, 32},
        {"pipe1  NB=32 U=8 G=2... (U=8 G=4 = 128 sgprs) skip", k_pipe1<32, 2, 32, 8>, 32, 256, 32},
        {"pipe   NB=32 groups of 4", k_pipe<16, 32, 4>, 16, 512, 32},
        {"pipe   NB=24 groups of 4", k_pipe<16, 24, 4>, 16, 512, 24},
        {"pairk  NB=24 scalar loads", k_pairk<16, 2, 24, 0>, 16, 512, 24},
        {"pairk  NB=24 no k loads (ceiling)", k_pairk<16, 2, 24, 1>, 16, 512, 24},
        {"pairk  NB=24 k from LDS", k_pairk<16, 2, 24, 2>, 16, 512, 24},
        {"pairk  NB=32 scalar loads", k_pairk<16, 2, 32, 0>, 16, 512, 32},
        {"pairk  NB=32 no k loads (ceiling)", k_pairk<16, 2, 32, 1>, 16, 512, 32},
        {"pairk  NB=32 k from LDS", k_pairk<16, 2, 32, 2>, 16, 512, 32},
        {"pairk  NB=32 k from LDS, 32 rows", k_pairk<32, 2, 32, 2>, 32, 512, 32},
        {"pair   NB=32 <16,2> 3 waves", k_pair<16, 2, 32, 3>, 16, 512, 32},
        {"pair   NB=32 <16,1>", k_pair<16, 1, 32, 1>, 16, 512, 32},
        {"pair   NB=28 <16,2>", k_pair<16, 2, 28, 1>, 16, 512, 28},
        {"single NB=32 <16,2>", k_single<16, 2, 32, 1>, 16, 256, 32},
        {"single NB=32 <16,4>", k_single<16, 4, 32, 1>, 16, 256, 32},
        {"single NB=32 <32,4>", k_single<32, 4, 32, 1>, 32, 256, 32},
        {"single NB=32 <16,4> 6 waves", k_single<16, 4, 32, 6>, 16, 256, 32},
        {"single NB=24 <16,4>", k_single<16, 4, 24, 1>, 16, 256, 24},
        {"twoph  2 x 16, 8 rows", k_twoph<8, 16, 1>, 8, 512, 32},
        {"twoph  2 x 16, 16 rows", k_twoph<16, 16, 1>, 16, 512, 32},
        {"twoph  2 x 16, 4 rows", k_twoph<4, 16, 1>, 4, 512, 32},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * m * W * 8;
    // the clocks take their time to come up from idle: the first variants of a run were measured 20 % slower than the SAME
    // code later in the list (identical ISA and descriptor: `pair NB=32 <16,2>` 132 us second in the list, 110 us twelfth)
    // -- half a second of the first variant before anything is timed
    {
        hipEventRecord(e0, 0);
        float ms = 0.f; int flip = 0;
        while (ms < 500.f) {
            for (int w = 0; w < 50; w++) hipLaunchKernelGGL(vs[0].f, dim3((W + vs[0].cols - 1) / vs[0].cols, m / vs[0].rows), dim3(256), 0, 0, tab, m, W, ld, E, K, (flip ^= 1));
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
    }
    for (auto & v : vs) {
        if (only && !strstr(v.name, only)) continue;
        dim3 g((W + v.cols - 1) / v.cols, v.rows < 0 ? -v.rows : m / v.rows);
        int flip = 0;
        for (int w = 0; w < 6; w++) hipLaunchKernelGGL(v.f, g, dim3(256), 0, 0, tab, m, W, ld, E, K, (flip ^= 1));
        CK(hipDeviceSynchronize());
        const int reps = 100;
        CK(hipEventRecord(e0, 0));
        for (int w = 0; w < reps; w++) hipLaunchKernelGGL(v.f, g, dim3(256), 0, 0, tab, m, W, ld, E, K, (flip ^= 1));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / reps;
        hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void *)v.f));
        if (strstr(v.name, "pipe   ")) {
            unsigned long long c[4]; CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof c));
            printf("    (one workgroup of the last launch: %.2f us from its first to its last instruction at %.0f MHz)\n", (c[3] - c[1]) * 0.01, (double)(c[2] - c[0]) / (double)(c[3] - c[1]) * 100.0);
        }
        printf("  %-38s %8.2f us/launch  frac %.3f   per pivot %5.2f us   %3d VGPRs %4zu B scratch\n", v.name, us, bytes / us / 1e6 / 8.0, us / v.nb, fa.numRegs, (size_t)fa.localSizeBytes);
    }
    return 0;
}
