// Lab bench for the blocked sweep (k_blk_sweep, lp_blocked.hip.h): times variants of the 16-stage body
// on the bench-sized tableau (4096 x 8192 fp64) with HIP events, outside the library, so that a register
// or scheduling idea can be tried in seconds. Not part of the product; build and run:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/_build/sweep_lab tools/lab/sweep_lab.hip
//   gpurun -- tools/_build/sweep_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../xpoly_amd/csrc/lp_blocked.hip.h"
using namespace xpg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// V0: the product body in a kernel of its own
template <int ROWS, int U, int NB, int WAVES> __global__ __launch_bounds__(256, WAVES)
void k_v0(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
          const double * __restrict__ K, LoopState * __restrict__ st)
{
    blk_sweep_body<ROWS, U, NB, false>(tab, m, W, ld, E, K, st);
}

// V2: stages outermost, U rows in lockstep (2U independent add chains)
template <int ROWS, int U, int NB, int WAVES> __global__ __launch_bounds__(256, WAVES)
void k_v2(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
          const double * __restrict__ K, LoopState * __restrict__ st)
{
    const int j = blockIdx.x * 512 + threadIdx.x * 2;
    if (j + 1 >= W) return;
    const int i0 = blockIdx.y * ROWS;
    double2 e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const double2 *>(E + (size_t)s * ld + j);
    double * base = tab + (size_t)i0 * ld + j;
    for (int i = i0; i < i0 + ROWS; i += U) {
        double2 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = *reinterpret_cast<const double2 *>(base + (size_t)u * ld);
#pragma unroll
        for (int s = 0; s < NB; s++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double k = K[(size_t)(i + u) * BLK_MAX + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                a[u].x = a[u].x + p0; a[u].y = a[u].y + p1;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<double2 *>(base + (size_t)u * ld) = a[u];
        base += (size_t)U * ld;
    }
}

// V3: T threads per workgroup, rows per workgroup at run time (grid.y row groups of `rows` rows)
template <int T, int U, int NB> __global__ __launch_bounds__(T)
void k_v3(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
          const double * __restrict__ K, int rows)
{
    const int j = blockIdx.x * (2 * T) + threadIdx.x * 2;
    if (j + 1 >= W) return;
    const int i0 = blockIdx.y * rows;
    const int iend = min(i0 + rows, m);
    double2 e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const double2 *>(E + (size_t)s * ld + j);
    double * base = tab + (size_t)i0 * ld + j;
    int i = i0;
    for (; i + U <= iend; i += U) {
        double2 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = *reinterpret_cast<const double2 *>(base + (size_t)u * ld);
#pragma unroll
        for (int s = 0; s < NB; s++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double k = K[(size_t)(i + u) * BLK_MAX + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                a[u].x = a[u].x + p0; a[u].y = a[u].y + p1;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<double2 *>(base + (size_t)u * ld) = a[u];
        base += (size_t)U * ld;
    }
    for (; i < iend; i++) {
        double2 a = *reinterpret_cast<const double2 *>(base);
#pragma unroll
        for (int s = 0; s < NB; s++) {
            const double k = K[(size_t)i * BLK_MAX + s];
            const double p0 = k * e[s].x, p1 = k * e[s].y;
            a.x = a.x + p0; a.y = a.y + p1;
        }
        *reinterpret_cast<double2 *>(base) = a;
        base += ld;
    }
}

// V4: ping-pong prefetch (the next U rows are in flight while this group is computed), optional
// non-temporal accesses and wave priority raised while loads are being issued.
template <int ROWS, int U, int NB, bool NT, bool PRIO, int WAVES> __global__ __launch_bounds__(256, WAVES)
void k_v4(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
          const double * __restrict__ K, LoopState * __restrict__ st)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const int j = blockIdx.x * 512 + threadIdx.x * 2;
    if (j + 1 >= W) return;
    const int i0 = blockIdx.y * ROWS;
    v2d e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const v2d *>(E + (size_t)s * ld + j);
    double * base = tab + (size_t)i0 * ld + j;
    auto load = [&](v2d (&d)[U], const double * p) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const v2d * q = reinterpret_cast<const v2d *>(p + (size_t)u * ld);
            d[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto apply = [&](v2d (&d)[U], double * p, int row0) {
#pragma unroll
        for (int s = 0; s < NB; s++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double k = K[(size_t)(row0 + u) * BLK_MAX + s];
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
    v2d a[U], b[U];
    load(a, base);
    static_assert(ROWS % (2 * U) == 0, "ROWS must hold whole ping-pong pairs");
#pragma unroll 1
    for (int i = i0; i < i0 + ROWS; i += 2 * U) {
        load(b, base + (size_t)U * ld);
        apply(a, base, i);
        if (i + 2 * U < i0 + ROWS) load(a, base + (size_t)2 * U * ld);
        apply(b, base + (size_t)U * ld, i + U);
        base += (size_t)2 * U * ld;
    }
}

// V5: the arithmetic alone (one load and one store per thread): the ALU floor of the 16-stage body
template <int ROWS, int U, int NB> __global__ __launch_bounds__(256)
void k_alu(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
           const double * __restrict__ K, LoopState * __restrict__ st)
{
    const int j = blockIdx.x * 512 + threadIdx.x * 2;
    if (j + 1 >= W) return;
    const int i0 = blockIdx.y * ROWS;
    double2 e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const double2 *>(E + (size_t)s * ld + j);
    double * base = tab + (size_t)i0 * ld + j;
    double2 a = *reinterpret_cast<const double2 *>(base);
    for (int i = i0; i < i0 + ROWS; i++) {
        const double * kr = K + (size_t)i * BLK_MAX;
#pragma unroll
        for (int s = 0; s < NB; s++) {
            const double k = kr[s];
            const double p0 = k * e[s].x, p1 = k * e[s].y;
            a.x = a.x + p0; a.y = a.y + p1;
        }
    }
    *reinterpret_cast<double2 *>(base) = a;
}

// V6: the memory traffic alone (the same loads and stores, one stage)
template <int ROWS, int U> __global__ __launch_bounds__(256)
void k_mem(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
           const double * __restrict__ K, LoopState * __restrict__ st)
{
    blk_sweep_body<ROWS, U, 1, false>(tab, m, W, ld, E, K, st);
}

typedef void (*kern_t)(double *, int, int, int, const double *, const double *, LoopState *);

int main()
{
    const int m = 4096, W = 8192, ld = 8192;
    double *tab, *E, *K; LoopState * st;
    CK(hipMalloc(&tab, (size_t)m * ld * 8)); CK(hipMalloc(&E, (size_t)16 * ld * 8));
    CK(hipMalloc(&K, (size_t)m * 16 * 8)); CK(hipMalloc(&st, sizeof(LoopState)));
    std::vector<double> h((size_t)m * ld);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 - 0.5; };
    for (auto & v : h) v = rnd();
    CK(hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)16 * ld; i++) h[i] = rnd();
    CK(hipMemcpy(E, h.data(), (size_t)16 * ld * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)m * 16; i++) h[i] = rnd() * 1e-3;
    CK(hipMemcpy(K, h.data(), (size_t)m * 16 * 8, hipMemcpyHostToDevice));
    LoopState hs; memset(&hs, 0, sizeof hs);
    for (int s = 0; s < 16; s++) hs.blk.r[s] = -1;
    CK(hipMemcpy(st, &hs, sizeof hs, hipMemcpyHostToDevice));

    struct V { const char * name; kern_t f; int rows; } vs[] = {
        {"mem-only <32,4> (1 stage)", k_mem<32, 4>, 32},
        {"mem-only <32,8> (1 stage)", k_mem<32, 8>, 32},
        {"alu-only 16 stages", k_alu<32, 4, 16>, 32},
        {"alu-only 8 stages", k_alu<32, 4, 8>, 32},
        {"v0 <32,4,16> waves>=1", k_v0<32, 4, 16, 1>, 32},
        {"v0 <32,2,16>", k_v0<32, 2, 16, 1>, 32},
        {"v0 <32,1,16>", k_v0<32, 1, 16, 1>, 32},
        {"v0 <16,4,16>", k_v0<16, 4, 16, 1>, 16},
        {"v0 <64,4,16>", k_v0<64, 4, 16, 1>, 64},
        {"v0 <32,4,8>", k_v0<32, 4, 8, 1>, 32},
        {"v0 <32,8,8>", k_v0<32, 8, 8, 1>, 32},
        {"v2 <32,2,16> lockstep", k_v2<32, 2, 16, 1>, 32},
        {"v2 <32,4,16> lockstep", k_v2<32, 4, 16, 1>, 32},
        {"v2 <16,2,16> lockstep", k_v2<16, 2, 16, 1>, 16},
        {"v2 <32,2,8> lockstep", k_v2<32, 2, 8, 1>, 32},
        {"v4 <32,2> pingpong", k_v4<32, 2, 16, false, false, 1>, 32},
        {"v4 <32,2> pingpong nt", k_v4<32, 2, 16, true, false, 1>, 32},
        {"v4 <32,2> pingpong prio", k_v4<32, 2, 16, false, true, 1>, 32},
        {"v4 <32,2> pingpong nt prio", k_v4<32, 2, 16, true, true, 1>, 32},
        {"v4 <16,2> pingpong", k_v4<16, 2, 16, false, false, 1>, 16},
        {"v4 <16,2> pingpong nt", k_v4<16, 2, 16, true, false, 1>, 16},
        {"v4 <16,2> pingpong prio", k_v4<16, 2, 16, false, true, 1>, 16},
        {"v4 <32,4> pingpong", k_v4<32, 4, 16, false, false, 1>, 32},
        {"v4 <32,4> pingpong nt prio", k_v4<32, 4, 16, true, true, 1>, 32},
        {"v4 <16,4> pingpong nt prio", k_v4<16, 4, 16, true, true, 1>, 16},
        {"v4 <32,1> pingpong", k_v4<32, 1, 16, false, false, 1>, 32},
        {"v4 <32,1> pingpong prio", k_v4<32, 1, 16, false, true, 1>, 32},
        {"v0 <32,4,16> again", k_v0<32, 4, 16, 1>, 32},
        {"v2 <16,2,16> again", k_v2<16, 2, 16, 1>, 16},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto & v : vs) {
        dim3 g((W + 511) / 512, (m + v.rows - 1) / v.rows);
        for (int w = 0; w < 5; w++) hipLaunchKernelGGL(v.f, g, dim3(256), 0, 0, tab, m, W, ld, E, K, st);
        CK(hipDeviceSynchronize());
        const int reps = 40;
        CK(hipEventRecord(e0, 0));
        for (int w = 0; w < reps; w++) hipLaunchKernelGGL(v.f, g, dim3(256), 0, 0, tab, m, W, ld, E, K, st);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %7.2f us/launch\n", v.name, ms * 1000.0f / reps);
    }
    // occupancy limited through a dynamic LDS request (160 KB per CU): o workgroups of 4 waves per CU
    struct VO { const char * name; kern_t f; int rows; } vos[] = {
        {"v0 <32,4,16>", k_v0<32, 4, 16, 1>, 32},
        {"v4 <32,4> pingpong", k_v4<32, 4, 16, false, false, 1>, 32},
        {"v4 <16,2> pingpong", k_v4<16, 2, 16, false, false, 1>, 16},
        {"v4 <16,4> pingpong", k_v4<16, 4, 16, false, false, 1>, 16},
        {"v4 <8,2> pingpong", k_v4<8, 2, 16, false, false, 1>, 8},
        {"v4 <8,4> pingpong", k_v4<8, 4, 16, false, false, 1>, 8},
    };
    for (auto & v : vos) {
        for (int o : {3, 4, 5, 8}) {
            const unsigned lds = o >= 8 ? 0 : (160 * 1024 / o - 1024);
            dim3 g((W + 511) / 512, (m + v.rows - 1) / v.rows);
            for (int w = 0; w < 5; w++) hipLaunchKernelGGL(v.f, g, dim3(256), lds, 0, tab, m, W, ld, E, K, st);
            CK(hipDeviceSynchronize());
            const int reps = 40;
            CK(hipEventRecord(e0, 0));
            for (int w = 0; w < reps; w++) hipLaunchKernelGGL(v.f, g, dim3(256), lds, 0, tab, m, W, ld, E, K, st);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-22s max %d wg/cu  %7.2f us/launch\n", v.name, o, ms * 1000.0f / reps);
        }
    }
    // the product kernels themselves, on a full batch whose 16 pivot rows are spread over the tableau
    {
        hs.status = ST_RUNNING; hs.blk.batch = 7; hs.blk.n = 16;
        for (int s = 0; s < 16; s++) hs.blk.r[s] = s * 251 + 5;
        CK(hipMemcpy(st, &hs, sizeof hs, hipMemcpyHostToDevice));
        typedef void (*kernp_t)(double *, int, int, int, const double *, const double *, const LoopState *, int);
        struct VP { const char * name; kernp_t f; int rows; } vps[] = {
            {"full <32,4> pingpong", k_blk_sweep_full<32, 4>, 32},
            {"full <16,4> pingpong", k_blk_sweep_full<16, 4>, 16},
            {"full <16,2> pingpong", k_blk_sweep_full<16, 2>, 16},
            {"full <8,2> pingpong", k_blk_sweep_full<8, 2>, 8},
            {"full <8,4> pingpong", k_blk_sweep_full<8, 4>, 8},
            {"full <32,2> pingpong", k_blk_sweep_full<32, 2>, 32},
        };
        for (int pass = 0; pass < 2; pass++)
        for (auto & v : vps) {
            dim3 g((W + 511) / 512, (m + v.rows - 1) / v.rows), g32((W + 511) / 512, (m + 31) / 32);
            for (int w = 0; w < 5; w++) hipLaunchKernelGGL(v.f, g, dim3(256), 0, 0, tab, m, W, ld, E, K, st, 7);
            CK(hipDeviceSynchronize());
            const int reps = 40;
            CK(hipEventRecord(e0, 0));
            for (int w = 0; w < reps; w++) {
                hipLaunchKernelGGL(v.f, g, dim3(256), 0, 0, tab, m, W, ld, E, K, st, 7);
                (void)g32;
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-30s %7.2f us/launch\n", v.name, ms * 1000.0f / reps);
        }
        dim3 g((W + 511) / 512, (m + 31) / 32);
        for (int w = 0; w < 5; w++) hipLaunchKernelGGL((k_blk_sweep<32, 4, 16>), g, dim3(256), 0, 0, tab, m, W, ld, E, K, st, 7);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int w = 0; w < 40; w++) hipLaunchKernelGGL((k_blk_sweep<32, 4, 16>), g, dim3(256), 0, 0, tab, m, W, ld, E, K, st, 7);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-30s %7.2f us/launch\n", "switch kernel <32,4,16>", ms * 1000.0f / 40);
        hs.status = 0; hs.blk.n = 0;
        for (int s = 0; s < 16; s++) hs.blk.r[s] = -1;
        CK(hipMemcpy(st, &hs, sizeof hs, hipMemcpyHostToDevice));
    }
    typedef void (*kern3_t)(double *, int, int, int, const double *, const double *, int);
    struct V3 { const char * name; kern3_t f; int threads; } v3s[] = {
        {"v3 T=256 U=2", k_v3<256, 2, 16>, 256}, {"v3 T=128 U=2", k_v3<128, 2, 16>, 128},
        {"v3 T=64 U=2", k_v3<64, 2, 16>, 64},    {"v3 T=128 U=4", k_v3<128, 4, 16>, 128},
        {"v3 T=64 U=4", k_v3<64, 4, 16>, 64},
    };
    const int groups[] = {64, 80, 96, 128, 160, 205, 256, 320, 512};
    for (auto & v : v3s) {
        if (!getenv("LAB_V3")) break;
        for (int G : groups) {
            const int rows = (m + G - 1) / G;
            dim3 g((W + 2 * v.threads - 1) / (2 * v.threads), (m + rows - 1) / rows);
            for (int w = 0; w < 5; w++) hipLaunchKernelGGL(v.f, g, dim3(v.threads), 0, 0, tab, m, W, ld, E, K, rows);
            CK(hipDeviceSynchronize());
            const int reps = 40;
            CK(hipEventRecord(e0, 0));
            for (int w = 0; w < reps; w++) hipLaunchKernelGGL(v.f, g, dim3(v.threads), 0, 0, tab, m, W, ld, E, K, rows);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-14s rows/wg=%3d wgs=%5d  %7.2f us/launch\n", v.name, rows, g.x * g.y, ms * 1000.0f / reps);
        }
    }
    return 0;
}
