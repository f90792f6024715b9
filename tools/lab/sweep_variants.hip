// GPU micro-benchmark (not product code): in-place vs ping-pong rank-1 sweep on a 4096 x 8192 fp64
// tableau, to decide whether an out-of-place pipeline could pay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#pragma clang fp contract(off)
template <int ROWS, int UNROLL> __global__ __launch_bounds__(256)
void sweep(const double * __restrict__ src, double * __restrict__ dst, int m, int W, int ld,
           const double * __restrict__ rowbuf, const double * __restrict__ colbuf)
{
    const int j = blockIdx.x * 512 + threadIdx.x * 2;
    if (j >= W) return;
    const int i0 = blockIdx.y * ROWS;
    const double2 e = *reinterpret_cast<const double2 *>(rowbuf + j);
    const double * s = src + (size_t)i0 * ld + j;
    double * d = dst + (size_t)i0 * ld + j;
    for (int i = i0; i + UNROLL <= i0 + ROWS; i += UNROLL) {
        double2 a[UNROLL]; double k[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { a[u] = *reinterpret_cast<const double2 *>(s + (size_t)u * ld); k[u] = colbuf[i + u]; }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            double2 o; const double p0 = k[u] * e.x, p1 = k[u] * e.y;
            o.x = a[u].x + p0; o.y = a[u].y + p1;
            *reinterpret_cast<double2 *>(d + (size_t)u * ld) = o;
        }
        s += (size_t)UNROLL * ld; d += (size_t)UNROLL * ld;
    }
}
int main()
{
    const int m = 4096, W = 8192, ld = 8192, reps = 300;
    double *A, *B, *row, *col;
    hipMalloc(&A, (size_t)m * ld * 8); hipMalloc(&B, (size_t)m * ld * 8); hipMalloc(&row, ld * 8); hipMalloc(&col, m * 8);
    hipMemset(A, 0, (size_t)m * ld * 8); hipMemset(B, 0, (size_t)m * ld * 8); hipMemset(row, 0, ld * 8); hipMemset(col, 0, m * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(W / 512, m / 32), block(256);
    for (int variant = 0; variant < 2; variant++) {
        for (int w = 0; w < 20; w++) hipLaunchKernelGGL((sweep<32, 8>), grid, block, 0, 0, A, variant ? B : A, m, W, ld, row, col);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; r++) {
            const double * s = variant ? ((r & 1) ? B : A) : A;
            double * d = variant ? ((r & 1) ? A : B) : A;
            hipLaunchKernelGGL((sweep<32, 8>), grid, block, 0, 0, s, d, m, W, ld, row, col);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per sweep, %.0f GB/s algorithmic\n", variant ? "ping-pong" : "in-place ", ms / reps * 1e3,
               2.0 * m * W * 8 / (ms / reps / 1e3) / 1e9);
    }
    return 0;
}
