// Lab: what a chain stage's COLUMN GATHER costs on one XCD -- 64 one-wave workgroups (rows 64 w .. 64 w + 63), each lane
// one cell of column c of a 4096 x ld fp64 matrix (row stride ld * 8 bytes) -- cold (first touch after the matrix was
// streamed through), warm (the same lines again: 16 columns share a 128-byte line), and for several leading dimensions
// (a power-of-two stride puts a column's 4096 lines on few L2 / HBM channels).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/gather_lab tools/lab/gather_lab.hip && tools/_build/gather_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned long long wall() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

// out[w][k]: 100 MHz ticks of gather k of worker w. cols[k]: the column of gather k.
__global__ __launch_bounds__(64) void k_gather(const double * tab, int ld, int m, const int * cols, int ng, unsigned long long * out, double * sink, int spread)
{
    if (!spread && (blockIdx.x & 7u)) return;
    const int w = spread ? (int)blockIdx.x : (int)(blockIdx.x >> 3), lane = threadIdx.x;
    const int i = w * 64 + lane;
    double acc = 0.0;
    for (int k = 0; k < ng; k++) {
        const int c = cols[k];
        __builtin_amdgcn_s_sleep(100);
        const unsigned long long t0 = wall();
        const double x = __builtin_nontemporal_load(&tab[0]) * 0.0 + tab[(size_t)i * ld + c];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += x;
        const unsigned long long t1 = wall();
        if (lane == 0) out[(size_t)w * ng + k] = t1 - t0;
        // a few microseconds between gathers, like a stage
        for (int z = 0; z < 40; z++) __builtin_amdgcn_s_sleep(127);
    }
    if (acc == 123.456) sink[0] = acc;
}
__global__ void k_stream(double * tab, size_t n) { for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) tab[k] = tab[k] * 1.0000001 + 1e-9; }

int main()
{
    const int m = 4096;
    const int lds[] = {8192, 8208, 8224, 8256, 12352};
    const int seq[] = {5, 5, 6, 21, 21, 5, 300, 300, 301};            // cold, same cell, same line, next group, again, back, far, again, same line
    const int ng = sizeof(seq) / sizeof(int);
    int * dcols; unsigned long long * dout; double * sink;
    hipMalloc(&dcols, sizeof(seq)); hipMemcpy(dcols, seq, sizeof(seq), hipMemcpyHostToDevice);
    hipMalloc(&dout, 64 * ng * 8); hipMalloc(&sink, 8);
    for (int spread = 0; spread < 2; spread++)
    for (int ld : lds) {
        double * tab; const size_t n = (size_t)m * ld;
        hipMalloc(&tab, n * 8); hipMemset(tab, 0, n * 8);
        std::vector<double> best(ng, 1e9), worst(ng, 0), mean(ng, 0);
        const int reps = 5;
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, tab, n);       // what a sweep leaves in the caches
            hipLaunchKernelGGL(k_gather, dim3(spread ? 64 : 64 * 8), dim3(64), 0, 0, tab, ld, m, dcols, ng, dout, sink, spread);
            std::vector<unsigned long long> h(64 * ng);
            hipMemcpy(h.data(), dout, 64 * ng * 8, hipMemcpyDeviceToHost);
            for (int k = 0; k < ng; k++) {
                double mx = 0, sum = 0;
                for (int w = 0; w < 64; w++) { const double us = h[(size_t)w * ng + k] * 0.01; mx = us > mx ? us : mx; sum += us; }
                if (mx < best[k]) best[k] = mx;
                mean[k] += sum / 64 / reps;
            }
        }
        printf("%s ld %5d (stride %6zu B): slowest worker's gather, best of %d [mean over workers] us:", spread ? "spread " : "one XCD", ld, (size_t)ld * 8, reps);
        for (int k = 0; k < ng; k++) printf("  c=%d %.2f [%.2f]", seq[k], best[k], mean[k]);
        printf("\n");
        hipFree(tab);
    }
    return 0;
}
