// Lab bench, round 5: what does the chip issue of NON-fused fp64 arithmetic -- the sweep's v_mul_f64 + v_add_f64 pairs (two
// roundings per update: the reference's arithmetic) -- with nothing else going on? Registers only, no memory in the loop.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/_build/valu_f64_lab tools/lab/valu_f64_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k_valu(double * out, const double * in, int iters)
{
    double a[8], e[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { a[q] = in[threadIdx.x + 256 * q]; e[q] = in[2048 + threadIdx.x + 256 * q]; }
    double k = in[5000 + (blockIdx.x & 7)];
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (MODE == 0) { const double p = k * e[q]; a[q] = a[q] + p; }                 // mul, add: two instructions
                else if (MODE == 1) a[q] = __builtin_fma(k, e[q], a[q]);                       // one fused instruction
                else if (MODE == 2) a[q] = a[q] + e[q];                                        // add only
                else a[q] = a[q] * e[q];                                                       // mul only
            }
        }
        k = -k;
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) s += a[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the sweep's dependency pattern: NCH accumulators (it has 4: two rows x a column pair), each the end of a chain of
// add <- mul pairs; 32 distinct e so that no product is shared
template <int NCH> __global__ __launch_bounds__(256) void k_chain(double * out, const double * in, int iters)
{
    double a[NCH], e[32];
#pragma unroll
    for (int q = 0; q < NCH; q++) a[q] = in[threadIdx.x + 256 * q];
#pragma unroll
    for (int q = 0; q < 32; q++) e[q] = in[2048 + ((threadIdx.x + 7 * q) & 2047)];
    double k = in[5000 + (blockIdx.x & 7)];
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32 / NCH; r++)
#pragma unroll
            for (int q = 0; q < NCH; q++) { const double p = k * e[r * NCH + q]; a[q] = a[q] + p; }
        k = -k;
    }
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NCH; q++) s += a[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    double *in, *out;
    CK(hipMalloc(&in, 8192 * 8)); CK(hipMalloc(&out, (size_t)8192 * 256 * 8));
    { double h[8192]; for (int i = 0; i < 8192; i++) h[i] = 1e-3 * (i % 97) + 0.5; CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    typedef void (*kern_t)(double *, const double *, int);
    struct V { const char * name; kern_t f; double ops; } vs[] = {
        {"mul + add pairs, 2 chains", k_chain<2>, 2.0}, {"mul + add pairs, 4 chains (the sweep's)", k_chain<4>, 2.0},
        {"mul + add pairs, 8 chains", k_chain<8>, 2.0}, {"mul + add pairs, 16 chains", k_chain<16>, 2.0}, {"v_fma_f64", k_valu<1>, 1.0}, {"v_add_f64", k_valu<2>, 1.0}, {"v_mul_f64", k_valu<3>, 1.0} };
    const int iters = 2000;
    for (int blocks : {1024, 3072, 4096}) for (auto & v : vs) {
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(v.f, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int w = 0; w < 5; w++) hipLaunchKernelGGL(v.f, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double insts = (double)blocks * 256 * iters * 32 * v.ops * 5;       // lane-instructions
        printf("  %4d workgroups  %-42s %7.2f T lane-instructions/s  (%.2f cycles per wave instruction at 2.4 GHz, 1024 SIMDs)\n", blocks, v.name,
               insts / (ms * 1e-3) / 1e12, 1024.0 * 2.4e9 * 64 / (insts / (ms * 1e-3)));
    }
    return 0;
}
