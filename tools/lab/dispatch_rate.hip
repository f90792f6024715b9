// How fast does the device start workgroups? Empty and near-empty kernels over grids of the rational sweep's shape.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_rate tools/lab/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int * p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_load(const int * __restrict__ q, int * p) { const int v = q[(blockIdx.x * 256 + threadIdx.x) & 0xfffff]; if (v == 12345) p[0] = 1; }
template <int SPIN> __global__ void k_spin(int * p)
{
    unsigned x = threadIdx.x | 1u;
    for (int i = 0; i < SPIN; i++) x = x * 3u + 1u;
    if (x == 0x12345u) p[0] = 1;
}
#define T(name, launch) do { for (int i = 0; i < 5; i++) launch; hipEventRecord(a, 0); for (int i = 0; i < 50; i++) launch; hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); printf("%-44s %8.2f us\n", name, ms * 20.0f); } while (0)
int main()
{
    int * p; hipMalloc(&p, 4 << 20); hipMemset(p, 0, 4 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    char nm[96];
    for (int wgs : {256, 1024, 2048, 4096, 8192, 16384, 32768}) {
        snprintf(nm, 96, "empty, %d x 256 threads", wgs); T(nm, hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, 0, p));
    }
    for (int wgs : {8192, 32768}) { snprintf(nm, 96, "empty, %d x 64 threads", wgs); T(nm, hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(64), 0, 0, p)); }
    for (int wgs : {2048, 8192}) { snprintf(nm, 96, "empty, %d x 1024 threads", wgs / 4); T(nm, hipLaunchKernelGGL(k_empty, dim3(wgs / 4), dim3(1024), 0, 0, p)); }
    for (int wgs : {1024, 4096, 8192}) { snprintf(nm, 96, "one load, %d x 256 threads", wgs); T(nm, hipLaunchKernelGGL(k_load, dim3(wgs), dim3(256), 0, 0, p, p)); }
    for (int wgs : {1024, 4096, 8192}) { snprintf(nm, 96, "spin 512 dependent VALU, %d x 256", wgs); T(nm, hipLaunchKernelGGL(k_spin<512>, dim3(wgs), dim3(256), 0, 0, p)); }
    for (int wgs : {1024, 4096, 8192}) { snprintf(nm, 96, "spin 2048 dependent VALU, %d x 256", wgs); T(nm, hipLaunchKernelGGL(k_spin<2048>, dim3(wgs), dim3(256), 0, 0, p)); }
    return 0;
}
