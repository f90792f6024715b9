// Which SIMD does wave w of a 4-wave workgroup land on? (the batch kernel's selecting wave is wave 0)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/simd_place tools/lab/simd_place.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned * out, int spin)
{
    extern __shared__ char lds[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned x = threadIdx.x | 1u;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;     // keep the workgroup resident for a while
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = id | (x == 7u ? 1u << 31 : 0u);
    if (x == 3u) lds[threadIdx.x] = 1;
}
int main()
{
    const int wgs = 1280;
    unsigned * d; hipMalloc(&d, wgs * 4 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 30 * 1024);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 30 * 1024, 0, d, 20000);
    std::vector<unsigned> h(wgs * 4); hipMemcpy(h.data(), d, wgs * 16, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < wgs; b++) for (int w = 0; w < 4; w++) hist[w][(h[b * 4 + w] >> 4) & 3]++;
    for (int w = 0; w < 4; w++) printf("wave %d of a workgroup: SIMD 0..3 = %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("first workgroups (wave0 hw_id): "); for (int b = 0; b < 8; b++) printf("%08x ", h[b * 4]); printf("\n");
    return 0;
}
