// GPU lab bench (not part of the library): what does ONE dependent step of a latency chain cost on
// MI355X, as a launch of its own and as a phase of a persistent kernel? The blocked simplex loop is
// such a chain (pick -> prep -> pick -> ...: each step reduces ~64-128 small records the previous
// step's workgroups left, gathers a strided column or a row, and leaves records of its own).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/launch_lab tools/lab/launch_lab.hip && tools/_build/launch_lab
//
// Steps measured (host wall time over N dependent steps, one stream):
//   A  trivial kernel, 1 workgroup
//   B  "record step" as a launch: NW one-wave workgroups, each reads all NW records of the previous
//      step (one lane per record), reduces them, gathers G strided doubles, writes its record
//   C  B replayed from a hipGraph of 32 nodes
//   D  the same step as a phase of ONE persistent launch: records published with sc1 stores,
//      arrival counter (agent-scope atomic add), sc1-load poll; workers = every workgroup
//   E  D with the workers on one XCD (workgroups with blockIdx % 8 == 0 of an 8x larger grid)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Rec { double v; unsigned long long tag; };

__global__ void k_trivial(int * p) { if (threadIdx.x == 0 && p[0] == -1) p[1] = 1; }

__device__ __forceinline__ double gather(const double * __restrict__ col, int w, int nw, int G, int ld)
{
    double s = 0.0;
    for (int i = w * 64 + (int)threadIdx.x; i < G; i += nw * 64) s += col[(size_t)i * ld];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// one step as its own launch: records of step t-1 in rec[(t-1)&1], written to rec[t&1]
__global__ __launch_bounds__(64) void k_step(Rec * rec, int t, const double * col, int G, int ld)
{
    const int nw = gridDim.x, w = blockIdx.x, lane = threadIdx.x;
    const Rec * in = rec + (size_t)((t - 1) & 1) * 256;
    double v = lane < nw ? in[lane].v : 0.0;
    const bool ok = lane < nw ? in[lane].tag == (unsigned long long)(t - 1) : true;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int sel = ((int)v) & 1023;                       // the "entering column": depends on the reduction
    const double g = gather(col + sel, w, nw, G, ld);
    if (lane == 0) { Rec r; r.v = (ok ? 1.0 : 1e9) + g * 1e-30; r.tag = (unsigned long long)t; rec[(size_t)(t & 1) * 256 + w] = r; }
}

// the same step as phases of one launch
__global__ __launch_bounds__(64) void k_persist(Rec * rec, unsigned * ctr, int steps, const double * col, int G, int ld,
                                                int spread, int nw, unsigned long long * stamps)
{
    if ((int)blockIdx.x % spread != 0) return;
    const int w = blockIdx.x / spread, lane = threadIdx.x;
    if (w >= nw) return;
    unsigned long long t0 = wall_clock64();
    for (int t = 1; t <= steps; t++) {
        const Rec * in = rec + (size_t)((t - 1) & 1) * 256;
        double v = 0.0;
        if (lane < nw) {
            // data-tagged granule: poll until the record of step t-1 is there (16-byte sc1 load)
            unsigned long long tag; double x;
            do {
                const unsigned long long * p = (const unsigned long long *)&in[lane];
                x = __builtin_bit_cast(double, __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                tag = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while (tag != (unsigned long long)(t - 1));
            v = x;
        }
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const int sel = ((int)v) & 1023;
        const double g = gather(col + sel, w, nw, G, ld);
        if (lane == 0) {
            unsigned long long * p = (unsigned long long *)&rec[(size_t)(t & 1) * 256 + w];
            __hip_atomic_store(p, __builtin_bit_cast(unsigned long long, 1.0 + g * 1e-30), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p + 1, (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (w == 0 && lane == 0) stamps[0] = wall_clock64() - t0;
    (void)ctr;
}

int main(int argc, char ** argv)
{
    const int N = 4000;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int * d_i; CK(hipMalloc(&d_i, 64)); CK(hipMemset(d_i, 0, 64));
    Rec * d_rec; CK(hipMalloc(&d_rec, 2 * 256 * sizeof(Rec)));
    unsigned * d_ctr; CK(hipMalloc(&d_ctr, 256));
    unsigned long long * d_st; CK(hipMalloc(&d_st, 64));
    const int M = 4096, LD = 8192;
    double * d_tab; CK(hipMalloc(&d_tab, (size_t)M * LD * 8)); CK(hipMemset(d_tab, 0, (size_t)M * LD * 8));
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto reset = [&](int tag0) {
        std::vector<Rec> h(512); for (auto & r : h) { r.v = 1.0; r.tag = (unsigned long long)tag0; }
        CK(hipMemcpy(d_rec, h.data(), sizeof(Rec) * 512, hipMemcpyHostToDevice));
    };
    // A
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d_i);
        CK(hipStreamSynchronize(s));
        if (rep) printf("A  trivial kernel, 1 WG:                      %6.2f us per step\n", (now() - t0) / N);
    }
    for (int G : {0, 4096}) for (int nw : {64, 128}) {
        // B
        for (int rep = 0; rep < 2; rep++) {
            reset(0);
            double t0 = now();
            for (int t = 1; t <= N; t++) hipLaunchKernelGGL(k_step, dim3(nw), dim3(64), 0, s, d_rec, t, d_tab, G, LD);
            CK(hipStreamSynchronize(s));
            if (rep) printf("B  launch per step, %3d WGs, gather %4d:       %6.2f us per step\n", nw, G, (now() - t0) / N);
        }
        // C: graph of 32 steps
        {
            hipGraph_t g; hipGraphExec_t ge;
            reset(0);
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int t = 1; t <= 32; t++) hipLaunchKernelGGL(k_step, dim3(nw), dim3(64), 0, s, d_rec, t, d_tab, G, LD);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; rep++) {
                double t0 = now();
                for (int i = 0; i < N / 32; i++) { CK(hipGraphLaunch(ge, s)); }
                CK(hipStreamSynchronize(s));
                if (rep) printf("C  hipGraph of 32 steps, %3d WGs, gather %4d:  %6.2f us per step (tags not chained across replays)\n", nw, G, (now() - t0) / (N / 32 * 32));
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        // D, E
        for (int spread : {1, 8}) {
            for (int rep = 0; rep < 2; rep++) {
                reset(0);
                double t0 = now();
                hipLaunchKernelGGL(k_persist, dim3(nw * spread), dim3(64), 0, s, d_rec, d_ctr, N, d_tab, G, LD, spread, nw, d_st);
                CK(hipStreamSynchronize(s));
                unsigned long long ticks = 0; CK(hipMemcpy(&ticks, d_st, 8, hipMemcpyDeviceToHost));
                if (rep) printf("%s  persistent, %3d workers%s, gather %4d:  %6.2f us per step (host), %6.2f (in-kernel clock)\n",
                                spread == 1 ? "D" : "E", nw, spread == 1 ? " over all XCDs" : " on one XCD  ", G, (now() - t0) / N, ticks * 0.01 / N);
            }
        }
    }
    return 0;
}
