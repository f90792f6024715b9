// GPU lab bench (not part of the library): the price of one all-to-all hand-off step of a persistent latency
// chain when every worker sits on ONE XCD and the granules stay in that XCD's L2.
//
// Round 2 measured "workers on one XCD" with sc1 granule stores (launch_lab.hip row E) and found next to nothing:
// an sc1 store drops the line from the XCD's L2, so a same-XCD reader still pays the fabric. Here the producer
// stores are PLAIN 16-byte stores (write through the CU's L1 into the XCD's L2 and stay there) and the consumers
// poll with sc1 loads (bypass their own L1, served by the L2): inside one XCD the L2 is the point of coherence,
// so the hand-off should cost an L2 round trip, not a fabric one.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/xcd_handoff_lab tools/lab/xcd_handoff_lab.hip && tools/_build/xcd_handoff_lab
//
// Step (what a chain stage does twice): every worker polls the granules of ALL nw workers of the previous step
// (lane l polls l, l+64, ...), reduces them, gathers G strided doubles of a 4096 x 8192 tableau (the entering
// column; its index depends on the reduction), and publishes its own granule.
// Modes: 0 = sc1 store, workers spread over all XCDs (the round-2/3 chain)
//        1 = sc1 store, workers on one XCD
//        2 = plain store, workers on one XCD
//        3 = sc0 store, workers on one XCD
// Every granule carries {value, step}; a poll that does not complete within ~1 s flags the run (stale line).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld_sc1(const void * p)
{
    u32x4 g;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
    return g;
}
template <int MODE> __device__ __forceinline__ void st_granule(void * p, u32x4 g)
{
    if (MODE == 0 || MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" :: "v"(p), "v"(g) : "memory");
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 3" :: "v"(p), "v"(g) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 3" :: "v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ int xcc_id()
{
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}

struct Out { unsigned long long ticks; int stuck; int xcc_min, xcc_max; };

// where granule k of a step's array lies: dense (16 bytes apart: a polling wave's load covers eight 128-byte lines, all in
// one or two L2 channels) or SPREAD: granule k at (k & 15) * page + (k >> 4) * 16 -- sixteen lines per load, in sixteen
// different `page`-sized blocks (L2 channels interleave at some such granularity: every poller of every worker reads the
// same 2 KB, so a dense array is one channel's queue)
__device__ __forceinline__ size_t gran_off(int k, int page) { return page ? (size_t)(k & 15) * page + (size_t)(k >> 4) * 16 : (size_t)k * 16; }

// grid = nw * spread workgroups of 64; workers are the workgroups with blockIdx % spread == 0
template <int MODE> __global__ __launch_bounds__(64)
void k_chain(char * gran, int steps, const double * tab, int G, int ld, int spread, int nw, Out * out, int * xccs, int page)
{
    if ((int)blockIdx.x % spread != 0) return;
    const int w = blockIdx.x / spread, lane = threadIdx.x;
    if (w >= nw) return;
    if (lane == 0) xccs[w] = xcc_id();
    const unsigned long long t0 = wall_clock64();
    const int nu = (nw + 63) >> 6;
    for (int t = 1; t <= steps; t++) {
        const char * in = gran + (size_t)((t - 1) & 1) * (1 << 17);
        double v = 0.0;
        unsigned spins = 0;
        for (int u = 0; u < nu; u++) {
            const int k = lane + 64 * u;
            const int kc = k < nw ? k : lane % nw;
            for (;;) {
                const u32x4 g = ld_sc1(in + gran_off(kc, page));
                if (__all(g.z == (unsigned)(t - 1))) { if (k < nw) v += __builtin_bit_cast(double, ((unsigned long long)g.y << 32) | g.x); break; }
                if (++spins > (1u << 22)) { if (lane == 0) out->stuck = t; return; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const int sel = ((int)v) & 1023;
        double s = 0.0;
        for (int i = w * 64 + lane; i < G; i += nw * 64) s += tab[(size_t)i * ld + sel];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const unsigned long long b = __builtin_bit_cast(unsigned long long, 1.0 + s * 1e-30);
            u32x4 g; g.x = (unsigned)b; g.y = (unsigned)(b >> 32); g.z = (unsigned)t; g.w = 0;
            st_granule<MODE>(gran + (size_t)(t & 1) * (1 << 17) + gran_off(w, page), g);
        }
    }
    if (w == 0 && lane == 0) out->ticks = wall_clock64() - t0;
}

int main()
{
    const int N = 4000;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    char * d_gran; CK(hipMalloc(&d_gran, 2 << 17));
    Out * d_out; CK(hipMalloc(&d_out, sizeof(Out)));
    int * d_x; CK(hipMalloc(&d_x, 512 * 4));
    const int M = 4096, LD = 8192;
    double * d_tab; CK(hipMalloc(&d_tab, (size_t)M * LD * 8)); CK(hipMemset(d_tab, 0, (size_t)M * LD * 8));
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto reset = [&] {
        std::vector<unsigned> h((2 << 17) / 4, 0u);
        for (size_t k = 0; k < h.size() / 4; k++) { const unsigned long long b = __builtin_bit_cast(unsigned long long, 1.0); h[4 * k] = (unsigned)b; h[4 * k + 1] = (unsigned)(b >> 32); h[4 * k + 2] = 0u; }
        CK(hipMemcpy(d_gran, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_out, 0, sizeof(Out)));
    };
    const char * names[4] = { "sc1 store, all XCDs ", "sc1 store, one XCD  ", "plain store, one XCD", "sc0 store, one XCD  " };
    for (int G : {0, 4096}) for (int nw : {64, 129, 193, 258}) for (int page : {0, 256, 1024, 4096}) for (int mode = 0; mode < 3; mode += 2) {
        const int spread = mode == 0 ? 1 : 8;
        for (int rep = 0; rep < 2; rep++) {
            reset();
            const double t0 = now();
            switch (mode) {
            case 0: hipLaunchKernelGGL(k_chain<0>, dim3(nw * spread), dim3(64), 0, s, d_gran, N, d_tab, G, LD, spread, nw, d_out, d_x, page); break;
            case 1: hipLaunchKernelGGL(k_chain<1>, dim3(nw * spread), dim3(64), 0, s, d_gran, N, d_tab, G, LD, spread, nw, d_out, d_x, page); break;
            case 2: hipLaunchKernelGGL(k_chain<2>, dim3(nw * spread), dim3(64), 0, s, d_gran, N, d_tab, G, LD, spread, nw, d_out, d_x, page); break;
            default: hipLaunchKernelGGL(k_chain<3>, dim3(nw * spread), dim3(64), 0, s, d_gran, N, d_tab, G, LD, spread, nw, d_out, d_x, page); break;
            }
            CK(hipStreamSynchronize(s));
            const double us = (now() - t0) / N;
            Out o; CK(hipMemcpy(&o, d_out, sizeof(o), hipMemcpyDeviceToHost));
            std::vector<int> x(512); CK(hipMemcpy(x.data(), d_x, 512 * 4, hipMemcpyDeviceToHost));
            int lo = 99, hi = -1; for (int k = 0; k < nw; k++) { lo = x[k] < lo ? x[k] : lo; hi = x[k] > hi ? x[k] : hi; }
            if (rep) printf("%s  %3d workers, gather %4d, granules %s%-4d: %6.2f us per step (host) %6.2f (device clock)  xcc %d..%d%s\n", names[mode], nw, G, page ? "spread over blocks of " : "dense ", page, us,
                            o.ticks * 0.01 / N, lo, hi, o.stuck ? "  STUCK (stale line)" : "");
        }
    }
    return 0;
}
