// The context handle (xpg_ctx) and the small host helpers every entry point shares: error macro, device binding,
// the LDS-limit cache, the leading-dimension rule. No kernel is referenced from here, so a translation unit that
// only needs the handle does not compile the LP kernels (the library is built as several TUs, see build.py).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/xpoly_amd.h"

namespace xpg { struct LoopState; }

struct xpg_ctx {
    int device;
    hipStream_t stream;
    std::string err;
    // scratch of the one-shot K1 entry points (xpg_pivot_*_dev)
    void * rowbuf; void * colbuf; xpg::LoopState * st; size_t row_cap, col_cap;
    void * stage; size_t stage_cap;   // grow-only device staging of the host-array batch entry points
    void * hstage; size_t hstage_cap; // its pinned host mirror (the MIP controller packs node batches into it)
    void * hpack = 0; size_t hpack_cap = 0;   // pinned host buffer of the packed-result entry points (the view they return)
    std::vector<xpg_ctx *> lanes;             // extra handles on the same device, one per concurrent shape class of a ragged call
    std::vector<std::pair<void *, size_t> > dev_cache;   // device blocks between host-array row-elimination calls (DevBuf)
    size_t dev_cache_bytes = 0;
    void * slice_buf = 0; size_t slice_cap = 0;   // k_batch's time slices: checkpoints, queue and counters (grow-only)
    int update_variant;     // tuning knob for the fp64 sweep (see launch_update_f64)
    int loop_mode;          // 0: pipelined fp64 loop (2 launches per pivot), 1: serial pick/prep/update
    int zigzag;             // pipelined sweep alternates its tile order (Infinity Cache reuse)
    int block_len;          // blocked loop (loop_mode 3): pivots staged per sweep, 1..16
    int loop_auto;          // XPG_LOOP unset: blocked loop where the sweep is what costs (large fp64 tableaux)
    int num_cus;            // compute units of the device
    int chain;              // blocked loop: stages 1.. of a batch in ONE persistent launch (lp_chain.hip.h); XPG_CHAIN=0 turns it off
    int chain_local = 1;        // the chain's workers all on ONE XCD, hand-offs through its L2 (XPG_CHAIN_XCD=0: spread over the chip, sc1 stores)
    int chain_fold = 1;         // a batch's stage 0 inside the chain launch wherever the batch before admitted it (XPG_CHAIN_FOLD=0: always as launches of its own)
    int chain_test_abort = 0;   // test hook XPG_CHAIN_TEST_ABORT=k (read when the handle is created): every k-th chain launch fails its roll call
    // xpg_profile_begin/end: event pairs around each sweep launch
    std::vector<hipEvent_t> ev0, ev1;
    int prof_cap, prof_n, prof_stride, prof_seen;
};

#define XPG_HIP(ctx, call)                                                         \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);        \
            return XPG_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

namespace xpg {

// Every extern "C" entry point binds the handle's device for its own duration and puts the caller's
// current device back: allocations, function attributes and launches of a handle created on device A
// must not land on whatever device the calling thread (or its host framework) selected last.
struct DeviceGuard {
    int prev = -1, mine = -1;
    explicit DeviceGuard(int device) : mine(device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (device >= 0 && prev != device) (void)hipSetDevice(device);
    }
    ~DeviceGuard() { if (prev >= 0 && mine >= 0 && prev != mine) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard & operator=(const DeviceGuard &) = delete;
};
#define XPG_BIND(ctx_) xpg::DeviceGuard xpg_bind_guard_((ctx_) ? (ctx_)->device : -1)

inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

// hipFuncAttributeMaxDynamicSharedMemorySize is one value per (function, device): two host threads -- two handles,
// or the _multi entry points given the same device twice -- setting "exactly what this launch needs" could lower
// it between the other thread's set and its launch. So the limit is only ever RAISED, under a mutex.
inline hipError_t lds_limit(const void * fn, int device, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> cur;
    std::lock_guard<std::mutex> g(mu);
    size_t & c = cur[std::make_pair(fn, device)];
    if (bytes <= c) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) c = bytes;
    return e;
}

// Leading dimension of a device tableau of W live columns. Always a multiple of 16 elements: rows start on 128-byte
// lines, every 16-byte access is aligned, and a thread whose first column is live owns a whole pair. A width that is
// itself a multiple of 16 is kept (4096 x 8192 runs best at ld = 8192: 78 us per blocked sweep against 82 at 8208);
// any other goes to the next multiple of 64 (rows on 512-byte boundaries: 4096 x 12289 126 us at 12352 against 131
// at 12304), stepping over the row strides the sweep was measured to run 10-20 % slower at (tools/lab/sweep_lab2.hip
// ldscan, profiles/round3_sweep_lab.txt: k * (32 KiB + 128 B) -- 8224, 12336, 16448 elements -- and 32 KiB - 128 B).
inline int pick_ld(int W)
{
    static const int align = [] { const char * s = xpg_hook("XPG_LD_ALIGN"); const int a = s ? atoi(s) : 64; return a >= 16 && a % 16 == 0 ? a : 64; }();
    int ld = W % 16 == 0 ? W : round_up(W, align);
    if (ld % 4112 == 0 || (ld + 16) % 4096 == 0) ld += 16;
    // A/B aid: XPG_LD_PAD=n (a multiple of 16 elements) widens every row by n
    static const int pad = [] { const char * s = xpg_hook("XPG_LD_PAD"); const int a = s ? atoi(s) : 0; return a > 0 && a % 16 == 0 ? a : 0; }();
    return ld + pad;
}

// Device scratch of one host-array call. The blocks come from, and go back to, a small cache the handle owns
// (freed with it): a caller that eliminates one system per call -- the drop-in adapter does -- would otherwise pay
// four hipMalloc / hipFree pairs per call, more than the kernel.
struct DevBuf {
    void * p; size_t cap; xpg_ctx * owner;
    DevBuf() : p(0), cap(0), owner(0) {}
    ~DevBuf()
    {
        if (!p) return;
        if (!owner || cap > ((size_t)1 << 30)) { (void)hipFree(p); return; }
        // park the block; when the cache is full (16 blocks / 1 GiB) the LARGEST parked blocks go first, so that the
        // small blocks of one-system callers are not crowded out by what a large batch left behind
        while (!owner->dev_cache.empty() && (owner->dev_cache.size() >= 16 || owner->dev_cache_bytes + cap > ((size_t)1 << 30))) {
            size_t big = 0;
            for (size_t i = 1; i < owner->dev_cache.size(); i++) if (owner->dev_cache[i].second > owner->dev_cache[big].second) big = i;
            (void)hipFree(owner->dev_cache[big].first);
            owner->dev_cache_bytes -= owner->dev_cache[big].second;
            owner->dev_cache.erase(owner->dev_cache.begin() + (long)big);
        }
        owner->dev_cache.push_back(std::make_pair(p, cap));
        owner->dev_cache_bytes += cap;
    }
    hipError_t alloc(xpg_ctx * ctx, size_t bytes)
    {
        if (bytes < 256) bytes = 256;
        owner = ctx;
        int best = -1;                                   // the smallest cached block that is large enough
        for (size_t i = 0; i < ctx->dev_cache.size(); i++)
            if (ctx->dev_cache[i].second >= bytes && (best < 0 || ctx->dev_cache[i].second < ctx->dev_cache[(size_t)best].second)) best = (int)i;
        if (best >= 0 && ctx->dev_cache[(size_t)best].second <= 2 * bytes + 4096) {
            p = ctx->dev_cache[(size_t)best].first; cap = ctx->dev_cache[(size_t)best].second;
            ctx->dev_cache_bytes -= cap;
            ctx->dev_cache.erase(ctx->dev_cache.begin() + best);
            return hipSuccess;
        }
        cap = bytes;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {                           // make room: drop the cache and try once more
            for (auto & b : ctx->dev_cache) (void)hipFree(b.first);
            ctx->dev_cache.clear(); ctx->dev_cache_bytes = 0;
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) { p = 0; cap = 0; }
        return e;
    }
};

inline int lineq_grid(int nb) { return nb < 256 * 16 ? nb : 256 * 16; }
// Lanes per system: the smallest of 16 / 32 / 64 that covers `width` (the columns for the column-parallel Gauss
// kernels; columns AND rows for reduce, whose duplicate-row and classification passes run one lane per row; fme
// always takes the whole wave for its P x N result rows); a wave then carries 64 / L systems -- as long as their
// LDS slices fit the 64 KB a workgroup gets by default. Groups of one wave that take different branches run one
// after the other, so packing pays where the control flow is mostly shared (measured: DESIGN.md section 4).
struct LineqGeom { int L, G; size_t lds; dim3 block; int grid; int sys_lds; };
inline LineqGeom lineq_geom(int nb, int width, size_t sys_lds)
{
    LineqGeom q;
    q.L = width <= 16 ? 16 : (width <= 32 ? 32 : 64);
    if (const char * e = xpg_hook("XPG_LINEQ_LANES")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) q.L = v > q.L ? v : q.L; }
    sys_lds = (sys_lds + 15) & ~(size_t)15;
    while (q.L < 64 && sys_lds * (size_t)(64 / q.L) > 64 * 1024) q.L *= 2;
    q.G = 64 / q.L;
    q.sys_lds = (int)sys_lds;
    q.lds = sys_lds * (size_t)q.G;
    q.block = dim3((unsigned)q.L, (unsigned)q.G);
    const int wgs = (nb + q.G - 1) / q.G;
    q.grid = lineq_grid(wgs);
    return q;
}

#define XPG_TRY(e_) do { hipError_t err_ = (e_); if (err_ != hipSuccess) { ctx->err = std::string(#e_) + ": " + hipGetErrorString(err_); return XPG_ERR_HIP; } } while (0)

} // namespace xpg
