// C ABI of libxpoly_amd.so (declared in include/xpoly_amd.h). gfx950 only.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see
// __graft_entry__.build()). fp-contract must stay off: the reference rounds
// after the multiply and after the add (lpsol.h:1485-1489).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <new>
#include <map>
#include <algorithm>
#include <thread>
#include <vector>
#include "../../include/xpoly_amd.h"
// The library is ONE shared object built from this file compiled four times in parallel (-DXPG_PART=0..3, build.py):
// the device code of all kernels together takes four minutes in one translation unit, and every part only includes
// the kernel headers its entry points launch. XPG_PART undefined = everything in one translation unit.
//   part 0  handle, K1 pivot, the device-resident LP (every loop of lp_*.hip.h), warm-started MIP, test and debug hooks
//   part 1  SIX::maxm / minm, the LDS-resident LP batches (k_batch), their multi-device and ragged forms
//   part 2  MIP (device tree walk + host controller), has_solution, DepPoly::is_empty front end
//   part 3  rational / integer row elimination (Lineq, rank / det / inv / null, hnf, gcd)
#ifndef XPG_PART
#define XPG_PART (-1)
#endif
#define XPG_IN(p_) (XPG_PART < 0 || XPG_PART == (p_))
#include "scalar.hip.h"
#include "ctx.hip.h"
#if XPG_IN(0)
#include "lp_kernels.hip.h"
#include "lp_pipe_r32.hip.h"
#include "lp_fused_r32.hip.h"
#include "lp_host.hip.h"
#include "warm_mip.hip.h"
#include "warm_mip_batch.hip.h"
#endif
#if XPG_IN(1) || XPG_IN(2)
#include "six_host.hip.h"
#include "batch_kernels.hip.h"
#endif
#if XPG_IN(3)
#include "lineq_host.hip.h"
#endif
#if XPG_IN(2)
#include "mip_host.hip.h"
#endif

using namespace xpg;

// batch_dev<S> launches k_batch<S>: part 1 compiles it, part 2 (the MIP controller's node batches, has_solution's
// LPs) calls part 1's instance
#if XPG_PART == 2
namespace xpg {
extern template int batch_dev<F64>(xpg_ctx *, int, int, const F64 *, const F64 *, int, int, unsigned, int32_t *, F64 *, F64 *, uint32_t *, int);
extern template int batch_dev<R32>(xpg_ctx *, int, int, const R32 *, const R32 *, int, int, unsigned, int32_t *, R32 *, R32 *, uint32_t *, int);
}
#elif XPG_PART == 1
namespace xpg {
template int batch_dev<F64>(xpg_ctx *, int, int, const F64 *, const F64 *, int, int, unsigned, int32_t *, F64 *, F64 *, uint32_t *, int);
template int batch_dev<R32>(xpg_ctx *, int, int, const R32 *, const R32 *, int, int, unsigned, int32_t *, R32 *, R32 *, uint32_t *, int);
}
#endif

static_assert(sizeof(F64) == 8 && sizeof(R32) == 8, "both scalars are 8 bytes");
static_assert(sizeof(xpg_rat32) == sizeof(R32), "ABI layout of a rational");

#if XPG_IN(0)
struct xpg_lp { LpBase * impl; };

extern "C" {

const char * xpg_version(void) { return "xpoly_amd 0.1 (gfx950)"; }

int xpg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int xpg_create(xpg_ctx ** out, int device)
{
    if (!out) return XPG_ERR_SHAPE;
    *out = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return XPG_ERR_NO_DEVICE;
    xpg::DeviceGuard bind(device);
    xpg_ctx * c = new (std::nothrow) xpg_ctx();
    if (!c) return XPG_ERR_ALLOC;
    c->device = device;
    c->rowbuf = c->colbuf = 0; c->st = 0; c->row_cap = c->col_cap = 0;
    c->stage = 0; c->stage_cap = 0;
    c->hstage = 0; c->hstage_cap = 0;
    const char * var = xpg_hook("XPG_UPDATE_VARIANT");
    c->update_variant = var ? atoi(var) : 0;
    // XPG_LOOP: "block" / "pipe" force the blocked / the pipelined loop whatever the size (unset: chosen by size). The
    // three-launch "serial" loop and the "split" form of the pipelined one (neither a default nor a fallback: kept for the
    // lab's A/B runs and the tests that compare the loops with each other) are selectable in the -DXPG_TEST_HOOKS build only.
    const char * lm = xpg_env("XPG_LOOP");
    if (lm && lm[0] == 's' && !xpg_hook("XPG_LOOP")) lm = nullptr;
    const char * zz = xpg_hook("XPG_ZIGZAG");
    c->zigzag = zz ? atoi(zz) : 0;                      // measured slower (79.7 vs 77.8 us per sweep): off
    c->loop_mode = (lm && lm[0] == 's' && lm[1] == 'e') ? 1 : ((lm && lm[0] == 's' && lm[1] == 'p') ? 2 : 0);
    if (lm && lm[0] == 'b') c->loop_mode = 3;          // "block": B pivots per sweep (lp_blocked.hip.h)
    if (lm && lm[0] == 'p') c->loop_mode = 0;          // "pipe": always the pipelined loop
    const char * ch = xpg_env("XPG_CHAIN");               // "0": launch-per-stage chain, for A/B runs
    c->chain = ch ? atoi(ch) : 1;
    const char * cx = xpg_hook("XPG_CHAIN_XCD");
    c->chain_local = cx ? (atoi(cx) != 0 ? 1 : 0) : 1;
    const char * cf = xpg_hook("XPG_CHAIN_FOLD");
    c->chain_fold = cf ? (atoi(cf) != 0 ? 1 : 0) : 1;
    const char * cta = xpg_hook("XPG_CHAIN_TEST_ABORT");
    c->chain_test_abort = cta ? atoi(cta) : 0;
    c->num_cus = 0;
    if (hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->num_cus = 0;
    c->loop_auto = lm ? 0 : 1;                         // unset: blocked loop for large fp64 tableaux, else pipelined
    const char * bl = xpg_env("XPG_BLOCK");
    c->block_len = bl ? atoi(bl) : BLK_DEFAULT;
    if (c->block_len < 1) c->block_len = 1;
    if (c->block_len > BLK_MAX) c->block_len = BLK_MAX;
    c->prof_cap = 0; c->prof_n = 0; c->prof_stride = 1; c->prof_seen = 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return XPG_ERR_HIP; }
    if (hipMalloc((void **)&c->st, sizeof(LoopState)) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return XPG_ERR_ALLOC; }
    *out = c;
    return 0;
}

void xpg_destroy(xpg_ctx * ctx)
{
    if (!ctx) return;
    XPG_BIND(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->ev0) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev1) (void)hipEventDestroy(e);
    if (ctx->rowbuf) (void)hipFree(ctx->rowbuf);
    if (ctx->colbuf) (void)hipFree(ctx->colbuf);
    if (ctx->st) (void)hipFree(ctx->st);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->hstage) (void)hipHostFree(ctx->hstage);
    if (ctx->hpack) (void)hipHostFree(ctx->hpack);
    if (ctx->slice_buf) (void)hipFree(ctx->slice_buf);
    for (xpg_ctx * l : ctx->lanes) xpg_destroy(l);
    ctx->lanes.clear();
    for (auto & b : ctx->dev_cache) (void)hipFree(b.first);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int xpg_profile_begin(xpg_ctx * ctx, int cap, int stride)
{
    XPG_BIND(ctx);
    if (!ctx || cap < 0 || stride < 1) return XPG_ERR_SHAPE;
    XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    while ((int)ctx->ev0.size() < cap) {
        hipEvent_t a, b;
        XPG_HIP(ctx, hipEventCreate(&a));
        XPG_HIP(ctx, hipEventCreate(&b));
        ctx->ev0.push_back(a); ctx->ev1.push_back(b);
    }
    ctx->prof_cap = cap; ctx->prof_n = 0; ctx->prof_stride = stride; ctx->prof_seen = 0;
    return 0;
}

int xpg_profile_end(xpg_ctx * ctx, int * launches, double * total_ms)
{
    XPG_BIND(ctx);
    if (!ctx) return XPG_ERR_SHAPE;
    XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (int i = 0; i < ctx->prof_n; i++) {
        float ms = 0.f;
        XPG_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0[i], ctx->ev1[i]));
        sum += ms;
    }
    if (launches) *launches = ctx->prof_n;
    if (total_ms) *total_ms = sum;
    ctx->prof_cap = 0; ctx->prof_n = 0;
    return 0;
}

const char * xpg_last_error(const xpg_ctx * ctx) { return ctx ? ctx->err.c_str() : "null context"; }
void * xpg_stream(const xpg_ctx * ctx) { return ctx ? (void *)ctx->stream : 0; }

int xpg_sync(xpg_ctx * ctx)
{
    XPG_BIND(ctx);
    if (!ctx) return XPG_ERR_SHAPE;
    XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int xpg_malloc(xpg_ctx * ctx, void ** dptr, size_t bytes)
{
    XPG_BIND(ctx);
    if (!ctx || !dptr) return XPG_ERR_SHAPE;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return XPG_ERR_ALLOC; }
    return 0;
}
int xpg_free(xpg_ctx * ctx, void * dptr)
{
    XPG_BIND(ctx);
    if (!ctx) return XPG_ERR_SHAPE;
    XPG_HIP(ctx, hipFree(dptr));
    return 0;
}
int xpg_upload(xpg_ctx * ctx, void * dst_dev, const void * src_host, size_t bytes)
{
    XPG_BIND(ctx);
    if (!ctx) return XPG_ERR_SHAPE;
    XPG_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int xpg_download(xpg_ctx * ctx, void * dst_host, const void * src_dev, size_t bytes)
{
    XPG_BIND(ctx);
    if (!ctx) return XPG_ERR_SHAPE;
    XPG_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

} // extern "C"

// ---- K1 on caller-owned device tableaux --------------------------------------------
namespace {

int ensure_scratch(xpg_ctx * ctx, int m, int W)
{
    const size_t rb = (size_t)round_up(W, 16) * 8, cb = (size_t)round_up(m, 16) * 8;
    if (rb > ctx->row_cap) {
        if (ctx->rowbuf) XPG_HIP(ctx, hipFree(ctx->rowbuf));
        ctx->rowbuf = 0; ctx->row_cap = 0;
        if (hipMalloc(&ctx->rowbuf, rb) != hipSuccess) { ctx->err = "hipMalloc(rowbuf)"; return XPG_ERR_ALLOC; }
        ctx->row_cap = rb;
    }
    if (cb > ctx->col_cap) {
        if (ctx->colbuf) XPG_HIP(ctx, hipFree(ctx->colbuf));
        ctx->colbuf = 0; ctx->col_cap = 0;
        if (hipMalloc(&ctx->colbuf, cb) != hipSuccess) { ctx->err = "hipMalloc(colbuf)"; return XPG_ERR_ALLOC; }
        ctx->col_cap = cb;
    }
    return 0;
}

template <class S> __global__ void k_stage_pivot(LpView<S> v, int row, int col)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LoopState * st = v.st;
    st->status = ST_RUNNING; st->row = row; st->col = col; st->leave = 0;
    st->cnv_bits = to_bits(v.obj[col]);
    st->piv_bits = to_bits(v.tab[(size_t)row * v.ld + col]);
}

template <class S>
int pivot_dev(xpg_ctx * ctx, S * tab, int m, int W, int ld, S * obj, int rhs, int row, int col)
{
    if (!ctx || !tab || !obj || m <= 0 || W <= 1 || ld < W || row < 0 || row >= m || col < 0 ||
        col >= W || rhs < 0 || rhs >= W)
        return XPG_ERR_SHAPE;
    // the fp64 sweep uses 16-byte accesses: rows must start 16-byte aligned
    if (is_f64<S>::value && ((ld & 1) || ((uintptr_t)tab & 15))) return XPG_ERR_SHAPE;
    int rc = ensure_scratch(ctx, m, W);
    if (rc) return rc;
    LpView<S> v;
    memset(&v, 0, sizeof(v));
    v.tab = tab; v.m = m; v.W = W; v.ld = ld; v.rhs = rhs; v.obj = obj;
    v.rowbuf = (S *)ctx->rowbuf; v.colbuf = (S *)ctx->colbuf; v.st = ctx->st;
    hipLaunchKernelGGL((k_stage_pivot<S>), dim3(1), dim3(64), 0, ctx->stream, v, row, col);
    const int span = W > m ? W : m;
    hipLaunchKernelGGL((k_prep<S>), dim3((span + 255) / 256), dim3(256), 0, ctx->stream, v, 0, 0, 0, 0);
    launch_update<S>(ctx, v, 0);
    XPG_HIP(ctx, hipGetLastError());
    return 0;
}

template <class S>
int pivot_host(xpg_ctx * ctx, S * tab, int m, int W, S * obj, int rhs, int row, int col)
{
    if (!ctx || !tab || !obj || m <= 0 || W <= 1) return XPG_ERR_SHAPE;
    const int ld = round_up(W, 16);
    S * d_tab = 0; S * d_obj = 0;
    XPG_HIP(ctx, hipMalloc((void **)&d_tab, (size_t)m * ld * sizeof(S)));
    if (hipMalloc((void **)&d_obj, (size_t)ld * sizeof(S)) != hipSuccess) { (void)hipFree(d_tab); return XPG_ERR_ALLOC; }
    int rc = 0;
    hipError_t e = hipMemcpy2DAsync(d_tab, (size_t)ld * sizeof(S), tab, (size_t)W * sizeof(S),
                                    (size_t)W * sizeof(S), m, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_obj, obj, (size_t)W * sizeof(S), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = pivot_dev<S>(ctx, d_tab, m, W, ld, d_obj, rhs, row, col);
    if (e == hipSuccess && rc == 0)
        e = hipMemcpy2DAsync(tab, (size_t)W * sizeof(S), d_tab, (size_t)ld * sizeof(S),
                             (size_t)W * sizeof(S), m, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(obj, d_obj, (size_t)W * sizeof(S), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_tab); (void)hipFree(d_obj);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return XPG_ERR_HIP; }
    return rc;
}

} // namespace

extern "C" {

int xpg_pivot_f64_dev(xpg_ctx * ctx, double * tab, int m, int W, int ld, double * obj,
                      int rhs_idx, int row, int col)
{
    XPG_BIND(ctx); return pivot_dev<F64>(ctx, (F64 *)tab, m, W, ld, (F64 *)obj, rhs_idx, row, col); }
int xpg_pivot_rat32_dev(xpg_ctx * ctx, xpg_rat32 * tab, int m, int W, int ld, xpg_rat32 * obj,
                        int rhs_idx, int row, int col)
{
    XPG_BIND(ctx); return pivot_dev<R32>(ctx, (R32 *)tab, m, W, ld, (R32 *)obj, rhs_idx, row, col); }
int xpg_pivot_f64(xpg_ctx * ctx, double * tab, int m, int W, double * obj, int rhs_idx, int row, int col)
{
    XPG_BIND(ctx); return pivot_host<F64>(ctx, (F64 *)tab, m, W, (F64 *)obj, rhs_idx, row, col); }
int xpg_pivot_rat32(xpg_ctx * ctx, xpg_rat32 * tab, int m, int W, xpg_rat32 * obj, int rhs_idx, int row, int col)
{
    XPG_BIND(ctx); return pivot_host<R32>(ctx, (R32 *)tab, m, W, (R32 *)obj, rhs_idx, row, col); }

// ---- device-resident LP ---------------------------------------------------------------
int xpg_lp_create(xpg_ctx * ctx, int kind, const void * leq, int m, int cols, const void * tgtf,
                  const void * vc_diag, const void * vc_rhs, int src_on_device, xpg_lp ** out)
{
    XPG_BIND(ctx);
    if (!ctx || !out || !leq || !tgtf || m <= 0 || cols < 2 || (kind != 0 && kind != 1))
        return XPG_ERR_SHAPE;
    *out = 0;
    xpg_lp * h = new (std::nothrow) xpg_lp();
    if (!h) return XPG_ERR_ALLOC;
    int rc;
    if (kind == 0) {
        Lp<F64> * p = new (std::nothrow) Lp<F64>();
        if (!p) { delete h; return XPG_ERR_ALLOC; }
        p->ctx = ctx; p->kind = 0; h->impl = p;
        rc = p->create(leq, m, cols, tgtf, vc_diag, vc_rhs, src_on_device);
    } else {
        Lp<R32> * p = new (std::nothrow) Lp<R32>();
        if (!p) { delete h; return XPG_ERR_ALLOC; }
        p->ctx = ctx; p->kind = 1; h->impl = p;
        rc = p->create(leq, m, cols, tgtf, vc_diag, vc_rhs, src_on_device);
    }
    if (rc) { delete h->impl; delete h; return rc; }
    *out = h;
    return 0;
}

void xpg_lp_destroy(xpg_lp * lp)
{
    if (!lp) return;
    XPG_BIND(lp->impl ? lp->impl->ctx : (xpg_ctx *)0);
    if (lp->impl) { (void)hipStreamSynchronize(lp->impl->ctx->stream); delete lp->impl; }
    delete lp;
}

#define XPG_DISPATCH(lp, expr)                                                   \
    do {                                                                         \
        if (!(lp) || !(lp)->impl) return XPG_ERR_SHAPE;                          \
        XPG_BIND((lp)->impl->ctx);                                               \
        if ((lp)->impl->kind == 0) { Lp<F64> * p = (Lp<F64> *)(lp)->impl; return expr; } \
        Lp<R32> * p = (Lp<R32> *)(lp)->impl; return expr;                        \
    } while (0)

int xpg_lp_two_stage(xpg_lp * lp, unsigned max_iter) { XPG_DISPATCH(lp, p->two_stage(max_iter)); }
int xpg_lp_begin(xpg_lp * lp) { XPG_DISPATCH(lp, p->begin()); }
int xpg_lp_iterate(xpg_lp * lp, unsigned pivots) { XPG_DISPATCH(lp, p->iterate(pivots)); }
int xpg_lp_read(xpg_lp * lp, void * tab, void * obj, uint8_t * nvset, uint8_t * bvset,
                int32_t * bv2eq, int32_t * eq2bv, void * maxv, void * sol)
{ XPG_DISPATCH(lp, p->read(tab, obj, nvset, bvset, bv2eq, eq2bv, maxv, sol)); }

int xpg_lp_shape(xpg_lp * lp, int * rows, int * W, int * rhs_idx)
{
    if (!lp || !lp->impl) return XPG_ERR_SHAPE;
    int m, w, r;
    if (lp->impl->kind == 0) { Lp<F64> * p = (Lp<F64> *)lp->impl; m = p->v.m; w = p->v.W; r = p->v.rhs; }
    else { Lp<R32> * p = (Lp<R32> *)lp->impl; m = p->v.m; w = p->v.W; r = p->v.rhs; }
    if (rows) *rows = m;
    if (W) *W = w;
    if (rhs_idx) *rhs_idx = r;
    return 0;
}

int xpg_lp_set_options(xpg_lp * lp, int pricing, double feas_rel_tol)
{
    if (!lp || !lp->impl) return XPG_ERR_SHAPE;
    XPG_BIND(lp->impl->ctx);
    if (lp->impl->kind == 0) return ((Lp<F64> *)lp->impl)->set_options(pricing, feas_rel_tol);
    return ((Lp<R32> *)lp->impl)->set_options(pricing, feas_rel_tol);
}

int xpg_lp_pivots_done(xpg_lp * lp, unsigned * out)
{
    if (!lp || !lp->impl || !out) return XPG_ERR_SHAPE;
    XPG_BIND(lp->impl->ctx);
    LoopState hs;
    int rc;
    if (lp->impl->kind == 0) rc = ((Lp<F64> *)lp->impl)->read_state(&hs);
    else rc = ((Lp<R32> *)lp->impl)->read_state(&hs);
    if (rc) return rc;
    *out = hs.total_pivots;
    return 0;
}

int xpg_lp_counters(xpg_lp * lp, unsigned * sweeps_full, unsigned * sweeps_partial)
{
    if (!lp || !lp->impl) return XPG_ERR_SHAPE;
    XPG_BIND(lp->impl->ctx);
    LoopState hs;
    int rc;
    if (lp->impl->kind == 0) rc = ((Lp<F64> *)lp->impl)->read_state(&hs);
    else rc = ((Lp<R32> *)lp->impl)->read_state(&hs);
    if (rc) return rc;
    if (sweeps_full) *sweeps_full = hs.blk.sweeps_full;
    if (sweeps_partial) *sweeps_partial = hs.blk.sweeps_part;
    return 0;
}

// Host-side views of two pieces of launch geometry, for the CPU test suite (no device needed): the blocked sweep's
// workgroup -> tile map (every tile of a strips x rowblocks tableau exactly once, whatever the shape) and the leading
// dimension a device tableau of W live columns gets.
int xpg_test_sweep_tile(int strips, int rowblocks, int rev, int lid, int * bx, int * by)
{
    if (!bx || !by || strips <= 0 || rowblocks <= 0) return XPG_ERR_SHAPE;
    if (lid < 0) return blk_sweep_grid(strips, rowblocks);      // lid < 0: the grid size
    int x = -1, y = -1;
    const bool live = blk_sweep_tile(lid, strips, rowblocks, rev, x, y);
    *bx = x; *by = y;
    return live ? 1 : 0;
}
int xpg_test_pick_ld(int W) { return W > 0 ? pick_ld(W) : XPG_ERR_SHAPE; }
} // extern "C"
namespace xpg { namespace {
__global__ __launch_bounds__(256) void k_test_canon_ops(int n, const R32 * a, const R32 * k, const R32 * e, R32 * out_fma, R32 * out_div)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (out_fma) out_fma[i] = fma_canon(a[i], k[i], e[i]);
    if (out_div) out_div[i] = k[i].num == 0 ? R32(0, 1) : div_canon(a[i], k[i]);
}
__global__ __launch_bounds__(256) void k_test_any_ops(int n, const R32 * a, const R32 * b, R32 * out_mul, R32 * out_add, R32 * out_div)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (out_mul) out_mul[i] = mul_any(a[i], b[i]);
    if (out_add) out_add[i] = add_any(a[i], b[i]);
    if (out_div) out_div[i] = div_any(a[i], b[i]);
}
} }
extern "C" {
int xpg_test_any_ops_rat32(xpg_ctx * ctx, int n, const xpg_rat32 * a, const xpg_rat32 * b, xpg_rat32 * out_mul, xpg_rat32 * out_add,
                           xpg_rat32 * out_div)
{
    if (!ctx || n < 0 || !a || !b) return XPG_ERR_SHAPE;
    XPG_BIND(ctx);
    if (n == 0) return 0;
    const size_t bytes = (size_t)n * sizeof(R32);
    DevBuf d;
    XPG_HIP(ctx, d.alloc(ctx, 5 * bytes));
    R32 * base = (R32 *)d.p;
    XPG_HIP(ctx, hipMemcpyAsync(base, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    XPG_HIP(ctx, hipMemcpyAsync(base + n, b, bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_test_any_ops, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, base, base + n,
                       out_mul ? base + 2 * (size_t)n : (R32 *)0, out_add ? base + 3 * (size_t)n : (R32 *)0, out_div ? base + 4 * (size_t)n : (R32 *)0);
    if (out_mul) XPG_HIP(ctx, hipMemcpyAsync(out_mul, base + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (out_add) XPG_HIP(ctx, hipMemcpyAsync(out_add, base + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (out_div) XPG_HIP(ctx, hipMemcpyAsync(out_div, base + 4 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream));
    XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int xpg_test_canon_ops_rat32(xpg_ctx * ctx, int n, const xpg_rat32 * a, const xpg_rat32 * k, const xpg_rat32 * e,
                             xpg_rat32 * out_fma, xpg_rat32 * out_div)
{
    if (!ctx || n < 0 || !a || !k || !e) return XPG_ERR_SHAPE;
    XPG_BIND(ctx);
    if (n == 0) return 0;
    const R32 * in[3] = { (const R32 *)a, (const R32 *)k, (const R32 *)e };
    for (int t = 0; t < 3; t++)
        for (int i = 0; i < n; i++)
            if (!canonical(in[t][i])) { ctx->err = "xpg_test_canon_ops_rat32: operand not canonical"; return XPG_ERR_SHAPE; }
    const size_t bytes = (size_t)n * sizeof(R32);
    DevBuf d;
    XPG_HIP(ctx, d.alloc(ctx, 5 * bytes));
    R32 * base = (R32 *)d.p;
    for (int t = 0; t < 3; t++) XPG_HIP(ctx, hipMemcpyAsync(base + (size_t)t * n, in[t], bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_test_canon_ops, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, base, base + n, base + 2 * (size_t)n,
                       out_fma ? base + 3 * (size_t)n : (R32 *)0, out_div ? base + 4 * (size_t)n : (R32 *)0);
    if (out_fma) XPG_HIP(ctx, hipMemcpyAsync(out_fma, base + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (out_div) XPG_HIP(ctx, hipMemcpyAsync(out_div, base + 4 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream));
    XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int xpg_lp_chain_aborts(xpg_lp * lp, unsigned * aborts, int * chain_off, unsigned * runs)
{
    if (!lp || !lp->impl) return XPG_ERR_SHAPE;
    XPG_BIND(lp->impl->ctx);
    if (runs) *runs = 0;
    if (lp->impl->kind != 0) { if (aborts) *aborts = 0; if (chain_off) *chain_off = 0; return 0; }
    Lp<F64> * p = (Lp<F64> *)lp->impl;
    LoopState hs;
    const int rc = p->read_state(&hs);
    if (rc) return rc;
    if (aborts) *aborts = hs.blk.ch_aborts;
    if (chain_off) *chain_off = p->chain_off ? 1 : 0;
    if (runs) *runs = hs.blk.ch_runs;
    return 0;
}

int xpg_lp_loop_info(xpg_lp * lp, int32_t * out, int n)
{
    if (!lp || !lp->impl || !out || n < 0) return XPG_ERR_SHAPE;
    int32_t f[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (lp->impl->kind == 0) ((Lp<F64> *)lp->impl)->loop_info(f);
    else ((Lp<R32> *)lp->impl)->loop_info(f);
    for (int k = 0; k < n && k < 10; k++) out[k] = f[k];
    return 0;
}

#ifdef XPG_STAMPS
// diagnostic builds only (not declared in the header): the phase tick sums of the blocked loop
int xpg_lp_debug(xpg_lp * lp, unsigned long long * out8)
{
    if (!lp || !lp->impl) return XPG_ERR_SHAPE;
    LoopState hs;
    int rc = lp->impl->kind == 0 ? ((Lp<F64> *)lp->impl)->read_state(&hs) : ((Lp<R32> *)lp->impl)->read_state(&hs);
    if (rc) return rc;
    for (int k = 0; k < 8; k++) out8[k] = hs.blk.dbg[k];
    return 0;
}
// the staged rows / columns of the last batch: E[BLK_MAX][ld], K[m][BLK_MAX]; *ld receives the leading dimension and
// *blk_max the stage capacity (BLK_MAX). The caller says how many stages its arrays were sized for (`cap`, from a first
// call with E = K = NULL); a capacity that is not the library's is refused instead of overrunning the caller's heap.
int xpg_lp_debug_staged(xpg_lp * lp, double * E, double * K, int * ld, int * blk_max, int cap)
{
    if (!lp || !lp->impl || lp->impl->kind != 0) return XPG_ERR_SHAPE;
    Lp<F64> * p = (Lp<F64> *)lp->impl;
    xpg_ctx * ctx = p->ctx;
    if (ld) *ld = p->v.ld;
    if (blk_max) *blk_max = BLK_MAX;
    if ((E || K) && cap != BLK_MAX) return XPG_ERR_SHAPE;
    if (E) XPG_HIP(ctx, hipMemcpy(E, p->v.blkE, (size_t)BLK_MAX * p->v.ld * 8, hipMemcpyDeviceToHost));
    if (K) XPG_HIP(ctx, hipMemcpy(K, p->v.blkK, (size_t)p->v.m * BLK_MAX * 8, hipMemcpyDeviceToHost));
    return 0;
}
int xpg_lp_debug_counts(xpg_lp * lp, int * rowcnt, int * colcnt, int n)
{
    if (!lp || !lp->impl || lp->impl->kind != 0) return XPG_ERR_SHAPE;
    Lp<F64> * p = (Lp<F64> *)lp->impl;
    xpg_ctx * ctx = p->ctx;
    XPG_HIP(ctx, hipMemcpy(rowcnt, p->v.rowcnt, (size_t)n * 4, hipMemcpyDeviceToHost));
    XPG_HIP(ctx, hipMemcpy(colcnt, p->v.colcnt, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}
int xpg_lp_debug_chain_ts(xpg_lp * lp, unsigned long long * out, int * blk_max, int cap)   // [4][BLK_MAX][8]; cap as above
{
    if (!lp || !lp->impl) return XPG_ERR_SHAPE;
    xpg_ctx * ctx = lp->impl->ctx;
    if (blk_max) *blk_max = BLK_MAX;
    if (!out) return 0;
    if (cap != BLK_MAX) return XPG_ERR_SHAPE;
    XPG_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ch_ts), sizeof(unsigned long long) * 4 * BLK_MAX * 8));
    return 0;
}
int xpg_lp_debug_rows(xpg_lp * lp, double * out)              // [4][8192]
{
    if (!lp || !lp->impl) return XPG_ERR_SHAPE;
    xpg_ctx * ctx = lp->impl->ctx;
    XPG_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_rows), sizeof(double) * 4 * 8192));
    return 0;
}
#endif

int xpg_lp_chain_folds(xpg_lp * lp, unsigned * folds)
{
    if (!lp || !lp->impl || !folds) return XPG_ERR_SHAPE;
    XPG_BIND(lp->impl->ctx);
    *folds = 0;
    if (lp->impl->kind != 0) return 0;
    LoopState hs;
    const int rc = ((Lp<F64> *)lp->impl)->read_state(&hs);
    if (rc) return rc;
    *folds = hs.blk.ch_folds;
    return 0;
}

int xpg_lp_trace(xpg_lp * lp, int32_t * pairs, int cap_pairs, int * n_pairs)
{
    if (!lp || !lp->impl || !n_pairs) return XPG_ERR_SHAPE;
    xpg_ctx * ctx = lp->impl->ctx;
    XPG_BIND(ctx);
    LoopState hs;
    int rc; int * d_trace; int cap;
    if (lp->impl->kind == 0) { Lp<F64> * p = (Lp<F64> *)lp->impl; rc = p->read_state(&hs); d_trace = p->v.trace; cap = p->v.trace_cap; }
    else { Lp<R32> * p = (Lp<R32> *)lp->impl; rc = p->read_state(&hs); d_trace = p->v.trace; cap = p->v.trace_cap; }
    if (rc) return rc;
    int n = (int)hs.total_pivots;
    *n_pairs = n;
    if (n > cap) n = cap;
    if (n > cap_pairs) n = cap_pairs;
    if (pairs && n > 0) {
        XPG_HIP(ctx, hipMemcpyAsync(pairs, d_trace, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int xpg_trim(xpg_ctx * ctx)
{
    XPG_BIND(ctx);
    if (!ctx) return XPG_ERR_SHAPE;
    XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto & b : ctx->dev_cache) (void)hipFree(b.first);
    ctx->dev_cache.clear(); ctx->dev_cache_bytes = 0;
    if (ctx->hpack) { (void)hipHostFree(ctx->hpack); ctx->hpack = 0; ctx->hpack_cap = 0; }
    return 0;
}

int xpg_mip_warm_f64(xpg_ctx * ctx, int is_max, const double * tgtf, const double * leq, int leq_rows, int cols, int is_bin,
                     double * out_v, double * out_sol, long long * out_stats)
{
    XPG_BIND(ctx);
    if (!ctx || !tgtf || !leq || leq_rows <= 0 || cols < 2 || !out_v) return XPG_ERR_SHAPE;
    std::vector<double> obj(tgtf, tgtf + cols);
    if (!is_max) for (int j = 0; j < cols; j++) obj[(size_t)j] = -obj[(size_t)j];       // min c.x = -max (-c).x
    WarmMip W(ctx);
    WarmStats S;
    double v = 0.0;
    const int st = W.solve(obj.data(), leq, leq_rows, cols, is_bin != 0, &v, out_sol, S);
    if (out_stats) { out_stats[0] = S.nodes; out_stats[1] = S.dual_pivots; out_stats[2] = S.root_pivots; out_stats[3] = S.max_depth; }
    if (st == XPG_IP_SUCC) *out_v = is_max ? v : -v;
    else *out_v = 0.0;
    return st;
}

int xpg_mip_warm_batch_f64(xpg_ctx * ctx, int nb, int is_max, const double * tgtf, const double * leq, int leq_rows, int cols, int is_bin,
                           int32_t * out_status, double * out_v, double * out_sol, long long * out_stats)
{
    XPG_BIND(ctx);
    return warm_mip_batch(ctx, nb, is_max, tgtf, leq, leq_rows, cols, is_bin, out_status, out_v, out_sol, out_stats);
}

} // extern "C"
#endif // part 0

#if XPG_IN(1)
extern "C" {
#ifdef XPG_LIFE
// diagnostic builds (-DXPG_LIFE) only: the per-LP pivot time marks of k_batch (4096 LPs x 32 marks), cleared on read
int xpg_life_debug(xpg_ctx * ctx, unsigned long long * out)
{
    XPG_BIND(ctx);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_life), sizeof(unsigned long long) * 4096 * 32) != hipSuccess) return XPG_ERR_HIP;
    std::vector<unsigned long long> z(4096 * 32, 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_life), z.data(), sizeof(unsigned long long) * 4096 * 32) != hipSuccess) return XPG_ERR_HIP;
    return 0;
}
#endif
#ifdef XPG_STAMPS
// diagnostic builds only: reads and clears sm_solve_lp's tick sums (phase one, plain build, main solve, pivots, counts)
int xpg_lp_solve_debug(xpg_ctx * ctx, unsigned long long * out8)
{
    XPG_BIND(ctx);
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lp_ticks), sizeof(z)) != hipSuccess) return XPG_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lp_ticks), z, sizeof(z)) != hipSuccess) return XPG_ERR_HIP;
    return 0;
}
// diagnostic builds only: reads and clears sm_fast_loop's counters
int xpg_fastloop_debug(xpg_ctx * ctx, unsigned long long * out16)
{
    XPG_BIND(ctx);
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fl), sizeof(z)) != hipSuccess) return XPG_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fl), z, sizeof(z)) != hipSuccess) return XPG_ERR_HIP;
    return 0;
}
#endif
// ---- SIX::maxm / minm ---------------------------------------------------------------------
// where the calling thread's last xpg_six_* call spent its time (six_host.hip.h SixProfile)
int xpg_six_last_profile(double * out_ms, int n)
{
    if (!out_ms || n < 0) return XPG_ERR_SHAPE;
    const SixProfile & p = six_profile();
    const double f[9] = { p.total_ms, p.reshape_ms, p.create_ms, p.dual_ms, p.solve_ms, p.read_ms, p.destroy_ms, (double)p.route, (double)p.pivots };
    for (int k = 0; k < n && k < 9; k++) out_ms[k] = f[k];
    return 0;
}
int xpg_test_normalize(xpg_ctx * ctx, int kind, const void * tgtf, const void * vc, int vc_rows, const void * eq, int eq_rows,
                       const void * leq, int leq_rows, int cols, void * out_dev_cells, void * out_host_cells, long long cap_cells,
                       int32_t * out_info)
{
    XPG_BIND(ctx);
    if (!ctx || !out_dev_cells || !out_host_cells || !out_info || (kind != 0 && kind != 1)) return XPG_ERR_SHAPE;
    if (kind == 0)
        return test_normalize<F64>(ctx, (const F64 *)tgtf, (const F64 *)vc, vc_rows, (const F64 *)eq, eq_rows, (const F64 *)leq, leq_rows, cols,
                                   (F64 *)out_dev_cells, (F64 *)out_host_cells, cap_cells, out_info);
    return test_normalize<R32>(ctx, (const R32 *)tgtf, (const R32 *)vc, vc_rows, (const R32 *)eq, eq_rows, (const R32 *)leq, leq_rows, cols,
                               (R32 *)out_dev_cells, (R32 *)out_host_cells, cap_cells, out_info);
}
int xpg_six_maxm_f64(xpg_ctx * ctx, const double * tgtf, const double * vc, int vc_rows,
                     const double * eq, int eq_rows, const double * leq, int leq_rows, int cols,
                     unsigned max_iter, double * out_v, double * out_sol)
{
    XPG_BIND(ctx);
    return six_solve<F64>(ctx, 0, true, (const F64 *)tgtf, (const F64 *)vc, vc_rows, (const F64 *)eq,
                          eq_rows, (const F64 *)leq, leq_rows, cols, max_iter, (F64 *)out_v, (F64 *)out_sol);
}
int xpg_six_minm_f64(xpg_ctx * ctx, const double * tgtf, const double * vc, int vc_rows,
                     const double * eq, int eq_rows, const double * leq, int leq_rows, int cols,
                     unsigned max_iter, double * out_v, double * out_sol)
{
    XPG_BIND(ctx);
    return six_solve<F64>(ctx, 0, false, (const F64 *)tgtf, (const F64 *)vc, vc_rows, (const F64 *)eq,
                          eq_rows, (const F64 *)leq, leq_rows, cols, max_iter, (F64 *)out_v, (F64 *)out_sol);
}
int xpg_six_maxm_rat32(xpg_ctx * ctx, const xpg_rat32 * tgtf, const xpg_rat32 * vc, int vc_rows,
                       const xpg_rat32 * eq, int eq_rows, const xpg_rat32 * leq, int leq_rows,
                       int cols, unsigned max_iter, xpg_rat32 * out_v, xpg_rat32 * out_sol)
{
    XPG_BIND(ctx);
    return six_solve<R32>(ctx, 1, true, (const R32 *)tgtf, (const R32 *)vc, vc_rows, (const R32 *)eq,
                          eq_rows, (const R32 *)leq, leq_rows, cols, max_iter, (R32 *)out_v, (R32 *)out_sol);
}
int xpg_six_minm_rat32(xpg_ctx * ctx, const xpg_rat32 * tgtf, const xpg_rat32 * vc, int vc_rows,
                       const xpg_rat32 * eq, int eq_rows, const xpg_rat32 * leq, int leq_rows,
                       int cols, unsigned max_iter, xpg_rat32 * out_v, xpg_rat32 * out_sol)
{
    XPG_BIND(ctx);
    return six_solve<R32>(ctx, 1, false, (const R32 *)tgtf, (const R32 *)vc, vc_rows, (const R32 *)eq,
                          eq_rows, (const R32 *)leq, leq_rows, cols, max_iter, (R32 *)out_v, (R32 *)out_sol);
}

// ---- batches ------------------------------------------------------------------------------
int xpg_six_batch_f64_dev(xpg_ctx * ctx, int is_max, int nb, const double * tgtf, const double * leq,
                          int m, int cols, unsigned max_iter, int32_t * out_status, double * out_v,
                          double * out_sol, uint32_t * out_pivots)
{
    XPG_BIND(ctx);
    return batch_dev<F64>(ctx, is_max, nb, (const F64 *)tgtf, (const F64 *)leq, m, cols, max_iter,
                          out_status, (F64 *)out_v, (F64 *)out_sol, out_pivots);
}
int xpg_six_batch_rat32_dev(xpg_ctx * ctx, int is_max, int nb, const xpg_rat32 * tgtf,
                            const xpg_rat32 * leq, int m, int cols, unsigned max_iter,
                            int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol,
                            uint32_t * out_pivots)
{
    XPG_BIND(ctx);
    return batch_dev<R32>(ctx, is_max, nb, (const R32 *)tgtf, (const R32 *)leq, m, cols, max_iter,
                          out_status, (R32 *)out_v, (R32 *)out_sol, out_pivots);
}
int xpg_six_batch_f64(xpg_ctx * ctx, int is_max, int nb, const double * tgtf, const double * leq,
                      int m, int cols, unsigned max_iter, int32_t * out_status, double * out_v,
                      double * out_sol)
{
    XPG_BIND(ctx);
    return batch_host<F64>(ctx, is_max, nb, (const F64 *)tgtf, (const F64 *)leq, m, cols, max_iter,
                           out_status, (F64 *)out_v, (F64 *)out_sol);
}
int xpg_six_batch_rat32(xpg_ctx * ctx, int is_max, int nb, const xpg_rat32 * tgtf,
                        const xpg_rat32 * leq, int m, int cols, unsigned max_iter,
                        int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol)
{
    XPG_BIND(ctx);
    return batch_host<R32>(ctx, is_max, nb, (const R32 *)tgtf, (const R32 *)leq, m, cols, max_iter,
                           out_status, (R32 *)out_v, (R32 *)out_sol);
}

} // extern "C"
#endif // part 1
// ---- the same batches over several devices: one context + host thread per shard ------------------
#if XPG_IN(1) || XPG_IN(2)
namespace {
// fn(ctx, lo, hi) runs shard [lo, hi) on its own context; the first error wins.
template <class F> int run_sharded(int ndev, const int * devices, int nb, F fn)
{
    if (ndev <= 0 || nb < 0) return XPG_ERR_SHAPE;
    std::vector<int> rcs((size_t)ndev, 0);
    std::vector<std::thread> th;
    const int base = nb / ndev, extra = nb % ndev;
    for (int g = 0; g < ndev; g++) {
        const int lo = g * base + (g < extra ? g : extra), hi = lo + base + (g < extra ? 1 : 0);
        const int dev = devices ? devices[g] : g;
        th.emplace_back([=, &rcs] {
            xpg_ctx * c = 0;
            int rc = xpg_create(&c, dev);
            if (rc == 0 && hi > lo) rc = fn(c, lo, hi);
            if (c) xpg_destroy(c);
            rcs[(size_t)g] = rc;
        });
    }
    for (auto & t : th) t.join();
    for (int rc : rcs) if (rc) return rc;
    return 0;
}
} // namespace
#endif
#if XPG_IN(1)
extern "C" {

int xpg_six_batch_f64_multi(int ndev, const int * devices, int is_max, int nb, const double * tgtf, const double * leq,
                            int m, int cols, unsigned max_iter, int32_t * out_status, double * out_v, double * out_sol)
{
    if (!tgtf || !leq || m <= 0 || cols < 2 || !out_status || !out_v || !out_sol) return XPG_ERR_SHAPE;
    return run_sharded(ndev, devices, nb, [=](xpg_ctx * c, int lo, int hi) {
        return xpg_six_batch_f64(c, is_max, hi - lo, tgtf + (size_t)lo * cols, leq + (size_t)lo * m * cols, m, cols, max_iter,
                                 out_status + lo, out_v + lo, out_sol + (size_t)lo * cols);
    });
}
int xpg_six_batch_rat32_multi(int ndev, const int * devices, int is_max, int nb, const xpg_rat32 * tgtf,
                              const xpg_rat32 * leq, int m, int cols, unsigned max_iter, int32_t * out_status,
                              xpg_rat32 * out_v, xpg_rat32 * out_sol)
{
    if (!tgtf || !leq || m <= 0 || cols < 2 || !out_status || !out_v || !out_sol) return XPG_ERR_SHAPE;
    return run_sharded(ndev, devices, nb, [=](xpg_ctx * c, int lo, int hi) {
        return xpg_six_batch_rat32(c, is_max, hi - lo, tgtf + (size_t)lo * cols, leq + (size_t)lo * m * cols, m, cols,
                                   max_iter, out_status + lo, out_v + lo, out_sol + (size_t)lo * cols);
    });
}
} // extern "C"
#endif
#if XPG_IN(2)
extern "C" {
int xpg_mip_batch_rat32_multi(int ndev, const int * devices, int nb, int is_max, int is_bin, const xpg_rat32 * tgtf,
                              const xpg_rat32 * leq, int leq_rows, int cols, int32_t * out_status, xpg_rat32 * out_v,
                              xpg_rat32 * out_sol, long long * out_nodes)
{
    if (!tgtf || !leq || leq_rows <= 0 || cols < 2 || !out_status || !out_v || !out_sol) return XPG_ERR_SHAPE;
    std::atomic<long long> nodes(0);
    const int rc = run_sharded(ndev, devices, nb, [=, &nodes](xpg_ctx * c, int lo, int hi) {
        long long n = 0;
        const int r = xpg_mip_batch_rat32(c, hi - lo, is_max, is_bin, tgtf + (size_t)lo * cols,
                                          leq + (size_t)lo * leq_rows * cols, leq_rows, cols, out_status + lo, out_v + lo,
                                          out_sol + (size_t)lo * cols, &n);
        nodes += n;
        return r;
    });
    if (out_nodes) *out_nodes = nodes.load();
    return rc;
}
int xpg_dep_is_empty_batch_rat32_multi(int ndev, const int * devices, int nb, const xpg_rat32 * mats, int rows, int cols,
                                       int32_t * out_empty, long long * out_nodes)
{
    if (!mats || rows <= 0 || cols < 2 || !out_empty) return XPG_ERR_SHAPE;
    std::atomic<long long> nodes(0);
    const int rc = run_sharded(ndev, devices, nb, [=, &nodes](xpg_ctx * c, int lo, int hi) {
        long long n = 0;
        const int r = xpg_dep_is_empty_batch_rat32(c, hi - lo, mats + (size_t)lo * rows * cols, rows, cols, out_empty + lo, &n);
        nodes += n;
        return r;
    });
    if (out_nodes) *out_nodes = nodes.load();
    return rc;
}


// ---- ragged batches: problems of DIFFERENT shapes in one call -----------------------------
// The dependence analysis of one SCoP emits polyhedra whose shape follows the statements' depths and the number of
// parameters (src/eng/poly.cpp:1120-1224, :1009-1053) and tests each (poly.cpp:268-314, :530-573); padding them to
// one shape is not parity-neutral (extra 0 <= 0 rows change the bug-compatible pivot sequence). A ragged call takes
// rows[nb], cols[nb] and the cell offsets of the concatenated systems, sorts the problems into shape classes and runs
// the classes CONCURRENTLY -- each on a lane: an extra handle (stream, scratch) on the same device that the caller's
// handle keeps, driven by a host thread for the duration of the call -- so that small classes share the chip instead
// of each waiting for the deepest problem of the one before. Results are scattered back in problem order.
} // extern "C"
#endif
#if XPG_IN(1) || XPG_IN(2) || XPG_IN(3)
namespace {
struct RaggedClass { int rows, cols; std::vector<int> idx; size_t work; };
inline int ragged_lanes()
{
    static const int n = [] { const char * e = xpg_env("XPG_RAGGED_LANES"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 32 ? 32 : v); }();
    return n;
}
// fn(lane_ctx, class) for every shape class; the first error wins.
template <class F> int run_ragged(xpg_ctx * ctx, int nb, const int32_t * rows, const int32_t * cols, F fn)
{
    std::map<std::pair<int, int>, size_t> where;
    std::vector<RaggedClass> cls;
    for (int b = 0; b < nb; b++) {
        if (rows[b] <= 0 || cols[b] < 2) return XPG_ERR_SHAPE;
        const auto key = std::make_pair((int)rows[b], (int)cols[b]);
        auto it = where.find(key);
        if (it == where.end()) { it = where.insert(std::make_pair(key, cls.size())).first; cls.push_back(RaggedClass{rows[b], cols[b], {}, 0}); }
        cls[it->second].idx.push_back(b);
    }
    if (cls.empty()) return 0;                                      // nb == 0
    for (auto & c : cls) c.work = c.idx.size() * (size_t)c.rows * c.cols;
    std::sort(cls.begin(), cls.end(), [](const RaggedClass & a, const RaggedClass & b) { return a.work > b.work; });
    const int nl = (int)cls.size() < ragged_lanes() ? (int)cls.size() : ragged_lanes();
    while ((int)ctx->lanes.size() < nl - 1) {                       // lane 0 is the caller's handle
        xpg_ctx * l = 0;
        const int rc = xpg_create(&l, ctx->device);
        if (rc) return rc;
        ctx->lanes.push_back(l);
    }
    std::vector<std::vector<int> > mine((size_t)nl);               // longest class first, each to the least loaded lane
    std::vector<size_t> load((size_t)nl, 0);
    for (int k = 0; k < (int)cls.size(); k++) {
        int best = 0;
        for (int l = 1; l < nl; l++) if (load[(size_t)l] < load[(size_t)best]) best = l;
        mine[(size_t)best].push_back(k); load[(size_t)best] += cls[(size_t)k].work;
    }
    std::vector<int> rcs((size_t)nl, 0);
    auto lane_body = [&](int l) {
        xpg_ctx * c = l == 0 ? ctx : ctx->lanes[(size_t)l - 1];
        xpg::DeviceGuard bind(c->device);
        for (int k : mine[(size_t)l]) { const int rc = fn(c, cls[(size_t)k]); if (rc) { rcs[(size_t)l] = rc; return; } }
    };
    std::vector<std::thread> th;
    for (int l = 1; l < nl; l++) th.emplace_back(lane_body, l);
    lane_body(0);
    for (auto & t : th) t.join();
    for (int l = 0; l < nl; l++)
        if (rcs[(size_t)l]) {                                        // the first failing lane's code and message (after the join: no race)
            if (l) ctx->err = ctx->lanes[(size_t)l - 1]->err;
            return rcs[(size_t)l];
        }
    return 0;
}
template <class T> inline void ragged_gather(std::vector<T> & buf, const T * src, const long long * off, const RaggedClass & c, size_t per)
{
    buf.resize(c.idx.size() * per);
    for (size_t k = 0; k < c.idx.size(); k++) memcpy((void *)(buf.data() + k * per), (const void *)(src + off[c.idx[k]]), per * sizeof(T));
}
} // namespace
#endif
#if XPG_IN(2)
extern "C" {

int xpg_dep_is_empty_batch_ragged_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, const int32_t * rows,
                                        const int32_t * cols, const long long * offsets, int32_t * out_empty,
                                        long long * out_nodes)
{
    XPG_BIND(ctx);
    if (out_nodes) *out_nodes = 0;
    if (!ctx || nb < 0) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;                                          // an empty SCoP: nothing to answer (mats may be null)
    if (!rows || !cols || !offsets || !out_empty) return XPG_ERR_SHAPE;
    // DepPoly::is_empty answers true for a polyhedron without rows before it looks at anything else (poly.cpp:533-535):
    // those are answered here and left out of the shape classes
    std::vector<int32_t> lrows, lcols; std::vector<long long> loff; std::vector<int> lidx;
    for (int b = 0; b < nb; b++) {
        if (rows[b] < 0) return XPG_ERR_SHAPE;
        if (rows[b] == 0) { out_empty[b] = 1; continue; }
        lrows.push_back(rows[b]); lcols.push_back(cols[b]); loff.push_back(offsets[b]); lidx.push_back(b);
    }
    if (lidx.empty()) return 0;
    if (!mats) return XPG_ERR_SHAPE;
    std::atomic<long long> nodes(0);
    const int rc = run_ragged(ctx, (int)lidx.size(), lrows.data(), lcols.data(), [&](xpg_ctx * c, const RaggedClass & g) {
        std::vector<R32> buf; std::vector<int32_t> emp(g.idx.size());
        ragged_gather(buf, (const R32 *)mats, loff.data(), g, (size_t)g.rows * g.cols);
        long n = 0;
        const int r = dep_is_empty_batch(c, (int)g.idx.size(), buf.data(), g.rows, g.cols, g.cols - 1, (const R32 *)0, emp.data(), &n);
        if (r) return r;
        for (size_t k = 0; k < g.idx.size(); k++) out_empty[lidx[(size_t)g.idx[k]]] = emp[k];
        nodes += n;
        return 0;
    });
    if (out_nodes) *out_nodes = nodes.load();
    return rc;
}

} // extern "C"
#endif
#if XPG_IN(1)
namespace {
template <class S>
int six_batch_ragged(xpg_ctx * ctx, int is_max, int nb, const S * tgtf, const S * leq, const int32_t * rows, const int32_t * cols,
                     const long long * leq_off, const long long * tg_off, unsigned max_iter, int32_t * out_status, S * out_v, S * out_sol)
{
    if (!ctx || nb < 0 || !tgtf || !leq || !rows || !cols || !leq_off || !tg_off || !out_status || !out_v || !out_sol) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    // ONE launch of the LDS-resident kernel with per-LP shapes (k_batch_ragged): the concatenated arrays go up as they
    // are, the LDS request is that of the largest LP. (Classes on concurrent lanes -- run_ragged, what the other ragged
    // entry points do -- remain the fallback for a batch whose largest LP does not fit one CU's LDS.)
    int max_R = 0, max_V = 0;
    long long leq_cells = 0, tg_cells = 0;
    for (int b = 0; b < nb; b++) {
        if (rows[b] <= 0 || cols[b] < 2 || leq_off[b] < 0 || tg_off[b] < 0) return XPG_ERR_SHAPE;
        const int n = cols[b] - 1, R = is_max ? rows[b] : n, V = is_max ? n : rows[b];
        max_R = R > max_R ? R : max_R; max_V = V > max_V ? V : max_V;
        const long long le = leq_off[b] + (long long)rows[b] * cols[b], te = tg_off[b] + cols[b];
        leq_cells = le > leq_cells ? le : leq_cells; tg_cells = te > tg_cells ? te : tg_cells;
    }
    if (small_lds_bytes<S>(max_R, max_V) <= 160 * 1024) {
        const size_t bl = (size_t)leq_cells * 8, bt = (size_t)tg_cells * 8;
        DevBuf dl, dt, ds, dv, dst, dr, dc, dlo, dto;
        XPG_TRY(dl.alloc(ctx, bl)); XPG_TRY(dt.alloc(ctx, bt)); XPG_TRY(ds.alloc(ctx, bt)); XPG_TRY(dv.alloc(ctx, (size_t)nb * 8));
        XPG_TRY(dst.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dr.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dc.alloc(ctx, (size_t)nb * 4));
        XPG_TRY(dlo.alloc(ctx, (size_t)nb * 8)); XPG_TRY(dto.alloc(ctx, (size_t)nb * 8));
        XPG_TRY(hipMemcpyAsync(dl.p, leq, bl, hipMemcpyHostToDevice, ctx->stream));
        XPG_TRY(hipMemcpyAsync(dt.p, tgtf, bt, hipMemcpyHostToDevice, ctx->stream));
        XPG_TRY(hipMemcpyAsync(ds.p, out_sol, bt, hipMemcpyHostToDevice, ctx->stream));     // (slots of unsolved LPs keep what they held)
        XPG_TRY(hipMemcpyAsync(dr.p, rows, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
        XPG_TRY(hipMemcpyAsync(dc.p, cols, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
        XPG_TRY(hipMemcpyAsync(dlo.p, leq_off, (size_t)nb * 8, hipMemcpyHostToDevice, ctx->stream));
        XPG_TRY(hipMemcpyAsync(dto.p, tg_off, (size_t)nb * 8, hipMemcpyHostToDevice, ctx->stream));
        const int rc = batch_dev_ragged<S>(ctx, is_max, nb, (const S *)dt.p, (const S *)dl.p, (const int *)dr.p, (const int *)dc.p,
                                           (const long long *)dlo.p, (const long long *)dto.p, max_R, max_V, max_iter,
                                           (int32_t *)dst.p, (S *)dv.p, (S *)ds.p);
        if (rc) return rc;
        XPG_TRY(hipMemcpyAsync(out_status, dst.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipMemcpyAsync(out_v, dv.p, (size_t)nb * 8, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipMemcpyAsync(out_sol, ds.p, bt, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipStreamSynchronize(ctx->stream));
        return 0;
    }
    return run_ragged(ctx, nb, rows, cols, [&](xpg_ctx * c, const RaggedClass & g) {
        const size_t ng = g.idx.size();
        std::vector<S> bl, bt, bs(ng * (size_t)g.cols), bv(ng); std::vector<int32_t> st(ng);
        ragged_gather(bl, leq, leq_off, g, (size_t)g.rows * g.cols);
        ragged_gather(bt, tgtf, tg_off, g, (size_t)g.cols);
        ragged_gather(bs, (const S *)out_sol, tg_off, g, (size_t)g.cols);     // (slots of unsolved LPs keep what they held)
        const int r = batch_host<S>(c, is_max, (int)ng, bt.data(), bl.data(), g.rows, g.cols, max_iter, st.data(), bv.data(), bs.data());
        if (r) return r;
        for (size_t k = 0; k < ng; k++) {
            const int b = g.idx[k];
            out_status[b] = st[k]; out_v[b] = bv[k];
            memcpy((void *)(out_sol + tg_off[b]), (const void *)(bs.data() + k * (size_t)g.cols), (size_t)g.cols * sizeof(S));
        }
        return 0;
    });
}
} // namespace
extern "C" {
int xpg_six_batch_f64_ragged(xpg_ctx * ctx, int is_max, int nb, const double * tgtf, const double * leq, const int32_t * rows,
                             const int32_t * cols, const long long * leq_offsets, const long long * tgtf_offsets,
                             unsigned max_iter, int32_t * out_status, double * out_v, double * out_sol)
{
    XPG_BIND(ctx);
    return six_batch_ragged<F64>(ctx, is_max, nb, (const F64 *)tgtf, (const F64 *)leq, rows, cols, leq_offsets, tgtf_offsets, max_iter,
                                 out_status, (F64 *)out_v, (F64 *)out_sol);
}
int xpg_six_batch_rat32_ragged(xpg_ctx * ctx, int is_max, int nb, const xpg_rat32 * tgtf, const xpg_rat32 * leq, const int32_t * rows,
                               const int32_t * cols, const long long * leq_offsets, const long long * tgtf_offsets,
                               unsigned max_iter, int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol)
{
    XPG_BIND(ctx);
    return six_batch_ragged<R32>(ctx, is_max, nb, (const R32 *)tgtf, (const R32 *)leq, rows, cols, leq_offsets, tgtf_offsets, max_iter,
                                 out_status, (R32 *)out_v, (R32 *)out_sol);
}
} // extern "C"
#endif
#if XPG_IN(3)
extern "C" {
// Lineq::reduce on systems of different shapes, in place (mats: the systems back to back, system b at cell
// offsets[b]); rhs_idx[b] (NULL: the last column of each).
int xpg_lineq_reduce_batch_ragged_rat32(xpg_ctx * ctx, int nb, xpg_rat32 * mats, const int32_t * rows, const int32_t * cols,
                                        const long long * offsets, const int32_t * rhs_idx, int is_intersect,
                                        int32_t * out_rows, int32_t * out_ok)
{
    XPG_BIND(ctx);
    if (!ctx || nb < 0 || !mats || !rows || !cols || !offsets || !out_rows || !out_ok) return XPG_ERR_SHAPE;
    // a class is (rows, cols, rhs): split the shape classes by constant column on the fly through the key of `cols`
    std::vector<int32_t> key((size_t)nb);
    for (int b = 0; b < nb; b++) {
        const int r = rhs_idx ? rhs_idx[b] : cols[b] - 1;
        if (cols[b] < 2 || cols[b] > 32767 || r < 0 || r >= cols[b]) return XPG_ERR_SHAPE;
        key[(size_t)b] = cols[b] | (r << 16);
    }
    return run_ragged(ctx, nb, rows, key.data(), [&](xpg_ctx * c, const RaggedClass & g) {
        const int gc = g.cols & 0xFFFF, rhs = g.cols >> 16;
        const size_t per = (size_t)g.rows * gc, ng = g.idx.size();
        std::vector<R32> buf; std::vector<int32_t> kr(ng), ko(ng);
        ragged_gather(buf, (const R32 *)mats, offsets, g, per);
        const int r = lineq_reduce_batch(c, (int)ng, buf.data(), g.rows, gc, rhs, 1, is_intersect, kr.data(), ko.data());
        if (r) return r;
        for (size_t k = 0; k < ng; k++) {
            const int b = g.idx[k];
            memcpy((void *)((R32 *)mats + offsets[b]), (const void *)(buf.data() + k * per), per * 8);
            out_rows[b] = kr[k]; out_ok[b] = ko[k];
        }
        return 0;
    });
}
// Lineq::fme on systems of different shapes, eliminating variable u[b] of system b. Packed result: system b's
// rows (cols[b] wide) start at cell out_cell_offsets[b] of outs (out_cell_offsets[nb] = cells in all); outs may
// be NULL or too small (outs_cap_cells): XPG_ERR_SHAPE with out_rows / out_cell_offsets filled.
int xpg_lineq_fme_batch_ragged_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, const int32_t * rows, const int32_t * cols,
                                     const long long * offsets, const int32_t * rhs_idx, const int32_t * u, int darkshadow,
                                     xpg_rat32 * outs, long long outs_cap_cells, long long * out_cell_offsets,
                                     int32_t * out_rows, int32_t * out_ok)
{
    XPG_BIND(ctx);
    if (!ctx || nb < 0 || !mats || !rows || !cols || !offsets || !u || !out_cell_offsets || !out_rows || !out_ok) return XPG_ERR_SHAPE;
    std::vector<int32_t> key((size_t)nb);
    for (int b = 0; b < nb; b++) {
        const int r = rhs_idx ? rhs_idx[b] : cols[b] - 1;
        if (cols[b] < 2 || cols[b] > 255 || r < 1 || r >= cols[b] || u[b] < 0 || u[b] >= r) return XPG_ERR_SHAPE;
        key[(size_t)b] = cols[b] | (r << 8) | (u[b] << 16);           // a class shares shape, constant column and variable
    }
    std::vector<std::vector<R32> > res((size_t)nb);
    const int rc = run_ragged(ctx, nb, rows, key.data(), [&](xpg_ctx * c, const RaggedClass & g) {
        const int gc = g.cols & 0xFF, rhs = (g.cols >> 8) & 0xFF, uu = g.cols >> 16;
        const size_t per = (size_t)g.rows * gc, ng = g.idx.size();
        std::vector<R32> buf; std::vector<int32_t> ko(ng); std::vector<long long> off(ng + 1);
        ragged_gather(buf, (const R32 *)mats, offsets, g, per);
        const R32 * view = 0;
        const int r = lineq_fme_batch_packed(c, (int)ng, buf.data(), g.rows, gc, rhs, uu, darkshadow, 0, (R32 *)0, 0, &view, off.data(), ko.data());
        if (r) return r;
        for (size_t k = 0; k < ng; k++) {
            const int b = g.idx[k];
            out_rows[b] = (int32_t)(off[k + 1] - off[k]); out_ok[b] = ko[k];
            res[(size_t)b].assign(view + off[k] * gc, view + off[k + 1] * gc);
        }
        return 0;
    });
    if (rc) return rc;
    long long tot = 0;
    for (int b = 0; b < nb; b++) { out_cell_offsets[b] = tot; tot += (long long)res[(size_t)b].size(); }
    out_cell_offsets[nb] = tot;
    if (!outs || outs_cap_cells < tot) return outs ? XPG_ERR_SHAPE : 0;
    for (int b = 0; b < nb; b++)
        if (!res[(size_t)b].empty()) memcpy((void *)((R32 *)outs + out_cell_offsets[b]), (const void *)res[(size_t)b].data(), res[(size_t)b].size() * 8);
    return 0;
}

} // extern "C"
#endif
#if XPG_IN(2)
extern "C" {
#ifdef XPG_STAMPS
// diagnostic builds only: reads and clears the tick sums of k_mip_tree (build, solve, feed)
int xpg_mip_debug(xpg_ctx * ctx, unsigned long long * out4)
{
    XPG_BIND(ctx);
    unsigned long long z[4] = {0};
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_mip_ticks), sizeof(z)) != hipSuccess) return XPG_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mip_ticks), z, sizeof(z)) != hipSuccess) return XPG_ERR_HIP;
    return 0;
}
#endif
// ---- MIP / has_solution -------------------------------------------------------------------------
int xpg_mip_maxm_rat32(xpg_ctx * ctx, const xpg_rat32 * tgtf, const xpg_rat32 * vc, int vc_rows,
                       const xpg_rat32 * eq, int eq_rows, const xpg_rat32 * leq, int leq_rows, int cols,
                       int is_bin, const uint8_t * ind, xpg_rat32 * out_v, xpg_rat32 * out_sol)
{
    XPG_BIND(ctx);
    return mip_solve<R32>(ctx, 1, true, is_bin != 0, (const R32 *)tgtf, (const R32 *)vc, vc_rows, (const R32 *)eq,
                          eq_rows, (const R32 *)leq, leq_rows, cols, ind, (R32 *)out_v, (R32 *)out_sol, 0);
}
int xpg_mip_minm_rat32(xpg_ctx * ctx, const xpg_rat32 * tgtf, const xpg_rat32 * vc, int vc_rows,
                       const xpg_rat32 * eq, int eq_rows, const xpg_rat32 * leq, int leq_rows, int cols,
                       int is_bin, const uint8_t * ind, xpg_rat32 * out_v, xpg_rat32 * out_sol)
{
    XPG_BIND(ctx);
    return mip_solve<R32>(ctx, 1, false, is_bin != 0, (const R32 *)tgtf, (const R32 *)vc, vc_rows, (const R32 *)eq,
                          eq_rows, (const R32 *)leq, leq_rows, cols, ind, (R32 *)out_v, (R32 *)out_sol, 0);
}
int xpg_mip_maxm_f64(xpg_ctx * ctx, const double * tgtf, const double * vc, int vc_rows, const double * eq,
                     int eq_rows, const double * leq, int leq_rows, int cols, int is_bin, const uint8_t * ind,
                     double * out_v, double * out_sol)
{
    XPG_BIND(ctx);
    return mip_solve<F64>(ctx, 0, true, is_bin != 0, (const F64 *)tgtf, (const F64 *)vc, vc_rows, (const F64 *)eq,
                          eq_rows, (const F64 *)leq, leq_rows, cols, ind, (F64 *)out_v, (F64 *)out_sol, 0);
}
int xpg_mip_minm_f64(xpg_ctx * ctx, const double * tgtf, const double * vc, int vc_rows, const double * eq,
                     int eq_rows, const double * leq, int leq_rows, int cols, int is_bin, const uint8_t * ind,
                     double * out_v, double * out_sol)
{
    XPG_BIND(ctx);
    return mip_solve<F64>(ctx, 0, false, is_bin != 0, (const F64 *)tgtf, (const F64 *)vc, vc_rows, (const F64 *)eq,
                          eq_rows, (const F64 *)leq, leq_rows, cols, ind, (F64 *)out_v, (F64 *)out_sol, 0);
}
int xpg_has_solution_rat32(xpg_ctx * ctx, const xpg_rat32 * leq, int leq_rows, const xpg_rat32 * eq, int eq_rows,
                           const xpg_rat32 * vc, int vc_rows, int cols, int rhs_idx, int is_int_sol,
                           int is_unique_sol)
{
    XPG_BIND(ctx);
    return has_solution(ctx, (const R32 *)leq, leq_rows, (const R32 *)eq, eq_rows, (const R32 *)vc, vc_rows, cols,
                        rhs_idx, is_int_sol != 0, is_unique_sol != 0);
}

int xpg_mip_batch_rat32(xpg_ctx * ctx, int nb, int is_max, int is_bin, const xpg_rat32 * tgtf, const xpg_rat32 * leq,
                        int leq_rows, int cols, int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol,
                        long long * out_nodes)
{
    XPG_BIND(ctx);
    return mip_batch<R32>(ctx, 1, nb, is_max != 0, is_bin != 0, (const R32 *)tgtf, (const R32 *)leq, leq_rows, cols,
                          out_status, (R32 *)out_v, (R32 *)out_sol, out_nodes);
}
int xpg_mip_batch_f64(xpg_ctx * ctx, int nb, int is_max, int is_bin, const double * tgtf, const double * leq,
                      int leq_rows, int cols, int32_t * out_status, double * out_v, double * out_sol, long long * out_nodes)
{
    XPG_BIND(ctx);
    return mip_batch<F64>(ctx, 0, nb, is_max != 0, is_bin != 0, (const F64 *)tgtf, (const F64 *)leq, leq_rows, cols,
                          out_status, (F64 *)out_v, (F64 *)out_sol, out_nodes);
}
int xpg_mip_batch_eq_rat32(xpg_ctx * ctx, int nb, int is_max, int is_bin, const xpg_rat32 * tgtf, const xpg_rat32 * leq,
                           int leq_rows, const xpg_rat32 * eq, int eq_rows, int cols, int32_t * out_status, xpg_rat32 * out_v,
                           xpg_rat32 * out_sol, long long * out_nodes)
{
    XPG_BIND(ctx);
    return mip_batch_eq<R32>(ctx, 1, nb, is_max != 0, is_bin != 0, (const R32 *)tgtf, (const R32 *)leq, leq_rows, (const R32 *)eq, eq_rows,
                             cols, out_status, (R32 *)out_v, (R32 *)out_sol, out_nodes);
}
int xpg_mip_batch_eq_f64(xpg_ctx * ctx, int nb, int is_max, int is_bin, const double * tgtf, const double * leq, int leq_rows,
                         const double * eq, int eq_rows, int cols, int32_t * out_status, double * out_v, double * out_sol,
                         long long * out_nodes)
{
    XPG_BIND(ctx);
    return mip_batch_eq<F64>(ctx, 0, nb, is_max != 0, is_bin != 0, (const F64 *)tgtf, (const F64 *)leq, leq_rows, (const F64 *)eq, eq_rows,
                             cols, out_status, (F64 *)out_v, (F64 *)out_sol, out_nodes);
}
int xpg_dep_is_empty_batch_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                                 int32_t * out_empty, long long * out_nodes)
{
    XPG_BIND(ctx);
    long n = 0;
    int rc = dep_is_empty_batch(ctx, nb, (const R32 *)mats, rows, cols, cols - 1, (const R32 *)0, out_empty, &n);
    if (out_nodes) *out_nodes = n;
    return rc;
}
int xpg_dep_is_empty_batch_ex_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                    const xpg_rat32 * vc, int32_t * out_empty, long long * out_nodes)
{
    XPG_BIND(ctx);
    long n = 0;
    int rc = dep_is_empty_batch(ctx, nb, (const R32 *)mats, rows, cols, rhs_idx, (const R32 *)vc, out_empty, &n);
    if (out_nodes) *out_nodes = n;
    return rc;
}
int xpg_dep_is_empty_batch_mode_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                      const xpg_rat32 * vc, int mode, int32_t * out_empty, long long * out_nodes)
{
    XPG_BIND(ctx);
    if (mode != XPG_DEP_PARITY && mode != XPG_DEP_SYMBOLS_AS_VARS) return XPG_ERR_SHAPE;
    long n = 0;
    int rc = dep_is_empty_batch(ctx, nb, (const R32 *)mats, rows, cols, rhs_idx, (const R32 *)vc, out_empty, &n, mode == XPG_DEP_SYMBOLS_AS_VARS ? 1 : 0);
    if (out_nodes) *out_nodes = n;
    return rc;
}
} // extern "C"
#endif
#if XPG_IN(3)
extern "C" {
#ifdef XPG_STAMPS
// diagnostic builds only: reads and clears the phase tick sums of k_fme_batch
int xpg_lineq_debug(xpg_ctx * ctx, unsigned long long * out16)
{
    XPG_BIND(ctx);
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lq_ticks), sizeof(z)) != hipSuccess) return XPG_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lq_ticks), z, sizeof(z)) != hipSuccess) return XPG_ERR_HIP;
    return 0;
}
#endif
int xpg_lineq_move2var_batch_rat32(xpg_ctx * ctx, int nb, xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                   int first_sym, int last_sym)
{
    if (!ctx || nb < 0 || !mats || rows <= 0 || cols < 2 || rhs_idx < 0 || first_sym <= rhs_idx || last_sym < first_sym ||
        last_sym >= cols)
        return XPG_ERR_SHAPE;                           // the reference's ASSERT (linsys.cpp:1185-1188)
    std::vector<R32> tmp((size_t)rows * cols);
    for (int b = 0; b < nb; b++) {
        R32 * m = (R32 *)mats + (size_t)b * rows * cols;
        move2var_one(m, tmp.data(), rows, cols, rhs_idx, first_sym, last_sym);
        memcpy(m, tmp.data(), sizeof(R32) * (size_t)rows * cols);
    }
    return 0;
}

// ---- rational row elimination ---------------------------------------------------------------
int xpg_lineq_reduce_batch_rat32(xpg_ctx * ctx, int nb, xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                 int is_intersect, int32_t * out_rows, int32_t * out_ok)
{
    XPG_BIND(ctx); return lineq_reduce_batch(ctx, nb, (R32 *)mats, rows, cols, rhs_idx, 1, is_intersect, out_rows, out_ok); }
int xpg_lineq_reduce_batch_packed_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                        int is_intersect, xpg_rat32 * outs, long long outs_cap_rows,
                                        const xpg_rat32 ** out_view, long long * row_offsets, int32_t * out_rows,
                                        int32_t * out_ok)
{
    XPG_BIND(ctx);
    return lineq_reduce_batch_packed(ctx, nb, (const R32 *)mats, rows, cols, rhs_idx, 1, is_intersect, (R32 *)outs, outs_cap_rows,
                                     (const R32 **)out_view, row_offsets, out_rows, out_ok);
}
int xpg_lineq_remove_iden_batch_rat32(xpg_ctx * ctx, int nb, xpg_rat32 * mats, int rows, int cols,
                                      int32_t * out_rows)
{
    XPG_BIND(ctx); return lineq_reduce_batch(ctx, nb, (R32 *)mats, rows, cols, 0, 0, 1, out_rows, 0); }
int xpg_lineq_fme_batch_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                              int u, int darkshadow, xpg_rat32 * outs, int cap_rows, int32_t * out_rows,
                              int32_t * out_ok)
{
    XPG_BIND(ctx);
    return lineq_fme_batch(ctx, nb, (const R32 *)mats, rows, cols, rhs_idx, u, darkshadow, (R32 *)outs, cap_rows,
                           out_rows, out_ok);
}
int xpg_lineq_calc_bound_batch_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                     int cap_rows, xpg_rat32 * bounds, int32_t * out_rows, int32_t * out_ok)
{
    XPG_BIND(ctx); return lineq_calc_bound_batch(ctx, nb, (const R32 *)mats, rows, cols, rhs_idx, cap_rows, (R32 *)bounds, out_rows, out_ok); }
int xpg_lineq_calc_bound_batch_packed_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                            int cap_rows, xpg_rat32 * outs, long long outs_cap_rows, const xpg_rat32 ** out_view,
                                            long long * row_offsets, int32_t * out_ok)
{
    XPG_BIND(ctx);
    if (!row_offsets) return XPG_ERR_SHAPE;
    if (cap_rows <= 0) cap_rows = 4 * rows + 16;
    return lineq_calc_bound_batch(ctx, nb, (const R32 *)mats, rows, cols, rhs_idx, cap_rows, (R32 *)outs, (int32_t *)0, out_ok,
                                  row_offsets, outs_cap_rows, (const R32 **)out_view);
}
int xpg_lineq_reduce_batch_rat32_dev(xpg_ctx * ctx, int nb, xpg_rat32 * d_mats, int rows, int cols, int rhs_idx,
                                     int is_intersect, int32_t * d_out_rows, int32_t * d_out_ok)
{
    XPG_BIND(ctx); return lineq_reduce_batch_dev(ctx, nb, (R32 *)d_mats, rows, cols, rhs_idx, 1, is_intersect, d_out_rows, d_out_ok); }
int xpg_lineq_fme_batch_packed_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx, int u,
                                     int darkshadow, int cap_rows, xpg_rat32 * outs, long long outs_cap_rows,
                                     const xpg_rat32 ** out_view, long long * row_offsets, int32_t * out_ok)
{
    XPG_BIND(ctx);
    return lineq_fme_batch_packed(ctx, nb, (const R32 *)mats, rows, cols, rhs_idx, u, darkshadow, cap_rows, (R32 *)outs,
                                  outs_cap_rows, (const R32 **)out_view, row_offsets, out_ok);
}
int xpg_lineq_fme_batch_rat32_dev(xpg_ctx * ctx, int nb, const xpg_rat32 * d_mats, int rows, int cols, int rhs_idx,
                                  int u, int darkshadow, xpg_rat32 * d_outs, int cap_rows, int32_t * d_out_rows,
                                  int32_t * d_out_ok)
{
    XPG_BIND(ctx); return lineq_fme_batch_dev(ctx, nb, (const R32 *)d_mats, rows, cols, rhs_idx, u, darkshadow, (R32 *)d_outs,
                                              cap_rows, d_out_rows, d_out_ok); }
int xpg_rat_rank_batch_dev(xpg_ctx * ctx, int nb, const xpg_rat32 * d_mats, int rows, int cols, int32_t * d_out_rank)
{
    XPG_BIND(ctx); return rat_rank_batch_dev(ctx, nb, (const R32 *)d_mats, rows, cols, d_out_rank); }
int xpg_rat_rank_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int32_t * out_rank)
{
    XPG_BIND(ctx); return out_rank ? gauss_batch(ctx, nb, (const R32 *)mats, rows, cols, 0, out_rank, 0, 0) : XPG_ERR_SHAPE; }
int xpg_rat_det_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int n, xpg_rat32 * out_det)
{
    XPG_BIND(ctx); return out_det ? gauss_batch(ctx, nb, (const R32 *)mats, n, n, 1, 0, (R32 *)out_det, 0) : XPG_ERR_SHAPE; }
int xpg_rat_inv_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int n, xpg_rat32 * out_inv, int32_t * out_ok)
{
    XPG_BIND(ctx); return (out_inv && out_ok) ? gauss_batch(ctx, nb, (const R32 *)mats, n, n, 2, out_ok, 0, (R32 *)out_inv) : XPG_ERR_SHAPE; }
int xpg_rat_rank_basis_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int is_unitarize,
                             int32_t * out_rank, xpg_rat32 * basis, int32_t * basis_rows)
{
    XPG_BIND(ctx);
    if (!out_rank || !basis || !basis_rows) return XPG_ERR_SHAPE;
    const int st = gauss_batch(ctx, nb, (const R32 *)mats, rows, cols, 3, out_rank, 0, (R32 *)basis, is_unitarize ? 1 : 0);
    if (st != 0) return st;
    for (int b = 0; b < nb; b++) basis_rows[b] = (!is_unitarize && out_rank[b] < rows) ? out_rank[b] : rows;
    return 0;
}
int xpg_rat_null_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, xpg_rat32 * ns)
{
    XPG_BIND(ctx); return ns ? gauss_batch(ctx, nb, (const R32 *)mats, rows, cols, 4, 0, 0, (R32 *)ns) : XPG_ERR_SHAPE; }
int xpg_int_hnf_batch(xpg_ctx * ctx, int nb, const int32_t * mats, int rows, int cols, int32_t * h, int32_t * u,
                      int32_t * status)
{
    XPG_BIND(ctx); return int_hnf_batch(ctx, nb, mats, rows, cols, h, u, status); }
int xpg_int_gcd_batch(xpg_ctx * ctx, int nb, int32_t * mats, int rows, int cols)
{
    XPG_BIND(ctx); return int_gcd_batch(ctx, nb, mats, rows, cols); }

} // extern "C"
#endif // part 3
