// Host orchestration of the device-resident LP (xpg_lp): owns the HBM buffers,
// queues the kernels of lp_kernels.hip.h on the context's stream and mirrors the
// control flow of SIX::TwoStageMethod / stage1 / constructBasicFeasibleSolution
// (src/com/lpsol.h:1907-1930, :1784-1844, :839-988). No arithmetic on tableau
// data happens on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
#include <stdio.h>
#include "../../include/xpoly_amd.h"
#include "ctx.hip.h"
#include "lp_kernels.hip.h"
#include "lp_blocked.hip.h"
#include "lp_chain.hip.h"

namespace xpg {

// fp64 sweep launch. Variants are (rows per workgroup, rows in flight); the
// default is what measured best on MI355X (profiles/), the others stay
// reachable through XPG_UPDATE_VARIANT for A/B runs.
inline void launch_update_f64(hipStream_t s, int variant, double * tab, int m, int W, int ld,
                              const double * rowbuf, const double * colbuf,
                              LoopState * st, int guarded, double * nextcol, double * bcol,
                              int rhs)
{
    const int strips = (W + 511) / 512;
#define XPG_LAUNCH(R, U)                                                                    \
    hipLaunchKernelGGL((k_update_f64<R, U>), dim3(strips, (m + R - 1) / R), dim3(256), 0, s, \
                       tab, m, W, ld, rowbuf, colbuf, st, guarded, nextcol, bcol, rhs)
    switch (variant) {
    case 1: XPG_LAUNCH(16, 4); break;
    case 2: XPG_LAUNCH(32, 4); break;
    case 3: XPG_LAUNCH(64, 8); break;
    case 4: XPG_LAUNCH(16, 8); break;
    case 5: XPG_LAUNCH(128, 8); break;
    case 6: XPG_LAUNCH(8, 8); break;
    case 7: XPG_LAUNCH(64, 16); break;
    default: XPG_LAUNCH(32, 8); break;
    }
#undef XPG_LAUNCH
}

template <class S> inline void launch_update(xpg_ctx * ctx, const LpView<S> & v, int guarded);
inline bool prof_open(xpg_ctx * ctx)
{
    if (ctx->prof_n >= ctx->prof_cap) return false;
    if ((ctx->prof_seen++ % ctx->prof_stride) != 0) return false;
    (void)hipEventRecord(ctx->ev0[ctx->prof_n], ctx->stream);
    return true;
}
inline void prof_close(xpg_ctx * ctx)
{
    (void)hipEventRecord(ctx->ev1[ctx->prof_n], ctx->stream);
    ctx->prof_n++;
}
template <> inline void launch_update<F64>(xpg_ctx * ctx, const LpView<F64> & v, int guarded)
{
    const bool timed = prof_open(ctx);
    launch_update_f64(ctx->stream, ctx->update_variant, (double *)v.tab, v.m, v.W, v.ld,
                      (const double *)v.rowbuf, (const double *)v.colbuf, v.st, guarded,
                      (double *)v.nextcol, (double *)v.bcol, v.rhs);
    if (timed) prof_close(ctx);
}
template <class S> inline void launch_pipe_sweep(xpg_ctx *, const LpView<S> &, int, int, bool) {}
template <> inline void launch_pipe_sweep<F64>(xpg_ctx * ctx, const LpView<F64> & v, int slot, int colstride,
                                               bool sample)
{
    const int strips = (v.W + 511) / 512;
    const int fused = ctx->loop_mode == 2 ? 0 : 1;
    // a sampled launch carries its own start/stop events (timestamps of the dispatch itself, as
    // rocprofv3 reports them) instead of being bracketed by two marker packets
    const bool timed = sample && ctx->prof_n < ctx->prof_cap && (ctx->prof_seen++ % ctx->prof_stride) == 0;
    hipEvent_t e0 = timed ? ctx->ev0[ctx->prof_n] : nullptr, e1 = timed ? ctx->ev1[ctx->prof_n] : nullptr;
    hipExtLaunchKernelGGL((k_pipe_sweep<32, 8>), dim3(strips, (v.m + 31) / 32 + 1), dim3(256), 0, ctx->stream, e0, e1, 0,
                          v, slot, colstride, fused, (double *)v.tab, (const double *)v.rowbuf,
                          (const double *)v.colbuf + (size_t)slot * colstride, ctx->zigzag ? (slot & 1) : 0);
    if (timed) ctx->prof_n++;
    if (!fused)
        hipLaunchKernelGGL((k_pipe_pick<0>), dim3(strips < PICK_MAX_WGS ? strips : PICK_MAX_WGS), dim3(256), 0, ctx->stream,
                           v, slot, colstride);
}
template <> inline void launch_pipe_sweep<R32>(xpg_ctx * ctx, const LpView<R32> & v, int slot, int colstride, bool sample)
{
    const bool timed = sample && prof_open(ctx);
    const int N = (v.m + 255) / 256 < PICK_MAX_WGS ? (v.m + 255) / 256 : PICK_MAX_WGS;
    hipLaunchKernelGGL(k_pipe_sweep_r32, dim3(v.m > N ? v.m : N, 1 + (v.W + 255) / 256), dim3(256), 0, ctx->stream, v, slot, colstride, N);
    if (timed) prof_close(ctx);
}
// Fused Rational loop (lp_fused_r32.hip.h): one launch per pivot; generic = the generic point before it (generic pick in
// place if the descriptor is idle, then staging of what it chose)
template <class S> inline void launch_fused(xpg_ctx *, const LpView<S> &, int, int, bool) {}
template <class S> inline void launch_side_home(xpg_ctx *, const LpView<S> &) {}
template <> inline void launch_fused<R32>(xpg_ctx * ctx, const LpView<R32> & v, int slot, int colstride, bool generic)
{
    const int NP = (v.W + 255) / 256;
    if (generic) hipLaunchKernelGGL(k_fused_generic, dim3(1), dim3(1024), 0, ctx->stream, v, slot, colstride);
    const bool timed = prof_open(ctx);
    const int N = (v.m + 63) / 64 < PICK_MAX_WGS ? (v.m + 63) / 64 : PICK_MAX_WGS;     // one-wave pick workgroups
    hipLaunchKernelGGL(k_pipe_fused_r32, dim3(v.m > N + NP ? v.m : N + NP, 1 + NP), dim3(256), 0, ctx->stream, v, slot, colstride, N, NP);
    if (timed) prof_close(ctx);
}
template <> inline void launch_side_home<R32>(xpg_ctx * ctx, const LpView<R32> & v)
{ hipLaunchKernelGGL(k_side_home, dim3(v.m, (v.W + 255) / 256), dim3(256), 0, ctx->stream, v); }
// What a batch of the blocked loop launches for an fp64 tableau of a given shape -- decided in ONE place: launch_blk_batch
// launches it, Lp::blocked_from_bytes asks whether the chain can run at all (the automatic loop choice), xpg_lp_loop_info
// reports it (bench.py's `shapes` leg).
// workgroup sizes of pick and prep: 64 = one wave per workgroup, no LDS round in the reductions
// (measured at 4096 x 8192: 58.7 k pivots/s with a 64-thread pick against 55.7 k with 256; prep +0.8 %)
inline int blk_tpb_pick() { static const int t = [] { const char * s = xpg_hook("XPG_BLK_TPB_PICK"); return s && atoi(s) == 256 ? 256 : 64; }(); return t; }
inline int blk_tpb_prep() { static const int t = [] { const char * s = xpg_hook("XPG_BLK_TPB_PREP"); return s && atoi(s) == 256 ? 256 : 64; }(); return t; }
// Stages 1 .. B-1 in one persistent launch: one one-wave worker per 64 rows / 64 columns, all resident at once -- they
// poll each other's records. The launch checks that itself (roll call, lp_chain.hip.h) and falls back when the device
// cannot seat them all, so the shape limits here are only those of the hand-off areas: 256 records (m <= 16 384), 511
// partial slots (W <= 32 640), and at most eight workers per CU. Taller or wider tableaux, and the opt-in Dantzig pricing,
// take the launch-per-stage path.
// shapes the blocked loop's hand-off areas can hold at all (launch-per-stage form: 510 partial slots of up to 256 columns)
inline bool blk_shape_ok(int W) { return (W + 255) / 256 <= 510; }
inline bool chain_shape_ok(const xpg_ctx * ctx, int m, int W)
{
    const int cpick = (m + 63) / 64, cprep = (W + 63) / 64;
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 1;
    return cpick <= BLK_REC_MAX && cprep <= 510 && cpick + cprep + 1 <= 8 * cus &&
           blk_tpb_prep() == 64;                   // stage 0's prep leaves one look-ahead partial per 64 columns, as the chain's workers do
}
struct BlkPlan {
    bool chain;            // a batch's stages in one persistent launch (else pick / prep kernels per stage)
    bool local;            // ... every worker on one XCD, hand-offs through that XCD's L2 (else the spread form, sc1 stores)
    int line;              // ... the entering column's line in the pick workers' LDS: 0 / 8 / 16 elements
    int cpick, cprep, workers;
    size_t lds;
    int sweep_rows;        // rows per workgroup of the full-batch pass (16 / 32)
};
inline BlkPlan blk_plan(const xpg_ctx * ctx, int m, int W, int ld, int B, bool ref_pricing, bool chain_off, bool chain_spread, bool fold)
{
    BlkPlan p;
    p.cpick = (m + 63) / 64; p.cprep = (W + 63) / 64; p.workers = p.cpick + p.cprep + 1;
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 1;
    p.chain = ctx->chain && !chain_off && ref_pricing && (B > 1 || fold) && chain_shape_ok(ctx, m, W);
    // every worker on one XCD, hand-offs through that XCD's L2 (lp_chain.hip.h) -- where one XCD seats them all: a
    // worker is one wave with 16 KB of LDS, an XCD has cus / 8 CUs of 160 KB; XPG_CHAIN_XCD=0, or a launch whose
    // placement check failed, selects the spread form with sc1 stores
    p.local = p.chain && ctx->chain_local && !chain_spread && p.workers <= (cus / 8) * 9;
    // the entering column's line in the pick workers' LDS (lp_chain.hip.h, LINE): where the row stride is a multiple of
    // 4 KiB -- the L1 then keeps next to nothing of a column -- and one XCD still seats every worker with the larger
    // LDS block; XPG_CHAIN_LINE=0|1|8 forces it off / on / on as a half line for A/B runs and tests
    static const int line_env = [] { const char * s = xpg_hook("XPG_CHAIN_LINE"); return s ? atoi(s) : -1; }();
    auto line_fits = [&](int cols) {
        return (size_t)p.workers * ch_lds_bytes(B, cols) <= (size_t)(cus / 8) * 160 * 1024 &&
               (size_t)p.workers <= (size_t)(cus / 8) * ((size_t)160 * 1024 / ch_lds_bytes(B, cols));
    };
    const bool want_line = p.local && (line_env == 1 || line_env == 8 || (line_env != 0 && ld % 512 == 0));      // (8: the half line wherever it fits, for tests)
    p.line = !want_line ? 0 : (line_env != 8 && line_fits(16)) ? 16 : line_fits(8) ? 8 : 0;
    p.lds = ch_lds_bytes(B, p.line);
    // 32 stages: 16 rows per workgroup where the tableau is of the Infinity Cache's size, 32 rows beyond it (see the launch)
    p.sweep_rows = (size_t)m * ld * sizeof(double) > ((size_t)320 << 20) ? 32 : 16;
    return p;
}
// One batch of the blocked loop: B x (pick, prep) -- the generic pick once after pick(0) -- and a sweep.
// fold: this batch's stage 0 is the chain launch's (t0 = 0) -- no pick / generic pick / prep launches. Speculative: the chain
// launch of the batch before admits it by a ticket once it has committed all its stages; without the ticket this batch's
// launches do nothing and the host, which sees the iteration budget not spent at its next status read, enqueues the rest
// with stage-0 launches again (Lp::iterate / run_loop). next_folds: the batch after this one will be enqueued that way.
template <class S> inline void launch_blk_batch(xpg_ctx *, const LpView<S> &, int, int, bool, bool, bool, bool, bool, bool) {}
template <> inline void launch_blk_batch<F64>(xpg_ctx * ctx, const LpView<F64> & v, int batch, int B, bool ref_pricing, bool closes_often, bool chain_off,
                                              bool chain_spread, bool fold, bool next_folds)
{
    const int strips = (v.W + 511) / 512;
    // prep workgroups leave one look-ahead partial each and there are 510 partial slots (BLK_PART_MAX): one wave per 64 columns
    // up to W = 32 640, four beyond (W <= 130 560; wider tableaux do not take the blocked loop, blk_shape_ok). Round 6: a
    // 1024 x 33793 tableau ran 529 one-wave prep workgroups over the 520 slots -- a memory fault.
    const int tq = (v.W + blk_tpb_prep() - 1) / blk_tpb_prep() <= 510 ? blk_tpb_prep() : 256, tp = blk_tpb_pick();
    const int want_pick = (v.m + tp - 1) / tp;
    const int npick = want_pick < BLK_PICK_WGS ? want_pick : BLK_PICK_WGS;
    const dim3 gprep((v.W + tq - 1) / tq);
    const BlkPlan pl = blk_plan(ctx, v.m, v.W, v.ld, B, ref_pricing, chain_off, chain_spread, fold);
    const int cpick = pl.cpick, cprep = pl.cprep, workers = pl.workers;
    for (int t = 0; t < B; t++) {
        if (pl.chain && (t == 1 || fold)) {
            const int t0 = fold ? 0 : 1;
            const int test_abort = ctx->chain_test_abort;
            const bool local = pl.local;
            // test hooks: k > 0 -- every k-th launch fails its roll call; k < 0 -- every |k|-th ONE-XCD launch "finds" its
            // workers on several XCDs (the spread form has no placement to check)
            const int fa = test_abort > 0 ? (batch % test_abort == test_abort - 1 ? 1 : 0)
                                          : (test_abort < 0 && local && batch % -test_abort == -test_abort - 1 ? 2 : 0);
            const int nparts0 = (int)gprep.x;
            const int fn = next_folds ? 1 : 0;
            const int line = pl.line;
            const size_t lds = pl.lds;
            if (local && line == 16) hipLaunchKernelGGL((k_blk_chain<true, 16>), dim3(8 * workers), dim3(64), lds, ctx->stream, v, batch, t0, B, cpick, cprep, nparts0, fa, fn);
            else if (local && line == 8) hipLaunchKernelGGL((k_blk_chain<true, 8>), dim3(8 * workers), dim3(64), lds, ctx->stream, v, batch, t0, B, cpick, cprep, nparts0, fa, fn);
            else if (local) hipLaunchKernelGGL((k_blk_chain<true, 0>), dim3(8 * workers), dim3(64), lds, ctx->stream, v, batch, t0, B, cpick, cprep, nparts0, fa, fn);
            else hipLaunchKernelGGL((k_blk_chain<false, 0>), dim3(workers), dim3(64), lds, ctx->stream, v, batch, t0, B, cpick, cprep, nparts0, fa, fn);
            break;
        }
        hipLaunchKernelGGL(k_blk_pick, dim3(npick), dim3(tp), 0, ctx->stream, v, batch, t, (int)gprep.x);
        if (t == 0) hipLaunchKernelGGL(k_blk_pick_generic, dim3(1), dim3(1024), 0, ctx->stream, v, batch, (int)gprep.x);
        hipLaunchKernelGGL(k_blk_prep, gprep, dim3(tq), 0, ctx->stream, v, batch, t);
    }
    // only sweeps of full-length batches are sampled by xpg_profile_begin / end (a shorter batch -- the tail
    // of an xpg_lp_iterate budget -- moves the same bytes for fewer pivots and runs a different kernel)
    const bool timed = B == ctx->block_len && ctx->prof_n < ctx->prof_cap && (ctx->prof_seen++ % ctx->prof_stride) == 0;
    hipEvent_t e0 = timed ? ctx->ev0[ctx->prof_n] : nullptr, e1 = timed ? ctx->ev1[ctx->prof_n] : nullptr;
    static const int rows_env = [] { const char * s = xpg_hook("XPG_BLK_ROWS"); return s ? atoi(s) : 32; }();
    // alternate passes walk the row blocks in opposite directions (Infinity Cache reuse across passes); XPG_SERPENTINE=0 for A/B runs
    static const int serpentine = [] { const char * s = xpg_hook("XPG_SERPENTINE"); return s ? atoi(s) : 1; }();
#define XPG_BLK_LAUNCH(ROWS_, UNR_, CAP_)                                                                                 \
    hipExtLaunchKernelGGL((k_blk_sweep<ROWS_, UNR_, CAP_>), dim3(strips, (v.m + ROWS_ - 1) / ROWS_), dim3(256), 0,        \
                          ctx->stream, e0, e1, 0, (double *)v.tab, v.m, v.W, v.ld, (const double *)v.blkE,                \
                          (const double *)v.blkK, v.st, batch, 0, ctx->block_len)
#define XPG_BLK_FULL(ROWS_, UNR_, NB_)                                                                                    \
    hipExtLaunchKernelGGL((k_blk_sweep_full<ROWS_, UNR_, NB_>), dim3(blk_sweep_grid(strips, (v.m + ROWS_ - 1) / ROWS_)), \
                          dim3(256), 0, ctx->stream, e0, e1, 0, (double *)v.tab, v.m, v.W, v.ld,                          \
                          (const double *)v.blkE, (const double *)v.blkK, v.st, batch, closes_often ? 1 : 0,              \
                          serpentine ? (batch & 1) : 0, ctx->block_len)
    // the full-batch kernels: 32 pivots per pass (the default; two register sets of e_s, 2-3 waves per SIMD: 110 / 166 us
    // per pass at 4096 x 8192 / 4096 x 12289 = 3.4 / 5.2 us per pivot against 4.9 / 7.9 with 16, tools/lab/sweep_lab2.hip) and
    // 16 (XPG_BLOCK=16); every other length -- the tail of an iteration budget -- goes through the switch kernel
    const bool full32 = B == 32 && rows_env != 1, full16 = B == 16 && rows_env != 1, full24 = B == 24 && rows_env != 1;
    // 32 stages: 16 rows per workgroup where the tableau is of the Infinity Cache's size (4096 x 8192: 93.9 us against 95.5
    // with 32 rows), 32 rows -- the e_s read from the L2 half as often -- where it is beyond it (4096 x 12289, 403 MB: 150.0
    // against 156.1 us, 105.7 k against 103.9 k pivots/s); XPG_BLK_ROWS = 162 / 322 / 164 force a form for A/B runs
    const bool beyond_mall = pl.sweep_rows == 32;
    if (full32) {
        if (rows_env == 322 || (rows_env == 32 && beyond_mall)) XPG_BLK_FULL(32, 2, 32);
        else if (rows_env == 164) XPG_BLK_FULL(16, 4, 32);
        else XPG_BLK_FULL(16, 2, 32);
    }
    else if (full24) { if (rows_env == 164) XPG_BLK_FULL(16, 4, 24); else XPG_BLK_FULL(16, 2, 24); }
    else if (full16) { if (rows_env == 324) XPG_BLK_FULL(32, 4, 16); else if (rows_env == 162) XPG_BLK_FULL(16, 2, 16); else XPG_BLK_FULL(16, 4, 16); }
    else if (B <= 8) XPG_BLK_LAUNCH(32, 8, 8);
    else XPG_BLK_LAUNCH(32, 4, 16);
    // An LP whose batches often close early (a rare branch of solveSlackForm met with pivots staged: 34-44 % of
    // the sweeps on whole solves of 300 x 300 and 1024 x 1500 LPs, 1 of 261 on the bench LP,
    // tools/lab/probe_partial_batches.py) gets a second launch with the stage count as a template switch for those
    // batches; the full-batch kernel above then leaves them alone.
    if (closes_often && (full32 || full24 || full16))
        hipLaunchKernelGGL((k_blk_sweep<32, 4, 16>), dim3(strips, (v.m + 31) / 32), dim3(256), 0, ctx->stream, (double *)v.tab,
                           v.m, v.W, v.ld, (const double *)v.blkE, (const double *)v.blkK, v.st, batch, B, ctx->block_len);
#undef XPG_BLK_LAUNCH
#undef XPG_BLK_FULL
    if (timed) ctx->prof_n++;
}
template <> inline void launch_update<R32>(xpg_ctx * ctx, const LpView<R32> & v, int guarded)
{
    const bool timed = prof_open(ctx);
    static const int rows = [] { const char * s = xpg_hook("XPG_R32_ROWS"); return s ? atoi(s) : 1; }();   // A/B knob: 8 / 4 / 2 / 1 rows per thread measured 15.6 / 16.5 / 16.8 / 17.2 k pivots/s at 1024 x 2048
    if (rows <= 1) hipLaunchKernelGGL((k_update_r32<1>), dim3(v.m, (v.W + 255) / 256), dim3(256), 0, ctx->stream, v, guarded);
    else if (rows == 2) hipLaunchKernelGGL((k_update_r32<2>), dim3((v.m + 1) / 2, (v.W + 255) / 256), dim3(256), 0, ctx->stream, v, guarded);
    else if (rows == 4) hipLaunchKernelGGL((k_update_r32<4>), dim3((v.m + 3) / 4, (v.W + 255) / 256), dim3(256), 0, ctx->stream, v, guarded);
    else hipLaunchKernelGGL((k_update_r32<8>), dim3((v.m + 7) / 8, (v.W + 255) / 256), dim3(256), 0, ctx->stream, v, guarded);
    if (timed) prof_close(ctx);
}

struct LpBase {
    virtual ~LpBase() {}
    xpg_ctx * ctx;
    int kind;
};

template <class S> struct Lp : LpBase {
    LpView<S> v;
    int n0;                 // structural variables of the input problem
    int m;
    S * d_leq; S * d_tgtf;  // the input, kept for the phase-1 objective rebuild
    S * d_maxv;
    std::vector<void *> owned;
    size_t tab_elems;
    int ld_cap = 0;         // leading dimension the buffers were allocated for (>= v.ld)
    int row_cap = 0;        // rows the tableau has room for (m + extra)
    bool began;
    int final_status;
    hipEvent_t throttle[2] = {nullptr, nullptr};
    unsigned pipe_t = 0;    // pipelined loop: iteration counter since reset_loop (slot = pipe_t & 1)
    bool irregular = false, irregular_known = false;   // fp64: the input held an inf / NaN (read back once per build)
    bool pipe_primed = false;
    LoopState * h_state = nullptr;    // pinned mirror for read_state
    bool fused_unavailable = false;   // fused Rational loop: its buffers could not be allocated
    bool fused_idles_often = false;   // ... this solve defers decisions often: a generic point before every launch
    unsigned fused_idle_seen = 0;
    int blk_batch = 0;      // blocked loop: id of the next batch since reset_loop
    bool closes_often = false;   // blocked loop: >= 5 % of this solve's sweeps so far were of a batch closed early
    bool chain_off = false;      // blocked loop: a chain launch of this solve failed its roll call (the device is shared): launch per stage from here on
    bool blocked_now = false;    // the last queue_iterations call went through the blocked loop
    bool chain_spread = false;   // blocked loop: a one-XCD chain launch found its workers on several XCDs: the spread (sc1) form from here on
    unsigned chain_misplaced_seen = 0;
    unsigned chain_aborts_seen = 0;
    int colstride = 0;      // elements per colbuf half
    int opt_pricing = 0;    // xpg_lp_set_options: 0 the reference's rule, 1 Dantzig (non-parity)
    double opt_feas_tol = 0.0;

    // Opt-in non-parity modes: fp64 pipelined loop only.
    int set_options(int pricing, double feas_tol)
    {
        if (pricing < 0 || pricing > 1 || !(feas_tol >= 0.0)) return XPG_ERR_SHAPE;
        if ((pricing != 0 || feas_tol != 0.0) && (!std::is_same<S, F64>::value || ctx->loop_mode == 1))
            return XPG_ERR_UNSUPPORTED;
        opt_pricing = pricing; opt_feas_tol = feas_tol;
        return 0;
    }

    // create() plans its ~30 device arrays and gets them as ONE allocation (alloc_commit): a handle made and released per
    // SIX::maxm call paid ~1 ms of hipMalloc and ~1 ms of hipFree calls around a 15 ms solve (bench.py six_e2e)
    struct AllocReq { void ** p; size_t bytes; };
    std::vector<AllocReq> plan;
    bool planning = false;
    int alloc_commit()
    {
        planning = false;
        size_t total = 0;
        for (const AllocReq & r : plan) total += (r.bytes + 255) & ~(size_t)255;
        void * base = nullptr;
        const int rc = alloc(&base, total);
        if (rc) { plan.clear(); return rc; }
        size_t off = 0;
        for (const AllocReq & r : plan) { *r.p = (char *)base + off; off += (r.bytes + 255) & ~(size_t)255; }
        plan.clear();
        return 0;
    }
    int alloc(void ** p, size_t bytes)
    {
        if (planning) { plan.push_back(AllocReq{p, bytes ? bytes : 8}); return 0; }
        hipError_t e = hipMalloc(p, bytes ? bytes : 8);
        if (e != hipSuccess && !ctx->dev_cache.empty()) {       // the blocks parked by host-array row-elimination calls make room
            for (auto & b : ctx->dev_cache) (void)hipFree(b.first);
            ctx->dev_cache.clear(); ctx->dev_cache_bytes = 0;
            e = hipMalloc(p, bytes ? bytes : 8);
        }
        if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return XPG_ERR_ALLOC; }
        owned.push_back(*p);
        return 0;
    }
    ~Lp()
    {
        for (void * p : owned) (void)hipFree(p);
        if (h_state) (void)hipHostFree(h_state);
        for (hipEvent_t e : throttle) if (e) (void)hipEventDestroy(e);
    }

    // `extra`: rows (and their slack columns) that may be appended later (warm-started branch and bound, warm_mip.hip.h)
    int create(const void * leq, int m_, int cols, const void * tgtf, const void * vcd,
               const void * vcr, int on_dev, int extra = 0)
    {
        m = m_; n0 = cols - 1; began = false; final_status = XPG_RUNNING;
        const int mcap = m + extra;
        const int Wmax = n0 + 1 + mcap + 1;         // with the phase-1 column
        // the allocation's leading dimension; build() picks the one in use from the width the slack form really has
        // (with or without the phase-1 column), so that a 4096 x 8192 tableau runs at ld = 8192, not 8208
        const int ld = pick_ld(Wmax) > pick_ld(Wmax - 1) ? pick_ld(Wmax) : pick_ld(Wmax - 1);
        ld_cap = ld;
        const int nmax = Wmax - 1;
        v.m = m; v.ld = ld; v.W = 0; v.rhs = 0; v.tab2 = nullptr; v.stage = nullptr;
        v.pw = (nmax + 31) / 32;
        v.trace_cap = 1 << 16;
        tab_elems = (size_t)mcap * ld;
        row_cap = mcap;
        int rc;
        planning = true;
        if ((rc = alloc((void **)&v.tab, tab_elems * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.obj, (size_t)ld * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.rowbuf, (size_t)ld * sizeof(S)))) return rc;
        colstride = round_up(mcap, 16);             // two halves: the pipelined loop double-buffers -column
        if ((rc = alloc((void **)&v.colbuf, (size_t)2 * colstride * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.x, (size_t)ld * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.nextcol, (size_t)round_up(mcap, 16) * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.bcol, (size_t)round_up(mcap, 16) * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.vcd, (size_t)ld * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.vcr, (size_t)ld * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.nv, ld))) return rc;
        if ((rc = alloc((void **)&v.bv, ld))) return rc;
        if ((rc = alloc((void **)&v.bv2eq, (size_t)ld * 4))) return rc;
        if ((rc = alloc((void **)&v.eq2bv, (size_t)round_up(mcap, 16) * 4))) return rc;
        if ((rc = alloc((void **)&v.rowcnt, (size_t)ld * 4))) return rc;
        if ((rc = alloc((void **)&v.colcnt, (size_t)ld * 4))) return rc;
        if ((rc = alloc((void **)&v.ppt, (size_t)nmax * v.pw * 4))) return rc;
        if ((rc = alloc((void **)&v.st, sizeof(LoopState)))) return rc;
        if ((rc = alloc((void **)&v.pickrec, (size_t)PICK_WORDS * 8))) return rc;
        if ((rc = alloc((void **)&v.blkK, (size_t)round_up(mcap, 16) * BLK_MAX * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.blkE, (size_t)BLK_MAX * ld * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&v.blkR, (size_t)BLK_REC_TOTAL_WORDS * 8))) return rc;
        if ((rc = alloc((void **)&v.blkP, (size_t)BLK_PART_TOTAL_INTS * 4))) return rc;
        if ((rc = alloc((void **)&v.trace, (size_t)v.trace_cap * 8))) return rc;
        if ((rc = alloc((void **)&d_leq, (size_t)m * cols * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&d_tgtf, (size_t)cols * sizeof(S)))) return rc;
        if ((rc = alloc((void **)&d_maxv, sizeof(S)))) return rc;
        if ((rc = alloc_commit())) return rc;
        hipStream_t s = ctx->stream;
        // on_dev: 0 both host arrays, 1 both device arrays, 2 the system a device array and the objective a host one
        const hipMemcpyKind kd = on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        XPG_HIP(ctx, hipMemcpyAsync(d_leq, leq, (size_t)m * cols * sizeof(S), kd, s));
        XPG_HIP(ctx, hipMemcpyAsync(d_tgtf, tgtf, (size_t)cols * sizeof(S), on_dev == 1 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        // vc(i,i) / vc(i,rhs): default -1 / 0; slack and xa entries are -1 / 0 (lpsol.h:1428, :867-868)
        std::vector<S> hd(ld, minus_one<S>()), hr(ld, zero<S>());
        if (vcd) for (int i = 0; i < n0; i++) hd[i] = ((const S *)vcd)[i];
        if (vcr) for (int i = 0; i < n0; i++) hr[i] = ((const S *)vcr)[i];
        XPG_HIP(ctx, hipMemcpyAsync(v.vcd, hd.data(), (size_t)ld * sizeof(S), hipMemcpyHostToDevice, s));
        XPG_HIP(ctx, hipMemcpyAsync(v.vcr, hr.data(), (size_t)ld * sizeof(S), hipMemcpyHostToDevice, s));
        XPG_HIP(ctx, hipMemsetAsync(v.st, 0, sizeof(LoopState), s));
        XPG_HIP(ctx, hipMemsetAsync(v.pickrec, 0, (size_t)PICK_WORDS * 8, s));
        // the cells of a row beyond W are padding: the sweeps update them along with the last live column (whole
        // 16-byte pairs) and NOBODY READS THEM -- their contents are undefined (E starts as zero there, but once phase
        // one's k_delete_col has shrunk W the column that becomes padding keeps what it held, in the tableau and in E)
        XPG_HIP(ctx, hipMemsetAsync(v.tab, 0, tab_elems * sizeof(S), s));
        XPG_HIP(ctx, hipMemsetAsync(v.blkE, 0, (size_t)BLK_MAX * ld * sizeof(S), s));
        // launch-throttle events, exercised once so their first use is not inside a solve
        for (int i = 0; i < 2; i++) {
            XPG_HIP(ctx, hipEventCreateWithFlags(&throttle[i], hipEventDisableTiming));
            XPG_HIP(ctx, hipEventRecord(throttle[i], s));
            XPG_HIP(ctx, hipEventSynchronize(throttle[i]));
        }
        XPG_HIP(ctx, hipStreamSynchronize(s));      // hd/hr are stack-owned
        return 0;
    }

    // the second tableau copy and the staging buffers of the fused Rational loop, on first use; false: not to be had
    // (the two-launch pipelined loop needs neither)
    bool fused_buffers()
    {
        if (v.tab2) return true;
        if (fused_unavailable) return false;
        void * t2 = nullptr; void * sg = nullptr;
        if (alloc(&t2, tab_elems * sizeof(S)) || alloc(&sg, (size_t)4 * ld_cap * sizeof(S))) { fused_unavailable = true; ctx->err.clear(); return false; }
        (void)hipMemsetAsync(t2, 0, tab_elems * sizeof(S), ctx->stream);
        (void)hipMemsetAsync(sg, 0, (size_t)4 * ld_cap * sizeof(S), ctx->stream);
        v.tab2 = (S *)t2; v.stage = (S *)sg;
        return true;
    }

    int read_state(LoopState * out)
    {
        // (through a pinned mirror: a copy into pageable memory is staged by the runtime, ~10 us more per status read)
        if (!h_state && hipHostMalloc((void **)&h_state, sizeof(LoopState)) != hipSuccess) { h_state = nullptr; (void)hipGetLastError(); }
        XPG_HIP(ctx, hipMemcpyAsync(h_state ? h_state : out, v.st, sizeof(LoopState), hipMemcpyDeviceToHost, ctx->stream));
        XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h_state) *out = *h_state;
        // fused Rational loop: side 1 of the ping-pong tableau is the current one -- copied onto side 0, which everything
        // outside that loop reads (stream-ordered before whatever the caller does next; both copies are then current)
        if (v.tab2 && out->r32_side) launch_side_home<S>(ctx, v);
        if (out->r32_idle - fused_idle_seen >= 8u) fused_idles_often = true;
        fused_idle_seen = out->r32_idle;
        closes_often = out->blk.sweeps_part >= 8 && out->blk.sweeps_part * 20u >= out->blk.sweeps_full + out->blk.sweeps_part;
        if (out->blk.ch_misplaced != chain_misplaced_seen) {        // (an abort of its own kind: it only changes the chain's form)
            chain_aborts_seen += out->blk.ch_misplaced - chain_misplaced_seen;
            chain_misplaced_seen = out->blk.ch_misplaced; chain_spread = true;
        }
        if (out->blk.ch_aborts != chain_aborts_seen) { chain_aborts_seen = out->blk.ch_aborts; chain_off = true; }
        if (out->status == XPG_ERR_CHAIN_STUCK)
            ctx->err = std::is_same<S, R32>::value
                ? "fused Rational loop: a launch met a pivot nobody had staged (nothing of that launch was written), or its stagers never saw the pick's records; rebuild the LP"
                : "blocked loop: a worker of the persistent chain launch stopped answering after the roll call (preempted queue?); rebuild the LP";
        return 0;
    }

    void build(int with_xa)
    {
        v.W = n0 + (with_xa ? 1 : 0) + m + 1;
        v.rhs = v.W - 1;
        // (a tableau that may grow -- warm-started branch and bound appends rows and slack columns -- keeps the allocation's)
        v.ld = row_cap == m ? pick_ld(v.W) : ld_cap;
        irregular_known = false;
        hipLaunchKernelGGL((k_build<S>), dim3(2048), dim3(256), 0, ctx->stream, v, d_leq, d_tgtf, n0, with_xa);
        hipLaunchKernelGGL((k_init_basis<S>), dim3(64), dim3(256), 0, ctx->stream, v, n0 + (with_xa ? 1 : 0));
    }
    void reset_loop(unsigned max_iter)
    {
        hipLaunchKernelGGL((k_reset_loop<S>), dim3(1024), dim3(256), 0, ctx->stream, v, max_iter, opt_pricing,
                           opt_feas_tol);
        pipe_t = 0; pipe_primed = false; fused_idles_often = false; fused_idle_seen = 0; blk_batch = 0; closes_often = false; chain_off = false; chain_aborts_seen = 0; chain_spread = false; chain_misplaced_seen = 0;
    }
    void queue_pivot(int guarded, int counted)
    {
        const int span = v.W > v.m ? v.W : v.m;
        hipLaunchKernelGGL((k_prep<S>), dim3((span + 255) / 256), dim3(256), 0, ctx->stream, v,
                           guarded, counted, 1, 0);
        launch_update<S>(ctx, v, guarded);
    }
    // Queues k loop iterations. The host runs far ahead of the GPU (3 launches cost
    // ~10 us, one iteration ~90 us), and an over-full HIP queue was measured to stall
    // the stream for tens of ms, so at most 2 x 64 iterations are kept in flight:
    // every 64 iterations an event is recorded and the one from two blocks back awaited.
    // Where the automatic choice takes the blocked loop: by the bytes one sweep moves. With the chain launch (one persistent
    // launch stages a batch's pivots) it wins from 48 x 64 LPs up -- whole solves of random, hard and dense LPs of 48 x 64 ...
    // 600 x 500 take 0.83 ... 0.50 of the pipelined loop's time, no single LP more than 0.92; 24 x 40: 1.06
    // (tools/lab/probe_loop_choice.py) -- so from 64 KB; where the chain cannot run (XPG_CHAIN=0, the opt-in Dantzig pricing, a
    // solve whose roll calls failed) the launch-per-stage form only pays where the sweep is what costs: 16 MB, rounds 2-4's
    // line. XPG_BLOCK_FROM_KB overrides both for A/B runs.
    size_t blocked_from_bytes() const
    {
        static const long env_kb = [] { const char * s = xpg_env("XPG_BLOCK_FROM_KB"); return s ? atol(s) : -1L; }();
        if (env_kb >= 0) return (size_t)env_kb << 10;
        const bool chain_usable = ctx->chain && !chain_off && opt_pricing == 0 && chain_shape_ok(ctx, v.m, v.W);   // (the same test launch_blk_batch applies)
        return chain_usable ? ((size_t)64 << 10) : ((size_t)16 << 20);
    }
    // xpg_lp_loop_info: what queue_iterations / launch_blk_batch would run for the LP as it stands
    void loop_info(int32_t * o)
    {
        const bool blocked = std::is_same<S, F64>::value && !irregular && blk_shape_ok(v.W) &&
                             (ctx->loop_mode == 3 || (ctx->loop_auto && (size_t)v.m * v.W * 16 >= blocked_from_bytes()));
        o[0] = blocked ? 3 : ctx->loop_mode == 1 ? 1 : 0;
        o[5] = v.ld;
        if (!blocked) return;
        const BlkPlan pl = blk_plan(ctx, v.m, v.W, v.ld, ctx->block_len, opt_pricing == 0, chain_off, chain_spread, true);
        o[1] = ctx->block_len; o[2] = !pl.chain ? 0 : pl.local ? 1 : 2; o[3] = pl.line; o[4] = ctx->block_len == 32 ? pl.sweep_rows : 16;
        o[6] = pl.workers; o[7] = pl.cpick; o[8] = pl.cprep; o[9] = (int32_t)pl.lds;
    }
    void queue_iterations(unsigned k)
    {
        unsigned blk = 0;
        // blocked loop: chosen explicitly, or by default from blocked_from_bytes() of sweep traffic up
        bool blocked = std::is_same<S, F64>::value && blk_shape_ok(v.W) &&
                       (ctx->loop_mode == 3 || (ctx->loop_auto && (size_t)v.m * v.W * 16 >= blocked_from_bytes()));
        if (blocked && !irregular_known) {                 // once per build: did k_build meet an inf or a NaN? (LoopState::noncanon)
            LoopState hs;
            if (read_state(&hs) == 0) irregular = hs.noncanon != 0;
            irregular_known = true;
        }
        if (blocked && irregular) blocked = false;          // NaN ratios need the generic pick's scan order: the pipelined loop has it
        blocked_now = blocked;
        if (blocked) { queue_blocked(k); return; }
        // (the rational scalar: XPG_R32_LOOP=pipe forces the two-launch loop, =fused the one-launch loop; =serial, the three-launch
        // loop -- neither a default nor a fallback -- exists in the -DXPG_TEST_HOOKS build only)
        static const bool r32_serial = [] { const char * s = xpg_hook("XPG_R32_LOOP"); return s && !strcmp(s, "serial"); }();
        static const bool r32_pipe = [] { const char * s = xpg_env("XPG_R32_LOOP"); return s && !strcmp(s, "pipe"); }();
        static const unsigned generic_every = [] { const char * s = xpg_hook("XPG_R32_GENERIC_EVERY"); const int n = s ? atoi(s) : 0; return (unsigned)(n > 0 ? n : 16); }();
        const bool pipelined = ctx->loop_mode != 1 && (std::is_same<S, F64>::value || !r32_serial);
        // The fused loop where it pays: its sweep copies the columns the in-place sweep skips (ping-pong tableau) and its
        // launch lasts as long as the pick -> staging chain inside it. Measured against the two-launch loop
        // (tools/lab/probe_rat_sizes.py, us per pivot fused / two-launch): 256 x 512 21.4 / 19.9, 384 x 785 21.0 / 21.0,
        // 512 x 1213 17.0 / 18.0, 768 x 1769 16.7 / 20.9, 1024 x 2048 19.7 / 24.7, 1280 x 2281 27.0 / 30.3, 1536 x 2637
        // 32.0 / 31.8, 2048 x 3549 51.5 / 46.4. XPG_R32_LOOP=fused / pipe force one or the other.
        static const bool r32_fused = [] { const char * s = xpg_env("XPG_R32_LOOP"); return s && !strcmp(s, "fused"); }();
        const size_t cells = (size_t)v.m * (size_t)v.W;
        const bool fused_pays = r32_fused || (cells >= 350000u && cells <= 5000000u);
        if (pipelined && !std::is_same<S, F64>::value && !r32_pipe && fused_pays && k > 0 && fused_buffers()) {
            // one launch per pivot (lp_fused_r32.hip.h); the first launch of a call and every generic_every-th one are
            // preceded by a generic point
            for (unsigned t = 0; t < k; t++) {
                const int slot = (int)(pipe_t++ & 1u);
                launch_fused<S>(ctx, v, slot, colstride, fused_idles_often || t % generic_every == 0);
                if ((t & 63) == 63) {
                    hipEvent_t e = throttle[blk & 1];
                    if (blk >= 2) (void)hipEventSynchronize(e);
                    (void)hipEventRecord(e, ctx->stream);
                    blk++;
                }
            }
            return;                                     // (read_state, which follows every call, brings the tableau home)
        }
        if (pipelined && k > 0 && !pipe_primed) {
            // the first pivot is chosen by a launch that has nothing to sweep (pd[1].row < 0): fp64 the sweep launch's
            // pick workgroup, Rational the prep launch's first workgroup
            if (std::is_same<S, F64>::value) launch_pipe_sweep(ctx, v, 1, colstride, false);
            else hipLaunchKernelGGL((k_pipe_prep<S>), dim3(1), dim3(256), 0, ctx->stream, v, 1, colstride);
            pipe_primed = true;
        }
        for (unsigned t = 0; t < k; t++) {
            if (pipelined) {
                // two launches per pivot; the next pivot is chosen inside the sweep launch
                const int slot = (int)(pipe_t++ & 1u);
                hipLaunchKernelGGL((k_pipe_prep<S>), dim3((v.W + 255) / 256), dim3(256), 0, ctx->stream, v, slot, colstride);
                launch_pipe_sweep(ctx, v, slot, colstride, true);
            } else {
                hipLaunchKernelGGL((k_pick<S>), dim3(1), dim3(1024), 0, ctx->stream, v);
                hipLaunchKernelGGL((k_prep<S>), dim3((v.W + 255) / 256), dim3(256), 0, ctx->stream, v, 1, 1, 0, 1);
                launch_update<S>(ctx, v, 1);
            }
            if ((t & 63) == 63) {
                hipEvent_t e = throttle[blk & 1];
                if (blk >= 2) (void)hipEventSynchronize(e);
                (void)hipEventRecord(e, ctx->stream);
                blk++;
            }
        }
    }

    // Blocked loop (lp_blocked.hip.h): at most k loop iterations as ceil(k / B) batches of
    // B x (pick, prep) + one sweep; the device-side budget stops the last batch where k ends.
    void queue_blocked(unsigned k)
    {
        if (k == 0) return;
        const int B = ctx->block_len;
        hipLaunchKernelGGL(k_blk_budget, dim3(1), dim3(64), 0, ctx->stream, v.st, k);
        const unsigned nb = (k + (unsigned)B - 1) / (unsigned)B;
        for (unsigned b = 0; b < nb; b++) {
            const int batch = blk_batch++;
            // the last batch of a budget that is not a multiple of B is enqueued at its own length, so its
            // sweep is the kernel specialised for that many stages (not the full-batch kernel's slow tail)
            const unsigned left = k - b * (unsigned)B;
            // a batch behind a full-length batch of this call may have its stage 0 done by its chain launch (launch_blk_batch)
            const bool can_fold = ctx->chain_fold && ctx->chain && !chain_off && !closes_often && opt_pricing == 0;
            const bool fold = can_fold && b > 0;
            const bool next_folds = can_fold && b + 1 < nb && left >= (unsigned)B;
            launch_blk_batch(ctx, v, batch, left < (unsigned)B ? (int)left : B, opt_pricing == 0, closes_often, chain_off, chain_spread, fold, next_folds);
            if ((b & 7) == 7) {                         // throttle: at most 2 x 8 batches in flight
                hipEvent_t e = throttle[(b >> 3) & 1];
                if ((b >> 3) >= 2) (void)hipEventSynchronize(e);
                (void)hipEventRecord(e, ctx->stream);
            }
        }
    }

    // Runs the queued loop until the device reports anything but ST_RUNNING,
    // then performs the optimum check. Returns a SIX_* status.
    int run_loop()
    {
        LoopState hs;
        unsigned chunk = 32;
        for (;;) {
            queue_iterations(chunk);
            int rc = read_state(&hs);
            if (rc) return rc;
            if (hs.status != ST_RUNNING) break;
            if (chunk < 1024) chunk *= 2;                // (queued launches behind a final status are no-ops)
        }
        return finish(hs.status);
    }
    int finish(int status)
    {
        if (status == ST_CHECK_OPT) {
            hipLaunchKernelGGL((k_solution<S>), dim3(64), dim3(256), 0, ctx->stream, v);
            hipLaunchKernelGGL((k_rowcheck<S>), dim3((v.m + 63) / 64), dim3(64), 0, ctx->stream, v);
            hipLaunchKernelGGL((k_finish<S>), dim3(1), dim3(1), 0, ctx->stream, v, d_maxv);
            LoopState hs;
            int rc = read_state(&hs);
            if (rc) return rc;
            status = hs.status;
        }
        return status;
    }

    // SIX::stage1's slack branch only (lpsol.h:1820-1841) and a fresh loop.
    int begin()
    {
        build(0);
        reset_loop(0xFFFFFFFFu);
        began = true; final_status = XPG_RUNNING;
        XPG_HIP(ctx, hipGetLastError());
        return 0;
    }
    // At most `pivots` further loop iterations, one host sync at the end.
    int iterate(unsigned pivots)
    {
        if (!began) return XPG_ERR_SHAPE;
        if (final_status != XPG_RUNNING) return final_status;
        queue_iterations(pivots);
        LoopState hs;
        int rc = read_state(&hs);
        if (rc) return rc;
        // blocked loop: a batch whose stage 0 was left to its chain launch does nothing when the batch before did not admit it
        // (it closed early: a rare branch, an aborted roll call) -- and neither does any batch behind it. What is left of the
        // budget is then enqueued again, starting with stage-0 launches, which always make progress or end the loop.
        for (int again = 0; blocked_now && hs.status == ST_RUNNING && hs.blk.budget != 0u && hs.blk.budget <= pivots && again < 1 << 20; again++) {
            queue_iterations(hs.blk.budget);
            rc = read_state(&hs);
            if (rc) return rc;
        }
        if (hs.status == ST_RUNNING) return XPG_RUNNING;
        final_status = finish(hs.status);
        return final_status;
    }

    // SIX::TwoStageMethod (lpsol.h:1907-1930).
    int two_stage(unsigned max_iter)
    {
        hipLaunchKernelGGL((k_need_phase1<S>), dim3(1), dim3(1024), 0, ctx->stream, d_leq, d_tgtf, m, n0, v.st);
        LoopState hs;
        int rc = read_state(&hs);
        if (rc) return rc;
        began = true;
        if (hs.aux) {
            rc = phase_one(max_iter);
            if (rc != 0) { final_status = rc; return rc; }
        } else {
            build(0);
        }
        reset_loop(max_iter);
        final_status = run_loop();
        return final_status;
    }

    // SIX::constructBasicFeasibleSolution (lpsol.h:839-988). Returns 0 when a
    // feasible slack form is in place, else SIX_NO_PRI_FEASIBLE_SOL (or an error).
    int phase_one(unsigned max_iter)
    {
        const int xa = n0;
        build(1);
        hipLaunchKernelGGL((k_force_pivot<S>), dim3(1), dim3(1024), 0, ctx->stream, v, xa);
        queue_pivot(0, 0);
        reset_loop(max_iter);
        int st = run_loop();
        if (st < 0) return st;
        if (st != XPG_SIX_SUCC) return XPG_SIX_NO_PRI_FEASIBLE_SOL;
        hipLaunchKernelGGL((k_phase1_exit<S>), dim3(1), dim3(64), 0, ctx->stream, v, xa, d_maxv);
        LoopState hs;
        int rc = read_state(&hs);
        if (rc) return rc;
        if (hs.aux == -1) return XPG_SIX_NO_PRI_FEASIBLE_SOL;
        if (hs.aux == -7) return XPG_ERR_REF_UNDEFINED;
        if (hs.aux == 1) queue_pivot(0, 0);
        hipLaunchKernelGGL((k_rebuild_obj<S>), dim3(1), dim3(1024), 0, ctx->stream, v, d_tgtf, n0);
        hipLaunchKernelGGL((k_delete_col<S>), dim3(v.m + 1), dim3(256), 0, ctx->stream, v, xa);
        v.W -= 1; v.rhs -= 1;
        XPG_HIP(ctx, hipGetLastError());
        return 0;
    }

    int read(void * tab, void * obj, uint8_t * nvset, uint8_t * bvset, int32_t * bv2eq,
             int32_t * eq2bv, void * maxv, void * sol)
    {
        hipStream_t s = ctx->stream;
        if (tab)
            XPG_HIP(ctx, hipMemcpy2DAsync(tab, (size_t)v.W * sizeof(S), v.tab, (size_t)v.ld * sizeof(S),
                                          (size_t)v.W * sizeof(S), v.m, hipMemcpyDeviceToHost, s));
        if (obj) XPG_HIP(ctx, hipMemcpyAsync(obj, v.obj, (size_t)v.W * sizeof(S), hipMemcpyDeviceToHost, s));
        if (nvset) XPG_HIP(ctx, hipMemcpyAsync(nvset, v.nv, v.rhs, hipMemcpyDeviceToHost, s));
        if (bvset) XPG_HIP(ctx, hipMemcpyAsync(bvset, v.bv, v.rhs, hipMemcpyDeviceToHost, s));
        if (bv2eq) XPG_HIP(ctx, hipMemcpyAsync(bv2eq, v.bv2eq, (size_t)v.rhs * 4, hipMemcpyDeviceToHost, s));
        if (eq2bv) XPG_HIP(ctx, hipMemcpyAsync(eq2bv, v.eq2bv, (size_t)v.m * 4, hipMemcpyDeviceToHost, s));
        if (maxv) {
            if (final_status == XPG_SIX_SUCC)
                XPG_HIP(ctx, hipMemcpyAsync(maxv, d_maxv, sizeof(S), hipMemcpyDeviceToHost, s));
            else *(S *)maxv = zero<S>();
        }
        if (sol) XPG_HIP(ctx, hipMemcpyAsync(sol, v.x, (size_t)v.W * sizeof(S), hipMemcpyDeviceToHost, s));
        XPG_HIP(ctx, hipStreamSynchronize(s));
        return 0;
    }
};

} // namespace xpg
