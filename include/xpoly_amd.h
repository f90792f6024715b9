/* xpoly_amd -- MI355X (gfx950) simplex / row-elimination kernels behind xpoly's
 * SIX<Mat,T>, MIP<Mat,T> and Lineq interfaces.  C ABI, no C++ or torch types.
 *
 * The reference (stevenknown/xpoly) has no FFI: its boundary is the C++ template
 * contract of SIX<Mat,T> (src/com/lpsol.h:204-338) over Matrix<T>'s public
 * row-major buffer (src/com/matt.h:152-156).  Each entry point below names the
 * reference member it stands in for; INTEGRATION.md shows the adapter a
 * maintainer would add on the xpoly side.
 *
 * Layouts (identical to the reference's in-memory layout):
 *   f64   : double[rows*cols], row-major                  (FloatMat, xmat.h:140)
 *   rat32 : struct {int32 num; int32 den;}[rows*cols]      (RMat, xmat.h:42; rational.h:66-67)
 *   every matrix of one problem has `cols` columns; the last one is the constant
 *   column (rhs_idx = cols-1, lpsol.h:1527-1552); `vc` is (cols-1) x cols with -1
 *   on the diagonal for x_i >= 0 and an all-zero column for a free variable
 *   (lpsol.h:1322); `eq`/`leq` may have 0 rows (pass NULL).
 *
 * Return values: the reference's own status integers (lpsol.h:198-202,
 * :2082-2085) or a negative XPG_ERR_* code.  No exceptions cross this boundary.
 * All buffers are caller-owned; the library never frees caller memory.
 * Functions with a _dev suffix take DEVICE pointers; the others take HOST
 * pointers and stage through HBM themselves.
 *
 * Threading: a handle (xpg_ctx, and every xpg_lp made from it) is used by ONE host
 * thread at a time -- it owns one HIP stream, grow-only staging areas and a small
 * cache of device blocks (xpg_trim returns them), none of which is locked.  Different
 * handles, on the same device or on different ones, may be used from different
 * threads concurrently (the _multi and _ragged entry points do exactly that inside
 * the library).  The reference itself is single-threaded with per-instance state
 * (lpsol.h:205-209).
 */
#ifndef XPOLY_AMD_H
#define XPOLY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SIX status codes -- src/com/lpsol.h:198-202 */
#define XPG_SIX_SUCC                  0
#define XPG_SIX_UNBOUND               1
#define XPG_SIX_NO_PRI_FEASIBLE_SOL   2
#define XPG_SIX_OPTIMAL_IS_INFEASIBLE 3
#define XPG_SIX_TIME_OUT              4
/* MIP status codes -- src/com/lpsol.h:2082-2085 */
#define XPG_IP_SUCC                    0
#define XPG_IP_UNBOUND                 1
#define XPG_IP_NO_PRI_FEASIBLE_SOL     2
#define XPG_IP_NO_BETTER_THAN_BEST_SOL 3
/* library errors (never overlap 0..4) */
#define XPG_ERR_HIP          (-1)  /* a HIP runtime call failed; see xpg_last_error */
#define XPG_ERR_ALLOC        (-2)
#define XPG_ERR_SHAPE        (-3)  /* malformed sizes (reference: ASSERT only, lpsol.h:1517-1558) */
#define XPG_ERR_UNSUPPORTED  (-4)
#define XPG_ERR_NO_DEVICE    (-5)
#define XPG_ERR_REF_UNDEFINED (-7) /* the reference's behaviour is undefined on this input */
#define XPG_ERR_CHAIN_STUCK  (-8)  /* a launch whose workgroups wait for each other stopped making progress (a preempted
                                      or hung queue): the persistent chain launch of the blocked fp64 loop AFTER its roll
                                      call, the fused Rational loop's pick -> staging hand-over (also: a pivot nobody
                                      staged), or a time-sliced batch launch whose queue did not move for 20 s.  A
                                      device LP must be rebuilt (xpg_lp_begin / _two_stage); a batch call returns it as
                                      the CALL's result and none of its outputs is valid (the _dev forms, which only
                                      enqueue, leave it as the status of every LP the launch did not finish) */

typedef struct xpg_ctx xpg_ctx;   /* one device + one HIP stream + scratch; re-entrant per handle */
typedef struct xpg_lp  xpg_lp;    /* one device-resident slack-form LP (tableau, objective, basis) */

typedef struct { int32_t num, den; } xpg_rat32;   /* src/com/rational.h:66-67 */

/* ---- context ---------------------------------------------------------------- */
int         xpg_device_count(void);
int         xpg_create(xpg_ctx ** out, int device);
void        xpg_destroy(xpg_ctx * ctx);
const char *xpg_last_error(const xpg_ctx * ctx);
const char *xpg_version(void);
void *      xpg_stream(const xpg_ctx * ctx);            /* the hipStream_t all work is queued on */
int         xpg_sync(xpg_ctx * ctx);
/* device buffers for hosts without their own allocator (benchmarks, C callers) */
int         xpg_malloc(xpg_ctx * ctx, void ** dptr, size_t bytes);
int         xpg_free(xpg_ctx * ctx, void * dptr);
int         xpg_upload(xpg_ctx * ctx, void * dst_dev, const void * src_host, size_t bytes);
int         xpg_download(xpg_ctx * ctx, void * dst_host, const void * src_dev, size_t bytes);

/* ---- in-library timing of the HBM-bound sweep (K1's update kernel) -------------------
 * Between begin and end every `stride`-th launch of the sweep on this context is
 * bracketed by a pair of HIP events recorded on the context's stream (at most `cap`
 * launches are sampled; an event pair costs ~6 us of stream time, hence the stride).
 * end synchronises the stream and returns the number of sampled launches and the
 * sum of their durations in milliseconds. */
int         xpg_profile_begin(xpg_ctx * ctx, int cap, int stride);
int         xpg_profile_end(xpg_ctx * ctx, int * launches, double * total_ms);

/* ---- K1: one pivot -- SIX::pivot arithmetic, src/com/lpsol.h:1471-1501 ------------
 * tab is m x W with leading dimension ld (elements); obj has W entries.  Row `row`
 * is scaled by 1/tab[row][col], column `col` is eliminated from every other row and
 * the scaled row is folded into obj.  Asynchronous on the context's stream. */
int xpg_pivot_f64_dev(xpg_ctx * ctx, double * tab_dev, int m, int W, int ld,
                      double * obj_dev, int rhs_idx, int row, int col);
int xpg_pivot_rat32_dev(xpg_ctx * ctx, xpg_rat32 * tab_dev, int m, int W, int ld,
                        xpg_rat32 * obj_dev, int rhs_idx, int row, int col);
int xpg_pivot_f64(xpg_ctx * ctx, double * tab, int m, int W, double * obj,
                  int rhs_idx, int row, int col);
int xpg_pivot_rat32(xpg_ctx * ctx, xpg_rat32 * tab, int m, int W, xpg_rat32 * obj,
                    int rhs_idx, int row, int col);

/* ---- device-resident LP: SIX::TwoStageMethod, src/com/lpsol.h:1907-1930 ------------
 * `leq` is m x cols (A | b), x >= 0 for every variable, `tgtf` has cols entries.
 * vc_diag/vc_rhs hold vc(i,i) and vc(i,rhs) (the only cells SIX::is_feasible
 * reads, lpsol.h:798-802); NULL means -1 / 0.  kind: 0 = f64, 1 = rat32.
 * src_on_device: 0 leq / tgtf are host arrays, 1 both are device pointers, 2 leq is a device
 * pointer and tgtf a host array (what SIX::minm's device-built dual hands over). */
int  xpg_lp_create(xpg_ctx * ctx, int kind, const void * leq, int m, int cols,
                   const void * tgtf, const void * vc_diag, const void * vc_rhs,
                   int src_on_device, xpg_lp ** out);
void xpg_lp_destroy(xpg_lp * lp);
/* stage1 + solveSlackForm with SIX::set_param(.., max_iter) (lpsol.h:380-385). */
int  xpg_lp_two_stage(xpg_lp * lp, unsigned max_iter);
/* Only SIX::stage1's slack construction (lpsol.h:1820-1841); then xpg_lp_iterate
 * runs at most `pivots` more iterations of solveSlackForm's loop (lpsol.h:1039-1188)
 * without a host round trip per pivot.  Returns -1000 while still running. */
int  xpg_lp_begin(xpg_lp * lp);
int  xpg_lp_iterate(xpg_lp * lp, unsigned pivots);
#define XPG_RUNNING (-1000)
/* Pivots the handle has made over its LIFETIME (every solve on it, phase 1's included): difference two reads for one solve. */
int  xpg_lp_pivots_done(xpg_lp * lp, unsigned * out);
/* Blocked loop bookkeeping since xpg_lp_begin / xpg_lp_two_stage: sweeps that applied a full batch of
 * staged pivots (32 by default, XPG_BLOCK = 1 .. 32: the batch's stages, the first included, are chosen and
 * staged by one persistent chain launch; lp_chain.hip.h), and sweeps that applied fewer (the tail of an iterate budget, or a batch closed
 * early by a rare branch of SIX::solveSlackForm, src/com/lpsol.h:1138-1151).  Either may be NULL. */
int  xpg_lp_counters(xpg_lp * lp, unsigned * sweeps_full, unsigned * sweeps_partial);
/* Blocked fp64 loop, diagnostics: persistent chain launches of the current solve that were given up at their roll call
 * because not every worker got a compute unit in time (the device is shared with other work), and whether the handle
 * has switched this solve to launch-per-stage kernels because of it; runs (may be NULL) = the launches that passed.
 * Results are the same either way. */
int  xpg_lp_chain_aborts(xpg_lp * lp, unsigned * aborts, int * chain_off, unsigned * runs);
/* ... and how many of the launches that passed did their batch's first stage themselves (no pick / prep launches for it). */
int  xpg_lp_chain_folds(xpg_lp * lp, unsigned * folds);
/* Which loop and which kernel instances the handle runs the LP in its current shape with (evidence for bench.py's `shapes`
 * leg; no reference counterpart).  out[0..n-1], n <= 10: loop (0 pipelined, 1 serial, 3 blocked), pivots per blocked pass,
 * chain form (0 launch-per-stage kernels, 1 one persistent launch on one XCD, 2 the same spread over the chip), columns of
 * the entering column's line kept in LDS (0 / 8 / 16), rows per workgroup of the pass (16 / 32), leading dimension, chain
 * workers, pick workers, prep workers, LDS bytes per chain worker. */
int  xpg_lp_loop_info(xpg_lp * lp, int32_t * out, int n);
/* Test view: SIX::normalize (src/com/lpsol.h:1290-1394, convertEq2Ineq :1197-1278) made both ways for one input -- the cells
 * the HBM route makes on the device (out_dev_cells) and the cells the LDS route makes on the host (out_host_cells), each
 * [rows][n + 1] (kind 0: double, 1: xpg_rat32; cap_cells = room of each buffer in cells).  out_info[0..6] = rows, n,
 * substitutions made, equalities kept as pairs, free variables, the device form's return code, the host form's
 * (XPG_ERR_REF_UNDEFINED where the reference reads past an equality's row, lpsol.h:1232). */
int  xpg_test_normalize(xpg_ctx * ctx, int kind, const void * tgtf, const void * vc, int vc_rows, const void * eq, int eq_rows,
                        const void * leq, int leq_rows, int cols, void * out_dev_cells, void * out_host_cells, long long cap_cells,
                        int32_t * out_info);
/* Launch geometry, host-side views for tests (no device needed).  xpg_test_sweep_tile: the blocked sweep's workgroup ->
 * tile map for a tableau of `strips` 512-column strips and `rowblocks` row blocks: lid < 0 returns the grid size, else
 * 1 / 0 = workgroup lid has / has no tile, written to (*bx, *by).  xpg_test_pick_ld: the leading dimension a device
 * tableau of W live columns gets. */
int  xpg_test_sweep_tile(int strips, int rowblocks, int rev, int lid, int * bx, int * by);
int  xpg_test_pick_ld(int W);
/* The device's canonical rational forms on n host triples (tests: the sweep's fused a + k * e and the ratio test's
 * b / a must equal the reference's two operations, src/com/rational.cpp:273-397, for canonical operands -- lowest
 * terms, den > 0, below the appro threshold; anything else in the inputs returns XPG_ERR_SHAPE):
 * out_fma[i] = a[i] + k[i] * e[i], out_div[i] = a[i] / k[i] (k[i] = 0: 0/1). Either output may be NULL. */
int  xpg_test_canon_ops_rat32(xpg_ctx * ctx, int n, const xpg_rat32 * a, const xpg_rat32 * k, const xpg_rat32 * e,
                              xpg_rat32 * out_fma, xpg_rat32 * out_div);
/* The device's GENERIC rational forms (any numerators, any denominators above INT32_MIN, zero and negative ones
 * included -- what a problem whose cells are not canonical runs on): out_mul[i] = a[i] * b[i], out_add[i] = a[i] + b[i],
 * out_div[i] = a[i] / b[i] as src/com/rational.cpp:273-397 computes them. Outputs may be NULL. */
int  xpg_test_any_ops_rat32(xpg_ctx * ctx, int n, const xpg_rat32 * a, const xpg_rat32 * b,
                            xpg_rat32 * out_mul, xpg_rat32 * out_add, xpg_rat32 * out_div);
/* OPT-IN, NON-PARITY (results are no longer the reference's bit for bit, and
 * nothing else in this header changes behaviour): before xpg_lp_begin / xpg_lp_two_stage,
 *   pricing = 1        Dantzig's rule -- the largest reduced cost enters -- instead of the reference's
 *                      first positive one (src/com/lpsol.h:1054-1069); the ratio test and the
 *                      anti-cycling pair table stay as they are;
 *   feas_rel_tol > 0   SIX::is_feasible (src/com/lpsol.h:784-822) with this relative tolerance
 *                      instead of Float's 1e-17 '==' (src/com/flty.cpp:41-58), which reports most
 *                      fp64 optima as SIX_OPTIMAL_IS_INFEASIBLE.
 * (0, 0.0) restores the reference's behaviour.  fp64 handles only (XPG_ERR_UNSUPPORTED otherwise). */
int  xpg_lp_set_options(xpg_lp * lp, int pricing, double feas_rel_tol);
/* shape of the live tableau: rows, columns W (= rhs_idx + 1), rhs_idx */
int  xpg_lp_shape(xpg_lp * lp, int * rows, int * W, int * rhs_idx);
/* download the live state; any pointer may be NULL.  tab: rows*W, obj: W,
 * nvset/bvset: rhs_idx bytes, bv2eq: rhs_idx, eq2bv: rows, maxv: 1, sol: W. */
int  xpg_lp_read(xpg_lp * lp, void * tab, void * obj, uint8_t * nvset, uint8_t * bvset,
                 int32_t * bv2eq, int32_t * eq2bv, void * maxv, void * sol);
/* the (entering, leaving) pairs pivoted so far, for parity tests */
int  xpg_lp_trace(xpg_lp * lp, int32_t * pairs, int cap_pairs, int * n_pairs);

/* ---- SIX::maxm / SIX::minm, src/com/lpsol.h:1993-2033 / :1662-1732 -------------------
 * Same argument meaning as the reference: out_v receives the optimum (0 on
 * non-success, lpsol.h:2024), out_sol the cols entries of `res` with a trailing
 * 1 in the constant slot (lpsol.h:1880-1887); out_sol is written only on success. */
int xpg_six_maxm_f64(xpg_ctx * ctx, const double * tgtf, const double * vc, int vc_rows,
                     const double * eq, int eq_rows, const double * leq, int leq_rows,
                     int cols, unsigned max_iter, double * out_v, double * out_sol);
int xpg_six_minm_f64(xpg_ctx * ctx, const double * tgtf, const double * vc, int vc_rows,
                     const double * eq, int eq_rows, const double * leq, int leq_rows,
                     int cols, unsigned max_iter, double * out_v, double * out_sol);
int xpg_six_maxm_rat32(xpg_ctx * ctx, const xpg_rat32 * tgtf, const xpg_rat32 * vc,
                       int vc_rows, const xpg_rat32 * eq, int eq_rows,
                       const xpg_rat32 * leq, int leq_rows, int cols, unsigned max_iter,
                       xpg_rat32 * out_v, xpg_rat32 * out_sol);
int xpg_six_minm_rat32(xpg_ctx * ctx, const xpg_rat32 * tgtf, const xpg_rat32 * vc,
                       int vc_rows, const xpg_rat32 * eq, int eq_rows,
                       const xpg_rat32 * leq, int leq_rows, int cols, unsigned max_iter,
                       xpg_rat32 * out_v, xpg_rat32 * out_sol);

/* Where the calling thread's last xpg_six_{maxm,minm}_* call spent its time, host clock, milliseconds:
 * out_ms[0..7] = total, host reshaping (SIX::normalize, lpsol.h:1290-1394: nothing but the vc diagonal when there are
 * no equalities and no free variables), handle creation incl. the upload of the system, the dual built on the device
 * (minm: upload + transpose; lpsol.h:1602-1629), stage 1 + pivot loop on the device, read-back, release, and the route
 * taken (1 = the LDS-resident batch kernel, 2 = the HBM-resident loop); out_ms[8] = the pivots the HBM-resident route's last
 * pivot loop made (SIX::solveSlackForm's cnt, lpsol.h:1187).  Evidence for bench.py and the tests; no reference counterpart. */
int xpg_six_last_profile(double * out_ms, int n);

/* ---- batches of independent small LPs (the dependence-test workload) ------------------
 * nb problems of identical shape: leq[nb][m][cols], tgtf[nb][cols], x >= 0, no
 * equalities -- what Lineq::has_solution hands to SIX (src/com/linsys.cpp:852-904).
 * Each LP is solved LDS-resident by one workgroup.  is_max selects maxm / minm.
 * out_status[nb], out_v[nb], out_sol[nb][cols] (rows of failed LPs untouched).
 * The _dev variants take device pointers for every array and are asynchronous. */
int xpg_six_batch_f64(xpg_ctx * ctx, int is_max, int nb, const double * tgtf,
                      const double * leq, int m, int cols, unsigned max_iter,
                      int32_t * out_status, double * out_v, double * out_sol);
int xpg_six_batch_rat32(xpg_ctx * ctx, int is_max, int nb, const xpg_rat32 * tgtf,
                        const xpg_rat32 * leq, int m, int cols, unsigned max_iter,
                        int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol);
int xpg_six_batch_f64_dev(xpg_ctx * ctx, int is_max, int nb, const double * tgtf,
                          const double * leq, int m, int cols, unsigned max_iter,
                          int32_t * out_status, double * out_v, double * out_sol,
                          uint32_t * out_pivots);
int xpg_six_batch_rat32_dev(xpg_ctx * ctx, int is_max, int nb, const xpg_rat32 * tgtf,
                            const xpg_rat32 * leq, int m, int cols, unsigned max_iter,
                            int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol,
                            uint32_t * out_pivots);

/* The same batches spread over the GPUs of one node from ONE caller thread -- what a C++ xpoly
 * caller of Lineq::has_solution (src/com/linsys.cpp:860-904) gets when it hands a SCoP's worth of
 * problems over at once.  devices[ndev] lists the HIP devices (NULL: 0 .. ndev-1; a device may be
 * listed twice).  Shard g is the contiguous range [g*nb/ndev + min(g, nb%ndev), ...) (sizes differ by
 * at most one, as xpoly_amd/shard.py); each shard runs on a context and host thread of its own and
 * writes straight into the caller's HOST arrays -- there is no collective and no second copy.
 * Returns 0, or the first shard's XPG_ERR_* (XPG_ERR_NO_DEVICE when a listed device is absent). */
int xpg_six_batch_f64_multi(int ndev, const int * devices, int is_max, int nb, const double * tgtf,
                            const double * leq, int m, int cols, unsigned max_iter,
                            int32_t * out_status, double * out_v, double * out_sol);
int xpg_six_batch_rat32_multi(int ndev, const int * devices, int is_max, int nb, const xpg_rat32 * tgtf,
                              const xpg_rat32 * leq, int m, int cols, unsigned max_iter,
                              int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol);

/* ---- MIP<Mat,T>::maxm / minm, src/com/lpsol.h:2636-2657 / :2681-2702 ---------------------
 * Depth-first branch and bound exactly as MIP::RecusivePart (lpsol.h:2427-2612): every
 * node is a from-scratch SIX solve (max_iter 10000, lpsol.h:2441) on the GPU; is_bin
 * selects 0-1 programming; rational_indicator (cols bytes, may be NULL) marks entries
 * allowed to stay fractional (lpsol.h:2369-2393).  Returns XPG_IP_* or XPG_ERR_*.
 * With vc = -I (x >= 0; what the reference's caller PolyTran::FeaSchedule passes, src/eng/poly.cpp:5118-5130), with
 * or without equalities, and node LPs within 64 KB of LDS the whole tree walk runs on the device in one launch;
 * any other vc is walked by the host controller (node LPs on the device, lock-step rounds): same results.
 * (The reference cannot instantiate MIP<FloatMat,Float>, lpsol.h:2242-2254; the f64
 * flavour follows the same template text.) */
int xpg_mip_maxm_rat32(xpg_ctx * ctx, const xpg_rat32 * tgtf, const xpg_rat32 * vc, int vc_rows,
                       const xpg_rat32 * eq, int eq_rows, const xpg_rat32 * leq, int leq_rows,
                       int cols, int is_bin, const uint8_t * rational_indicator,
                       xpg_rat32 * out_v, xpg_rat32 * out_sol);
int xpg_mip_minm_rat32(xpg_ctx * ctx, const xpg_rat32 * tgtf, const xpg_rat32 * vc, int vc_rows,
                       const xpg_rat32 * eq, int eq_rows, const xpg_rat32 * leq, int leq_rows,
                       int cols, int is_bin, const uint8_t * rational_indicator,
                       xpg_rat32 * out_v, xpg_rat32 * out_sol);
int xpg_mip_maxm_f64(xpg_ctx * ctx, const double * tgtf, const double * vc, int vc_rows,
                     const double * eq, int eq_rows, const double * leq, int leq_rows, int cols,
                     int is_bin, const uint8_t * rational_indicator, double * out_v, double * out_sol);
int xpg_mip_minm_f64(xpg_ctx * ctx, const double * tgtf, const double * vc, int vc_rows,
                     const double * eq, int eq_rows, const double * leq, int leq_rows, int cols,
                     int is_bin, const uint8_t * rational_indicator, double * out_v, double * out_sol);
/* OPT-IN, NON-PARITY: branch and bound that re-optimises every node from its parent's final
 * tableau with the dual simplex instead of the fresh SIX per node of src/com/lpsol.h:2440-2448.  fp64; maximise (or
 * minimise) tgtf . x subject to leq (A | b), x >= 0 and integral (0/1 bounds are rows of leq, as for the parity MIP).
 * The tableau stays in HBM for the whole tree; a child is its parent's solved state plus one bound row.  Best
 * incumbent, bounding by the relaxation, floor child first -- a different (sane) walk from the reference's, so its
 * answers are checked against the optimum itself (tests/test_gpu_warm_mip.py: scipy / HiGHS), not against
 * MIP::RecusivePart.  out_stats (may be NULL): nodes, dual pivots over all nodes, primal pivots of the root, depth. */
int xpg_mip_warm_f64(xpg_ctx * ctx, int is_max, const double * tgtf, const double * leq, int leq_rows,
                     int cols, int is_bin, double * out_v, double * out_sol, long long * out_stats);
/* The same method for a BATCH: nb programs of one shape (tgtf [nb][cols], leq [nb][leq_rows][cols]), each tree walked by ONE
 * WORKGROUP inside one launch with its current tableau in LDS (warm_mip_batch.hip.h) -- the root's two-phase solve, the
 * depth-first stack, snapshots in HBM, bounding -- as the parity walk does it for MIP::RecusivePart (xpg_mip_batch_*).
 * is_bin: the program is 0-1 (its x_j <= 1 rows are rows of leq): a path then appends at most one bound row per variable,
 * which sizes the LDS block; 0: what 64 KB allow (a tree that needs more ends XPG_ERR_UNSUPPORTED in out_status).
 * out_status[b] = XPG_IP_* (or XPG_ERR_UNSUPPORTED), out_v[b], out_sol[b][cols]; out_stats (may be NULL): nodes, dual
 * pivots and root pivots summed over the batch, the deepest path. */
int xpg_mip_warm_batch_f64(xpg_ctx * ctx, int nb, int is_max, const double * tgtf, const double * leq, int leq_rows, int cols,
                           int is_bin, int32_t * out_status, double * out_v, double * out_sol, long long * out_stats);
/* Lineq::has_solution(leq, eq, vc, rhs_idx, is_int_sol, is_unique_sol),
 * src/com/linsys.cpp:830-906.  Returns 1 / 0, or XPG_ERR_*. */
int xpg_has_solution_rat32(xpg_ctx * ctx, const xpg_rat32 * leq, int leq_rows, const xpg_rat32 * eq,
                           int eq_rows, const xpg_rat32 * vc, int vc_rows, int cols, int rhs_idx,
                           int is_int_sol, int is_unique_sol);

/* Batches: nb independent problems of one shape, x >= 0, inequalities only
 * (tgtf[nb][cols], leq[nb][leq_rows][cols]).  Every tree is walked on the device by one
 * workgroup (node rebuild, normalisation, LDS solve, the recursion of lpsol.h:2427-2612 as a
 * stack machine); problems whose node LPs do not fit 64 KB of LDS advance in lock step from the
 * host instead, the node LPs of a round sharing one launch.
 * out_nodes (may be NULL) receives the total number of node LPs solved. */
int xpg_mip_batch_rat32(xpg_ctx * ctx, int nb, int is_max, int is_bin, const xpg_rat32 * tgtf,
                        const xpg_rat32 * leq, int leq_rows, int cols, int32_t * out_status,
                        xpg_rat32 * out_v, xpg_rat32 * out_sol, long long * out_nodes);
int xpg_mip_batch_f64(xpg_ctx * ctx, int nb, int is_max, int is_bin, const double * tgtf,
                      const double * leq, int leq_rows, int cols, int32_t * out_status,
                      double * out_v, double * out_sol, long long * out_nodes);
/* The same for nb MIPs with eq_rows EQUALITIES each at the root, eq[nb][eq_rows][cols] (x >= 0; the shape
 * PolyTran::FeaSchedule hands to MIP::maxm / minm, src/eng/poly.cpp:5118-5130); leq may be NULL with
 * leq_rows = 0.  Every node runs SIX::convertEq2Ineq (src/com/lpsol.h:1197-1278) over the root's and the
 * branches' equalities, as the reference's recursion does. */
int xpg_mip_batch_eq_rat32(xpg_ctx * ctx, int nb, int is_max, int is_bin, const xpg_rat32 * tgtf,
                           const xpg_rat32 * leq, int leq_rows, const xpg_rat32 * eq, int eq_rows, int cols,
                           int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol, long long * out_nodes);
int xpg_mip_batch_eq_f64(xpg_ctx * ctx, int nb, int is_max, int is_bin, const double * tgtf,
                         const double * leq, int leq_rows, const double * eq, int eq_rows, int cols,
                         int32_t * out_status, double * out_v, double * out_sol, long long * out_nodes);
/* DepPoly::is_empty(keepit, vc = NULL), src/eng/poly.cpp:530-573, for nb dependence polyhedra
 * mats[nb][rows][cols] without constant symbols (constant in the last column):
 * Lineq::reduce pre-filter, then Lineq::has_solution(is_int_sol, is_unique_sol) = MIP::maxm
 * and, failing that, MIP::minm.  out_empty[b] = 1 / 0, or XPG_ERR_REF_UNDEFINED. */
int xpg_dep_is_empty_batch_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                                 int32_t * out_empty, long long * out_nodes);

/* The same with the two arguments of the reference spelled out: the constant is column rhs_idx and the columns
 * after it are constant symbols, which Lineq::move2var (src/com/linsys.cpp:1177-1200) first turns into variables
 * (i + j <= 1 + M + N  ->  i + j - M - N <= 1, poly.cpp:536-548); vc [rhs_idx][rhs_idx + 1] are the caller's
 * variable constraints (NULL: -x_i <= 0, poly.cpp:559-567).  With symbols the reference hands has_solution a
 * matrix with several constant columns, which SIX::verify only ASSERTs (lpsol.h:1526-1552): systems that
 * Lineq::reduce does not decide then get XPG_ERR_REF_UNDEFINED in out_empty. */
int xpg_dep_is_empty_batch_ex_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                                    int rhs_idx, const xpg_rat32 * vc, int32_t * out_empty,
                                    long long * out_nodes);
/* The same with a choice of what happens to a parametrised polyhedron (rhs_idx < cols - 1) that Lineq::reduce does not decide.
 *   XPG_DEP_PARITY           what the reference does: undefined (XPG_ERR_REF_UNDEFINED in out_empty).  Built with
 *                            -fsanitize=address the reference heap-overflows there: MIP::verify -> Matrix::is_colequ, from
 *                            src/com/linsys.cpp:864 -- vc is sized for rhs_idx variables, the system has rhs_idx + symbols.
 *   XPG_DEP_SYMBOLS_AS_VARS  OPT-IN, NOT PARITY: the evident intent of src/eng/poly.cpp:530-573 -- after move2var the symbols
 *                            are variables of unknown sign, so has_solution(int, unique) is asked about the widened system:
 *                            rhs_idx = cols - 1, vc widened by all-zero (= free) rows and columns for the symbols.  Checked
 *                            against the CPU restatement's move2var + reduce + has_solution on that widened system. */
enum { XPG_DEP_PARITY = 0, XPG_DEP_SYMBOLS_AS_VARS = 1 };
int xpg_dep_is_empty_batch_mode_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                      const xpg_rat32 * vc, int mode, int32_t * out_empty, long long * out_nodes);
/* Multi-device forms of the two batches above (sharding and devices[] as xpg_six_batch_*_multi;
 * out_nodes receives the sum over the shards). */
int xpg_mip_batch_rat32_multi(int ndev, const int * devices, int nb, int is_max, int is_bin,
                              const xpg_rat32 * tgtf, const xpg_rat32 * leq, int leq_rows, int cols,
                              int32_t * out_status, xpg_rat32 * out_v, xpg_rat32 * out_sol,
                              long long * out_nodes);
int xpg_dep_is_empty_batch_rat32_multi(int ndev, const int * devices, int nb, const xpg_rat32 * mats,
                                       int rows, int cols, int32_t * out_empty, long long * out_nodes);

/* ---- ragged batches: problems of different shapes in one call -------------------------------------------------
 * What one SCoP hands the dependence analysis: DepPolyMgr::build emits polyhedra whose shape follows the
 * statements' depths and the parameter count (src/eng/poly.cpp:1120-1224, :1009-1053) and DepGraph::rebuild tests
 * each (poly.cpp:268-314, :530-573).  Padding to one shape is not parity-neutral, so shapes are per problem:
 * rows[b], cols[b], and the CELL offset of problem b in the concatenated array (offsets[b]; tgtf / sol / v arrays of
 * the LP form have their own offsets, cols[b] cells each).  The library sorts the problems into shape classes and
 * runs the classes concurrently, each on its own stream of the handle's device; results come back in problem order.
 * Semantics per problem are those of the uniform entry points above. */
int xpg_six_batch_f64_ragged(xpg_ctx * ctx, int is_max, int nb, const double * tgtf, const double * leq,
                             const int32_t * rows, const int32_t * cols, const long long * leq_offsets,
                             const long long * tgtf_offsets, unsigned max_iter, int32_t * out_status,
                             double * out_v, double * out_sol);
int xpg_six_batch_rat32_ragged(xpg_ctx * ctx, int is_max, int nb, const xpg_rat32 * tgtf, const xpg_rat32 * leq,
                               const int32_t * rows, const int32_t * cols, const long long * leq_offsets,
                               const long long * tgtf_offsets, unsigned max_iter, int32_t * out_status,
                               xpg_rat32 * out_v, xpg_rat32 * out_sol);
/* DepPoly::is_empty (poly.cpp:530-573) on nb polyhedra of mixed shapes, constant in the last column of each. */
int xpg_dep_is_empty_batch_ragged_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, const int32_t * rows,
                                        const int32_t * cols, const long long * offsets, int32_t * out_empty,
                                        long long * out_nodes);
/* Lineq::reduce in place (system b keeps its rows[b] x cols[b] cells at offsets[b], the survivors packed at their
 * front); rhs_idx[b] is its constant column (NULL: the last column of each). */
int xpg_lineq_reduce_batch_ragged_rat32(xpg_ctx * ctx, int nb, xpg_rat32 * mats, const int32_t * rows,
                                        const int32_t * cols, const long long * offsets, const int32_t * rhs_idx,
                                        int is_intersect, int32_t * out_rows, int32_t * out_ok);
/* Lineq::fme eliminating variable u[b] of system b.  Packed result: out_rows[b] rows of cols[b] cells starting at
 * cell out_cell_offsets[b] of outs (out_cell_offsets[nb] = all cells).  outs NULL: a sizing call (returns 0);
 * outs_cap_cells too small: XPG_ERR_SHAPE with out_rows / out_cell_offsets filled. */
int xpg_lineq_fme_batch_ragged_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, const int32_t * rows,
                                     const int32_t * cols, const long long * offsets, const int32_t * rhs_idx,
                                     const int32_t * u, int darkshadow, xpg_rat32 * outs, long long outs_cap_cells,
                                     long long * out_cell_offsets, int32_t * out_rows, int32_t * out_ok);

/* ---- rational row elimination, batches of small systems (one wavefront each) -------------
 * mats is [nb][rows][cols] of xpg_rat32 on the host; rhs_idx is the constant column,
 * columns after it are constant symbols (src/com/linsys.h:64-70).
 *   reduce      : Lineq::reduce(m, rhs_idx, is_intersect), src/com/linsys.cpp:359-626, in
 *                 place; out_rows[b] = surviving rows (packed at the front of system b),
 *                 out_ok[b] = its bool (consistent).
 *   remove_iden : Lineq::removeIdenRow, src/com/linsys.cpp:1209-1268, in place.
 *   fme         : Lineq::fme(u, res, darkshadow), src/com/linsys.cpp:656-774; outs is
 *                 [nb][cap_rows][cols]; out_rows[b] < 0 means the result needs -out_rows[b]
 *                 rows (> cap_rows) and was not written.
 *   rank/det/inv: Matrix<Rational>::rank / det / inv, src/com/matt.h:2614-2726, :1621-1736,
 *                 :1743-1845 (Gauss-Jordan with the reference's pivot preference). */
int xpg_lineq_reduce_batch_rat32(xpg_ctx * ctx, int nb, xpg_rat32 * mats, int rows, int cols,
                                 int rhs_idx, int is_intersect, int32_t * out_rows, int32_t * out_ok);
int xpg_lineq_remove_iden_batch_rat32(xpg_ctx * ctx, int nb, xpg_rat32 * mats, int rows, int cols,
                                      int32_t * out_rows);
/* Lineq::move2var, src/com/linsys.cpp:1177-1200, in place on every system: columns first_sym..last_sym (constant
 * symbols, behind the constant column rhs_idx) are multiplied by -1 and moved in front of column rhs_idx.  A
 * reshaping of the host arrays (no kernel); the multiplication is the reference's Rational '*'. */
int xpg_lineq_move2var_batch_rat32(xpg_ctx * ctx, int nb, xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                   int first_sym, int last_sym);
int xpg_lineq_fme_batch_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                              int rhs_idx, int u, int darkshadow, xpg_rat32 * outs, int cap_rows,
                              int32_t * out_rows, int32_t * out_ok);
/* Lineq::fme (src/com/linsys.cpp:656-774) with a PACKED result -- the form a host-resident caller wants: the
 * worst case of a result is rows^2/4 + rows rows, the typical one a third of it, and the slots of the entry point
 * above come back whole.  Here row_offsets[nb + 1] (in rows) and only the live rows of every system, back to back
 * ([row_offsets[nb]][cols]), travel; both directions go through pinned memory of the handle.
 *   outs      (may be NULL) receives the rows; outs_cap_rows is its capacity in rows.  Too small: XPG_ERR_SHAPE with
 *             row_offsets filled, so the caller can size the buffer and call again.
 *   out_view  (may be NULL) receives a pointer to the same rows in the handle's pinned buffer, valid until the next
 *             packed call on this handle (or xpg_destroy): the zero-copy form the C++ adapter uses.
 *   cap_rows  rows of the device slot per system; <= 0: the worst case.  A result that needs more: XPG_ERR_UNSUPPORTED.
 * out_ok[b] as above.  One handle is used by one host thread at a time. */
int xpg_lineq_fme_batch_packed_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                                     int rhs_idx, int u, int darkshadow, int cap_rows, xpg_rat32 * outs,
                                     long long outs_cap_rows, const xpg_rat32 ** out_view, long long * row_offsets,
                                     int32_t * out_ok);
/* Lineq::reduce (src/com/linsys.cpp:359-626) with a PACKED result, the form a host-resident caller of many systems wants
 * (round 6): mats is read only; row_offsets[nb + 1] (in rows) and the surviving rows of every system, back to back, come
 * back -- the device writes them straight into pinned memory of the handle, so the call synchronises once.  outs /
 * outs_cap_rows / out_view as for the packed fme above; out_rows (may be NULL) receives the per-system counts
 * (= row_offsets[b + 1] - row_offsets[b]); out_ok[b] is Lineq::reduce's bool.  The in-place entry point
 * xpg_lineq_reduce_batch_rat32 is this call plus a copy of every system's survivors to the front of its slot. */
int xpg_lineq_reduce_batch_packed_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                        int is_intersect, xpg_rat32 * outs, long long outs_cap_rows,
                                        const xpg_rat32 ** out_view, long long * row_offsets, int32_t * out_rows,
                                        int32_t * out_ok);
/* Gives the device blocks and pinned staging a handle keeps between host-array calls back to the runtime (they are
 * kept to spare one-system callers four hipMalloc / hipFree pairs per call; at most 1 GiB / 16 blocks). */
int xpg_trim(xpg_ctx * ctx);
/* Lineq::calcBound, src/com/linsys.cpp:1047-1078, for nb systems: for each variable j every
 * other variable is eliminated (innermost first) by chained fme launches that stay on the
 * device.  bounds is [nb][rhs_idx][cap_rows][cols], out_rows is [nb][rhs_idx];
 * out_ok[b] = 1, 0 (an elimination found the system inconsistent) or -rows_needed. */
int xpg_lineq_calc_bound_batch_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                                     int rhs_idx, int cap_rows, xpg_rat32 * bounds, int32_t * out_rows,
                                     int32_t * out_ok);
/* The same with a PACKED result (round 5): the nb * rhs_idx bound systems back to back instead of worst-case slots --
 * row_offsets[b * rhs_idx + j] is the first row of variable j's bounds of system b, row_offsets[nb * rhs_idx] the total; the
 * slots stay in HBM and are compacted there, only live rows cross the link.  outs (may be NULL) has room for outs_cap_rows
 * rows (too small: XPG_ERR_SHAPE with row_offsets filled); out_view (may be NULL) receives a pointer into the handle's
 * pinned buffer, valid until the handle's next call; cap_rows <= 0: 4 * rows + 16.  out_ok[b] = 1, 0 (inconsistent: its
 * chains count as empty) or -rows_needed (then nothing is packed and every offset is 0: call again with that cap_rows). */
int xpg_lineq_calc_bound_batch_packed_rat32(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols, int rhs_idx,
                                            int cap_rows, xpg_rat32 * outs, long long outs_cap_rows, const xpg_rat32 ** out_view,
                                            long long * row_offsets, int32_t * out_ok);
/* The same three on DEVICE arrays (xpg_malloc / any HIP allocation of the handle's device), enqueue only: the
 * results are there after xpg_sync.  For chains of eliminations that stay in HBM, and for measuring the kernels
 * without PCIe.  d_outs is not cleared: rows of a slot past d_out_rows[b] keep what they held. */
int xpg_lineq_reduce_batch_rat32_dev(xpg_ctx * ctx, int nb, xpg_rat32 * d_mats, int rows, int cols, int rhs_idx,
                                     int is_intersect, int32_t * d_out_rows, int32_t * d_out_ok);
int xpg_lineq_fme_batch_rat32_dev(xpg_ctx * ctx, int nb, const xpg_rat32 * d_mats, int rows, int cols, int rhs_idx,
                                  int u, int darkshadow, xpg_rat32 * d_outs, int cap_rows, int32_t * d_out_rows,
                                  int32_t * d_out_ok);
int xpg_rat_rank_batch_dev(xpg_ctx * ctx, int nb, const xpg_rat32 * d_mats, int rows, int cols, int32_t * d_out_rank);
int xpg_rat_rank_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                       int32_t * out_rank);
int xpg_rat_det_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int n, xpg_rat32 * out_det);
int xpg_rat_inv_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int n, xpg_rat32 * out_inv,
                      int32_t * out_ok);
/* Matrix<Rational>::rank(&basis, is_unitarize), src/com/matt.h:2614-2726 (caller
 * src/eng/ldtran.cpp:440-449).  basis is [nb][rows][cols]; basis_rows[b] = rows, except
 * without unitarising and rank < rows, where the reference returns the rank original rows in
 * pivot order (matt.h:2710-2719).  Rows past basis_rows[b] are zero. */
int xpg_rat_rank_basis_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                             int is_unitarize, int32_t * out_rank, xpg_rat32 * basis,
                             int32_t * basis_rows);
/* Matrix<Rational>::null, src/com/matt.h:2546-2584: ns is [nb][cols][cols], column convention. */
int xpg_rat_null_batch(xpg_ctx * ctx, int nb, const xpg_rat32 * mats, int rows, int cols,
                       xpg_rat32 * ns);
/* INTMat::hnf, src/com/xmat.cpp:912-992 (with gen_elim_mat :853-868, exgcd comf.cpp:295-321;
 * caller src/eng/ldtran.cpp:208): h = mat * u, h [nb][rows][cols] lower triangular, u
 * [nb][cols][cols] unimodular.  INT arithmetic is two's-complement 32 bit.  status[b] = 0, or
 * XPG_ERR_REF_UNDEFINED where the reference divides by zero (zero diagonal below row 0) or
 * multiplies by its rows x cols identity read out of bounds (cols > rows, negative diagonal);
 * h and u of such a matrix are left zero. */
int xpg_int_hnf_batch(xpg_ctx * ctx, int nb, const int32_t * mats, int rows, int cols, int32_t * h,
                      int32_t * u, int32_t * status);
/* INTMat::gcd, src/com/xmat.cpp:996-1030: every row divided by the gcd of its nonzero
 * magnitudes, in place. */
int xpg_int_gcd_batch(xpg_ctx * ctx, int nb, int32_t * mats, int rows, int cols);

#ifdef __cplusplus
}
#endif
#endif /* XPOLY_AMD_H */
