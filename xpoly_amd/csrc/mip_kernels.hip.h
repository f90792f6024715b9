// MIP<Mat,T>::RecusivePart (src/com/lpsol.h:2427-2612) for a batch of independent problems, one WORKGROUP per
// problem and the whole depth-first tree walk on the device: the node's problem is rebuilt from the root and the
// branch rows of the path, normalised as SIX::normalize would (for 0-1 problems that is convertEq2Ineq's
// substitution of the branch equalities, lpsol.h:1197-1278, quirks included), solved in LDS by the same code the
// batch kernel runs (sm_solve_lp), and the answer is fed to the reference's recursion written as a stack machine
// -- the very logic of MipTask::on_lp (mip_host.hip.h), statement for statement, run by thread 0 on state that
// lives in a per-problem workspace in HBM.
//
// Why: with the tree walk on the host a batch advances in lock-step rounds -- normalise on the host, one launch,
// feed back on the host -- and at 4-5 nodes per tree the host half and the rounds' tails cost more than the node
// LPs (1024 knapsacks of 24 variables: 22-29 ms, against 13 ms for the reference arithmetic restated on all 256 host cores). Here a tree
// never leaves its workgroup and the trees do not wait for each other.
//
// Scope: x >= 0 (the caller's vc is -I: what xpg_mip_batch_* and the reference's own caller, PolyTran::FeaSchedule,
// src/eng/poly.cpp:5118-5130, pass), inequalities and -- round 3 -- equalities at the root, binary or integer
// branching, with or without a rational_indicator (lpsol.h:2369-2393). General variable constraints (free
// variables, bounds other than x >= 0) keep the host controller.
#pragma once
#include "batch_kernels.hip.h"

namespace xpg {

// Per-problem workspace, in units of 8 bytes (S and long long are both 8): see mip_ws_words.
template <class S> struct MipWs {
    S * L;            // [rmax][cols]  the node's inequalities
    S * y;            // [cols]        raw LP solution of the node
    S * sol;          // [cols]        the recursion's by-reference solution
    S * best_sol;     // [cols]
    S * kept_sol;     // [depth][cols] per frame
    S * kept_v;       // [depth]
    S * vals;         // v, best_v
    int * frame;      // [depth][6]: stage, col, lo, hi, kept, -
    int * forks;      // [cols]
    int * ctl;        // have_best, top, nodes, final_status, speculation sequence number
    S * spec_y;       // [depth][cols] raw LP solution of frame f's CEILING child, solved ahead by a helper workgroup
    int * spec_ctl;   // [depth][2]: (sequence number << 3) | state, the child's SIX status
};
__host__ __device__ inline size_t mip_ws_words(int rmax, int cols, int depth)
{
    size_t w = (size_t)rmax * cols + 3 * (size_t)cols + (size_t)depth * cols + depth + 2;
    w += (size_t)depth * cols;                                   // spec_y
    w += ((size_t)depth * 6 + cols + 8 + (size_t)depth * 2 + 1) / 2;
    return (w + 1) & ~(size_t)1;
}
template <class S> __device__ __forceinline__ MipWs<S> mip_ws_carve(unsigned long long * base, int rmax, int cols, int depth)
{
    MipWs<S> w;
    S * p = (S *)base;
    w.L = p; p += (size_t)rmax * cols;
    w.y = p; p += cols;
    w.sol = p; p += cols;
    w.best_sol = p; p += cols;
    w.kept_sol = p; p += (size_t)depth * cols;
    w.kept_v = p; p += depth;
    w.vals = p; p += 2;
    w.spec_y = p; p += (size_t)depth * cols;
    int * q = (int *)p;
    w.frame = q; q += depth * 6;
    w.forks = q; q += cols;
    w.ctl = q; q += 8;
    w.spec_ctl = q;
    return w;
}

#ifdef XPG_STAMPS
__device__ unsigned long long g_mip_ticks[4];                 // diagnostic builds: 100 MHz ticks in build / solve / feed, summed over nodes
#define MIP_T0 unsigned long long mt_ = wall_clock64();
#define MIP_T(k) { const unsigned long long n_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&g_mip_ticks[k], n_ - mt_); mt_ = n_; }
#else
#define MIP_T0
#define MIP_T(k)
#endif
enum { MF_STAGE = 0, MF_COL = 1, MF_LO = 2, MF_HI = 3, MF_KEPT = 4 };
enum { MC_HAVE_BEST = 0, MC_TOP = 1, MC_NODES = 2, MC_FINAL = 3, MC_SEQ = 4 };
// Speculative solves of ceiling children (round 4). MIP::RecusivePart solves the floor child of a branching node, its whole
// subtree, and then ALWAYS the ceiling child (lpsol.h:2506-2560) -- whose LP depends only on the path to it. While the walk
// is in the floor subtree a HELPER workgroup (extra workgroups at the end of the grid, which get a CU once trees finish)
// solves the ceiling child's LP and parks (status, raw solution) in the tree's workspace; when the walk arrives there it
// takes the parked answer instead of building and solving the node. Everything that depends on the order of the walk --
// is_satisfying, the best-so-far test, forks, remember -- still runs in the walk (mip_feed), so results are the walk's.
// States of a frame's slot: requested by the walk -> claimed by a helper -> done; the walk cancels a request no helper has
// claimed when it gets there first, and waits for one that is claimed (the helper started earlier than the walk could).
enum { SP_NONE = 0, SP_REQ = 1, SP_CLAIMED = 2, SP_DONE = 3, SP_CANCEL = 4, SP_QCAP = 1 << 16 };
// queue (ints): [0] tail, [1] head, [2] trees finished, [3] parked answers taken, then valid[SP_QCAP], then entries[SP_QCAP][4] {block, frame, seq, -}
__device__ __forceinline__ int * spq_valid(int * q) { return q + 4; }
__device__ __forceinline__ int * spq_entry(int * q, int slot) { return q + 4 + SP_QCAP + 4 * slot; }
__host__ __device__ inline size_t spq_bytes() { return (size_t)(4 + SP_QCAP + 4 * SP_QCAP) * 4; }
__device__ __forceinline__ int sp_load(const int * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sp_store(int * p, int x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool sp_cas(int * p, int expect, int want)
{ return __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool mip_int_cast_ok(F64) { return true; }
__device__ __forceinline__ bool mip_int_cast_ok(R32 a) { return a.den != 0; }

// The node's problem (MipTask::push_branch + normalize_host): root inequalities, then -- integer branching -- one
// bound row per ancestor, or -- 0-1 branching -- the ancestors' equalities x_col = b substituted into the
// inequalities column by column (fold_eq of six_host.hip.h; every branch equality has one variable, so each is
// "the only nonzero of its column" and none is left to become a pair of inequalities). All threads; returns the
// row count, or a negative status where the reference's behaviour is undefined.
template <class S> __device__ int mip_build_node_eq(const MipWs<S> & w, const S * root_eq, int eq_rows, int cols, bool is_bin,
                                                     int top, int rows, int * sh_flag);
template <class S> __device__ int mip_build_node(const MipWs<S> & w, const S * root_leq, int leq_rows, const S * root_eq, int eq_rows,
                                                  int cols, bool is_bin, int top, int * sh_flag)
{   // (a helper builds the ceiling child of frame f from a COPY of the frames with frame f's stage set to 2, top = f + 1)
    const int rhs0 = cols - 1;
    for (int t = threadIdx.x; t < leq_rows * cols; t += blockDim.x) w.L[t] = root_leq[t];
    if (threadIdx.x == 0) *sh_flag = 0;
    int rows = leq_rows;
    if (eq_rows > 0) {                                       // equalities at the root: the general convertEq2Ineq
        if (!is_bin)
            for (int f = 0; f < top; f++) {
                const int * fr = w.frame + f * 6;
                const bool ceiling = fr[MF_STAGE] == 2;
                for (int j = threadIdx.x; j < cols; j += blockDim.x) {
                    S val = zero<S>();
                    if (j == fr[MF_COL]) val = ceiling ? minus_one<S>() : one<S>();
                    if (j == rhs0) val = S::from_int(ceiling ? -fr[MF_HI] : fr[MF_LO]);
                    w.L[(size_t)rows * cols + j] = val;
                }
                rows++;
            }
        __syncthreads();
        return mip_build_node_eq<S>(w, root_eq, eq_rows, cols, is_bin, top, rows, sh_flag);
    }
    if (!is_bin) {
        for (int f = 0; f < top; f++) {                      // frame f's child row (lpsol.h:2514-2520, :2555-2559)
            const int * fr = w.frame + f * 6;
            const bool ceiling = fr[MF_STAGE] == 2;
            for (int j = threadIdx.x; j < cols; j += blockDim.x) {
                S val = zero<S>();
                if (j == fr[MF_COL]) val = ceiling ? minus_one<S>() : one<S>();
                if (j == rhs0) val = S::from_int(ceiling ? -fr[MF_HI] : fr[MF_LO]);
                w.L[(size_t)rows * cols + j] = val;
            }
            rows++;
        }
        __syncthreads();
        return rows;
    }
    __syncthreads();
    for (int j = 0; j < rhs0; j++) {                         // convertEq2Ineq, lpsol.h:1218-1262
        int at = -1;
        for (int f = 0; f < top; f++) if (w.frame[f * 6 + MF_COL] == j) { at = f; break; }
        if (at < 0) continue;
        const int * fr = w.frame + at * 6;
        const S b = S::from_int(fr[MF_STAGE] == 2 ? fr[MF_HI] : fr[MF_LO]);   // lpsol.h:2506-2512, :2548-2553
        for (int q = threadIdx.x; q < rows; q += blockDim.x) {
            S * Lq = w.L + (size_t)q * cols;
            const S coef = Lq[j];
            if (eq(coef, zero<S>())) continue;
            if (q >= cols) { *sh_flag = 1; continue; }       // the reference reads the equality at the ROW's index
            // the equality row: 1 in column j, b in the constant column
            const S lead = q == j ? one<S>() : (q == rhs0 ? b : zero<S>());
            const bool rescale = ne(lead, one<S>());
            const S x1 = rescale ? q_div(false, one<S>(), lead) : one<S>();
            const int m1 = rescale ? scale_mode(x1) : SCALE_KEEP, m2 = scale_mode(coef);
            Lq[j] = zero<S>();
            for (int k = 0; k < cols; k++) {
                S t = k == j ? one<S>() : (k == rhs0 ? b : zero<S>());
                t = q_scaled(false, t, x1, m1);
                t = q_scaled(false, t, coef, m2);
                if (k >= rhs0) t = neg(t);
                Lq[k] = q_add(false, t, Lq[k]);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    return *sh_flag ? XPG_ERR_REF_UNDEFINED : rows;
}

// The same with equalities at the root (MipTask::push_branch + fold_eq of six_host.hip.h = SIX::convertEq2Ineq,
// lpsol.h:1197-1278, on the node's equality list: the root's rows, then -- 0-1 branching -- one x_col = b row per
// ancestor). Columns are visited in order; a column in which exactly ONE not-yet-used equality has a nonzero is
// substituted into every inequality that mentions it (the reference reads the equality's leading entry at the
// INEQUALITY's row index, lpsol.h:1232: reproduced, refused once it would leave the row), and the equalities left
// over become pairs of opposite inequalities below the others. Wave 0 finds the next (column, equality) with
// ballots over the equality rows; the substitution runs over the inequality rows in parallel. MIP_EQ_MAX bounds the
// equality list (root rows + depth).
enum { MIP_EQ_MAX = 256 };
template <class S> __device__ __forceinline__ S mip_eq_cell(const S * root_eq, int eq_rows, int cols, const int * frame, int i, int k)
{
    if (i < eq_rows) return root_eq[(size_t)i * cols + k];
    const int * fr = frame + (i - eq_rows) * 6;              // a branch equality: 1 in its column, b in the constant column
    if (k == fr[MF_COL]) return one<S>();
    if (k == cols - 1) return S::from_int(fr[MF_STAGE] == 2 ? fr[MF_HI] : fr[MF_LO]);   // lpsol.h:2506-2512, :2548-2553
    return zero<S>();
}
template <class S> __device__ int mip_build_node_eq(const MipWs<S> & w, const S * root_eq, int eq_rows, int cols, bool is_bin,
                                                     int top, int rows, int * sh_flag)
{
    __shared__ unsigned char sh_used[MIP_EQ_MAX];
    __shared__ short sh_left[MIP_EQ_MAX];
    __shared__ int sh_pick[3];                               // column, equality, equalities left over
    const int rhs0 = cols - 1, lane = threadIdx.x & 63;
    const int ne_rows = eq_rows + (is_bin ? top : 0);
    for (int i = threadIdx.x; i < ne_rows; i += blockDim.x) sh_used[i] = 0;
    __syncthreads();
    int from = 0;
    while (rows > 0) {                                       // (no inequality at all: nothing to substitute into)
        if (threadIdx.x < 64) {
            int fj = -1, fat = -1;
            for (int j = from; j < rhs0 && fj < 0; j++) {
                int hits = 0, at = -1;
                for (int base = 0; base < ne_rows; base += 64) {
                    const int i = base + lane;
                    const bool nz = i < ne_rows && !sh_used[i] && ne(mip_eq_cell<S>(root_eq, eq_rows, cols, w.frame, i, j), zero<S>());
                    const unsigned long long bal = __ballot(nz);
                    hits += __popcll(bal);
                    if (bal) at = base + 63 - __clzll((long long)bal);
                }
                if (hits == 1) { fj = j; fat = at; }
            }
            if (lane == 0) { sh_pick[0] = fj; sh_pick[1] = fat; if (fat >= 0) sh_used[fat] = 1; }
        }
        __syncthreads();
        const int j = sh_pick[0], at = sh_pick[1];
        if (j < 0) break;
        for (int q = threadIdx.x; q < rows; q += blockDim.x) {
            S * Lq = w.L + (size_t)q * cols;
            const S coef = Lq[j];
            if (eq(coef, zero<S>())) continue;
            if (q >= cols) { *sh_flag = 1; continue; }       // the reference reads the equality at the ROW's index
            const S lead = mip_eq_cell<S>(root_eq, eq_rows, cols, w.frame, at, q);
            const bool rescale = ne(lead, one<S>());
            const S x1 = rescale ? q_div(false, one<S>(), lead) : one<S>();
            const int m1 = rescale ? scale_mode(x1) : SCALE_KEEP, m2 = scale_mode(coef);
            Lq[j] = zero<S>();
            for (int k = 0; k < cols; k++) {
                S t = mip_eq_cell<S>(root_eq, eq_rows, cols, w.frame, at, k);
                t = q_scaled(false, t, x1, m1);
                t = q_scaled(false, t, coef, m2);
                if (k >= rhs0) t = neg(t);
                Lq[k] = q_add(false, t, Lq[k]);
            }
        }
        __syncthreads();
        from = j + 1;
    }
    // the equalities left over, in order: -row then +row (lpsol.h:1264-1277)
    if (threadIdx.x < 64) {
        int left = 0;
        for (int base = 0; base < ne_rows; base += 64) {
            const int i = base + lane;
            const bool open = i < ne_rows && !sh_used[i];
            const unsigned long long bal = __ballot(open);
            if (open) sh_left[left + __popcll(bal & ((1ull << lane) - 1ull))] = (short)i;
            left += __popcll(bal);
        }
        if (lane == 0) sh_pick[2] = left;
    }
    __syncthreads();
    const int left = sh_pick[2];
    for (int t = threadIdx.x; t < left * cols; t += blockDim.x) {
        const int u = t / cols, k = t - u * cols;
        const S c = mip_eq_cell<S>(root_eq, eq_rows, cols, w.frame, sh_left[u], k);
        w.L[(size_t)(rows + 2 * u) * cols + k] = q_mul(false, c, minus_one<S>());
        w.L[(size_t)(rows + 2 * u + 1) * cols + k] = c;
    }
    rows += 2 * left;
    __syncthreads();
    if (*sh_flag) return XPG_ERR_REF_UNDEFINED;
    return rows > 0 ? rows : XPG_ERR_SHAPE;
}

// First half of finish_host for a solved node, by all threads: sol = (y, 1); the products sol[j] * tgtf[j] of the objective
// (into y: the raw values are not needed again) and the reduced solution entries. The sum of the products keeps the
// reference's order and stays with thread 0 (mip_feed); a third of the feed-back's serial time was these loops.
template <class S> __device__ __forceinline__ void mip_feed_products(const MipWs<S> & w, const S * tgtf, int cols)
{
    const int n0 = cols - 1;
    for (int j = threadIdx.x; j < cols; j += blockDim.x) {
        const S x = j < n0 ? w.y[j] : one<S>();
        w.y[j] = q_mul(false, x, tgtf[j]);
        S t = x; reduce(t);
        w.sol[j] = t;
    }
    __syncthreads();
}

// MipTask::on_lp: feeds the node's answer to the recursion and runs it until the next LP is needed (returns
// false) or the tree ends (returns true, ctl[MC_FINAL] set). Thread 0 only.
template <class S> __device__ bool mip_feed(const MipWs<S> & w, int cols, bool is_max, bool is_bin, int st, const uint8_t * allow)
{
    int top = w.ctl[MC_TOP];
    S v = zero<S>();
    int ret;
    {
        int * f = w.frame + top * 6;
        if (st == XPG_SIX_SUCC) {                            // finish_host (SIX::calcFinalSolution, lpsol.h:1851-1899), second half:
            for (int j = 0; j < cols; j++) v = q_add(false, v, w.y[j]);   // the sum, in order, of the products mip_feed_products left
            reduce(v);
        }
        if (st < 0) ret = st;
        else if (st == XPG_SIX_UNBOUND) ret = XPG_IP_UNBOUND;
        else if (st == XPG_SIX_TIME_OUT) ret = XPG_ERR_REF_UNDEFINED;         // UNREACH() in the reference
        else if (st != XPG_SIX_SUCC) ret = XPG_IP_NO_PRI_FEASIBLE_SOL;
        else {
            int col = 0;
            bool sat = true;                                 // MIP::is_satisfying (lpsol.h:2364-2408)
            for (int j = 0; j < cols; j++) {
                S t = w.sol[j];
                if (allow || is_bin) { reduce(t); w.sol[j] = t; }
                bool bad;
                if (allow) bad = !allow[j] && (!is_int(t) || (is_bin && ne(t, zero<S>()) && ne(t, one<S>())));   // lpsol.h:2369-2393
                else bad = is_bin ? (ne(t, zero<S>()) && ne(t, one<S>())) : !is_int(t);
                if (bad) { col = j; sat = false; break; }
            }
            const bool have_best = w.ctl[MC_HAVE_BEST] != 0;
            const S best_v = w.vals[1];
            if (sat) ret = XPG_IP_SUCC;
            else if (have_best && (is_max ? le(v, best_v) : ge(v, best_v))) ret = XPG_IP_NO_BETTER_THAN_BEST_SOL;
            else if (w.forks[col] >= 1) ret = XPG_IP_NO_PRI_FEASIBLE_SOL;                  // lpsol.h:2486-2496
            else if (!is_bin && !mip_int_cast_ok(w.sol[col])) ret = XPG_ERR_REF_UNDEFINED;
            else {
                w.forks[col] += 1;
                f[MF_COL] = col; f[MF_LO] = 0; f[MF_HI] = 1;
                if (!is_bin) { f[MF_LO] = to_int(w.sol[col]); f[MF_HI] = f[MF_LO] + 1; }
                f[MF_STAGE] = 1;
                int * c = w.frame + (top + 1) * 6;           // push_branch(floor)
                c[MF_STAGE] = 0; c[MF_COL] = 0; c[MF_LO] = 0; c[MF_HI] = 1; c[MF_KEPT] = 0;
                w.kept_v[top + 1] = zero<S>();
                w.ctl[MC_TOP] = top + 1;
                w.vals[0] = v;
                return false;
            }
        }
    }
    for (;;) {                                               // hand `ret` to the callers up the stack
        top -= 1;
        if (top < 0) { w.ctl[MC_FINAL] = ret; w.ctl[MC_TOP] = 0; w.vals[0] = v; return true; }
        int * p = w.frame + top * 6;
        if (ret < 0) continue;
        if (p[MF_STAGE] == 1) {                              // floor branch came back, lpsol.h:2527-2543
            if (ret == XPG_IP_SUCC) {
                for (int j = 0; j < cols; j++) w.kept_sol[(size_t)top * cols + j] = w.sol[j];
                w.kept_v[top] = v; p[MF_KEPT] = 1;
                const bool have_best = w.ctl[MC_HAVE_BEST] != 0;                           // remember()
                if (!have_best || (is_max ? lt(w.vals[1], v) : gt(w.vals[1], v))) {
                    for (int j = 0; j < cols; j++) w.best_sol[j] = w.sol[j];
                    w.vals[1] = v; w.ctl[MC_HAVE_BEST] = 1;
                }
            }
            p[MF_STAGE] = 2;
            int * c = w.frame + (top + 1) * 6;               // push_branch(ceiling)
            c[MF_STAGE] = 0; c[MF_COL] = 0; c[MF_LO] = 0; c[MF_HI] = 1; c[MF_KEPT] = 0;
            w.kept_v[top + 1] = zero<S>();
            w.ctl[MC_TOP] = top + 1;
            w.vals[0] = v;
            return false;
        }
        bool rem = false;
        if (ret == XPG_IP_SUCC) {                            // ceiling branch came back, lpsol.h:2563-2611
            if (p[MF_KEPT] && (is_max ? gt(w.kept_v[top], v) : lt(w.kept_v[top], v))) {
                v = w.kept_v[top];
                for (int j = 0; j < cols; j++) w.sol[j] = w.kept_sol[(size_t)top * cols + j];
            }
            rem = true;
        } else if (p[MF_KEPT]) {
            v = w.kept_v[top];
            for (int j = 0; j < cols; j++) w.sol[j] = w.kept_sol[(size_t)top * cols + j];
            rem = true; ret = XPG_IP_SUCC;
        }
        if (rem) {
            const bool have_best = w.ctl[MC_HAVE_BEST] != 0;
            if (!have_best || (is_max ? lt(w.vals[1], v) : gt(w.vals[1], v))) {
                for (int j = 0; j < cols; j++) w.best_sol[j] = w.sol[j];
                w.vals[1] = v; w.ctl[MC_HAVE_BEST] = 1;
            }
        }
    }
}

template <class S> __global__ __launch_bounds__(256, 2)
void k_mip_tree(int nb, const S * tgtf_all, const S * leq_all, int leq_rows, int cols, int is_max, int is_bin, int rmax,
                int depth, unsigned long long * ws_all, size_t ws_words, int32_t * out_status, S * out_v, S * out_sol,
                int * out_nodes, const int * rows_of, const int * active, const uint8_t * allow, const S * eq_all, int eq_rows,
                int nmain, int * spq)
{
    // eq_all / eq_rows (may be NULL / 0): eq_rows equalities per problem at the root.
    // allow (may be NULL): MIP's rational_indicator, one row of cols flags shared by the batch -- variables whose flag is
    // set may stay fractional (lpsol.h:2369-2393).
    // rows_of (may be NULL): problem b has rows_of[b] of its leq_rows-row slot live (ragged batches: the systems
    // Lineq::reduce leaves); active (may be NULL): only problems whose entry is 1 are walked, the others' outputs
    // are left alone.
    // nmain: the first nmain workgroups walk trees (tree b by workgroup b % nmain); spq != NULL: speculation (see SP_*): then
    // nmain == nb (one tree per walking workgroup, its workspace never reused) and the workgroups behind them are helpers.
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int sh_ctl[8];
    __shared__ unsigned long long sh_v;                      // the node's own objective: recomputed by mip_feed
    const int n = cols - 1;
    Small<S> P;
    sm_carve(P, lds, is_max ? rmax : n, is_max ? n : rmax);
    if ((int)blockIdx.x >= nmain) {
        // ================================ a helper ======================================================
        if (!spq) return;
        const MipWs<S> hw = mip_ws_carve<S>(ws_all + (size_t)blockIdx.x * ws_words, rmax, cols, depth);
        for (;;) {
            if (threadIdx.x == 0) {
                int verdict = 0, blk = 0, f = 0, seq = 0;   // 0: leave, 1: next task, 2: solve
                const int my = atomicAdd(&spq[1], 1);
                unsigned spins = 0;
                for (;;) {
                    if (my < SP_QCAP && sp_load(&spq_valid(spq)[my]) == my + 1) { verdict = 1; break; }
                    if (my >= SP_QCAP || (sp_load(&spq[2]) >= nmain && sp_load(&spq[0]) <= my)) break;   // nothing more will come
                    if (++spins > (1u << 24)) break;        // (seconds: a walk that never ends its tree; leave rather than hang)
                    __builtin_amdgcn_s_sleep(8);
                }
                if (verdict == 1) {
                    __threadfence();
                    const int * e = spq_entry(spq, my);
                    blk = sp_load(&e[0]); f = sp_load(&e[1]); seq = sp_load(&e[2]);
                    const MipWs<S> tw = mip_ws_carve<S>(ws_all + (size_t)blk * ws_words, rmax, cols, depth);
                    if (sp_cas(&tw.spec_ctl[2 * f], (seq << 3) | SP_REQ, (seq << 3) | SP_CLAIMED)) verdict = 2;   // (else: cancelled)
                }
                sh_ctl[4] = verdict; sh_ctl[5] = blk; sh_ctl[6] = f; sh_ctl[7] = seq;
            }
            __syncthreads();
            const int verdict = sh_ctl[4], blk = sh_ctl[5], f = sh_ctl[6], seq = sh_ctl[7];
            __syncthreads();
            if (verdict == 0) return;
            if (verdict == 1) continue;
            __threadfence();                                 // acquire: the walk's frames (published before the request)
            const MipWs<S> tw = mip_ws_carve<S>(ws_all + (size_t)blk * ws_words, rmax, cols, depth);
            const int b = blk;                               // (one tree per walking workgroup)
            // the frames of the path, copied; frame f in its ceiling stage
            for (int t = threadIdx.x; t < (f + 1) * 6; t += blockDim.x) {
                int x = tw.frame[t];
                if (t == f * 6 + MF_STAGE) x = 2;
                hw.frame[t] = x;
            }
            __syncthreads();
            const S * tgtf = tgtf_all + (size_t)b * cols;
            const S * root = leq_all + (size_t)b * leq_rows * cols;
            const int my_rows = rows_of ? rows_of[b] : leq_rows;
            int st = mip_build_node<S>(hw, root, my_rows, (const S *)0, 0, cols, is_bin != 0, f + 1, &sh_ctl[1]);
            if (st >= 0) {
                Source<S> src;
                src.leq = hw.L; src.tgtf = tgtf; src.m = st; src.cols = cols; src.is_max = is_max;
                st = sm_solve_lp<S>(P, src, 10000u, /*raw_sol=*/1, hw.y, (S *)&sh_v);
            }
            __syncthreads();
            for (int j = threadIdx.x; j < cols; j += blockDim.x) tw.spec_y[(size_t)f * cols + j] = hw.y[j];
            if (threadIdx.x == 0) tw.spec_ctl[2 * f + 1] = st;
            __threadfence();                                 // release: the answer before the state
            __syncthreads();
            if (threadIdx.x == 0) sp_store(&tw.spec_ctl[2 * f], (seq << 3) | SP_DONE);
        }
    }
    for (int b = blockIdx.x; b < nb; b += nmain) {
        if (active && active[b] != 1) { if (spq && threadIdx.x == 0) atomicAdd(&spq[2], 1); continue; }
        const S * tgtf = tgtf_all + (size_t)b * cols;
        const S * root = leq_all + (size_t)b * leq_rows * cols;
        const S * root_eq = eq_rows > 0 ? eq_all + (size_t)b * eq_rows * cols : (const S *)0;
        const int my_rows = rows_of ? rows_of[b] : leq_rows;
        const MipWs<S> w = mip_ws_carve<S>(ws_all + (size_t)blockIdx.x * ws_words, rmax, cols, depth);
        // MipTask::start
        for (int j = threadIdx.x; j < cols; j += blockDim.x) w.forks[j] = 0;
        for (int j = threadIdx.x; j < 2 * depth; j += blockDim.x) w.spec_ctl[j] = 0;
        if (threadIdx.x == 0) {
            w.ctl[MC_HAVE_BEST] = 0; w.ctl[MC_TOP] = 0; w.ctl[MC_NODES] = 0; w.ctl[MC_FINAL] = 0; w.ctl[MC_SEQ] = 0;
            w.vals[0] = zero<S>(); w.vals[1] = zero<S>();
            int * f = w.frame;
            f[MF_STAGE] = 0; f[MF_COL] = 0; f[MF_LO] = 0; f[MF_HI] = 1; f[MF_KEPT] = 0;
            w.kept_v[0] = zero<S>();
        }
        __syncthreads();
        for (;;) {
            const int top = w.ctl[MC_TOP];
            __syncthreads();
            if (threadIdx.x == 0) w.ctl[MC_NODES] += 1;
            MIP_T0
            int st = 0;
            bool parked = false;
            if (spq && top > 0) {
                // is this node a ceiling child whose LP a helper was asked to solve?
                if (threadIdx.x == 0) {
                    int got = 0;
                    int * sc = w.spec_ctl + 2 * (top - 1);
                    const int mine = sp_load(&sc[0]);
                    if (w.frame[(top - 1) * 6 + MF_STAGE] == 2 && (mine & 7) != SP_NONE) {
                        const int seq = mine >> 3;
                        unsigned spins = 0;
                        for (;;) {
                            const int cur = sp_load(&sc[0]);
                            if (cur == ((seq << 3) | SP_DONE)) { got = 1; break; }
                            if (cur == ((seq << 3) | SP_REQ)) { if (sp_cas(&sc[0], cur, (seq << 3) | SP_CANCEL)) break; continue; }
                            if (cur != ((seq << 3) | SP_CLAIMED)) break;
                            if (++spins > (1u << 26)) break;                  // (a helper that never finishes: solve it here after all)
                            __builtin_amdgcn_s_sleep(4);
                        }
                        if (got) { __threadfence(); atomicAdd(&spq[3], 1); }   // acquire: the parked answer behind its state (spq[3]: answers taken)
                        // (a slot left CLAIMED by the timeout above is never requested again: its late answer can land)
                        if (sp_load(&sc[0]) != ((seq << 3) | SP_CLAIMED)) sp_store(&sc[0], SP_NONE);
                    }
                    sh_ctl[2] = got; sh_ctl[3] = got ? sc[1] : 0;
                }
                __syncthreads();
                parked = sh_ctl[2] != 0;
                if (parked) {
                    st = sh_ctl[3];
                    for (int j = threadIdx.x; j < cols; j += blockDim.x) w.y[j] = w.spec_y[(size_t)(top - 1) * cols + j];
                }
                __syncthreads();
            }
            if (!parked) {
                st = mip_build_node<S>(w, root, my_rows, root_eq, eq_rows, cols, is_bin != 0, top, &sh_ctl[1]);
                MIP_T(0)
                if (st >= 0) {
                    Source<S> src;
                    src.leq = w.L; src.tgtf = tgtf; src.m = st; src.cols = cols; src.is_max = is_max;
                    st = sm_solve_lp<S>(P, src, 10000u, /*raw_sol=*/1, w.y, (S *)&sh_v);
                }
            }
            MIP_T(1)
            if (st == XPG_SIX_SUCC) mip_feed_products<S>(w, tgtf, cols);       // (st is the same in every thread)
            if (threadIdx.x == 0) {
                const bool ended = mip_feed<S>(w, cols, is_max != 0, is_bin != 0, st, allow);
                sh_ctl[0] = ended ? 1 : 0;
                if (spq && !ended) {
                    // the walk has just branched (the frame under the new top is in its floor stage): ask for its ceiling child
                    const int nt = w.ctl[MC_TOP];
                    if (nt > 0 && w.frame[(nt - 1) * 6 + MF_STAGE] == 1 && sp_load(&w.spec_ctl[2 * (nt - 1)]) == SP_NONE) {
                        const int slot = atomicAdd(&spq[0], 1);
                        if (slot < SP_QCAP) {
                            const int seq = ++w.ctl[MC_SEQ];
                            w.spec_ctl[2 * (nt - 1) + 1] = 0;
                            sp_store(&w.spec_ctl[2 * (nt - 1)], (seq << 3) | SP_REQ);
                            int * e = spq_entry(spq, slot);
                            e[0] = (int)blockIdx.x; e[1] = nt - 1; e[2] = seq; e[3] = 0;
                            __threadfence();                 // release: frames, slot state and entry before the ticket
                            sp_store(&spq_valid(spq)[slot], slot + 1);
                        }
                    }
                }
            }
            __syncthreads();
            MIP_T(2)
            if (sh_ctl[0]) break;
        }
        if (threadIdx.x == 0) {
            const int fin = w.ctl[MC_FINAL];
            out_status[b] = fin;
            out_v[b] = w.vals[0];
            out_nodes[b] = w.ctl[MC_NODES];
            if (spq) atomicAdd(&spq[2], 1);
        }
        if (w.ctl[MC_FINAL] == XPG_IP_SUCC && out_sol)
            for (int j = threadIdx.x; j < cols; j += blockDim.x) out_sol[(size_t)b * cols + j] = w.sol[j];
        __syncthreads();
    }
}

// ---- DepPoly::is_empty (src/eng/poly.cpp:530-573) after the reduce, on the device ------------------------------
// Per polyhedron b with reduce's outputs kept[b] / ok[b]: the verdict reduce alone gives, else the feasibility
// objective of Lineq::has_solution (SIX::reviseTargetFunc on all ones, lpsol.h:2053-2074 / linsys.cpp:851-862: 1 for
// every variable that occurs in some inequality) and the "still open" mark for the MIP walks that follow.
__global__ void k_dep_prepare(int nb, const R32 * mats, int rows, int cols, const int * kept, const int * ok, R32 * tgtf,
                              int * active, int32_t * empty)
{
    const int b = blockIdx.x * blockDim.y + threadIdx.y;
    if (b >= nb) return;
    const int last = cols - 1, k = kept[b];
    const bool open = ok[b] != 0 && k > 0;
    if (threadIdx.x == 0) {
        active[b] = open ? 1 : 0;
        empty[b] = !ok[b] ? 1 : (k == 0 ? 0 : 1);       // inconsistent bounds: empty; only redundant constraints: not
    }
    const R32 * m = mats + (size_t)b * rows * cols;
    for (int j = threadIdx.x; j < cols; j += blockDim.x) {
        bool nz = false;
        if (open && j < last)
            for (int i = 0; i < k && !nz; i++) nz = ne(m[(size_t)i * cols + j], R32(0, 1));
        tgtf[(size_t)b * cols + j] = nz ? R32(1, 1) : R32(0, 1);
    }
}
// After a walk (maxm, then minm; linsys.cpp:864-876): success = a solution exists = not empty, decided; a negative
// status = the reference is undefined on this system, decided; anything else stays open for the next walk.
__global__ void k_dep_update(int nb, const int32_t * status, int * active, int32_t * empty)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb || active[b] != 1) return;
    const int st = status[b];
    if (st < 0) { empty[b] = st; active[b] = 0; }
    else if (st == XPG_IP_SUCC) { empty[b] = 0; active[b] = 0; }
}

} // namespace xpg
