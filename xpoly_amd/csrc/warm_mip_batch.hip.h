// OPT-IN, NON-PARITY, batch form: nb independent integer programs
//     maximise c . x   subject to   A x <= b,  x >= 0,  x integer
// each solved by ONE WORKGROUP with the tree's current tableau in LDS: branch and bound that re-optimises every node from
// its parent's final tableau with the DUAL simplex (the reference builds a fresh SIX per node and solves the grown problem
// from the slack form, src/com/lpsol.h:2440-2448; its MIP::RecusivePart is lpsol.h:2427-2612). warm_mip.hip.h is the same
// method for ONE tree on an HBM-resident tableau, driven from the host with a round trip per six dual pivots; here the
// whole walk -- root solve, the depth-first stack, snapshots, bounding -- runs inside the launch, a tree per workgroup, as
// the parity walk k_mip_tree does (mip_kernels.hip.h), and a batch of 1024 knapsacks is one launch.
//
// Layout of a tree's tableau (doubles, LDS): rows i < m of  x_B + sum a_ij x_N = b_i  over columns
//     [0, n0) structural | [n0, n0 + mcap) one slack per POSSIBLE row (row i's slack is column n0 + i) | constant
// so that a bound row appended at depth d uses row m0 + d and the slack column that is already there: nothing shifts.
// The objective is kept as the z-row  z - sum d_j x_j = z_c  (o_j = -d_j, constant z_c): one elimination rule for every
// row. A child is its parent's solved state plus ONE bound row written in the parent's basis
//     x_j <= floor(x_j*)      or      -x_j <= -ceil(x_j*)
// (if x_j is basic in row q the row is -/+ row q with the x_j coefficient cancelled: the new slack is basic with a
// negative constant), after which the basis is still dual feasible and a few dual pivots restore primal feasibility.
// Depth first, floor child first, best incumbent, bounding by the relaxation; a node on the stack is a snapshot of its
// solved state in the tree's HBM workspace (live rows only), restored by the workgroup when it comes back to it.
// Results are checked against the mathematics (scipy's HiGHS milp), not against the reference's walk, whose answers
// depend on its fork counter (lpsol.h:2474-2497): tests/test_gpu_warm_mip.py.
#pragma once
#include <vector>
#include "ctx.hip.h"
#include "scalar.hip.h"

namespace xpg {

struct WbShape {
    int n0, m0, mcap, wcap;       // variables, rows of the root, row capacity, row stride (n0 + mcap + 1)
    int depth_cap;                // bound rows a path may append (mcap - m0) = stack levels
    unsigned long long snap_stride, tree_stride;   // doubles per snapshot / per tree in the workspace
};
enum { WB_THREADS = 256, WB_MAX_PIVOTS = 20000 };

// (value, index) minimum over the workgroup, lowest index among equal values; idx < 0 on every thread = no candidate.
// red: 8 doubles + 8 ints of LDS. Every thread returns the same pair.
__device__ __forceinline__ void wb_argmin(double & val, int & idx, double * red_v, int * red_i)
{
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(val, o); const int i2 = __shfl_xor(idx, o);
        const bool take = i2 >= 0 && (idx < 0 || v2 < val || (v2 == val && i2 < idx));
        if (take) { val = v2; idx = i2; }
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();                                   // (the scratch of the reduction before this one has been read)
    if ((threadIdx.x & 63) == 0) { red_v[w] = val; red_i[w] = idx; }
    __syncthreads();
    double bv = red_v[0]; int bi = red_i[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); k++) {
        const double v2 = red_v[k]; const int i2 = red_i[k];
        if (i2 >= 0 && (bi < 0 || v2 < bv || (v2 == bv && i2 < bi))) { bv = v2; bi = i2; }
    }
    val = bv; idx = bi;
}

struct WbLds {
    double * T; double * obj; double * prow; double * pcol; double * c0; double * red_v;
    int * bv_row; int * eq2bv; int * red_i;
};
inline size_t wb_lds_bytes(const WbShape & S)
{
    size_t b = ((size_t)S.mcap * S.wcap + 2 * (size_t)S.wcap + S.mcap + S.n0 + 8) * 8;
    b += ((size_t)S.wcap + S.mcap + 8) * 4;
    return (b + 15) & ~(size_t)15;
}
__device__ __forceinline__ void wb_carve(WbLds & L, unsigned char * lds, const WbShape & S)
{
    double * d = (double *)lds;
    L.T = d; d += (size_t)S.mcap * S.wcap;
    L.obj = d; d += S.wcap;
    L.prow = d; d += S.wcap;
    L.pcol = d; d += S.mcap;
    L.c0 = d; d += S.n0;
    L.red_v = d; d += 8;
    int * q = (int *)d;
    L.bv_row = q; q += S.wcap;
    L.eq2bv = q; q += S.mcap;
    L.red_i = q;
}

// One pivot on (r, e): row r scaled, every other row and the z-row eliminated, the basis swapped. m live rows; columns of
// slacks beyond row m - 1 do not exist yet (all zero) and are skipped with the rest of a row's tail by `live`.
__device__ __forceinline__ void wb_pivot(const WbLds & L, const WbShape & S, int m, int r, int e)
{
    const int cst = S.wcap - 1, live = S.n0 + m;            // live columns [0, live) + the constant
    const double piv = L.T[(size_t)r * S.wcap + e];
    for (int j = threadIdx.x; j <= live; j += blockDim.x) {
        const int c = j < live ? j : cst;
        L.prow[c] = L.T[(size_t)r * S.wcap + c] / piv;
    }
    for (int i = threadIdx.x; i <= m; i += blockDim.x) L.pcol[i] = i < m ? L.T[(size_t)i * S.wcap + e] : L.obj[e];
    __syncthreads();
    // a wave per row (rows w, w + waves, ...; the z-row is row m), a lane per column: no index arithmetic, the row's factor is
    // wave-uniform -- and a row whose entry in the pivot column is zero is left alone (most rows of a 0-1 program: its x_j <= 1
    // rows touch one structural column each)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    for (int i = wv; i <= m; i += nwv) {
        const double f = L.pcol[i];
        if (i != r && f == 0.0) continue;
        double * row = i < m ? &L.T[(size_t)i * S.wcap] : L.obj;
        for (int jj = lane; jj <= live; jj += 64) {
            const int c = jj < live ? jj : cst;
            if (i == r) row[c] = c == e ? 1.0 : L.prow[c];
            else row[c] = c == e ? 0.0 : row[c] - f * L.prow[c];
        }
    }
    if (threadIdx.x == 0) { const int lv = L.eq2bv[r]; L.bv_row[lv] = -1; L.bv_row[e] = r; L.eq2bv[r] = e; }
    __syncthreads();
}

// Dual simplex from a dual-feasible basis: 0 = primal feasible again (optimal), 2 = the node is infeasible, 4 = gave up.
__device__ __forceinline__ int wb_dual(const WbLds & L, const WbShape & S, int m, double tol, unsigned & pivots)
{
    const int cst = S.wcap - 1, live = S.n0 + m;
    for (int it = 0; it < WB_MAX_PIVOTS; it++) {
        double bv = 0.0; int bi = -1;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const double b = L.T[(size_t)i * S.wcap + cst];
            if (b < -tol && (bi < 0 || b < bv)) { bv = b; bi = i; }
        }
        wb_argmin(bv, bi, L.red_v, L.red_i);
        if (bi < 0) return 0;
        const int r = bi;
        double ev = 0.0; int ei = -1;
        for (int j = threadIdx.x; j < live; j += blockDim.x) {
            if (L.bv_row[j] >= 0) continue;
            const double a = L.T[(size_t)r * S.wcap + j];
            if (a < -tol) {
                const double q = L.obj[j] / -a;            // o_j >= 0 while the basis is dual feasible
                if (ei < 0 || q < ev) { ev = q; ei = j; }
            }
        }
        wb_argmin(ev, ei, L.red_v, L.red_i);
        if (ei < 0) return 2;
        wb_pivot(L, S, m, r, ei);
        pivots++;
    }
    return 4;
}
// Primal simplex (Dantzig's rule) from a primal-feasible basis: 0 optimal, 1 unbounded, 4 gave up.
__device__ __forceinline__ int wb_primal(const WbLds & L, const WbShape & S, int m, double tol, unsigned & pivots)
{
    const int cst = S.wcap - 1, live = S.n0 + m;
    for (int it = 0; it < WB_MAX_PIVOTS; it++) {
        double ev = 0.0; int ei = -1;
        for (int j = threadIdx.x; j < live; j += blockDim.x) {
            if (L.bv_row[j] >= 0) continue;
            const double o = L.obj[j];
            if (o < -tol && (ei < 0 || o < ev)) { ev = o; ei = j; }
        }
        wb_argmin(ev, ei, L.red_v, L.red_i);
        if (ei < 0) return 0;
        double rv = 0.0; int ri = -1;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const double a = L.T[(size_t)i * S.wcap + ei];
            if (a > tol) {
                const double q = L.T[(size_t)i * S.wcap + cst] / a;
                if (ri < 0 || q < rv) { rv = q; ri = i; }
            }
        }
        wb_argmin(rv, ri, L.red_v, L.red_i);
        if (ri < 0) return 1;
        wb_pivot(L, S, m, ri, ei);
        pivots++;
    }
    return 4;
}
// The z-row of objective c0 in the current basis: o_j = sum_i c_B(i) a_ij - c_j, z_c = sum_i c_B(i) b_i.
__device__ __forceinline__ void wb_price_out(const WbLds & L, const WbShape & S, int m)
{
    const int cst = S.wcap - 1, live = S.n0 + m;
    for (int j = threadIdx.x; j <= live; j += blockDim.x) {
        const int c = j < live ? j : cst;
        double s = 0.0;
        for (int i = 0; i < m; i++) { const int k = L.eq2bv[i]; if (k < S.n0) s += L.c0[k] * L.T[(size_t)i * S.wcap + c]; }
        L.obj[c] = (c < S.n0 && L.bv_row[c] < 0) ? s - L.c0[c] : (c == cst ? s : (L.bv_row[c] >= 0 ? 0.0 : s));
    }
    __syncthreads();
}

// Workspace of a tree (doubles): [0] incumbent value, [1] have, [2 .. 2 + n0) incumbent point, then depth_cap snapshots of
// snap_stride doubles: T rows (mcap x wcap), obj (wcap), node record (8), bv_row and eq2bv as ints behind them.
__global__ __launch_bounds__(WB_THREADS) void k_warm_mip_batch(int nb, const double * __restrict__ tgtf, const double * __restrict__ leq, WbShape S,
                                                                 int is_max, double * __restrict__ ws, int32_t * out_status, double * out_v,
                                                                 double * out_sol, unsigned * out_stats)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wb_lds_raw[];
    WbLds L;
    wb_carve(L, wb_lds_raw, S);
    const double tol = 1e-9, int_tol = 1e-6;
    const int n0 = S.n0, m0 = S.m0, wcap = S.wcap, cst = S.wcap - 1, cols = n0 + 1;
    const int tid = threadIdx.x;
    for (int tree = blockIdx.x; tree < nb; tree += gridDim.x) {
        double * W = ws + (size_t)tree * S.tree_stride;
        const double * A = leq + (size_t)tree * m0 * cols;
        const double * c = tgtf + (size_t)tree * cols;
        __syncthreads();
        // ---- the slack form
        for (int t = tid; t < S.mcap * wcap; t += blockDim.x) L.T[t] = 0.0;
        for (int j = tid; j < wcap; j += blockDim.x) { L.obj[j] = 0.0; L.bv_row[j] = -1; }
        __syncthreads();
        for (int t = tid; t < m0 * cols; t += blockDim.x) {
            const int i = t / cols, j = t - i * cols;
            L.T[(size_t)i * wcap + (j < n0 ? j : cst)] = A[t];
        }
        for (int i = tid; i < m0; i += blockDim.x) { L.T[(size_t)i * wcap + n0 + i] = 1.0; L.bv_row[n0 + i] = i; L.eq2bv[i] = n0 + i; }
        for (int j = tid; j < n0; j += blockDim.x) L.c0[j] = is_max ? c[j] : -c[j];
        if (tid == 0) { W[0] = 0.0; W[1] = 0.0; }
        __syncthreads();
        unsigned root_piv = 0, dual_piv = 0, nodes = 0, max_depth = 0;
        int m = m0, status = -1;
        // ---- the root relaxation: to a primal-feasible basis by the dual simplex under a zero objective (dual feasible by
        // construction) where the origin is not feasible, then the true objective priced out, then the primal simplex
        {
            double nb_ = 0.0; int ni = -1;
            for (int i = tid; i < m0; i += blockDim.x) if (L.T[(size_t)i * wcap + cst] < -tol) { ni = i; nb_ = -1.0; }
            wb_argmin(nb_, ni, L.red_v, L.red_i);
            int st1 = 0;
            if (ni >= 0) st1 = wb_dual(L, S, m, tol, root_piv);
            if (st1 == 2) status = XPG_IP_NO_PRI_FEASIBLE_SOL;
            else if (st1 == 4) status = XPG_ERR_UNSUPPORTED;
            else {
                wb_price_out(L, S, m);
                const int st2 = wb_primal(L, S, m, tol, root_piv);
                if (st2 == 1) status = XPG_IP_UNBOUND;
                else if (st2 == 4) status = XPG_ERR_UNSUPPORTED;
            }
        }
        // ---- the walk: either a freshly solved state is in LDS (have_state: consider it -- prune, new incumbent, or push), or the
        // node on top of the stack is taken up again (its next child). Every decision below is computed by every thread alike
        int depth = 0, height = 0;
        bool have_state = status == -1;
        while (status == -1) {
            if (have_state) {
                // consider: value, bound test, first fractional variable
                nodes++;
                if ((unsigned)depth > max_depth) max_depth = (unsigned)depth;
                const double value = L.obj[cst];
                const double best = W[0]; const bool have = W[1] != 0.0;
                bool pruned = have && value <= best + 1e-9 * fmax(1.0, fabs(best));
                double fv = 0.0; int fi = -1;
                if (!pruned) {
                    for (int j = tid; j < n0; j += blockDim.x) {
                        const int q = L.bv_row[j];
                        const double x = q >= 0 ? L.T[(size_t)q * wcap + cst] : 0.0;
                        const double f = x - floor(x + int_tol);
                        if (f > int_tol && (fi < 0 || j < fi)) { fi = j; fv = (double)j; }
                    }
                    wb_argmin(fv, fi, L.red_v, L.red_i);
                }
                __syncthreads();
                if (!pruned && fi < 0) {                     // integral: the new incumbent
                    for (int j = tid; j < n0; j += blockDim.x) { const int q = L.bv_row[j]; W[2 + j] = q >= 0 ? L.T[(size_t)q * wcap + cst] : 0.0; }
                    if (tid == 0) { W[0] = value; W[1] = 1.0; }
                    __threadfence_block();
                    __syncthreads();
                } else if (!pruned) {
                    if (height >= S.depth_cap || m >= S.mcap) { status = XPG_ERR_UNSUPPORTED; break; }
                    // push: snapshot of the solved state (live rows), the node record behind it
                    double * sn = W + 2 + n0 + (size_t)height * S.snap_stride;
                    for (int t = tid; t < m * wcap; t += blockDim.x) sn[t] = L.T[t];
                    double * so = sn + (size_t)S.mcap * wcap;
                    for (int j = tid; j < wcap; j += blockDim.x) so[j] = L.obj[j];
                    double * rec = so + wcap;
                    const int q = L.bv_row[fi];
                    const double xf = q >= 0 ? L.T[(size_t)q * wcap + cst] : 0.0;
                    if (tid == 0) { rec[0] = (double)fi; rec[1] = xf; rec[2] = value; rec[3] = (double)depth; rec[4] = 0.0; rec[5] = (double)m; }
                    int * si = (int *)(rec + 8);
                    for (int j = tid; j < wcap; j += blockDim.x) si[j] = L.bv_row[j];
                    for (int i = tid; i < m; i += blockDim.x) si[wcap + i] = L.eq2bv[i];
                    height++;
                    __threadfence_block();
                    __syncthreads();
                }
                have_state = false;
            }
            // ---- pop: the node on top of the stack, its next child
            if (height == 0) break;
            double * sn = W + 2 + n0 + (size_t)(height - 1) * S.snap_stride;
            double * so = sn + (size_t)S.mcap * wcap;
            double * rec = so + wcap;
            const int var = (int)rec[0], which = (int)rec[4], pm = (int)rec[5];
            const double val = rec[1], bound = rec[2];
            const int pdepth = (int)rec[3];
            const double best = W[0]; const bool have = W[1] != 0.0;
            __syncthreads();                                 // (everyone has read the record before thread 0 advances it)
            if (which > 1 || (have && bound <= best + 1e-9 * fmax(1.0, fabs(best)))) { height--; continue; }
            if (tid == 0) rec[4] = (double)(which + 1);
            // restore the parent's solved state
            const int * si = (const int *)(rec + 8);
            for (int t = tid; t < pm * wcap; t += blockDim.x) L.T[t] = sn[t];
            for (int j = tid; j < wcap; j += blockDim.x) { L.obj[j] = so[j]; L.bv_row[j] = si[j]; }
            for (int i = tid; i < pm; i += blockDim.x) L.eq2bv[i] = si[wcap + i];
            m = pm;
            __syncthreads();
            // the bound row in this basis: floor child x_j <= lo, ceiling child -x_j <= -(lo + 1)
            const double lo = floor(val + int_tol);
            const int sign = which == 0 ? 1 : -1;
            const double d = which == 0 ? lo : lo + 1.0;
            const int q = L.bv_row[var];
            const int live = n0 + m;
            for (int cc = tid; cc < wcap; cc += blockDim.x) {    // the whole row: what a deeper path left beyond `live` must go
                double v;
                if (cc == cst) v = q >= 0 ? sign * d - sign * L.T[(size_t)q * wcap + cst] : sign * d;
                else if (cc == live) v = 1.0;                   // its own slack
                else if (cc > live) v = 0.0;
                else if (cc == var) v = q >= 0 ? 0.0 : (double)sign;
                else v = q >= 0 ? -sign * L.T[(size_t)q * wcap + cc] : 0.0;
                L.T[(size_t)m * wcap + cc] = v;
            }
            if (tid == 0) { L.bv_row[n0 + m] = m; L.eq2bv[m] = n0 + m; L.obj[n0 + m] = 0.0; }
            __syncthreads();
            m += 1;
            const int ds = wb_dual(L, S, m, tol, dual_piv);
            if (ds == 4) { status = XPG_ERR_UNSUPPORTED; break; }
            if (ds == 2) continue;                           // infeasible child
            depth = pdepth + 1;
            have_state = true;
            if (nodes > 2000000u) { status = XPG_ERR_UNSUPPORTED; break; }
        }
        __syncthreads();
        if (status == -1) status = W[1] != 0.0 ? XPG_IP_SUCC : XPG_IP_NO_PRI_FEASIBLE_SOL;
        if (status == XPG_IP_SUCC) {
            for (int j = tid; j < n0; j += blockDim.x) out_sol[(size_t)tree * cols + j] = W[2 + j];
            if (tid == 0) { out_sol[(size_t)tree * cols + n0] = 1.0; out_v[tree] = is_max ? W[0] : -W[0]; }
        } else if (tid == 0) out_v[tree] = 0.0;
        if (tid == 0) {
            out_status[tree] = status;
            out_stats[4 * (size_t)tree] = nodes; out_stats[4 * (size_t)tree + 1] = dual_piv; out_stats[4 * (size_t)tree + 2] = root_piv; out_stats[4 * (size_t)tree + 3] = max_depth;
        }
    }
}

// Host side: shapes, workspace, launch, results. Trees are solved `chunk` at a time where the snapshot workspace of the
// whole batch would exceed 4 GB.
inline int warm_mip_batch(xpg_ctx * ctx, int nb, int is_max, const double * tgtf, const double * leq, int rows, int cols, int is_bin,
                          int32_t * out_status, double * out_v, double * out_sol, long long * out_stats)
{
    if (!ctx || nb < 0 || !tgtf || !leq || rows <= 0 || cols < 2 || !out_status || !out_v || !out_sol) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    WbShape S;
    S.n0 = cols - 1; S.m0 = rows;
    // bound rows a path may append: a 0-1 program (its x_j <= 1 rows are rows of the problem) branches on a variable at
    // most once per path; a general integer program gets what the one-tree form allows, as far as 64 KB of LDS go
    int depth = is_bin ? S.n0 + 2 : 2 * S.n0 + 8;
    for (;;) {
        S.depth_cap = depth; S.mcap = S.m0 + depth; S.wcap = S.n0 + S.mcap + 1;
        if (wb_lds_bytes(S) <= 64 * 1024 || depth <= 4) break;
        depth -= 2;
    }
    if (wb_lds_bytes(S) > 64 * 1024) { ctx->err = "warm-started branch and bound, batch form: the tableau of a tree does not fit 64 KB of LDS"; return XPG_ERR_UNSUPPORTED; }
    S.snap_stride = (unsigned long long)S.mcap * S.wcap + S.wcap + 8 + ((unsigned long long)S.wcap + S.mcap + 1) / 2 + 1;
    S.tree_stride = 2 + S.n0 + (unsigned long long)S.depth_cap * S.snap_stride;
    S.tree_stride = (S.tree_stride + 15) & ~15ull;
    const size_t lds = wb_lds_bytes(S);
    const size_t per_tree = (size_t)S.tree_stride * 8;
    int chunk = nb;
    while ((size_t)chunk * per_tree > ((size_t)4 << 30) && chunk > 64) chunk = (chunk + 1) / 2;
    const size_t in_l = (size_t)rows * cols * 8, in_t = (size_t)cols * 8;
    char * dev = nullptr;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_leq = 0, o_tg = o_leq + up(in_l * chunk), o_sol = o_tg + up(in_t * chunk), o_v = o_sol + up(in_t * chunk),
                 o_st = o_v + up((size_t)chunk * 8), o_stats = o_st + up((size_t)chunk * 4), o_ws = o_stats + up((size_t)chunk * 16),
                 total = o_ws + (size_t)chunk * per_tree;
    if (hipMalloc((void **)&dev, total) != hipSuccess) { (void)hipGetLastError(); ctx->err = "hipMalloc(warm branch and bound workspace)"; return XPG_ERR_ALLOC; }
    hipError_t e = lds_limit((const void *)k_warm_mip_batch, ctx->device, lds);
    long long tot[4] = {0, 0, 0, 0};
    std::vector<unsigned> hstats;
    for (int lo = 0; lo < nb && e == hipSuccess; lo += chunk) {
        const int n = nb - lo < chunk ? nb - lo : chunk;
        e = hipMemcpyAsync(dev + o_leq, leq + (size_t)lo * rows * cols, in_l * n, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dev + o_tg, tgtf + (size_t)lo * cols, in_t * n, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_warm_mip_batch, dim3(n), dim3(WB_THREADS), lds, ctx->stream, n, (const double *)(dev + o_tg), (const double *)(dev + o_leq), S,
                           is_max ? 1 : 0, (double *)(dev + o_ws), (int32_t *)(dev + o_st), (double *)(dev + o_v), (double *)(dev + o_sol),
                           (unsigned *)(dev + o_stats));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out_status + lo, dev + o_st, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_v + lo, dev + o_v, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out_sol + (size_t)lo * cols, dev + o_sol, in_t * n, hipMemcpyDeviceToHost, ctx->stream);
        hstats.resize((size_t)n * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(hstats.data(), dev + o_stats, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) break;
        for (int b = 0; b < n; b++) {
            tot[0] += hstats[4 * (size_t)b]; tot[1] += hstats[4 * (size_t)b + 1]; tot[2] += hstats[4 * (size_t)b + 2];
            if ((long long)hstats[4 * (size_t)b + 3] > tot[3]) tot[3] = hstats[4 * (size_t)b + 3];
        }
    }
    (void)hipFree(dev);
    if (e != hipSuccess) { ctx->err = std::string("warm branch and bound, batch form: ") + hipGetErrorString(e); return XPG_ERR_HIP; }
    if (out_stats) for (int k = 0; k < 4; k++) out_stats[k] = tot[k];
    return 0;
}

} // namespace xpg
