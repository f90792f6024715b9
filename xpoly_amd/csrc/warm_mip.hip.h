// OPT-IN, NON-PARITY: branch and bound that re-optimises every node from its parent's
// final tableau with the DUAL simplex, instead of a fresh SIX per node (src/com/lpsol.h:2440-2448 builds a new
// SIX and solves the grown problem from the slack form each time). fp64, x >= 0, inequalities only.
//
// The tableau stays in HBM for the whole tree. A child is its parent's solved state plus ONE bound row
//     x_j <= floor(x_j*)      or      -x_j <= -ceil(x_j*)
// written in the parent's basis (k_warm_new_row: if x_j is basic in row q, the row is -/+ row q with the x_j
// coefficient cancelled, so the new slack is basic with a negative constant), after which the basis is still dual
// feasible and a few dual pivots restore primal feasibility: k_dual_pick chooses the leaving row (most negative
// constant) and the entering column (least c_j / a_rj over a_rj < 0), and the pivot itself is the library's own
// K1 (k_prep + k_update_f64, the kernels every other loop uses). Depth-first, floor child first; a node is kept
// as a snapshot of its solved state (tableau, objective row, basis) and restored by device-to-device copies.
// This is a different pivoting rule and a sane branch and bound (best incumbent, bounding by the relaxation),
// so results are checked against the mathematics (scipy's HiGHS milp), not against the reference's depth-first
// walk, whose answers depend on its fork counter (lpsol.h:2474-2497).
#pragma once
#include <cmath>
#include <vector>
#include "lp_host.hip.h"

namespace xpg {

enum { ST_DUAL_DONE = -1002 };

// The new row (index v.m) for the bound sign * x_j <= sign * d, in the current basis. Reads only.
__global__ void k_warm_new_row(LpView<F64> v, int j, int sign, double d)
{
    const int rhs = v.rhs, m = v.m;
    const int q = v.bv[j] ? v.bv2eq[j] : -1;
    const double * tab = (const double *)v.tab;
    double * nr = (double *)v.tab + (size_t)m * v.ld;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= rhs + 1; k += gridDim.x * blockDim.x) {
        double val;
        if (k == rhs + 1) val = q >= 0 ? sign * d - sign * tab[(size_t)q * v.ld + rhs] : sign * d;     // the constant
        else if (k == rhs) val = 1.0;                                                                  // the new slack
        else if (k == j) val = q >= 0 ? 0.0 : (double)sign;
        else val = q >= 0 ? -sign * tab[(size_t)q * v.ld + k] : 0.0;
        nr[k] = val;
    }
}
// Makes room for the new slack's column in front of the constant column (rows < m and the objective row) and
// enters the slack into the basis.
__global__ void k_warm_shift(LpView<F64> v)
{
    const int rhs = v.rhs, m = v.m;
    double * tab = (double *)v.tab;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= m; i += gridDim.x * blockDim.x) {
        if (i < m) { tab[(size_t)i * v.ld + rhs + 1] = tab[(size_t)i * v.ld + rhs]; tab[(size_t)i * v.ld + rhs] = 0.0; }
        else {
            double * o = (double *)v.obj;
            o[rhs + 1] = o[rhs]; o[rhs] = 0.0;
            v.nv[rhs] = 0; v.bv[rhs] = 1; v.bv2eq[rhs] = m; v.eq2bv[m] = rhs;
        }
    }
}
__global__ void k_warm_begin(LoopState * st) { if (threadIdx.x == 0 && blockIdx.x == 0) { st->status = ST_RUNNING; st->row = -1; st->infeasible = 0; } }

// One dual simplex choice: leaving row = the most negative constant (lowest row on ties), entering column = the
// least c_j / a_rj over nonbasic j with a_rj < -tol (lowest column on ties). Leaves the pivot in LoopState for
// k_prep / k_update, or a final status: ST_DUAL_DONE (primal feasible again: optimal) or 2 (no entering column:
// the node is infeasible).
__global__ __launch_bounds__(1024) void k_dual_pick(LpView<F64> v, double tol)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_raw[16 * sizeof(Cand<F64>)];
    Cand<F64> * sh = (Cand<F64> *)sh_raw;
    LoopState * st = v.st;
    if (st->status != ST_RUNNING) return;
    const double * tab = (const double *)v.tab;
    Cand<F64> best; best.q = zero<F64>(); best.idx = INT_MAX;
    for (int i = threadIdx.x; i < v.m; i += blockDim.x) {
        const double b = tab[(size_t)i * v.ld + v.rhs];
        if (b < -tol) { Cand<F64> c; c.q = F64(b); c.idx = i; best = better(best, c); }
    }
    best = block_argmin(best, sh);
    if (best.idx == INT_MAX) { if (threadIdx.x == 0) { st->status = ST_DUAL_DONE; st->row = -1; } return; }
    const int r = best.idx;
    __syncthreads();
    Cand<F64> e; e.q = zero<F64>(); e.idx = INT_MAX;
    for (int j = threadIdx.x; j < v.rhs; j += blockDim.x) {
        if (!v.nv[j]) continue;
        const double a = tab[(size_t)r * v.ld + j];
        if (a < -tol) { Cand<F64> c; c.q = F64(((const double *)v.obj)[j] / a); c.idx = j; e = better(e, c); }
    }
    e = block_argmin(e, sh);
    if (threadIdx.x == 0) {
        if (e.idx == INT_MAX) { st->status = 2; st->row = -1; }
        else {
            st->row = r; st->col = e.idx; st->leave = v.eq2bv[r];
            st->cnv_bits = to_bits(v.obj[e.idx]);
            st->piv_bits = to_bits(v.tab[(size_t)r * v.ld + e.idx]);
        }
    }
}

struct WarmStats { long long nodes, dual_pivots, root_pivots, max_depth; };

class WarmMip {
    xpg_ctx * ctx;
    Lp<F64> L;
    int n0, m0;
    struct Snap { double * tab; double * obj; uint8_t * nv; uint8_t * bv; int * bv2eq; int * eq2bv; int m, W, rhs; };
    std::vector<Snap> pool;            // every snapshot ever allocated (freed at the end)
    std::vector<int> free_slots;
    double tol, int_tol;

    size_t tab_bytes() const { return (size_t)L.row_cap * L.v.ld * 8; }
    int new_slot()
    {
        if (!free_slots.empty()) { const int s = free_slots.back(); free_slots.pop_back(); return s; }
        Snap S;
        if (hipMalloc((void **)&S.tab, tab_bytes()) != hipSuccess || hipMalloc((void **)&S.obj, (size_t)L.v.ld * 8) != hipSuccess ||
            hipMalloc((void **)&S.nv, L.v.ld) != hipSuccess || hipMalloc((void **)&S.bv, L.v.ld) != hipSuccess ||
            hipMalloc((void **)&S.bv2eq, (size_t)L.v.ld * 4) != hipSuccess || hipMalloc((void **)&S.eq2bv, (size_t)round_up(L.row_cap, 16) * 4) != hipSuccess)
            return -1;
        pool.push_back(S);
        return (int)pool.size() - 1;
    }
    void copy_state(bool save, int slot)
    {
        Snap & S = pool[slot];
        hipStream_t s = ctx->stream;
        const hipMemcpyKind k = hipMemcpyDeviceToDevice;
        if (save) { S.m = L.v.m; S.W = L.v.W; S.rhs = L.v.rhs; }
        else { L.v.m = S.m; L.v.W = S.W; L.v.rhs = S.rhs; }
        const size_t rows = (size_t)(save ? L.v.m : S.m) * L.v.ld * 8;
#define XPG_CP(dev_, snap_, bytes_) (void)hipMemcpyAsync(save ? (void *)(snap_) : (void *)(dev_), save ? (const void *)(dev_) : (const void *)(snap_), bytes_, k, s)
        XPG_CP(L.v.tab, S.tab, rows);
        XPG_CP(L.v.obj, S.obj, (size_t)L.v.ld * 8);
        XPG_CP(L.v.nv, S.nv, (size_t)L.v.ld);
        XPG_CP(L.v.bv, S.bv, (size_t)L.v.ld);
        XPG_CP(L.v.bv2eq, S.bv2eq, (size_t)L.v.ld * 4);
        XPG_CP(L.v.eq2bv, S.eq2bv, (size_t)round_up(L.row_cap, 16) * 4);
#undef XPG_CP
    }
    // value and x[0 .. n0) of the state now in L
    int read_point(double & value, std::vector<double> & x)
    {
        hipLaunchKernelGGL((k_solution<F64>), dim3(64), dim3(256), 0, ctx->stream, L.v);
        x.resize((size_t)n0);
        XPG_HIP(ctx, hipMemcpyAsync(x.data(), L.v.x, (size_t)n0 * 8, hipMemcpyDeviceToHost, ctx->stream));
        XPG_HIP(ctx, hipMemcpyAsync(&value, (const double *)L.v.obj + L.v.rhs, 8, hipMemcpyDeviceToHost, ctx->stream));
        XPG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    // appends the bound and re-optimises; returns 0 (optimal), 2 (infeasible) or an error
    int branch(int j, int sign, double d, WarmStats & S)
    {
        if (L.v.m >= L.row_cap) return XPG_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_warm_new_row, dim3(8), dim3(256), 0, ctx->stream, L.v, j, sign, d);
        hipLaunchKernelGGL(k_warm_shift, dim3(8), dim3(256), 0, ctx->stream, L.v);
        L.v.m += 1; L.v.W += 1; L.v.rhs += 1;
        hipLaunchKernelGGL(k_warm_begin, dim3(1), dim3(64), 0, ctx->stream, L.v.st);
        LoopState hs;
        for (int round = 0; round < 4096; round++) {
            for (int k = 0; k < 6; k++) {
                hipLaunchKernelGGL(k_dual_pick, dim3(1), dim3(1024), 0, ctx->stream, L.v, tol);
                L.queue_pivot(1, 1);
            }
            int rc = L.read_state(&hs);
            if (rc) return rc;
            if (hs.status != ST_RUNNING) break;
        }
        if (hs.status == ST_RUNNING) return XPG_SIX_TIME_OUT;
        return hs.status == ST_DUAL_DONE ? 0 : hs.status;
    }

public:
    WarmMip(xpg_ctx * c) : ctx(c), n0(0), m0(0), tol(1e-9), int_tol(1e-6) {}
    ~WarmMip()
    {
        for (Snap & S : pool) { (void)hipFree(S.tab); (void)hipFree(S.obj); (void)hipFree(S.nv); (void)hipFree(S.bv); (void)hipFree(S.bv2eq); (void)hipFree(S.eq2bv); }
    }

    // maximise tgtf . x subject to leq (A | b), x >= 0, x integer (0/1 when is_bin). Returns XPG_IP_*.
    int solve(const double * tgtf, const double * leq, int rows, int cols, bool is_bin, double * out_v, double * out_sol, WarmStats & S)
    {
        n0 = cols - 1; m0 = rows;
        S.nodes = S.dual_pivots = S.root_pivots = S.max_depth = 0;
        // room for one bound row per level of a depth-first path: two per variable is more than any path uses
        const int extra = 2 * n0 + 8;
        L.ctx = ctx; L.kind = 0;
        int rc = L.create(leq, rows, cols, tgtf, 0, 0, 0, extra);
        if (rc) return rc;
        L.set_options(1, 1e-9);                             // Dantzig pricing, tolerant feasibility for the root
        int st = L.two_stage(0xFFFFFFFFu);
        if (st < 0) return st;
        LoopState hs;
        if ((rc = L.read_state(&hs))) return rc;
        S.root_pivots = hs.total_pivots;
        if (st == XPG_SIX_UNBOUND) return XPG_IP_UNBOUND;
        if (st != XPG_SIX_SUCC) return XPG_IP_NO_PRI_FEASIBLE_SOL;
        unsigned pivots_before = hs.total_pivots;
        struct Node { int slot; int var; double val; double bound; int depth; int next; };   // next: 0 floor pending, 1 ceiling pending
        std::vector<Node> stack;
        bool have = false; double best = 0.0; std::vector<double> best_x, x;
        double value;
        // evaluates the state in L: incumbent / prune / push
        auto consider = [&](int depth) -> int {
            int rc2 = read_point(value, x);
            if (rc2) return rc2;
            S.nodes++;
            if (depth > S.max_depth) S.max_depth = depth;
            if (have && value <= best + 1e-9 * std::fmax(1.0, std::fabs(best))) return 0;     // bounded by the incumbent
            int frac = -1;
            for (int j = 0; j < n0; j++) {
                const double f = x[(size_t)j] - std::floor(x[(size_t)j] + int_tol);
                if (f > int_tol) { frac = j; break; }
            }
            if (frac < 0) { have = true; best = value; best_x = x; return 0; }
            const int slot = new_slot();
            if (slot < 0) return XPG_ERR_ALLOC;
            copy_state(true, slot);
            Node N; N.slot = slot; N.var = frac; N.val = x[(size_t)frac]; N.bound = value; N.depth = depth; N.next = 0;
            stack.push_back(N);
            return 0;
        };
        if ((rc = consider(0))) return rc;
        while (!stack.empty()) {
            Node & N = stack.back();
            if (N.next > 1 || (have && N.bound <= best + 1e-9 * std::fmax(1.0, std::fabs(best)))) {
                free_slots.push_back(N.slot);
                stack.pop_back();
                continue;
            }
            const int which = N.next++;
            const int var = N.var, depth = N.depth, slot = N.slot;
            const double lo = std::floor(N.val + int_tol);
            copy_state(false, slot);
            // floor child: x_j <= lo; ceiling child: -x_j <= -(lo + 1)   (0-1 programs: x_j <= 0 / x_j >= 1 the same way)
            (void)is_bin;
            const int brc = which == 0 ? branch(var, +1, lo, S) : branch(var, -1, lo + 1.0, S);
            if (brc < 0 || brc == XPG_SIX_TIME_OUT) return brc < 0 ? brc : XPG_ERR_HIP;
            if (brc != 0) continue;                          // infeasible child
            if ((rc = consider(depth + 1))) return rc;
            if (S.nodes > 2000000) return XPG_ERR_UNSUPPORTED;
        }
        if ((rc = L.read_state(&hs))) return rc;
        S.dual_pivots = (long long)hs.total_pivots - (long long)pivots_before;
        if (!have) return XPG_IP_NO_PRI_FEASIBLE_SOL;
        *out_v = best;
        if (out_sol) { for (int j = 0; j < n0; j++) out_sol[j] = best_x[(size_t)j]; out_sol[n0] = 1.0; }
        return XPG_IP_SUCC;
    }
};

} // namespace xpg
