// Host controller of xpoly's branch-and-bound MIP<Mat,T> (src/com/lpsol.h:2087-2702)
// and of Lineq::has_solution (src/com/linsys.cpp:830-906). The tree walk is the
// reference's depth-first recursion -- its results depend on DFS order through the
// shared fork_count row and the incumbent (lpsol.h:2474-2497) -- and every node's
// LP relaxation is a from-scratch SIX solve with max_iter = 10000 (lpsol.h:2441)
// executed on the GPU (six_solve: one LDS-resident launch at these sizes).
#pragma once
#include <vector>
#include "six_host.hip.h"

namespace xpg {

template <class S> struct MipProblem {
    int cols;
    std::vector<S> tgtf, vc, eq, leq;      // flat row-major; vc has cols-1 rows
    int eq_rows, leq_rows;
};

template <class S> struct MipHost {
    xpg_ctx * ctx;
    int kind;
    bool have_best;
    std::vector<S> best_sol;
    S best_v;
    const uint8_t * allow_rational;         // 1 x cols, or null (lpsol.h:2369-2393)
    int rhs0;
    long nodes;

    MipHost(xpg_ctx * c, int k, const uint8_t * allow, int rhs)
        : ctx(c), kind(k), have_best(false), best_v(zero<S>()), allow_rational(allow), rhs0(rhs), nodes(0) {}

    // MIP::is_satisfying (lpsol.h:2364-2408); `col` is the first offending entry.
    bool satisfied(std::vector<S> & s, bool is_bin, int & col) const
    {
        for (size_t j = 0; j < s.size(); j++) {
            if (allow_rational || is_bin) reduce(s[j]);
            if (allow_rational) {
                if (allow_rational[j]) continue;
                if (!is_int(s[j])) { col = (int)j; return false; }
                if (is_bin && ne(s[j], zero<S>()) && ne(s[j], one<S>())) { col = (int)j; return false; }
            } else if (is_bin) {
                if (ne(s[j], zero<S>()) && ne(s[j], one<S>())) { col = (int)j; return false; }
            } else if (!is_int(s[j])) { col = (int)j; return false; }     // {R,Float}Mat::is_imat
        }
        return true;
    }

    void remember(const std::vector<S> & s, S v, bool is_max)
    {
        if (!have_best || (is_max ? lt(best_v, v) : gt(best_v, v))) { best_sol = s; best_v = v; have_best = true; }
    }

    static void add_row(std::vector<S> & rows, int & nrows, int cols, int col, S coef, int rhs, S b)
    {
        rows.resize((size_t)(nrows + 1) * cols, zero<S>());
        for (int j = 0; j < cols; j++) rows[(size_t)nrows * cols + j] = zero<S>();
        rows[(size_t)nrows * cols + col] = coef;
        rows[(size_t)nrows * cols + rhs] = b;
        nrows++;
    }

    // MIP::RecusivePart (lpsol.h:2427-2612).
    int node(const MipProblem<S> & Q, bool is_max, bool is_bin, std::vector<int> & forks, S & v, std::vector<S> & sol)
    {
        nodes++;
        std::vector<S> out_sol(Q.cols);
        S out_v = zero<S>();
        int st = six_solve<S>(ctx, kind, is_max, Q.tgtf.data(), Q.vc.data(), Q.cols - 1,
                              Q.eq_rows ? Q.eq.data() : (const S *)0, Q.eq_rows,
                              Q.leq_rows ? Q.leq.data() : (const S *)0, Q.leq_rows, Q.cols, 10000u, &out_v,
                              out_sol.data());
        v = out_v;
        if (st < 0) return st;
        if (st != XPG_SIX_SUCC) {
            if (st == XPG_SIX_UNBOUND) return XPG_IP_UNBOUND;
            if (st == XPG_SIX_TIME_OUT) return XPG_ERR_REF_UNDEFINED;         // UNREACH() in the reference
            return XPG_IP_NO_PRI_FEASIBLE_SOL;
        }
        sol = out_sol;
        int col = 0;
        if (satisfied(sol, is_bin, col)) return XPG_IP_SUCC;
        if (have_best && (is_max ? le(v, best_v) : ge(v, best_v))) return XPG_IP_NO_BETTER_THAN_BEST_SOL;
        if (forks[col] >= 1) return XPG_IP_NO_PRI_FEASIBLE_SOL;              // lpsol.h:2486-2496
        forks[col]++;
        int lo = 0, hi = 1;
        MipProblem<S> L = Q;                                                  // floor branch, lpsol.h:2503-2521
        if (is_bin) add_row(L.eq, L.eq_rows, Q.cols, col, one<S>(), rhs0, S::from_int(lo));
        else {
            if (!int_cast_ok(sol[col])) return XPG_ERR_REF_UNDEFINED;
            lo = to_int(sol[col]); hi = lo + 1;
            add_row(L.leq, L.leq_rows, Q.cols, col, one<S>(), rhs0, S::from_int(lo));
        }
        std::vector<S> kept_sol; S kept_v = zero<S>(); bool kept = false;
        st = node(L, is_max, is_bin, forks, v, sol);
        if (st < 0) return st;
        if (st == XPG_IP_SUCC) { kept_sol = sol; kept_v = v; kept = true; remember(sol, v, is_max); }
        MipProblem<S> H = Q;                                                  // ceiling branch, lpsol.h:2545-2560
        if (is_bin) add_row(H.eq, H.eq_rows, Q.cols, col, one<S>(), rhs0, S::from_int(hi));
        else add_row(H.leq, H.leq_rows, Q.cols, col, minus_one<S>(), rhs0, S::from_int(-hi));
        st = node(H, is_max, is_bin, forks, v, sol);
        if (st < 0) return st;
        if (st == XPG_IP_SUCC) {                                              // lpsol.h:2563-2592
            if (kept && (is_max ? gt(kept_v, v) : lt(kept_v, v))) { v = kept_v; sol = kept_sol; }
            remember(sol, v, is_max);
            return XPG_IP_SUCC;
        }
        if (kept) { v = kept_v; sol = kept_sol; remember(sol, v, is_max); return XPG_IP_SUCC; }
        return st;
    }

    static bool int_cast_ok(F64) { return true; }
    static bool int_cast_ok(R32 a) { return a.den != 0; }
};

template <class S>
MipProblem<S> make_problem(const S * tgtf, const S * vc, int vc_rows, const S * eqs, int eq_rows, const S * leq,
                           int leq_rows, int cols)
{
    MipProblem<S> Q;
    Q.cols = cols; Q.eq_rows = eq_rows; Q.leq_rows = leq_rows;
    Q.tgtf.assign(tgtf, tgtf + cols);
    Q.vc.assign(vc, vc + (size_t)vc_rows * cols);
    if (eq_rows) Q.eq.assign(eqs, eqs + (size_t)eq_rows * cols);
    if (leq_rows) Q.leq.assign(leq, leq + (size_t)leq_rows * cols);
    return Q;
}

// MIP::maxm / minm (lpsol.h:2636-2657, :2681-2702).
template <class S>
int mip_solve(xpg_ctx * ctx, int kind, bool is_max, bool is_bin, const S * tgtf, const S * vc, int vc_rows,
              const S * eqs, int eq_rows, const S * leq, int leq_rows, int cols, const uint8_t * allow_rational,
              S * out_v, S * out_sol, long * out_nodes)
{
    if (!ctx || !tgtf || !vc || !out_v || cols < 2 || vc_rows != cols - 1 || eq_rows < 0 || leq_rows < 0 ||
        (eq_rows == 0 && leq_rows == 0) || (eq_rows > 0 && !eqs) || (leq_rows > 0 && !leq))
        return XPG_ERR_SHAPE;
    MipProblem<S> Q = make_problem(tgtf, vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols);
    MipHost<S> M(ctx, kind, allow_rational, cols - 1);
    std::vector<int> forks(cols, 0);
    S v = zero<S>();
    std::vector<S> sol;
    int st = M.node(Q, is_max, is_bin, forks, v, sol);
    *out_v = v;
    if (st == XPG_IP_SUCC && out_sol && (int)sol.size() == cols)
        for (int j = 0; j < cols; j++) out_sol[j] = sol[j];
    if (out_nodes) *out_nodes = M.nodes;
    return st;
}

// Lineq::has_solution (linsys.cpp:830-906): objective sum(x) with unconstrained columns
// zeroed (SIX::reviseTargetFunc, lpsol.h:2053-2074), maxm then minm; success, or an
// unbounded answer when a unique solution is not demanded, means "has a solution".
inline int has_solution(xpg_ctx * ctx, const R32 * leq, int leq_rows, const R32 * eqs, int eq_rows, const R32 * vc,
                        int vc_rows, int cols, int rhs, bool is_int, bool is_unique)
{
    if (!ctx || !vc || cols < 2 || rhs != cols - 1 || vc_rows != rhs) return XPG_ERR_SHAPE;
    if (leq_rows == 0 && eq_rows == 0) return 0;
    if (leq_rows == 0) return XPG_ERR_REF_UNDEFINED;      // the reference sizes tgtf from leq (linsys.cpp:851)
    std::vector<R32> tgtf(cols, R32(0, 1));
    for (int j = 0; j < rhs; j++) {
        bool nz = false;
        for (int i = 0; i < leq_rows && !nz; i++) nz = !eq(leq[(size_t)i * cols + j], R32(0, 1));
        for (int i = 0; i < eq_rows && !nz; i++) nz = !eq(eqs[(size_t)i * cols + j], R32(0, 1));
        tgtf[j] = nz ? R32(1, 1) : R32(0, 1);
    }
    R32 v; std::vector<R32> sol(cols);
    for (int pass = 0; pass < 2; pass++) {
        int st = is_int
            ? mip_solve<R32>(ctx, 1, pass == 0, false, tgtf.data(), vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols,
                             (const uint8_t *)0, &v, sol.data(), (long *)0)
            : six_solve<R32>(ctx, 1, pass == 0, tgtf.data(), vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols,
                             0xFFFFFFFFu, &v, sol.data());
        if (st < 0) return st;
        if (st == 0) return 1;
        if (!is_unique && st == 1) return 1;
    }
    return 0;
}

} // namespace xpg
