// TEST INFRASTRUCTURE -- C ABI of the CPU restatement ("port"), entry points
// mirror oracle/ref_driver.cpp one-to-one with the prefix orc_ so the tests can
// drive either checker through the same Python binding (oracle/checker.py).
// Built by oracle/Makefile into oracle/_build/libxpoly_oracle.so. Never linked
// into or loaded by the product.
#include <string.h>
#include "oracle.h"
#include "oracle_scalar.h"
#include "oracle_lp.h"
#include "oracle_lineq.h"

using namespace orc;

namespace {

template <class S> struct Raw;
template <> struct Raw<F64> {
    static F64 get(const void * p, size_t i) { return F64(((const double*)p)[i]); }
    static void put(void * p, size_t i, F64 v) { ((double*)p)[i] = v.v; }
};
template <> struct Raw<R32> {
    static R32 get(const void * p, size_t i)
    { const int32_t * q = (const int32_t*)p + 2 * i; return R32(q[0], q[1]); }
    static void put(void * p, size_t i, R32 v)
    { int32_t * q = (int32_t*)p + 2 * i; q[0] = v.num; q[1] = v.den; }
};

template <class S> Mat<S> load(const void * p, int rows, int cols)
{
    if (rows <= 0 || !p) return Mat<S>();
    Mat<S> m(rows, cols);
    for (size_t k = 0; k < m.a.size(); k++) m.a[k] = Raw<S>::get(p, k);
    return m;
}
template <class S> void store(const Mat<S> & m, void * p)
{
    for (size_t k = 0; k < m.a.size(); k++) Raw<S>::put(p, k, m.a[k]);
}
template <class S> void store(const std::vector<S> & v, void * p)
{
    for (size_t k = 0; k < v.size(); k++) Raw<S>::put(p, k, v[k]);
}

template <class S>
Problem<S> load_problem(const void * tgtf, const void * vc, int vc_rows, const void * eq,
                        int eq_rows, const void * leq, int leq_rows, int cols)
{
    Problem<S> Q;
    Q.cols = cols;
    Q.tgtf.resize(cols);
    for (int j = 0; j < cols; j++) Q.tgtf[j] = Raw<S>::get(tgtf, j);
    Q.vc = load<S>(vc, vc_rows, cols);
    Q.eq = load<S>(eq, eq_rows, cols);
    Q.leq = load<S>(leq, leq_rows, cols);
    return Q;
}

template <class S>
int t_six_solve(int is_max, const void * tgtf, const void * vc, int vc_rows, const void * eq,
                int eq_rows, const void * leq, int leq_rows, int cols, unsigned max_iter,
                void * out_v, void * out_sol)
{
    Problem<S> Q = load_problem<S>(tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows, cols);
    S v; std::vector<S> sol;
    int st = is_max ? six_maxm(Q, max_iter, v, sol) : six_minm(Q, max_iter, v, sol);
    Raw<S>::put(out_v, 0, v);
    if (st == SIX_SUCC) store(sol, out_sol);
    return st;
}

template <class S>
int t_two_stage(const void * leq, int m, int cols, const void * vc, const void * tgtf,
                unsigned max_iter, void * out_tab, int * out_rows, int * out_cols,
                void * out_tgtf, uint8_t * out_nv, uint8_t * out_bv, int32_t * out_bv2eq,
                int32_t * out_eq2bv, int * out_rhs, void * out_maxv, void * out_sol,
                int32_t * out_trace, int trace_cap, int * out_trace_len)
{
    Slack<S> P;
    P.eq = load<S>(leq, m, cols);
    P.rhs = cols - 1;
    P.obj.resize(cols);
    for (int j = 0; j < cols; j++) P.obj[j] = Raw<S>::get(tgtf, j);
    Mat<S> V = load<S>(vc, cols - 1, cols);
    P.vcd.resize(cols - 1); P.vcr.resize(cols - 1);
    for (int i = 0; i < cols - 1; i++) { P.vcd[i] = V.at(i, i); P.vcr[i] = V.at(i, cols - 1); }
    S best = S(0); std::vector<S> x;
    int st = two_stage(P, max_iter, best, x);
    *out_rows = P.eq.r; *out_cols = P.eq.c; *out_rhs = P.rhs;
    store(P.eq, out_tab);
    store(P.obj, out_tgtf);
    for (int i = 0; i < P.rhs && i < (int)P.nv.size(); i++) {
        out_nv[i] = P.nv[i]; out_bv[i] = P.bv[i]; out_bv2eq[i] = P.bv2eq[i];
    }
    for (size_t i = 0; i < P.eq2bv.size(); i++) out_eq2bv[i] = P.eq2bv[i];
    Raw<S>::put(out_maxv, 0, best);
    if (out_sol && !x.empty()) store(x, out_sol);
    if (out_trace_len) {
        int n = (int)P.trace.size() < trace_cap ? (int)P.trace.size() : trace_cap;
        for (int i = 0; i < n; i++) out_trace[i] = P.trace[i];
        *out_trace_len = (int)P.trace.size();
    }
    return st;
}

template <class S>
int t_mip_solve(int is_max, int is_bin, const void * tgtf, const void * vc, int vc_rows,
                const void * eq, int eq_rows, const void * leq, int leq_rows, int cols,
                const uint8_t * rat_ind, void * out_v, void * out_sol, long * nodes, int * max_leq_rows = 0)
{
    // (max_leq_rows, when given, is three ints: the most inequality rows of a node LP, the deepest recursion level, the node
    // LPs on the sequential chain when ceiling children are solved ahead)
    Problem<S> Q = load_problem<S>(tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows, cols);
    S v; std::vector<S> sol;
    int st = mip_solve(Q, is_max != 0, is_bin != 0, rat_ind, v, sol, nodes, max_leq_rows, max_leq_rows ? max_leq_rows + 1 : 0);
    Raw<S>::put(out_v, 0, v);
    if (st == IP_SUCC && (int)sol.size() == cols) store(sol, out_sol);
    return st;
}

} // namespace

extern "C" {

int orc_six_solve(int kind, int is_max, const void * tgtf, const void * vc, int vc_rows,
                  const void * eq, int eq_rows, const void * leq, int leq_rows, int cols,
                  unsigned max_iter, void * out_v, void * out_sol)
{
    if (kind == 0)
        return t_six_solve<F64>(is_max, tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows, cols,
                                max_iter, out_v, out_sol);
    return t_six_solve<R32>(is_max, tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows, cols,
                            max_iter, out_v, out_sol);
}

int orc_two_stage(int kind, const void * leq, int m, int cols, const void * vc,
                  const void * tgtf, unsigned max_iter, void * out_tab, int * out_rows,
                  int * out_cols, void * out_tgtf, uint8_t * out_nvset, uint8_t * out_bvset,
                  int32_t * out_bv2eq, int32_t * out_eq2bv, int * out_rhs, void * out_maxv,
                  void * out_sol)
{
    if (kind == 0)
        return t_two_stage<F64>(leq, m, cols, vc, tgtf, max_iter, out_tab, out_rows, out_cols,
            out_tgtf, out_nvset, out_bvset, out_bv2eq, out_eq2bv, out_rhs, out_maxv, out_sol,
            0, 0, 0);
    return t_two_stage<R32>(leq, m, cols, vc, tgtf, max_iter, out_tab, out_rows, out_cols,
        out_tgtf, out_nvset, out_bvset, out_bv2eq, out_eq2bv, out_rhs, out_maxv, out_sol,
        0, 0, 0);
}

// Same as orc_two_stage but also returns the (entering, leaving) pivot pairs.
int orc_two_stage_trace(int kind, const void * leq, int m, int cols, const void * vc,
                        const void * tgtf, unsigned max_iter, void * out_tab, int * out_rows,
                        int * out_cols, void * out_tgtf, uint8_t * out_nvset,
                        uint8_t * out_bvset, int32_t * out_bv2eq, int32_t * out_eq2bv,
                        int * out_rhs, void * out_maxv, void * out_sol, int32_t * out_trace,
                        int trace_cap, int * out_trace_len)
{
    if (kind == 0)
        return t_two_stage<F64>(leq, m, cols, vc, tgtf, max_iter, out_tab, out_rows, out_cols,
            out_tgtf, out_nvset, out_bvset, out_bv2eq, out_eq2bv, out_rhs, out_maxv, out_sol,
            out_trace, trace_cap, out_trace_len);
    return t_two_stage<R32>(leq, m, cols, vc, tgtf, max_iter, out_tab, out_rows, out_cols,
        out_tgtf, out_nvset, out_bvset, out_bv2eq, out_eq2bv, out_rhs, out_maxv, out_sol,
        out_trace, trace_cap, out_trace_len);
}

int orc_mip_solve(int kind, int is_max, int is_bin, const void * tgtf, const void * vc,
                  int vc_rows, const void * eq, int eq_rows, const void * leq, int leq_rows,
                  int cols, const uint8_t * rat_ind, void * out_v, void * out_sol)
{
    if (kind == 0)
        return t_mip_solve<F64>(is_max, is_bin, tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows,
                                cols, rat_ind, out_v, out_sol, 0);
    return t_mip_solve<R32>(is_max, is_bin, tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows,
                            cols, rat_ind, out_v, out_sol, 0);
}

// the same solve with tree statistics (how many nodes, the most inequality rows a node LP had): for tests that must
// know a tree went deep enough to reach a particular kernel path
int orc_mip_solve_stats(int kind, int is_max, int is_bin, const void * tgtf, const void * vc,
                        int vc_rows, const void * eq, int eq_rows, const void * leq, int leq_rows,
                        int cols, const uint8_t * rat_ind, void * out_v, void * out_sol, long * nodes, int * max_leq_rows)
{
    if (kind == 0)
        return t_mip_solve<F64>(is_max, is_bin, tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows,
                                cols, rat_ind, out_v, out_sol, nodes, max_leq_rows);
    return t_mip_solve<R32>(is_max, is_bin, tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows,
                            cols, rat_ind, out_v, out_sol, nodes, max_leq_rows);
}

void orc_rat_op(int op, int32_t an, int32_t ad, int32_t bn, int32_t bd, int32_t * rn,
                int32_t * rd)
{
    R32 a(an, ad), b(bn, bd), r;
    switch (op) {
    case 0: r = mul(a, b); break;
    case 1: r = div(a, b); break;
    case 2: r = add(a, b); break;
    case 3: r = sub(a, b); break;
    default: r = a; reduce(r); break;
    }
    *rn = r.num; *rd = r.den;
}

int orc_rat_cmp(int cmp, int32_t an, int32_t ad, int32_t bn, int32_t bd)
{
    R32 a(an, ad), b(bn, bd);
    switch (cmp) {
    case 0: return lt(a, b);
    case 1: return le(a, b);
    case 2: return gt(a, b);
    case 3: return ge(a, b);
    case 4: return eq(a, b);
    default: return ne(a, b);
    }
}

int orc_flt_cmp(int cmp, double x, double y)
{
    F64 a(x), b(y);
    switch (cmp) {
    case 0: return lt(a, b);
    case 1: return le(a, b);
    case 2: return gt(a, b);
    case 3: return ge(a, b);
    case 4: return eq(a, b);
    default: return ne(a, b);
    }
}

// Bare K1 on a caller-owned tableau, in place: the arithmetic of
// lpsol.h:1471-1501 without the basis bookkeeping. The CPU baseline of bench.py.
void orc_pivot_f64(double * tab, int m, int W, double * obj, int rhs_idx, int row, int col)
{
    static_assert(sizeof(F64) == sizeof(double), "F64 must be a bare double");
    pivot_cells((F64*)tab, m, W, (F64*)obj, rhs_idx, row, col);
}

void orc_pivot_rat32(int32_t * tab, int m, int W, int32_t * obj, int rhs_idx, int row, int col)
{
    static_assert(sizeof(R32) == 2 * sizeof(int32_t), "R32 must be two int32");
    pivot_cells((R32*)tab, m, W, (R32*)obj, rhs_idx, row, col);
}

void orc_set_strict(int on) { strict_mode() = on != 0; }
long long orc_appro_count(void) { return counters().appro_calls; }
long long orc_reduce_count(void) { return counters().reduce_calls; }
long long orc_pivot_count(void) { return counters().pivots; }

} // extern "C"

#include "oracle_lineq_abi.inc"
