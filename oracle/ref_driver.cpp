// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// C-ABI driver around the *real* xpoly reference (stevenknown/xpoly), compiled
// by oracle/Makefile against the sources where they lie under
// /root/reference/src/com (never copied into this repo). The output is
// oracle/_ref/libxpoly_ref.so, used only by tests/, tools/gen_golden.py,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
//
// Why a driver of our own: the reference ships no FFI and its only executable
// (src/example/example.cpp) neither builds (missing strbuf.h include) nor runs
// correctly on x86-64 (sete() walks the caller stack), see SURVEY.md section 0.5.
// This file therefore fills matrices via set()/setr() only.
//
// Scalar kinds: 0 = Float (fp64, flat double[]), 1 = Rational (flat
// int32 {num,den} pairs).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "ltype.h"
#include "comf.h"
#include "smempool.h"
#include "strbuf.h"
#include "rational.h"
#include "flty.h"
#include "sstl.h"
#include "matt.h"
#include "xmat.h"
#include "bs.h"
#include "sbs.h"
#include "sgraph.h"
#include "lpsol.h"
#include "linsys.h"

using namespace xcom;

namespace xcom {
extern LONGLONG g_appro_count;
extern LONGLONG g_red_count;
}

namespace {

struct R32 { int32_t num, den; };

void load(FloatMat & m, const void * src, int rows, int cols)
{
    m.reinit(rows, cols);
    const double * p = (const double*)src;
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++)
            m.set(i, j, Float(p[(size_t)i * cols + j]));
}

void load(RMat & m, const void * src, int rows, int cols)
{
    m.reinit(rows, cols);
    const R32 * p = (const R32*)src;
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++)
            m.setr(i, j, p[(size_t)i * cols + j].num, p[(size_t)i * cols + j].den);
}

void store(const FloatMat & m, void * dst)
{
    double * p = (double*)dst;
    for (UINT i = 0; i < m.get_row_size(); i++)
        for (UINT j = 0; j < m.get_col_size(); j++)
            p[(size_t)i * m.get_col_size() + j] = m.get(i, j).f();
}

void store(const RMat & m, void * dst)
{
    R32 * p = (R32*)dst;
    for (UINT i = 0; i < m.get_row_size(); i++)
        for (UINT j = 0; j < m.get_col_size(); j++) {
            Rational r = m.get(i, j);
            p[(size_t)i * m.get_col_size() + j].num = r.num();
            p[(size_t)i * m.get_col_size() + j].den = r.den();
        }
}

void store1(Float v, void * dst) { *(double*)dst = v.f(); }
void store1(Rational v, void * dst) { ((R32*)dst)->num = v.num(); ((R32*)dst)->den = v.den(); }

template <class Mat, class T>
int six_solve(int is_max, const void * tgtf, const void * vc, int vc_rows,
              const void * eq, int eq_rows, const void * leq, int leq_rows,
              int cols, unsigned max_iter, void * out_v, void * out_sol)
{
    Mat mt, mvc, meq, mleq, sol;
    load(mt, tgtf, 1, cols);
    load(mvc, vc, vc_rows, cols);
    if (eq_rows > 0) load(meq, eq, eq_rows, cols);
    if (leq_rows > 0) load(mleq, leq, leq_rows, cols);
    SIX<Mat, T> six;
    six.set_param(0, max_iter);
    T v;
    UINT st = is_max ? six.maxm(v, sol, mt, mvc, meq, mleq)
                     : six.minm(v, sol, mt, mvc, meq, mleq);
    store1(v, out_v);
    if (st == SIX_SUCC && sol.get_col_size() == (UINT)cols) store(sol, out_sol);
    return (int)st;
}

template <class Mat, class T>
int two_stage(const void * leq, int m, int cols, const void * vc,
              const void * tgtf, unsigned max_iter,
              void * out_tab, int * out_rows, int * out_cols,
              void * out_tgtf, uint8_t * out_nvset, uint8_t * out_bvset,
              int32_t * out_bv2eq, int32_t * out_eq2bv, int * out_rhs,
              void * out_maxv, void * out_sol)
{
    Mat mleq, mvc, mt, sol;
    load(mleq, leq, m, cols);
    load(mvc, vc, cols - 1, cols);
    load(mt, tgtf, 1, cols);
    SIX<Mat, T> six;
    six.set_param(0, max_iter);
    T maxv = 0;
    Vector<bool> nvset, bvset;
    Vector<INT> bv2eq, eq2bv;
    INT rhs = cols - 1;
    UINT st = six.TwoStageMethod(mleq, mvc, mt, sol, maxv, nvset, bvset,
                                 bv2eq, eq2bv, rhs);
    *out_rows = mleq.get_row_size();
    *out_cols = mleq.get_col_size();
    *out_rhs = rhs;
    store(mleq, out_tab);
    store(mt, out_tgtf);
    for (INT i = 0; i < rhs; i++) {
        out_nvset[i] = nvset.get(i) ? 1 : 0;
        out_bvset[i] = bvset.get(i) ? 1 : 0;
        out_bv2eq[i] = bv2eq.get(i);
    }
    for (UINT i = 0; i < mleq.get_row_size(); i++) out_eq2bv[i] = eq2bv.get(i);
    store1(maxv, out_maxv);
    if (sol.size() > 0 && out_sol) store(sol, out_sol);
    return (int)st;
}

template <class Mat, class T>
int mip_solve(int is_max, int is_bin, const void * tgtf, const void * vc,
              int vc_rows, const void * eq, int eq_rows, const void * leq,
              int leq_rows, int cols, const uint8_t * rat_ind,
              void * out_v, void * out_sol)
{
    Mat mt, mvc, meq, mleq, sol;
    load(mt, tgtf, 1, cols);
    load(mvc, vc, vc_rows, cols);
    if (eq_rows > 0) load(meq, eq, eq_rows, cols);
    if (leq_rows > 0) load(mleq, leq, leq_rows, cols);
    BMat ind;
    if (rat_ind) {
        ind.reinit(1, cols);
        for (int j = 0; j < cols; j++) ind.set(0, j, rat_ind[j] != 0);
    }
    MIP<Mat, T> mip;
    T v;
    UINT st = is_max
        ? mip.maxm(v, sol, mt, mvc, meq, mleq, is_bin != 0, rat_ind ? &ind : NULL)
        : mip.minm(v, sol, mt, mvc, meq, mleq, is_bin != 0, rat_ind ? &ind : NULL);
    store1(v, out_v);
    if (st == IP_SUCC && sol.get_col_size() == (UINT)cols) store(sol, out_sol);
    return (int)st;
}

} // namespace

extern "C" {

int ref_six_solve(int kind, int is_max, const void * tgtf, const void * vc,
                  int vc_rows, const void * eq, int eq_rows, const void * leq,
                  int leq_rows, int cols, unsigned max_iter, void * out_v,
                  void * out_sol)
{
    if (kind == 0)
        return six_solve<FloatMat, Float>(is_max, tgtf, vc, vc_rows, eq, eq_rows,
                                          leq, leq_rows, cols, max_iter, out_v, out_sol);
    return six_solve<RMat, Rational>(is_max, tgtf, vc, vc_rows, eq, eq_rows,
                                     leq, leq_rows, cols, max_iter, out_v, out_sol);
}

// Runs the public SIX::TwoStageMethod (lpsol.h:1907) on an already normalised
// problem (x >= 0 for all variables) and hands back the live tableau after at
// most max_iter pivots. out_tab needs room for m * (cols + m + 1) elements.
int ref_two_stage(int kind, const void * leq, int m, int cols, const void * vc,
                  const void * tgtf, unsigned max_iter, void * out_tab,
                  int * out_rows, int * out_cols, void * out_tgtf,
                  uint8_t * out_nvset, uint8_t * out_bvset, int32_t * out_bv2eq,
                  int32_t * out_eq2bv, int * out_rhs, void * out_maxv,
                  void * out_sol)
{
    if (kind == 0)
        return two_stage<FloatMat, Float>(leq, m, cols, vc, tgtf, max_iter, out_tab,
            out_rows, out_cols, out_tgtf, out_nvset, out_bvset, out_bv2eq,
            out_eq2bv, out_rhs, out_maxv, out_sol);
    return two_stage<RMat, Rational>(leq, m, cols, vc, tgtf, max_iter, out_tab,
        out_rows, out_cols, out_tgtf, out_nvset, out_bvset, out_bv2eq,
        out_eq2bv, out_rhs, out_maxv, out_sol);
}

int ref_mip_solve(int kind, int is_max, int is_bin, const void * tgtf,
                  const void * vc, int vc_rows, const void * eq, int eq_rows,
                  const void * leq, int leq_rows, int cols,
                  const uint8_t * rat_ind, void * out_v, void * out_sol)
{
    // MIP<FloatMat,Float> does not compile in the reference as shipped: the
    // virtual MIP::dump_end_six (lpsol.h:2242-2254) passes a StrBuf to
    // Float::format(CHAR*). Only the Rational instantiation exists.
    if (kind == 0) return -100;
    return mip_solve<RMat, Rational>(is_max, is_bin, tgtf, vc, vc_rows, eq,
        eq_rows, leq, leq_rows, cols, rat_ind, out_v, out_sol);
}

// Scalar ops of Rational: op 0 '*', 1 '/', 2 '+', 3 '-', 4 reduce(a).
void ref_rat_op(int op, int32_t an, int32_t ad, int32_t bn, int32_t bd,
                int32_t * rn, int32_t * rd)
{
    Rational a, b, r;
    a.num() = an; a.den() = ad; b.num() = bn; b.den() = bd;
    switch (op) {
    case 0: r = a * b; break;
    case 1: r = a / b; break;
    case 2: r = a + b; break;
    case 3: r = a - b; break;
    default: r = a; r.reduce(); break;
    }
    *rn = r.num(); *rd = r.den();
}

// cmp: 0 '<', 1 '<=', 2 '>', 3 '>=', 4 '==', 5 '!='
int ref_rat_cmp(int cmp, int32_t an, int32_t ad, int32_t bn, int32_t bd)
{
    Rational a, b;
    a.num() = an; a.den() = ad; b.num() = bn; b.den() = bd;
    switch (cmp) {
    case 0: return a < b;
    case 1: return a <= b;
    case 2: return a > b;
    case 3: return a >= b;
    case 4: return a == b;
    default: return a != b;
    }
}

int ref_flt_cmp(int cmp, double x, double y)
{
    Float a(x), b(y);
    switch (cmp) {
    case 0: return a < b;
    case 1: return a <= b;
    case 2: return a > b;
    case 3: return a >= b;
    case 4: return a == b;
    default: return a != b;
    }
}

// Lineq::fme (linsys.cpp:656): eliminate variable u from the rows x cols
// system whose constant column is rhs_idx. out must hold cap_rows rows.
// Returns 1/0 = fme()'s bool, -1 if the result does not fit.
int ref_fme(const void * in, int rows, int cols, int rhs_idx, int u,
            int darkshadow, void * out, int cap_rows, int * out_rows,
            int * out_cols)
{
    RMat m, res;
    load(m, in, rows, cols);
    Lineq lin(&m, rhs_idx);
    bool ok = lin.fme((UINT)u, res, darkshadow != 0);
    *out_rows = res.get_row_size();
    *out_cols = res.get_col_size();
    if ((int)res.get_row_size() > cap_rows) return -1;
    if (res.size() > 0) store(res, out);
    return ok ? 1 : 0;
}

// Lineq::reduce (linsys.cpp:359) in place; returns its bool (consistent).
int ref_reduce(void * inout, int rows, int cols, int rhs_idx, int is_intersect,
               int * out_rows, int * out_cols)
{
    RMat m;
    load(m, inout, rows, cols);
    Lineq lin(NULL);
    bool ok = lin.reduce(m, (UINT)rhs_idx, is_intersect != 0);
    *out_rows = m.get_row_size();
    *out_cols = m.get_col_size();
    if (m.size() > 0 && (int)m.get_row_size() <= rows) store(m, inout);
    return ok ? 1 : 0;
}

// Lineq::removeIdenRow (linsys.cpp:1209) in place.
void ref_remove_iden_row(void * inout, int rows, int cols, int * out_rows)
{
    RMat m;
    load(m, inout, rows, cols);
    Lineq lin(NULL);
    lin.removeIdenRow(m);
    *out_rows = m.get_row_size();
    if (m.size() > 0) store(m, inout);
}

// Lineq::move2var (linsys.cpp:1177-1200): constant symbols first_sym..last_sym become variables in front of the
// constant column rhs_idx. In place (the shape does not change).
void ref_move2var(void * inout, int rows, int cols, int rhs_idx, int first_sym, int last_sym)
{
    RMat m;
    load(m, inout, rows, cols);
    Lineq lin(NULL);
    lin.move2var(m, (UINT)rhs_idx, (UINT)first_sym, (UINT)last_sym, NULL, NULL);
    store(m, inout);
}

// Lineq::has_solution (linsys.cpp:830).
int ref_has_solution(const void * leq, int leq_rows, const void * eq,
                     int eq_rows, const void * vc, int vc_rows, int cols,
                     int rhs_idx, int is_int_sol, int is_unique_sol)
{
    RMat mleq, meq, mvc;
    if (leq_rows > 0) load(mleq, leq, leq_rows, cols);
    if (eq_rows > 0) load(meq, eq, eq_rows, cols);
    load(mvc, vc, vc_rows, cols);
    Lineq lin(NULL);
    return lin.has_solution(mleq, meq, mvc, (UINT)rhs_idx, is_int_sol != 0,
                            is_unique_sol != 0) ? 1 : 0;
}

// Lineq::calcBound (linsys.cpp:1047): out is [rhs_idx][cap_rows][cols], out_rows[rhs_idx].
int ref_calc_bound(const void * in, int rows, int cols, int rhs_idx, void * out, int cap_rows, int * out_rows)
{
    RMat m;
    load(m, in, rows, cols);
    RMat * v = new RMat[rhs_idx];
    List<RMat*> bd;
    for (int i = 0; i < rhs_idx; i++) bd.append_tail(&v[i]);
    Lineq lin(&m, rhs_idx);
    bool ok = lin.calcBound(bd);
    int rc = ok ? 1 : 0;
    for (int j = 0; j < rhs_idx; j++) {
        out_rows[j] = v[j].get_row_size();
        if ((int)v[j].get_row_size() > cap_rows) { rc = -1; break; }
        if (v[j].size() > 0) store(v[j], (char*)out + (size_t)j * cap_rows * cols * sizeof(R32));
    }
    delete [] v;
    return rc;
}

// Matrix<Rational>::rank / det / inv (matt.h:2614, :1621, :1743).
int ref_rat_rank(const void * in, int rows, int cols)
{
    RMat m;
    load(m, in, rows, cols);
    return (int)m.rank();
}

void ref_rat_det(const void * in, int n, int32_t * num, int32_t * den)
{
    RMat m;
    load(m, in, n, n);
    Rational d = m.det();
    *num = d.num(); *den = d.den();
}

int ref_rat_inv(const void * in, int n, void * out)
{
    RMat m, e;
    load(m, in, n, n);
    bool ok = m.inv(e);
    if (ok) store(e, out);
    return ok ? 1 : 0;
}

// Matrix<Rational>::rank(&basis, is_unitarize) (matt.h:2614) and null (matt.h:2546).
int ref_rat_rank_basis(const void * in, int rows, int cols, int unitarize, void * out, int * out_rows)
{
    RMat m, b;
    load(m, in, rows, cols);
    int rk = (int)m.rank(&b, unitarize != 0);
    *out_rows = (int)b.get_row_size();
    if (b.get_row_size() > 0 && b.get_col_size() > 0) store(b, out);
    return rk;
}

void ref_rat_null(const void * in, int rows, int cols, void * out)
{
    RMat m, ns;
    load(m, in, rows, cols);
    m.null(ns);
    store(ns, out);
}

// INTMat::hnf / gcd (xmat.cpp:912, :996).
int ref_int_hnf(const int32_t * in, int rows, int cols, int32_t * h, int32_t * u)
{
    INTMat a(rows, cols), hh, uu;
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) a.set(i, j, in[i * cols + j]);
    a.hnf(hh, uu);
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) h[i * cols + j] = hh.get(i, j);
    for (int i = 0; i < cols; i++) for (int j = 0; j < cols; j++) u[i * cols + j] = uu.get(i, j);
    return 0;
}

void ref_int_gcd(int32_t * inout, int rows, int cols)
{
    INTMat a(rows, cols);
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) a.set(i, j, inout[i * cols + j]);
    a.gcd();
    for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) inout[i * cols + j] = a.get(i, j);
}

long long ref_appro_count(void) { return g_appro_count; }
long long ref_reduce_count(void) { return g_red_count; }

} // extern "C"
