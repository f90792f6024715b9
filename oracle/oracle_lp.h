// TEST INFRASTRUCTURE -- CPU restatement ("port") of xpoly's simplex solver
// SIX<Mat,T> and branch-and-bound MIP<Mat,T> (reference: src/com/lpsol.h).
// Never included by the product (xpoly_amd/).
//
// This is a bug-compatible replay, not a good LP solver: same entering rule
// (first positive reduced cost whose pivot-pair row still has a free slot),
// same ratio test (strict '>' so the lowest row wins ties, tolerant '<=' skip),
// same anti-cycling pair table, same order of every multiply and add.
#ifndef XPOLY_ORACLE_LP_H
#define XPOLY_ORACLE_LP_H

#include <vector>
#include <stdint.h>
#include <stddef.h>
#include "oracle_scalar.h"

namespace orc {

enum { SIX_SUCC = 0, SIX_UNBOUND = 1, SIX_NO_PRI_FEASIBLE_SOL = 2,
       SIX_OPTIMAL_IS_INFEASIBLE = 3, SIX_TIME_OUT = 4 };           // lpsol.h:198-202
enum { IP_SUCC = 0, IP_UNBOUND = 1, IP_NO_PRI_FEASIBLE_SOL = 2,
       IP_NO_BETTER_THAN_BEST_SOL = 3 };                             // lpsol.h:2082-2085
// Returned when the reference would read outside a buffer or divide an int
// by zero (undefined there, so there is nothing to be compatible with).
enum { ORC_REF_UNDEFINED = -7 };
// strict (default): report ORC_REF_UNDEFINED wherever the x86-64 reference is
// undefined. Non-strict: follow the evident intent instead (used only to check
// the product on inputs the reference cannot handle, e.g. free variables, whose
// vcmap is filled by the stack-walking sete() at lpsol.h:1376-1378).
inline bool & strict_mode() { static bool s = true; return s; }

// Dense row-major matrix (matt.h:152-156, :289-295). New cells are always
// the scalar zero (0.0 or 0/1): matt.h:587-591, :851-860, xmat.cpp:216-224.
template <class S> struct Mat {
    int r, c;
    std::vector<S> a;
    Mat() : r(0), c(0) {}
    Mat(int rows, int cols) : r(rows), c(cols), a((size_t)rows * cols) {}
    S & at(int i, int j) { return a[(size_t)i * c + j]; }
    const S & at(int i, int j) const { return a[(size_t)i * c + j]; }
    size_t size() const { return (size_t)r * c; }
    S * row(int i) { return &a[(size_t)i * c]; }
    const S * row(int i) const { return &a[(size_t)i * c]; }
};

template <class S> void insert_cols(Mat<S> & m, int before, int n)   // matt.h:2824-2844
{
    if (n == 0) return;
    Mat<S> t(m.r, m.c + n);
    for (int i = 0; i < m.r; i++) {
        for (int j = 0; j < before; j++) t.at(i, j) = m.at(i, j);
        for (int j = before; j < m.c; j++) t.at(i, j + n) = m.at(i, j);
    }
    m = t;
}
template <class S> void delete_col(Mat<S> & m, int col)               // matt.h:884-905
{
    Mat<S> t(m.r, m.c - 1);
    for (int i = 0; i < m.r; i++)
        for (int j = 0, k = 0; j < m.c; j++)
            if (j != col) t.at(i, k++) = m.at(i, j);
    m = t;
}
template <class S> void append_rows(Mat<S> & m, int n)                // matt.h:557-591
{
    if (n == 0) return;
    if (m.r == 0) { m = Mat<S>(n, m.c ? m.c : 1); return; }
    m.a.resize((size_t)(m.r + n) * m.c);
    m.r += n;
}

// Scale a run of cells the way Matrix::mul / mulOfRow / mulOfColumn(s) do
// (matt.h:1331-1348, :1353-1368, :1395-1412, :1415-1432): x == 1 is a no-op,
// x "==" 0 stores zeros, otherwise cell * x.  mulOfRow tests == 1 first, mul
// tests == 0 first; the two conditions exclude each other for both scalars.
template <class S> void scale_cells(S * p, int n, int stride, S x)
{
    if (eq(x, S(1))) return;
    if (eq(x, S(0))) { for (int k = 0; k < n; k++) p[(size_t)k * stride] = S(0); return; }
    for (int k = 0; k < n; k++) p[(size_t)k * stride] = mul(p[(size_t)k * stride], x);
}

// PivotPairTab (lpsol.h:68-154): n x n bool, rows = entering, cols = leaving.
struct PairTab {
    int n;
    std::vector<uint8_t> used;
    explicit PairTab(int nvars) : n(nvars), used((size_t)nvars * nvars, 0) {}
    void mark(int nv, int bv) { used[(size_t)nv * n + bv] = 1; }            // :100-104
    bool seen(int nv, int bv) const { return used[(size_t)nv * n + bv] != 0; } // :107-111
    void close_row(int nv)                                                   // :114-121
    {
        for (int j = 0; j < n; j++) if (j != nv) used[(size_t)nv * n + j] = 1;
    }
    bool row_open(int nv) const                                              // :124-137
    {
        for (int j = 0; j < n; j++) if (j != nv && !used[(size_t)nv * n + j]) return true;
        return false;
    }
    bool col_open(int bv) const                                              // :140-153
    {
        for (int i = 0; i < n; i++) if (i != bv && !used[(size_t)i * n + bv]) return true;
        return false;
    }
};

// The slack-form state SIX threads through its private methods: tableau,
// objective row, the two live columns of 'vc', the basis bookkeeping.
template <class S> struct Slack {
    Mat<S> eq;                 // m x W, constant in column rhs (W = rhs + 1)
    std::vector<S> obj;        // W
    std::vector<S> vcd, vcr;   // vc(i,i) and vc(i,rhs): all is_feasible reads (lpsol.h:798-802)
    std::vector<uint8_t> nv, bv;
    std::vector<int> bv2eq, eq2bv;
    int rhs;
    // trace of (entering, leaving) pairs actually pivoted, for the tests
    std::vector<int> trace;
};

// The arithmetic of SIX::pivot (lpsol.h:1471-1501) on a bare m x W tableau:
// scale row r by the reciprocal of the pivot, eliminate column nv from every
// other row, fold the pivot row into the objective.
template <class S>
void pivot_cells(S * tab, int m, int W, S * obj, int rhs, int r, int nv)
{
    S * prow = tab + (size_t)r * W;
    scale_cells(prow, W, 1, div(S(1), prow[nv]));                          // :1471
    std::vector<S> e(prow, prow + W);                                      // :1473-1474
    for (int i = 0; i < m; i++) {                                          // :1481-1490
        if (i == r) continue;
        S * row = tab + (size_t)i * W;
        S k = neg(row[nv]);
        for (int j = 0; j < W; j++) row[j] = add(row[j], mul(k, e[j]));
    }
    scale_cells(e.data(), W, 1, S(-1));                                    // :1496
    for (int j = rhs; j < W; j++) e[j] = neg(e[j]);                        // :1497-1499
    scale_cells(e.data(), W, 1, obj[nv]);                                  // :1500
    for (int j = 0; j < W; j++) obj[j] = add(e[j], obj[j]);                // :1501, matt.h:1450-1460
}

// SIX::pivot (lpsol.h:1456-1511): the above plus the basis bookkeeping.
template <class S> void pivot(Slack<S> & P, int nv, int bv)
{
    const int r = P.bv2eq[bv];
    counters().pivots++;                                                  // telemetry only (tools/gen_golden_end.py)
    pivot_cells(P.eq.a.data(), P.eq.r, P.eq.c, P.obj.data(), P.rhs, r, nv);
    P.nv[nv] = 0; P.nv[bv] = 1; P.bv[nv] = 1; P.bv[bv] = 0;               // :1504-1507
    P.eq2bv[r] = nv; P.bv2eq[nv] = r; P.bv2eq[bv] = -1;                   // :1508-1510
    P.trace.push_back(nv); P.trace.push_back(bv);
}

// SIX::findPivotBV (lpsol.h:553-663).
template <class S> int ratio_test(const Slack<S> & P, const PairTab & T, int nv)
{
    const Mat<S> & E = P.eq;
    int best = -1;
    S bestv;
    for (int i = 0; i < E.r; i++) {                                       // :571-612
        S a = E.at(i, nv);
        if (le(a, S(0))) continue;
        int b = P.eq2bv[i];
        if (T.seen(nv, b) || !T.col_open(b)) continue;
        S q = div(E.at(i, P.rhs), a);
        if (best < 0 || gt(bestv, q)) { bestv = q; best = i; }
    }
    if (best < 0) {                                                       // :623-658
        for (int i = 0; i < E.r; i++) {
            int b = P.eq2bv[i];
            if (T.seen(nv, b) || !T.col_open(b)) continue;
            S a = E.at(i, nv);
            if (eq(a, S(0))) continue;
            S q = div(E.at(i, P.rhs), a);
            if (best < 0 || gt(bestv, q)) { bestv = q; best = i; }
        }
        if (best < 0) return -1;
    }
    return P.eq2bv[best];
}

// SIX::findPivotNVandBVPair (lpsol.h:671-773); ALLOW_RELAX_CURRENT_SOL is
// not defined (:756), so negative coefficients are never tried.
template <class S> bool find_pair(const Slack<S> & P, const PairTab & T, int & nv, int & bv)
{
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < P.rhs; i++) {
            if (P.bv[i]) continue;
            if (!T.row_open(i)) continue;
            S c = P.obj[i];
            bool take;
            if (gt(c, S(0))) take = true;
            else if (eq(c, S(0))) take = (pass == 1);
            else take = false;
            if (!take) continue;
            int b = ratio_test(P, T, i);
            if (b < 0) continue;
            nv = i; bv = b;
            return true;
        }
    }
    return false;
}

// SIX::is_feasible in equality mode (lpsol.h:784-822). Rational::reduce of the
// constant column is written back into the tableau (xmat.cpp:619-625).
template <class S> bool feasible(Slack<S> & P, const std::vector<S> & x)
{
    for (int i = 0; i < P.rhs; i++)
        if (gt(mul(P.vcd[i], x[i]), P.vcr[i])) return false;
    for (int i = 0; i < P.eq.r; i++) {
        S sum = S(0);
        const S * row = P.eq.row(i);
        for (int j = 0; j < P.rhs; j++) sum = add(sum, mul(row[j], x[j]));
        reduce(sum);
        reduce(P.eq.at(i, P.rhs));
        if (ne(sum, P.eq.at(i, P.rhs))) return false;
    }
    return true;
}

// SIX::solveSlackForm (lpsol.h:1008-1191).
template <class S>
int solve_slack(Slack<S> & P, unsigned max_iter, S & maxv, std::vector<S> & x)
{
    PairTab T(P.rhs);                                                     // :1021, :390-399
    maxv = S(0);
    x.assign(P.obj.size(), S(0));
    unsigned done = 0;
    while (done < max_iter) {
        int nv = -1, bv = -1;
        bool none_positive = true;
        for (int j = 0; j < P.rhs; j++) {                                 // :1054-1069
            if (!P.nv[j]) { P.obj[j] = S(0); continue; }
            if (gt(P.obj[j], S(0))) {
                none_positive = false;
                if (T.row_open(j)) { nv = j; break; }
            }
        }
        if (nv < 0) {
            if (none_positive) {                                          // :1089-1128
                for (size_t j = 0; j < x.size(); j++) x[j] = S(0);
                for (int j = 0; j < P.rhs; j++)
                    if (P.bv[j]) x[j] = P.eq.at(P.bv2eq[j], P.rhs);
                if (!feasible(P, x)) return SIX_OPTIMAL_IS_INFEASIBLE;
                maxv = P.obj[P.rhs];
                return SIX_SUCC;
            }
            if (!find_pair(P, T, nv, bv)) return SIX_UNBOUND;             // :1138-1141
        } else {
            bv = ratio_test(P, T, nv);                                    // :1145-1151
            if (bv < 0) { T.close_row(nv); continue; }
        }
        T.mark(nv, bv);                                                   // :1156
        pivot(P, nv, bv);                                                 // :1170
        done++;
    }
    return SIX_TIME_OUT;
}

// SIX::slack (lpsol.h:1406-1433): one identity column per row before the
// constant column; vc gains a "-s <= 0" row for each.
template <class S> void add_slacks(Slack<S> & P)
{
    const int m = P.eq.r, at = P.rhs;
    insert_cols(P.eq, at, m);
    P.obj.insert(P.obj.begin() + at, (size_t)m, S(0));
    for (int i = 0; i < m; i++) {
        P.eq.at(i, at + i) = S(1);
        P.vcd.push_back(S(-1));
        P.vcr.push_back(S(0));
    }
    P.rhs += m;
}

template <class S> void init_basis(Slack<S> & P, int first_slack)        // :880-890, :1830-1841
{
    P.nv.assign(P.rhs, 0); P.bv.assign(P.rhs, 0);
    P.bv2eq.assign(P.rhs, 0); P.eq2bv.assign(P.eq.r, 0);
    for (int i = 0; i < first_slack; i++) { P.bv2eq[i] = -1; P.nv[i] = 1; }
    for (int i = first_slack, j = 0; i < P.rhs; i++, j++) {
        P.bv[i] = 1; P.nv[i] = 0; P.eq2bv[j] = i; P.bv2eq[i] = j;
    }
}

// {R,Float}Mat::substit on a one-row target with is_eq == false
// (xmat.cpp:571-599, :1491-1519).
template <class S>
void substitute(std::vector<S> & f, const S * expr, int var, int rhs)
{
    const int W = (int)f.size();
    scale_cells(&f[rhs], W - rhs, 1, S(-1));
    if (ne(f[var], S(0)) && !eq(expr[var], S(0))) {
        std::vector<S> t(expr, expr + W);
        if (ne(f[var], t[var])) {
            S k = div(neg(f[var]), t[var]);
            if (eq(k, S(0))) { for (int j = 0; j < W; j++) t[j] = S(0); }   // Matrix::mul order, matt.h:1335-1341
            else if (!eq(k, S(1))) { for (int j = 0; j < W; j++) t[j] = mul(t[j], k); }
        } else {
            for (int j = 0; j < W; j++) t[j] = mul(t[j], S(-1));
        }
        for (int j = 0; j < W; j++) f[j] = add(t[j], f[j]);
    }
    scale_cells(&f[rhs], W - rhs, 1, S(-1));
}

// SIX::constructBasicFeasibleSolution (lpsol.h:839-988).
template <class S> int phase_one(Slack<S> & P, unsigned max_iter)
{
    const std::vector<S> obj0 = P.obj;
    const int rhs0 = P.rhs, m = P.eq.r;
    const int xa = rhs0;
    insert_cols(P.eq, rhs0, 1);                                            // :860-861
    for (int i = 0; i < m; i++) P.eq.at(i, xa) = S(-1);
    P.obj.assign(obj0.size() + 1, S(0));                                   // :862-864
    P.obj[xa] = S(-1);
    P.vcd.push_back(S(-1)); P.vcr.push_back(S(0));                         // :865-868
    P.rhs = rhs0 + 1;
    const int first_slack = P.rhs;
    add_slacks(P);                                                         // :875
    init_basis(P, first_slack);
    int row = 0;                                                           // :894-904
    for (int i = 1; i < m; i++)
        if (gt(P.eq.at(row, P.rhs), P.eq.at(i, P.rhs))) row = i;
    pivot(P, xa, P.eq2bv[row]);                                            // :906-908
    S best; std::vector<S> x;
    if (solve_slack(P, max_iter, best, x) != SIX_SUCC) return 0;           // :912-915
    reduce(best);
    if (ne(best, S(0))) return 0;                                          // :919-922
    if (P.bv[xa]) {                                                        // :924-941
        int r = P.bv2eq[xa], cand = 0;
        for (; cand < P.rhs; cand++) {
            if (!P.nv[cand]) continue;
            reduce(P.eq.at(r, cand));
            if (ne(P.eq.at(r, cand), S(0))) break;
        }
        if (cand >= P.rhs) return ORC_REF_UNDEFINED;   // reference pivots on the constant column
        pivot(P, cand, xa);
    }
    const int W = P.eq.c;                                                  // :944-953
    std::vector<S> f(obj0.begin(), obj0.begin() + rhs0);
    f.insert(f.end(), (size_t)(W - (int)obj0.size()), S(0));
    f.insert(f.end(), obj0.begin() + rhs0, obj0.end());
    for (int i = 0; i < P.rhs; i++) {
        reduce(f[i]);
        if (ne(f[i], S(0)) && P.bv[i]) substitute(f, P.eq.row(P.bv2eq[i]), i, P.rhs);
    }
    P.obj = f;
    P.obj.erase(P.obj.begin() + xa);                                       // :956-959
    delete_col(P.eq, xa);
    P.vcd.erase(P.vcd.begin() + xa); P.vcr.erase(P.vcr.begin() + xa);
    P.nv.erase(P.nv.begin() + xa); P.bv.erase(P.bv.begin() + xa);          // :962-975
    P.bv2eq.erase(P.bv2eq.begin() + xa);
    for (size_t i = 0; i < P.eq2bv.size(); i++) if (P.eq2bv[i] > xa) P.eq2bv[i]--; // :978-983
    P.rhs--;
    return 1;
}

// SIX::stage1 + SIX::TwoStageMethod (lpsol.h:1784-1844, :1907-1930).
template <class S>
int two_stage(Slack<S> & P, unsigned max_iter, S & maxv, std::vector<S> & x)
{
    bool any_pos = false;
    for (int i = 0; i < P.rhs && !any_pos; i++) any_pos = gt(P.obj[i], S(0));
    bool rhs_ok = true;                                                    // :1760-1770
    for (int i = 0; i < P.eq.r && rhs_ok; i++) rhs_ok = !lt(P.eq.at(i, P.rhs), S(0));
    if (!any_pos || !rhs_ok) {
        int ok = phase_one(P, max_iter);
        if (ok == ORC_REF_UNDEFINED) return ORC_REF_UNDEFINED;
        if (!ok) return SIX_NO_PRI_FEASIBLE_SOL;
    } else {
        const int first_slack = P.rhs;
        add_slacks(P);
        init_basis(P, first_slack);
    }
    return solve_slack(P, max_iter, maxv, x);
}

// A linear program in the reference's calling convention (lpsol.h:1979-1991):
// every matrix has 'cols' columns, the last being the constant.
template <class S> struct Problem {
    int cols;
    std::vector<S> tgtf;       // 1 x cols
    Mat<S> vc, eq, leq;        // vc is (cols-1) x cols
};

template <class S> bool col_is_zero(const Mat<S> & m, int col)           // matt.h:2360-2370
{
    for (int i = 0; i < m.r; i++) if (!eq(m.at(i, col), S(0))) return false;
    return true;
}

// SIX::convertEq2Ineq (lpsol.h:1197-1278). 'rhs' is m_rhs_idx. The reference
// indexes the equality row with the *inequality row number* at :1232 -- kept.
template <class S> int fold_equalities(Mat<S> & leq, const Mat<S> & E, int rhs)
{
    if (E.size() == 0) return 0;
    std::vector<uint8_t> gone(E.r, 0);
    int left = E.r;
    if (leq.size() > 0) {
        for (int j = 0; j < rhs; j++) {
            int cnt = 0, pos = 0;
            for (int i = 0; i < E.r; i++) {
                if (gone[i]) continue;
                if (ne(E.at(i, j), S(0))) { cnt++; pos = i; }
            }
            if (cnt != 1) continue;
            gone[pos] = 1; left--;
            for (int mrow = 0; mrow < leq.r; mrow++) {
                S v = leq.at(mrow, j);
                if (eq(v, S(0))) continue;
                if (mrow >= E.c) return ORC_REF_UNDEFINED;   // out-of-bounds read in the reference
                std::vector<S> t(E.row(pos), E.row(pos) + E.c);
                S lead = t[mrow];
                if (ne(lead, S(1))) scale_cells(t.data(), E.c, 1, div(S(1), lead));
                scale_cells(t.data(), E.c, 1, v);
                leq.at(mrow, j) = S(0);
                for (int k = rhs; k < E.c; k++) t[k] = neg(t[k]);
                for (int k = 0; k < E.c; k++) leq.at(mrow, k) = add(t[k], leq.at(mrow, k));
            }
        }
    }
    if (left > 0) {
        int c = leq.r;
        if (leq.size() == 0) leq = Mat<S>(left * 2, E.c);
        else append_rows(leq, left * 2);
        for (int i = 0; i < E.r; i++) {
            if (gone[i]) continue;
            for (int k = 0; k < E.c; k++) leq.at(c, k) = E.at(i, k);
            scale_cells(leq.row(c), E.c, 1, S(-1));
            for (int k = 0; k < E.c; k++) leq.at(c + 1, k) = E.at(i, k);
            c += 2;
        }
    }
    return 0;
}

// Result of SIX::normalize (lpsol.h:1290-1394): inequalities only, all
// variables non-negative, free variables split v = v' - v''.
template <class S> struct Normal {
    Mat<S> leq;
    std::vector<S> obj, vcd, vcr;
    std::vector<int> split;   // triples (orig, plus, minus)   -- 'vcmap'
    int rhs;
};

template <class S> int normalize(const Problem<S> & Q, Normal<S> & N)
{
    const int rhs0 = Q.cols - 1;
    N.leq = Q.leq;
    int rc = fold_equalities(N.leq, Q.eq, rhs0);
    if (rc) return rc;
    std::vector<int> free_vars;
    for (int i = 0; i < rhs0; i++) if (col_is_zero(Q.vc, i)) free_vars.push_back(i);
    const int extra = (int)free_vars.size();
    N.vcd.assign(rhs0 + extra, S(0)); N.vcr.assign(rhs0 + extra, S(0));
    for (int i = 0; i < rhs0 && i < Q.vc.r; i++) { N.vcd[i] = Q.vc.at(i, i); N.vcr[i] = Q.vc.at(i, rhs0); }
    Mat<S> tmp = N.leq;                                                    // pre-insertion copy, :1382
    insert_cols(N.leq, rhs0, extra);
    N.obj = Q.tgtf;
    N.obj.insert(N.obj.begin() + rhs0, (size_t)extra, S(0));
    N.split.clear();
    int last = rhs0 - 1;
    for (int i = 0; i < rhs0; i++) {
        if (!col_is_zero(Q.vc, i)) continue;
        N.vcd[i] = S(-1); N.vcd[last + 1] = S(-1);                         // :1372-1373
        N.split.push_back(i); N.split.push_back(i); N.split.push_back(last + 1);
        for (int r = 0; r < tmp.r; r++) N.leq.at(r, last + 1) = tmp.at(r, i);   // :1381-1384
        scale_cells(&N.leq.a[last + 1], N.leq.r, N.leq.c, S(-1));
        N.obj[last + 1] = Q.tgtf[i];                                       // :1387-1389
        scale_cells(&N.obj[last + 1], 1, 1, S(-1));
        last++;
    }
    N.rhs = last + 1;
    return 0;
}

// SIX::calcFinalSolution (lpsol.h:1851-1899); rhs0 is m_rhs_idx.
template <class S>
void final_solution(std::vector<S> & sol, S & v, std::vector<S> & x,
                    const std::vector<int> & split, const std::vector<S> & tgtf, int rhs0)
{
    for (size_t k = 0; k + 2 < split.size(); k += 3)
        x[split[k]] = sub(x[split[k + 1]], x[split[k + 2]]);
    const int cols = (int)tgtf.size();
    sol.assign(cols, S(0));
    for (int i = 0; i < rhs0; i++) sol[i] = x[i];
    for (int k = rhs0; k < cols; k++) sol[k] = S(1);
    v = S(0);
    for (int j = 0; j < cols; j++) v = add(v, mul(sol[j], tgtf[j]));
}

template <class S> Slack<S> slack_from(const Normal<S> & N)
{
    Slack<S> P;
    P.eq = N.leq; P.obj = N.obj; P.vcd = N.vcd; P.vcr = N.vcr; P.rhs = N.rhs;
    return P;
}

// SIX::maxm (lpsol.h:1993-2033).
template <class S>
int six_maxm(const Problem<S> & Q, unsigned max_iter, S & v, std::vector<S> & sol,
             std::vector<int> * trace = 0)
{
    Normal<S> N;
    v = S(0);
    int rc = normalize(Q, N);
    if (rc) return rc;
    Slack<S> P = slack_from(N);
    S best = S(0); std::vector<S> x;
    int st = two_stage(P, max_iter, best, x);
    if (trace) *trace = P.trace;
    v = S(0);
    if (st == SIX_SUCC && !N.split.empty() && strict_mode()) return ORC_REF_UNDEFINED;
    if (st == SIX_SUCC) {
        final_solution(sol, v, x, N.split, Q.tgtf, Q.cols - 1);
        reduce(v);
        for (size_t i = 0; i < sol.size(); i++) reduce(sol[i]);
    }
    return st;
}

// SIX::calcDualMaxm + SIX::minm (lpsol.h:1586-1655, :1662-1732).
template <class S>
int six_minm(const Problem<S> & Q, unsigned max_iter, S & v, std::vector<S> & sol,
             std::vector<int> * trace = 0)
{
    Normal<S> N;
    v = S(0);
    int rc = normalize(Q, N);
    if (rc) return rc;
    const int n = N.rhs, m = N.leq.r;        // primal: m rows, n variables
    Slack<S> D;                              // dual: n rows, m variables
    D.eq = Mat<S>(n, m + 1);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) D.eq.at(i, j) = N.leq.at(j, i);
    scale_cells(D.eq.a.data(), (int)D.eq.a.size(), 1, S(-1));            // :1607
    for (int i = 0; i < n; i++) D.eq.at(i, m) = N.obj[i];                 // :1610-1613
    D.obj.assign(m + 1, S(0));                                            // :1616-1619
    for (int j = 0; j < m; j++) D.obj[j] = N.leq.at(j, n);
    scale_cells(D.obj.data(), m + 1, 1, S(-1));
    D.vcd.assign(m, S(-1)); D.vcr.assign(m, S(0));                        // :1626-1629
    D.rhs = m;
    S best = S(0); std::vector<S> x;
    int st = two_stage(D, max_iter, best, x);
    if (trace) *trace = D.trace;
    if (st != SIX_SUCC) return st;
    if (!N.split.empty() && strict_mode()) return ORC_REF_UNDEFINED;
    // primal values are minus the reduced costs of the dual slacks, :1713-1716
    const int nd = m;                        // dual_num_nv
    const int norig = (int)D.obj.size() - 1 - nd;
    std::vector<S> y(norig + 1, S(0));
    for (int k = 0; k < norig; k++) y[k] = neg(D.obj[nd + k]);
    final_solution(sol, v, y, N.split, Q.tgtf, Q.cols - 1);
    reduce(v);
    for (size_t i = 0; i < sol.size(); i++) reduce(sol[i]);
    return st;
}

// ---------------------------------------------------------------------------
// MIP<Mat,T> (lpsol.h:2087-2702): depth-first branch and bound; every node is
// a from-scratch SIX solve with max_iter = 10000 (:2441).
// ---------------------------------------------------------------------------
template <class S> struct Mip {
    bool have_best;
    std::vector<S> best_sol;
    S best_v;
    const uint8_t * allow_rational;   // 1 x cols or null
    int rhs0;
    long nodes;
    int max_leq_rows;                 // the most inequality rows any node LP had (test statistics only)
    int depth, max_depth;             // recursion depth of the node being solved / the deepest one (statistics only)
    long spec_chain;                  // node LPs left on the walk's sequential chain when every ceiling child's LP is solved ahead
                                      // (the device walk's speculation, mip_kernels.hip.h): set by node() for its subtree
    Mip() : have_best(false), allow_rational(0), rhs0(0), nodes(0), max_leq_rows(0), depth(0), max_depth(0), spec_chain(0) {}
    struct DepthGuard { Mip & m; explicit DepthGuard(Mip & m_) : m(m_) { if (++m.depth > m.max_depth) m.max_depth = m.depth; } ~DepthGuard() { --m.depth; } };

    bool all_int(const std::vector<S> & s, int & col) const               // xmat.cpp:603-616, :1523-1536
    {
        for (size_t j = 0; j < s.size(); j++) if (!is_int(s[j])) { col = (int)j; return false; }
        return true;
    }

    bool satisfied(std::vector<S> & s, bool is_bin, int & col) const      // lpsol.h:2364-2408
    {
        if (allow_rational) {
            for (size_t j = 0; j < s.size(); j++) {
                reduce(s[j]);
                if (allow_rational[j]) continue;
                if (!is_int(s[j])) { col = (int)j; return false; }
                if (is_bin && ne(s[j], S(0)) && ne(s[j], S(1))) { col = (int)j; return false; }
            }
            return true;
        }
        if (is_bin) {
            for (size_t j = 0; j < s.size(); j++) {
                reduce(s[j]);
                if (ne(s[j], S(0)) && ne(s[j], S(1))) { col = (int)j; return false; }
            }
            return true;
        }
        return all_int(s, col);
    }

    void keep_best(const std::vector<S> & s, S v, bool is_max)
    {
        if (!have_best || (is_max ? lt(best_v, v) : gt(best_v, v))) {
            best_sol = s; best_v = v; have_best = true;
        }
    }

    int node(const Problem<S> & Q, bool is_max, bool is_bin, std::vector<int> & forks,
             S & v, std::vector<S> & sol)                                  // lpsol.h:2427-2612
    {
        nodes++;
        spec_chain = 1;                                                    // (a node that does not branch: its own LP)
        DepthGuard depth_guard(*this);
        if (Q.leq.r > max_leq_rows) max_leq_rows = Q.leq.r;
        int st = is_max ? six_maxm(Q, 10000u, v, sol) : six_minm(Q, 10000u, v, sol);
        if (st < 0) return st;
        if (st != SIX_SUCC) {
            if (st == SIX_UNBOUND) return IP_UNBOUND;
            if (st == SIX_TIME_OUT) return ORC_REF_UNDEFINED;             // UNREACH() at :2461-2463
            return IP_NO_PRI_FEASIBLE_SOL;
        }
        int col = 0;
        if (satisfied(sol, is_bin, col)) return IP_SUCC;
        if (have_best) {                                                   // :2474-2485
            if (is_max ? le(v, best_v) : ge(v, best_v)) return IP_NO_BETTER_THAN_BEST_SOL;
        }
        if (forks[col] >= 1) return IP_NO_PRI_FEASIBLE_SOL;                // :2486-2496
        forks[col]++;
        int lo = 0, hi = 1;
        Problem<S> L = Q;                                                  // floor branch
        if (is_bin) {
            if (L.eq.r == 0) L.eq = Mat<S>(0, Q.cols);
            grow_row(L.eq, Q.cols); L.eq.at(L.eq.r - 1, col) = S(1); L.eq.at(L.eq.r - 1, rhs0) = S(lo);
        } else {
            if (!int_cast_defined(sol[col])) return ORC_REF_UNDEFINED;   // int32 division by zero
            lo = to_int(sol[col]); hi = lo + 1;
            if (L.leq.r == 0) L.leq = Mat<S>(0, Q.cols);
            grow_row(L.leq, Q.cols); L.leq.at(L.leq.r - 1, col) = S(1); L.leq.at(L.leq.r - 1, rhs0) = S(lo);
        }
        std::vector<S> keep_sol; S keep_v; bool kept = false;
        st = node(L, is_max, is_bin, forks, v, sol);
        const long chain_floor = spec_chain;
        if (st < 0) return st;
        if (st == IP_SUCC) { keep_sol = sol; keep_v = v; kept = keep_sol.size() != 0; keep_best(sol, v, is_max); }
        Problem<S> H = Q;                                                  // ceiling branch
        if (is_bin) {
            if (H.eq.r == 0) H.eq = Mat<S>(0, Q.cols);
            grow_row(H.eq, Q.cols); H.eq.at(H.eq.r - 1, col) = S(1); H.eq.at(H.eq.r - 1, rhs0) = S(hi);
        } else {
            if (H.leq.r == 0) H.leq = Mat<S>(0, Q.cols);
            grow_row(H.leq, Q.cols); H.leq.at(H.leq.r - 1, col) = S(-1); H.leq.at(H.leq.r - 1, rhs0) = S(-hi);
        }
        st = node(H, is_max, is_bin, forks, v, sol);
        spec_chain = 1 + chain_floor + (spec_chain - 1);                   // this node's LP, the floor subtree, the ceiling subtree less its root's LP
        if (st < 0) return st;
        if (st == IP_SUCC) {                                               // :2563-2592
            if (kept && (is_max ? gt(keep_v, v) : lt(keep_v, v))) { v = keep_v; sol = keep_sol; }
            keep_best(sol, v, is_max);
            return IP_SUCC;
        }
        if (kept) { v = keep_v; sol = keep_sol; keep_best(sol, v, is_max); return IP_SUCC; } // :2593-2608
        return st;
    }

    static void grow_row(Mat<S> & m, int cols)
    {
        if (m.r == 0) { m = Mat<S>(1, cols); return; }
        append_rows(m, 1);
    }
};

template <class S>
int mip_solve(const Problem<S> & Q, bool is_max, bool is_bin, const uint8_t * allow_rational,
              S & v, std::vector<S> & sol, long * nodes = 0, int * max_leq_rows = 0, int * max_depth_out = 0)
{
    Mip<S> M;
    M.allow_rational = allow_rational;
    M.rhs0 = Q.cols - 1;
    std::vector<int> forks(Q.cols, 0);
    v = S(0);
    int st = M.node(Q, is_max, is_bin, forks, v, sol);
    if (nodes) *nodes = M.nodes;
    if (max_leq_rows) { max_leq_rows[0] = M.max_leq_rows; }
    if (max_depth_out) { max_depth_out[0] = M.max_depth; max_depth_out[1] = (int)M.spec_chain; }
    return st;
}

} // namespace orc
#endif
