// Host-side mirror of the problem reshaping SIX does around its pivot loop:
// SIX::normalize / convertEq2Ineq / calcDualMaxm / calcFinalSolution
// (src/com/lpsol.h:1290-1394, :1197-1278, :1586-1655, :1851-1899). These are
// O(rows x cols) one-shot copies on the caller's host buffers; every pivot,
// ratio test and pricing scan of the solve itself runs on the GPU through Lp<S>.
#pragma once
#include <vector>
#include <stdlib.h>
#include "lp_host.hip.h"
#include "batch_kernels.hip.h"

namespace xpg {

template <class S> struct HostMat {
    int r, c;
    std::vector<S> a;
    HostMat() : r(0), c(0) {}
    HostMat(int rows, int cols) : r(rows), c(cols), a((size_t)rows * cols, zero<S>()) {}
    HostMat(const S * p, int rows, int cols) : r(rows), c(cols), a(p, p + (size_t)rows * cols) {}
    S & operator()(int i, int j) { return a[(size_t)i * c + j]; }
    const S & operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};

// One strided run of cells times a scalar with Matrix::mul's shortcuts.
template <class S> inline void scale_run(S * p, int n, int stride, S x)
{
    const int mode = scale_mode(x);
    if (mode == SCALE_KEEP) return;
    for (int k = 0; k < n; k++) p[(size_t)k * stride] = scaled(p[(size_t)k * stride], x, mode);
}

// SIX::convertEq2Ineq (lpsol.h:1197-1278): an equality whose column j is the
// only nonzero among the remaining equalities is substituted into the
// inequalities; the rest become a pair of opposite inequalities. The reference
// reads the equality row at the *inequality's row index* (lpsol.h:1232); that is
// reproduced, and refused once it would leave the row.
template <class S> int fold_eq(HostMat<S> & L, const HostMat<S> & E, int rhs)
{
    if (E.r == 0) return 0;
    std::vector<char> used(E.r, 0);
    int remaining = E.r;
    if (L.r > 0) {
        for (int j = 0; j < rhs; j++) {
            int hits = 0, at = 0;
            for (int i = 0; i < E.r; i++)
                if (!used[i] && ne(E(i, j), zero<S>())) { hits++; at = i; }
            if (hits != 1) continue;
            used[at] = 1; remaining--;
            for (int q = 0; q < L.r; q++) {
                const S coef = L(q, j);
                if (eq(coef, zero<S>())) continue;
                if (q >= E.c) return XPG_ERR_REF_UNDEFINED;
                std::vector<S> t(&E.a[(size_t)at * E.c], &E.a[(size_t)at * E.c] + E.c);
                const S lead = t[q];
                if (ne(lead, one<S>())) scale_run(t.data(), E.c, 1, div(one<S>(), lead));
                scale_run(t.data(), E.c, 1, coef);
                L(q, j) = zero<S>();
                for (int k = rhs; k < E.c; k++) t[k] = neg(t[k]);
                for (int k = 0; k < E.c; k++) L(q, k) = add(t[k], L(q, k));
            }
        }
    }
    if (remaining > 0) {
        const int base = L.r;
        HostMat<S> G(base + 2 * remaining, E.c);
        for (size_t k = 0; k < L.a.size(); k++) G.a[k] = L.a[k];
        int at = base;
        for (int i = 0; i < E.r; i++) {
            if (used[i]) continue;
            for (int k = 0; k < E.c; k++) { G(at, k) = E(i, k); G(at + 1, k) = E(i, k); }
            scale_run(&G(at, 0), E.c, 1, minus_one<S>());
            at += 2;
        }
        L = G;
    }
    return 0;
}

template <class S>
int six_solve(xpg_ctx * ctx, int kind, bool is_max, const S * tgtf, const S * vc, int vc_rows,
              const S * eqs, int eq_rows, const S * leq, int leq_rows, int cols, unsigned max_iter,
              S * out_v, S * out_sol)
{
    if (!ctx || !tgtf || !vc || !out_v || cols < 2 || vc_rows != cols - 1 || eq_rows < 0 ||
        leq_rows < 0 || (eq_rows == 0 && leq_rows == 0) || (eq_rows > 0 && !eqs) ||
        (leq_rows > 0 && !leq))
        return XPG_ERR_SHAPE;
    *out_v = zero<S>();
    const int n0 = cols - 1;
    HostMat<S> L = leq_rows ? HostMat<S>(leq, leq_rows, cols) : HostMat<S>();
    const HostMat<S> E = eq_rows ? HostMat<S>(eqs, eq_rows, cols) : HostMat<S>();
    int rc = fold_eq(L, E, n0);
    if (rc) return rc;
    if (L.r == 0) return XPG_ERR_SHAPE;

    // ---- free variables: v = v' - v'' (lpsol.h:1321-1392)
    std::vector<int> free_var;
    for (int j = 0; j < n0; j++) {
        bool all_zero = true;
        for (int i = 0; i < vc_rows && all_zero; i++) all_zero = eq(vc[(size_t)i * cols + j], zero<S>());
        if (all_zero) free_var.push_back(j);
    }
    const int extra = (int)free_var.size(), n = n0 + extra;
    HostMat<S> N(L.r, n + 1);
    std::vector<S> obj(n + 1, zero<S>()), vcd(n, zero<S>()), vcr(n, zero<S>());
    for (int i = 0; i < L.r; i++) {
        for (int j = 0; j < n0; j++) N(i, j) = L(i, j);
        N(i, n) = L(i, n0);
    }
    for (int j = 0; j < n0; j++) { obj[j] = tgtf[j]; vcd[j] = vc[(size_t)j * cols + j]; vcr[j] = vc[(size_t)j * cols + n0]; }
    obj[n] = tgtf[n0];
    for (int k = 0; k < extra; k++) {
        const int j = free_var[k], twin = n0 + k;
        vcd[j] = minus_one<S>(); vcd[twin] = minus_one<S>();
        for (int i = 0; i < L.r; i++) N(i, twin) = L(i, j);
        scale_run(&N(0, twin), L.r, N.c, minus_one<S>());
        obj[twin] = tgtf[j];
        scale_run(&obj[twin], 1, 1, minus_one<S>());
    }

    // ---- small problems (the dependence-test and branch-and-bound node sizes): one launch of
    // the LDS-resident batch kernel with nb = 1, which builds the dual itself for minm
    bool plain_vc = true;
    for (int j = 0; j < n && plain_vc; j++) plain_vc = eq(vcd[j], minus_one<S>()) && eq(vcr[j], zero<S>());
    {
        const int R = is_max ? N.r : n, V = is_max ? n : N.r;
        const char * force = getenv("XPG_FORCE_DEVICE_LP");     // test hook: always take the HBM-resident path
        if (plain_vc && !(force && force[0] == '1') && small_lds_bytes<S>(R, V) <= 64 * 1024) {
            int32_t st1 = 0; S v1 = zero<S>();
            std::vector<S> raw(n + 1, zero<S>());
            rc = batch_host<S>(ctx, is_max ? 1 : 0, 1, obj.data(), N.a.data(), N.r, n + 1, max_iter, &st1, &v1,
                               raw.data(), /*raw_sol=*/1);
            if (rc) return rc;
            if (st1 != XPG_SIX_SUCC) return st1;
            std::vector<S> y(raw.begin(), raw.begin() + n);
            y.push_back(zero<S>());
            for (int k = 0; k < extra; k++) y[free_var[k]] = sub(y[free_var[k]], y[n0 + k]);
            S v = zero<S>();
            std::vector<S> sol(cols);
            for (int j = 0; j < n0; j++) sol[j] = y[j];
            sol[n0] = one<S>();
            for (int j = 0; j < cols; j++) v = add(v, mul(sol[j], tgtf[j]));
            reduce(v);
            *out_v = v;
            if (out_sol) for (int j = 0; j < cols; j++) { reduce(sol[j]); out_sol[j] = sol[j]; }
            return XPG_SIX_SUCC;
        }
    }

    // ---- the slack form handed to the GPU: primal for maxm, dual for minm
    HostMat<S> P; std::vector<S> pobj, pd, pr;
    if (is_max) { P = N; pobj = obj; pd = vcd; pr = vcr; }
    else {                                                       // lpsol.h:1602-1629
        const int mm = N.r;
        P = HostMat<S>(n, mm + 1);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < mm; j++) P(i, j) = N(j, i);
        scale_run(P.a.data(), (int)P.a.size(), 1, minus_one<S>());
        for (int i = 0; i < n; i++) P(i, mm) = obj[i];
        pobj.assign(mm + 1, zero<S>());
        for (int j = 0; j < mm; j++) pobj[j] = N(j, n);
        scale_run(pobj.data(), mm + 1, 1, minus_one<S>());
        pd.assign(mm, minus_one<S>()); pr.assign(mm, zero<S>());
    }
    xpg_lp * lp = 0;
    rc = xpg_lp_create(ctx, kind, P.a.data(), P.r, P.c, pobj.data(), pd.data(), pr.data(), 0, &lp);
    if (rc) return rc;
    int st = xpg_lp_two_stage(lp, max_iter);
    if (st != XPG_SIX_SUCC) { xpg_lp_destroy(lp); return st; }
    int rows, W, rhs;
    xpg_lp_shape(lp, &rows, &W, &rhs);
    std::vector<S> x(W), fobj(W);
    rc = xpg_lp_read(lp, 0, fobj.data(), 0, 0, 0, 0, 0, x.data());
    xpg_lp_destroy(lp);
    if (rc) return rc;
    std::vector<S> y;
    if (is_max) y = x;
    else {                                                       // lpsol.h:1713-1716
        const int nd = N.r;
        y.assign(n + 1, zero<S>());
        for (int k = 0; k < n; k++) y[k] = neg(fobj[nd + k]);
    }
    // ---- SIX::calcFinalSolution (lpsol.h:1851-1899)
    for (int k = 0; k < extra; k++) y[free_var[k]] = sub(y[free_var[k]], y[n0 + k]);
    S v = zero<S>();
    std::vector<S> sol(cols);
    for (int j = 0; j < n0; j++) sol[j] = y[j];
    sol[n0] = one<S>();
    for (int j = 0; j < cols; j++) v = add(v, mul(sol[j], tgtf[j]));
    reduce(v);
    *out_v = v;
    if (out_sol) for (int j = 0; j < cols; j++) { reduce(sol[j]); out_sol[j] = sol[j]; }
    return XPG_SIX_SUCC;
}

} // namespace xpg
