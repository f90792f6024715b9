// Host-side mirror of the problem reshaping SIX does around its pivot loop:
// SIX::normalize / convertEq2Ineq / calcDualMaxm / calcFinalSolution
// (src/com/lpsol.h:1290-1394, :1197-1278, :1586-1655, :1851-1899). These are
// O(rows x cols) one-shot copies on the caller's host buffers; every pivot,
// ratio test and pricing scan of the solve itself runs on the GPU: small
// problems through the LDS-resident batch kernel, large ones through Lp<S>.
#pragma once
#include <vector>
#include <stdlib.h>
#include "ctx.hip.h"
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

// The result of SIX::normalize (lpsol.h:1290-1394): inequalities only, every variable
// non-negative, free variables split v = v' - v''.
template <class S> struct NormalForm {
    int n0, n, cols;                      // original / normalised variable counts
    HostMat<S> N;                         // rows x (n + 1)
    std::vector<S> obj, vcd, vcr;         // objective (n + 1), vc(i,i), vc(i,rhs)
    std::vector<int> free_var;
    bool plain_vc;                        // every variable constraint is exactly -x_i <= 0
    bool fits_lds(bool is_max) const
    {
        const int R = is_max ? N.r : n, V = is_max ? n : N.r;
        return plain_vc && small_lds_bytes<S>(R, V) <= 64 * 1024;
    }
};

template <class S>
int normalize_host(const S * tgtf, const S * vc, int vc_rows, const S * eqs, int eq_rows, const S * leq,
                   int leq_rows, int cols, NormalForm<S> & F)
{
    if (!tgtf || !vc || cols < 2 || vc_rows != cols - 1 || eq_rows < 0 || leq_rows < 0 ||
        (eq_rows == 0 && leq_rows == 0) || (eq_rows > 0 && !eqs) || (leq_rows > 0 && !leq))
        return XPG_ERR_SHAPE;
    const int n0 = cols - 1;
    HostMat<S> L = leq_rows ? HostMat<S>(leq, leq_rows, cols) : HostMat<S>();
    const HostMat<S> E = eq_rows ? HostMat<S>(eqs, eq_rows, cols) : HostMat<S>();
    int rc = fold_eq(L, E, n0);
    if (rc) return rc;
    if (L.r == 0) return XPG_ERR_SHAPE;
    F.free_var.clear();
    for (int j = 0; j < n0; j++) {                               // lpsol.h:1321-1339
        bool all_zero = true;
        for (int i = 0; i < vc_rows && all_zero; i++) all_zero = eq(vc[(size_t)i * cols + j], zero<S>());
        if (all_zero) F.free_var.push_back(j);
    }
    const int extra = (int)F.free_var.size(), n = n0 + extra;
    F.n0 = n0; F.n = n; F.cols = cols;
    F.N = HostMat<S>(L.r, n + 1);
    F.obj.assign(n + 1, zero<S>()); F.vcd.assign(n, zero<S>()); F.vcr.assign(n, zero<S>());
    for (int i = 0; i < L.r; i++) {
        for (int j = 0; j < n0; j++) F.N(i, j) = L(i, j);
        F.N(i, n) = L(i, n0);
    }
    for (int j = 0; j < n0; j++) { F.obj[j] = tgtf[j]; F.vcd[j] = vc[(size_t)j * cols + j]; F.vcr[j] = vc[(size_t)j * cols + n0]; }
    F.obj[n] = tgtf[n0];
    for (int k = 0; k < extra; k++) {                            // lpsol.h:1365-1392
        const int j = F.free_var[k], twin = n0 + k;
        F.vcd[j] = minus_one<S>(); F.vcd[twin] = minus_one<S>();
        for (int i = 0; i < L.r; i++) F.N(i, twin) = L(i, j);
        scale_run(&F.N(0, twin), L.r, F.N.c, minus_one<S>());
        F.obj[twin] = tgtf[j];
        scale_run(&F.obj[twin], 1, 1, minus_one<S>());
    }
    F.plain_vc = true;
    for (int j = 0; j < n && F.plain_vc; j++) F.plain_vc = eq(F.vcd[j], minus_one<S>()) && eq(F.vcr[j], zero<S>());
    return 0;
}

// SIX::calcFinalSolution (lpsol.h:1851-1899) from the raw values y[0..n) of the normalised
// variables: undo the free-variable split, recompute the objective on the ORIGINAL tgtf.
template <class S>
void finish_host(const NormalForm<S> & F, const S * tgtf, std::vector<S> y, S * out_v, S * out_sol)
{
    y.resize(F.n + 1, zero<S>());
    for (size_t k = 0; k < F.free_var.size(); k++) y[F.free_var[k]] = sub(y[F.free_var[k]], y[F.n0 + (int)k]);
    S v = zero<S>();
    std::vector<S> sol(F.cols);
    for (int j = 0; j < F.n0; j++) sol[j] = y[j];
    sol[F.n0] = one<S>();
    for (int j = 0; j < F.cols; j++) v = add(v, mul(sol[j], tgtf[j]));
    reduce(v);
    *out_v = v;
    if (out_sol) for (int j = 0; j < F.cols; j++) { reduce(sol[j]); out_sol[j] = sol[j]; }
}

// The HBM-resident route for problems that do not fit one CU's LDS.
template <class S>
int solve_large(xpg_ctx * ctx, int kind, bool is_max, const NormalForm<S> & F, unsigned max_iter, std::vector<S> & y)
{
    const int n = F.n;
    HostMat<S> P; std::vector<S> pobj, pd, pr;
    if (is_max) { P = F.N; pobj = F.obj; pd = F.vcd; pr = F.vcr; }
    else {                                                       // SIX::calcDualMaxm, lpsol.h:1602-1629
        const int mm = F.N.r;
        P = HostMat<S>(n, mm + 1);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < mm; j++) P(i, j) = F.N(j, i);
        scale_run(P.a.data(), (int)P.a.size(), 1, minus_one<S>());
        for (int i = 0; i < n; i++) P(i, mm) = F.obj[i];
        pobj.assign(mm + 1, zero<S>());
        for (int j = 0; j < mm; j++) pobj[j] = F.N(j, n);
        scale_run(pobj.data(), mm + 1, 1, minus_one<S>());
        pd.assign(mm, minus_one<S>()); pr.assign(mm, zero<S>());
    }
    xpg_lp * lp = 0;
    int rc = xpg_lp_create(ctx, kind, P.a.data(), P.r, P.c, pobj.data(), pd.data(), pr.data(), 0, &lp);
    if (rc) return rc;
    int st = xpg_lp_two_stage(lp, max_iter);
    if (st != XPG_SIX_SUCC) { xpg_lp_destroy(lp); return st; }
    int rows, W, rhs;
    xpg_lp_shape(lp, &rows, &W, &rhs);
    std::vector<S> x(W), fobj(W);
    rc = xpg_lp_read(lp, 0, fobj.data(), 0, 0, 0, 0, 0, x.data());
    xpg_lp_destroy(lp);
    if (rc) return rc;
    if (is_max) y.assign(x.begin(), x.begin() + n);
    else {                                                       // lpsol.h:1713-1716
        const int nd = F.N.r;
        y.assign(n, zero<S>());
        for (int k = 0; k < n; k++) y[k] = neg(fobj[nd + k]);
    }
    return XPG_SIX_SUCC;
}

// SIX::maxm / minm (lpsol.h:1993-2033, :1662-1732).
template <class S>
int six_solve(xpg_ctx * ctx, int kind, bool is_max, const S * tgtf, const S * vc, int vc_rows,
              const S * eqs, int eq_rows, const S * leq, int leq_rows, int cols, unsigned max_iter,
              S * out_v, S * out_sol)
{
    if (!ctx || !out_v) return XPG_ERR_SHAPE;
    *out_v = zero<S>();
    NormalForm<S> F;
    int rc = normalize_host(tgtf, vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols, F);
    if (rc) return rc;
    std::vector<S> y;
    const char * force = getenv("XPG_FORCE_DEVICE_LP");          // test hook: always take the HBM-resident path
    if (!(force && force[0] == '1') && F.fits_lds(is_max)) {
        // dependence-test / branch-and-bound node sizes: one launch of the LDS-resident batch
        // kernel with nb = 1 (it builds the dual itself for minm)
        int32_t st1 = 0; S v1 = zero<S>();
        std::vector<S> raw(F.n + 1, zero<S>());
        rc = batch_host<S>(ctx, is_max ? 1 : 0, 1, F.obj.data(), F.N.a.data(), F.N.r, F.n + 1, max_iter, &st1, &v1,
                           raw.data(), /*raw_sol=*/1);
        if (rc) return rc;
        if (st1 != XPG_SIX_SUCC) return st1;
        y.assign(raw.begin(), raw.begin() + F.n);
    } else {
        int st = solve_large(ctx, kind, is_max, F, max_iter, y);
        if (st != XPG_SIX_SUCC) return st;
    }
    finish_host(F, tgtf, y, out_v, out_sol);
    return XPG_SIX_SUCC;
}

} // namespace xpg
