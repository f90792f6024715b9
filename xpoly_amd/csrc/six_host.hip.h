// Host-side mirror of the problem reshaping SIX does around its pivot loop:
// SIX::normalize / convertEq2Ineq / calcDualMaxm / calcFinalSolution
// (src/com/lpsol.h:1290-1394, :1197-1278, :1586-1655, :1851-1899). These are
// O(rows x cols) one-shot copies on the caller's host buffers; every pivot,
// ratio test and pricing scan of the solve itself runs on the GPU: small
// problems through the LDS-resident batch kernel, large ones through Lp<S>.
#pragma once
#include <vector>
#include <stdlib.h>
#include <time.h>
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
    int rows;                             // inequalities after normalisation
    const S * Np;                         // rows x (n + 1), row-major: N's cells, or -- no equalities, no free variable: nothing
                                          // to reshape -- the CALLER's inequalities where they lie (round 5: a 4096 x 8193 system
                                          // was copied three times on the host, 268 MB each, before its upload)
    HostMat<S> N;                         // owns the cells unless Np is the caller's
    std::vector<S> obj, vcd, vcr;         // objective (n + 1), vc(i,i), vc(i,rhs)
    std::vector<int> free_var;
    bool plain_vc;                        // every variable constraint is exactly -x_i <= 0
    bool fits_lds(bool is_max) const
    {
        const int R = is_max ? rows : n, V = is_max ? n : rows;
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
    F.free_var.clear();
    for (int j = 0; j < n0; j++) {                               // lpsol.h:1321-1339: a column of vc without a nonzero = a free variable
        // (the diagonal first: with the usual -x_j <= 0 rows it settles the column in one read instead of a strided scan of a
        // matrix that is 537 MB at 8192 variables; the answer is the scan's)
        bool all_zero = eq(vc[(size_t)j * cols + j], zero<S>());
        for (int i = 0; i < vc_rows && all_zero; i++) all_zero = eq(vc[(size_t)i * cols + j], zero<S>());
        if (all_zero) F.free_var.push_back(j);
    }
    const int extra = (int)F.free_var.size(), n = n0 + extra;
    F.n0 = n0; F.n = n; F.cols = cols;
    F.obj.assign(n + 1, zero<S>()); F.vcd.assign(n, zero<S>()); F.vcr.assign(n, zero<S>());
    for (int j = 0; j < n0; j++) { F.obj[j] = tgtf[j]; F.vcd[j] = vc[(size_t)j * cols + j]; F.vcr[j] = vc[(size_t)j * cols + n0]; }
    F.obj[n] = tgtf[n0];
    if (eq_rows == 0 && extra == 0) {
        // nothing to fold, nothing to split: the normal form IS the caller's system
        F.N = HostMat<S>(); F.rows = leq_rows; F.Np = leq;
    } else {
        HostMat<S> L = leq_rows ? HostMat<S>(leq, leq_rows, cols) : HostMat<S>();
        const HostMat<S> E = eq_rows ? HostMat<S>(eqs, eq_rows, cols) : HostMat<S>();
        int rc = fold_eq(L, E, n0);
        if (rc) return rc;
        if (L.r == 0) return XPG_ERR_SHAPE;
        F.N = HostMat<S>(L.r, n + 1);
        for (int i = 0; i < L.r; i++) {
            for (int j = 0; j < n0; j++) F.N(i, j) = L(i, j);
            F.N(i, n) = L(i, n0);
        }
        for (int k = 0; k < extra; k++) {                        // lpsol.h:1365-1392
            const int j = F.free_var[k], twin = n0 + k;
            F.vcd[j] = minus_one<S>(); F.vcd[twin] = minus_one<S>();
            for (int i = 0; i < L.r; i++) F.N(i, twin) = L(i, j);
            scale_run(&F.N(0, twin), L.r, F.N.c, minus_one<S>());
            F.obj[twin] = tgtf[j];
            scale_run(&F.obj[twin], 1, 1, minus_one<S>());
        }
        F.rows = F.N.r; F.Np = F.N.a.data();
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

// SIX::calcDualMaxm (lpsol.h:1602-1629) on the device: P = (-N^T | c), n x (mm + 1), from the mm x (n + 1) system N --
// every cell times -1 with Matrix::mul's own shortcuts and arithmetic (scaled(): the host path's cells exactly), tile by
// tile through LDS so that both sides are read and written in whole lines.
template <class S> __global__ __launch_bounds__(256) void k_dual_build(const S * __restrict__ N, int mm, int n, const S * __restrict__ obj, S * __restrict__ P)
{
    __shared__ unsigned long long tile[32][33];                          // (cells as bits: S has constructors)
    const S m1 = minus_one<S>();
    const int mode = scale_mode(m1);
    const int j0 = (int)blockIdx.x * 32, i0 = (int)blockIdx.y * 32;      // j: row of N (column of P), i: column of N (row of P)
    const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int j = j0 + k, i = i0 + tx;
        if (j < mm && i < n) tile[k][tx] = to_bits(N[(size_t)j * (n + 1) + i]);
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int i = i0 + k, j = j0 + tx;
        if (i < n && j < mm) P[(size_t)i * (mm + 1) + j] = scaled(from_bits<S>(tile[tx][k]), m1, mode);
    }
    if (blockIdx.x == 0) {                                               // the constant column of the dual: the primal objective
        const int i = i0 + (int)threadIdx.x;
        if (threadIdx.x < 32 && i < n) P[(size_t)i * (mm + 1) + mm] = obj[i];
    }
}

// where a call's time went (xpg_six_last_profile): milliseconds, host clock, synchronised at the marks
struct SixProfile { double reshape_ms, create_ms, upload_ms, dual_ms, solve_ms, read_ms, destroy_ms, total_ms; int route; };
inline SixProfile & six_profile() { static thread_local SixProfile p; return p; }
inline double six_now_ms()
{
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// The HBM-resident route for problems that do not fit one CU's LDS. The system goes up ONCE, from where the caller (or
// normalize_host) holds it; the dual of minm is built on the device.
template <class S>
int solve_large(xpg_ctx * ctx, int kind, bool is_max, const NormalForm<S> & F, unsigned max_iter, std::vector<S> & y)
{
    SixProfile & pf = six_profile();
    const int n = F.n, mm = F.rows;
    xpg_lp * lp = 0;
    int rc;
    double t0 = six_now_ms();
    if (is_max) {
        rc = xpg_lp_create(ctx, kind, F.Np, mm, n + 1, F.obj.data(), F.vcd.data(), F.vcr.data(), 0, &lp);
        if (rc) return rc;
        pf.create_ms = six_now_ms() - t0;
    } else {                                                     // SIX::calcDualMaxm, lpsol.h:1602-1629
        std::vector<S> pobj(mm + 1, zero<S>());
        for (int j = 0; j < mm; j++) pobj[j] = F.Np[(size_t)j * (n + 1) + n];
        scale_run(pobj.data(), mm + 1, 1, minus_one<S>());
        const size_t cells = (size_t)mm * (n + 1), pcells = (size_t)n * (mm + 1);
        // (one allocation: the system as uploaded, its dual, the objective)
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        char * dbase = nullptr;
        S * dN = nullptr; S * dP = nullptr; S * dobj = nullptr;
        hipError_t e = hipMalloc((void **)&dbase, up(cells * sizeof(S)) + up(pcells * sizeof(S)) + up((size_t)(n + 1) * sizeof(S)));
        if (e == hipSuccess) { dP = (S *)dbase; dN = (S *)(dbase + up(pcells * sizeof(S))); dobj = (S *)(dbase + up(pcells * sizeof(S)) + up(cells * sizeof(S))); }
        if (e == hipSuccess) e = hipMemcpyAsync(dN, F.Np, cells * sizeof(S), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dobj, F.obj.data(), (size_t)(n + 1) * sizeof(S), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((k_dual_build<S>), dim3((mm + 31) / 32, (n + 31) / 32), dim3(256), 0, ctx->stream, dN, mm, n, dobj, dP);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        pf.dual_ms = six_now_ms() - t0;
        if (e == hipSuccess) {
            std::vector<S> pd(mm, minus_one<S>()), pr(mm, zero<S>());
            t0 = six_now_ms();
            // (src_on_device = 2: the system is a device array, the objective a host one)
            rc = xpg_lp_create(ctx, kind, dP, n, mm + 1, pobj.data(), pd.data(), pr.data(), 2, &lp);
            pf.create_ms = six_now_ms() - t0;
        } else { ctx->err = std::string("dual on the device: ") + hipGetErrorString(e); rc = e == hipErrorOutOfMemory ? XPG_ERR_ALLOC : XPG_ERR_HIP; (void)hipGetLastError(); }
        if (dbase) (void)hipFree(dbase);
        if (rc) return rc;
    }
    t0 = six_now_ms();
    int st = xpg_lp_two_stage(lp, max_iter);
    pf.solve_ms = six_now_ms() - t0;
    if (st != XPG_SIX_SUCC) { t0 = six_now_ms(); xpg_lp_destroy(lp); pf.destroy_ms = six_now_ms() - t0; return st; }
    int rows, W, rhs;
    xpg_lp_shape(lp, &rows, &W, &rhs);
    std::vector<S> x(W), fobj(W);
    t0 = six_now_ms();
    rc = xpg_lp_read(lp, 0, fobj.data(), 0, 0, 0, 0, 0, x.data());
    pf.read_ms = six_now_ms() - t0;
    t0 = six_now_ms();
    xpg_lp_destroy(lp);
    pf.destroy_ms = six_now_ms() - t0;
    if (rc) return rc;
    if (is_max) y.assign(x.begin(), x.begin() + n);
    else {                                                       // lpsol.h:1713-1716
        const int nd = mm;
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
    SixProfile & pf = six_profile();
    pf = SixProfile();
    const double t_in = six_now_ms();
    NormalForm<S> F;
    int rc = normalize_host(tgtf, vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols, F);
    if (rc) return rc;
    pf.reshape_ms = six_now_ms() - t_in;
    std::vector<S> y;
    const char * force = getenv("XPG_FORCE_DEVICE_LP");          // test hook: always take the HBM-resident path
    if (!(force && force[0] == '1') && F.fits_lds(is_max)) {
        // dependence-test / branch-and-bound node sizes: one launch of the LDS-resident batch
        // kernel with nb = 1 (it builds the dual itself for minm)
        int32_t st1 = 0; S v1 = zero<S>();
        std::vector<S> raw(F.n + 1, zero<S>());
        pf.route = 1;
        const double t0 = six_now_ms();
        rc = batch_host<S>(ctx, is_max ? 1 : 0, 1, F.obj.data(), F.Np, F.rows, F.n + 1, max_iter, &st1, &v1,
                           raw.data(), /*raw_sol=*/1);
        pf.solve_ms = six_now_ms() - t0;
        pf.total_ms = six_now_ms() - t_in;
        if (rc) return rc;
        if (st1 != XPG_SIX_SUCC) return st1;
        y.assign(raw.begin(), raw.begin() + F.n);
    } else {
        pf.route = 2;
        int st = solve_large(ctx, kind, is_max, F, max_iter, y);
        pf.total_ms = six_now_ms() - t_in;
        if (st != XPG_SIX_SUCC) return st;
    }
    finish_host(F, tgtf, y, out_v, out_sol);
    pf.total_ms = six_now_ms() - t_in;
    return XPG_SIX_SUCC;
}

} // namespace xpg
