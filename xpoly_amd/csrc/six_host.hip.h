// Host-side mirror of the problem reshaping SIX does around its pivot loop:
// SIX::normalize / convertEq2Ineq / calcDualMaxm / calcFinalSolution
// (src/com/lpsol.h:1290-1394, :1197-1278, :1586-1655, :1851-1899). These are
// O(rows x cols) one-shot copies on the caller's host buffers; every pivot,
// ratio test and pricing scan of the solve itself runs on the GPU: small
// problems through the LDS-resident batch kernel, large ones through Lp<S>.
#pragma once
#include <vector>
#include <stdlib.h>
#include <string.h>
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
struct FoldStep { int j, at; };           // convertEq2Ineq: equality `at` is substituted for variable j (lpsol.h:1209-1252)
template <class S> struct NormalForm {
    int n0, n, cols;                      // original / normalised variable counts
    int rows;                             // inequalities after normalisation
    const S * Np;                         // rows x (n + 1), row-major: N's cells, or -- no equalities, no free variable: nothing
                                          // to reshape -- the CALLER's inequalities where they lie (round 5: a 4096 x 8193 system
                                          // was copied three times on the host, 268 MB each, before its upload); NULL: the cells
                                          // have not been made (the HBM route makes them on the device, normalize_device)
    HostMat<S> N;                         // owns the cells unless Np is the caller's
    std::vector<S> obj, vcd, vcr;         // objective (n + 1), vc(i,i), vc(i,rhs)
    std::vector<int> free_var;
    std::vector<FoldStep> steps;          // the substitutions convertEq2Ineq makes, in its order (they depend on eq alone)
    std::vector<int> rest;                // equalities it keeps as a pair of opposite inequalities
    int leq_rows, eq_rows;
    bool plain_vc;                        // every variable constraint is exactly -x_i <= 0
    bool fits_lds(bool is_max) const
    {
        const int R = is_max ? rows : n, V = is_max ? n : rows;
        return plain_vc && small_lds_bytes<S>(R, V) <= 64 * 1024;
    }
};

// What SIX::normalize will do, from vc and eq alone -- sizes, the free variables, the substitution steps, the objective and the
// variable constraints of the normal form; the CELLS are made by normalize_cells_host (small problems: the LDS route) or on
// the device (normalize_device below: the HBM route), unless there is nothing to make (Np = the caller's leq).
template <class S>
int normalize_plan(const S * tgtf, const S * vc, int vc_rows, const S * eqs, int eq_rows, const S * leq,
                   int leq_rows, int cols, NormalForm<S> & F)
{
    if (!tgtf || !vc || cols < 2 || vc_rows != cols - 1 || eq_rows < 0 || leq_rows < 0 ||
        (eq_rows == 0 && leq_rows == 0) || (eq_rows > 0 && !eqs) || (leq_rows > 0 && !leq))
        return XPG_ERR_SHAPE;
    const int n0 = cols - 1;
    F.free_var.clear(); F.steps.clear(); F.rest.clear();
    F.leq_rows = leq_rows; F.eq_rows = eq_rows;
    {   // lpsol.h:1321-1339: a column of vc without a nonzero = a free variable. The diagonal first: with the usual -x_j <= 0 rows it
        // settles a column in one read instead of a strided scan of a matrix that is 537 MB at 8192 variables; the columns it leaves
        // open are then scanned TOGETHER, row by row, the next rows' cells requested ahead (every read of such a scan misses the
        // cache: 32 free variables of 4096 were 1.0 of the 1.5 ms of this plan one column after the other). The answer is the scan's.
        std::vector<int> cand;
        for (int j = 0; j < n0; j++) {
            if (j + 8 < n0) __builtin_prefetch(&vc[(size_t)(j + 8) * cols + (j + 8)]);
            if (eq(vc[(size_t)j * cols + j], zero<S>())) cand.push_back(j);
        }
        std::vector<char> open(cand.size(), 1);
        size_t left = cand.size();
        for (int i = 0; i < vc_rows && left > 0; i++) {
            if (i + 4 < vc_rows) for (size_t c = 0; c < cand.size(); c++) if (open[c]) __builtin_prefetch(&vc[(size_t)(i + 4) * cols + cand[c]]);
            for (size_t c = 0; c < cand.size(); c++)
                if (open[c] && !eq(vc[(size_t)i * cols + cand[c]], zero<S>())) { open[c] = 0; left--; }
        }
        for (size_t c = 0; c < cand.size(); c++) if (open[c]) F.free_var.push_back(cand[c]);
    }
    const int extra = (int)F.free_var.size(), n = n0 + extra;
    F.n0 = n0; F.n = n; F.cols = cols;
    F.obj.assign(n + 1, zero<S>()); F.vcd.assign(n, zero<S>()); F.vcr.assign(n, zero<S>());
    for (int j = 0; j < n0; j++) {
        if (j + 8 < n0) __builtin_prefetch(&vc[(size_t)(j + 8) * cols + n0]);
        F.obj[j] = tgtf[j]; F.vcd[j] = vc[(size_t)j * cols + j]; F.vcr[j] = vc[(size_t)j * cols + n0];
    }
    F.obj[n] = tgtf[n0];
    for (int k = 0; k < extra; k++) {                            // lpsol.h:1365-1392
        const int j = F.free_var[k], twin = n0 + k;
        F.vcd[j] = minus_one<S>(); F.vcd[twin] = minus_one<S>();
        F.obj[twin] = tgtf[j];
        scale_run(&F.obj[twin], 1, 1, minus_one<S>());
    }
    // convertEq2Ineq's choices (lpsol.h:1209-1222): column by column, the one not yet used equality with a nonzero there
    std::vector<char> used((size_t)eq_rows, 0);
    if (eq_rows > 0 && leq_rows > 0) {
        for (int j = 0; j < n0; j++) {
            int hits = 0, at = 0;
            for (int i = 0; i < eq_rows; i++)
                if (!used[(size_t)i] && ne(eqs[(size_t)i * cols + j], zero<S>())) { hits++; at = i; }
            if (hits != 1) continue;
            used[(size_t)at] = 1;
            F.steps.push_back(FoldStep{j, at});
        }
    }
    for (int i = 0; i < eq_rows; i++) if (!used[(size_t)i]) F.rest.push_back(i);
    F.rows = leq_rows + 2 * (int)F.rest.size();
    F.N = HostMat<S>();
    F.Np = (eq_rows == 0 && extra == 0) ? leq : (const S *)0;    // nothing to fold, nothing to split: the normal form IS the caller's system
    if (F.rows == 0) return XPG_ERR_SHAPE;
    F.plain_vc = true;
    for (int j = 0; j < n && F.plain_vc; j++) F.plain_vc = eq(F.vcd[j], minus_one<S>()) && eq(F.vcr[j], zero<S>());
    return 0;
}

// The cells of the normal form on the host (the LDS route's small problems).
template <class S>
int normalize_cells_host(const S * eqs, const S * leq, NormalForm<S> & F)
{
    if (F.Np) return 0;
    const int cols = F.cols, n0 = F.n0, n = F.n, extra = (int)F.free_var.size();
    HostMat<S> L = F.leq_rows ? HostMat<S>(leq, F.leq_rows, cols) : HostMat<S>();
    const HostMat<S> E = F.eq_rows ? HostMat<S>(eqs, F.eq_rows, cols) : HostMat<S>();
    int rc = fold_eq(L, E, n0);
    if (rc) return rc;
    if (L.r == 0) return XPG_ERR_SHAPE;
    F.N = HostMat<S>(L.r, n + 1);
    for (int i = 0; i < L.r; i++) {
        for (int j = 0; j < n0; j++) F.N(i, j) = L(i, j);
        F.N(i, n) = L(i, n0);
    }
    for (int k = 0; k < extra; k++) {                            // lpsol.h:1365-1392
        const int j = F.free_var[k], twin = n0 + k;
        for (int i = 0; i < L.r; i++) F.N(i, twin) = L(i, j);
        scale_run(&F.N(0, twin), L.r, F.N.c, minus_one<S>());
    }
    F.rows = F.N.r; F.Np = F.N.a.data();
    return 0;
}
// plan + host cells: what callers that always work on host cells use (the MIP controller's nodes)
template <class S>
int normalize_host(const S * tgtf, const S * vc, int vc_rows, const S * eqs, int eq_rows, const S * leq,
                   int leq_rows, int cols, NormalForm<S> & F)
{
    const int rc = normalize_plan(tgtf, vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols, F);
    return rc ? rc : normalize_cells_host(eqs, leq, F);
}

// ---- the same cells made ON THE DEVICE (round 6: the HBM route; a 4096 x 8192 call with a hundred equalities spent seconds
// in fold_eq on one host core in front of a 13 ms device solve) ---------------------------------------------------------
// convertEq2Ineq's substitutions (lpsol.h:1224-1250). A substitution changes inequality q only, from its own cells and the
// equality's (which nothing changes): rows are independent, so ONE launch runs every step on every row -- a workgroup per
// row, thread t owning the columns k = t (mod 256), the step's coefficient L(q, j) handed round through LDS by the thread
// that owns column j. The reference reads the equality at the INEQUALITY's row index for the leading value (:1232); that is
// reproduced, and flagged once it would leave the row (the host form returns XPG_ERR_REF_UNDEFINED there).
template <class S> __global__ __launch_bounds__(256)
void k_fold_eq(S * __restrict__ L, int lrows, int cols, int rhs, const S * __restrict__ E, const FoldStep * __restrict__ steps, int nsteps,
               int * __restrict__ flag)
{
    __shared__ unsigned long long coef_bits;
    const int tid = (int)threadIdx.x;
    for (int q = (int)blockIdx.x; q < lrows; q += (int)gridDim.x) {
        S * row = L + (size_t)q * cols;
        for (int s = 0; s < nsteps; s++) {
            const int j = steps[s].j, at = steps[s].at;
            __syncthreads();
            if (tid == (j & 255)) coef_bits = to_bits(row[j]);
            __syncthreads();
            const S coef = from_bits<S>(coef_bits);
            if (eq(coef, zero<S>())) continue;
            if (q >= cols) { if (tid == 0) atomicOr(flag, 1); break; }
            const S * e = E + (size_t)at * cols;
            const S lead = e[q];
            const bool rescale = ne(lead, one<S>());
            const S inv = div(one<S>(), lead);
            const int m1 = rescale ? scale_mode(inv) : (int)SCALE_KEEP, m2 = scale_mode(coef);
            for (int k = tid; k < cols; k += 256) {
                S t = scaled(scaled(e[k], inv, m1), coef, m2);
                const S cur = k == j ? zero<S>() : row[k];
                if (k >= rhs) t = neg(t);
                row[k] = add(t, cur);
            }
        }
        __syncthreads();
    }
}
// The normal form N [rows x (n + 1)] from the folded inequalities L [lrows x cols], the kept equalities as pairs -e / e
// (lpsol.h:1254-1268) and the twins of the free variables (-column, lpsol.h:1380-1386).
template <class S> __global__ __launch_bounds__(256)
void k_normal_form(const S * __restrict__ L, int lrows, int cols, const S * __restrict__ E, const int * __restrict__ rest, int nrest,
                   const int * __restrict__ free_var, int extra, S * __restrict__ N)
{
    const int rows = lrows + 2 * nrest, n0 = cols - 1, n = n0 + extra;
    const S m1 = minus_one<S>();
    const int mode = scale_mode(m1);
    for (int i = (int)blockIdx.x; i < rows; i += (int)gridDim.x) {
        const bool from_eq = i >= lrows;
        const S * src = from_eq ? E + (size_t)rest[(i - lrows) >> 1] * cols : L + (size_t)i * cols;
        const bool negate = from_eq && (((i - lrows) & 1) == 0);
        S * dst = N + (size_t)i * (n + 1);
        for (int c = (int)threadIdx.x; c <= n; c += 256) {
            const int sc = c < n0 ? c : (c == n ? n0 : free_var[c - n0]);
            S x = src[sc];
            if (negate) x = scaled(x, m1, mode);
            if (c >= n0 && c < n) x = scaled(x, m1, mode);
            dst[c] = x;
        }
    }
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
struct SixProfile { double reshape_ms, create_ms, upload_ms, dual_ms, solve_ms, read_ms, destroy_ms, total_ms; int route; unsigned pivots; };
inline SixProfile & six_profile() { static thread_local SixProfile p; return p; }
inline double six_now_ms()
{
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// SIX::normalize's cells on the device: the caller's inequalities and equalities go up as they lie, k_fold_eq and
// k_normal_form make N [F.rows x (F.n + 1)] in *dN (hipMalloc'ed here, the caller frees it). XPG_ERR_REF_UNDEFINED where the
// reference's substitution reads past an equality's row (lpsol.h:1232).
template <class S>
int normalize_device(xpg_ctx * ctx, const NormalForm<S> & F, const S * leq, const S * eqs, S ** dN)
{
    *dN = nullptr;
    const int cols = F.cols, lrows = F.leq_rows, erows = F.eq_rows, nst = (int)F.steps.size(), nrest = (int)F.rest.size(), extra = (int)F.free_var.size();
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t bL = up((size_t)(lrows > 0 ? lrows : 1) * cols * sizeof(S)), bE = up((size_t)(erows > 0 ? erows : 1) * cols * sizeof(S));
    const size_t bS = up((size_t)(nst > 0 ? nst : 1) * sizeof(FoldStep)), bR = up((size_t)(nrest > 0 ? nrest : 1) * 4), bF = up((size_t)(extra > 0 ? extra : 1) * 4);
    const size_t bN = up((size_t)F.rows * (F.n + 1) * sizeof(S));
    char * base = nullptr;
    hipError_t e = hipMalloc((void **)&base, bN + bL + bE + bS + bR + bF + 256);
    if (e != hipSuccess) { ctx->err = std::string("normalize on the device: ") + hipGetErrorString(e); (void)hipGetLastError(); return XPG_ERR_ALLOC; }
    S * dNN = (S *)base; S * dL = (S *)(base + bN); S * dE = (S *)(base + bN + bL);
    FoldStep * dS = (FoldStep *)(base + bN + bL + bE); int * dR = (int *)(base + bN + bL + bE + bS); int * dF = (int *)(base + bN + bL + bE + bS + bR);
    int * dflag = (int *)(base + bN + bL + bE + bS + bR + bF);
    hipStream_t st = ctx->stream;
    if (lrows > 0) e = hipMemcpyAsync(dL, leq, (size_t)lrows * cols * sizeof(S), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && erows > 0) e = hipMemcpyAsync(dE, eqs, (size_t)erows * cols * sizeof(S), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nst > 0) e = hipMemcpyAsync(dS, F.steps.data(), (size_t)nst * sizeof(FoldStep), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nrest > 0) e = hipMemcpyAsync(dR, F.rest.data(), (size_t)nrest * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && extra > 0) e = hipMemcpyAsync(dF, F.free_var.data(), (size_t)extra * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(dflag, 0, 4, st);
    int flag = 0;
    if (e == hipSuccess) {
        if (nst > 0 && lrows > 0)
            hipLaunchKernelGGL((k_fold_eq<S>), dim3(lrows < 65535 ? lrows : 65535), dim3(256), 0, st, dL, lrows, cols, F.n0, dE, dS, nst, dflag);
        hipLaunchKernelGGL((k_normal_form<S>), dim3(F.rows < 65535 ? F.rows : 65535), dim3(256), 0, st, dL, lrows, cols, dE, dR, nrest, dF, extra, dNN);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, dflag, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);                  // (the steps / rest / free_var vectors are the caller's: read by now)
    if (e != hipSuccess) { ctx->err = std::string("normalize on the device: ") + hipGetErrorString(e); (void)hipGetLastError(); (void)hipFree(base); return XPG_ERR_HIP; }
    if (flag) { (void)hipFree(base); return XPG_ERR_REF_UNDEFINED; }
    *dN = dNN;
    return 0;
}

// The HBM-resident route for problems that do not fit one CU's LDS. The system goes up ONCE, from where the caller holds it:
// as it lies when there is nothing to reshape (F.Np), else through normalize_device; the dual of minm is built on the device.
template <class S>
int solve_large(xpg_ctx * ctx, int kind, bool is_max, const NormalForm<S> & F, const S * leq, const S * eqs, unsigned max_iter, std::vector<S> & y)
{
    SixProfile & pf = six_profile();
    const int n = F.n, mm = F.rows;
    xpg_lp * lp = 0;
    int rc;
    double t0 = six_now_ms();
    S * devN = nullptr;                                          // the normal form made on the device (F.Np == NULL)
    if (!F.Np) {
        rc = normalize_device(ctx, F, leq, eqs, &devN);          // (its time is part of create_ms / dual_ms below)
        if (rc) return rc;
    }
    if (is_max) {
        rc = devN ? xpg_lp_create(ctx, kind, devN, mm, n + 1, F.obj.data(), F.vcd.data(), F.vcr.data(), 2, &lp)
                  : xpg_lp_create(ctx, kind, F.Np, mm, n + 1, F.obj.data(), F.vcd.data(), F.vcr.data(), 0, &lp);
        if (devN) (void)hipFree(devN);
        if (rc) return rc;
        pf.create_ms = six_now_ms() - t0;
    } else {                                                     // SIX::calcDualMaxm, lpsol.h:1602-1629
        std::vector<S> pobj(mm + 1, zero<S>());
        hipError_t e = hipSuccess;
        if (devN) {                                              // the constant column of N: one strided copy down
            e = hipMemcpy2DAsync(pobj.data(), sizeof(S), devN + n, (size_t)(n + 1) * sizeof(S), sizeof(S), mm, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        } else
            for (int j = 0; j < mm; j++) pobj[j] = F.Np[(size_t)j * (n + 1) + n];
        scale_run(pobj.data(), mm + 1, 1, minus_one<S>());
        const size_t cells = (size_t)mm * (n + 1), pcells = (size_t)n * (mm + 1);
        // (one allocation: the system as uploaded, its dual, the objective)
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        char * dbase = nullptr;
        S * dN = nullptr; S * dP = nullptr; S * dobj = nullptr;
        if (e == hipSuccess) e = hipMalloc((void **)&dbase, (devN ? 0 : up(cells * sizeof(S))) + up(pcells * sizeof(S)) + up((size_t)(n + 1) * sizeof(S)));
        if (e == hipSuccess) {
            dP = (S *)dbase; dobj = (S *)(dbase + up(pcells * sizeof(S)));
            dN = devN ? devN : (S *)(dbase + up(pcells * sizeof(S)) + up((size_t)(n + 1) * sizeof(S)));
        }
        if (e == hipSuccess && !devN) e = hipMemcpyAsync(dN, F.Np, cells * sizeof(S), hipMemcpyHostToDevice, ctx->stream);
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
        if (devN) (void)hipFree(devN);
        if (rc) return rc;
    }
    t0 = six_now_ms();
    int st = xpg_lp_two_stage(lp, max_iter);
    pf.solve_ms = six_now_ms() - t0;
    (void)xpg_lp_pivots_done(lp, &pf.pivots);                    // pivots of the loop that ended the solve (after phase 1, if any)
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

// Test view of SIX::normalize's two implementations (six_host.hip.h): the cells normalize_device makes in HBM and the cells
// normalize_cells_host makes (the LDS route's, which every small-LP parity test exercises), for the same input.
template <class S>
int test_normalize(xpg_ctx * ctx, const S * tgtf, const S * vc, int vc_rows, const S * eq, int eq_rows, const S * leq, int leq_rows,
                          int cols, S * out_dev, S * out_host, long long cap_cells, int32_t * out_info)
{
    NormalForm<S> F;
    int rc = normalize_plan(tgtf, vc, vc_rows, eq, eq_rows, leq, leq_rows, cols, F);
    if (rc) return rc;
    out_info[0] = F.rows; out_info[1] = F.n; out_info[2] = (int32_t)F.steps.size(); out_info[3] = (int32_t)F.rest.size();
    out_info[4] = (int32_t)F.free_var.size();
    const size_t cells = (size_t)F.rows * (F.n + 1);
    if ((long long)cells > cap_cells) return XPG_ERR_SHAPE;
    if (F.Np) { out_info[5] = out_info[6] = 0; memcpy(out_dev, F.Np, cells * sizeof(S)); memcpy(out_host, F.Np, cells * sizeof(S)); return 0; }
    S * dN = nullptr;
    out_info[5] = normalize_device(ctx, F, leq, eq, &dN);
    if (out_info[5] == 0) {
        hipError_t e = hipMemcpyAsync(out_dev, dN, cells * sizeof(S), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(dN);
        if (e != hipSuccess) return XPG_ERR_HIP;
    }
    out_info[6] = normalize_cells_host(eq, leq, F);
    if (out_info[6] == 0) memcpy(out_host, F.Np, cells * sizeof(S));
    return 0;
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
    int rc = normalize_plan(tgtf, vc, vc_rows, eqs, eq_rows, leq, leq_rows, cols, F);
    if (rc) return rc;
    pf.reshape_ms = six_now_ms() - t_in;
    std::vector<S> y;
    const char * force = xpg_env("XPG_FORCE_DEVICE_LP");          // test hook: always take the HBM-resident path
    if (!(force && force[0] == '1') && F.fits_lds(is_max)) {
        // dependence-test / branch-and-bound node sizes: one launch of the LDS-resident batch
        // kernel with nb = 1 (it builds the dual itself for minm)
        int32_t st1 = 0; S v1 = zero<S>();
        std::vector<S> raw(F.n + 1, zero<S>());
        pf.route = 1;
        rc = normalize_cells_host(eqs, leq, F);                  // (a few KB: SIX::normalize's copies on the host)
        if (rc) return rc;
        pf.reshape_ms = six_now_ms() - t_in;
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
        int st = solve_large(ctx, kind, is_max, F, leq, eqs, max_iter, y);
        pf.total_ms = six_now_ms() - t_in;
        if (st != XPG_SIX_SUCC) return st;
    }
    finish_host(F, tgtf, y, out_v, out_sol);
    pf.total_ms = six_now_ms() - t_in;
    return XPG_SIX_SUCC;
}

} // namespace xpg
