// Device-resident slack-form simplex for ONE large tableau (configs 2 and 4):
// every step of SIX::solveSlackForm's loop (src/com/lpsol.h:1039-1188) is a
// kernel template over the scalar S (xpg::F64 or xpg::R32); the host only
// queues launches and polls a status word every few dozen pivots.
//
// Per loop iteration two launches, communicating through LoopState:
//   k_pick   : pricing scan (lpsol.h:1054-1069) + ratio test (lpsol.h:553-663)
//              + pivot-pair table upkeep (lpsol.h:68-154) + staging of the pivot
//              (row * 1/pivot -> rowbuf, -column -> colbuf, objective row update,
//              basis swap: lpsol.h:1471-1474, :1485, :1496-1510) + look-ahead
//              pricing                                           -- 1 workgroup
//   k_update : a_ij += (-a_i,nv) * e_j for every i != r, all j   (lpsol.h:1481-1490)
//              -- the HBM-bound sweep, >= 2048 workgroups; it also exports the
//              look-ahead column and the constant column contiguously so the
//              next k_pick's ratio test reads 2 x 32 KB coalesced instead of
//              2 x 4096 64-byte sectors
// (k_prep is the multi-workgroup staging used by one-shot and phase-1 pivots.)
//
// HBM layout: tableau row-major, leading dimension ld (multiple of 16 elements
// = 128 B so every row starts on a cache line and 16-byte vector accesses are
// aligned), objective row / rowbuf / colbuf contiguous. The pivot-pair table is
// a bit matrix (n x ceil(n/32) words) with per-row and per-column counters so
// canBeNVCandidate / canBeBVCandidate (lpsol.h:124-153) are O(1).
#pragma once
#include "scalar.hip.h"
#include <limits.h>

namespace xpg {

enum { ST_RUNNING = -1000, ST_CHECK_OPT = -1001 };

struct LoopState {
    int status;            // ST_RUNNING, ST_CHECK_OPT or a final SIX_* code
    unsigned done;         // pivots performed in this solveSlackForm call ('cnt')
    unsigned max_iter;
    int row, col, leave;   // this iteration's pivot (row < 0: no pivot this time)
    unsigned long long cnv_bits;   // objective coefficient of the entering column
    unsigned long long piv_bits;   // pivot element
    int infeasible;        // set by the feasibility kernels
    unsigned total_pivots; // over the handle's lifetime (trace index)
    int aux;               // scratch result for phase-1 helper kernels
    // look-ahead pricing, filled by k_prep's atomics after the objective update:
    int next_first;        // lowest eligible entering column of the next iteration
                           // (INT_MAX: none, NF_UNKNOWN: not computed)
    int anypos;            // some nonbasic reduced cost is > 0
    int cached_col;        // column currently held in nextcol[] (-1: none), set by the sweep
    int bcol_valid;        // bcol[] holds the current constant column
};
enum { NF_UNKNOWN = -2 };

template <class S> struct LpView {
    S * tab; int m, W, ld, rhs;
    S * obj;
    uint8_t * nv; uint8_t * bv; int * bv2eq; int * eq2bv;
    uint32_t * ppt; int pw; int * rowcnt; int * colcnt;
    S * rowbuf; S * colbuf; S * x; S * vcd; S * vcr;
    S * nextcol; S * bcol;   // contiguous copies of the predicted entering column / constant column
    LoopState * st;
    int * trace; int trace_cap;
};

template <class S> __device__ __forceinline__ S from_bits(unsigned long long b)
{ S s; __builtin_memcpy(&s, &b, 8); return s; }
template <class S> __device__ __forceinline__ unsigned long long to_bits(S s)
{ unsigned long long b; __builtin_memcpy(&b, &s, 8); return b; }

template <class S> __device__ __forceinline__ S shfl_xor_s(S v, int mask)
{
    int w[2];
    __builtin_memcpy(w, &v, 8);
    w[0] = __shfl_xor(w[0], mask); w[1] = __shfl_xor(w[1], mask);
    __builtin_memcpy(&v, w, 8);
    return v;
}

// ---- arg-min with the reference's tie-break ("first row wins", lpsol.h:604-611)
template <class S> struct Cand { S q; int idx; };
template <class S> __device__ __forceinline__ Cand<S> better(Cand<S> a, Cand<S> b)
{
    if (b.idx == INT_MAX) return a;
    if (a.idx == INT_MAX) return b;
    if (gt(a.q, b.q)) return b;
    if (gt(b.q, a.q)) return a;
    return a.idx < b.idx ? a : b;
}
template <class S> __device__ Cand<S> block_argmin(Cand<S> c, Cand<S> * sh)
{
    for (int o = 32; o > 0; o >>= 1) {
        Cand<S> t; t.q = shfl_xor_s(c.q, o); t.idx = __shfl_xor(c.idx, o);
        c = better(c, t);
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = c;
    __syncthreads();
    Cand<S> r = sh[0];
    for (int k = 1; k < nw; k++) r = better(r, sh[k]);
    return r;
}
__device__ inline int block_min_int(int v, int * sh)
{
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    int r = sh[0];
    for (int k = 1; k < nw; k++) r = min(r, sh[k]);
    return r;
}

template <class S> __device__ __forceinline__ bool ppt_seen(const LpView<S> & v, int nv, int b)
{ return (v.ppt[(size_t)nv * v.pw + (b >> 5)] >> (b & 31)) & 1u; }

// SIX::findPivotBV (lpsol.h:553-663) by one workgroup: rows strided over
// threads, arg-min of b_i / a_i,nv with the lowest row winning ties.
// The column and the constant column come from the contiguous copies the
// previous sweep exported when they are current (col_cached / b_cached), else
// from the tableau (one 64-byte sector per element).
template <class S> __device__ int ratio_test(const LpView<S> & v, int nv, Cand<S> * sh,
                                             bool col_cached = false, bool b_cached = false)
{
    const int lim = v.rhs - 1;
    for (int pass = 0; pass < 2; pass++) {
        Cand<S> best; best.q = zero<S>(); best.idx = INT_MAX;
        for (int i = threadIdx.x; i < v.m; i += blockDim.x) {
            S a = col_cached ? v.nextcol[i] : v.tab[(size_t)i * v.ld + nv];
            if (pass == 0 ? le(a, zero<S>()) : eq(a, zero<S>())) continue;
            int b = v.eq2bv[i];
            if (ppt_seen(v, nv, b) || v.colcnt[b] >= lim) continue;
            Cand<S> c; c.q = div(b_cached ? v.bcol[i] : v.tab[(size_t)i * v.ld + v.rhs], a); c.idx = i;
            best = better(best, c);
        }
        best = block_argmin(best, sh);
        if (best.idx != INT_MAX) return v.eq2bv[best.idx];
    }
    return -1;
}

// In-workgroup pricing scan (lpsol.h:1054-1069 without the side effect): lowest
// nonbasic j with c_j > 0 whose pair-table row is still open, and whether any
// nonbasic c_j > 0 exists at all.
template <class S> __device__ void price_scan(const LpView<S> & v, int * sh_i, int * sh_flag,
                                              int & first, int & anypos)
{
    const int rhs = v.rhs, lim = rhs - 1;
    int f = INT_MAX, any = 0;
    for (int j = threadIdx.x; j < rhs; j += blockDim.x)
        if (v.nv[j] && gt(v.obj[j], zero<S>())) {
            any = 1;
            if (v.rowcnt[j] < lim) f = min(f, j);
        }
    first = block_min_int(f, sh_i);
    if (threadIdx.x == 0) *sh_flag = 0;
    __syncthreads();
    if (any) *sh_flag = 1;
    __syncthreads();
    anypos = *sh_flag;
}

// One workgroup per loop iteration, deliberately light on memory traffic (one CU
// moves only ~30 GB/s): it consumes the look-ahead pricing result k_prep left in
// LoopState, runs the ratio test (lpsol.h:553-663) on the contiguous column
// copies the last sweep exported, keeps the pair table (lpsol.h:68-154), swaps
// the basis (lpsol.h:1504-1510) and writes -column to colbuf (lpsol.h:1485).
// Every rare branch of solveSlackForm (optimum, findPivotNVandBVPair, relaxed
// ratio pass, disableNV) is handled here by the generic single-workgroup code.
template <class S> __global__ __launch_bounds__(1024) void k_pick(LpView<S> v)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<S>)];
    Cand<S> * sh_c = (Cand<S> *)sh_c_raw;
    __shared__ int sh_i[16];
    __shared__ int sh_flag;
    LoopState * st = v.st;
    if (st->status != ST_RUNNING) return;
    if (st->done >= st->max_iter) {                    // while (cnt < m_max_iter), lpsol.h:1039
        __syncthreads();
        if (threadIdx.x == 0) { st->status = 4; st->row = -1; }
        return;
    }
    const int rhs = v.rhs, lim = rhs - 1;
    const int cached_col = st->cached_col;
    const bool b_cached = st->bcol_valid != 0;
    int first = st->next_first, anypos = st->anypos;
    __syncthreads();
    if (first == NF_UNKNOWN) price_scan(v, sh_i, &sh_flag, first, anypos);
    const int stop = first == INT_MAX ? rhs : first;
    for (int j = threadIdx.x; j < stop; j += blockDim.x)
        if (!v.nv[j]) v.obj[j] = zero<S>();           // lpsol.h:1055-1060
    __syncthreads();
    int enter = -1, leave = -1;
    if (first == INT_MAX) {
        if (!anypos) {                                 // optimum reached: lpsol.h:1089
            if (threadIdx.x == 0) { st->status = ST_CHECK_OPT; st->row = -1; }
            return;
        }
        // SIX::findPivotNVandBVPair (lpsol.h:671-773)
        for (int pass = 0; pass < 2 && enter < 0; pass++) {
            for (int i = 0; i < rhs; i++) {
                if (v.bv[i] || v.rowcnt[i] >= lim) continue;
                S c = v.obj[i];
                bool take = gt(c, zero<S>()) ? true : (eq(c, zero<S>()) ? pass == 1 : false);
                if (!take) continue;
                int b = ratio_test(v, i, sh_c, i == cached_col, b_cached);
                if (b < 0) continue;
                enter = i; leave = b;
                break;
            }
        }
        if (enter < 0) {
            if (threadIdx.x == 0) { st->status = 1; st->row = -1; }   // SIX_UNBOUND
            return;
        }
    } else {
        leave = ratio_test(v, first, sh_c, first == cached_col, b_cached);
        if (leave < 0) {                               // disableNV + continue, lpsol.h:1146-1151
            int add = 0;
            for (int j = threadIdx.x; j < rhs; j += blockDim.x) {
                if (j == first || ppt_seen(v, first, j)) continue;
                atomicOr(&v.ppt[(size_t)first * v.pw + (j >> 5)], 1u << (j & 31));
                v.colcnt[j] += 1;
                add++;
            }
            if (add) atomicAdd(&v.rowcnt[first], add);
            __threadfence_block();
            __syncthreads();
            int nf, any;                               // no sweep follows: re-price here
            price_scan(v, sh_i, &sh_flag, nf, any);
            if (threadIdx.x == 0) { st->row = -1; st->next_first = nf; st->anypos = any; }
            return;
        }
        enter = first;
    }
    // ---- pivot (enter, leave) chosen
    const bool col_cached = enter == cached_col;
    const int r = v.bv2eq[leave];
    __syncthreads();                                   // everyone has read bv2eq[leave]
    for (int i = threadIdx.x; i < v.m; i += blockDim.x)
        v.colbuf[i] = neg(col_cached ? v.nextcol[i] : v.tab[(size_t)i * v.ld + enter]);   // :1485
    if (threadIdx.x == 0) {
        if (!ppt_seen(v, enter, leave)) {              // genPair, lpsol.h:100-104
            v.ppt[(size_t)enter * v.pw + (leave >> 5)] |= 1u << (leave & 31);
            v.rowcnt[enter] += 1; v.colcnt[leave] += 1;
        }
        st->row = r; st->col = enter; st->leave = leave;
        st->cnv_bits = to_bits(v.obj[enter]);
        st->piv_bits = to_bits(col_cached ? v.nextcol[r] : v.tab[(size_t)r * v.ld + enter]);
        v.nv[enter] = 0; v.nv[leave] = 1; v.bv[enter] = 1; v.bv[leave] = 0;   // :1504-1510
        v.eq2bv[r] = enter; v.bv2eq[enter] = r; v.bv2eq[leave] = -1;
        const unsigned t = st->total_pivots;
        if ((int)t < v.trace_cap) { v.trace[2 * t] = enter; v.trace[2 * t + 1] = leave; }
        st->total_pivots = t + 1;
        st->done += 1;
        st->next_first = INT_MAX; st->anypos = 0;      // k_prep's look-ahead fills these
    }
}

// Row / column staging, objective update and basis swap for the pivot chosen in
// LoopState (lpsol.h:1468-1474, :1485, :1496-1510). Reads the tableau only.
// guarded: only while the loop is running; counted: the pivot counts towards 'cnt'
// (the forced pivots of phase 1 are neither, lpsol.h:906-908, :939).
template <class S> __global__ __launch_bounds__(256)
void k_prep(LpView<S> v, int guarded, int counted, int bookkeeping, int lookahead)
{
    LoopState * st = v.st;
    if ((guarded && st->status != ST_RUNNING) || st->row < 0) return;
    const int r = st->row, c = st->col;
    const S s = div(one<S>(), from_bits<S>(st->piv_bits));   // 1/(eq.get(eqnum, nv)), :1471
    const int smode = scale_mode(s);
    const S cnv = from_bits<S>(st->cnv_bits);
    const int cmode = scale_mode(cnv);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const int lim = v.rhs - 1;
    int nf = INT_MAX, any = 0;
    for (int j = gid; j < v.W; j += gsz) {
        S e = scaled(v.tab[(size_t)r * v.ld + j], s, smode);
        v.rowbuf[j] = e;
        S t = mul(e, minus_one<S>());                          // nvexp.mul(-1), :1496
        if (j >= v.rhs) t = neg(t);                            // :1497-1499
        t = scaled(t, cnv, cmode);                             // nvexp.mul(tgtf(nv)), :1500
        const S o = add(t, v.obj[j]);                          // addRowToRow, :1501
        v.obj[j] = o;
        // look-ahead pricing of the next iteration (basis already swapped by k_pick)
        if (lookahead && j < v.rhs && v.nv[j] && gt(o, zero<S>())) {
            any = 1;
            if (v.rowcnt[j] < lim) nf = min(nf, j);
        }
    }
    if (lookahead) {
        for (int o = 32; o > 0; o >>= 1) { nf = min(nf, __shfl_xor(nf, o)); any |= __shfl_xor(any, o); }
        if ((threadIdx.x & 63) == 0) {
            if (nf != INT_MAX) atomicMin(&st->next_first, nf);
            if (any) atomicOr(&st->anypos, 1);
        }
        return;                                                // colbuf / basis were done by k_pick
    }
    for (int i = gid; i < v.m; i += gsz)
        v.colbuf[i] = neg(v.tab[(size_t)i * v.ld + c]);        // coeff_of_nv, :1485
    if (bookkeeping && gid == 0) {
        const int leave = st->leave;                           // :1504-1510
        v.nv[c] = 0; v.nv[leave] = 1; v.bv[c] = 1; v.bv[leave] = 0;
        v.eq2bv[r] = c; v.bv2eq[c] = r; v.bv2eq[leave] = -1;
        const unsigned t = st->total_pivots;
        if ((int)t < v.trace_cap) { v.trace[2 * t] = c; v.trace[2 * t + 1] = leave; }
        st->total_pivots = t + 1;
        if (counted) st->done += 1;
    }
    if (gid == 0) { st->next_first = NF_UNKNOWN; st->cached_col = -1; st->bcol_valid = 0; }
}

// K1, generic scalar: one element per thread per row, rows looped per block.
// In loop mode (guarded) the sweep also writes the post-update values of the
// look-ahead column and of the constant column to contiguous arrays.
template <class S, int ROWS> __global__ __launch_bounds__(256)
void k_update(LpView<S> v, int guarded)
{
    const LoopState * st = v.st;
    if ((guarded && st->status != ST_RUNNING) || st->row < 0) return;
    const int r = st->row;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= v.W) return;
    const int xcol = guarded ? st->next_first : -1;
    const bool ex_col = guarded && j == xcol, ex_b = guarded && j == v.rhs;
    if (guarded && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        v.st->cached_col = (xcol >= 0 && xcol < v.W) ? xcol : -1; v.st->bcol_valid = 1;
    }
    const S e = v.rowbuf[j];
    const int i0 = blockIdx.y * ROWS;
    for (int ii = 0; ii < ROWS; ii++) {
        const int i = i0 + ii;
        if (i >= v.m) break;
        S * p = v.tab + (size_t)i * v.ld + j;
        const S o = (i == r) ? e : add(*p, mul(v.colbuf[i], e));
        *p = o;
        if (ex_col) v.nextcol[i] = o;
        if (ex_b) v.bcol[i] = o;
    }
}

// K1, fp64: the judged HBM-bound sweep. 256 threads cover a 512-column strip
// with one 16-byte access each; e_j lives in two registers for the whole
// row loop, -a_i,nv arrives through the scalar cache (wave-uniform index), and
// UNROLL rows of loads are in flight before the first use. mul then add are
// two roundings (file compiled with -ffp-contract=off).
template <int ROWS, int UNROLL> __global__ __launch_bounds__(256)
void k_update_f64(double * __restrict__ tab, int m, int W, int ld,
                  const double * __restrict__ rowbuf, const double * __restrict__ colbuf,
                  LoopState * __restrict__ st, int guarded,
                  double * __restrict__ nextcol, double * __restrict__ bcol, int rhs)
{
    if ((guarded && st->status != ST_RUNNING) || st->row < 0) return;
    const int r = st->row;
    const int j = blockIdx.x * 512 + threadIdx.x * 2;
    if (j >= W) return;
    // contiguous export of the look-ahead column and the constant column (loop mode)
    int xc = guarded ? st->next_first : -1;
    if (xc >= W) xc = -1;                                     // INT_MAX: nothing to price next
    const int xb = guarded ? rhs : -1;
    if (guarded && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        st->cached_col = xc >= 0 ? xc : -1; st->bcol_valid = 1;
    }
    const bool ex_col = xc >= 0 && (xc >> 1) == (j >> 1), ex_b = xb >= 0 && (xb >> 1) == (j >> 1);
    const int i0 = blockIdx.y * ROWS;
    const int iend = min(i0 + ROWS, m);
    if (j + 1 < W) {
        const double2 e = *reinterpret_cast<const double2 *>(rowbuf + j);
        double * base = tab + (size_t)i0 * ld + j;
        int i = i0;
        for (; i + UNROLL <= iend; i += UNROLL) {
            double2 a[UNROLL];
            double k[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                a[u] = *reinterpret_cast<const double2 *>(base + (size_t)u * ld);
                k[u] = colbuf[i + u];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                double2 o;
                const double p0 = k[u] * e.x, p1 = k[u] * e.y;
                o.x = a[u].x + p0; o.y = a[u].y + p1;
                if (i + u == r) o = e;
                *reinterpret_cast<double2 *>(base + (size_t)u * ld) = o;
                if (ex_col) nextcol[i + u] = (xc & 1) ? o.y : o.x;
                if (ex_b) bcol[i + u] = (xb & 1) ? o.y : o.x;
            }
            base += (size_t)UNROLL * ld;
        }
        for (; i < iend; i++) {
            double2 a = *reinterpret_cast<const double2 *>(base);
            const double k = colbuf[i];
            double2 o;
            const double p0 = k * e.x, p1 = k * e.y;
            o.x = a.x + p0; o.y = a.y + p1;
            if (i == r) o = e;
            *reinterpret_cast<double2 *>(base) = o;
            if (ex_col) nextcol[i] = (xc & 1) ? o.y : o.x;
            if (ex_b) bcol[i] = (xb & 1) ? o.y : o.x;
            base += ld;
        }
    } else {                                                  // odd last column
        const double e = rowbuf[j];
        for (int i = i0; i < iend; i++) {
            double * p = tab + (size_t)i * ld + j;
            const double q = colbuf[i] * e;
            const double o = (i == r) ? e : (*p + q);
            *p = o;
            if (ex_col) nextcol[i] = o;
            if (ex_b) bcol[i] = o;
        }
    }
}

// ---- optimum: solution read-out + SIX::is_feasible (lpsol.h:1104-1110, :784-822)
template <class S> __global__ void k_solution(LpView<S> v)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int j = gid; j < v.W; j += gsz) {
        S x = zero<S>();
        if (j < v.rhs && v.bv[j]) x = v.tab[(size_t)v.bv2eq[j] * v.ld + v.rhs];
        v.x[j] = x;
        if (j < v.rhs && gt(mul(v.vcd[j], x), v.vcr[j])) v.st->infeasible = 1;
    }
}
// One thread per row; the sum runs over j ascending exactly as the reference
// does, skipping nonbasic j whose x_j is an exact zero (adding a*0 leaves the
// running sum unchanged for finite a).
template <class S> __global__ void k_rowcheck(LpView<S> v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.m) return;
    S sum = zero<S>();
    const S * row = v.tab + (size_t)i * v.ld;
    for (int j = 0; j < v.rhs; j++)
        if (v.bv[j]) sum = add(sum, mul(row[j], v.x[j]));
    reduce(sum);
    S b = row[v.rhs];
    reduce(b);
    v.tab[(size_t)i * v.ld + v.rhs] = b;                       // lc.reduce(i, rhs_idx) writes back
    if (ne(sum, b)) v.st->infeasible = 1;
}
template <class S> __global__ void k_finish(LpView<S> v, S * maxv)
{
    LoopState * st = v.st;
    if (st->status != ST_CHECK_OPT) return;
    if (st->infeasible) { st->status = 3; return; }            // SIX_OPTIMAL_IS_INFEASIBLE
    *maxv = v.obj[v.rhs];                                      // lpsol.h:1119
    st->status = 0;
}

// ---- slack-form construction (SIX::slack, lpsol.h:1406-1433; the xa column
// of constructBasicFeasibleSolution, lpsol.h:860-868) straight into HBM.
template <class S> __global__ void k_build(LpView<S> v, const S * leq, const S * tgtf,
                                           int n, int with_xa)
{
    const int W = v.W, cols = n + 1;
    const int first_slack = n + (with_xa ? 1 : 0);
    const size_t total = (size_t)(v.m + 1) * W;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / W), j = (int)(t % W);
        S val = zero<S>();
        if (i < v.m) {
            if (j < n) val = leq[(size_t)i * cols + j];
            else if (with_xa && j == n) val = minus_one<S>();
            else if (j == v.rhs) val = leq[(size_t)i * cols + n];
            else if (j - first_slack == i) val = one<S>();
            v.tab[(size_t)i * v.ld + j] = val;
        } else {
            if (with_xa) { if (j == n) val = minus_one<S>(); }
            else if (j < n) val = tgtf[j];
            else if (j == v.rhs) val = tgtf[n];
            v.obj[j] = val;
        }
    }
}
template <class S> __global__ void k_init_basis(LpView<S> v, int first_slack)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int i = gid; i < v.rhs; i += gsz) {
        const bool slack = i >= first_slack;
        v.nv[i] = slack ? 0 : 1; v.bv[i] = slack ? 1 : 0;
        v.bv2eq[i] = slack ? i - first_slack : -1;
        if (slack) v.eq2bv[i - first_slack] = i;
    }
}
template <class S> __global__ void k_reset_loop(LpView<S> v, unsigned max_iter)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int i = gid; i < v.rhs; i += gsz) { v.rowcnt[i] = 0; v.colcnt[i] = 0; }
    const size_t words = (size_t)v.rhs * v.pw;
    for (size_t t = gid; t < words; t += gsz) v.ppt[t] = 0u;
    if (gid == 0) {
        LoopState * st = v.st;
        st->status = ST_RUNNING; st->done = 0; st->max_iter = max_iter;
        st->row = -1; st->infeasible = 0;
        st->next_first = NF_UNKNOWN; st->anypos = 0; st->cached_col = -1; st->bcol_valid = 0;
    }
}

// stage1's trigger (lpsol.h:1794-1803): aux = 1 if phase 1 is needed.
template <class S> __global__ void k_need_phase1(const S * leq, const S * tgtf, int m, int n,
                                                 LoopState * st)
{
    __shared__ int anypos, anyneg;
    if (threadIdx.x == 0) { anypos = 0; anyneg = 0; }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x)
        if (gt(tgtf[j], zero<S>())) anypos = 1;
    for (int i = threadIdx.x; i < m; i += blockDim.x)
        if (lt(leq[(size_t)i * (n + 1) + n], zero<S>())) anyneg = 1;
    __syncthreads();
    if (threadIdx.x == 0) st->aux = (!anypos || anyneg) ? 1 : 0;
}

// Forced first pivot of phase 1: row of the smallest constant, lowest index on
// ties, entering xa (lpsol.h:894-908).
template <class S> __global__ __launch_bounds__(1024) void k_force_pivot(LpView<S> v, int xa)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_raw[16 * sizeof(Cand<S>)];
    Cand<S> * sh = (Cand<S> *)sh_raw;
    Cand<S> best; best.q = zero<S>(); best.idx = INT_MAX;
    for (int i = threadIdx.x; i < v.m; i += blockDim.x) {
        Cand<S> c; c.q = v.tab[(size_t)i * v.ld + v.rhs]; c.idx = i;
        best = better(best, c);
    }
    best = block_argmin(best, sh);
    if (threadIdx.x == 0) {
        LoopState * st = v.st;
        const int r = best.idx;
        st->row = r; st->col = xa; st->leave = v.eq2bv[r];
        st->cnv_bits = to_bits(v.obj[xa]);
        st->piv_bits = to_bits(v.tab[(size_t)r * v.ld + xa]);
    }
}

// After phase 1 solved: maxv.reduce() != 0 -> aux = -1 (no feasible solution);
// xa still basic -> choose the first nonbasic column with a nonzero (reduced)
// coefficient in xa's row and stage that pivot: aux = 1; else aux = 0.
// (lpsol.h:919-941)
template <class S> __global__ void k_phase1_exit(LpView<S> v, int xa, S * maxv)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    LoopState * st = v.st;
    S best = *maxv;
    reduce(best);
    st->row = -1;
    if (ne(best, zero<S>())) { st->aux = -1; return; }
    if (!v.bv[xa]) { st->aux = 0; return; }
    const int r = v.bv2eq[xa];
    int cand = 0;
    for (; cand < v.rhs; cand++) {
        if (!v.nv[cand]) continue;
        S a = v.tab[(size_t)r * v.ld + cand];
        reduce(a);
        v.tab[(size_t)r * v.ld + cand] = a;
        if (ne(a, zero<S>())) break;
    }
    if (cand >= v.rhs) { st->aux = -7; return; }               // reference: undefined
    st->row = r; st->col = cand; st->leave = xa;
    st->cnv_bits = to_bits(v.obj[cand]);
    st->piv_bits = to_bits(v.tab[(size_t)r * v.ld + cand]);
    st->aux = 1;
}

// Objective rebuild at the end of phase 1 (lpsol.h:944-953 with
// {R,Float}Mat::substit, xmat.cpp:571-599 / :1491-1519). One workgroup;
// basic variables are substituted in ascending index order.
template <class S> __global__ __launch_bounds__(1024)
void k_rebuild_obj(LpView<S> v, const S * tgtf0, int n0)
{
    __shared__ unsigned long long sh_f, sh_e;
    const int W = v.W, rhs = v.rhs;
    for (int j = threadIdx.x; j < W; j += blockDim.x)
        v.obj[j] = j < n0 ? tgtf0[j] : (j == rhs ? tgtf0[n0] : zero<S>());
    __syncthreads();
    for (int i = 0; i < rhs; i++) {
        if (threadIdx.x == 0) { S f = v.obj[i]; reduce(f); v.obj[i] = f; sh_f = to_bits(f); }
        __syncthreads();
        const S f = from_bits<S>(sh_f);
        if (ne(f, zero<S>()) && v.bv[i]) {                     // uniform branch
            const S * expr = v.tab + (size_t)v.bv2eq[i] * v.ld;
            if (threadIdx.x == 0) {
                v.obj[rhs] = mul(v.obj[rhs], minus_one<S>());  // mulOfColumns(rhs.., -1)
                sh_e = to_bits(expr[i]);
            }
            __syncthreads();
            const S ev = from_bits<S>(sh_e);
            if (!eq(ev, zero<S>())) {
                S k; int mode;
                if (ne(f, ev)) {
                    k = div(neg(f), ev);
                    mode = eq(k, zero<S>()) ? SCALE_ZERO : (eq(k, one<S>()) ? SCALE_KEEP : SCALE_MUL);
                } else { k = minus_one<S>(); mode = SCALE_MUL; }
                for (int j = threadIdx.x; j < W; j += blockDim.x)
                    v.obj[j] = add(scaled(expr[j], k, mode), v.obj[j]);
            }
            __syncthreads();
            if (threadIdx.x == 0) v.obj[rhs] = mul(v.obj[rhs], minus_one<S>());
        }
        __syncthreads();
    }
}

// Physical removal of column xa from the tableau (row i = blockIdx.x; the extra
// block m handles the objective row and the per-variable arrays), lpsol.h:955-986.
template <class S> __global__ __launch_bounds__(256) void k_delete_col(LpView<S> v, int xa)
{
    const int W = v.W;
    S * row = blockIdx.x < v.m ? v.tab + (size_t)blockIdx.x * v.ld : v.obj;
    for (int c0 = xa; c0 < W - 1; c0 += blockDim.x) {
        const int j = c0 + threadIdx.x;
        S t = zero<S>();
        if (j < W - 1) t = row[j + 1];
        __syncthreads();
        if (j < W - 1) row[j] = t;
        __syncthreads();
    }
    if (blockIdx.x == v.m) {
        for (int c0 = xa; c0 < v.rhs - 1; c0 += blockDim.x) {
            const int j = c0 + threadIdx.x;
            uint8_t a = 0, b = 0; int q = 0; S d = zero<S>(), e = zero<S>();
            if (j < v.rhs - 1) { a = v.nv[j + 1]; b = v.bv[j + 1]; q = v.bv2eq[j + 1];
                                 d = v.vcd[j + 1]; e = v.vcr[j + 1]; }
            __syncthreads();
            if (j < v.rhs - 1) { v.nv[j] = a; v.bv[j] = b; v.bv2eq[j] = q; v.vcd[j] = d; v.vcr[j] = e; }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < v.m; i += blockDim.x)
            if (v.eq2bv[i] > xa) v.eq2bv[i] -= 1;
    }
}

} // namespace xpg
