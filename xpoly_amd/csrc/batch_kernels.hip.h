// Batches of independent small LPs -- the polyhedral dependence-test workload
// (Lineq::has_solution -> SIX::maxm / minm, src/com/linsys.cpp:852-904; one call
// per DepPoly::is_empty, src/eng/poly.cpp:530-573). One workgroup owns one LP
// from input to answer: the whole slack tableau, objective row, basis maps and
// pivot-pair table live in LDS (32x64 LP: ~29 KB of the CU's 160 KB), so HBM
// sees only the 16 KB problem read and the ~0.5 KB result write. Every step of
// SIX::maxm / minm for an x >= 0, inequality-only problem runs inside the
// kernel: slack construction, the auxiliary-variable phase 1, solveSlackForm,
// the feasibility check and the final objective.
#pragma once
#include "lp_kernels.hip.h"
#include "rat_ops.hip.h"

namespace xpg {

#ifdef XPG_STAMPS
static __device__ unsigned long long g_fl[16];                  // diagnostic builds: sm_fast_loop counters / 100 MHz ticks (tools/lab/probe_fastloop.py)
#define FL_ADD(k_, v_) atomicAdd(&g_fl[k_], (unsigned long long)(v_))
#else
#define FL_ADD(k_, v_) do { } while (0)
#endif
#ifdef XPG_LIFE
// diagnostic builds (-DXPG_LIFE, tools/lab/probe_batch_life.py): 100 MHz time of every 256th pivot of the first 4096 LPs
static __device__ unsigned long long g_life[4096 * 32];
#define LIFE_MARK(lp_, piv_) do { if (threadIdx.x == 0 && (lp_) < 4096 && ((piv_) & 255) == 0 && ((piv_) >> 8) < 31) g_life[(lp_) * 32 + ((piv_) >> 8)] = wall_clock64(); } while (0)
#define LIFE_END(lp_) do { if (threadIdx.x == 0 && (lp_) < 4096) g_life[(lp_) * 32 + 31] = wall_clock64(); } while (0)
#else
#define LIFE_MARK(lp_, piv_) do { } while (0)
#define LIFE_END(lp_) do { } while (0)
#endif

template <class S> struct Small {
    S * tab; int R, W, ld, rhs;     // R rows, W live columns, row stride ld
    S * obj; S * e; S * k; S * x;
    uint8_t * nv; uint8_t * bv; int * bv2eq; int * eq2bv;
    uint32_t * ppt; int pw; int * rowcnt; int * colcnt;
    Cand<S> * sh_c; int * sh_i;     // reduction scratch (16 entries each)
    int * sh_w;                     // 8 words of broadcast scratch
    unsigned pivots;
    unsigned closes;                // iterations that ended in disableNV (no pivot), for profiling
    bool cn;                        // Rational: every input cell canonical -> the 32-bit forms (rat_ops.hip.h)
};

template <class S> __device__ __forceinline__ bool sm_seen(const Small<S> & P, int nv, int b)
{ return (P.ppt[nv * P.pw + (b >> 5)] >> (b & 31)) & 1u; }

// SIX::pivot (lpsol.h:1456-1511) on the LDS tableau; all threads participate.
template <class S> __device__ __forceinline__ void sm_pivot(Small<S> & P, int nv, int bv)
{
    const int r = P.bv2eq[bv], W = P.W, ld = P.ld;
    const S piv = P.tab[r * ld + nv];
    const S cnv = P.obj[nv];
    __syncthreads();
    const S s = q_div(P.cn, one<S>(), piv);
    const int smode = scale_mode(s), cmode = scale_mode(cnv);
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        const S ej = q_scaled(P.cn, P.tab[r * ld + j], s, smode);
        P.e[j] = ej;
        P.tab[r * ld + j] = ej;
        S t = q_mul(P.cn, ej, minus_one<S>());
        if (j >= P.rhs) t = neg(t);
        t = q_scaled(P.cn, t, cnv, cmode);
        P.obj[j] = q_add(P.cn, t, P.obj[j]);
    }
    for (int i = threadIdx.x; i < P.R; i += blockDim.x)
        if (i != r) P.k[i] = neg(P.tab[i * ld + nv]);
    __syncthreads();
    // sweep: a lane owns a column (e_j stays in a register), lane groups of CW lanes take
    // alternate rows; k_i is an LDS broadcast. Consecutive lanes touch consecutive cells.
    {
        const int CW = blockDim.x >= 128 && W > 64 ? 128 : 64;
        const int tx = threadIdx.x % CW, ty = threadIdx.x / CW, ny = blockDim.x / CW;
        for (int j = tx; j < W; j += CW) {
            const S ej = P.e[j];
            S * p = P.tab + ty * ld + j;
            for (int i = ty; i < P.R; i += ny, p += ny * ld)
                if (i != r) *p = q_fma(P.cn, *p, P.k[i], ej);
        }
    }
    if (threadIdx.x == 0) {
        P.nv[nv] = 0; P.nv[bv] = 1; P.bv[nv] = 1; P.bv[bv] = 0;
        P.eq2bv[r] = nv; P.bv2eq[nv] = r; P.bv2eq[bv] = -1;
        XPG_TRACE_PIVOT("lds", nv, bv, r);
    }
    P.pivots++;
    __syncthreads();
}

// SIX::findPivotBV (lpsol.h:553-663)
template <class S> __device__ __forceinline__ int sm_ratio(const Small<S> & P, int nv)
{
    const int lim = P.rhs - 1;
    for (int pass = 0; pass < 2; pass++) {
        Cand<S> best; best.q = zero<S>(); best.idx = INT_MAX;
        bool weird = false;
        for (int i = threadIdx.x; i < P.R; i += blockDim.x) {
            const S a = P.tab[i * P.ld + nv];
            if (pass == 0 ? le(a, zero<S>()) : eq(a, zero<S>())) continue;
            const int b = P.eq2bv[i];
            if (sm_seen(P, nv, b) || P.colcnt[b] >= lim) continue;
            Cand<S> c; c.q = q_div(P.cn, P.tab[i * P.ld + P.rhs], a); c.idx = i;
            weird |= unordered_value(c.q);
            best = better(best, c);
        }
        best = block_argmin(best, P.sh_c);
        if (__builtin_expect(__syncthreads_or(weird ? 1 : 0), 0)) {   // an unordered quotient: the reference's scan itself (lp_kernels.hip.h)
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                int sbest = INT_MAX; S sq = zero<S>();
                for (int base = 0; base < P.R; base += 64) {
                    const int i = min(base + lane, P.R - 1);
                    const S a = P.tab[i * P.ld + nv];
                    const int b = P.eq2bv[i];
                    bool ok = base + lane < P.R && !(pass == 0 ? le(a, zero<S>()) : eq(a, zero<S>()));
                    ok = ok && !(sm_seen(P, nv, b) || P.colcnt[b] >= lim);
                    const S q = ok ? q_div(P.cn, P.tab[i * P.ld + P.rhs], a) : zero<S>();
                    scan_step_in_order(q, ok, base, sbest, sq);
                }
                if (lane == 0) P.sh_i[0] = sbest;
            }
            __syncthreads();
            best.idx = P.sh_i[0];
            __syncthreads();
        }
        if (best.idx != INT_MAX) return P.eq2bv[best.idx];
    }
    return -1;
}

// ---- fast path for the dependence-test sizes (rhs <= 128 variables, R <= 64 rows) ------------
// Wave 0 alone does the selection with wave-level primitives (ballot / shuffles, no
// workgroup barrier); the other waves join for the staging and the sweep. Three barriers
// per pivot instead of ten.
enum { ACT_PIVOT = 0, ACT_OPT = 1, ACT_FINDPAIR = 2, ACT_CLOSE = 3, ACT_TIMEOUT = 4, ACT_UNBOUND = 5 };

// Pricing (lpsol.h:1054-1069) + ratio test (lpsol.h:553-663) + genPair by wave 0.
// Results in sh_w[0..2] = action, entering column, leaving variable; the pivot element and
// c_nv are parked in sh_c so that nobody re-reads them after the row has been rescaled.
template <class S> __device__ __forceinline__ void sm_select_wave0(Small<S> & P)
{
    // The selection is a latency chain: every LDS load of a stage is issued before the first use
    // (indices clamped instead of tested: the reference's short-circuit order would serialise them)
    // and the bookkeeping takes its values from the lanes' registers through v_readlane.
    const int lane = threadIdx.x, rhs = P.rhs, lim = rhs - 1, R = P.R;
    const int j0 = lane, j1 = lane + 64;
    const bool in0 = j0 < rhs, in1 = j1 < rhs;
    const int q0 = in0 ? j0 : 0, q1 = in1 ? j1 : 0;
    const int nv0 = P.nv[q0], nv1 = P.nv[q1];
    const S ob0 = P.obj[q0], ob1 = P.obj[q1];
    const int rc0 = P.rowcnt[q0], rc1 = P.rowcnt[q1];
    const bool nb0 = in0 && nv0 != 0, nb1 = in1 && nv1 != 0;
    const bool c0 = nb0 && gt(ob0, zero<S>()), c1 = nb1 && gt(ob1, zero<S>());
    const bool o0 = c0 && rc0 < lim, o1 = c1 && rc1 < lim;
    const unsigned long long m0 = __ballot(o0), m1 = __ballot(o1), any = __ballot(c0 || c1);
    const int first = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : INT_MAX);
    const int stop = first == INT_MAX ? rhs : first;
    if (in0 && j0 < stop && !nb0) P.obj[j0] = zero<S>();             // lpsol.h:1055-1060
    if (in1 && j1 < stop && !nb1) P.obj[j1] = zero<S>();
    if (first == INT_MAX) {
        if (lane == 0) { P.sh_w[0] = any ? ACT_FINDPAIR : ACT_OPT; P.sh_w[1] = first; P.sh_w[2] = -1; P.sh_w[3] = -1; }
        return;
    }
    // ratio test (lpsol.h:553-663): two LDS rounds, one division, arg-min on the VALU
    const int li = lane < R ? lane : 0;
    const S a = P.tab[li * P.ld + first], bc = P.tab[li * P.ld + rhs];
    const int b = P.eq2bv[li];
    const uint32_t w = P.ppt[first * P.pw + (b >> 5)];
    const int cc = P.colcnt[b];
    const bool open = lane < R && !((w >> (b & 31)) & 1u) && cc < lim;
    const bool nonzero = open && !eq(a, zero<S>());
    const S qr = nonzero ? q_div(P.cn, bc, a) : zero<S>();
    int row = wave_argmin_row(qr, nonzero && !le(a, zero<S>()), lane);
    if (row == INT_MAX) row = wave_argmin_row(qr, nonzero, lane);    // relaxed second pass: a != 0
    if (row == INT_MAX) {
        if (lane == 0) { P.sh_w[0] = ACT_CLOSE; P.sh_w[1] = first; P.sh_w[2] = -1; P.sh_w[3] = -1; }
        return;
    }
    const int leave = __builtin_amdgcn_readlane(b, row);
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)w, row);
    const int ccv = __builtin_amdgcn_readlane(cc, row);
    Cand<S> pa; pa.q = a; pa.idx = 0;
    Cand<S> po0; po0.q = ob0; po0.idx = rc0;
    Cand<S> po1; po1.q = ob1; po1.idx = rc1;
    const S piv = read_lane(pa, row).q;
    const Cand<S> pf = first < 64 ? read_lane(po0, first) : read_lane(po1, first - 64);
    if (lane == 0) {
        P.sh_w[0] = ACT_PIVOT; P.sh_w[1] = first; P.sh_w[2] = leave; P.sh_w[3] = row;
        // genPair (lpsol.h:100-104): a candidate row was by construction not yet paired
        P.ppt[first * P.pw + (leave >> 5)] = wv | (1u << (leave & 31));
        P.rowcnt[first] = pf.idx + 1; P.colcnt[leave] = ccv + 1;
        S * park = (S *)P.sh_c;
        park[0] = piv;
        park[1] = pf.q;
        park[2] = q_div(P.cn, one<S>(), piv);              // 1/(eq.get(eqnum, nv)), lpsol.h:1471: once, here (see sm_fast_loop_body)
    }
}

// SIX::pivot (lpsol.h:1456-1511) with the pivot element and c_nv handed in.
template <class S> __device__ __forceinline__ void sm_pivot_fast(Small<S> & P, int nv, int bv, int r, S piv, S cnv)
{
    const int W = P.W, ld = P.ld;
    const S s = q_div(P.cn, one<S>(), piv);
    const int smode = scale_mode(s), cmode = scale_mode(cnv);
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        const S ej = q_scaled(P.cn, P.tab[r * ld + j], s, smode);
        P.e[j] = ej;
        P.tab[r * ld + j] = ej;
        S t = q_mul(P.cn, ej, minus_one<S>());
        if (j >= P.rhs) t = neg(t);
        t = q_scaled(P.cn, t, cnv, cmode);
        P.obj[j] = q_add(P.cn, t, P.obj[j]);
    }
    for (int i = threadIdx.x; i < P.R; i += blockDim.x)
        P.k[i] = i != r ? neg(P.tab[i * ld + nv]) : zero<S>();
    __syncthreads();
    {
        const int CW = blockDim.x >= 128 && W > 64 ? 128 : 64;
        const int tx = threadIdx.x % CW, ty = threadIdx.x / CW, ny = blockDim.x / CW;
        // Four rows of LDS reads are issued before the first use: the loop is latency-bound (one dependent
        // ds_read -> mul -> add -> ds_write chain per cell otherwise). No clamp, select or range test per cell:
        // four rows while all four are in range, then the tail; the pivot row goes through a + 0*e (k_r = 0)
        // and is put back to e afterwards.
        const int R = P.R;
        for (int j = tx; j < W; j += CW) {
            const S ej = P.e[j];
            S * col = P.tab + j;
            int i = ty;
            for (; i + 3 * ny < R; i += 4 * ny) {
                const int i1 = i + ny, i2 = i + 2 * ny, i3 = i + 3 * ny;
                const S a0 = col[i * ld], a1 = col[i1 * ld], a2 = col[i2 * ld], a3 = col[i3 * ld];
                const S k0 = P.k[i], k1 = P.k[i1], k2 = P.k[i2], k3 = P.k[i3];
                col[i * ld] = q_fma(P.cn, a0, k0, ej);
                col[i1 * ld] = q_fma(P.cn, a1, k1, ej);
                col[i2 * ld] = q_fma(P.cn, a2, k2, ej);
                col[i3 * ld] = q_fma(P.cn, a3, k3, ej);
            }
            for (; i < R; i += ny) col[i * ld] = q_fma(P.cn, col[i * ld], P.k[i], ej);
            if (r >= ty && (r - ty) % ny == 0) col[r * ld] = ej;
        }
    }
    if (threadIdx.x == 0) {
        P.nv[nv] = 0; P.nv[bv] = 1; P.bv[nv] = 1; P.bv[bv] = 0;
        P.eq2bv[r] = nv; P.bv2eq[nv] = r; P.bv2eq[bv] = -1;
        XPG_TRACE_PIVOT("lds", nv, bv, r);
    }
    P.pivots++;
    __syncthreads();
}

// SIX::findPivotBV (lpsol.h:553-663) for column nv by one wavefront (R <= 64, lane = row): both passes, DPP
// arg-min, no workgroup barrier. Returns the pivot row or INT_MAX; b / w / cc are this lane's basic variable, pair
// word and counter (the caller fetches the winner's by v_readlane).
// (b, cc, bc -- this lane's basic variable, its pair counter and its constant-column entry -- do not depend on the column:
// a caller that tries many columns in a row fetches them once)
template <class S> __device__ __forceinline__ int sm_ratio_wave_with(const Small<S> & P, int nv, int lane, int b, int cc, S bc, uint32_t & w, S & a)
{
    const int lim = P.rhs - 1, R = P.R, li = lane < R ? lane : 0;
    a = P.tab[li * P.ld + nv];
    w = P.ppt[nv * P.pw + (b >> 5)];
    const bool open = lane < R && !((w >> (b & 31)) & 1u) && cc < lim;
    const bool nonzero = open && !eq(a, zero<S>());
    // no open row with a nonzero entry: both passes fail, and findPivotNVandBVPair's scan meets many such columns
    // late in a long LP (an LP of the dependence-test family was measured at 12 us per pivot after 5 000 pivots, ~50
    // failed candidates each) -- they leave before the quotient and the two arg-min chains
    if (__ballot(nonzero) == 0ull) return INT_MAX;
    const S qr = nonzero ? q_div(P.cn, bc, a) : zero<S>();
    int row = wave_argmin_row(qr, nonzero && !le(a, zero<S>()), lane);
    if (row == INT_MAX) row = wave_argmin_row(qr, nonzero, lane);           // relaxed second pass: a != 0
    return row;
}
template <class S> __device__ __forceinline__ int sm_ratio_wave(const Small<S> & P, int nv, int lane, int & b, uint32_t & w, int & cc, S & a)
{
    const int li = lane < P.R ? lane : 0;
    const S bc = P.tab[li * P.ld + P.rhs];
    b = P.eq2bv[li];
    cc = P.colcnt[b];
    return sm_ratio_wave_with(P, nv, lane, b, cc, bc, w, a);
}

// SIX::findPivotNVandBVPair (lpsol.h:671-773) by wave 0 alone: candidates 64 per ballot in ascending order, positive
// reduced costs first, then zero ones; each tried with the wave-level ratio test. On success the pivot is staged
// exactly as sm_select_wave0 stages one (sh_w, parked pivot element and c_nv, pair table); else sh_w[0] = -1.
template <class S> __device__ __forceinline__ void sm_findpair_wave0(Small<S> & P)
{
    const int lane = threadIdx.x, rhs = P.rhs, lim = rhs - 1;
    const int li = lane < P.R ? lane : 0;
    const int b = P.eq2bv[li], cc = P.colcnt[b];               // nothing the candidates share changes before one succeeds
    const S bc = P.tab[li * P.ld + rhs];
    for (int pass = 0; pass < 2; pass++)
        for (int base = 0; base < rhs; base += 64) {
            const int i = base + lane;
            bool take = false;
            if (i < rhs && !P.bv[i] && P.rowcnt[i] < lim) {
                const S c = P.obj[i];
                take = gt(c, zero<S>()) ? true : (eq(c, zero<S>()) ? pass == 1 : false);
            }
            unsigned long long mask = __ballot(take);
            while (mask) {
                const int cand = base + __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                uint32_t w; S a;
                const int row = sm_ratio_wave_with(P, cand, lane, b, cc, bc, w, a);
#ifdef XPG_STAMPS
                if (lane == 0) FL_ADD(2, 1);
#endif
                if (row == INT_MAX) continue;
                const int leave = __builtin_amdgcn_readlane(b, row);
                const uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)w, row);
                const int ccv = __builtin_amdgcn_readlane(cc, row);
                Cand<S> pa; pa.q = a; pa.idx = 0;
                const S piv = read_lane(pa, row).q;
                if (lane == 0) {
                    P.sh_w[0] = ACT_PIVOT; P.sh_w[1] = cand; P.sh_w[2] = leave; P.sh_w[3] = row;
                    if (!((wv >> (leave & 31)) & 1u)) {        // genPair (a candidate row was not yet paired: always taken)
                        P.ppt[cand * P.pw + (leave >> 5)] = wv | (1u << (leave & 31));
                        P.rowcnt[cand] += 1; P.colcnt[leave] = ccv + 1;
                    }
                    S * pk = (S *)P.sh_c;
                    pk[0] = piv;
                    pk[1] = P.obj[cand];
                    pk[2] = q_div(P.cn, one<S>(), piv);
                }
                return;
            }
        }
    if (lane == 0) P.sh_w[0] = -1;
}

// ---- the overlapped fast loop (two or more wavefronts per LP, rhs <= 127 variables, R <= 64 rows) ----------
// One pivot used to be: wave 0 selects (pricing, ratio test: a chain of ~30 dependent LDS round trips) while the
// other waves wait -> barrier -> everybody stages (scaled row, -column, objective row) -> barrier -> everybody
// sweeps -> barrier. PMC at 5 LPs per CU: 1 194 VALU + 770 SALU + 287 LDS wave-instructions per pivot in 21.8 k
// cycles -- 27 % issue utilisation: the chain, not the arithmetic, sets the rate. Here the selection of pivot t+1
// runs UNDER the sweep of pivot t, as the pipelined large-tableau loop does it (lp_kernels.hip.h):
//   stage A   wave 0: objective row of pivot t (two columns per lane; it rescales its pivot-row entries itself) and
//                     the pricing of pivot t+1 on the basis as it will be after the swap (lpsol.h:1054-1069, with
//                     the zeroing of :1055-1060) -> the entering column `first` is known BEFORE the sweep starts;
//             others: scaled pivot row -> e, -column -> k
//   barrier
//   stage C   wave 0: basis swap of pivot t; the two columns the next choice needs -- `first` and the constant
//                     column -- updated for its rows with the sweep's own q_fma(P.cn, a, k, e); ratio test
//                     (lpsol.h:553-663) on those fresh values, pair-table upkeep -> pivot t+1;
//             others: the sweep of every other column (row r := e)
//   barrier
// Two barriers per pivot, and the longest dependent chain of a pivot is max(selection, sweep) instead of their sum.
// Anything but "pivot chosen" leaves the loop with the tableau fully swept and the basis consistent, and the
// generic code of sm_solve takes over exactly as it did behind sm_select_wave0. (Rotating the selecting wave over
// the SIMDs with the workgroup index was tried: no difference.)
// CR / CLD / CT: rows, row stride and workgroup size as compile-time constants (0: run-time values). The 32 x 64 LPs of
// BASELINE configs[2] (R = 32, ld = 97, 256 threads) run the specialised instance: row offsets become immediates, loops
// unroll, and the sweep takes the two-lanes-per-column form below (round 3: 1 040 -> 542 VALU, 521 -> 348 SALU, 263 -> 189 LDS
// wave-instructions per pivot on the dense family, 1 055 / 507 / 264 -> 646 / 424 / 211 on the dependence-test-like one).
template <class S, int CR, int CLD, int CT> __device__ __forceinline__ int sm_fast_loop_body(Small<S> & P, unsigned max_iter, unsigned & done, bool preselected)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const bool w0 = tid < 64;
    const int st = tid - 64, nsw = (CT ? CT : (int)blockDim.x) - 64;       // sweeper index / count
    const int nswaves = nsw >> 6, swave = st >> 6;
    const int rhs = P.rhs, lim = rhs - 1, R = CR ? CR : P.R, W = P.W, ld = CLD ? CLD : P.ld;
    if (CR == 32 && CLD == 97) {
        // the specialised instance: 32 rows and 63 variables, with (W = 97) or without (96) stage 1's column -- lane-constant
        // column tests fold (a wave-0 lane's first column always exists and is never the constant one)
        __builtin_assume(W == 96 || W == 97);
        __builtin_assume(rhs == W - 1);
    }
    if (!preselected) {                                         // the first pivot: nothing to overlap with
        if (w0) sm_select_wave0(P);
        __syncthreads();
    }
    for (;;) {
#ifdef XPG_STAMPS
        unsigned long long fl_t0 = wall_clock64();
#endif
        int action = P.sh_w[0];
        if (action == ACT_FINDPAIR) {
            // every positive column is exhausted: findPivotNVandBVPair right here, on the tableau the barrier above
            // has just completed (the basic objective entries were zeroed by stage A's pricing: stop = rhs)
            __syncthreads();                                    // everybody has read sh_w[0]
            if (w0) sm_findpair_wave0(P);
            __syncthreads();
            action = P.sh_w[0] == ACT_PIVOT ? ACT_PIVOT : ACT_UNBOUND;
#ifdef XPG_STAMPS
            if (tid == 0) { FL_ADD(1, 1); FL_ADD(3, wall_clock64() - fl_t0); }
#endif
        }
#ifdef XPG_STAMPS
        else if (tid == 0 && action == ACT_PIVOT) FL_ADD(0, 1);
        if (action != ACT_PIVOT && tid == 0) FL_ADD(8 + (action % 7), 1);
        unsigned long long fl_t1 = wall_clock64();
#endif
        if (action != ACT_PIVOT) return action;
        const int enter = P.sh_w[1], leave = P.sh_w[2], r = P.sh_w[3];
        const S * park = (const S *)P.sh_c;
        const S cnv = park[1];
        // 1/(eq.get(eqnum, nv)), lpsol.h:1471: computed ONCE by the wave that chose the pivot and parked beside it (round 5:
        // every wave of the workgroup used to divide for itself -- 3 x ~25 VALU wave-instructions of the ~530 per pivot)
        const S s = park[2];
        const int smode = scale_mode(s), cmode = scale_mode(cnv);
        const bool last = done + 1 >= max_iter;                 // while (cnt < m_max_iter), lpsol.h:1039: no pricing after the last pivot
        // ---- stage A
        S o0 = zero<S>(), o1 = zero<S>(); int rc0 = 0, rc1 = 0;
        if (w0) {
            const int j0 = lane, j1 = lane + 64;
            const bool in0 = j0 < W, in1 = j1 < W;
            const int q0 = in0 ? j0 : 0, q1 = in1 ? j1 : 0;
            const S t0 = P.tab[r * ld + q0], t1 = P.tab[r * ld + q1];
            const S ob0 = P.obj[q0], ob1 = P.obj[q1];
            const int nvm0 = P.nv[q0 < rhs ? q0 : 0], nvm1 = P.nv[q1 < rhs ? q1 : 0];
            rc0 = P.rowcnt[q0 < rhs ? q0 : 0]; rc1 = P.rowcnt[q1 < rhs ? q1 : 0];
            S x0 = q_mul(P.cn, q_scaled(P.cn, t0, s, smode), minus_one<S>()), x1 = q_mul(P.cn, q_scaled(P.cn, t1, s, smode), minus_one<S>());   // nvexp.mul(-1), lpsol.h:1496
            if (j0 >= rhs) x0 = neg(x0);                        // :1497-1499
            if (j1 >= rhs) x1 = neg(x1);
            o0 = q_add(P.cn, q_scaled(P.cn, x0, cnv, cmode), ob0);              // :1500-1501
            o1 = q_add(P.cn, q_scaled(P.cn, x1, cnv, cmode), ob1);
            int first = INT_MAX; bool anyc = false;
            if (!last) {
                // basis after this pivot's swap
                const bool nb0 = in0 && j0 < rhs && (j0 == enter ? false : (j0 == leave ? true : nvm0 != 0));
                const bool nb1 = in1 && j1 < rhs && (j1 == enter ? false : (j1 == leave ? true : nvm1 != 0));
                const bool c0 = nb0 && gt(o0, zero<S>()), c1 = nb1 && gt(o1, zero<S>());
                const bool e0 = c0 && rc0 < lim, e1 = c1 && rc1 < lim;
                const unsigned long long m0 = __ballot(e0), m1 = __ballot(e1);
                anyc = __ballot(c0 || c1) != 0ull;
                first = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : INT_MAX);
                const int stop = first == INT_MAX ? rhs : first;
                if (in0 && j0 < stop && j0 < rhs && !nb0) o0 = zero<S>();     // lpsol.h:1055-1060
                if (in1 && j1 < stop && j1 < rhs && !nb1) o1 = zero<S>();
            }
            if (in0) P.obj[j0] = o0;
            if (in1) P.obj[j1] = o1;
            if (lane == 0) { P.sh_w[5] = first; P.sh_w[6] = anyc ? 1 : 0; }
        } else {
            for (int j = st; j < W; j += nsw) P.e[j] = q_scaled(P.cn, P.tab[r * ld + j], s, smode);
            for (int i = st; i < R; i += nsw) P.k[i] = i != r ? neg(P.tab[i * ld + enter]) : zero<S>();
        }
#ifdef XPG_STAMPS
        if (tid == 0) FL_ADD(4, wall_clock64() - fl_t1);
        if (tid == 64) FL_ADD(15, wall_clock64() - fl_t1);
#endif
        __syncthreads();
#ifdef XPG_STAMPS
        unsigned long long fl_t2 = wall_clock64();
#endif
        const int first = P.sh_w[5];
        const bool have_first = first != INT_MAX;
        // ---- stage C
        if (w0) {
            if (lane == 0) {                                    // lpsol.h:1504-1510
                P.nv[enter] = 0; P.nv[leave] = 1; P.bv[enter] = 1; P.bv[leave] = 0;
                P.eq2bv[r] = enter; P.bv2eq[enter] = r; P.bv2eq[leave] = -1;
                XPG_TRACE_PIVOT("lds-fast", enter, leave, r);
            }
            const int li = lane < R ? lane : 0;
            const int fc = have_first ? first : rhs;
            const S kb = P.k[li], ef = P.e[fc], eb = P.e[rhs];
            const S a_old = P.tab[li * ld + fc], b_old = P.tab[li * ld + rhs];
            const int b = P.eq2bv[li];
            const S a = li == r ? ef : q_fma(P.cn, a_old, kb, ef);
            const S bc = li == r ? eb : q_fma(P.cn, b_old, kb, eb);
            if (lane < R) { if (have_first) P.tab[li * ld + fc] = a; P.tab[li * ld + rhs] = bc; }
            if (last) {
                if (lane == 0) P.sh_w[0] = ACT_TIMEOUT;
            } else if (!have_first) {
                if (lane == 0) { P.sh_w[0] = P.sh_w[6] ? ACT_FINDPAIR : ACT_OPT; P.sh_w[1] = first; P.sh_w[2] = -1; P.sh_w[3] = -1; }
            } else {
                // ratio test (lpsol.h:553-663) on the fresh column, exactly as sm_select_wave0
                const uint32_t w = P.ppt[first * P.pw + (b >> 5)];
                const int cc = P.colcnt[b];
                const bool open = lane < R && !((w >> (b & 31)) & 1u) && cc < lim;
                const bool nonzero = open && !eq(a, zero<S>());
                const S qr = nonzero ? q_div(P.cn, bc, a) : zero<S>();
                int row = wave_argmin_row(qr, nonzero && !le(a, zero<S>()), lane);
                if (row == INT_MAX) row = wave_argmin_row(qr, nonzero, lane);       // relaxed second pass: a != 0
                if (row == INT_MAX) {
                    if (lane == 0) { P.sh_w[0] = ACT_CLOSE; P.sh_w[1] = first; P.sh_w[2] = -1; P.sh_w[3] = -1; }
                } else {
                    const int nleave = __builtin_amdgcn_readlane(b, row);
                    const uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)w, row);
                    const int ccv = __builtin_amdgcn_readlane(cc, row);
                    Cand<S> pa; pa.q = a; pa.idx = 0;
                    Cand<S> po0; po0.q = o0; po0.idx = rc0;
                    Cand<S> po1; po1.q = o1; po1.idx = rc1;
                    const S npiv = read_lane(pa, row).q;
                    const Cand<S> pf = first < 64 ? read_lane(po0, first) : read_lane(po1, first - 64);
                    if (lane == 0) {
                        P.sh_w[0] = ACT_PIVOT; P.sh_w[1] = first; P.sh_w[2] = nleave; P.sh_w[3] = row;
                        P.ppt[first * P.pw + (nleave >> 5)] = wv | (1u << (nleave & 31));      // genPair, lpsol.h:100-104
                        P.rowcnt[first] = pf.idx + 1; P.colcnt[nleave] = ccv + 1;
                        S * pk = (S *)P.sh_c;
                        pk[0] = npiv;
                        pk[1] = pf.q;
                        pk[2] = q_div(P.cn, one<S>(), npiv);
                    }
                }
            }
        } else {
            // the sweep: wave w of the sweepers takes rows w, w + nswaves, ...; a lane owns columns lane and lane + 64
            // (row pairs through ds_read2 with -a_i,nv fetched by v_readlane instead of an LDS read per row were
            // tried: 5 % slower -- the sweep is not what the LDS pipe is short of)
            // Four rows in flight while all four are in range, then the tail row by row; the pivot row goes through
            // the same a + k*e with k_r = 0 (staged so) and is then overwritten with e -- no clamp, select or range
            // test per cell (14 -> 7 instructions per cell on the fp64 ISA).
            if (CR == 32 && CT == 256 && (W == 96 || W == 97)) {
                // the 32 x 96 tableau (32 x 97 during phase one, whose last column is the constant one that wave 0 keeps)
                // on 192 sweeper lanes: two lanes per column, 16 rows each -- every lane busy, row offsets immediates,
                // one -a_i,nv read per cell and no loop control (the generic form below walks 2 x 11 row steps per lane
                // with half the lanes idle in the second column pass)
                const int j = st < 96 ? st : st - 96, h = st < 96 ? 0 : 16;
                if (!(j == rhs || (have_first && j == first))) {
                    const S ej = P.e[j];
                    S * c = P.tab + h * ld + j;
                    const S * kk = P.k + h;
#pragma unroll
                    for (int n = 0; n < 16; n++) c[n * ld] = q_fma(P.cn, c[n * ld], kk[n], ej);
                    if ((r & 16) == h) c[(r & 15) * ld] = ej;
                }
            } else
            for (int j = st & 63; j < W; j += 64) {
                if (j == rhs || (have_first && j == first)) continue;
                const S ej = P.e[j];
                S * col = P.tab + j;
                int i = swave;
                for (; i + 3 * nswaves < R; i += 4 * nswaves) {
                    const int i1 = i + nswaves, i2 = i + 2 * nswaves, i3 = i + 3 * nswaves;
                    const S a0 = col[i * ld], a1 = col[i1 * ld], a2 = col[i2 * ld], a3 = col[i3 * ld];
                    const S k0 = P.k[i], k1 = P.k[i1], k2 = P.k[i2], k3 = P.k[i3];
                    col[i * ld] = q_fma(P.cn, a0, k0, ej);
                    col[i1 * ld] = q_fma(P.cn, a1, k1, ej);
                    col[i2 * ld] = q_fma(P.cn, a2, k2, ej);
                    col[i3 * ld] = q_fma(P.cn, a3, k3, ej);
                }
                for (; i < R; i += nswaves) col[i * ld] = q_fma(P.cn, col[i * ld], P.k[i], ej);
                if (r >= swave && (r - swave) % nswaves == 0) col[r * ld] = ej;
            }
        }
        P.pivots++;
        done++;
        LIFE_MARK((int)blockIdx.x, P.pivots);
#ifdef XPG_STAMPS
        if (tid == 0) FL_ADD(5, wall_clock64() - fl_t2);
        if (tid == 64) FL_ADD(6, wall_clock64() - fl_t2);
#endif
        __syncthreads();
#ifdef XPG_STAMPS
        if (tid == 0) FL_ADD(7, wall_clock64() - fl_t0);
#endif
    }
}

template <class S, int CR, int CLD, int CT> __device__ __forceinline__ void sm_carve_fwd(Small<S> & P, unsigned char * lds);
template <class S, int CR, int CLD, int CT> __device__ __forceinline__ int sm_fast_loop(Small<S> & P0, unsigned max_iter, unsigned & done, bool preselected)
{
    if constexpr (CR != 0) {
        // the specialised instance re-derives every LDS array from ONE base with compile-time offsets (the carve of a
        // 32-row, 63-variable LP is a set of constants): the twenty pointers of Small<S> stop competing for scalar
        // registers -- the run-time form spills them to VGPR lanes and pays v_readlane / v_writelane in the pivot loop
        Small<S> P = P0;
        sm_carve_fwd<S, CR, CLD, CT>(P, (unsigned char *)P0.tab);
        const int action = sm_fast_loop_body<S, CR, CLD, CT>(P, max_iter, done, preselected);
        P0.pivots = P.pivots; P0.closes = P.closes;
        return action;
    } else {
        return sm_fast_loop_body<S, 0, 0, 0>(P0, max_iter, done, preselected);
    }
}

// The specialised instance as a function of its own: its registers are allocated for the pivot loop alone instead of
// together with every cold branch of the solve (inlined, code added to those branches moved spill traffic into the loop:
// 523 -> 600 VALU per pivot, and the kernel's 96 registers held 336-432 bytes of scratch; the function needs 88 registers
// and 40 bytes). The LDS block comes in as an address-space-3 pointer, so the loop's accesses stay ds_* instructions
// behind the call boundary (as generic pointers through a Small<S> reference they became flat loads: 23 k LPs/s);
// scalars by value, results by value. 8192 LPs: 94.4 k -> 108.1 k dependence-test, 284.5 k -> 312.9 k dense LPs/s.
struct FastLoopRet { int action; unsigned pivots, closes, done; };
template <class S> __device__ __noinline__ FastLoopRet sm_fast_loop_32x97x256(__attribute__((address_space(3))) unsigned char * lds, int W, int rhs,
                                                                             bool cn, unsigned pivots, unsigned closes, unsigned max_iter,
                                                                             unsigned done, bool preselected)
{
    Small<S> P;
    sm_carve(P, (unsigned char *)lds, 32, 97 - 32 - 2);
    P.R = 32; P.W = W; P.rhs = rhs; P.cn = cn; P.pivots = pivots; P.closes = closes;
    unsigned d = done;
    FastLoopRet r;
    r.action = sm_fast_loop_body<S, 32, 97, 256>(P, max_iter, d, preselected);
    r.pivots = P.pivots; r.closes = P.closes; r.done = d;
    return r;
}

// SIX::solveSlackForm (lpsol.h:1008-1191) incl. is_feasible (lpsol.h:784-822,
// vc = "-x_i <= 0" for every variable). Returns a SIX_* status; maxv on success.
// Time slices (k_batch): with slice != SM_NO_SLICE the loop hands back SM_SUSPEND at its top once `slice` more iterations
// are done -- everything it needs to go on is then in the LDS arrays and in (done_io, P.pivots, P.closes): the pricing
// scan at the top of the loop (and of the pivot loop, which is told to stop there: no look-ahead after its last pivot)
// starts from the tableau alone. resume: re-entered after such a hand-back (the pair table and counters are the LP's).
enum { SM_SUSPEND = -100 };
#define SM_NO_SLICE 0xFFFFFFFFu
template <class S> __device__ __forceinline__ int sm_solve(Small<S> & P, unsigned max_iter, S & maxv, unsigned & done_io, unsigned slice, bool resume)
{
    const int rhs = P.rhs, lim = rhs - 1;
    if (!resume) {
        for (int i = threadIdx.x; i < rhs; i += blockDim.x) { P.rowcnt[i] = 0; P.colcnt[i] = 0; }
        for (int t = threadIdx.x; t < rhs * P.pw; t += blockDim.x) P.ppt[t] = 0u;
    }
    maxv = zero<S>();
    __syncthreads();
    unsigned done = resume ? done_io : 0u;
    const bool fast = rhs <= 128 && P.R <= 64;
    const bool overlapped = fast && rhs <= 127 && blockDim.x >= 128;
    const bool sliced = slice != SM_NO_SLICE && overlapped;
    const unsigned stop_at = sliced && max_iter - done > slice ? done + slice : max_iter;
    bool preselected = false;                                   // the generic code below has staged a pivot in sh_w
    while (done < max_iter) {
        if (sliced && !preselected && done >= stop_at) { done_io = done; return SM_SUSPEND; }
        if (overlapped) {
            int action;
            // the specialised loop re-derives every array from the tableau base as sm_carve(32, 63) lays them out: the LDS
            // must really have been carved for 32 rows (a MIP node with 32 live rows in a block carved for rmax = 60 has
            // R == 32 and ld == 97 too, and its objective row sits 60 rows behind the base, not 32)
            const bool carved_32x63 = (const unsigned char *)P.obj - (const unsigned char *)P.tab == (ptrdiff_t)32 * 97 * 8;
            if (!is_f64<S>::value || P.R != 32 || P.ld != 97 || blockDim.x != 256 || !carved_32x63 || (P.W != 96 && P.W != 97) || P.rhs != P.W - 1) action = sm_fast_loop<S, 0, 0, 0>(P, stop_at, done, preselected);
            else {
                const FastLoopRet fr = sm_fast_loop_32x97x256<S>((__attribute__((address_space(3))) unsigned char *)P.tab, P.W, P.rhs, P.cn, P.pivots,
                                                                 P.closes, stop_at, done, preselected);
                action = fr.action; P.pivots = fr.pivots; P.closes = fr.closes; done = fr.done;
            }
            preselected = false;
            if (action == ACT_TIMEOUT) {
                if (done >= max_iter) return 4;
                __syncthreads();
                done_io = done;                                 // the slice is over (stop_at < max_iter)
                return SM_SUSPEND;
            }
            if (action == ACT_UNBOUND) return 1;                // SIX_UNBOUND, lpsol.h:1138-1142
            // rare outcomes fall through to the generic code below, which redoes the (idempotent) pricing scan
            __syncthreads();
        } else if (fast) {
            if (threadIdx.x < 64) sm_select_wave0(P);
            __syncthreads();
            const int action = P.sh_w[0];
            if (action == ACT_PIVOT) {
                const int enter_f = P.sh_w[1], leave_f = P.sh_w[2];
                const S * park = (const S *)P.sh_c;
                const S piv = park[0], cnv = park[1];
                sm_pivot_fast(P, enter_f, leave_f, P.sh_w[3], piv, cnv);
                done++;
                continue;
            }
            // rare outcomes fall through to the generic code below, which redoes the (idempotent)
            // pricing scan with all threads
            __syncthreads();
        }
        int first = INT_MAX, anypos = 0;
        for (int j = threadIdx.x; j < rhs; j += blockDim.x)
            if (P.nv[j] && gt(P.obj[j], zero<S>())) {
                anypos = 1;
                if (P.rowcnt[j] < lim) first = min(first, j);
            }
        first = block_min_int(first, P.sh_i);
        if (threadIdx.x == 0) P.sh_w[0] = 0;
        __syncthreads();
        if (anypos) P.sh_w[0] = 1;
        const int stop = first == INT_MAX ? rhs : first;
        for (int j = threadIdx.x; j < stop; j += blockDim.x)
            if (!P.nv[j]) P.obj[j] = zero<S>();
        __syncthreads();
        int enter = -1, leave = -1;
        if (first == INT_MAX) {
            if (!P.sh_w[0]) {
                // optimum: x_B = b, feasibility (lpsol.h:1104-1126)
                if (threadIdx.x == 0) P.sh_w[1] = 0;
                __syncthreads();
                for (int j = threadIdx.x; j < P.W; j += blockDim.x) {
                    S xv = zero<S>();
                    if (j < rhs && P.bv[j]) xv = P.tab[P.bv2eq[j] * P.ld + rhs];
                    P.x[j] = xv;
                    if (j < rhs && gt(q_mul(P.cn, minus_one<S>(), xv), zero<S>())) P.sh_w[1] = 1;
                }
                __syncthreads();
                for (int i = threadIdx.x; i < P.R; i += blockDim.x) {
                    S sum = zero<S>();
                    const S * row = P.tab + i * P.ld;
                    for (int j = 0; j < rhs; j++) sum = q_fma(P.cn, sum, row[j], P.x[j]);
                    reduce(sum);
                    S b = row[rhs];
                    reduce(b);
                    P.tab[i * P.ld + rhs] = b;
                    if (ne(sum, b)) P.sh_w[1] = 1;
                }
                __syncthreads();
                if (P.sh_w[1]) return 3;
                maxv = P.obj[rhs];
                return 0;
            }
            if (overlapped) {
                // by wave 0 with wave-level ratio tests, the chosen pivot staged for the fast loop
                if (threadIdx.x < 64) sm_findpair_wave0(P);
                __syncthreads();
                if (P.sh_w[0] != ACT_PIVOT) return 1;
                preselected = true;
                continue;
            }
            // SIX::findPivotNVandBVPair (lpsol.h:671-773): columns in ascending order, positive reduced costs first,
            // then zero ones; the first whose ratio test finds a row. The scan is 64 columns per ballot (every wave
            // computes the same masks from the same LDS words, so no barrier is needed) instead of one column per
            // dependent LDS round: on dependence-test-like LPs most pivots come from here -- the pair table has
            // exhausted every positive column -- and the serial scan alone was ~190 rounds per pivot.
            for (int pass = 0; pass < 2 && enter < 0; pass++)
                for (int base = 0; base < rhs && enter < 0; base += 64) {
                    const int i = base + (int)(threadIdx.x & 63);
                    bool take = false;
                    if (i < rhs && !P.bv[i] && P.rowcnt[i] < lim) {
                        const S c = P.obj[i];
                        take = gt(c, zero<S>()) ? true : (eq(c, zero<S>()) ? pass == 1 : false);
                    }
                    unsigned long long mask = __ballot(take);
                    while (mask) {
                        const int cand = base + __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const int b = sm_ratio(P, cand);
                        if (b < 0) continue;
                        enter = cand; leave = b;
                        break;
                    }
                }
            if (enter < 0) return 1;
        } else {
            leave = sm_ratio(P, first);
            if (leave < 0) {                                            // lpsol.h:1146-1151
                int add_n = 0;
                for (int j = threadIdx.x; j < rhs; j += blockDim.x) {
                    if (j == first || sm_seen(P, first, j)) continue;
                    atomicOr(&P.ppt[first * P.pw + (j >> 5)], 1u << (j & 31));
                    P.colcnt[j] += 1;
                    add_n++;
                }
                if (add_n) atomicAdd(&P.rowcnt[first], add_n);
                __syncthreads();
                P.closes++;
                continue;
            }
            enter = first;
        }
        if (threadIdx.x == 0 && !sm_seen(P, enter, leave)) {
            P.ppt[enter * P.pw + (leave >> 5)] |= 1u << (leave & 31);
            P.rowcnt[enter] += 1; P.colcnt[leave] += 1;
        }
        __syncthreads();
        sm_pivot(P, enter, leave);
        done++;
    }
    return 4;
}
template <class S> __device__ __forceinline__ int sm_solve(Small<S> & P, unsigned max_iter, S & maxv)
{
    unsigned done = 0;
    return sm_solve<S>(P, max_iter, maxv, done, SM_NO_SLICE, false);
}

// Source of the slack form: the primal (is_max) or the dual built the way
// SIX::calcDualMaxm does (lpsol.h:1602-1629) straight from the caller's arrays.
template <class S> struct Source {
    const S * leq; const S * tgtf; int m, cols, is_max; bool cn;
    __device__ int rows() const { return is_max ? m : cols - 1; }
    __device__ int vars() const { return is_max ? cols - 1 : m; }
    __device__ S A(int i, int j) const
    { return is_max ? leq[i * cols + j] : q_mul(cn, leq[j * cols + i], minus_one<S>()); }
    __device__ S b(int i) const { return is_max ? leq[i * cols + cols - 1] : tgtf[i]; }
    __device__ S c(int j) const
    { return is_max ? tgtf[j] : q_mul(cn, leq[j * cols + cols - 1], minus_one<S>()); }
    __device__ S c0() const { return is_max ? tgtf[cols - 1] : q_mul(cn, zero<S>(), minus_one<S>()); }
};

template <class S> __device__ __forceinline__ void sm_build(Small<S> & P, const Source<S> & src, int with_xa)
{
    const int V = src.vars(), R = src.rows();
    const int first_slack = V + (with_xa ? 1 : 0);
    P.R = R; P.W = first_slack + R + 1; P.rhs = P.W - 1;
    for (int i = 0; i < R; i++)
        for (int j = threadIdx.x; j < P.W; j += blockDim.x) {
            S val = zero<S>();
            if (j < V) val = src.A(i, j);
            else if (with_xa && j == V) val = minus_one<S>();
            else if (j == P.rhs) val = src.b(i);
            else if (j - first_slack == i) val = one<S>();
            P.tab[i * P.ld + j] = val;
        }
    for (int j = threadIdx.x; j < P.W; j += blockDim.x) {
        S val = zero<S>();
        if (with_xa) { if (j == V) val = minus_one<S>(); }
        else if (j < V) val = src.c(j);
        else if (j == P.rhs) val = src.c0();
        P.obj[j] = val;
    }
    for (int i = threadIdx.x; i < P.rhs; i += blockDim.x) {
        const bool slack = i >= first_slack;
        P.nv[i] = slack ? 0 : 1; P.bv[i] = slack ? 1 : 0;
        P.bv2eq[i] = slack ? i - first_slack : -1;
        if (slack) P.eq2bv[i - first_slack] = i;
    }
    __syncthreads();
}

// SIX::constructBasicFeasibleSolution (lpsol.h:839-988). Returns 1 when a
// feasible slack form stands in P, 0 when there is none, -7 where the
// reference's behaviour is undefined.
// In three parts -- the auxiliary LP with x_a pivoted in, its solve, what follows it -- so that k_batch can run the solve
// in time slices (sm_solve_lp).
template <class S> __device__ __forceinline__ void sm_phase_one_pre(Small<S> & P, const Source<S> & src)
{
    const int V = src.vars(), xa = V;
    sm_build(P, src, 1);
    Cand<S> best; best.q = zero<S>(); best.idx = INT_MAX;
    for (int i = threadIdx.x; i < P.R; i += blockDim.x) {
        Cand<S> c; c.q = P.tab[i * P.ld + P.rhs]; c.idx = i;
        best = better(best, c);
    }
    best = block_argmin(best, P.sh_c);
    {
        bool weird = false;
        for (int i = threadIdx.x; i < P.R; i += blockDim.x) weird |= unordered_value(P.tab[i * P.ld + P.rhs]);
        if (__builtin_expect(__syncthreads_or(weird ? 1 : 0), 0)) {   // lpsol.h:894-904 as written: row = 0; if (b[row] > b[i]) row = i
            if (threadIdx.x < 64) {
                int sbest = INT_MAX; S sq = zero<S>();
                for (int base = 0; base < P.R; base += 64) {
                    const int i = min(base + (int)threadIdx.x, P.R - 1);
                    scan_step_in_order(P.tab[i * P.ld + P.rhs], base + (int)threadIdx.x < P.R, base, sbest, sq);
                }
                if (threadIdx.x == 0) P.sh_i[0] = sbest;
            }
            __syncthreads();
            best.idx = P.sh_i[0];
            __syncthreads();
        }
    }
    sm_pivot(P, xa, P.eq2bv[best.idx]);
}
template <class S> __device__ __forceinline__ int sm_phase_one_post(Small<S> & P, const Source<S> & src, int solve_status, S top)
{
    const int V = src.vars(), xa = V;
    if (solve_status != 0) return 0;
    reduce(top);
    if (ne(top, zero<S>())) return 0;
    if (P.bv[xa]) {
        const int r = P.bv2eq[xa];
        if (threadIdx.x == 0) {
            int cand = 0;
            for (; cand < P.rhs; cand++) {
                if (!P.nv[cand]) continue;
                S a = P.tab[r * P.ld + cand];
                reduce(a);
                P.tab[r * P.ld + cand] = a;
                if (ne(a, zero<S>())) break;
            }
            P.sh_w[2] = cand;
        }
        __syncthreads();
        const int cand = P.sh_w[2];
        if (cand >= P.rhs) return -7;
        sm_pivot(P, cand, xa);
    }
    // objective rebuild (lpsol.h:944-953; substit: xmat.cpp:571-599 / :1491-1519)
    const int W = P.W, rhs = P.rhs;
    for (int j = threadIdx.x; j < W; j += blockDim.x)
        P.obj[j] = j < V ? src.c(j) : (j == rhs ? src.c0() : zero<S>());
    __syncthreads();
    for (int i = 0; i < rhs; i++) {
        if (threadIdx.x == 0) { S f = P.obj[i]; reduce(f); P.obj[i] = f; }
        __syncthreads();
        const S f = P.obj[i];
        const bool go = ne(f, zero<S>()) && P.bv[i];
        __syncthreads();
        if (go) {
            const S * expr = P.tab + P.bv2eq[i] * P.ld;
            const S ev = expr[i];
            if (threadIdx.x == 0) P.obj[rhs] = q_mul(P.cn, P.obj[rhs], minus_one<S>());
            __syncthreads();
            if (!eq(ev, zero<S>())) {
                S kk; int mode;
                if (ne(f, ev)) {
                    kk = q_div(P.cn, neg(f), ev);
                    mode = eq(kk, zero<S>()) ? SCALE_ZERO : (eq(kk, one<S>()) ? SCALE_KEEP : SCALE_MUL);
                } else { kk = minus_one<S>(); mode = SCALE_MUL; }
                for (int j = threadIdx.x; j < W; j += blockDim.x)
                    P.obj[j] = q_add(P.cn, q_scaled(P.cn, expr[j], kk, mode), P.obj[j]);
            }
            __syncthreads();
            if (threadIdx.x == 0) P.obj[rhs] = q_mul(P.cn, P.obj[rhs], minus_one<S>());
            __syncthreads();
        }
    }
    // drop column xa (lpsol.h:955-986)
    for (int i = 0; i <= P.R; i++) {
        S * row = i < P.R ? P.tab + i * P.ld : P.obj;
        for (int c0 = xa; c0 < W - 1; c0 += blockDim.x) {
            const int j = c0 + threadIdx.x;
            S t = zero<S>();
            if (j < W - 1) t = row[j + 1];
            __syncthreads();
            if (j < W - 1) row[j] = t;
            __syncthreads();
        }
    }
    for (int c0 = xa; c0 < rhs - 1; c0 += blockDim.x) {
        const int j = c0 + threadIdx.x;
        uint8_t a = 0, b = 0; int q = 0;
        if (j < rhs - 1) { a = P.nv[j + 1]; b = P.bv[j + 1]; q = P.bv2eq[j + 1]; }
        __syncthreads();
        if (j < rhs - 1) { P.nv[j] = a; P.bv[j] = b; P.bv2eq[j] = q; }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < P.R; i += blockDim.x)
        if (P.eq2bv[i] > xa) P.eq2bv[i] -= 1;
    P.W -= 1; P.rhs -= 1;
    __syncthreads();
    return 1;
}
template <class S> __device__ __forceinline__ int sm_phase_one(Small<S> & P, const Source<S> & src, unsigned max_iter)
{
    sm_phase_one_pre(P, src);
    S top;
    const int st = sm_solve<S>(P, max_iter, top);
    return sm_phase_one_post(P, src, st, top);
}

template <class S> __host__ __device__ inline size_t small_lds_bytes(int R, int V)
{
    const int Wmax = V + 1 + R + 1, nmax = Wmax - 1, pw = (nmax + 31) / 32;
    size_t b = (size_t)R * Wmax * 8;      // tab
    b += (size_t)Wmax * 8 * 3;            // obj, e, x
    b += (size_t)((R + 1) & ~1) * 8;      // k
    b += 16 * sizeof(Cand<S>);            // sh_c
    b += (size_t)nmax * 4 * 3;            // bv2eq, rowcnt, colcnt
    b += (size_t)R * 4;                   // eq2bv
    b += (size_t)nmax * pw * 4;           // ppt
    b += 16 * 4 + 8 * 4;                  // sh_i, sh_w
    b += (size_t)((nmax + 3) & ~3) * 2;   // nv, bv
    return (b + 15) & ~(size_t)15;
}

#ifdef XPG_STAMPS
static __device__ unsigned long long g_lp_ticks[8];     // diagnostic builds: ticks in phase one / plain build / main solve, pivots, counts
#endif
// The LDS arrays of one LP with at most R rows and V variables (small_lds_bytes is their size).
template <class S> __device__ __forceinline__ void sm_carve(Small<S> & P, unsigned char * lds, int R, int V)
{
    const int Wmax = V + 1 + R + 1, nmax = Wmax - 1;
    unsigned char * p = lds;
    P.tab = (S *)p; p += (size_t)R * Wmax * 8;
    P.obj = (S *)p; p += (size_t)Wmax * 8;
    P.e = (S *)p; p += (size_t)Wmax * 8;
    P.x = (S *)p; p += (size_t)Wmax * 8;
    P.k = (S *)p; p += (size_t)((R + 1) & ~1) * 8;
    P.sh_c = (Cand<S> *)p; p += 16 * sizeof(Cand<S>);
    P.bv2eq = (int *)p; p += (size_t)nmax * 4;
    P.rowcnt = (int *)p; p += (size_t)nmax * 4;
    P.colcnt = (int *)p; p += (size_t)nmax * 4;
    P.eq2bv = (int *)p; p += (size_t)R * 4;
    P.pw = (nmax + 31) / 32;
    P.ppt = (uint32_t *)p; p += (size_t)nmax * P.pw * 4;
    P.sh_i = (int *)p; p += 16 * 4;
    P.sh_w = (int *)p; p += 8 * 4;
    P.nv = (uint8_t *)p; p += (size_t)((nmax + 3) & ~3);
    P.bv = (uint8_t *)p;
    P.ld = Wmax;
}
template <class S, int CR, int CLD, int CT> __device__ __forceinline__ void sm_carve_fwd(Small<S> & P, unsigned char * lds)
{
    sm_carve(P, lds, CR, CLD - CR - 2);                          // (ld = V + 1 + R + 1)
}

// One LP by the whole workgroup: SIX::maxm / minm of an x >= 0, inequality-only problem (src: m rows, cols - 1
// variables + constant), stage 1 included. Returns the SIX status (or XPG_ERR_REF_UNDEFINED). On status 0 the
// solution goes to sol[0 .. cols) (global; raw_sol: entries not reduced, the caller finishes calcFinalSolution
// itself) and the objective to *v_out; otherwise *v_out = 0 and sol is left alone.
// Time slices (k_batch): slice != SM_NO_SLICE lets the two solves -- stage 1's auxiliary LP and the LP's own -- hand the
// LP back as SM_SUSPEND with (*stage_io, *done_io) saying where; a call with *stage_io != 0 goes on from there (the LDS
// arrays and P's scalars restored by the caller).
template <class S> __device__ __forceinline__ int sm_solve_lp(Small<S> & P, Source<S> & src, unsigned max_iter, int raw_sol,
                                                              S * sol, S * v_out, unsigned slice = SM_NO_SLICE, unsigned * done_io = nullptr,
                                                              int * stage_io = nullptr, unsigned slice1 = SM_NO_SLICE)
{
    const int m = src.m, cols = src.cols, is_max = src.is_max, n = cols - 1;
    const int R = is_max ? m : n, V = is_max ? n : m;
    const int entry = stage_io ? *stage_io : 0;        // 0: a fresh LP, 1: back in stage 1's solve, 2: back in its own
    int status = -1, stage = entry;
    if (entry == 0) {
        P.pivots = 0; P.closes = 0;
        __syncthreads();
        // stage1 trigger (lpsol.h:1794-1803)
        if (threadIdx.x == 0) { P.sh_w[3] = 0; P.sh_w[4] = 0; P.sh_w[5] = 0; }
        __syncthreads();
        if (!is_f64<S>::value) {                       // one non-canonical input cell sends the LP down the generic forms
            bool bad = false;
            for (int t = threadIdx.x; t < m * cols; t += blockDim.x) bad |= !q_canonical(src.leq[t]);
            for (int t = threadIdx.x; t < cols; t += blockDim.x) bad |= !q_canonical(src.tgtf[t]);
            if (bad) P.sh_w[5] = 1;
            __syncthreads();
        }
        P.cn = src.cn = !is_f64<S>::value && P.sh_w[5] == 0;
        for (int j = threadIdx.x; j < V; j += blockDim.x) if (gt(src.c(j), zero<S>())) P.sh_w[3] = 1;
        for (int i = threadIdx.x; i < R; i += blockDim.x) if (lt(src.b(i), zero<S>())) P.sh_w[4] = 1;
        __syncthreads();
        const bool phase1 = !P.sh_w[3] || P.sh_w[4];
        __syncthreads();
        if (phase1) { sm_phase_one_pre(P, src); stage = 1; }
        else { sm_build(P, src, 0); stage = 2; }
    }
#ifdef XPG_STAMPS
    unsigned long long lp_t_ = wall_clock64();
#endif
    if (stage == 1) {
        S top1 = zero<S>();
        unsigned done = entry == 1 ? *done_io : 0u;
        const int st1 = sm_solve<S>(P, max_iter, top1, done, slice1, entry == 1);
        if (st1 == SM_SUSPEND) { *done_io = done; *stage_io = 1; return SM_SUSPEND; }
        const int ok = sm_phase_one_post<S>(P, src, st1, top1);
        if (ok == 0) status = 2;
        else if (ok < 0) status = XPG_ERR_REF_UNDEFINED;
        stage = 2;
#ifdef XPG_STAMPS
        { const unsigned long long n_ = wall_clock64(); if (threadIdx.x == 0) { atomicAdd(&g_lp_ticks[0], n_ - lp_t_); atomicAdd(&g_lp_ticks[4], 1ull); } lp_t_ = n_; }
#endif
    }
    S top = zero<S>();
    if (status == -1) {
        unsigned done = entry == 2 ? *done_io : 0u;
        status = sm_solve<S>(P, max_iter, top, done, slice, entry == 2);
        if (status == SM_SUSPEND) { *done_io = done; *stage_io = 2; return SM_SUSPEND; }
    }
#ifdef XPG_STAMPS
    { const unsigned long long n_ = wall_clock64(); if (threadIdx.x == 0) { atomicAdd(&g_lp_ticks[2], n_ - lp_t_); atomicAdd(&g_lp_ticks[3], (unsigned long long)P.pivots); } }
#endif
    // SIX::calcFinalSolution (lpsol.h:1851-1899) / minm read-out (lpsol.h:1713-1716)
    if (status == 0) {
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            S val = is_max ? P.x[j] : neg(P.obj[m + j]);
            P.e[j] = val;
            if (!raw_sol) reduce(val);          // raw_sol: the host finishes calcFinalSolution itself
            sol[j] = val;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            sol[n] = one<S>();
            S v = zero<S>();
            for (int j = 0; j < n; j++) v = q_fma(P.cn, v, P.e[j], src.tgtf[j]);
            v = q_fma(P.cn, v, one<S>(), src.tgtf[n]);
            reduce(v);
            *v_out = v;
        }
    } else if (threadIdx.x == 0) {
        *v_out = zero<S>();
    }
    __syncthreads();
    return status;
}

// WAVES: waves per SIMD the kernel's registers allow = workgroups of 256 threads per CU. Five where five LPs fit a CU's
// LDS (the 32 x 64 LPs of BASELINE configs[2]: 30 KB each): 96 registers and 336 bytes of scratch instead of 128 and 208,
// and the fifth LP more than pays for the spills (8192 LPs: 92.9 k -> 98.3 k dependence-test, 252.9 k -> 286.4 k dense
// LPs/s) -- the pivot is a latency chain, and what a CU lacks is LPs in flight. Four where LDS seats fewer anyway.
// Time slices (BatchSlices::slice != SM_NO_SLICE): an LP whose own solve has run `slice` iterations gives its LDS slot
// back -- the whole LDS block and seven words go to HBM, its index into a queue -- and CONTINUATION workgroups (the last
// ncont of the grid: dispatched once every LP has had its first turn, resident until all LPs are done) take queued LPs
// in turn, a slice at a time. Lengths of LPs are not known in advance and a launch ends with its longest LPs; taking
// turns, all long LPs advance together and end together, so the launch lasts about (total pivots) / (slots x rate)
// instead of (rounds of long LPs) x (longest LP): 8192 dense 32 x 63 LPs hold 1841 LPs of > 5000 pivots for 1280 slots --
// two rounds of 12 ms with 40 % of the chip idle in the second. Results are the same bit for bit (sm_solve).
// Queue: a ring of (ticket + 1) << 32 | (lp + 1) words; a pusher takes a ticket from ctl[0] and waits for its slot to be
// empty, a popper takes one from ctl[1] and waits for that ticket's word (or for ctl[2] == nb: every LP is done).
struct BatchSlices {
    unsigned slice; unsigned slice1;        // iterations per turn of the LP's own solve / of stage 1's (SM_NO_SLICE: not sliced)
    int nmain; unsigned char * ckpt; unsigned long long stride; unsigned long long * queue; unsigned qmask; unsigned * ctl;
};
enum { CK_HEADER = 64 };
// A watchdog for a state the protocol cannot reach (every LP is in exactly one place: a seat, the queue, or done) --
// but a launch that never ends takes the device with it. It measures LACK OF PROGRESS, not time: a waiting popper or
// pusher restarts its clock whenever the launch as a whole has moved (ctl[0], pushes, or ctl[2], LPs done), so a long
// tail of slow LPs -- Rational LPs of 64 rows under a large max_iter, a shared or preempted device -- never trips it.
// After 20 s without any movement (100 MHz ticks) the waiter raises ctl[3], the launch-wide ABORT word: every popper and
// pusher sees it in its own loop and leaves at once (nobody burns a second 20 s), LPs left behind keep the sentinel
// status XPG_ERR_CHAIN_STUCK the host wrote before the launch, and the host-array callers turn a sentinel they find
// into a call-level error (batch_stuck_check).
#define SLICE_WATCHDOG_TICKS 2000000000ull
template <class S> __device__ __forceinline__ void sm_checkpoint(const Small<S> & P, const unsigned char * lds, size_t lds_bytes, unsigned char * ck, unsigned done, int stage)
{
    const uint4 * src = (const uint4 *)lds;
    uint4 * dst = (uint4 *)(ck + CK_HEADER);
    for (size_t t = threadIdx.x; t < lds_bytes / 16; t += blockDim.x) dst[t] = src[t];
    if (threadIdx.x == 0) {
        unsigned * h = (unsigned *)ck;
        h[0] = done; h[1] = P.pivots; h[2] = P.closes; h[3] = (unsigned)P.R; h[4] = (unsigned)P.W; h[5] = (unsigned)P.rhs; h[6] = P.cn ? 1u : 0u; h[7] = (unsigned)stage;
    }
    __threadfence();                                   // every wave releases its own stores (the pusher's atomic follows the barrier)
    __syncthreads();
}
template <class S> __device__ __forceinline__ unsigned sm_restore(Small<S> & P, unsigned char * lds, size_t lds_bytes, const unsigned char * ck, int & stage)
{
    __threadfence();                                   // acquire: the checkpoint was written by another workgroup of this launch
    const uint4 * src = (const uint4 *)(ck + CK_HEADER);
    uint4 * dst = (uint4 *)lds;
    for (size_t t = threadIdx.x; t < lds_bytes / 16; t += blockDim.x) dst[t] = src[t];
    const unsigned * h = (const unsigned *)ck;
    P.pivots = h[1]; P.closes = h[2]; P.R = (int)h[3]; P.W = (int)h[4]; P.rhs = (int)h[5]; P.cn = h[6] != 0u;
    const unsigned done = h[0];
    stage = (int)h[7];
    __syncthreads();
    return done;
}
__device__ __forceinline__ unsigned slices_progress(const BatchSlices & Q)
{
    return __hip_atomic_load(&Q.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
           __hip_atomic_load(&Q.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool slices_aborted(const BatchSlices & Q)
{ return __hip_atomic_load(&Q.ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
// true: waited SLICE_WATCHDOG_TICKS without the launch moving (the caller raises ABORT); t0 / seen are the waiter's clock
__device__ __forceinline__ bool slices_starved(const BatchSlices & Q, unsigned long long & t0, unsigned & seen)
{
    const unsigned long long now = wall_clock64();
    if (now - t0 <= SLICE_WATCHDOG_TICKS) return false;
    const unsigned p = slices_progress(Q);
    if (p != seen) { seen = p; t0 = now; return false; }
    __hip_atomic_store(&Q.ctl[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
__device__ __forceinline__ void slices_push(const BatchSlices & Q, int lp)
{
    const unsigned t = atomicAdd(&Q.ctl[0], 1u);
    unsigned long long * e = Q.queue + (t & Q.qmask);
    unsigned long long t0 = wall_clock64();
    unsigned seen = slices_progress(Q);
    while (__hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
        // (never seen: the ring holds 2 x nb slots; the LP keeps the sentinel status the host wrote)
        if (slices_aborted(Q) || slices_starved(Q, t0, seen)) return;
        __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_store(e, ((unsigned long long)(t + 1u) << 32) | (unsigned)(lp + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// -1: every LP of the launch is done (or the launch was aborted)
__device__ __forceinline__ int slices_pop(const BatchSlices & Q, int nb)
{
    const unsigned t = atomicAdd(&Q.ctl[1], 1u);
    unsigned long long * e = Q.queue + (t & Q.qmask);
    unsigned long long t0 = wall_clock64();
    unsigned seen = slices_progress(Q);
    for (;;) {
        const unsigned long long w = __hip_atomic_load(e, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w >> 32) == t + 1u) {
            __hip_atomic_store(e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return (int)(unsigned)w - 1;
        }
        if (__hip_atomic_load(&Q.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)nb) return -1;
        if (slices_aborted(Q) || slices_starved(Q, t0, seen)) return -1;
        __builtin_amdgcn_s_sleep(32);
    }
}

// WAVES: waves per SIMD the kernel's registers allow = workgroups of 256 threads per CU. Five where five LPs fit a CU's
// LDS (the 32 x 64 LPs of BASELINE configs[2]: 30 KB each): 96 registers and 336 bytes of scratch instead of 128 and 208,
// and the fifth LP more than pays for the spills (8192 LPs: 92.9 k -> 98.3 k dependence-test, 252.9 k -> 286.4 k dense
// LPs/s) -- the pivot is a latency chain, and what a CU lacks is LPs in flight. Four where LDS seats fewer anyway.
template <class S, int WAVES> __global__ __launch_bounds__(256, WAVES) void k_batch(int nb, const S * tgtf, const S * leq, int m, int cols,
                                           int is_max, unsigned max_iter, int32_t * out_status,
                                           S * out_v, S * out_sol, uint32_t * out_pivots, int raw_sol, BatchSlices Q)
{
    const bool getenv_closes = (raw_sol & 2) != 0;     // profiling: report disableNV iterations instead
    raw_sol &= 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int sh_next;
    const int n = cols - 1;
    Small<S> P;
    const int R0 = is_max ? m : n, V0 = is_max ? n : m;
    sm_carve(P, lds, R0, V0);
    const bool slicing = Q.slice != SM_NO_SLICE;
    const size_t lds_bytes = small_lds_bytes<S>(R0, V0);
    const int nmain = slicing ? Q.nmain : (int)gridDim.x;
    const bool cont = (int)blockIdx.x >= nmain;
    int lp = cont ? -1 : (int)blockIdx.x;
    for (;;) {
        bool resume = false;
        if (cont) {
            __syncthreads();                           // (everybody is through with the LDS block and sh_next)
            if (threadIdx.x == 0) sh_next = slices_pop(Q, nb);
            __syncthreads();
            lp = sh_next;
            if (lp < 0) break;
            resume = true;
        } else if (lp >= nb) break;
        Source<S> src;
        src.leq = leq + (size_t)lp * m * cols; src.tgtf = tgtf + (size_t)lp * cols;
        src.m = m; src.cols = cols; src.is_max = is_max;
        unsigned done = 0;
        int stage = 0;
        if (resume) { done = sm_restore(P, lds, lds_bytes, Q.ckpt + (size_t)lp * Q.stride, stage); src.cn = P.cn; }
        else LIFE_MARK(lp, 0);
        const int status = sm_solve_lp<S>(P, src, max_iter, raw_sol, out_sol + (size_t)lp * cols, out_v + lp, Q.slice, &done, &stage, Q.slice1);
        if (status == SM_SUSPEND) {
            sm_checkpoint(P, lds, lds_bytes, Q.ckpt + (size_t)lp * Q.stride, done, stage);
            if (threadIdx.x == 0) slices_push(Q, lp);
        } else {
            LIFE_END(lp);
            if (threadIdx.x == 0) {
                out_status[lp] = status;
                if (out_pivots) out_pivots[lp] = getenv_closes ? P.closes : P.pivots;
                if (slicing) { __threadfence(); atomicAdd(&Q.ctl[2], 1u); }
            }
        }
        __syncthreads();
        if (!cont) lp += nmain;
    }
}

// The same kernel for LPs of DIFFERENT shapes in one launch (round 3): LP lp is rows[lp] x cols[lp], its inequalities at
// cell leq_off[lp] of leq, its objective -- and its solution slot -- at cell tg_off[lp] of tgtf / out_sol. The LDS
// arrays are carved per LP; the launch reserves the LDS of the largest one.
template <class S> __global__ __launch_bounds__(256, 4) void k_batch_ragged(int nb, const S * tgtf, const S * leq, const int * rows,
                                           const int * cols_of, const long long * leq_off, const long long * tg_off,
                                           int is_max, unsigned max_iter, int32_t * out_status, S * out_v, S * out_sol, int raw_sol)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int lp = blockIdx.x; lp < nb; lp += gridDim.x) {
        const int m = rows[lp], cols = cols_of[lp], n = cols - 1;
        Small<S> P;
        sm_carve(P, lds, is_max ? m : n, is_max ? n : m);
        Source<S> src;
        src.leq = leq + leq_off[lp]; src.tgtf = tgtf + tg_off[lp];
        src.m = m; src.cols = cols; src.is_max = is_max;
        const int status = sm_solve_lp<S>(P, src, max_iter, raw_sol, out_sol + tg_off[lp], out_v + lp);
        if (threadIdx.x == 0) out_status[lp] = status;
        __syncthreads();
    }
}
// Device arrays in and out, enqueue only. max_rows / max_cols: the largest shape of the batch (sizes the LDS request).
template <class S>
int batch_dev_ragged(xpg_ctx * ctx, int is_max, int nb, const S * tgtf, const S * leq, const int * d_rows, const int * d_cols,
                     const long long * d_leq_off, const long long * d_tg_off, int max_R, int max_V, unsigned max_iter,
                     int32_t * out_status, S * out_v, S * out_sol)
{
    if (!ctx || nb < 0 || !tgtf || !leq || !d_rows || !d_cols || !d_leq_off || !d_tg_off || max_R <= 0 || max_V <= 0 ||
        !out_status || !out_v || !out_sol)
        return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const size_t lds = small_lds_bytes<S>(max_R, max_V);
    if (lds > 160 * 1024) return XPG_ERR_UNSUPPORTED;
    const int cells = max_R * (max_V + max_R + 2);
    int threads = cells >= 2048 ? 256 : (cells >= 1024 ? 128 : 64);
    const int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
    int grid = 256 * (per_cu > 16 ? 16 : per_cu) * 64;
    if (grid > nb) grid = nb;
    XPG_HIP(ctx, lds_limit((const void *)k_batch_ragged<S>, ctx->device, lds));
    hipLaunchKernelGGL((k_batch_ragged<S>), dim3(grid), dim3(threads), lds, ctx->stream, nb, tgtf, leq, d_rows, d_cols, d_leq_off,
                       d_tg_off, is_max ? 1 : 0, max_iter, out_status, out_v, out_sol, 0);
    XPG_HIP(ctx, hipGetLastError());
    return 0;
}

template <class S>
int batch_dev(xpg_ctx * ctx, int is_max, int nb, const S * tgtf, const S * leq, int m, int cols,
              unsigned max_iter, int32_t * out_status, S * out_v, S * out_sol, uint32_t * out_pivots,
              int raw_sol = 0)
{
    if (!ctx || nb < 0 || !tgtf || !leq || m <= 0 || cols < 2 || !out_status || !out_v || !out_sol)
        return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const int n = cols - 1;
    const int R = is_max ? m : n, V = is_max ? n : m;
    size_t lds = small_lds_bytes<S>(R, V);
    if (lds > 160 * 1024) return XPG_ERR_UNSUPPORTED;     // one LP must fit one CU's LDS
    if (const char * pad = xpg_hook("XPG_BATCH_LDS_KB")) {  // A/B aid: request more LDS per LP = fewer LPs per CU (occupancy scaling curve)
        const size_t want = (size_t)atoi(pad) * 1024;
        if (want > lds && want <= 160 * 1024) lds = want;
    }
    // measured on MI355X (32x64 LPs): 64 / 128 / 256 threads -> 61k / 91k / 118k LPs/s
    const int cells = R * (V + R + 2);
    int threads = cells >= 2048 ? 256 : (cells >= 1024 ? 128 : 64);
    if (const char * c = xpg_hook("XPG_BATCH_COUNT_CLOSES")) { if (c[0] == '1') raw_sol |= 2; }   // profiling aid
    if (const char * t = xpg_hook("XPG_BATCH_THREADS")) { const int v = atoi(t); if (v >= 64 && v <= 256 && v % 64 == 0) threads = v; }
    const int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
    // One workgroup per LP up to 64 per CU-slot: the hardware's dispatcher then balances LPs of very different
    // lengths (the dependence-test family mixes phase-1 failures of a few dozen pivots with runs of thousands)
    // better than a fixed grid-stride assignment does.
    int grid = 256 * (per_cu > 16 ? 16 : per_cu) * 64;
    if (const char * g = xpg_hook("XPG_BATCH_GRID_X")) { const int v = atoi(g); if (v >= 1) grid = 256 * (per_cu > 16 ? 16 : per_cu) * v; }
    if (grid > nb) grid = nb;
    static const int waves_env = [] { const char * e = xpg_hook("XPG_BATCH_WAVES"); return e ? atoi(e) : 0; }();   // A/B: 4 or 5
    const bool five = (per_cu >= 5 && waves_env != 4) || waves_env == 5;
    // Time slices (k_batch): where the pivot loop can hand an LP back (sm_solve's `overlapped` shapes) and the launch holds
    // more LPs than the chip seats at once, so that LPs wait for slots at all. XPG_BATCH_SLICE=0 turns them off, =n sets
    // the slice (iterations of the LP's own solve per turn; 8192 dense LPs: 416.6 k LPs/s at 256, 418.4 k at 384-512, 414.6 k
    // at 768, 408.5 k at 1024, 313.5 k unsliced).
    static const unsigned slice_env = [] { const char * e = xpg_env("XPG_BATCH_SLICE"); return e ? (unsigned)atoi(e) : 512u; }();
    BatchSlices Q;
    Q.slice = SM_NO_SLICE; Q.slice1 = SM_NO_SLICE; Q.nmain = grid; Q.ckpt = nullptr; Q.stride = 0; Q.queue = nullptr; Q.qmask = 0; Q.ctl = nullptr;
    const int seats = ctx->num_cus * (five ? (per_cu < 5 ? per_cu : 5) : (per_cu < 4 ? per_cu : 4));
    static const bool slice_force = [] { const char * e = xpg_hook("XPG_BATCH_SLICE_FORCE"); return e && e[0] == '1'; }();   // tests: also when every LP has a seat
    if (slice_env != 0u && threads >= 128 && R <= 64 && R + V <= 127 && (nb > seats + seats / 4 || slice_force) && grid == nb) {
        const size_t stride = (CK_HEADER + lds + 255) & ~(size_t)255;
        size_t qcap = 1; while (qcap < (size_t)2 * nb) qcap <<= 1;
        const size_t need = stride * nb + qcap * 8 + 256;
        if (need <= ((size_t)6 << 30)) {
            if (need > ctx->slice_cap) {
                if (ctx->slice_buf) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->slice_buf); ctx->slice_buf = nullptr; ctx->slice_cap = 0; }
                if (hipMalloc(&ctx->slice_buf, need) == hipSuccess) ctx->slice_cap = need; else { ctx->slice_buf = nullptr; (void)hipGetLastError(); }
            }
            if (ctx->slice_buf) {
                unsigned char * base = (unsigned char *)ctx->slice_buf;
                // stage 1's solve in longer turns (8192 dependence-test LPs: 112 k LPs/s at 512, 124.7 k at 1024, 124.2 k at 2048, 122 k
                // at 4096, 121 k unsliced; XPG_BATCH_SLICE_STAGE1=n, 0: not sliced; a forced test slice applies to both)
                static const unsigned slice1_env = [] { const char * e = xpg_hook("XPG_BATCH_SLICE_STAGE1"); return e ? (unsigned)atoi(e) : 1536u; }();
                Q.slice = slice_env; Q.slice1 = slice1_env == 0u ? SM_NO_SLICE : (slice_force ? slice_env : slice1_env); Q.ckpt = base; Q.stride = stride; Q.queue = (unsigned long long *)(base + stride * nb);
                Q.qmask = (unsigned)(qcap - 1); Q.ctl = (unsigned *)(base + stride * nb + qcap * 8);
                XPG_HIP(ctx, hipMemsetAsync(Q.queue, 0, qcap * 8 + 256, ctx->stream));
                XPG_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)out_status, (int)XPG_ERR_CHAIN_STUCK, (size_t)nb, ctx->stream));   // (SLICE_WATCHDOG_TICKS)
                grid += seats;                              // the continuation workgroups, dispatched behind every LP's own
            }
        }
    }
    if (five) {
        XPG_HIP(ctx, lds_limit((const void *)k_batch<S, 5>, ctx->device, lds));
        hipLaunchKernelGGL((k_batch<S, 5>), dim3(grid), dim3(threads), lds, ctx->stream, nb, tgtf, leq, m, cols,
                           is_max ? 1 : 0, max_iter, out_status, out_v, out_sol, out_pivots, raw_sol, Q);
    } else {
        XPG_HIP(ctx, lds_limit((const void *)k_batch<S, 4>, ctx->device, lds));
        hipLaunchKernelGGL((k_batch<S, 4>), dim3(grid), dim3(threads), lds, ctx->stream, nb, tgtf, leq, m, cols,
                           is_max ? 1 : 0, max_iter, out_status, out_v, out_sol, out_pivots, raw_sol, Q);
    }
    XPG_HIP(ctx, hipGetLastError());
    return 0;
}

// A sentinel left in the statuses a sliced launch brought back = the launch aborted (SLICE_WATCHDOG_TICKS): a call-level
// error, not a per-LP status the callers of a batch (has_solution, the MIP controller) would mistake for an answer.
inline int batch_stuck_check(xpg_ctx * ctx, const int32_t * st, int nb)
{
    for (int b = 0; b < nb; b++)
        if (st[b] == (int32_t)XPG_ERR_CHAIN_STUCK) {
            ctx->err = "batched LPs: the time-sliced launch made no progress for 20 s and was aborted (preempted or hung device); no result of this call is valid";
            return XPG_ERR_CHAIN_STUCK;
        }
    return 0;
}

// The same call for a caller that builds its batch itself (the MIP controller, one batch per lock-step round):
// batch_stage_prepare hands out pinned host arrays laid out like the device staging, the caller fills leq / tgtf,
// batch_stage_run copies them down in ONE transfer, launches, and brings status / value / solution back in one.
// (Pageable std::vector staging cost the controller six transfers of ~0.1 ms each per round.) Solutions of LPs
// whose status is not 0 are whatever the slot held before.
template <class S> struct BatchStage {
    int nb, m, cols;
    S * h_leq; S * h_tgtf; S * h_sol; S * h_v; int32_t * h_st;      // pinned host
    S * d_leq; S * d_tgtf; S * d_sol; S * d_v; int32_t * d_st;      // device
    size_t in_bytes, out_bytes;
};
template <class S>
int batch_stage_prepare(xpg_ctx * ctx, int nb, int m, int cols, BatchStage<S> & bs)
{
    if (!ctx || nb <= 0 || m <= 0 || cols < 2) return XPG_ERR_SHAPE;
    const size_t bl = (size_t)nb * m * cols * 8, bt = (size_t)nb * cols * 8;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t need = up(bl) + 2 * up(bt) + up((size_t)nb * 8) + up((size_t)nb * 4);
    if (need > ctx->stage_cap) {
        if (ctx->stage) (void)hipFree(ctx->stage);
        ctx->stage = 0; ctx->stage_cap = 0;
        const size_t cap = need + need / 2;
        if (hipMalloc(&ctx->stage, cap) != hipSuccess) { ctx->err = "hipMalloc(batch staging)"; return XPG_ERR_ALLOC; }
        ctx->stage_cap = cap;
    }
    if (need > ctx->hstage_cap) {
        if (ctx->hstage) (void)hipHostFree(ctx->hstage);
        ctx->hstage = 0; ctx->hstage_cap = 0;
        const size_t cap = need + need / 2;
        if (hipHostMalloc(&ctx->hstage, cap, hipHostMallocDefault) != hipSuccess) { ctx->err = "hipHostMalloc(batch staging)"; return XPG_ERR_ALLOC; }
        ctx->hstage_cap = cap;
    }
    bs.nb = nb; bs.m = m; bs.cols = cols;
    char * d = (char *)ctx->stage; char * h = (char *)ctx->hstage;
    size_t o = 0;
    bs.d_leq = (S *)(d + o); bs.h_leq = (S *)(h + o); o += up(bl);
    bs.d_tgtf = (S *)(d + o); bs.h_tgtf = (S *)(h + o); o += up(bt);
    bs.in_bytes = o;
    bs.d_sol = (S *)(d + o); bs.h_sol = (S *)(h + o); o += up(bt);
    bs.d_v = (S *)(d + o); bs.h_v = (S *)(h + o); o += up((size_t)nb * 8);
    bs.d_st = (int32_t *)(d + o); bs.h_st = (int32_t *)(h + o); o += up((size_t)nb * 4);
    bs.out_bytes = o - bs.in_bytes;
    return 0;
}
template <class S>
int batch_stage_run(xpg_ctx * ctx, BatchStage<S> & bs, int is_max, unsigned max_iter, int raw_sol)
{
    hipError_t e = hipMemcpyAsync(bs.d_leq, bs.h_leq, bs.in_bytes, hipMemcpyHostToDevice, ctx->stream);
    int rc = 0;
    if (e == hipSuccess) rc = batch_dev<S>(ctx, is_max, bs.nb, bs.d_tgtf, bs.d_leq, bs.m, bs.cols, max_iter, bs.d_st, bs.d_v, bs.d_sol, 0, raw_sol);
    if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(bs.h_sol, bs.d_sol, bs.out_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return XPG_ERR_HIP; }
    if (rc == 0) rc = batch_stuck_check(ctx, bs.h_st, bs.nb);
    return rc;
}

// Host-array form: staged through a grow-only scratch area the context owns (one hipMalloc per growth instead of
// five hipMalloc / hipFree pairs per call: the MIP controller makes a call per lock-step round).
template <class S>
int batch_host(xpg_ctx * ctx, int is_max, int nb, const S * tgtf, const S * leq, int m, int cols,
               unsigned max_iter, int32_t * out_status, S * out_v, S * out_sol, int raw_sol = 0)
{
    if (!ctx || nb < 0 || m <= 0 || cols < 2) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const size_t bl = (size_t)nb * m * cols * 8, bt = (size_t)nb * cols * 8;
    if (bl + bt <= ((size_t)1 << 20)) {
        // a small batch -- the drop-in adapter's ONE problem per call above all: through the context's pinned staging, one transfer
        // down, one up (six pageable transfers of a few hundred bytes cost ~200 us per call: bench.py leg one_call, round 6)
        BatchStage<S> bs;
        int rc = batch_stage_prepare<S>(ctx, nb, m, cols, bs);
        if (rc) return rc;
        memcpy(bs.h_leq, leq, bl); memcpy(bs.h_tgtf, tgtf, bt);
        rc = batch_stage_run<S>(ctx, bs, is_max, max_iter, raw_sol);
        if (rc) return rc;
        memcpy(out_status, bs.h_st, (size_t)nb * 4); memcpy(out_v, bs.h_v, (size_t)nb * 8);
        for (int b = 0; b < nb; b++)                                  // (out_sol is written on success only, include/xpoly_amd.h)
            if (bs.h_st[b] == 0) memcpy(out_sol + (size_t)b * cols, bs.h_sol + (size_t)b * cols, (size_t)cols * 8);
        return 0;
    }
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t need = up(bl) + 2 * up(bt) + up((size_t)nb * 8) + up((size_t)nb * 4);
    if (need > ctx->stage_cap) {
        if (ctx->stage) (void)hipFree(ctx->stage);
        ctx->stage = 0; ctx->stage_cap = 0;
        const size_t cap = need + need / 2;
        if (hipMalloc(&ctx->stage, cap) != hipSuccess) { ctx->err = "hipMalloc(batch staging)"; return XPG_ERR_ALLOC; }
        ctx->stage_cap = cap;
    }
    char * p = (char *)ctx->stage;
    S * d_leq = (S *)p; p += up(bl);
    S * d_tgtf = (S *)p; p += up(bt);
    S * d_sol = (S *)p; p += up(bt);
    S * d_v = (S *)p; p += up((size_t)nb * 8);
    int32_t * d_st = (int32_t *)p;
    int rc = 0;
    hipError_t e = hipMemcpyAsync(d_leq, leq, bl, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tgtf, tgtf, bt, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sol, out_sol, bt, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = batch_dev<S>(ctx, is_max, nb, d_tgtf, d_leq, m, cols, max_iter, d_st, d_v, d_sol, 0, raw_sol);
    if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(out_status, d_st, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(out_v, d_v, (size_t)nb * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(out_sol, d_sol, bt, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return XPG_ERR_HIP; }
    if (rc == 0) rc = batch_stuck_check(ctx, out_status, nb);
    return rc;
}


} // namespace xpg
