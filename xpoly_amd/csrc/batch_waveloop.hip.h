// One wavefront per LP: the register-resident pivot loop of the batched small-LP kernel for
// R <= 32 rows and W <= 128 columns (the dependence-test shapes; included by batch_kernels.hip.h).
//
// Lane l owns columns l and l + 64 for all 32 (padded) rows: 2 x 32 fp64/rational cells = 128
// VGPRs, held as four 16-row ext_vectors (a 32-dword tuple is the largest legal vector register;
// anything larger, or a C array, ends up in scratch memory -- measured). The rank-1 update
// (lpsol.h:1481-1490) is then a mul+add per cell with no LDS traffic but the k_i broadcasts, and,
// with a single wave, NOTHING in the loop waits for another wave: the workgroup barriers of the
// multi-wave loop (four per pivot, ~60 % of its time) are gone, and the LPs sharing a CU run
// completely independently of each other.
// LDS carries only what crosses lanes in a pivot: the entering column (P.k, written by its owner
// lane), the constant column (kept current in P.x with the sweep's own
// arithmetic), the objective row, basis maps and the pair table. The sweep runs over all rows
// unconditionally (k_i negated in the register, lpsol.h:1485); the scaled pivot row is put back
// afterwards. Row r of a lane's registers is reached with log2(16) select masks (wave-uniform
// index), never with a dynamic vector index.
// Leaves -- tableau written back to LDS -- for everything that wants the whole tableau (optimum
// check, findPivotNVandBVPair) or when the iteration budget is spent.
#pragma once

namespace xpg {

typedef unsigned long long u64_t;
typedef u64_t V16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ u64_t pick16(const V16 & v, int k4)
{
    u64_t t[8];
    {
        const bool hi = (k4 & 8) != 0;
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = hi ? v[k + 8] : v[k];
    }
#pragma unroll
    for (int span = 4; span >= 1; span /= 2) {
        const bool hi = (k4 & span) != 0;
#pragma unroll
        for (int k = 0; k < span; k++) t[k] = hi ? t[k + span] : t[k];
    }
    return t[0];
}
__device__ __forceinline__ void put16(V16 & v, int k4, u64_t x)
{
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = (k == k4) ? x : v[k];
}

// rows [16*H, 16*H + 16) of column `col` from / to the LDS tableau
template <class S, int H> __device__ __forceinline__ void load16(V16 & v, const Small<S> & P, int col, bool owns)
{
    const S * p = P.tab + (16 * H) * P.ld + (owns ? col : 0);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v[k] = to_bits<S>((owns && 16 * H + k < P.R) ? *p : zero<S>());
        p += P.ld;
    }
}
template <class S, int H> __device__ __forceinline__ void store16(const V16 & v, const Small<S> & P, int col, bool owns)
{
    S * p = P.tab + (16 * H) * P.ld + (owns ? col : 0);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (owns && 16 * H + k < P.R) *p = from_bits<S>(v[k]);
        p += P.ld;
    }
}
template <class S, int H> __device__ __forceinline__ void export16(const V16 & v, S * kcol)
{
#pragma unroll
    for (int k = 0; k < 16; k++) kcol[16 * H + k] = from_bits<S>(v[k]);
}
// v[k] += (-a_k) * e for the 16 rows, four k_i in flight at a time
template <class S, int H> __device__ __forceinline__ void sweep16(V16 & v, const S * kcol, S e)
{
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) {
        S kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) kv[u] = neg(kcol[16 * H + k0 + u]);          // -a_i,nv (lpsol.h:1485)
#pragma unroll
        for (int u = 0; u < 4; u++) v[k0 + u] = to_bits<S>(add(from_bits<S>(v[k0 + u]), mul(kv[u], e)));
    }
}
template <class S, int H> __device__ __forceinline__ void sweep16x2(V16 & va, V16 & vb, const S * kcol, S ea, S eb)
{
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) {
        S kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) kv[u] = neg(kcol[16 * H + k0 + u]);          // -a_i,nv (lpsol.h:1485)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            va[k0 + u] = to_bits<S>(add(from_bits<S>(va[k0 + u]), mul(kv[u], ea)));
            vb[k0 + u] = to_bits<S>(add(from_bits<S>(vb[k0 + u]), mul(kv[u], eb)));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// TWO: the tableau is wider than 64 columns (lane l also owns column l + 64).
template <class S, bool TWO> __device__ __forceinline__
int sm_wave_loop(Small<S> & P, unsigned max_iter, unsigned & done)
{
    const int lane = threadIdx.x;
    const int R = P.R, W = P.W, rhs = P.rhs, lim = rhs - 1;
    S * bcol = P.x;
    const bool own0 = lane < W, own1 = TWO && lane + 64 < W;
    V16 a0, a1, b0, b1;                            // column lane: rows 0-15, 16-31; column lane+64 likewise
    load16<S, 0>(a0, P, lane, own0); load16<S, 1>(a1, P, lane, own0);
    if constexpr (TWO) { load16<S, 0>(b0, P, lane + 64, own1); load16<S, 1>(b1, P, lane + 64, own1); }
    if (lane < R) bcol[lane] = P.tab[lane * P.ld + rhs];
    __syncthreads();
    // lane-private indices, clamped so that every load below is unconditional
    const int j0 = lane, j1 = lane + 64;
    const bool in0 = j0 < rhs, in1 = j1 < rhs;
    const int q0 = in0 ? j0 : 0, q1 = in1 ? j1 : 0;
    const int li = lane < R ? lane : 0;
    int action;
#ifdef XPG_EXP_STAMPS       /* diagnostic build: cycles per phase (tools/probe_stamps_reg.py) */
    long long st_[5] = {0, 0, 0, 0, 0};
#define WL_STAMP(q) do { const long long now_ = clock64(); st_[q] += now_ - t_last_; t_last_ = now_; } while (0)
#else
#define WL_STAMP(q) do { } while (0)
#endif
    for (;;) {
#ifdef XPG_EXP_STAMPS
        long long t_last_ = clock64();
#endif
        if (done >= max_iter) { action = ACT_BUDGET; break; }
        // ---- pricing (lpsol.h:1054-1069): one LDS round + ballots
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
        if (first == INT_MAX) { action = any ? ACT_FINDPAIR : ACT_OPT; break; }
        WL_STAMP(0);
        // ---- the entering column, out of its owner lane's registers (rows past R: zeros)
        if (lane == (first & 63)) {
            if (!TWO || first < 64) { export16<S, 0>(a0, P.k); export16<S, 1>(a1, P.k); }
            else { export16<S, 0>(b0, P.k); export16<S, 1>(b1, P.k); }
        }
        __syncthreads();
        WL_STAMP(1);
        // ---- ratio test (lpsol.h:553-663): two LDS rounds, one division
        const S a = P.k[li], bc = bcol[li];
        const int b = P.eq2bv[li];
        const uint32_t w = P.ppt[first * P.pw + (b >> 5)];
        const int cc = P.colcnt[b];
        const bool open = lane < R && !((w >> (b & 31)) & 1u) && cc < lim;
        const bool nonzero = open && !eq(a, zero<S>());
        Cand<S> c; c.q = nonzero ? div(bc, a) : zero<S>();
        c.idx = (nonzero && !le(a, zero<S>())) ? lane : INT_MAX;
        Cand<S> best = wave_argmin(c);
        if (best.idx == INT_MAX) {                                       // relaxed second pass: a != 0
            c.idx = nonzero ? lane : INT_MAX;
            best = wave_argmin(c);
        }
        const int r = __builtin_amdgcn_readfirstlane(best.idx);
        if (r == INT_MAX) {                                              // disableNV, lpsol.h:1146-1151
            int add_n = 0;
            for (int j = lane; j < rhs; j += 64) {
                if (j == first || sm_seen(P, first, j)) continue;
                atomicOr(&P.ppt[first * P.pw + (j >> 5)], 1u << (j & 31));
                P.colcnt[j] += 1;
                add_n++;
            }
            if (add_n) atomicAdd(&P.rowcnt[first], add_n);
            P.closes++;
            __syncthreads();
            continue;
        }
        // everything the bookkeeping needs is in some lane's registers already
        const int leave = __builtin_amdgcn_readlane(b, r);
        const uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)w, r);
        const int ccv = __builtin_amdgcn_readlane(cc, r);
        const S piv = readlane_s(a, r);
        const S cnv = first < 64 ? readlane_s(ob0, first) : readlane_s(ob1, first - 64);
        const int rcf = first < 64 ? __builtin_amdgcn_readlane(rc0, first) : __builtin_amdgcn_readlane(rc1, first - 64);
        if (lane == 0) {
            // genPair (lpsol.h:100-104): a candidate row was by construction not yet paired
            P.ppt[first * P.pw + (leave >> 5)] = wv | (1u << (leave & 31));
            P.rowcnt[first] = rcf + 1; P.colcnt[leave] = ccv + 1;
            P.nv[first] = 0; P.nv[leave] = 1; P.bv[first] = 1; P.bv[leave] = 0;     // lpsol.h:1504-1510
            P.eq2bv[r] = first; P.bv2eq[first] = r; P.bv2eq[leave] = -1;
        }
        WL_STAMP(2);
        // ---- SIX::pivot (lpsol.h:1456-1511): scaled pivot row out of register row r
        const S s = div(one<S>(), piv);
        const int smode = scale_mode(s), cmode = scale_mode(cnv);
        const int r4 = r & 15;
        S e0, e1 = zero<S>();
        if (r < 16) e0 = from_bits<S>(pick16(a0, r4)); else e0 = from_bits<S>(pick16(a1, r4));
        e0 = scaled(e0, s, smode);
        if constexpr (TWO) {
            if (r < 16) e1 = from_bits<S>(pick16(b0, r4)); else e1 = from_bits<S>(pick16(b1, r4));
            e1 = scaled(e1, s, smode);
        }
        // e_rhs for the constant column: straight out of its owner lane's register
        const S erhs = (!TWO || rhs < 64) ? readlane_s(e0, rhs & 63) : readlane_s(e1, rhs - 64);
        WL_STAMP(3);
        // ---- rank-1 update over all rows (P.k is zero-padded past R), then row r := scaled row
        if constexpr (TWO) { sweep16x2<S, 0>(a0, b0, P.k, e0, e1); sweep16x2<S, 1>(a1, b1, P.k, e0, e1); }
        else { sweep16<S, 0>(a0, P.k, e0); sweep16<S, 1>(a1, P.k, e0); }
        if (r < 16) { put16(a0, r4, to_bits<S>(e0)); if constexpr (TWO) put16(b0, r4, to_bits<S>(e1)); }
        else { put16(a1, r4, to_bits<S>(e0)); if constexpr (TWO) put16(b1, r4, to_bits<S>(e1)); }
        {                                                                // objective row, lpsol.h:1496-1501
            S t = mul(e0, minus_one<S>());
            if (j0 >= rhs) t = neg(t);
            t = scaled(t, cnv, cmode);
            if (own0) P.obj[j0] = add(t, P.obj[j0]);
            if constexpr (TWO) {
                S t1 = mul(e1, minus_one<S>());
                if (j1 >= rhs) t1 = neg(t1);
                t1 = scaled(t1, cnv, cmode);
                if (own1) P.obj[j1] = add(t1, P.obj[j1]);
            }
            if (lane < R) bcol[lane] = lane == r ? erhs : add(bcol[lane], mul(neg(P.k[lane]), erhs));
        }
        P.pivots++;
        done++;
        __syncthreads();
        WL_STAMP(4);
    }
#ifdef XPG_EXP_STAMPS
    if (lane == 0) {
        extern __shared__ __attribute__((aligned(16))) unsigned char lds_st[];
        int * dd = (int *)lds_st + 32;
        for (int q = 0; q < 5; q++) atomicAdd(&dd[q], (int)(st_[q] >> 4));
    }
#endif
#undef WL_STAMP
    store16<S, 0>(a0, P, lane, own0); store16<S, 1>(a1, P, lane, own0);
    if constexpr (TWO) { store16<S, 0>(b0, P, lane + 64, own1); store16<S, 1>(b1, P, lane + 64, own1); }
    __syncthreads();
    return action;
}

} // namespace xpg
