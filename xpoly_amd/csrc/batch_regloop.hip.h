// Register-resident pivot loop of the batched small-LP kernel (included by batch_kernels.hip.h
// after Small<S>, sm_seen and wave_argmin).
//
// For the dependence-test shapes the tableau leaves LDS for the duration of SIX::solveSlackForm's
// loop. A workgroup of BT threads is CW = BT / G columns wide and G row groups deep: thread t owns
// the NC columns (t % CW) + c*CW, c < NC, for the rows g + G*k (g = t / CW, k < RT) in NC*RT
// registers, so the rank-1 update (lpsol.h:1481-1490) is a mul+add per cell with no LDS traffic
// but the k_i broadcasts. Two shapes are instantiated:
//   BT = 64,  G = 1, NC = 1|2, RT = 32: ONE wavefront per LP (R <= 32, W <= 128). No workgroup
//             barrier does anything, nothing waits for another wave, 2 waves/SIMD of registers;
//   BT = 256, G = 2|4, NC = 1, RT <= 32: four wavefronts per LP for up to 64 rows.
// LDS carries only what crosses lanes in a pivot: the entering column (P.k, exported by its
// owners), the scaled pivot row (P.e), the constant column (kept current in P.x with the sweep's
// own arithmetic), the objective row, basis maps and the pair table.
//
// The loop is a latency chain, so every stage issues all its LDS loads before the first use
// (the reference's short-circuit tests would serialise them) and hands values on through
// v_readlane instead of re-reading LDS: pricing is one LDS round + ballots, the ratio test two
// rounds + one fp64 division + a DPP arg-min, its bookkeeping pure stores. The sweep runs over
// ALL rows unconditionally (k_i negated in the register, lpsol.h:1485) and the scaled pivot row
// is put back afterwards, which is cheaper than a test per cell.
// Leaves -- tableau written back to LDS -- for everything that wants the whole tableau (optimum
// check, findPivotNVandBVPair) or when the iteration budget is spent.
#pragma once

namespace xpg {

enum { ACT_BUDGET = 4 };

template <class S> __device__ __forceinline__ S readlane_s(S v, int lane)
{
    int w[2];
    __builtin_memcpy(w, &v, 8);
    w[0] = __builtin_amdgcn_readlane(w[0], lane); w[1] = __builtin_amdgcn_readlane(w[1], lane);
    S t;
    __builtin_memcpy(&t, w, 8);
    return t;
}

// The row registers of a thread are one ext_vector of 64-bit lanes: a C array is left in scratch
// memory by the compiler here (measured: every rank-1 update went through scratch_load /
// scratch_store), a vector type is not.
template <int N> struct RowRegs { typedef unsigned long long type __attribute__((ext_vector_type(N))); };

// v[BASE + kr] for a wave-uniform kr by halving on the bits of kr: log2(N) select masks. (A dynamic
// vector index makes the compiler dump the whole vector to scratch; a tree of branches makes it
// copy the vector per leaf.)
template <int BASE, int N, class V> __device__ __forceinline__ unsigned long long row_pick(const V & v, int kr)
{
    if constexpr (N < 2) return v[BASE];
    unsigned long long t[N / 2 > 0 ? N / 2 : 1];
    {
        const bool hi = (kr & (N / 2)) != 0;
#pragma unroll
        for (int k = 0; k < N / 2; k++) t[k] = hi ? v[BASE + k + N / 2] : v[BASE + k];
    }
#pragma unroll
    for (int span = N / 4; span >= 1; span /= 2) {
        const bool hi = (kr & span) != 0;
#pragma unroll
        for (int k = 0; k < span; k++) t[k] = hi ? t[k + span] : t[k];
    }
    return t[0];
}
template <int BASE, int N, class V> __device__ __forceinline__ void row_put(V & v, int kr, unsigned long long x)
{
    constexpr int STEP = N < 8 ? N : 8;
#pragma unroll
    for (int k0 = 0; k0 < N; k0 += STEP) {
#pragma unroll
        for (int u = 0; u < STEP; u++) v[BASE + k0 + u] = (k0 + u == kr) ? x : v[BASE + k0 + u];
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <class S, int RT, int G, int NC, int BT> __device__ __forceinline__
int sm_reg_loop(Small<S> & P, unsigned max_iter, unsigned & done)
{
    constexpr int CW = BT / G, NW = BT / 64;
    const int tid = threadIdx.x, lane = tid & 63, jcol = tid % CW, g = tid / CW;
    const int R = P.R, W = P.W, ld = P.ld, rhs = P.rhs, lim = rhs - 1;
    S * bcol = P.x;
    typename RowRegs<NC * RT>::type reg;
#define REG_GET(c, k) from_bits<S>(reg[(c) * RT + (k)])
#define REG_SET(c, k, v) reg[(c) * RT + (k)] = to_bits<S>(v)
    bool owns[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        owns[c] = jcol + c * CW < W;
        const S * p = P.tab + g * ld + (owns[c] ? jcol + c * CW : 0);
#pragma unroll
        for (int k = 0; k < RT; k++) {
            REG_SET(c, k, (owns[c] && g + G * k < R) ? *p : zero<S>());
            p += G * ld;
        }
    }
    for (int i = tid; i < R; i += BT) bcol[i] = P.tab[i * ld + rhs];
    __syncthreads();
    // lane-private indices, clamped so that every load below is unconditional
    const int j0 = lane, j1 = lane + 64;
    const bool in0 = j0 < rhs, in1 = j1 < rhs;
    const int q0 = in0 ? j0 : 0, q1 = in1 ? j1 : 0;
    const int li = lane < R ? lane : 0;
    const S * kmine = P.k + g;                     // k_i of my rows: kmine[G * k]
    int action;
#ifdef XPG_EXP_STAMPS       /* diagnostic build: cycles per phase, first and last wave (tools/probe_stamps_reg.py) */
    long long st_[5] = {0, 0, 0, 0, 0};
#define RL_STAMP(q) do { const long long now_ = clock64(); st_[q] += now_ - t_last_; t_last_ = now_; } while (0)
#else
#define RL_STAMP(q) do { } while (0)
#endif
    // The serial part of a pivot (zeroing, ratio test, bookkeeping) is done by ONE wave; with several
    // waves per LP it rotates, so that the selections of the LPs sharing a CU spread over its SIMDs.
    unsigned it = 0;
    for (;; it++) {
#ifdef XPG_EXP_STAMPS
        long long t_last_ = clock64();
#endif
        const bool selector = NW == 1 || (unsigned)(tid >> 6) == (it & (unsigned)(NW - 1));
        if (done >= max_iter) { action = ACT_BUDGET; break; }
        // ---- pricing (lpsol.h:1054-1069): every wave for itself, one LDS round, no barrier
        const int nv0 = P.nv[q0], nv1 = P.nv[q1];
        const S ob0 = P.obj[q0], ob1 = P.obj[q1];
        const int rc0 = P.rowcnt[q0], rc1 = P.rowcnt[q1];
        const bool nb0 = in0 && nv0 != 0, nb1 = in1 && nv1 != 0;
        const bool c0 = nb0 && gt(ob0, zero<S>()), c1 = nb1 && gt(ob1, zero<S>());
        const bool o0 = c0 && rc0 < lim, o1 = c1 && rc1 < lim;
        const unsigned long long m0 = __ballot(o0), m1 = __ballot(o1), any = __ballot(c0 || c1);
        const int first = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : INT_MAX);
        const int stop = first == INT_MAX ? rhs : first;
        if (selector) {                                                  // lpsol.h:1055-1060
            if (in0 && j0 < stop && !nb0) P.obj[j0] = zero<S>();
            if (in1 && j1 < stop && !nb1) P.obj[j1] = zero<S>();
        }
        if (first == INT_MAX) { action = any ? ACT_FINDPAIR : ACT_OPT; break; }
        RL_STAMP(0);
        // ---- the entering column, out of its owners' registers (rows past R: zeros into the padding)
        {
            const int cf = first / CW;                                   // wave-uniform
            if (jcol == first - cf * CW) {
                S * p = P.k + g;
#pragma unroll
                for (int k = 0; k < RT; k++) {
                    S v = REG_GET(0, k);
#pragma unroll
                    for (int c = 1; c < NC; c++) if (cf == c) v = REG_GET(c, k);
                    p[G * k] = v;
                }
            }
        }
        __syncthreads();
        RL_STAMP(1);
        if (selector) {
            // ---- ratio test (lpsol.h:553-663) by one wave: two LDS rounds, one division
            const S a = P.k[li], bc = bcol[li];
            const int b = P.eq2bv[li];
            const uint32_t w = P.ppt[first * P.pw + (b >> 5)];
            const int cc = P.colcnt[b];
            const bool open = lane < R && !((w >> (b & 31)) & 1u) && cc < lim;
            const bool nonzero = open && !eq(a, zero<S>());
            Cand<S> c; c.q = nonzero ? div(bc, a) : zero<S>();
            c.idx = (nonzero && !le(a, zero<S>())) ? lane : INT_MAX;
            Cand<S> best = wave_argmin(c);
            if (best.idx == INT_MAX) {                                   // relaxed second pass: a != 0
                c.idx = nonzero ? lane : INT_MAX;
                best = wave_argmin(c);
            }
            const int row = __builtin_amdgcn_readfirstlane(best.idx);
            if (row == INT_MAX) {
                if (lane == 0) P.sh_w[0] = ACT_CLOSE;
            } else {
                // everything the bookkeeping needs is in some lane's registers already
                const int leave = __builtin_amdgcn_readlane(b, row);
                const uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)w, row);
                const int ccv = __builtin_amdgcn_readlane(cc, row);
                const S piv = readlane_s(a, row);
                const S cnv = first < 64 ? readlane_s(ob0, first) : readlane_s(ob1, first - 64);
                const int rcf = first < 64 ? __builtin_amdgcn_readlane(rc0, first) : __builtin_amdgcn_readlane(rc1, first - 64);
                if (lane == 0) {
                    P.sh_w[0] = ACT_PIVOT; P.sh_w[2] = leave; P.sh_w[3] = row;
                    // genPair (lpsol.h:100-104): a candidate row was by construction not yet paired
                    P.ppt[first * P.pw + (leave >> 5)] = wv | (1u << (leave & 31));
                    P.rowcnt[first] = rcf + 1; P.colcnt[leave] = ccv + 1;
                    S * park = (S *)P.sh_c;
                    park[0] = piv; park[1] = cnv;
                }
            }
        }
        __syncthreads();
        RL_STAMP(2);
        if (P.sh_w[0] == ACT_CLOSE) {                                    // disableNV, lpsol.h:1146-1151
            int add_n = 0;
            for (int j = tid; j < rhs; j += BT) {
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
        // ---- SIX::pivot (lpsol.h:1456-1511)
        const int leave = __builtin_amdgcn_readfirstlane(P.sh_w[2]), r = __builtin_amdgcn_readfirstlane(P.sh_w[3]);
        const S piv = ((const S *)P.sh_c)[0], cnv = ((const S *)P.sh_c)[1];
        const S s = div(one<S>(), piv);
        const int smode = scale_mode(s), cmode = scale_mode(cnv);
        const int kr = r / G, gr = r % G;
        S ej[NC];
        if (g == gr) {                                                   // the owners of row r scale it
            ej[0] = scaled(from_bits<S>(row_pick<0, RT>(reg, kr)), s, smode);
            if (owns[0]) P.e[jcol] = ej[0];
            if constexpr (NC == 2) {
                ej[1] = scaled(from_bits<S>(row_pick<RT, RT>(reg, kr)), s, smode);
                if (owns[1]) P.e[jcol + CW] = ej[1];
            }
        }
        __syncthreads();
        RL_STAMP(3);
        if (g != gr) {
#pragma unroll
            for (int c = 0; c < NC; c++) ej[c] = P.e[owns[c] ? jcol + c * CW : 0];
        }
        // rank-1 update in registers over all rows, k_i through immediate-offset LDS reads (P.k is
        // padded with zeros to G*RT rows; rows past R compute on zeros and are never stored)
#pragma unroll
        for (int k0 = 0; k0 < RT; k0 += 4) {
            S kv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) kv[u] = neg(kmine[G * (k0 + u)]);    // -a_i,nv (lpsol.h:1485)
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int u = 0; u < 4; u++) REG_SET(c, k0 + u, add(REG_GET(c, k0 + u), mul(kv[u], ej[c])));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g == gr) {                                                   // row r is the scaled row, not swept
            row_put<0, RT>(reg, kr, to_bits<S>(ej[0]));
            if constexpr (NC == 2) row_put<RT, RT>(reg, kr, to_bits<S>(ej[1]));
        }
        {                                                                // objective row, lpsol.h:1496-1501
            const S erhs = P.e[rhs];
            for (int j = tid; j < W; j += BT) {
                S t = mul(P.e[j], minus_one<S>());
                if (j >= rhs) t = neg(t);
                t = scaled(t, cnv, cmode);
                P.obj[j] = add(t, P.obj[j]);
            }
            if (tid < R) bcol[tid] = tid == r ? erhs : add(bcol[tid], mul(neg(P.k[tid]), erhs));
        }
        if (tid == 0) {
            P.nv[first] = 0; P.nv[leave] = 1; P.bv[first] = 1; P.bv[leave] = 0;
            P.eq2bv[r] = first; P.bv2eq[first] = r; P.bv2eq[leave] = -1;
        }
        P.pivots++;
        done++;
        __syncthreads();
        RL_STAMP(4);
    }
#ifdef XPG_EXP_STAMPS
    if (lane == 0 && (tid == 0 || tid == BT - 64)) {
        extern __shared__ __attribute__((aligned(16))) unsigned char lds_st[];
        int * dd = (int *)lds_st + 32 + ((tid && NW > 1) ? 5 : 0);
        for (int q = 0; q < 5; q++) atomicAdd(&dd[q], (int)(st_[q] >> 4));
    }
#endif
#undef RL_STAMP
#pragma unroll
    for (int c = 0; c < NC; c++) {
        S * p = P.tab + g * ld + (owns[c] ? jcol + c * CW : 0);
#pragma unroll
        for (int k = 0; k < RT; k++) {
            if (owns[c] && g + G * k < R) *p = REG_GET(c, k);
            p += G * ld;
        }
    }
    __syncthreads();
#undef REG_GET
#undef REG_SET
    return action;
}

} // namespace xpg
