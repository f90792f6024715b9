// Blocked fp64 simplex loop: B pivots per pass over the tableau.
//
// The rank-1 sweep (lpsol.h:1481-1490) is HBM-bound at 2*m*W*8 bytes per pivot. Nothing in
// SIX::solveSlackForm needs the swept tableau to CHOOSE the next pivot except one row and two
// columns of it -- and those can be had from the un-swept tableau by replaying the pending
// updates on just that row / column with the sweep's own arithmetic:
//     x := tab[i][c];  for each staged pivot s:  x := (i == r_s) ? e_s[c] : x + k_s[i] * e_s[c]
// (k_s = -column of pivot s as the reference negates it, e_s = its scaled row; both rounded exactly
// as a sweep would have left them, because every operand is itself produced this way). So a batch
// stages up to B pivots -- pick, prep, pick, prep, ... each O((m + W) * staged) -- and then ONE
// sweep applies them all to every cell in registers, in order, with the same two roundings per
// update:
//     a := tab[i][j];  for s < n:  a := (i == r_s) ? e_s[j] : a + k_s[i] * e_s[j];  tab[i][j] := a
// HBM traffic per pivot drops by the batch length; results stay bit-identical to n separate sweeps.
//
// Per batch (host-enqueued, no host round trip), t = 0 .. B-1:
//   k_blk_pick(t)        runs iff exactly t pivots are staged: ratio test (lpsol.h:553-663) on the
//                        replayed entering and constant columns by <= 16 workgroups with the record /
//                        last-adder protocol of the pipelined loop; commits the basis swap at once (the
//                        chain is serial), stages k = -column, makes the batch t + 1 long
//   k_blk_pick_generic   after pick(0) only, when that asked for it: the generic single-workgroup
//                        pick_body for whatever the fast path does not do (second ratio pass, disableNV,
//                        findPivotNVandBVPair, optimum, iteration limits) -- legal because nothing is
//                        staged, i.e. the tableau is fully swept
//   k_blk_prep(t)        runs iff t + 1 pivots are staged: replayed pivot row -> scaled row e_t,
//                        objective row (with the zeroing of lpsol.h:1055-1060), look-ahead pricing
//   k_blk_sweep          applies the staged pivots
// A pick that cannot take the fast path while pivots are staged closes the batch: the remaining
// launches of the batch do nothing, the sweep applies what is staged and the next batch starts with
// pick(0) + generic on a swept tableau.
#pragma once
#include "lp_kernels.hip.h"

namespace xpg {

// State of `batch` as every workgroup of a kernel sees it at its start (uniform).
struct BlkView { int n; bool closed; };
__device__ __forceinline__ BlkView blk_view(const LoopState * st, int batch)
{
    BlkView b;
    const bool mine = st->blk.batch == batch;
    b.n = mine ? st->blk.n : 0;
    b.closed = mine && st->blk.closed != 0;
    return b;
}
// First writer of a batch (one thread, in a kernel where nobody else writes blk).
__device__ __forceinline__ void blk_open(LoopState * st, int batch)
{
    if (st->blk.batch != batch) {
        st->blk.batch = batch; st->blk.n = 0; st->blk.closed = 0; st->blk.generic = 0; st->blk.from_generic = 0;
    }
}

// x := tableau value of row i in some column after the n staged pivots; kr = blkK row of i,
// ec[s] = e_s[that column], rs[s] = r_s.
__device__ __forceinline__ double blk_replay(double x, int i, int n, const double * __restrict__ kr,
                                            const double * ec, const int * rs)
{
    for (int s = 0; s < n; s++) {
        const double p = kr[s] * ec[s];
        x = (i == rs[s]) ? ec[s] : (x + p);
    }
    return x;
}

// ---- pick(t): fast path by up to PICK_MAX_WGS workgroups of 256 threads ------------------------
__global__ __launch_bounds__(256) void k_blk_pick(LpView<F64> v, int batch, int t)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<F64>)];
    __shared__ double sh_ec[BLK_MAX], sh_eb[BLK_MAX];
    __shared__ int sh_rs[BLK_MAX];
    __shared__ unsigned long long sh_cnv;
    __shared__ int sh_rc;
    Cand<F64> * sh_c = (Cand<F64> *)sh_c_raw;
    LoopState * st = v.st;
    const int status = st->status;
    const BlkView B = blk_view(st, batch);
    const unsigned budget = st->blk.budget, done = st->done, max_iter = st->max_iter;
    const unsigned long long pkey = st->blk.price_key;         // Dantzig mode: the prep's atomicMax key decides
    const int first = pkey ? dz_col(pkey) : st->next_first;
    const int tid = threadIdx.x, p = blockIdx.x, N = gridDim.x;
    if (status != ST_RUNNING || B.closed || B.n != t || budget == 0) return;
    const int rhs = v.rhs, ld = v.ld, m = v.m, lim = v.rhs - 1, n = t;
    const bool fast = first >= 0 && first < rhs && done < max_iter;
    if (!fast) {
        // nothing staged: the generic pick (next launch) decides; else close the batch and sweep first
        if (p == 0 && tid == 0) {
            blk_open(st, batch);
            if (n == 0) st->blk.generic = 1; else st->blk.closed = 1;
        }
        return;
    }
    const double * __restrict__ tab = (const double *)v.tab;
    double * __restrict__ K = (double *)v.blkK;
    const double * __restrict__ E = (const double *)v.blkE;
    if (tid < n) {
        sh_ec[tid] = E[(size_t)tid * ld + first]; sh_eb[tid] = E[(size_t)tid * ld + rhs];
        sh_rs[tid] = st->blk.r[tid];
    }
    if (tid == 0) { sh_cnv = to_bits(v.obj[first]); sh_rc = v.rowcnt[first]; }
    __syncthreads();
    // fused pass over this workgroup's rows: replayed entering column (its negation staged as k_t),
    // replayed constant column, first pass of the ratio test
    Cand<F64> best; best.q = zero<F64>(); best.idx = INT_MAX;
    double best_a = 0.0; int best_b = 0, best_cc = 0; uint32_t best_w = 0;
    for (int i = p * 256 + tid; i < m; i += 256 * N) {
        const double x0 = tab[(size_t)i * ld + first], b0 = tab[(size_t)i * ld + rhs];
        const int bi = v.eq2bv[i];
        const uint32_t w = v.ppt[(size_t)first * v.pw + (bi >> 5)];
        const int cc = v.colcnt[bi];
        const double * kr = K + (size_t)i * BLK_MAX;
        const double a = blk_replay(x0, i, n, kr, sh_ec, sh_rs);
        const double bc = blk_replay(b0, i, n, kr, sh_eb, sh_rs);
        K[(size_t)i * BLK_MAX + n] = -a;                                  // -a_i,nv (lpsol.h:1485)
        if (le(F64(a), zero<F64>())) continue;                            // findPivotBV, lpsol.h:553-663
        if (((w >> (bi & 31)) & 1u) || cc >= lim) continue;
        Cand<F64> c; c.q = div(F64(bc), F64(a)); c.idx = i;
        const Cand<F64> nbest = better(best, c);
        if (nbest.idx != best.idx) { best_a = a; best_b = bi; best_cc = cc; best_w = w; }
        best = nbest;
    }
    const Cand<F64> wbest = block_argmin(best, sh_c);
    const bool publisher = wbest.idx != INT_MAX ? (best.idx == wbest.idx) : (tid == 0);
    if (!publisher) return;
    unsigned long long * rec = v.pickrec + (size_t)p * PICK_REC_WORDS;
    unsigned long long * ctr = v.pickrec + PICK_CTR_OFF;
    __hip_atomic_store(rec + 0, to_bits(wbest.q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 1, to_bits(F64(best_a)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 2, ((unsigned long long)(unsigned)wbest.idx << 32) | (unsigned)best_b, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 3, ((unsigned long long)best_w << 32) | (unsigned)best_cc, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the counter only ever grows and every fast pick adds exactly N to it (k_reset_loop zeroes it):
    // the add that completes a multiple of N is the last one of this pick
    const unsigned long long arrived = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((arrived + 1) % (unsigned long long)N != 0) return;
    // ---- last adder: combine in workgroup order (ties: lowest row, lpsol.h:604-611) and commit
    Cand<F64> g; g.q = zero<F64>(); g.idx = INT_MAX;
    double g_a = 0.0; int g_b = 0, g_cc = 0; uint32_t g_w = 0;
    for (int k = 0; k < N; k++) {
        const unsigned long long * rk = v.pickrec + (size_t)k * PICK_REC_WORDS;
        const unsigned long long w0 = __hip_atomic_load(rk + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w1 = __hip_atomic_load(rk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w2 = __hip_atomic_load(rk + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w3 = __hip_atomic_load(rk + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Cand<F64> c; c.q = from_bits<F64>(w0); c.idx = (int)(unsigned)(w2 >> 32);
        const Cand<F64> ng = better(g, c);
        if (ng.idx != g.idx) { g_a = from_bits<F64>(w1).v; g_b = (int)(unsigned)w2; g_w = (uint32_t)(w3 >> 32); g_cc = (int)(unsigned)w3; }
        g = ng;
    }
    blk_open(st, batch);
    if (g.idx == INT_MAX) {                            // first pass empty: second pass / disableNV are generic
        if (n == 0) st->blk.generic = 1; else st->blk.closed = 1;
        return;
    }
    const int enter = first, leave = g_b, r = g.idx;
    if (!((g_w >> (leave & 31)) & 1u)) {               // genPair, lpsol.h:100-104
        v.ppt[(size_t)enter * v.pw + (leave >> 5)] = g_w | (1u << (leave & 31));
        v.rowcnt[enter] = sh_rc + 1; v.colcnt[leave] = g_cc + 1;
    }
    st->row = r; st->col = enter; st->leave = leave;
    st->cnv_bits = sh_cnv; st->piv_bits = to_bits(F64(g_a));
    v.nv[enter] = 0; v.nv[leave] = 1; v.bv[enter] = 1; v.bv[leave] = 0;       // lpsol.h:1504-1510
    v.eq2bv[r] = enter; v.bv2eq[enter] = r; v.bv2eq[leave] = -1;
    const unsigned tp = st->total_pivots;
    if ((int)tp < v.trace_cap) { v.trace[2 * tp] = enter; v.trace[2 * tp + 1] = leave; }
    st->total_pivots = tp + 1;
    st->done = done + 1;
    st->blk.budget = budget - 1;
    st->blk.from_generic = 0;
    st->blk.r[n] = r; st->blk.n = n + 1;
    st->next_first = INT_MAX; st->anypos = 0;          // the prep's look-ahead fills these
    st->blk.price_key = 0ull;
}

// ---- the generic pick, only when pick(0) of this batch asked for it ------------------------------
__global__ __launch_bounds__(1024) void k_blk_pick_generic(LpView<F64> v, int batch)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<F64>)];
    __shared__ int sh_i[16];
    __shared__ int sh_flag;
    LoopState * st = v.st;
    if (st->status != ST_RUNNING) return;
    if (!(st->blk.batch == batch && st->blk.generic && st->blk.n == 0)) return;
    const unsigned budget = st->blk.budget;
    if (budget == 0) return;
    const PickOut o = { &st->status, &st->row, &st->col, &st->leave, &st->next_first, &st->anypos,
                        &st->cnv_bits, &st->piv_bits };
    const unsigned long long pkey = st->blk.price_key;
    const int first = pkey ? dz_col(pkey) : st->next_first, anypos = st->anypos;
    __syncthreads();
    if (threadIdx.x == 0) { st->blk.generic = 0; st->row = -1; }
    __syncthreads();
    // the tableau is fully swept (nothing staged): every column comes from it
    const bool chosen = pick_body<F64>(v, first, anypos, -1, false, false, o, v.colbuf, (Cand<F64> *)sh_c_raw, sh_i, &sh_flag);
    if (threadIdx.x == 0) {
        st->blk.budget = budget - 1;
        st->blk.price_key = 0ull;
        if (chosen) { st->blk.from_generic = 1; st->blk.r[0] = st->row; st->blk.n = 1; }
    }
}

// ---- prep(t): replayed pivot row -> e_t, objective row, look-ahead pricing ------------------------
__global__ __launch_bounds__(256) void k_blk_prep(LpView<F64> v, int batch, int t)
{
    __shared__ double sh_k[BLK_MAX];
    __shared__ int sh_rs[BLK_MAX];
    LoopState * st = v.st;
    const int status = st->status, pricing = st->pricing;
    const BlkView B = blk_view(st, batch);
    const int r = st->row, enter = st->col, leave = st->leave, from_generic = st->blk.from_generic;
    const unsigned long long piv_bits = st->piv_bits, cnv_bits = st->cnv_bits;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    if (status != ST_RUNNING || B.n != t + 1 || r < 0) return;
    const int n = t, W = v.W, rhs = v.rhs, ld = v.ld, m = v.m, lim = v.rhs - 1;
    double * __restrict__ K = (double *)v.blkK;
    double * __restrict__ E = (double *)v.blkE;
    if (threadIdx.x < n) { sh_k[threadIdx.x] = K[(size_t)r * BLK_MAX + threadIdx.x]; sh_rs[threadIdx.x] = st->blk.r[threadIdx.x]; }
    __syncthreads();
    const F64 s = div(one<F64>(), from_bits<F64>(piv_bits));  // 1/(eq.get(eqnum, nv)), lpsol.h:1471
    const int smode = scale_mode(s);
    const F64 cnv = from_bits<F64>(cnv_bits);
    const int cmode = scale_mode(cnv);
    const bool dantzig = pricing == 1;
    int nf = INT_MAX, any = 0;
    unsigned long long key = 0;
    for (int j = gid; j < W; j += gsz) {
        double x = ((const double *)v.tab)[(size_t)r * ld + j];
        F64 oj = v.obj[j];
        const bool nvj = j < rhs && v.nv[j] != 0;              // basis AFTER this pivot's swap (the pick committed it)
        const int rcj = v.rowcnt[j < rhs ? j : 0];
        for (int q = 0; q < n; q++) {                          // the pivot row as the pending sweeps would leave it
            const double e_q = E[(size_t)q * ld + j];
            const double pr = sh_k[q] * e_q;
            x = (r == sh_rs[q]) ? e_q : (x + pr);
        }
        const F64 e = scaled(F64(x), s, smode);
        E[(size_t)n * ld + j] = e.v;
        F64 tt = mul(e, minus_one<F64>());                     // nvexp.mul(-1), lpsol.h:1496
        if (j >= rhs) tt = neg(tt);                            // :1497-1499
        tt = scaled(tt, cnv, cmode);                           // nvexp.mul(tgtf(nv)), :1500
        // lpsol.h:1055-1060, left to this kernel by the fast pick: entries basic BEFORE the swap
        // (the leaving variable was, the entering one was not) below the entering index
        if (!from_generic && j < enter && (j == leave || !nvj)) oj = zero<F64>();
        const F64 o = add(tt, oj);                             // addRowToRow, :1501
        v.obj[j] = o;
        if (j < rhs && nvj && gt(o, zero<F64>())) {            // look-ahead pricing of the next pivot
            any = 1;
            if (rcj < lim) {
                if (dantzig) { const unsigned long long kj = dz_key(o.v, j); key = kj > key ? kj : key; }
                else nf = min(nf, j);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { nf = min(nf, __shfl_xor(nf, o)); any |= __shfl_xor(any, o); }
    if (dantzig) key = wave_max_u64(key);
    if ((threadIdx.x & 63) == 0) {
        if (nf != INT_MAX) atomicMin(&st->next_first, nf);
        if (key) atomicMax(&st->blk.price_key, key);
        if (any) atomicOr(&st->anypos, 1);
    }
    // -column from the generic pick's colbuf when it chose this pivot
    if (from_generic)
        for (int i = gid; i < m; i += gsz) K[(size_t)i * BLK_MAX + n] = ((const double *)v.colbuf)[i];
}

// ---- the sweep: every cell once, all staged pivots in order --------------------------------------
// NB is the batch length as a compile-time constant (the kernel switches on the wave-uniform count),
// so the update loop is straight-line code on NB register-resident row pairs e_s; -a_i,nv arrives
// through the scalar cache (blkK row of 16 doubles, wave-uniform address).
template <int ROWS, int UNROLL, int NB, bool HASR> __device__ __forceinline__
void blk_sweep_body(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
                    const double * __restrict__ K, const LoopState * __restrict__ st)
{
    const int j = blockIdx.x * 512 + threadIdx.x * 2;
    if (j >= W) return;
    int rs[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) rs[s] = st->blk.r[s];
    const int i0 = blockIdx.y * ROWS;
    const int iend = min(i0 + ROWS, m);
    if (j + 1 < W) {
        double2 e[NB];
#pragma unroll
        for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const double2 *>(E + (size_t)s * ld + j);
        double * base = tab + (size_t)i0 * ld + j;
        int i = i0;
        for (; i + UNROLL <= iend; i += UNROLL) {
            double2 a[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) a[u] = *reinterpret_cast<const double2 *>(base + (size_t)u * ld);
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const double * kr = K + (size_t)(i + u) * BLK_MAX;
#pragma unroll
                for (int s = 0; s < NB; s++) {
                    const double k = kr[s];
                    const double p0 = k * e[s].x, p1 = k * e[s].y;
                    double2 o;
                    o.x = a[u].x + p0; o.y = a[u].y + p1;
                    a[u] = (HASR && i + u == rs[s]) ? e[s] : o;
                }
                *reinterpret_cast<double2 *>(base + (size_t)u * ld) = a[u];
            }
            base += (size_t)UNROLL * ld;
        }
        for (; i < iend; i++) {
            double2 a = *reinterpret_cast<const double2 *>(base);
            const double * kr = K + (size_t)i * BLK_MAX;
#pragma unroll
            for (int s = 0; s < NB; s++) {
                const double k = kr[s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                double2 o;
                o.x = a.x + p0; o.y = a.y + p1;
                a = (HASR && i == rs[s]) ? e[s] : o;
            }
            *reinterpret_cast<double2 *>(base) = a;
            base += ld;
        }
    } else {                                                  // odd last column
        for (int i = i0; i < iend; i++) {
            double * p = tab + (size_t)i * ld + j;
            double a = *p;
#pragma unroll
            for (int s = 0; s < NB; s++) {
                const double es = E[(size_t)s * ld + j];
                const double q = K[(size_t)i * BLK_MAX + s] * es;
                a = (i == rs[s]) ? es : (a + q);
            }
            *p = a;
        }
    }
}

template <int ROWS, int UNROLL, int BCAP> __global__ __launch_bounds__(256)
void k_blk_sweep(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
                 const double * __restrict__ K, LoopState * __restrict__ st, int batch)
{
    const int status = st->status;
    const BlkView B = blk_view(st, batch);
    if (status != ST_RUNNING || B.n == 0) return;
    // only the row blocks that hold one of the pivot rows pay for the "row r := e" test per cell
    bool hasr = false;
    {
        const int lo = blockIdx.y * ROWS, hi = lo + ROWS;
        for (int s = 0; s < B.n; s++) { const int r = st->blk.r[s]; hasr = hasr || (r >= lo && r < hi); }
    }
#define XPG_BLK_CASE(NB_) case NB_: if constexpr (NB_ <= BCAP) {                                              \
        if (hasr) blk_sweep_body<ROWS, UNROLL, NB_, true>(tab, m, W, ld, E, K, st);                          \
        else blk_sweep_body<ROWS, UNROLL, NB_, false>(tab, m, W, ld, E, K, st); } break;
    switch (B.n) {
        XPG_BLK_CASE(1) XPG_BLK_CASE(2) XPG_BLK_CASE(3) XPG_BLK_CASE(4)
        XPG_BLK_CASE(5) XPG_BLK_CASE(6) XPG_BLK_CASE(7) XPG_BLK_CASE(8)
        XPG_BLK_CASE(9) XPG_BLK_CASE(10) XPG_BLK_CASE(11) XPG_BLK_CASE(12)
        XPG_BLK_CASE(13) XPG_BLK_CASE(14) XPG_BLK_CASE(15) XPG_BLK_CASE(16)
        default: break;
    }
#undef XPG_BLK_CASE
}

// Host-set budget of loop iterations (xpg_lp_iterate).
__global__ void k_blk_budget(LoopState * st, unsigned budget) { if (threadIdx.x == 0 && blockIdx.x == 0) st->blk.budget = budget; }

} // namespace xpg
