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
//   k_blk_pick(t)        runs iff exactly t pivots are staged: first pass of the ratio test
//                        (lpsol.h:553-663) on the replayed entering and constant columns by <= 64 one-wave
//                        workgroups; each stages its rows of k_t = -column and leaves ONE record (its
//                        best row) tagged with (batch, t). No atomics, no last-adder: the launch
//                        boundary orders the records for ...
//   k_blk_prep(t)        ... whose every workgroup combines the <= 64 records (lowest row wins ties,
//                        lpsol.h:604-611), replays the pivot row -> scaled row e_t, updates the
//                        objective row (with the zeroing of lpsol.h:1055-1060) and leaves its look-ahead
//                        pricing partial (lowest eligible column / Dantzig key) for the next pick; one
//                        thread commits the pivot: pair table, basis swap, trace, counters, batch length
//   k_blk_pick_generic   after pick(0) only, when that asked for it: the generic single-workgroup
//                        pick_body for whatever the fast path does not do (second ratio pass, disableNV,
//                        findPivotNVandBVPair, optimum, iteration limits) -- legal because nothing is
//                        staged, i.e. the tableau is fully swept
//   k_blk_sweep          applies the staged pivots
// Control state is only ever written by one thread of a kernel whose other workgroups do not read what
// it writes (they decide from the records / partials of the previous launch), so there is no
// intra-launch hand-off anywhere. (Two fields once broke that rule -- want_generic, cleared by the
// pick's thread 0 and read by every pick workgroup; from_generic / row, cleared by prep's committing
// thread and read by every prep workgroup -- and a late workgroup then took a different branch from
// the others; both are now cleared by the single-workgroup generic-pick launch.) A pick that cannot
// take the fast path while pivots are staged closes
// the batch: the remaining launches of the batch do nothing, the sweep applies what is staged and the
// next batch starts with pick(0) + generic on a swept tableau.
#pragma once
#include <type_traits>
#include "lp_kernels.hip.h"

namespace xpg {

// Diagnostic builds (-DXPG_STAMPS, tools/lab/run_stamps.sh): 100 MHz ticks between points of pick / prep, summed by
// workgroup 0 lane 0 into LoopState::blk.dbg (a full memory wait precedes every stamp).
#ifdef XPG_STAMPS
#define XPG_STAMP_DECL unsigned long long stamp_prev_ = wall_clock64()
#define XPG_STAMP(st_, slot_) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const unsigned long long now_ = wall_clock64(); \
        if (blockIdx.x == 0 && threadIdx.x == 0) (st_)->blk.dbg[slot_] += now_ - stamp_prev_; stamp_prev_ = now_; } while (0)
#else
#define XPG_STAMP_DECL do { } while (0)
#define XPG_STAMP(st_, slot_) do { } while (0)
#endif

#ifdef XPG_STAMPS
__device__ double g_dbg_rows[4][8192];                      // per row of the last pick: bc, bi, pair word, counter
#endif

__device__ __forceinline__ unsigned blk_epoch(int batch, int t) { return (((unsigned)batch << 5) | (unsigned)t) + 1u; }
// the ticket (and roll-call tag) of a chain launch that starts at stage 0 of `batch`: no stage's epoch (bit 31; an epoch would
// need batch >= 2^26)
__device__ __forceinline__ unsigned blk_ticket0(int batch) { return 0x80000000u | (unsigned)batch; }

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long x, int src)
{
    const unsigned lo = __shfl((unsigned)x, src), hi = __shfl((unsigned)(x >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// The look-ahead the last prep left: every wave reduces the partials itself (one load round).
struct BlkLook { int first; int anypos; };
__device__ __forceinline__ BlkLook blk_lookahead(const LpView<F64> & v, unsigned want_epoch, int nparts)
{
    const int lane = threadIdx.x & 63;
    int nf = INT_MAX, any = 0;
    unsigned long long key = 0;
    for (int k = lane; k < nparts; k += 64) {
        const int * P = v.blkP + (size_t)k * 4;             // layout: lp_kernels.hip.h, the partial store of blk_prep_body
        const int * Q = v.blkP + BLK_PART_PAY + (size_t)k * BLK_PART_PAY_INTS;
        const bool ok = (unsigned)P[2] == want_epoch;
        const int pn = P[0], pa = P[1];
        const unsigned long long pk = ((unsigned long long)(unsigned)Q[13] << 32) | (unsigned)Q[12];
        if (ok) { nf = min(nf, pn); any |= pa; key = pk > key ? pk : key; }
    }
    for (int o = 32; o > 0; o >>= 1) { nf = min(nf, __shfl_xor(nf, o)); any |= __shfl_xor(any, o); }
    key = wave_max_u64(key);
    BlkLook L;
    L.first = key ? dz_col(key) : nf;
    L.anypos = any;
    return L;
}

// ---- pick(t): first ratio-test pass by up to BLK_PICK_WGS workgroups (one wave each by default) ----
// p of N: this workgroup's share of the rows (p >= N: no rows -- a prep-only worker of the chain kernel,
// which still needs the decision). Returns whether the fast path ran; the answer is the same in every
// workgroup, because it depends on the committed state and the look-ahead partials only.
__device__ __forceinline__ bool blk_pick_body(const LpView<F64> & v, int batch, int t, int nparts, int p, int N,
                                              Cand<F64> * sh_c)
{
    LoopState * st = v.st;
    XPG_STAMP_DECL;
    const int status = st->status;
    const int bb = st->blk.batch, bn = st->blk.n, bclosed = st->blk.closed, want_generic = st->blk.want_generic;
    const int la_state = st->blk.la_from_state;
    const unsigned la_epoch = st->blk.la_epoch;
    const unsigned budget = st->blk.budget, done = st->done, max_iter = st->max_iter;
    const int sfirst = st->next_first;
    int rs[BLK_MAX];
#pragma unroll
    for (int s = 0; s < BLK_MAX; s++) rs[s] = st->blk.r[s];
    const int tid = threadIdx.x;
    // in the same round as the state: the look-ahead partials, and what this thread's first row needs
    // that does not depend on the entering column
    const BlkLook look = blk_lookahead(v, la_epoch, nparts);
    const int tpb = (int)blockDim.x;                    // 64 (one wave, no LDS round in the reduction) or 256
    const int i_pre = p * tpb + tid;
    const int i_clamped = i_pre < v.m ? i_pre : 0;
    const int bi_pre = v.eq2bv[i_clamped];
    double k_pre[BLK_MAX];
#pragma unroll
    for (int s = 0; s < BLK_MAX; s++) k_pre[s] = s < t ? ((const double *)v.blkK)[(size_t)i_clamped * BLK_MAX + s] : 0.0;
    const int n = (bb == batch) ? bn : 0;
    XPG_STAMP(st, 0);                                   // state + partials (reduced) + this thread's eq2bv / K row
    if (status != ST_RUNNING || (bb == batch && bclosed) || n != t || budget == 0) return false;
    int first = sfirst;
    if (!la_state) first = look.first;
    const int rhs = v.rhs, ld = v.ld, m = v.m, lim = v.rhs - 1;
    const bool fast = !want_generic && first >= 0 && first < rhs && done < max_iter;
    if (!fast) {
        // nothing staged: the generic pick (next launch) decides; else close the batch and sweep first
        if (p == 0 && tid == 0) {
            if (bb != batch) { st->blk.batch = batch; st->blk.n = 0; st->blk.closed = 0; st->blk.from_generic = 0; st->blk.generic = 0; }
            // (want_generic is read by every workgroup of this launch: the generic pick clears it, not this thread)
            if (n == 0) st->blk.generic = 1;
            else st->blk.closed = 1;
        }
        return false;
    }
    if (p >= N) return true;
    const double * __restrict__ tab = (const double *)v.tab;
    double * __restrict__ K = (double *)v.blkK;
    const double * __restrict__ E = (const double *)v.blkE;
    double ec[BLK_MAX], eb[BLK_MAX];                            // e_s[first], e_s[rhs]: wave-uniform loads
#pragma unroll
    for (int s = 0; s < BLK_MAX; s++) {
        ec[s] = s < n ? E[(size_t)s * ld + first] : 0.0;
        eb[s] = s < n ? E[(size_t)s * ld + rhs] : 0.0;
    }
    const unsigned long long cnv_bits = to_bits(v.obj[first]);
    // fused pass over this workgroup's rows: replayed entering column (its negation staged as k_t),
    // replayed constant column, first pass of the ratio test
    Cand<F64> best; best.q = zero<F64>(); best.idx = INT_MAX;
    double best_a = 0.0; int best_b = 0, best_cc = 0; uint32_t best_w = 0;
    bool unordered = false;                                     // a candidate whose ratio is NaN: see below
    for (int i = p * tpb + tid; i < m; i += tpb * N) {
        // eq2bv and the blkK row of this thread's first row were loaded with the state (they do not depend
        // on the entering column): everything that does goes out in ONE further round
        const bool pre = i == i_pre;
        const int bi = pre ? bi_pre : v.eq2bv[i];
        const double x0 = tab[(size_t)i * ld + first], b0 = tab[(size_t)i * ld + rhs];
        const uint32_t w = v.ppt[(size_t)first * v.pw + (bi >> 5)];
        const int cc = v.colcnt[bi];
        const double * kr = K + (size_t)i * BLK_MAX;
        double a = x0, bc = b0;
#pragma unroll
        for (int s = 0; s < BLK_MAX; s++) {
            if (s < n) {
                const double k = pre ? k_pre[s] : kr[s];
                const double pa = k * ec[s], pb = k * eb[s];
                a = (i == rs[s]) ? ec[s] : (a + pa);
                bc = (i == rs[s]) ? eb[s] : (bc + pb);
            }
        }
        K[(size_t)i * BLK_MAX + n] = -a;                                  // -a_i,nv (lpsol.h:1485)
#ifdef XPG_STAMPS
        if (i < 8192) { g_dbg_rows[0][i] = bc; g_dbg_rows[1][i] = (double)bi; g_dbg_rows[2][i] = (double)w; g_dbg_rows[3][i] = (double)cc; }
#endif
        XPG_STAMP(st, 1);                               // round 2 (column gathers, E, pair word, counter) + replay
        if (le(F64(a), zero<F64>())) continue;                            // findPivotBV, lpsol.h:553-663
        if (((w >> (bi & 31)) & 1u) || cc >= lim) continue;
        Cand<F64> c; c.q = div(F64(bc), F64(a)); c.idx = i;
        unordered = unordered || c.q.v != c.q.v;
        const Cand<F64> nbest = better(best, c);
        if (nbest.idx != best.idx) { best_a = a; best_b = bi; best_cc = cc; best_w = w; }
        best = nbest;
    }
    XPG_STAMP(st, 2);                                   // division + candidate
    const Cand<F64> wbest = block_argmin(best, sh_c);
    // A candidate ratio that is NaN (an overflow's inf - inf, mid-solve) has no place in any order: findPivotBV's answer
    // then depends on the order of its scan (lpsol.h:599-611, `minbval > v` is false either way round), which no reduction
    // reproduces. The record says so and prep hands the pivot to the generic pick, which scans in order.
    const int any_unordered = __syncthreads_or(unordered ? 1 : 0);
    XPG_STAMP(st, 3);                                   // arg-min
    const bool publisher = wbest.idx != INT_MAX ? (best.idx == wbest.idx) : (tid == 0);
    if (publisher) {
        // six {data, tag} granules of 16 bytes (the layout the chain kernel polls, lp_chain.hip.h)
        unsigned long long * rec = v.blkR + (size_t)p * BLK_REC_WORDS;
        const unsigned long long tag = (unsigned long long)blk_epoch(batch, t);
        rec[0] = to_bits(wbest.q); rec[1] = tag;
        rec[2] = to_bits(F64(best_a)); rec[3] = tag;
        rec[4] = ((unsigned long long)(unsigned)wbest.idx << 32) | (unsigned)best_b; rec[5] = tag;
        rec[6] = ((unsigned long long)best_w << 32) | (unsigned)best_cc; rec[7] = tag;
        rec[8] = cnv_bits; rec[9] = tag;
        rec[10] = (unsigned long long)(unsigned)first | (any_unordered ? 1ull << 32 : 0ull); rec[11] = tag;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_blk_pick(LpView<F64> v, int batch, int t, int nparts)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<F64>)];
    (void)blk_pick_body(v, batch, t, nparts, (int)blockIdx.x, (int)gridDim.x, (Cand<F64> *)sh_c_raw);
}

// ---- the generic pick, only when pick(0) of this batch asked for it ------------------------------
__global__ __launch_bounds__(1024) void k_blk_pick_generic(LpView<F64> v, int batch, int nparts)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<F64>)];
    __shared__ int sh_i[16];
    __shared__ int sh_flag;
    __shared__ int sh_la[2];
    LoopState * st = v.st;
    if (st->status != ST_RUNNING) return;
    // Housekeeping that must not happen inside a multi-workgroup launch whose other workgroups read the
    // fields: an earlier batch's "chosen by the generic pick" mark (prep(0) of that batch read it in
    // every workgroup, so its committing thread could not clear it).
    if (threadIdx.x == 0 && st->blk.from_generic && st->blk.batch != batch) { st->blk.from_generic = 0; st->row = -1; }
    if (!(st->blk.batch == batch && st->blk.generic && st->blk.n == 0)) return;
    const unsigned budget = st->blk.budget;
    if (budget == 0) return;
    const PickOut o = { &st->status, &st->row, &st->col, &st->leave, &st->next_first, &st->anypos,
                        &st->cnv_bits, &st->piv_bits };
    int first = st->next_first, anypos = st->anypos;
    const int la_state = st->blk.la_from_state;
    const unsigned la_epoch = st->blk.la_epoch;
    if (!la_state) {
        if (threadIdx.x < 64) {
            const BlkLook L = blk_lookahead(v, la_epoch, nparts);
            if (threadIdx.x == 0) { sh_la[0] = L.first; sh_la[1] = L.anypos; }
        }
        __syncthreads();
        first = sh_la[0]; anypos = sh_la[1];
    }
    __syncthreads();
    // the tableau is fully swept (nothing staged): every column comes from it. Whatever look-ahead
    // pick_body leaves (re-pricing after disableNV, ...) goes to next_first / anypos.
    if (threadIdx.x == 0) {
        st->blk.generic = 0; st->blk.want_generic = 0; st->row = -1; st->blk.la_from_state = 1;
        st->next_first = first; st->anypos = anypos;
    }
    __syncthreads();
    const bool chosen = pick_body<F64>(v, first, anypos, -1, false, false, o, v.colbuf, (Cand<F64> *)sh_c_raw, sh_i, &sh_flag);
    if (threadIdx.x == 0) {
        st->blk.budget = budget - 1;
        if (chosen) { st->blk.from_generic = 1; st->blk.r[0] = st->row; st->blk.n = 1; }
    }
}

// ---- prep(t): combine the pick's records, replayed pivot row -> e_t, objective row, pricing --------
// p of nwork workgroups of 256 threads. Returns whether a pivot was staged (the same answer in every
// workgroup: it depends on the committed state and the pick's records only).
__device__ __forceinline__ bool blk_prep_body(const LpView<F64> & v, int batch, int t, int p, int nwork)
{
    __shared__ int sh_nf[4], sh_any[4];
    __shared__ unsigned long long sh_key[4];
    LoopState * st = v.st;
    XPG_STAMP_DECL;
    const int status = st->status, pricing = st->pricing;
    const int bb = st->blk.batch, bn = st->blk.n, from_generic = st->blk.from_generic;
    const int srow = st->row, scol = st->col, sleave = st->leave;
    const unsigned long long spiv = st->piv_bits, scnv = st->cnv_bits;
    const unsigned budget = st->blk.budget, done = st->done, tp = st->total_pivots;
    int rs[BLK_MAX];
#pragma unroll
    for (int s = 0; s < BLK_MAX; s++) rs[s] = st->blk.r[s];
    const int gid = p * (int)blockDim.x + (int)threadIdx.x, gsz = nwork * (int)blockDim.x;
    // in the same round as the state and the records: everything of this thread's first column that does
    // not depend on the pivot row
    const int j_pre = gid < v.W ? gid : 0;
    const F64 obj_pre = v.obj[j_pre];
    const int nv_pre = v.nv[j_pre < v.rhs ? j_pre : 0], rc_pre = v.rowcnt[j_pre < v.rhs ? j_pre : 0];
    double e_pre[BLK_MAX];
#pragma unroll
    for (int q = 0; q < BLK_MAX; q++) e_pre[q] = q < t ? ((const double *)v.blkE)[(size_t)q * v.ld + j_pre] : 0.0;
    const unsigned epoch = blk_epoch(batch, t);
    int r, enter, leave, g_cc = 0; uint32_t g_w = 0;
    unsigned long long piv_bits, cnv_bits;
    bool generic_pivot = false;
    // the fast pick's records, if it ran: every wave combines them the same way -- lane l loads record l
    // (one round of loads), a 64-lane butterfly picks the best row, the winner's payload comes by shuffle
    Cand<F64> g; g.q = zero<F64>(); g.idx = INT_MAX;
    double g_a = 0.0; int g_b = 0, cand_first = -1; unsigned long long g_cnv = 0;
    bool any_rec = false, unordered_rec = false;
    {
        const int lane = threadIdx.x & 63;
        unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0;
        bool valid = false;
        if (lane < BLK_PICK_WGS) {
            const unsigned long long * rk = v.blkR + (size_t)lane * BLK_REC_WORDS;
            w0 = rk[0]; w1 = rk[2]; w2 = rk[4]; w3 = rk[6]; w4 = rk[8]; w5 = rk[10];
            valid = (unsigned)rk[11] == epoch;
        }
        const unsigned long long vmask = __ballot(valid);
        any_rec = vmask != 0;
        unordered_rec = __ballot(valid && ((w5 >> 32) & 1ull)) != 0ull;
        if (any_rec) {
            Cand<F64> c; c.q = from_bits<F64>(w0); c.idx = valid ? (int)(unsigned)(w2 >> 32) : INT_MAX;
            g = c;
            for (int o = BLK_PICK_WGS / 2; o > 0; o >>= 1) {
                Cand<F64> tq; tq.q = shfl_xor_s(g.q, o); tq.idx = __shfl_xor(g.idx, o);
                g = better(g, tq);
            }
            g.idx = __shfl(g.idx, 0); g.q = from_bits<F64>(shfl_u64(to_bits(g.q), 0));
            const int fv = __ffsll((long long)vmask) - 1;                 // any valid record: first, cnv
            cand_first = (int)(unsigned)shfl_u64(w5, fv); g_cnv = shfl_u64(w4, fv);
            if (g.idx != INT_MAX) {
                const unsigned long long wm = __ballot(valid && c.idx == g.idx);
                const int src = __ffsll((long long)wm) - 1;
                const unsigned long long p1 = shfl_u64(w1, src), p2 = shfl_u64(w2, src), p3 = shfl_u64(w3, src);
                g_a = from_bits<F64>(p1).v; g_b = (int)(unsigned)p2; g_w = (uint32_t)(p3 >> 32); g_cc = (int)(unsigned)p3;
            }
        }
    }
    XPG_STAMP(st, 4);                                   // state + records (combined) + this thread's obj / nv / rowcnt / E column
    if (status != ST_RUNNING) return false;
    if (any_rec) {
        if (g.idx == INT_MAX || unordered_rec) {               // first pass empty (second pass / disableNV), or a NaN among the candidates: generic
            if (gid == 0) {
                st->blk.closed = 1;
                if (t == 0) { st->blk.want_generic = 1; st->blk.batch = batch; st->blk.n = 0; }
            }
            return false;
        }
        r = g.idx; enter = cand_first; leave = g_b; piv_bits = to_bits(F64(g_a)); cnv_bits = g_cnv;
    } else if (t == 0 && bb == batch && from_generic && bn == 1) {
        generic_pivot = true;                                  // the generic pick chose pivot 0 of this batch
        r = srow; enter = scol; leave = sleave; piv_bits = spiv; cnv_bits = scnv;
    } else {
        return false;                                          // this batch's pick(t) did not run
    }
    const int n = t, W = v.W, rhs = v.rhs, ld = v.ld, m = v.m, lim = v.rhs - 1;
    // what the committing thread needs of the pair table goes out with the pivot row (one round fewer on
    // the thread every other workgroup ends up waiting for)
    const int rc_enter = v.rowcnt[(unsigned)enter < (unsigned)rhs ? enter : 0];
    double * __restrict__ K = (double *)v.blkK;
    double * __restrict__ E = (double *)v.blkE;
    double kq[BLK_MAX];                                        // k_q[r]: wave-uniform loads
#pragma unroll
    for (int q = 0; q < BLK_MAX; q++) kq[q] = q < n ? K[(size_t)r * BLK_MAX + q] : 0.0;
    const F64 s = div(one<F64>(), from_bits<F64>(piv_bits));  // 1/(eq.get(eqnum, nv)), lpsol.h:1471
    const int smode = scale_mode(s);
    const F64 cnv = from_bits<F64>(cnv_bits);
    const int cmode = scale_mode(cnv);
    const bool dantzig = pricing == 1;
    int nf = INT_MAX, any = 0;
    unsigned long long key = 0;
    int my_j = -1; F64 my_e = zero<F64>(), my_o = zero<F64>();  // this thread's (last) column, its new e and objective entry
    for (int j = gid; j < W; j += gsz) {
        const bool pre = j == j_pre;                           // this thread's first column: loaded with the state
        double x = ((const double *)v.tab)[(size_t)r * ld + j];
        F64 oj = pre ? obj_pre : v.obj[j];
        // the basis before and after this pivot's swap, without using the two entries the committing
        // thread rewrites (the generic pick has swapped already)
        const bool in = j < rhs;
        const bool nv_mem = in && j != enter && j != leave && (pre ? nv_pre : (int)v.nv[j]) != 0;
        const bool nv_old = in && (j == enter ? true : (j == leave ? false : nv_mem));
        const bool nv_new = in && (j == enter ? false : (j == leave ? true : nv_mem));
        const int rcj = (in && j != enter) ? (pre ? rc_pre : v.rowcnt[j]) : INT_MAX;
#pragma unroll
        for (int q = 0; q < BLK_MAX; q++) {                    // the pivot row as the pending sweeps would leave it
            if (q < n) {
                const double e_q = pre ? e_pre[q] : E[(size_t)q * ld + j];
                const double pr = kq[q] * e_q;
                x = (r == rs[q]) ? e_q : (x + pr);
            }
        }
        const F64 e = scaled(F64(x), s, smode);
        E[(size_t)n * ld + j] = e.v;
        F64 tt = mul(e, minus_one<F64>());                     // nvexp.mul(-1), lpsol.h:1496
        if (j >= rhs) tt = neg(tt);                            // :1497-1499
        tt = scaled(tt, cnv, cmode);                           // nvexp.mul(tgtf(nv)), :1500
        // lpsol.h:1055-1060, left to this kernel by the fast pick: entries basic BEFORE the swap below the
        // entering index (the generic pick has done its own zeroing)
        if (!generic_pivot && j < enter && in && !nv_old) oj = zero<F64>();
        const F64 o = add(tt, oj);                             // addRowToRow, :1501
        v.obj[j] = o;
        my_j = j; my_e = e; my_o = o;
        XPG_STAMP(st, 5);                               // round 2 (pivot row gather, K row of r) + replay + objective
        if (nv_new && gt(o, zero<F64>())) {                    // look-ahead pricing of the next pivot
            any = 1;
            if (rcj < lim) {
                if (dantzig) { const unsigned long long kj = dz_key(o.v, j); key = kj > key ? kj : key; }
                else nf = min(nf, j);
            }
        }
    }
    // this workgroup's pricing partial for the next pick
    for (int o = 32; o > 0; o >>= 1) { nf = min(nf, __shfl_xor(nf, o)); any |= __shfl_xor(any, o); }
    key = wave_max_u64(key);
    if ((threadIdx.x & 63) == 0) { sh_nf[threadIdx.x >> 6] = nf; sh_any[threadIdx.x >> 6] = any; sh_key[threadIdx.x >> 6] = key; }
    __syncthreads();
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) { nf = min(nf, sh_nf[k]); any |= sh_any[k]; key = sh_key[k] > key ? sh_key[k] : key; }
    // The partial, in the granule layout the chain kernel polls (lp_kernels.hip.h): g0 {nf, any, epoch, 0} in the dense
    // array, and in this workgroup's payload g1 {e_t[nf], epoch}, g2 {objective entry of nf, epoch}, g3 {e_t[rhs], epoch}
    // (the workgroup that owns the constant column), then the Dantzig key. g1..g3 are what the chain's first stage
    // needs fresh of this one.
    {
        int * P = v.blkP + (size_t)p * 4;
        int * Q = v.blkP + BLK_PART_PAY + (size_t)p * BLK_PART_PAY_INTS;
        if (my_j >= 0 && my_j == nf) {
            const unsigned long long eb_ = to_bits(my_e), ob_ = to_bits(my_o);
            Q[0] = (int)(unsigned)eb_; Q[1] = (int)(unsigned)(eb_ >> 32); Q[2] = (int)epoch; Q[3] = 0;
            Q[4] = (int)(unsigned)ob_; Q[5] = (int)(unsigned)(ob_ >> 32); Q[6] = (int)epoch; Q[7] = 0;
        }
        if (my_j == rhs) {
            const unsigned long long eb_ = to_bits(my_e);
            Q[8] = (int)(unsigned)eb_; Q[9] = (int)(unsigned)(eb_ >> 32); Q[10] = (int)epoch; Q[11] = 0;
        }
        if (threadIdx.x == 0) {
            Q[12] = (int)(unsigned)key; Q[13] = (int)(unsigned)(key >> 32);
            P[0] = nf; P[1] = any; P[2] = (int)epoch; P[3] = 0;
        }
    }
    // -column from the generic pick's colbuf when it chose this pivot
    if (generic_pivot)
        for (int i = gid; i < m; i += gsz) K[(size_t)i * BLK_MAX + n] = ((const double *)v.colbuf)[i];
    XPG_STAMP(st, 6);                                   // pricing partial
    // ---- one thread commits the pivot
    if (gid == 0) {
        if (!generic_pivot) {
            if (!((g_w >> (leave & 31)) & 1u)) {               // genPair, lpsol.h:100-104
                v.ppt[(size_t)enter * v.pw + (leave >> 5)] = g_w | (1u << (leave & 31));
                v.rowcnt[enter] = rc_enter + 1; v.colcnt[leave] = g_cc + 1;
            }
            v.nv[enter] = 0; v.nv[leave] = 1; v.bv[enter] = 1; v.bv[leave] = 0;       // lpsol.h:1504-1510
            v.eq2bv[r] = enter; v.bv2eq[enter] = r; v.bv2eq[leave] = -1;
            if ((int)tp < v.trace_cap) { v.trace[2 * tp] = enter; v.trace[2 * tp + 1] = leave; }
            st->total_pivots = tp + 1;
            st->done = done + 1;
            st->blk.budget = budget - 1;
            if (bb != batch) { st->blk.batch = batch; st->blk.closed = 0; st->blk.generic = 0; }
            st->blk.r[n] = r; st->blk.n = n + 1;
        }
        // ticket of the chain kernel (stages 1.. of this batch in one launch): stage 0 staged a pivot, and
        // the counters as they stand after it -- fields the chain itself never writes, so a late worker
        // of that launch reads what the early ones read
        if (t == 0) {
            for (int x = 0; x < 8; x++) st->blk.ch_arrive[x] = 0u;   // (the previous batch's chain launch has completed: stream order)
            st->blk.ch_decide = 0u;
            st->blk.ch_epoch = epoch;
            st->blk.ch_budget = generic_pivot ? budget : budget - 1;
            st->blk.ch_done = generic_pivot ? done : done + 1;
            st->blk.ch_tp = generic_pivot ? tp : tp + 1;
        }
        // (from_generic and row are read by every workgroup of this launch in the generic case: they are
        // left alone here and cleared by the next batch's generic-pick launch, a single workgroup)
        st->blk.la_from_state = 0;
        st->blk.la_epoch = epoch;
    }
    XPG_STAMP(st, 7);                                   // commit
    return true;
}

__global__ __launch_bounds__(256) void k_blk_prep(LpView<F64> v, int batch, int t)
{
    (void)blk_prep_body(v, batch, t, (int)blockIdx.x, (int)gridDim.x);
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
    {                                                         // (ld is a multiple of 16: a live first column owns a whole pair; the cells beyond W are padding)
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
    }
}

// The full batch (n == BLK_MAX, the steady state) as a kernel of its own, shaped by measurements on the
// bench tableau (tools/lab/sweep_lab.hip):
//  * compiled alone the 16-stage body fits 128 VGPRs (four wavefronts per SIMD) with next to no SGPR
//    spills; inside the switch kernel below the allocator settles on the union of all 32 bodies
//    (131 VGPRs + 102 spilled SGPRs);
//  * ping-pong row groups: the loads of the next U rows are in flight while this group's 16 x 2U
//    multiply-adds run (the arithmetic alone is ~40 us of a ~77 us pass, so it must overlap);
//  * stages outermost, the U rows in lockstep: 2U independent add chains; every cell still sees its
//    stages in order, so the bits are those of the row-by-row form;
//  * 16-row blocks: 4096 workgroups keep the tail of the launch short (32- and 64-row blocks were
//    slower although they re-read E less often); the first row group is requested before the state
//    and E are (it comes from HBM, they come from the L2);
//  * no "row r := e" selects and no row list in the loop: the stream treats the staged pivot rows like
//    any other row (what it leaves there is meaningless) and the workgroup that owns them rewrites
//    them once its stream is through -- their final contents depend on E and K only (e_s, then stages
//    s+1.. applied to it with the same two roundings per stage), and e is still in registers.
// Which tile a workgroup takes (round 3; tools/lab/sweep_lab2.hip, profiles/round3_sweep_lab.txt). Two things decide
// whether this pass runs at the rate of a plain in-place copy of the tableau or 15-30 % below it:
//  * E must stay in the L2 of the XCD that uses it. A workgroup re-reads 16 x 4 KB of E for every 64 KB of tableau
//    it streams, i.e. a third of its L2 traffic. Workgroups are dealt round-robin over the 8 XCDs, so with S column
//    strips in a plain 2-D grid an XCD meets strip s every 8 / gcd(S, 8) row blocks: S = 24 -> every row block (E
//    hits), S = 25 (the 4096 x 12289 tableau of BASELINE configs[1]) -> every 8th, by which time 4.6 MB have gone
//    through its 4 MB L2 and E comes from the fabric again: 146 us against 122 us at S = 24. Here strip s below
//    F = 8 * floor(S / 8) ALWAYS runs on XCD s % 8 (linear id -> XCD id % 8, then that XCD's own strips row block by
//    row block), and the S - F leftover strips are dealt over the XCDs by row block ((q + y) % 8), so the load stays
//    even for every S and an odd last column costs no XCD more than an eighth of a strip.
//  * Alternate passes walk the row blocks in opposite directions (rev): the tail of pass t is the head of pass
//    t + 1, so whatever the 256 MiB Infinity Cache still holds of the tableau is read back from it instead of
//    being evicted unread by a cyclic sweep (4096 x 12289: 154 -> 129 us; 8192 x 8192: 193 -> 160 us).
// One-dimensional grid of blk_sweep_grid() workgroups.
__host__ __device__ __forceinline__ int blk_sweep_grid(int strips, int rowblocks)
{
    return 8 * ((rowblocks + 7) / 8) * (8 * (strips >> 3) + (strips & 7));
}
__host__ __device__ __forceinline__ bool blk_sweep_tile(int lid, int strips, int rowblocks, int rev, int & bx, int & by)
{
    const int c = lid & 7, k = lid >> 3;
    const int nown = strips >> 3, L = strips & 7, per = 8 * nown + L;
    const int super = k / per, rem = k - super * per;
    int yy;
    if (rem < 8 * nown) { yy = rem / nown; bx = c + 8 * (rem - yy * nown); }
    else { const int q = rem - 8 * nown; yy = (c - q) & 7; bx = 8 * nown + q; }
    by = 8 * super + yy;
    if (by >= rowblocks) return false;
    if (rev) by = rowblocks - 1 - by;
    return true;
}

template <int ROWS, int U, int NB> __global__ __launch_bounds__(256)
void k_blk_sweep_full(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
                      const double * __restrict__ K, LoopState * __restrict__ st, int batch, int only_full, int rev, int full_n)
{
    static_assert(NB <= BLK_MAX, "a full batch is at most BLK_MAX pivots");
    static_assert(ROWS % (2 * U) == 0, "a row block holds whole ping-pong pairs");
    int bx, by;
    if (!blk_sweep_tile((int)blockIdx.x, (W + 511) / 512, (m + ROWS - 1) / ROWS, rev, bx, by)) return;
    const int j = bx * 512 + threadIdx.x * 2;
    const int i0 = by * ROWS;
    const int iend = min(i0 + ROWS, m);
    // the first row group goes out before anything else: it comes from HBM, the state and E from the L2.
    // ld is a multiple of 16 and the cells of a row beyond W are padding nobody reads (undefined contents), so a thread
    // whose first column is live always owns a whole 16-byte pair: an odd W has no scalar last column.
    const bool full = i0 + ROWS <= m && j < W;
    double2 a[U], b[U];
    if (full) {
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = *reinterpret_cast<const double2 *>(tab + (size_t)(i0 + u) * ld + j);
    }
    const int status = st->status;
    const int n = (st->blk.batch == batch) ? st->blk.n : 0;
    if (status != ST_RUNNING || n == 0 || j >= W) return;
    // only_full: a second launch (k_blk_sweep, the stage count as a template switch) follows for the batches that
    // closed early -- the host turns that on for LPs that close batches often (Lp::queue_blocked)
    if (only_full && n != NB) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                        // xpg_lp_counters
        if (n == full_n) st->blk.sweeps_full += 1u; else st->blk.sweeps_part += 1u;   // (full = the configured batch length, xpg_lp_counters)
    }

    if (n != NB) {
        // A partial batch (the iteration budget ran out, or a pick closed the batch early): the plain form,
        // stage count and row list read at run time, e_s re-read from the cache per row. Rare, and kept light
        // on registers so that it does not weigh on the path below.
        for (int i = i0; i < iend; i++) {
            double * p = tab + (size_t)i * ld + j;
            double ax = p[0], ay = p[1];
            for (int s = 0; s < n; s++) {
                const double k = K[(size_t)i * BLK_MAX + s];
                const double ex = E[(size_t)s * ld + j], ey = E[(size_t)s * ld + j + 1];
                const double p0 = k * ex, p1 = k * ey;
                const bool piv = st->blk.r[s] == i;
                ax = piv ? ex : ax + p0;
                ay = piv ? ey : ay + p1;
            }
            p[0] = ax; p[1] = ay;
        }
        return;
    }

    double2 e[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) e[s] = *reinterpret_cast<const double2 *>(E + (size_t)s * ld + j);
    auto load = [&](double2 (&d)[U], const double * p) {
#pragma unroll
        for (int u = 0; u < U; u++) d[u] = *reinterpret_cast<const double2 *>(p + (size_t)u * ld);
    };
    // The rows' k_s = -a_i,nv (wave-uniform: scalar loads) in groups of G stages through TWO sets of scalar registers: the
    // group g + 1 is requested before the arithmetic of group g. Left alone the compiler keeps one set -- a group is
    // requested where its registers were last used and waited for on the spot, 150 cycles per 256 of arithmetic at 32 stages
    // -- and scalar loads return out of order, so the only wait there is is "all of them": the empty asm below asks for
    // group g's registers and so puts that wait IN FRONT of the requests of group g + 1 (everything is the compiler's own
    // load and wait; nothing is in flight that it does not know of). tools/lab/sweep_lab3.hip `pipe`: 108 -> 96 us at 32
    // stages, 85.3 -> 84.4 at 24.
    constexpr int G = (NB % 16 == 0 && U <= 2) ? 8 : 4, NG = NB / G;      // (2 sets x U rows x G stages x 2 scalar registers: 64 of ~100)
    static_assert(NB % G == 0 && NG % 2 == 0, "an even number of groups: a row group starts and ends in set 0");
    double kq[2][U][G];
    auto kload = [&](int set, int row0, int g) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < G; q++) kq[set][u][q] = K[(size_t)(row0 + u) * BLK_MAX + g * G + q];
    };
    // group 0 of the rows row0 .. is in set 0 on entry; leaves group 0 of the rows next_row0 .. in set 0
    auto apply = [&](double2 (&d)[U], double * p, int row0, int next_row0) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int q = 0; q < G; q += 4)
                    asm volatile("" :: "s"(kq[g & 1][u][q]), "s"(kq[g & 1][u][q + 1]), "s"(kq[g & 1][u][q + 2]), "s"(kq[g & 1][u][q + 3]));
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) kload((g + 1) & 1, row0, g + 1); else kload((g + 1) & 1, next_row0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; q++) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const double k = kq[g & 1][u][q];
                    const double p0 = k * e[g * G + q].x, p1 = k * e[g * G + q].y;
                    d[u].x = d[u].x + p0; d[u].y = d[u].y + p1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) *reinterpret_cast<double2 *>(p + (size_t)u * ld) = d[u];
    };
    double * base = tab + (size_t)i0 * ld + j;
    if (full) {
        kload(0, i0, 0);
#pragma unroll 1
        for (int i = i0; i < i0 + ROWS; i += 2 * U) {
            load(b, base + (size_t)U * ld);
            apply(a, base, i, i + U);
            if (i + 2 * U < i0 + ROWS) load(a, base + (size_t)2 * U * ld);
            apply(b, base + (size_t)U * ld, i + U, i + 2 * U < i0 + ROWS ? i + 2 * U : i0);     // (the last one: any rows of this block -- nobody reads them)
            base += (size_t)2 * U * ld;
        }
    } else {                                                  // the short last row block
        for (int i = i0; i < iend; i++) {
            double2 a = *reinterpret_cast<const double2 *>(base);
#pragma unroll
            for (int s = 0; s < NB; s++) {
                const double k = K[(size_t)i * BLK_MAX + s];
                const double p0 = k * e[s].x, p1 = k * e[s].y;
                a.x = a.x + p0; a.y = a.y + p1;
            }
            *reinterpret_cast<double2 *>(base) = a;
            base += ld;
        }
    }

    // The staged pivot rows of this block, now that the stream is through (the row list is not read
    // before this point, so it is not live in the loop): row r_s is e_s after stage s, then every later
    // stage s2 does x := x + K[r_s][s2] * e_s2 -- unless the row pivots again at s2, in which case that
    // later stage's value is the one that survives. Same thread, same address as the streamed store
    // above, so program order puts this one last.
    asm volatile("" ::: "memory");
    int rs[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) rs[s] = st->blk.r[s];
    bool hasr = false;
#pragma unroll
    for (int s = 0; s < NB; s++) hasr = hasr || (rs[s] >= i0 && rs[s] < iend);
    if (!hasr) return;
#pragma unroll
    for (int s = 0; s < NB; s++) {
        const int r = rs[s];
        bool skip = r < i0 || r >= iend;
#pragma unroll
        for (int s2 = s + 1; s2 < NB; s2++) skip = skip || (rs[s2] == r);
        if (skip) continue;
        double2 x = e[s];
#pragma unroll
        for (int s2 = s + 1; s2 < NB; s2++) {
            const double k = K[(size_t)r * BLK_MAX + s2];
            const double p0 = k * e[s2].x, p1 = k * e[s2].y;
            x.x = x.x + p0; x.y = x.y + p1;
        }
        *reinterpret_cast<double2 *>(tab + (size_t)r * ld + j) = x;
    }
}

// The batch length as a template switch (block lengths below 16, and the A/B switch XPG_BLK_ROWS=1).
template <int ROWS, int UNROLL, int BCAP> __global__ __launch_bounds__(256)
void k_blk_sweep(double * __restrict__ tab, int m, int W, int ld, const double * __restrict__ E,
                 const double * __restrict__ K, LoopState * __restrict__ st, int batch, int skip_full, int full_n)
{
    const int status = st->status;
    const int n = (st->blk.batch == batch) ? st->blk.n : 0;
    if (status != ST_RUNNING || n == 0) return;
    if (skip_full && n == skip_full) return;            // the full-batch kernel launched just before did this one (skip_full = its length)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {     // xpg_lp_counters
        if (n == full_n) st->blk.sweeps_full += 1u; else st->blk.sweeps_part += 1u;
    }
    // only the row blocks that hold one of the pivot rows pay for the "row r := e" test per cell
    bool hasr = false;
    {
        const int lo = blockIdx.y * ROWS, hi = lo + ROWS;
        for (int s = 0; s < n; s++) { const int r = st->blk.r[s]; hasr = hasr || (r >= lo && r < hi); }
    }
#define XPG_BLK_CASE(NB_) case NB_: if constexpr (NB_ <= BCAP) {                                              \
        if (hasr) blk_sweep_body<ROWS, UNROLL, NB_, true>(tab, m, W, ld, E, K, st);                      \
        else blk_sweep_body<ROWS, UNROLL, NB_, false>(tab, m, W, ld, E, K, st); } break;
    switch (n) {
        XPG_BLK_CASE(1) XPG_BLK_CASE(2) XPG_BLK_CASE(3) XPG_BLK_CASE(4)
        XPG_BLK_CASE(5) XPG_BLK_CASE(6) XPG_BLK_CASE(7) XPG_BLK_CASE(8)
        XPG_BLK_CASE(9) XPG_BLK_CASE(10) XPG_BLK_CASE(11) XPG_BLK_CASE(12)
        XPG_BLK_CASE(13) XPG_BLK_CASE(14) XPG_BLK_CASE(15) XPG_BLK_CASE(16)
        default: break;
    }
    if (n > BCAP) {
        // a batch longer than the switch is compiled for (17 .. 31 staged pivots: a batch of 32 closed early, or the tail
        // of an iteration budget): stage count and row list read at run time, e_s re-read from the cache per row
        const int j = blockIdx.x * 512 + threadIdx.x * 2;
        if (j >= W) return;
        const int i0 = blockIdx.y * ROWS, iend = min(i0 + ROWS, m);
        for (int i = i0; i < iend; i++) {
            double * p = tab + (size_t)i * ld + j;
            double ax = p[0], ay = p[1];
            for (int s = 0; s < n; s++) {
                const double k = K[(size_t)i * BLK_MAX + s];
                const double ex = E[(size_t)s * ld + j], ey = E[(size_t)s * ld + j + 1];
                const double p0 = k * ex, p1 = k * ey;
                const bool piv = st->blk.r[s] == i;
                ax = piv ? ex : ax + p0;
                ay = piv ? ey : ay + p1;
            }
            p[0] = ax; p[1] = ay;
        }
    }
#undef XPG_BLK_CASE
}

// Host-set budget of loop iterations (xpg_lp_iterate).
__global__ void k_blk_budget(LoopState * st, unsigned budget) { if (threadIdx.x == 0 && blockIdx.x == 0) st->blk.budget = budget; }

} // namespace xpg
