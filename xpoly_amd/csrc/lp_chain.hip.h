// Blocked fp64 simplex loop, stages 1 .. B-1 of a batch in ONE launch.
//
// The launch-per-stage chain of lp_blocked.hip.h (pick(t) -> prep(t) -> pick(t+1) ...) is a latency
// chain: measured at 4096 x 8192 (tools/probe_stamps.py) a pick or prep launch lasts 5.8-6.0 us of
// which ~2 us each are its two dependent memory rounds -- first touches of a fresh launch (cold L2,
// cold TLBs) -- and a launch boundary. The same chain step as a phase of a persistent launch costs
// 2.2-2.8 us in the lab (tools/launch_lab.hip, rows D against B). This kernel is that persistent form.
//
// Workers are one-wave workgroups, one per 64 rows (pick role) and one per 64 columns (prep role);
// worker w plays both roles while it has rows and columns. NOTHING is a barrier: every hand-off is a
// data-tagged record that its consumers poll (MI355X_MICROARCH.md, "Valid forms": every byte of the
// hand-off stored sc1 and drained by s_waitcnt vmcnt(0) before the tag granule; every load of it a
// global sc1 load to registers).
//   stage t, pick role:  poll the look-ahead partials of stage t-1 (16-byte granules {nf, any, epoch})
//                        -> entering column `first`; replay that column and the constant column for
//                        the lane's own row (its k_s[i] live in registers, e_s[first] / e_s[rhs] come
//                        by one sc1 load per stage, lane s loads stage s); ratio test (lpsol.h:553-663);
//                        DPP arg-min; -a_i,nv -> K (sc1); ONE lane publishes the worker's record:
//                        six 16-byte granules {data, tag}
//   stage t, prep role:  poll the <= 256 records (lane l polls record l, all six granules in one
//                        round); combine (lowest row wins ties, lpsol.h:604-611); replay the pivot row
//                        for the lane's own column (its e_s[j] live in registers, k_s[r] by one sc1
//                        load, lane s loads stage s); scaled row -> E (sc1), objective row, look-ahead
//                        pricing (lpsol.h:1054-1069); DPP min; ONE lane publishes the partial granule.
//                        Lane 0 of worker 0 commits the pivot exactly as blk_prep_body does.
// What each worker reads of the committed state at entry comes from fields this kernel never writes
// (the ticket blk.ch_* left by stage 0's prep launch), so a worker that starts late sees what the early
// ones saw. Basis words a stage needs (pair word, counters, nv[j], rowcnt[j]) are re-read every stage with
// sc1 loads and only once the hand-off that orders them behind the previous stage's commit has been seen
// (the partials for the pick role, the records for the prep role: a prep-only worker does not wait for
// partials, so at the top of a stage the previous commit may still be in flight); the two entries the
// committing lane rewrites DURING a stage (enter, leave) are the ones that stage does not read. The basic
// variable of the lane's own row is kept in a register: every worker learns (row, entering) of each stage.
// Anything but "fast pick found a row" ends the launch for every worker alike: the pick role publishes
// a CLOSE record instead (budget spent, iteration limit, no eligible column), an empty first ratio pass
// is seen by everyone in the records. The batch then has fewer than B staged pivots, the sweep applies
// them, and the next batch starts with the launch-per-stage kernels, which own every rare branch.
// A poll that does not complete within ~0.5 s flags ST_CHAIN_STUCK (XPG_ERR_HIP for the caller).
#pragma once
#include "lp_blocked.hip.h"

namespace xpg {

enum { ST_CHAIN_STUCK = -1, CH_SPIN_LIMIT = 1 << 21, CH_CLOSE = 0x7FFFFFFF };
typedef unsigned int ch_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long ch_lo64(ch_u32x4 g) { return ((unsigned long long)g.y << 32) | g.x; }
__device__ __forceinline__ unsigned long long ch_hi64(ch_u32x4 g) { return ((unsigned long long)g.w << 32) | g.z; }

// all six granules of one record in one round (the asm block waits for its own loads)
__device__ __forceinline__ void ch_load_record(const void * p, ch_u32x4 (&g)[6])
{
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\t"
                 "global_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %6, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %6, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %5, %6, off offset:80 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]) : "v"(p) : "memory");
}
// up to four partial granules (32 bytes apart in units of one partial) in one round
__device__ __forceinline__ void ch_load_partials(const void * p0, const void * p1, const void * p2, const void * p3,
                                                 ch_u32x4 (&g)[4])
{
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
// One 16-byte granule {data, tag} by one store. The s_nop belongs to the store: a VMEM store of more than 8
// bytes reads its data registers after issue, the hardware does not interlock a VALU write to them in the next
// cycles, and the compiler's hazard recogniser cannot see a store inside inline asm (seen without it: the next
// granule's tag move landed in this granule's data).
__device__ __forceinline__ void ch_store_granule(void * p, unsigned long long lo, unsigned long long hi)
{
    ch_u32x4 g;
    g.x = (unsigned)lo; g.y = (unsigned)(lo >> 32); g.z = (unsigned)hi; g.w = (unsigned)(hi >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" :: "v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ void ch_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <class T> __device__ __forceinline__ T ch_ld(const T * p)      // agent-scope (sc1) load of 1, 4 or 8 bytes
{
    if constexpr (sizeof(T) == 1) {
        return (T)__hip_atomic_load((const unsigned char *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (sizeof(T) == 4) {
        const unsigned u = __hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        T r; __builtin_memcpy(&r, &u, 4); return r;
    } else {
        static_assert(sizeof(T) == 8, "1-, 4- or 8-byte objects");
        const unsigned long long u = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        T r; __builtin_memcpy(&r, &u, 8); return r;
    }
}
template <class T> __device__ __forceinline__ void ch_st(T * p, T x)    // agent-scope (sc1, write-through) store
{
    if constexpr (sizeof(T) == 1) {
        unsigned char u; __builtin_memcpy(&u, &x, 1);
        __hip_atomic_store((unsigned char *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (sizeof(T) == 4) {
        unsigned u; __builtin_memcpy(&u, &x, 4);
        __hip_atomic_store((unsigned *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        static_assert(sizeof(T) == 8, "1-, 4- or 8-byte objects");
        unsigned long long u; __builtin_memcpy(&u, &x, 8);
        __hip_atomic_store((unsigned long long *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ double ch_readlane_f64(double x, int lane)
{
    int w[2];
    __builtin_memcpy(w, &x, 8);
    w[0] = __builtin_amdgcn_readlane(w[0], lane); w[1] = __builtin_amdgcn_readlane(w[1], lane);
    __builtin_memcpy(&x, w, 8);
    return x;
}
__device__ __forceinline__ unsigned long long ch_readlane_u64(unsigned long long x, int lane)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// npick = ceil(m / 64) <= 256 pick workers, nprep = ceil(W / 64) <= 256 prep workers; grid = max of the two.
__global__ __launch_bounds__(64) void k_blk_chain(LpView<F64> v, int batch, int t0, int B, int npick, int nprep)
{
    LoopState * st = v.st;
    const int w = (int)blockIdx.x, lane = (int)threadIdx.x;
    const int m = v.m, W = v.W, rhs = v.rhs, ld = v.ld, lim = v.rhs - 1;
    // ---- the ticket: stage 0 of THIS batch staged a pivot (fields this launch never writes)
    if (st->blk.ch_epoch != blk_epoch(batch, t0 - 1) || st->status != ST_RUNNING || st->pricing != 0) return;
    unsigned budget = st->blk.ch_budget, done = st->blk.ch_done, tp = st->blk.ch_tp;
    const unsigned max_iter = st->max_iter;
    const bool picker = w < npick, prepper = w < nprep;
    const int i = w * 64 + lane, j = w * 64 + lane;         // this lane's row (pick role) and column (prep role)
    const bool has_row = picker && i < m, has_col = prepper && j < W;
    const int ic = has_row ? i : 0, jc = has_col ? j : 0;
    const double * __restrict__ tab = (const double *)v.tab;
    double * K = (double *)v.blkK;
    double * E = (double *)v.blkE;
    // own data of earlier stages (written by previous launches): into registers once
    double kreg[BLK_MAX], ereg[BLK_MAX];
#pragma unroll
    for (int s = 0; s < BLK_MAX; s++) {
        kreg[s] = (s < t0 && has_row) ? K[(size_t)ic * BLK_MAX + s] : 0.0;
        ereg[s] = (s < t0 && has_col) ? E[(size_t)s * ld + jc] : 0.0;
    }
    int rs[BLK_MAX];                                        // pivot rows staged so far (wave-uniform)
#pragma unroll
    for (int s = 0; s < BLK_MAX; s++) rs[s] = s < t0 ? st->blk.r[s] : -1;
    F64 oj = has_col ? v.obj[jc] : zero<F64>();            // this lane's objective entry: own data throughout
    int bi = v.eq2bv[ic];                                   // basic variable of this lane's row (stage 0's commit is in)
    bool stuck = false;

#pragma unroll 1
    for (int t = t0; t < B; t++) {
        const unsigned want_part = blk_epoch(batch, t - 1), tag = blk_epoch(batch, t);
        int first = -1;
        // =========================== pick role =====================================================
        if (picker) {
            // ---- poll the partials of stage t-1: <= 4 per lane
            int nf = INT_MAX, any = 0;
            {
                const char * base = (const char *)v.blkP;
                const int k0 = lane, k1 = lane + 64, k2 = lane + 128, k3 = lane + 192;
                const void * p0 = base + (size_t)(k0 < nprep ? k0 : 0) * (BLK_PART_INTS * 4);
                const void * p1 = base + (size_t)(k1 < nprep ? k1 : 0) * (BLK_PART_INTS * 4);
                const void * p2 = base + (size_t)(k2 < nprep ? k2 : 0) * (BLK_PART_INTS * 4);
                const void * p3 = base + (size_t)(k3 < nprep ? k3 : 0) * (BLK_PART_INTS * 4);
                ch_u32x4 g[4];
                unsigned spins = 0;
                for (;;) {
                    ch_load_partials(p0, p1, p2, p3, g);
                    const bool ok = g[0].z == want_part && g[1].z == want_part && g[2].z == want_part && g[3].z == want_part;
                    if (__all(ok)) break;
                    if (++spins > CH_SPIN_LIMIT) { stuck = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (stuck) break;
                if (k0 < nprep) { nf = min(nf, (int)g[0].x); any |= (int)g[0].y; }
                if (k1 < nprep) { nf = min(nf, (int)g[1].x); any |= (int)g[1].y; }
                if (k2 < nprep) { nf = min(nf, (int)g[2].x); any |= (int)g[2].y; }
                if (k3 < nprep) { nf = min(nf, (int)g[3].x); any |= (int)g[3].y; }
            }
            first = wave_min_int(nf);
            (void)any;
            // the fast path of blk_pick_body, or the end of the batch for everyone
            const bool fast = first >= 0 && first < rhs && done < max_iter && budget != 0;
            unsigned long long * rec = v.blkR + (size_t)w * BLK_REC_WORDS;
            if (!fast) {
                if (lane == 0) {
                    // (budget spent: the launch-per-stage pick leaves the batch open; otherwise it closes it)
                    if (w == 0 && budget != 0) ch_st(&st->blk.closed, 1);
                    ch_drain();
                    for (int q = 0; q < 6; q++) ch_store_granule(rec + 2 * q, q == 5 ? (unsigned long long)CH_CLOSE : 0ull, (unsigned long long)tag);
                }
                first = CH_CLOSE;
            } else {
                // ---- one round: the column gathers, e_s[first] / e_s[rhs] (lane s / lane 16+s), c_nv, pair word, counter
                const double x0 = tab[(size_t)ic * ld + first], b0 = tab[(size_t)ic * ld + rhs];
                double ev = 0.0;
                if (lane < t) ev = ch_ld(&E[(size_t)lane * ld + first]);
                else if (lane >= 16 && lane - 16 < t) ev = ch_ld(&E[(size_t)(lane - 16) * ld + rhs]);
                const unsigned long long cnv_bits = to_bits(ch_ld(&v.obj[first]));
                const uint32_t pw_word = ch_ld(&v.ppt[(size_t)first * v.pw + (bi >> 5)]);
                const int cc = ch_ld(&v.colcnt[bi]);
                double a = x0, bc = b0;
#pragma unroll
                for (int s = 0; s < BLK_MAX; s++) {
                    if (s < t) {
                        const double ec = ch_readlane_f64(ev, s), eb = ch_readlane_f64(ev, 16 + s);
                        const double k = kreg[s];
                        const double pa = k * ec, pb = k * eb;
                        a = (i == rs[s]) ? ec : (a + pa);
                        bc = (i == rs[s]) ? eb : (bc + pb);
                    }
                }
#pragma unroll
                for (int s = 0; s < BLK_MAX; s++) if (s == t) kreg[s] = -a;      // -a_i,nv (lpsol.h:1485)
                if (has_row) ch_st(&K[(size_t)i * BLK_MAX + t], -a);
#ifdef XPG_STAMPS
                if (has_row && i < 8192) { g_dbg_rows[0][i] = bc; g_dbg_rows[1][i] = (double)bi; g_dbg_rows[2][i] = (double)pw_word; g_dbg_rows[3][i] = (double)cc; }
#endif
                // findPivotBV's first pass (lpsol.h:553-663)
                Cand<F64> c; c.q = zero<F64>(); c.idx = INT_MAX;
                if (has_row && !le(F64(a), zero<F64>()) && !((pw_word >> (bi & 31)) & 1u) && cc < lim) {
                    c.q = div(F64(bc), F64(a)); c.idx = i;
                }
                const Cand<F64> best = wave_argmin(c);
                const int bidx = __builtin_amdgcn_readfirstlane(best.idx);
                const bool publisher = bidx != INT_MAX ? (i == bidx) : (lane == 0);
                ch_drain();                                                       // the wave's K stores are out
                if (publisher) {
                    const unsigned long long tg = (unsigned long long)tag;
                    ch_store_granule(rec + 0, to_bits(best.q), tg);
                    ch_store_granule(rec + 2, to_bits(F64(a)), tg);
                    ch_store_granule(rec + 4, ((unsigned long long)(unsigned)bidx << 32) | (unsigned)bi, tg);
                    ch_store_granule(rec + 6, ((unsigned long long)pw_word << 32) | (unsigned)cc, tg);
                    ch_store_granule(rec + 8, cnv_bits, tg);
                    ch_store_granule(rec + 10, (unsigned long long)(unsigned)first, tg);
                }
            }
        }
        // =========================== prep role (every worker reads the records: all must leave together) ====
        Cand<F64> g; g.q = zero<F64>(); g.idx = INT_MAX;
        double g_a = 0.0; int g_b = 0, g_cc = 0, enter = -1; uint32_t g_w = 0; unsigned long long g_cnv = 0;
        {
            Cand<F64> mine; mine.q = zero<F64>(); mine.idx = INT_MAX;
            unsigned long long m_a = 0, m_ib = 0, m_wc = 0, m_cnv = 0, m_first = 0;
            unsigned spins = 0;
            for (int u = 0; u < 4; u++) {                   // <= 256 records, lane l polls l, l + 64, ...
                const int k = lane + 64 * u;
                if (64 * u >= npick) break;                 // wave-uniform
                const void * p = v.blkR + (size_t)(k < npick ? k : 0) * BLK_REC_WORDS;
                ch_u32x4 gr[6];
                for (;;) {
                    ch_load_record(p, gr);
                    const bool ok = gr[0].z == tag && gr[1].z == tag && gr[2].z == tag && gr[3].z == tag && gr[4].z == tag && gr[5].z == tag;
                    if (__all(ok)) break;
                    if (++spins > CH_SPIN_LIMIT) { stuck = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (stuck) break;
                if (k < npick) {
                    const unsigned long long fw = ch_lo64(gr[5]);
                    Cand<F64> c; c.q = from_bits<F64>(ch_lo64(gr[0])); c.idx = (int)(unsigned)(ch_lo64(gr[2]) >> 32);
                    if ((unsigned)fw == (unsigned)CH_CLOSE) { c.idx = INT_MAX; m_first = fw; }
                    else if (m_first != (unsigned long long)CH_CLOSE) m_first = fw;
                    const Cand<F64> nb = better(mine, c);
                    if (nb.idx != mine.idx) { m_a = ch_lo64(gr[1]); m_ib = ch_lo64(gr[2]); m_wc = ch_lo64(gr[3]); }
                    m_cnv = ch_lo64(gr[4]);
                    mine = nb;
                }
            }
            if (stuck) break;
            // every pick worker takes the same fast / close decision, so any record tells which it was
            const unsigned f0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m_first);
            if (f0 == (unsigned)CH_CLOSE) break;
            enter = (int)f0;
            g = wave_argmin(mine);
            const int gi = __builtin_amdgcn_readfirstlane(g.idx);
            if (gi == INT_MAX) {
                // first ratio pass empty: the second pass / disableNV are the generic pick's (next batch)
                if (w == 0 && lane == 0) ch_st(&st->blk.closed, 1);
                break;
            }
            const unsigned long long hit = __ballot(mine.idx == gi);
            const int src = __ffsll((long long)hit) - 1;
            g.idx = gi; g.q = from_bits<F64>(ch_readlane_u64(to_bits(g.q), 0));
            g_a = from_bits<F64>(ch_readlane_u64(m_a, src)).v;
            const unsigned long long ib = ch_readlane_u64(m_ib, src), wc = ch_readlane_u64(m_wc, src);
            g_b = (int)(unsigned)ib; g_w = (uint32_t)(wc >> 32); g_cc = (int)(unsigned)wc;
            g_cnv = ch_readlane_u64(m_cnv, 0);
        }
        const int r = g.idx, leave = g_b, n = t;
#pragma unroll
        for (int s = 0; s < BLK_MAX; s++) if (s == t) rs[s] = r;
        if (i == r) bi = enter;                             // lpsol.h:1508, as every later pick of this launch sees it
        if (prepper) {
            // ---- one round: the pivot row gather, k_q[r] (lane q), this column's basis words (the previous
            // stage's commit is behind the records just read), the committing lane's counter
            const double x0 = tab[(size_t)r * ld + jc];
            const int nvj = has_col && j < rhs ? (int)ch_ld(&v.nv[jc]) : 0;
            const int rcj0 = has_col && j < rhs ? ch_ld(&v.rowcnt[jc]) : INT_MAX;
            double kv = 0.0;
            if (lane < t) kv = ch_ld(&K[(size_t)r * BLK_MAX + lane]);
            int rc_enter = 0;
            if (w == 0 && lane == 0) rc_enter = ch_ld(&v.rowcnt[enter]);
            const F64 sc = div(one<F64>(), F64(g_a));       // 1/(eq.get(eqnum, nv)), lpsol.h:1471
            const int smode = scale_mode(sc);
            const F64 cnv = from_bits<F64>(g_cnv);
            const int cmode = scale_mode(cnv);
            double x = x0;
#pragma unroll
            for (int q = 0; q < BLK_MAX; q++) {             // the pivot row as the pending sweeps would leave it
                if (q < t) {
                    const double kq = ch_readlane_f64(kv, q);
                    const double pr = kq * ereg[q];
                    x = (r == rs[q]) ? ereg[q] : (x + pr);
                }
            }
            const F64 e = scaled(F64(x), sc, smode);
#pragma unroll
            for (int q = 0; q < BLK_MAX; q++) if (q == t) ereg[q] = e.v;
            int nf = INT_MAX, any = 0;
            if (has_col) {
                ch_st(&E[(size_t)n * ld + j], e.v);
                F64 tt = mul(e, minus_one<F64>());          // nvexp.mul(-1), lpsol.h:1496
                if (j >= rhs) tt = neg(tt);                 // :1497-1499
                tt = scaled(tt, cnv, cmode);                // nvexp.mul(tgtf(nv)), :1500
                const bool in = j < rhs;
                const bool nv_mem = in && j != enter && j != leave && nvj != 0;
                const bool nv_old = in && (j == enter ? true : (j == leave ? false : nv_mem));
                const bool nv_new = in && (j == enter ? false : (j == leave ? true : nv_mem));
                const int rcj = (in && j != enter) ? rcj0 : INT_MAX;
                if (j < enter && in && !nv_old) oj = zero<F64>();     // lpsol.h:1055-1060
                oj = add(tt, oj);                           // addRowToRow, :1501
                ch_st(&v.obj[j], oj);
                if (nv_new && gt(oj, zero<F64>())) { any = 1; if (rcj < lim) nf = j; }
            }
            nf = wave_min_int(nf);
            any = __ballot(any != 0) != 0ull ? 1 : 0;
            // ---- worker 0, lane 0 commits the pivot (as blk_prep_body)
            if (w == 0 && lane == 0) {
                if (!((g_w >> (leave & 31)) & 1u)) {        // genPair, lpsol.h:100-104
                    ch_st(&v.ppt[(size_t)enter * v.pw + (leave >> 5)], g_w | (1u << (leave & 31)));
                    ch_st(&v.rowcnt[enter], rc_enter + 1); ch_st(&v.colcnt[leave], g_cc + 1);
                }
                ch_st(&v.nv[enter], (uint8_t)0); ch_st(&v.nv[leave], (uint8_t)1);      // lpsol.h:1504-1510
                ch_st(&v.bv[enter], (uint8_t)1); ch_st(&v.bv[leave], (uint8_t)0);
                ch_st(&v.eq2bv[r], enter); ch_st(&v.bv2eq[enter], r); ch_st(&v.bv2eq[leave], -1);
                if ((int)tp < v.trace_cap) { v.trace[2 * tp] = enter; v.trace[2 * tp + 1] = leave; }
                st->total_pivots = tp + 1;
                st->done = done + 1;
                st->blk.budget = budget - 1;
                st->blk.r[n] = r; st->blk.n = n + 1;
                st->blk.la_from_state = 0;
                st->blk.la_epoch = tag;
            }
            ch_drain();                                     // this wave's E / obj (and the commit) are out
            if (lane == 0) {
                int * P = v.blkP + (size_t)w * BLK_PART_INTS;
                ch_store_granule(P, ((unsigned long long)(unsigned)any << 32) | (unsigned)nf, (unsigned long long)tag);
            }
        }
        tp += 1; done += 1; budget -= 1;
    }
    if (stuck && lane == 0) ch_st(&st->status, (int)ST_CHAIN_STUCK);
}

} // namespace xpg
