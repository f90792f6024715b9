// Fused loop for the rational scalar (config 4): ONE launch per pivot. k_pipe_fused_r32 sweeps pivot t, chooses pivot
// t + 1 (one-wave pick workgroups, fused_pick_r32 below) and STAGES it -- scaled pivot row, objective row, look-ahead pricing
// of t + 2 -- inside the same launch, which is what k_pipe_prep did in a launch of its own (6-7 us per pivot beside a
// 17.6 us sweep at 1024 x 2048: launch latency and a drained chip, the staging itself is 2048 cells).
//
// What that needs:
//   * The staging needs row r' of the tableau AFTER the running sweep, and r' is only known once the pick is done.
//     The sweep therefore does not work in place: it reads one copy of the tableau and writes the other (LpView::tab /
//     tab2, PipeDesc::side names the current one), so the stagers compute a'_r',j = a_r',j + k_r' * e_j themselves from
//     the immutable input side with the sweep's own operation -- no ordering against the workgroup that sweeps that row.
//     A column whose e_j is zero is copied instead of skipped.
//   * A hand-over inside the launch: every pick workgroup publishes its best row as four self-validating 16-byte
//     granules (write-through stores); ceil(W / 256) stager workgroups -- dispatched right behind the pick workgroups,
//     polling with s_sleep meanwhile -- each combine the N records themselves (no counter, no last adder, no flag: one
//     store-to-load hop) and take it from there. Their look-ahead goes through agent-scope atomics on accumulators that
//     only atomics touch; the LAST stager to finish writes the whole next descriptor (one writer, plain stores: the
//     launch boundary publishes it); stager 0 also does the pick's bookkeeping (genPair, or the deferral).
//   * Nothing a reader of the handle sees may run ahead of the sweep: the staged scaled row and objective row live in
//     staging buffers per slot (LpView::stage); the stagers of the launch that SWEEPS a pivot commit its objective row
//     to v.obj first. The basis swap was already committed that way (pick workgroup 0).
//   * The generic pick (relaxed second pass, disableNV, findPivotNVandBVPair: 131 registers and scratch) must not be
//     compiled into this launch. The host enqueues a generic point -- k_fused_generic: generic pick in place and the
//     staging of what it chose -- before the first launch of every queue_iterations call and every
//     XPG_R32_GENERIC_EVERY (16) launches; a deferred decision idles through the fused launches until then (workgroup
//     (0,0) carries the descriptor over to the other slot and counts it: a solve that idles often gets a generic point
//     before every launch from the host's next status read on).
//   * Everything outside this loop reads v.tab: the host's status read after each batch of launches (Lp::read_state)
//     sees which side the last launch left current and enqueues k_side_home when it is side 1 (both copies are then
//     current).
// Lp::queue_iterations takes this loop for tableaux of 350 k - 5 M cells (below, both loops are a chain of latencies; above,
// the copies of the columns an in-place sweep skips are HBM traffic the two-launch loop of lp_pipe_r32.hip.h does not have).
#pragma once
#include "lp_pipe_r32.hip.h"

namespace xpg {

__device__ __forceinline__ unsigned long long * go_block(const LpView<R32> & v, int slot)
{ return v.pickrec + PICK_GO_OFF + (size_t)PICK_GO_WORDS * slot; }
__device__ __forceinline__ void go_store(unsigned long long * p, unsigned long long x)
{ __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long go_load(const unsigned long long * p)
{ return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A pick workgroup's ratio-test record: four 16-byte granules {8 bytes of data, ticket, kind}, each stored with ONE
// write-through store and valid by itself (ticket = the handle's pivot count once the swept pivot is committed: unique
// per launch, never 0) -- no counter, no flag, no ordering between them: the stagers poll the granules of all N pick
// workgroups and every stager workgroup combines them itself.
typedef unsigned int fr_u32x4 __attribute__((ext_vector_type(4)));
enum { FREC_CAND = 1, FREC_NONE = 2 };
__device__ __forceinline__ char * frec(const LpView<R32> & v, int p) { return (char *)(v.pickrec + PICK_FREC_OFF + (size_t)PICK_FREC_WORDS * p); }
__device__ __forceinline__ void fr_store(char * p, unsigned long long data, unsigned ticket, int kind)
{
    fr_u32x4 g; g.x = (unsigned)data; g.y = (unsigned)(data >> 32); g.z = ticket; g.w = (unsigned)kind;
    // (s_nop: the hardware reads the data registers of a > 8-byte VMEM store after issue, and the compiler's hazard
    // recogniser does not look into asm -- without it the next granule's moves can land in this one: ch_store_u32x4,
    // lp_chain.hip.h, found that the hard way)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" :: "v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ fr_u32x4 fr_load(const char * p)
{
    fr_u32x4 g;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
    return g;
}
// diagnostic builds (-DXPG_STAMPS): 100 MHz wall-clock stamps of the LAST launch's chain in LoopState::blk.dbg
// (0 pick start, 1 pick workgroup 0 through its rows, 2 its record stored, 3 stager 0 start, 4 stager 0 has every
//  record, 5 stager 0 through its column, 6 descriptor written, 7 pick workgroup 0's record stored (2: the LAST record);
//  tools/lab/probe_rat_chain.py)
#ifdef XPG_STAMPS
#define FUSED_STAMP(st_, k_) do { (st_)->blk.dbg[k_] = wall_clock64(); } while (0)
#else
#define FUSED_STAMP(st_, k_) do { } while (0)
#endif

// The pick workgroups (p = 0 .. N-1) of a fused launch: ONE WAVE each (the other three of the workgroup leave at once) --
// up to 16 waves on 16 compute units, a row of the ratio test per lane at 1024 rows, the wave's best by shuffles: no LDS
// round, no barrier, and the N <= 16 records are exactly the 64 granules one stager wave polls.
// A: the tableau side the launch reads, B: the one it writes.
__device__ inline void fused_pick_r32(const LpView<R32> & v, int slot, int colstride, int p, int N,
                                      const R32 * __restrict__ A, R32 * __restrict__ B)
{
    LoopState * st = v.st;
    PipeDesc & I = st->pd[slot];
    PipeDesc & O = st->pd[slot ^ 1];
    const int r = I.row, ienter = I.col, ileave = I.leave, first = desc_first(I), anypos = I.anypos, stop = I.stop, side = I.side;
    const unsigned done_now = I.done_after, total_now = I.total_after, max_iter = st->max_iter;
    const bool cn = st->noncanon == 0;
    const int W = v.W, rhs = v.rhs, ld = v.ld, m = v.m, lim = v.rhs - 1;
    R32 * __restrict__ nextcol = v.nextcol;
    R32 * __restrict__ bcol = v.bcol;
    R32 * __restrict__ cbo = v.colbuf + (size_t)(slot ^ 1) * colstride;
    const int tid = threadIdx.x;
    if (tid >= 64) return;

    if (stop != 0) {
        // wave 0 of workgroup 0 alone promotes the deferred final status (as k_pipe_prep does)
        if (p != 0) return;
        const int zu = I.zero_upto;
        for (int j = tid; j < zu; j += 64)
            if (!v.nv[j]) v.obj[j] = zero<R32>();              // lpsol.h:1055-1060, deferred by the pick
        if (tid == 0) { O = I; st->status = stop; }
        return;
    }
    if (r < 0) {                                       // a deferred decision waits for the next generic point: carry it over
        if (p == 0 && tid == 0) { O = I; st->r32_idle += 1u; }
        return;
    }

    // (a pivot nobody has staged is not a state the host's launch sequence can produce -- a generic point stages what
    // it picks, a fused launch what its pick chooses; if it ever is seen, the solve ends with an error and NOTHING of this
    // launch is written: the sweep workgroups and the stagers test the same flag, so the handle's tableau, basis and
    // descriptors stay those of the last good pivot)
    if (!I.staged) { if (p == 0 && tid == 0) st->status = XPG_ERR_CHAIN_STUCK; return; }
    // ---- a sweep is running around us
    if (p == 0 && tid == 0) {                          // commit this iteration's pivot (lpsol.h:1504-1510)
        FUSED_STAMP(st, 0);
        v.nv[ienter] = 0; v.nv[ileave] = 1; v.bv[ienter] = 1; v.bv[ileave] = 0;
        v.eq2bv[r] = ienter; v.bv2eq[ienter] = r; v.bv2eq[ileave] = -1;
        XPG_TRACE_PIVOT("hbm-fused", ienter, ileave, r);
        const unsigned t = total_now - 1;
        if ((int)t < v.trace_cap) { v.trace[2 * t] = ienter; v.trace[2 * t + 1] = ileave; }
        st->total_pivots = total_now; st->done = done_now;
    }
    const int xc = (first >= 0 && first < W) ? first : -1;
    const R32 * __restrict__ cb = v.colbuf + (size_t)slot * colstride;
    const R32 * __restrict__ rb = v.stage + (size_t)slot * ld;
    const R32 eb = rb[rhs];
    const int stride = 64 * N;                         // rows are dealt to the waves in blocks of 64
    if (xc < 0 || done_now >= max_iter) {
        // no ratio test this time: keep the columns current, workgroup 0 records the outcome; the stagers are told
        if (xc >= 0) {
            const R32 e0 = rb[xc];
            for (int i = p * 64 + tid; i < m; i += stride) {
                const R32 n0 = (i == r) ? e0 : l_fma(cn, A[(size_t)i * ld + xc], cb[i], e0);
                B[(size_t)i * ld + xc] = n0;
                nextcol[i] = n0;
            }
        }
        for (int i = p * 64 + tid; i < m; i += stride)
            bcol[i] = (i == r) ? eb : l_fma(cn, bcol[i], cb[i], eb);
        if (tid < 4) fr_store(frec(v, p) + 16 * tid, 0ull, total_now, FREC_NONE);
        if (p == 0 && tid == 0) {
            if (done_now >= max_iter)                  // while (cnt < m_max_iter), lpsol.h:1039
                write_desc(O, -1, 0, 0, first, anypos, 4, xc, 0, done_now, total_now, 0ull, 0ull);
            else if (first == INT_MAX && !anypos)      // optimum reached: lpsol.h:1089
                write_desc(O, -1, 0, 0, first, anypos, ST_CHECK_OPT, -1, rhs, done_now, total_now, 0ull, 0ull);
            else                                       // findPivotNVandBVPair needs the whole tableau: next generic point
                write_desc(O, -1, 0, 0, first, anypos, 0, -1, 0, done_now, total_now, 0ull, 0ull);
            O.side = side ^ 1; O.staged = 0; st->r32_side = side ^ 1;
        }
        return;
    }

    // ---- fused pass over this workgroup's rows: new constant column, new entering column, -column, first ratio pass
    const R32 e0 = rb[xc];
    Cand<R32> best; best.q = zero<R32>(); best.idx = INT_MAX;
    R32 best_a = zero<R32>(); int best_b = 0, best_cc = 0; uint32_t best_w = 0;
    bool weird = false;                                // a quotient with den <= 0: no order to reduce by (lp_kernels.hip.h)
    for (int i = p * 64 + tid; i < m; i += stride) {
        const R32 k = cb[i], bo = bcol[i], c0 = A[(size_t)i * ld + xc];        // every load of the row in flight before the first use
        int bi = v.eq2bv[i];
        if (i == r) bi = ienter;                       // the commit above, seen without waiting for it
        const uint32_t w = v.ppt[(size_t)xc * v.pw + (bi >> 5)];
        const int cc = v.colcnt[bi];
        R32 nb = l_fma(cn, bo, k, eb), a = l_fma(cn, c0, k, e0);               // the sweep's a + k*e
        if (i == r) { nb = eb; a = e0; }
        bcol[i] = nb;
        B[(size_t)i * ld + xc] = a;
        nextcol[i] = a;
        cbo[i] = neg(a);                                                       // -column, lpsol.h:1485
        if (le(a, zero<R32>())) continue;                                      // findPivotBV, lpsol.h:553-663
        if (((w >> (bi & 31)) & 1u) || cc >= lim) continue;
        Cand<R32> c; c.q = l_div(cn, nb, a); c.idx = i;
        weird |= unordered_value(c.q);
        const Cand<R32> nbest = better(best, c);
        if (nbest.idx != best.idx) { best_a = a; best_b = bi; best_cc = cc; best_w = w; }
        best = nbest;
    }
    if (p == 0 && tid == 0) FUSED_STAMP(st, 1);
    const Cand<R32> wbest = wave_argmin(best);                                 // four DPP steps, no LDS
    const bool wg_weird = __ballot(weird) != 0ull;
    // one lane publishes this workgroup's record: the owner of the winning row, else lane 0
    const bool publisher = wbest.idx != INT_MAX ? (best.idx == wbest.idx) : (tid == 0);
    if (!publisher) return;
    char * rec = frec(v, p);
    fr_store(rec, to_bits(wbest.q), total_now, FREC_CAND);
    fr_store(rec + 16, to_bits(best_a), total_now, FREC_CAND);
    fr_store(rec + 32, ((unsigned long long)(unsigned)wbest.idx << 32) | (unsigned)best_b, total_now, FREC_CAND);
    fr_store(rec + 48, ((unsigned long long)best_w << 32) | (unsigned)best_cc | (wg_weird ? 0x80000000u : 0u), total_now, FREC_CAND);
#ifdef XPG_STAMPS
    atomicMax(&st->blk.dbg[2], (unsigned long long)wall_clock64());        // the LAST record
    if (p == 0) FUSED_STAMP(st, 7);
#endif
}

// The stager workgroups (q = 0 .. NP-1, one column per thread) of a fused launch.
__device__ inline void fused_stage_r32(const LpView<R32> & v, int slot, int colstride, int q, int N, int NP, const R32 * __restrict__ A)
{
    __shared__ unsigned long long sh_res[4];           // kind | row << 32, leave, pivot bits, pair-table word << 32 | column count
    LoopState * st = v.st;
    PipeDesc & I = st->pd[slot];
    PipeDesc & O = st->pd[slot ^ 1];
    const int r = I.row, ienter = I.col, ileave = I.leave, first = desc_first(I), anypos = I.anypos, stop = I.stop, side = I.side;
    const unsigned done_now = I.done_after, total_now = I.total_after;
    const bool canon = st->noncanon == 0;
    if (r < 0 || stop != 0 || !I.staged) return;       // (unstaged: pick workgroup 0 reports it, nobody writes)
    const int W = v.W, rhs = v.rhs, ld = v.ld, lim = v.rhs - 1;
    const int tid = threadIdx.x, j = q * 256 + tid;
    const bool in = j < W;
    const int jc = in ? j : 0;
    const R32 * __restrict__ rin = v.stage + (size_t)slot * ld;
    R32 * __restrict__ rout = v.stage + (size_t)(slot ^ 1) * ld;
    const R32 * __restrict__ oin = v.stage + (size_t)(2 + slot) * ld;
    R32 * __restrict__ oout = v.stage + (size_t)(2 + (slot ^ 1)) * ld;
    const int enter2 = first;                          // (a launch with a ratio test has first in range; else the records say NONE)
    const int ec = (enter2 >= 0 && enter2 < W) ? enter2 : 0;
    // this pivot's objective row becomes the handle's; what does not depend on the pick is fetched before the wait
    const R32 o_in = oin[jc];
    if (in) v.obj[j] = o_in;
    const R32 e = rin[jc];
    const R32 cnv = oin[ec];                           // objective coefficient of the entering column (lpsol.h:1496)
    const bool nv_raw = jc < rhs && v.nv[jc] != 0;
    const int rcj = v.rowcnt[jc < rhs ? jc : 0];
    // basis after THIS pivot's swap (pick workgroup 0 commits it in this launch) = before the next one's
    const bool nvj = jc < rhs && (jc == ienter ? false : (jc == ileave ? true : nv_raw));

    if (q == 0 && tid == 0) FUSED_STAMP(st, 3);
    if (tid < 64) {
        // wave 0 polls the 4 N granules, one per lane, then combines the records in workgroup order (ties: lowest row,
        // lpsol.h:604-611) -- every lane the same
        const bool mine = tid < 4 * N;
        const char * gp = frec(v, mine ? tid >> 2 : 0) + 16 * (tid & 3);
        const unsigned long long t0 = wall_clock64();
        fr_u32x4 g;
        bool stuck = false;
        for (;;) {
            g = fr_load(gp);
            if (__all(!mine || g.z == total_now)) break;
            if (wall_clock64() - t0 > 400000000ull) { stuck = true; break; }      // 4 s: the pick never answered
            __builtin_amdgcn_s_sleep(4);
        }
        // every lane of a record's quad gets the record's four words, then a butterfly over the quads with the
        // reference's tie-break (lowest row, lpsol.h:604-611; better() is symmetric: both partners keep the same one)
        const int lo = (int)g.x, hi = (int)g.y;
        const int base = tid & ~3;
        int wl[4], wh[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { wl[k] = __shfl(lo, base + k); wh[k] = __shfl(hi, base + k); }
        const bool none = stuck || __ballot(mine && (int)g.w == FREC_NONE) != 0ull;
        const bool any_weird = __ballot(mine && (tid & 3) == 3 && (g.x & 0x80000000u) != 0u) != 0ull;
        Cand<R32> c; R32 g_a; int g_b, g_cc; uint32_t g_w;
        { R32 t; t.num = wl[0]; t.den = wh[0]; c.q = t; }
        c.idx = mine ? wh[2] : INT_MAX;
        g_a.num = wl[1]; g_a.den = wh[1];
        g_b = wl[2]; g_w = (uint32_t)wh[3]; g_cc = wl[3] & 0x7fffffff;
        for (int o = 4; o < 64; o <<= 1) {
            Cand<R32> t; t.q = shfl_xor_s(c.q, o); t.idx = __shfl_xor(c.idx, o);
            const R32 ta = shfl_xor_s(g_a, o);
            const int tb = __shfl_xor(g_b, o), tcc = __shfl_xor(g_cc, o); const uint32_t tw = (uint32_t)__shfl_xor((int)g_w, o);
            const Cand<R32> ng = better(c, t);
            if (ng.idx != c.idx) { g_a = ta; g_b = tb; g_cc = tcc; g_w = tw; }
            c = ng;
        }
        if (tid == 0) {
            const int kind = stuck ? -1 : (none ? FREC_NONE : ((c.idx == INT_MAX || any_weird) ? 0 : FREC_CAND));
            sh_res[0] = ((unsigned long long)(unsigned)c.idx << 32) | (unsigned)kind;
            sh_res[1] = (unsigned long long)(unsigned)g_b;
            sh_res[2] = to_bits(g_a);
            sh_res[3] = ((unsigned long long)g_w << 32) | (unsigned)g_cc;
        }
    }
    __syncthreads();
    const int kind = (int)(unsigned)sh_res[0], r2 = (int)(unsigned)(sh_res[0] >> 32), leave2 = (int)(unsigned)sh_res[1];
    const R32 piv = from_bits<R32>(sh_res[2]);
    if (q == 0 && tid == 0) FUSED_STAMP(st, 4);
    if (kind == -1) { if (q == 0 && tid == 0) st->status = XPG_ERR_CHAIN_STUCK; return; }
    if (kind == FREC_NONE) return;                     // no ratio test in this launch: pick workgroup 0 wrote the descriptor
    if (kind == 0) {                                   // first pass empty (second pass / disableNV), or candidates that only the
                                                       // reference's own scan order decides: the generic pick of the next generic point
        if (q == 0 && tid == 0) {
            write_desc(O, -1, 0, 0, first, anypos, 0, enter2, 0, done_now, total_now, 0ull, 0ull);
            O.side = side ^ 1; O.staged = 0; st->r32_side = side ^ 1;
        }
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    if (q == 0 && tid == 0) {                          // genPair, lpsol.h:100-104
        const uint32_t g_w = (uint32_t)(sh_res[3] >> 32); const int g_cc = (int)(unsigned)sh_res[3];
        if (!((g_w >> (leave2 & 31)) & 1u)) {
            v.ppt[(size_t)enter2 * v.pw + (leave2 >> 5)] = g_w | (1u << (leave2 & 31));
            v.rowcnt[enter2] = v.rowcnt[enter2] + 1; v.colcnt[leave2] = g_cc + 1;
        }
    }
    unsigned long long * go = go_block(v, slot);
    const R32 s = div(one<R32>(), piv);                        // 1/(eq.get(eqnum, nv)), lpsol.h:1471
    const int smode = scale_mode(s), cmode = scale_mode(cnv);
    const R32 kr = v.colbuf[(size_t)slot * colstride + r2];
    int nf = INT_MAX, any = 0;
    if (in) {
        const R32 aold = A[(size_t)r2 * ld + j];
        // row r2 after the running sweep, by the sweep's own rule (k_pipe_fused_r32 below; the pick's column likewise)
        const R32 a = r2 == r ? e : ((canon && e.num == 0 && j != enter2) ? aold : l_fma(canon, aold, kr, e));
        const R32 e2 = scaled_c(a, s, smode, canon);
        rout[j] = e2;
        R32 oj = o_in;
        if (j < enter2 && !nvj) oj = zero<R32>();              // lpsol.h:1055-1060, deferred by the pick (zero_upto = the entering column)
        const R32 o = obj_update_c(e2, j >= rhs, cnv, cmode, oj, canon);       // lpsol.h:1496-1501
        oout[j] = o;
        const bool nv_next = j == enter2 ? false : (j == leave2 ? true : nvj);
        if (j < rhs && nv_next && gt(o, zero<R32>())) { any = 1; if (rcj < lim) nf = j; }
    }
    for (int o = 32; o > 0; o >>= 1) { nf = min(nf, __shfl_xor(nf, o)); any |= __shfl_xor(any, o); }
    // the look-ahead accumulators rest at (INT_MAX, 0) between launches: only atomics touch them
    if ((tid & 63) == 0) {
        if (nf != INT_MAX) __hip_atomic_fetch_min(go + GO_NF, (unsigned long long)(unsigned)nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (any) __hip_atomic_fetch_or(go + GO_ANY, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid != 0) return;
    if (q == 0) FUSED_STAMP(st, 5);
    const unsigned long long arrived = __hip_atomic_fetch_add(go + GO_ARRIVED, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived != (unsigned long long)(NP - 1)) return;
    const int nfirst = (int)(unsigned)go_load(go + GO_NF), anyp = (int)go_load(go + GO_ANY);
    go_store(go + GO_ARRIVED, 0ull); go_store(go + GO_NF, (unsigned long long)(unsigned)INT_MAX); go_store(go + GO_ANY, 0ull);
    // (zero_upto = 0: the deferred zeroing of the objective row is already in the staged row)
    write_desc(O, r2, enter2, leave2, nfirst, anyp, 0, enter2, 0, done_now + 1, total_now + 1, to_bits(cnv), to_bits(piv));
    O.side = side ^ 1; O.staged = 1; st->r32_side = side ^ 1;
    FUSED_STAMP(st, 6);
}

// Grid (max(m, N + NP), 1 + column blocks): rows are the fast index (every column block spread over the 8 XCDs);
// blockIdx.y == 0 is dispatched first and holds the N pick workgroups, then the NP stagers (the rest of that row exits).
__global__ __launch_bounds__(256) XPG_R32_PIPE_ATTR
void k_pipe_fused_r32(LpView<R32> v, int slot, int colstride, int N, int NP)
{
    LoopState * st = v.st;
    const PipeDesc & D = st->pd[slot];
    const int status = st->status, r = D.row, first = desc_first(D), noncanon = st->noncanon, side = D.side, stop = D.stop, staged = D.staged;
    if (status != ST_RUNNING) return;
    const R32 * __restrict__ A = side ? v.tab2 : v.tab;
    R32 * __restrict__ B = side ? v.tab : v.tab2;
    if (blockIdx.y == 0) {
        const int x = blockIdx.x;
        if (x < N) {
            __builtin_amdgcn_s_setprio(3);
            fused_pick_r32(v, slot, colstride, x, N, A, B);
        } else if (x < N + NP) {
            fused_stage_r32(v, slot, colstride, x - N, N, NP, A);
        }
        return;
    }
    if (r < 0 || stop != 0 || !staged) return;               // (unstaged: see fused_pick_r32 -- nothing is written)
    const int i = blockIdx.x;
    if (i >= v.m) return;
    const int j = ((int)blockIdx.y - 1) * 256 + threadIdx.x;
    if (j >= v.W) return;
    const int xc = (first >= 0 && first < v.W) ? first : -1;
    if (j == xc) return;                                      // the pick workgroups' column
    const bool canon = noncanon == 0;
    const R32 e = v.stage[(size_t)slot * v.ld + j];
    const size_t off = (size_t)i * v.ld + j;
    const R32 a = A[off];
    // (the written side is not read again in this launch: streamed out, so that the launch does not end on the write-back
    // of 16 MB of dirty L2 lines -- XPG_FUSED_PLAIN_STORES builds keep the plain store for A/B runs)
    R32 o;
    if (i == r) o = e;                                        // the pivot row := e
    else if (canon && e.num == 0) o = a;                      // a + k * 0 = a exactly: copied, not computed
    else o = l_fma(canon, a, v.colbuf[(size_t)slot * colstride + i], e);
#ifdef XPG_FUSED_PLAIN_STORES
    B[off] = o;
#else
    __builtin_nontemporal_store(to_bits(o), (unsigned long long *)(B + off));
#endif
}

// The generic point: one workgroup of 1024 threads (the generic pick is a latency chain of strided gathers and per-row
// quotients: four rows per thread cost 25 us at 1024 rows with 256 threads). An idle descriptor gets its pivot IN PLACE
// and, in the same launch, its staging (scaled pivot row, objective row, look-ahead: what k_pipe_prep<R32> does, two
// columns per thread at W = 2049 -- a launch of its own cost 8.6 us for 3 us of work); a deferred final status is promoted.
__global__ __launch_bounds__(1024) void k_fused_generic(LpView<R32> v, int slot, int colstride)
{
    __shared__ int sh_d[4];
    __shared__ unsigned long long sh_b[2];
    __shared__ int sh_red[32];
    LoopState * st = v.st;
    PipeDesc & D = st->pd[slot];
    const int status = st->status, stop = D.stop, r0 = D.row, zu0 = D.zero_upto, side = D.side;
    const bool canon = st->noncanon == 0;
    if (status != ST_RUNNING) return;
    if (stop != 0) {
        for (int j = threadIdx.x; j < zu0; j += blockDim.x)
            if (!v.nv[j]) v.obj[j] = zero<R32>();              // lpsol.h:1055-1060, deferred by the pick
        __syncthreads();
        if (threadIdx.x == 0) st->status = stop;
        return;
    }
    if (r0 >= 0) return;
    LpView<R32> w = v;
    w.tab = side ? v.tab2 : v.tab;
    prep_idle<R32>(w, slot, colstride, true);
    __syncthreads();
    if (threadIdx.x == 0) {                                    // what the pick left in the descriptor, through the thread that wrote it
        sh_d[0] = D.row; sh_d[1] = D.col; sh_d[2] = D.leave; sh_d[3] = D.stop ? -1 : D.zero_upto;
        sh_b[0] = D.piv_bits; sh_b[1] = D.cnv_bits;
    }
    __syncthreads();
    const int r = sh_d[0], enter = sh_d[1], leave = sh_d[2], zu = sh_d[3];
    if (r < 0 || zu < 0) return;                               // nothing chosen (deferred again, or a final status)
    // ---- staging, with k_pipe_prep's arithmetic, into this slot's staging buffers (v.obj is committed by the launch that sweeps)
    const R32 * __restrict__ tab = w.tab;
    R32 * __restrict__ rowbuf = v.stage + (size_t)slot * v.ld;
    R32 * __restrict__ objout = v.stage + (size_t)(2 + slot) * v.ld;
    const R32 s = div(one<R32>(), from_bits<R32>(sh_b[0]));    // 1/(eq.get(eqnum, nv)), lpsol.h:1471
    const int smode = scale_mode(s);
    const R32 cnv = from_bits<R32>(sh_b[1]);
    const int cmode = scale_mode(cnv), lim = v.rhs - 1;
    int nf = INT_MAX, any = 0;
    for (int j = threadIdx.x; j < v.W; j += blockDim.x) {
        const R32 a = tab[(size_t)r * v.ld + j];
        R32 oj = v.obj[j];
        const bool nvj = j < v.rhs && v.nv[j] != 0;            // basis BEFORE this pivot's swap
        const int rcj = v.rowcnt[j < v.rhs ? j : 0];
        const R32 e = scaled_c(a, s, smode, canon);
        rowbuf[j] = e;
        if (j < zu && !nvj) oj = zero<R32>();                  // lpsol.h:1055-1060, deferred by the pick
        const R32 o = obj_update_c(e, j >= v.rhs, cnv, cmode, oj, canon);      // lpsol.h:1496-1501
        objout[j] = o;
        const bool nv_next = j == enter ? false : (j == leave ? true : nvj);
        if (j < v.rhs && nv_next && gt(o, zero<R32>())) { any = 1; if (rcj < lim) nf = min(nf, j); }
    }
    for (int o = 32; o > 0; o >>= 1) { nf = min(nf, __shfl_xor(nf, o)); any |= __shfl_xor(any, o); }
    if ((threadIdx.x & 63) == 0) { sh_red[threadIdx.x >> 6] = nf; sh_red[16 + (threadIdx.x >> 6)] = any; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)(blockDim.x >> 6);
        for (int k = 1; k < nw; k++) { nf = min(nf, sh_red[k]); any |= sh_red[16 + k]; }
        D.next_first = nf; D.anypos = any; D.staged = 1;
    }
}

// After the last fused launch of a call: side 0 is what everything else reads.
__global__ __launch_bounds__(256) void k_side_home(LpView<R32> v)
{
    if (!v.st->r32_side) return;
    const int i = blockIdx.x, j = (int)blockIdx.y * 256 + threadIdx.x;
    if (j >= v.W) return;
    const size_t off = (size_t)i * v.ld + j;
    v.tab[off] = v.tab2[off];
}

} // namespace xpg
