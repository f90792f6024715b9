// Fused loop for the rational scalar (config 4): ONE launch per pivot. k_pipe_fused_r32 sweeps pivot t, chooses pivot
// t + 1 (the pick workgroups of lp_pipe_r32.hip.h) and STAGES it -- scaled pivot row, objective row, look-ahead pricing
// of t + 2 -- inside the same launch, which is what k_pipe_prep did in a launch of its own (6-7 us per pivot beside a
// 17.6 us sweep at 1024 x 2048: launch latency and a drained chip, the staging itself is 2048 cells).
//
// What that needs:
//   * The staging needs row r' of the tableau AFTER the running sweep, and r' is only known once the pick is done.
//     The sweep therefore does not work in place: it reads one copy of the tableau and writes the other (LpView::tab /
//     tab2, PipeDesc::side names the current one), so the stagers compute a'_r',j = a_r',j + k_r' * e_j themselves from
//     the immutable input side with the sweep's own operation -- no ordering against the workgroup that sweeps that row.
//     A column whose e_j is zero is copied instead of skipped.
//   * A hand-over inside the launch: the pick's last adder publishes the chosen pivot as a few agent-scope stores and a
//     flag (pickrec GO_*); ceil(W / 256) stager workgroups -- dispatched right behind the pick workgroups, spinning with
//     s_sleep meanwhile -- take it from there. Their look-ahead goes through agent-scope atomics on accumulators that
//     only atomics touch; the LAST stager to finish writes the whole next descriptor (one writer, plain stores: the
//     launch boundary publishes it).
//   * Nothing a reader of the handle sees may run ahead of the sweep: the staged scaled row and objective row live in
//     staging buffers per slot (LpView::stage); the stagers of the launch that SWEEPS a pivot commit its objective row
//     to v.obj first. The basis swap was already committed that way (pick workgroup 0).
//   * The generic pick (relaxed second pass, disableNV, findPivotNVandBVPair: 131 registers and scratch) must not be
//     compiled into this launch. The host enqueues a generic point -- k_pipe_prep<R32>(fused) twice: generic pick in
//     place, then staging of what it chose -- before the first launch of every queue_iterations call and every
//     XPG_R32_GENERIC_EVERY (8) launches; a deferred decision idles through the fused launches until then (workgroup
//     (0,0) carries the descriptor over to the other slot and counts it: a solve that idles often gets a generic point
//     before every launch from the host's next status read on).
//   * After the last launch of a call k_side_home copies side 1 onto side 0 when side 1 is the current one (both copies
//     are then current): everything outside this loop reads v.tab.
#pragma once
#include "lp_pipe_r32.hip.h"

namespace xpg {

__device__ __forceinline__ unsigned long long * go_block(const LpView<R32> & v, int slot)
{ return v.pickrec + PICK_GO_OFF + (size_t)PICK_GO_WORDS * slot; }
__device__ __forceinline__ void go_store(unsigned long long * p, unsigned long long x)
{ __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long go_load(const unsigned long long * p)
{ return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
enum { GO_NONE = 1, GO_PIVOT = 2 };
// the flag: kind in the high word, the ticket (total pivots of the handle once the swept pivot is committed: never
// repeats on a handle, never 0) in the low one
__device__ __forceinline__ void go_signal(unsigned long long * go, int kind, unsigned ticket)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    go_store(go + GO_FLAG, ((unsigned long long)(unsigned)kind << 32) | ticket);
}

// The pick workgroups (p = 0 .. N-1) of a fused launch. A: the tableau side the launch reads, B: the one it writes.
__device__ inline void fused_pick_r32(const LpView<R32> & v, int slot, int colstride, int p, int N, int NP,
                                      const R32 * __restrict__ A, R32 * __restrict__ B)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<R32>)];
    __shared__ unsigned long long sh_cnv;
    __shared__ int sh_rc;
    Cand<R32> * sh_c = (Cand<R32> *)sh_c_raw;
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
    const R32 * __restrict__ ostage = v.stage + (size_t)(2 + slot) * ld;       // the objective row with this pivot applied
    const int tid = threadIdx.x;

    if (stop != 0) {
        // workgroup 0 alone promotes the deferred final status (as k_pipe_prep does)
        if (p != 0) return;
        const int zu = I.zero_upto;
        for (int j = tid; j < zu; j += 256)
            if (!v.nv[j]) v.obj[j] = zero<R32>();              // lpsol.h:1055-1060, deferred by the pick
        __syncthreads();
        if (tid == 0) { O = I; st->status = stop; }
        return;
    }
    if (r < 0) {                                       // a deferred decision waits for the next generic point: carry it over
        if (p == 0 && tid == 0) { O = I; st->r32_idle += 1u; }
        return;
    }

    // ---- a sweep is running around us
    if (p == 0 && tid == 0) {                          // commit this iteration's pivot (lpsol.h:1504-1510)
        v.nv[ienter] = 0; v.nv[ileave] = 1; v.bv[ienter] = 1; v.bv[ileave] = 0;
        v.eq2bv[r] = ienter; v.bv2eq[ienter] = r; v.bv2eq[ileave] = -1;
        XPG_TRACE_PIVOT("hbm-fused", ienter, ileave, r);
        const unsigned t = total_now - 1;
        if ((int)t < v.trace_cap) { v.trace[2 * t] = ienter; v.trace[2 * t + 1] = ileave; }
        st->total_pivots = total_now; st->done = done_now;
    }
    unsigned long long * go = go_block(v, slot);
    const int xc = (first >= 0 && first < W) ? first : -1;
    const R32 * __restrict__ cb = v.colbuf + (size_t)slot * colstride;
    const R32 * __restrict__ rb = v.stage + (size_t)slot * ld;
    const R32 eb = rb[rhs];
    const int stride = 256 * N;                        // rows are dealt to the workgroups in blocks of 256
    if (xc < 0 || done_now >= max_iter) {
        // no ratio test this time: keep the columns current, workgroup 0 records the outcome
        if (xc >= 0) {
            const R32 e0 = rb[xc];
            for (int i = p * 256 + tid; i < m; i += stride) {
                const R32 n0 = (i == r) ? e0 : l_fma(cn, A[(size_t)i * ld + xc], cb[i], e0);
                B[(size_t)i * ld + xc] = n0;
                nextcol[i] = n0;
            }
        }
        for (int i = p * 256 + tid; i < m; i += stride)
            bcol[i] = (i == r) ? eb : l_fma(cn, bcol[i], cb[i], eb);
        if (p == 0 && tid == 0) {
            if (done_now >= max_iter)                  // while (cnt < m_max_iter), lpsol.h:1039
                write_desc(O, -1, 0, 0, first, anypos, 4, xc, 0, done_now, total_now, 0ull, 0ull);
            else if (first == INT_MAX && !anypos)      // optimum reached: lpsol.h:1089
                write_desc(O, -1, 0, 0, first, anypos, ST_CHECK_OPT, -1, rhs, done_now, total_now, 0ull, 0ull);
            else                                       // findPivotNVandBVPair needs the whole tableau: next generic point
                write_desc(O, -1, 0, 0, first, anypos, 0, -1, 0, done_now, total_now, 0ull, 0ull);
            O.side = side ^ 1; O.staged = 0; st->r32_side = side ^ 1;
            go_signal(go, GO_NONE, total_now);
        }
        return;
    }

    // ---- fused pass over this workgroup's rows: new constant column, new entering column, -column, first ratio pass
    const R32 e0 = rb[xc];
    unsigned long long cnv_bits = 0; int rc_enter = 0;
    if (tid == 0) { cnv_bits = to_bits(ostage[xc]); rc_enter = v.rowcnt[xc]; }   // for the last adder's tail
    Cand<R32> best; best.q = zero<R32>(); best.idx = INT_MAX;
    R32 best_a = zero<R32>(); int best_b = 0, best_cc = 0; uint32_t best_w = 0;
    bool weird = false;                                // a quotient with den <= 0: no order to reduce by (lp_kernels.hip.h)
    for (int i = p * 256 + tid; i < m; i += stride) {
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
    const Cand<R32> wbest = block_argmin(best, sh_c);
    const int wg_weird = __syncthreads_or(weird ? 1 : 0);
    // one lane publishes this workgroup's record: the owner of the winning row, else lane 0
    const bool publisher = wbest.idx != INT_MAX ? (best.idx == wbest.idx) : (tid == 0);
    if (tid == 0) { sh_cnv = cnv_bits; sh_rc = rc_enter; }
    __syncthreads();
    if (!publisher) return;
    unsigned long long * rec = v.pickrec + (size_t)p * PICK_REC_WORDS;
    unsigned long long * ctr = v.pickrec + PICK_CTR_OFF + 16 * slot;
    go_store(rec + 0, to_bits(wbest.q));
    go_store(rec + 1, to_bits(best_a));
    go_store(rec + 2, ((unsigned long long)(unsigned)wbest.idx << 32) | (unsigned)best_b);
    go_store(rec + 3, ((unsigned long long)best_w << 32) | (unsigned)best_cc | (wg_weird ? 0x80000000u : 0u));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long arrived = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived != (unsigned long long)(N - 1)) return;
    go_store(ctr, 0ull);                               // (no prep launch zeroes it in this loop)
    // ---- last adder: combine the records in workgroup order (ties: lowest row, lpsol.h:604-611)
    Cand<R32> g; g.q = zero<R32>(); g.idx = INT_MAX;
    R32 g_a = zero<R32>(); int g_b = 0, g_cc = 0; uint32_t g_w = 0;
    bool any_weird = false;
    for (int k = 0; k < N; k++) {
        const unsigned long long * rk = v.pickrec + (size_t)k * PICK_REC_WORDS;
        const unsigned long long w0 = go_load(rk + 0), w1 = go_load(rk + 1), w2 = go_load(rk + 2), w3 = go_load(rk + 3);
        Cand<R32> c; c.q = from_bits<R32>(w0); c.idx = (int)(unsigned)(w2 >> 32);
        any_weird |= ((unsigned)w3 & 0x80000000u) != 0u;
        const Cand<R32> ng = better(g, c);
        if (ng.idx != g.idx) { g_a = from_bits<R32>(w1); g_b = (int)(unsigned)w2; g_w = (uint32_t)(w3 >> 32); g_cc = (int)((unsigned)w3 & 0x7fffffffu); }
        g = ng;
    }
    if (g.idx == INT_MAX || any_weird) {               // first pass empty (second pass / disableNV), or candidates that only the
                                                       // reference's own scan order decides: the generic pick of the next generic point
        write_desc(O, -1, 0, 0, first, anypos, 0, xc, 0, done_now, total_now, 0ull, 0ull);
        O.side = side ^ 1; O.staged = 0; st->r32_side = side ^ 1;
        go_signal(go, GO_NONE, total_now);
        return;
    }
    const int enter = xc, leave = g_b;
    if (!((g_w >> (leave & 31)) & 1u)) {               // genPair, lpsol.h:100-104
        v.ppt[(size_t)enter * v.pw + (leave >> 5)] = g_w | (1u << (leave & 31));
        v.rowcnt[enter] = sh_rc + 1; v.colcnt[leave] = g_cc + 1;
    }
    // the stagers take it from here; the last of them writes the descriptor
    go_store(go + GO_ROWLEAVE, ((unsigned long long)(unsigned)g.idx << 32) | (unsigned)leave);
    go_store(go + GO_PIV, to_bits(g_a));
    go_store(go + GO_CNV, sh_cnv);
    go_store(go + GO_NF, (unsigned long long)(unsigned)INT_MAX);
    go_store(go + GO_ANY, 0ull);
    go_signal(go, GO_PIVOT, total_now);
    (void)NP;
}

// The stager workgroups (q = 0 .. NP-1, one column per thread) of a fused launch.
__device__ inline void fused_stage_r32(const LpView<R32> & v, int slot, int colstride, int q, int NP, const R32 * __restrict__ A)
{
    __shared__ unsigned long long sh_flag;
    LoopState * st = v.st;
    PipeDesc & I = st->pd[slot];
    PipeDesc & O = st->pd[slot ^ 1];
    const int r = I.row, ienter = I.col, ileave = I.leave, first = desc_first(I), stop = I.stop, side = I.side;
    const unsigned done_now = I.done_after, total_now = I.total_after;
    const bool canon = st->noncanon == 0;
    if (r < 0 || stop != 0) return;
    const int W = v.W, rhs = v.rhs, ld = v.ld, lim = v.rhs - 1;
    const int tid = threadIdx.x, j = q * 256 + tid;
    const bool in = j < W;
    const int jc = in ? j : 0;
    const R32 * __restrict__ rin = v.stage + (size_t)slot * ld;
    R32 * __restrict__ rout = v.stage + (size_t)(slot ^ 1) * ld;
    const R32 * __restrict__ oin = v.stage + (size_t)(2 + slot) * ld;
    R32 * __restrict__ oout = v.stage + (size_t)(2 + (slot ^ 1)) * ld;
    // this pivot's objective row becomes the handle's; what does not depend on the pick is fetched before the wait
    const R32 o_in = oin[jc];
    if (in) v.obj[j] = o_in;
    const R32 e = rin[jc];
    const bool nv_raw = jc < rhs && v.nv[jc] != 0;
    const int rcj = v.rowcnt[jc < rhs ? jc : 0];
    // basis after THIS pivot's swap (pick workgroup 0 commits it in this launch) = before the next one's
    const bool nvj = jc < rhs && (jc == ienter ? false : (jc == ileave ? true : nv_raw));

    unsigned long long * go = go_block(v, slot);
    if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        unsigned long long f;
        for (;;) {
            f = go_load(go + GO_FLAG);
            if ((unsigned)f == total_now) break;
            if (wall_clock64() - t0 > 400000000ull) { f = 0ull; break; }      // 4 s: the pick never answered
            __builtin_amdgcn_s_sleep(8);
        }
        sh_flag = f;
    }
    __syncthreads();
    const unsigned long long flag = sh_flag;
    if (flag == 0ull) { if (q == 0 && tid == 0) st->status = XPG_ERR_CHAIN_STUCK; return; }
    if ((int)(flag >> 32) != GO_PIVOT) return;         // the pick wrote the descriptor itself
    __builtin_amdgcn_s_setprio(3);
    const unsigned long long rl = go_load(go + GO_ROWLEAVE);
    const int r2 = (int)(unsigned)(rl >> 32), leave2 = (int)(unsigned)rl, enter2 = first;
    const R32 piv = from_bits<R32>(go_load(go + GO_PIV)), cnv = from_bits<R32>(go_load(go + GO_CNV));
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
    if ((tid & 63) == 0) {
        if (nf != INT_MAX) __hip_atomic_fetch_min(go + GO_NF, (unsigned long long)(unsigned)nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (any) __hip_atomic_fetch_or(go + GO_ANY, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid != 0) return;
    const unsigned long long arrived = __hip_atomic_fetch_add(go + GO_ARRIVED, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived != (unsigned long long)(NP - 1)) return;
    go_store(go + GO_ARRIVED, 0ull);
    const int nfirst = (int)(unsigned)go_load(go + GO_NF), anyp = (int)go_load(go + GO_ANY);
    // (zero_upto = 0: the deferred zeroing of the objective row is already in the staged row)
    write_desc(O, r2, enter2, leave2, nfirst, anyp, 0, enter2, 0, done_now + 1, total_now + 1, to_bits(cnv), to_bits(piv));
    O.side = side ^ 1; O.staged = 1; st->r32_side = side ^ 1;
}

// Grid (max(m, N + NP), 1 + column blocks): rows are the fast index (every column block spread over the 8 XCDs);
// blockIdx.y == 0 is dispatched first and holds the N pick workgroups, then the NP stagers (the rest of that row exits).
__global__ __launch_bounds__(256) XPG_R32_PIPE_ATTR
void k_pipe_fused_r32(LpView<R32> v, int slot, int colstride, int N, int NP)
{
    LoopState * st = v.st;
    const PipeDesc & D = st->pd[slot];
    const int status = st->status, r = D.row, first = desc_first(D), noncanon = st->noncanon, side = D.side, stop = D.stop;
    if (status != ST_RUNNING) return;
    const R32 * __restrict__ A = side ? v.tab2 : v.tab;
    R32 * __restrict__ B = side ? v.tab : v.tab2;
    if (blockIdx.y == 0) {
        const int x = blockIdx.x;
        if (x < N) {
            __builtin_amdgcn_s_setprio(3);
            fused_pick_r32(v, slot, colstride, x, N, NP, A, B);
        } else if (x < N + NP) {
            fused_stage_r32(v, slot, colstride, x - N, NP, A);
        }
        return;
    }
    if (r < 0 || stop != 0) return;
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
    if (i == r) { B[off] = e; return; }                       // the pivot row := e
    if (canon && e.num == 0) { B[off] = a; return; }          // a + k * 0 = a exactly: copied, not computed
    const R32 k = v.colbuf[(size_t)slot * colstride + i];
    B[off] = l_fma(canon, a, k, e);
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
