// Pipelined loop for the rational scalar (config 4): k_pipe_prep<R32> -> k_pipe_sweep_r32, two launches per pivot,
// the NEXT pivot chosen by a few pick workgroups inside the sweep launch -- the shape of the fp64 pipelined loop
// (lp_kernels.hip.h, "Pipelined fp64 loop"), with the same descriptors, deferrals and record hand-off.
//
// Why it pays here: the serial loop spent 10.7 + 6.6 us per pivot in its single-workgroup pick and its prep, both
// latency (dependent round trips and ~1500 dependent integer instructions per thread), next to a ~19 us sweep that
// is bound by integer issue. Under the sweep the pick costs nothing but one compute unit's share.
//   * What the pick needs of the post-sweep tableau it computes itself with the sweep's own operation
//     (fma_canon / the generic pair): the constant column lives in bcol[] for the whole loop, and the predicted
//     entering column is left alone by the sweep (one cell per row) and updated by the pick workgroups, who also keep
//     it in nextcol[] and write its negation as the next sweep's -column (the other half of colbuf).
//   * The pick's registers are the kernel's: the sweep keeps its speed down to 6 waves per SIMD (tools/lab/
//     rat_sweep_lab.hip: 8 -> 6 waves 0 %, 5 waves +8 %, 3 waves +20 %), so the kernel is held to 80 VGPRs.
//   * The pick waves raise their priority (s_setprio): they share their SIMDs with seven sweep waves each.
// Anything but "the first ratio pass found a row" is deferred by one iteration to a launch that sweeps nothing and
// runs the generic single-workgroup pick_body, exactly as in the fp64 loop.
#pragma once
#include "lp_kernels.hip.h"

namespace xpg {

__device__ __forceinline__ int pick_wgs_r32(int m) { const int n = (m + 255) / 256; return n < PICK_MAX_WGS ? n : PICK_MAX_WGS; }

// Nothing is being swept (descriptor row < 0: the loop's first iteration, or a decision the sweep-time pick deferred):
// the generic single-workgroup pick on a quiescent tableau. It runs in k_pipe_prep's first workgroup -- that launch
// has nothing to stage then -- so that its registers (pick_body's relaxed pass, disableNV, findPivotNVandBVPair)
// are not the sweep launch's.
// inplace (the fused loop's generic point, lp_fused_r32.hip.h): the pivot goes into pd[slot] itself and its -column into
// this slot's half of colbuf -- the launch that consumes it is the next one on the SAME slot.
template <> __device__ inline void prep_idle<R32>(const LpView<R32> & v, int slot, int colstride, bool inplace)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<R32>)];
    __shared__ int sh_i[16];
    __shared__ int sh_flag;
    Cand<R32> * sh_c = (Cand<R32> *)sh_c_raw;
    LoopState * st = v.st;
    PipeDesc & I = st->pd[slot];
    PipeDesc & O = st->pd[inplace ? slot : slot ^ 1];
    const int first = desc_first(I), anypos = I.anypos;
    const unsigned done_now = I.done_after, total_now = I.total_after, max_iter = st->max_iter;
    const int rhs = v.rhs, ld = v.ld, m = v.m;
    R32 * __restrict__ tab = v.tab;
    R32 * __restrict__ bcol = v.bcol;
    R32 * __restrict__ cbo = v.colbuf + (size_t)(inplace ? slot : slot ^ 1) * colstride;
    const int tid = threadIdx.x;
    const int izero = I.zero_upto, cached_col = I.cached_col;
    const bool b_was_cached = I.bcol_valid != 0;
    if (inplace) __syncthreads();                  // every thread holds its copy of the descriptor before it is rewritten
    for (int j = tid; j < izero; j += (int)blockDim.x)
        if (!v.nv[j]) v.obj[j] = zero<R32>();                              // deferred lpsol.h:1055-1060
    if (!b_was_cached)
        for (int i = tid; i < m; i += (int)blockDim.x) bcol[i] = tab[(size_t)i * ld + rhs];
    if (tid == 0) {
        write_desc(O, -1, 0, 0, INT_MAX, 0, 0, cached_col, 0, done_now, total_now, 0ull, 0ull);
        O.staged = 0;
    }
    __threadfence_block();
    __syncthreads();
    if (done_now >= max_iter) {                    // while (cnt < m_max_iter), lpsol.h:1039
        if (tid == 0) O.stop = 4;
        return;
    }
    const PickOut o = { &O.stop, &O.row, &O.col, &O.leave, &O.next_first, &O.anypos, &O.cnv_bits, &O.piv_bits };
    const bool chosen = pick_body<R32>(v, first, anypos, cached_col, true, false, o, cbo, sh_c, sh_i,
                                       &sh_flag, &O.zero_upto);
    if (chosen && tid == 0) { O.done_after = done_now + 1; O.total_after = total_now + 1; }
}

__device__ inline void pipe_pick_r32(const LpView<R32> & v, int slot, int colstride, int p, int N)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<R32>)];
    __shared__ unsigned long long sh_cnv;
    __shared__ int sh_rc;
    Cand<R32> * sh_c = (Cand<R32> *)sh_c_raw;
    LoopState * st = v.st;
    PipeDesc & I = st->pd[slot];
    PipeDesc & O = st->pd[slot ^ 1];
    const int r = I.row, ienter = I.col, ileave = I.leave, first = desc_first(I), anypos = I.anypos;
    const unsigned done_now = I.done_after, total_now = I.total_after, max_iter = st->max_iter;
    const bool cn = st->noncanon == 0;
    const int W = v.W, rhs = v.rhs, ld = v.ld, m = v.m, lim = v.rhs - 1;
    R32 * __restrict__ tab = v.tab;
    R32 * __restrict__ nextcol = v.nextcol;
    R32 * __restrict__ bcol = v.bcol;
    R32 * __restrict__ cbo = v.colbuf + (size_t)(slot ^ 1) * colstride;
    const int tid = threadIdx.x;

    if (r < 0) return;                                 // nothing swept: k_pipe_prep ran the generic pick (prep_idle)

    // ---- a sweep is running around us
    if (p == 0 && tid == 0) {                          // commit this iteration's pivot (lpsol.h:1504-1510)
        v.nv[ienter] = 0; v.nv[ileave] = 1; v.bv[ienter] = 1; v.bv[ileave] = 0;
        v.eq2bv[r] = ienter; v.bv2eq[ienter] = r; v.bv2eq[ileave] = -1;
        XPG_TRACE_PIVOT("hbm-pipe", ienter, ileave, r);
        const unsigned t = total_now - 1;
        if ((int)t < v.trace_cap) { v.trace[2 * t] = ienter; v.trace[2 * t + 1] = ileave; }
        st->total_pivots = total_now; st->done = done_now;
    }
    const int xc = (first >= 0 && first < W) ? first : -1;
    const R32 * __restrict__ cb = v.colbuf + (size_t)slot * colstride;
    const R32 * __restrict__ rb = v.rowbuf;
    const R32 eb = rb[rhs];
    const int stride = 256 * N;                        // rows are dealt to the workgroups in blocks of 256
    if (xc < 0 || done_now >= max_iter) {
        // no ratio test this time: keep the columns current, workgroup 0 records the outcome
        if (xc >= 0) {
            const R32 e0 = rb[xc];
            for (int i = p * 256 + tid; i < m; i += stride) {
                R32 * q = tab + (size_t)i * ld + xc;
                const R32 n0 = (i == r) ? e0 : l_fma(cn, *q, cb[i], e0);
                *q = n0;
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
            else                                       // findPivotNVandBVPair needs the whole tableau: next launch
                write_desc(O, -1, 0, 0, first, anypos, 0, -1, 0, done_now, total_now, 0ull, 0ull);
        }
        return;
    }

    // ---- fused pass over this workgroup's rows: new constant column, new entering column, -column, first ratio pass
    const R32 e0 = rb[xc];
    unsigned long long cnv_bits = 0; int rc_enter = 0;
    if (tid == 0) { cnv_bits = to_bits(v.obj[xc]); rc_enter = v.rowcnt[xc]; }   // for the last adder's tail
    Cand<R32> best; best.q = zero<R32>(); best.idx = INT_MAX;
    R32 best_a = zero<R32>(); int best_b = 0, best_cc = 0; uint32_t best_w = 0;
    bool weird = false;                                // a quotient with den <= 0: no order to reduce by (lp_kernels.hip.h)
    for (int i = p * 256 + tid; i < m; i += stride) {
        R32 * q = tab + (size_t)i * ld + xc;
        const R32 k = cb[i], bo = bcol[i], c0 = *q;    // every load of the row in flight before the first use
        int bi = v.eq2bv[i];
        if (i == r) bi = ienter;                       // the commit above, seen without waiting for it
        const uint32_t w = v.ppt[(size_t)xc * v.pw + (bi >> 5)];
        const int cc = v.colcnt[bi];
        R32 nb = l_fma(cn, bo, k, eb), a = l_fma(cn, c0, k, e0);               // the sweep's a + k*e
        if (i == r) { nb = eb; a = e0; }
        bcol[i] = nb;
        *q = a;
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
    __hip_atomic_store(rec + 0, to_bits(wbest.q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 1, to_bits(best_a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 2, ((unsigned long long)(unsigned)wbest.idx << 32) | (unsigned)best_b, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 3, ((unsigned long long)best_w << 32) | (unsigned)best_cc | (wg_weird ? 0x80000000u : 0u), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long arrived = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived != (unsigned long long)(N - 1)) return;
    // ---- last adder: combine the records in workgroup order (ties: lowest row, lpsol.h:604-611)
    Cand<R32> g; g.q = zero<R32>(); g.idx = INT_MAX;
    R32 g_a = zero<R32>(); int g_b = 0, g_cc = 0; uint32_t g_w = 0;
    bool any_weird = false;
    for (int k = 0; k < N; k++) {
        const unsigned long long * rk = v.pickrec + (size_t)k * PICK_REC_WORDS;
        const unsigned long long w0 = __hip_atomic_load(rk + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w1 = __hip_atomic_load(rk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w2 = __hip_atomic_load(rk + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w3 = __hip_atomic_load(rk + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Cand<R32> c; c.q = from_bits<R32>(w0); c.idx = (int)(unsigned)(w2 >> 32);
        any_weird |= ((unsigned)w3 & 0x80000000u) != 0u;
        const Cand<R32> ng = better(g, c);
        if (ng.idx != g.idx) { g_a = from_bits<R32>(w1); g_b = (int)(unsigned)w2; g_w = (uint32_t)(w3 >> 32); g_cc = (int)((unsigned)w3 & 0x7fffffffu); }
        g = ng;
    }
    if (g.idx == INT_MAX || any_weird) {               // first pass empty (second pass / disableNV), or candidates that only the
                                                       // reference's own scan order decides: the generic pick of the next launch
        write_desc(O, -1, 0, 0, first, anypos, 0, xc, 0, done_now, total_now, 0ull, 0ull);
        return;
    }
    const int enter = xc, leave = g_b;
    if (!((g_w >> (leave & 31)) & 1u)) {               // genPair, lpsol.h:100-104
        v.ppt[(size_t)enter * v.pw + (leave >> 5)] = g_w | (1u << (leave & 31));
        v.rowcnt[enter] = sh_rc + 1; v.colcnt[leave] = g_cc + 1;
    }
    write_desc(O, g.idx, enter, leave, INT_MAX, 0, 0, xc, first, done_now + 1, total_now + 1, sh_cnv, to_bits(g_a));
}

// Grid (m, 1 + column blocks): rows are the fast index (every column block spread over the 8 XCDs, lp_kernels.hip.h
// k_update_r32); blockIdx.y == 0 is dispatched first and holds the N pick workgroups (the rest of that row exits).
#ifndef XPG_R32_PIPE_ATTR
#define XPG_R32_PIPE_ATTR __attribute__((amdgpu_waves_per_eu(6, 8)))
#endif
__global__ __launch_bounds__(256) XPG_R32_PIPE_ATTR
void k_pipe_sweep_r32(LpView<R32> v, int slot, int colstride, int N)
{
    LoopState * st = v.st;
    const PipeDesc & D = st->pd[slot];
    const int status = st->status, r = D.row, first = desc_first(D), noncanon = st->noncanon;    // one round trip, then branch
    if (status != ST_RUNNING) return;
    if (blockIdx.y == 0) {
        if ((int)blockIdx.x < N) {
            __builtin_amdgcn_s_setprio(3);
            pipe_pick_r32(v, slot, colstride, blockIdx.x, N);
        }
        return;
    }
    if (r < 0) return;
    const int j = ((int)blockIdx.y - 1) * 256 + threadIdx.x;
    if (j >= v.W) return;
    const int xc = (first >= 0 && first < v.W) ? first : -1;
    if (j == xc) return;                                      // the pick workgroups' column
    const int i = blockIdx.x;
    const bool canon = noncanon == 0;
    const R32 e = v.rowbuf[j];
    R32 * p = v.tab + (size_t)i * v.ld + j;
    if (canon && e.num == 0) {
        if (i == r) *p = e;                                   // the pivot row's own cell := e
        return;
    }
    const R32 k = v.colbuf[(size_t)slot * colstride + i];
    *p = (i == r) ? e : l_fma(canon, *p, k, e);
}

} // namespace xpg
