// Device-resident slack-form simplex for ONE large tableau (configs 2 and 4):
// every step of SIX::solveSlackForm's loop (src/com/lpsol.h:1039-1188) is a
// kernel template over the scalar S (xpg::F64 or xpg::R32); the host only
// queues launches and polls a status word every few dozen pivots.
//
// Serial loop (rational scalar, XPG_LOOP=serial): three launches per iteration,
// communicating through LoopState:
//   k_pick   : ratio test (lpsol.h:553-663) on the look-ahead column, pivot-pair table
//              upkeep (lpsol.h:68-154), basis swap, -column -> colbuf   -- 1 workgroup
//   k_prep   : row * 1/pivot -> rowbuf, objective row update (lpsol.h:1471-1474,
//              :1496-1501) and look-ahead pricing of the next iteration (lpsol.h:1054-1069)
//   k_update : a_ij += (-a_i,nv) * e_j for every i != r, all j   (lpsol.h:1481-1490)
//              -- the HBM-bound sweep, >= 2048 workgroups; it also exports the
//              look-ahead column and the constant column contiguously so the
//              next k_pick's ratio test reads 2 x 32 KB coalesced instead of
//              2 x 4096 64-byte sectors
// Pipelined fp64 loop (default): k_pipe_prep + k_pipe_sweep, the next pivot being chosen
// by extra workgroups inside the sweep launch -- see "Pipelined fp64 loop" below.
//
// HBM layout: tableau row-major, leading dimension ld (multiple of 16 elements
// = 128 B so every row starts on a cache line and 16-byte vector accesses are
// aligned), objective row / rowbuf / colbuf contiguous. The pivot-pair table is
// a bit matrix (n x ceil(n/32) words) with per-row and per-column counters so
// canBeNVCandidate / canBeBVCandidate (lpsol.h:124-153) are O(1).
#pragma once
#include "scalar.hip.h"
#include "rat_ops.hip.h"
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
    int noncanon;          // rational LP: some input entry is not in lowest terms with den > 0 (k_build); the sweep then
                           // keeps to the reference's two generic operations per cell (no fma_canon, no zero-column skip)
    // look-ahead pricing, filled by k_prep's atomics after the objective update:
    int next_first;        // lowest eligible entering column of the next iteration
                           // (INT_MAX: none, NF_UNKNOWN: not computed)
    int anypos;            // some nonbasic reduced cost is > 0
    int cached_col;        // column currently held in nextcol[] (-1: none), set by the sweep
    int bcol_valid;        // bcol[] holds the current constant column
    // Pipelined loop (k_pipe_prep / k_pipe_sweep): iteration t works from pd[t & 1] while the
    // pick workgroup riding in its sweep launch writes pd[(t + 1) & 1].
    struct PipeDesc {
        int row, col, leave;           // the pivot of this iteration (row < 0: none)
        int next_first, anypos;        // look-ahead pricing for the following iteration
        int stop;                      // != 0: final status, promoted to `status` by k_pipe_prep
        int cached_col, bcol_valid;    // what nextcol[] / bcol[] hold for the following pick
        int zero_upto;                 // basic objective entries below this index are still to be zeroed
        unsigned done_after, total_after;   // LoopState::done / total_pivots once this pivot is committed
        int side;                      // fused Rational loop (lp_fused_r32.hip.h): the tableau copy that is current for this descriptor
        int staged, pad_;              // ... and: its scaled pivot row / objective row are already in the staging buffers
        unsigned long long cnv_bits, piv_bits;
        unsigned long long price_key;  // Dantzig look-ahead (atomicMax of dz_key), 0: none / not used
    } pd[2];
    // opt-in NON-PARITY modes of the fp64 loop; both 0 = the reference's behaviour
    int pricing;           // 1: Dantzig's rule (largest reduced cost) instead of the first positive one
    int r32_side;          // fused Rational loop: the tableau copy the LAST launch left current (k_side_home reads it)
    unsigned r32_idle;     // fused Rational loop: launches that found a deferred decision and carried it over
    double feas_tol;       // > 0: SIX::is_feasible with this relative tolerance instead of Float's 1e-17 '=='
    // Blocked fp64 loop (lp_blocked.hip.h): up to BLK_MAX pivots are chosen and staged against the
    // un-swept tableau, then ONE sweep applies them all. Fields tagged with the batch they belong to.
    struct Blk {
        int batch;             // batch the fields below belong to (stale otherwise: n = 0, open)
        int n;                 // pivots staged so far in this batch
        int closed;            // no further pivot may join this batch (rare branch met): sweep, then retry
        int generic;           // the first pick of the batch hands over to the generic single-workgroup pick
        int from_generic;      // the pivot now in row/col/leave was chosen by the generic pick (column in colbuf)
        unsigned budget;       // loop iterations the host still allows (xpg_lp_iterate)
        int r[32];             // pivot rows of the staged pivots (BLK_MAX)
        unsigned long long price_key;   // Dantzig look-ahead of the blocked loop's prep
        unsigned la_epoch;     // tag of the prep whose partials (blkP) hold the current look-ahead
        int want_generic;      // a fast pick found no row in its first pass: the next batch starts generic
        int la_from_state;     // the current look-ahead is next_first / anypos (left by reset or the generic
                               // pick), not the per-workgroup partials of the last prep
        unsigned ch_epoch;     // chain kernel ticket: epoch of the stage-0 prep that staged this batch's first pivot
        unsigned ch_budget, ch_done, ch_tp;   // budget / done / total_pivots after that stage
        unsigned ch_arrive[8]; // chain kernel: workers of this batch's launch that have started, by the XCD they run on (the spread
                               // form counts in [0] only; zeroed by stage 0's prep)
        unsigned ch_misplaced; // one-XCD chain launches aborted because their workers were NOT all on one XCD
        // the ticket of a chain launch that does a batch's stage 0 itself (lp_chain.hip.h, t0 = 0): written by the committer of the
        // chain before it, once that one has committed its last stage
        unsigned ch0_ticket;   // blk_ticket0(batch) of the batch it admits
        unsigned ch0_la_epoch; // the tag its look-ahead partials carry
        unsigned ch0_budget, ch0_done, ch0_tp;
        unsigned ch_folds;     // chain launches that did their batch's stage 0 themselves (xpg_lp_chain_aborts)
        unsigned ch_decide;    // the roll call's verdict for this batch's launch: 0 open, CH_GO, CH_ABORT -- set ONCE, by compare-and-swap
                               // (the committer when everybody has counted in or its patience ends; a worker that has waited
                               // for the verdict far longer than that: the committer itself never got a CU); zeroed by stage 0's prep
        unsigned ch_aborts;    // chain launches given up before their first stage because not every worker got a CU in time
        unsigned ch_runs;      // chain launches that passed their roll call
        unsigned sweeps_full;  // sweeps that applied a full batch of BLK_MAX pivots (xpg_lp_counters)
        unsigned sweeps_part;  // sweeps that applied fewer (budget ran out, or a pick closed the batch early)
        unsigned long long dbg[8];   // diagnostic builds (-DXPG_STAMPS): 100 MHz ticks between points of pick / prep
    } blk;
};
enum { BLK_MAX = 32,           // the most pivots a batch stages (XPG_BLOCK up to this)
       BLK_DEFAULT = 32,       // the default batch length. Rounds 4-5 ran 24: the pass applying 32 staged pivots (32 register pairs of
                               // e_s, three waves per SIMD) took 110 / 172 us at 4096 x 8192 / 4096 x 12289 against 83 / 134 us with 24 --
                               // it waited for the rows' scalar k_s in front of every eight stages. With those requested a group ahead
                               // (k_blk_sweep_full) it takes 93.5 / 156 us: 134.4 k / 128.7 k / 107.2 k pivots/s at 4096 x 8192 with
                               // 32 / 24 / 16, 104.2 k / 97.3 k / 79.5 k at 4096 x 12289 (profiles/round5_block_length_ab.txt); 32 is
                               // ahead from 512 x 1024 up (tools/lab/probe_nap_sizes.py)
       BLK_REC_WORDS = 16,
       BLK_PICK_WGS = 64,      // pick workgroups (= records) of the launch-per-stage path: one lane of a wave combines each
       BLK_REC_MAX = 256,      // records of the chain kernel: one per pick worker
       BLK_PART_MAX = 520,     // partial slots: one per prep worker (<= 510), the commit granule behind them; the last slot is the roll-call decision
       BLK_DECISION_SLOT = BLK_PART_MAX - 1,
       BLK_TPB_MIN = 64 };     // smallest workgroup of pick / prep: sizes the partial array
// Hand-off areas of the blocked loop. What MANY waves poll lies DENSE -- one 16-byte granule {data, tag} per producer,
// back to back, so that a polling wave's load covers 64 granules in eight 128-byte lines instead of 40 (80-byte
// partials) or 64 (128-byte records) -- and what only the consumer of ONE producer reads (the payload) lies behind it.
//   blkR (8-byte words): [0, 64 x 16)        the launch-per-stage records (blk_pick_body -> blk_prep_body), 16 words each
//                        + 2 w               chain record w, g0 {ratio key, row, tag}: polled by every worker
//                        + 2 x 256 + 8 w     its payload g1 {pivot element, leaving} g2 {pair word, counter, entering | last
//                                            stage} g3 {c_nv}: fetched of the winner only
//   blkP (ints):         4 w                 partial w, g0 {lowest eligible column, any c_j > 0, tag, 0}; the chain's commit granule, in the
//                                            slot behind its last prep worker: {INT_MAX, 0, tag, pivot row | CH_CLOSE_ROW}
//                        4 x 520 + 16 w      its payload g1 {e_t[that column]} g2 {its objective entry} g3 {e_t[rhs]} (the
//                                            owner of the constant column), then the Dantzig key (2 ints)
enum { BLK_REC_G0 = BLK_PICK_WGS * BLK_REC_WORDS, BLK_REC_PAY = BLK_REC_G0 + 2 * BLK_REC_MAX, BLK_REC_PAY_WORDS = 8,
       BLK_REC_TOTAL_WORDS = BLK_REC_PAY + BLK_REC_PAY_WORDS * BLK_REC_MAX,
       BLK_PART_PAY = 4 * BLK_PART_MAX, BLK_PART_PAY_INTS = 16, BLK_PART_TOTAL_INTS = BLK_PART_PAY + BLK_PART_PAY_INTS * BLK_PART_MAX };
enum { NF_UNKNOWN = -2 };
typedef LoopState::PipeDesc PipeDesc;
// LpView::pickrec layout (8-byte words): PICK_MAX_WGS records of PICK_REC_WORDS, then one arrival
// counter per descriptor slot, each on a 128-byte line of its own; the fused Rational loop adds one hand-over block of
// PICK_GO_WORDS per slot (GO_*: the look-ahead accumulators, the stagers' arrival counter) and the pick workgroups'
// records as tagged granules (PICK_FREC_*).
enum { PICK_MAX_WGS = 16, PICK_REC_WORDS = 4, PICK_CTR_OFF = 128, PICK_GO_OFF = PICK_CTR_OFF + 32, PICK_GO_WORDS = 32,
       PICK_FREC_OFF = PICK_GO_OFF + 2 * PICK_GO_WORDS, PICK_FREC_WORDS = 8, PICK_WORDS = PICK_FREC_OFF + PICK_MAX_WGS * PICK_FREC_WORDS };
enum { GO_NF = 16, GO_ANY = 17, GO_ARRIVED = 18 };

template <class S> struct LpView {
    S * tab; int m, W, ld, rhs;
    S * obj;
    uint8_t * nv; uint8_t * bv; int * bv2eq; int * eq2bv;
    uint32_t * ppt; int pw; int * rowcnt; int * colcnt;
    S * rowbuf; S * colbuf; S * x; S * vcd; S * vcr;
    S * nextcol; S * bcol;   // contiguous copies of the predicted entering column / constant column
    unsigned long long * pickrec;   // pipelined loop: per-workgroup ratio-test records + arrival counters
    S * blkK; S * blkE;             // blocked loop: -column of staged pivot s at blkK[i * BLK_MAX + s], its scaled row at blkE[s * ld + j]
    unsigned long long * blkR;      // blocked loop: ratio-test records of the pick workgroups (BLK_REC_WORDS each)
    int * blkP;                     // blocked loop: look-ahead pricing partials of the prep workgroups (BLK_PART_INTS each)
    LoopState * st;
    int * trace; int trace_cap;
    S * tab2;                       // fused Rational loop: the other copy of the ping-pong tableau (null: not in use)
    S * stage;                      // fused Rational loop: scaled pivot row [2][ld], then staged objective row [2][ld]
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

// Wave-wide arg-min on the VALU: four DPP row_shr steps reduce each 16-lane row into its last
// lane, four v_readlane pairs fetch the row results, the final combine is scalar. This
// replaces six dependent ds_bpermute stages (LDS-latency each) in the selection chain.
template <int CTRL> __device__ __forceinline__ int dpp_row_shr(int own)
{ return __builtin_amdgcn_update_dpp(own, own, CTRL, 0xf, 0xf, false); }   // lanes without a source keep `own`

template <class S, int CTRL> __device__ __forceinline__ Cand<S> dpp_step(Cand<S> c)
{
    int w[2];
    __builtin_memcpy(w, &c.q, 8);
    w[0] = dpp_row_shr<CTRL>(w[0]); w[1] = dpp_row_shr<CTRL>(w[1]);
    Cand<S> t;
    __builtin_memcpy(&t.q, w, 8);
    t.idx = dpp_row_shr<CTRL>(c.idx);
    return better(c, t);
}
template <class S> __device__ __forceinline__ Cand<S> read_lane(Cand<S> c, int lane)
{
    int w[2];
    __builtin_memcpy(w, &c.q, 8);
    w[0] = __builtin_amdgcn_readlane(w[0], lane); w[1] = __builtin_amdgcn_readlane(w[1], lane);
    Cand<S> t;
    __builtin_memcpy(&t.q, w, 8);
    t.idx = __builtin_amdgcn_readlane(c.idx, lane);
    return t;
}
template <class S> __device__ __forceinline__ Cand<S> wave_argmin(Cand<S> c)
{
    c = dpp_step<S, 0x111>(c);      // row_shr:1
    c = dpp_step<S, 0x112>(c);      // row_shr:2
    c = dpp_step<S, 0x114>(c);      // row_shr:4
    c = dpp_step<S, 0x118>(c);      // row_shr:8 -> lanes 15, 31, 47, 63 hold their row's winner
    return better(better(read_lane(c, 15), read_lane(c, 31)), better(read_lane(c, 47), read_lane(c, 63)));
}

// fp64 ratio tests, the cheap form: the ratio as a 64-bit key whose UNSIGNED order is better()'s order on ratios that are
// not NaN (-0 and +0 are one value, then the usual monotone map of IEEE doubles), reduced by four DPP steps on 64-bit
// integers, the lowest lane among the minima by ballot -- a third of the instructions of the Cand<F64> tree (three
// registers through every step, two fp64 compares and an index compare per step). A NaN ratio compares "equal" to
// everything in better() (both `>` are false), which no key -- and no tree -- can mimic: if any candidate ratio of the wave
// is a NaN the rows are scanned in the reference's order (scan_step_in_order below is the same loop).
__device__ __forceinline__ unsigned long long ratio_key_f64(double q)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, q + 0.0);   // (-0) + (+0) = +0
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
template <int CTRL> __device__ __forceinline__ unsigned long long u64_min_step(unsigned long long x)
{
    const unsigned lo = (unsigned)dpp_row_shr<CTRL>((int)(unsigned)x), hi = (unsigned)dpp_row_shr<CTRL>((int)(unsigned)(x >> 32));
    const unsigned long long t = ((unsigned long long)hi << 32) | lo;
    return t < x ? t : x;
}
__device__ __forceinline__ unsigned long long wave_min_u64_dpp(unsigned long long x)
{
    x = u64_min_step<0x111>(x); x = u64_min_step<0x112>(x); x = u64_min_step<0x114>(x); x = u64_min_step<0x118>(x);
    unsigned long long r[4];
    for (int k = 0; k < 4; k++) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, 16 * k + 15);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 16 * k + 15);
        r[k] = ((unsigned long long)hi << 32) | lo;
    }
    const unsigned long long ab = r[0] < r[1] ? r[0] : r[1], cd = r[2] < r[3] ? r[2] : r[3];
    return ab < cd ? ab : cd;
}
// The row (= lane) of the least ratio among the lanes with `valid`, lowest row on ties; INT_MAX if there is none.
__device__ __forceinline__ int wave_argmin_row(F64 q, bool valid, int lane)
{
    if (__builtin_expect(__ballot(valid && q.v != q.v) != 0ull, 0)) {   // a NaN among the candidates: no order, the reference's scan itself
        int best = INT_MAX; F64 bestq = F64(0.0);
        unsigned long long m = __ballot(valid);
        Cand<F64> c; c.q = q; c.idx = 0;
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const F64 ql = read_lane(c, l).q;
            if (best == INT_MAX || gt(bestq, ql)) { bestq = ql; best = l; }
        }
        return best;
    }
    const unsigned long long key = valid ? ratio_key_f64(q.v) : ~0ull;
    const unsigned long long kmin = wave_min_u64_dpp(key);
    if (kmin == ~0ull) return INT_MAX;
    return __ffsll((long long)__ballot(key == kmin)) - 1;
}
// Rational comparisons order by cross-multiplication (rational.h: a.num * b.den < a.den * b.num), which is a strict
// weak order only while every denominator is positive. The reference's own operations can leave den = 0 behind
// (convertEq2Ineq divides by an equality's entry at the INEQUALITY's row index, lpsol.h:1232, which may be 0; MIP's
// substituted nodes), and then +n/0 > 5 > -n/0 while +n/0 and -n/0 compare "equal": the winner of a scan depends on the
// order the candidates are met in, and only the reference's own order -- rows ascending, `if (best > q) best = q` --
// reproduces it. Every arg-min over Rational candidates therefore checks for such a candidate and, if there is one,
// replays the sequential scan (64 rows per step: the lanes fetch, one scalar loop combines them in row order).
template <class S> __device__ __forceinline__ bool unordered_value(S) { return false; }
template <> __device__ __forceinline__ bool unordered_value<R32>(R32 q) { return q.den <= 0; }
template <> __device__ __forceinline__ bool unordered_value<F64>(F64 q) { return q.v != q.v; }   // a NaN ratio (inf / inf, 0 * inf upstream): `>` is false both ways
// One step of that scan over the 64 candidates of a wave (lane l = candidate base + l): best / bestq carry over.
template <class S> __device__ __forceinline__ void scan_step_in_order(S q, bool valid, int base, int & best, S & bestq)
{
    unsigned long long m = __ballot(valid);
    Cand<S> c; c.q = q; c.idx = 0;
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const S ql = read_lane(c, l).q;
        if (best == INT_MAX || gt(bestq, ql)) { bestq = ql; best = base + l; }
    }
}
template <class S> __device__ __forceinline__ int wave_argmin_row(S q, bool valid, int lane)   // Rational: the Cand tree
{
    if (__builtin_expect(__ballot(valid && unordered_value(q)) != 0ull, 0)) {   // not an order: the reference's scan itself
        int best = INT_MAX; S bestq = zero<S>();
        scan_step_in_order(q, valid, 0, best, bestq);
        return best;
    }
    Cand<S> c; c.q = q; c.idx = valid ? lane : INT_MAX;
    return __builtin_amdgcn_readfirstlane(wave_argmin(c).idx);
}

// Wave-wide integer minimum the same way.
__device__ __forceinline__ int wave_min_int(int x)
{
    x = min(x, dpp_row_shr<0x111>(x));
    x = min(x, dpp_row_shr<0x112>(x));
    x = min(x, dpp_row_shr<0x114>(x));
    x = min(x, dpp_row_shr<0x118>(x));
    return min(min(__builtin_amdgcn_readlane(x, 15), __builtin_amdgcn_readlane(x, 31)),
               min(__builtin_amdgcn_readlane(x, 47), __builtin_amdgcn_readlane(x, 63)));
}

template <class S> __device__ __forceinline__ bool ppt_seen(const LpView<S> & v, int nv, int b)
{ return (v.ppt[(size_t)nv * v.pw + (b >> 5)] >> (b & 31)) & 1u; }

// SIX::findPivotBV (lpsol.h:553-663) by one workgroup: rows strided over
// threads, arg-min of b_i / a_i,nv with the lowest row winning ties.
// The column and the constant column come from the contiguous copies the
// previous sweep exported when they are current (col_cached / b_cached), else
// from the tableau (one 64-byte sector per element).
// pass0: -2 run both passes; -1 the caller already ran the first pass and found no row.
template <class S> __device__ int ratio_test(const LpView<S> & v, int nv, Cand<S> * sh,
                                             bool col_cached = false, bool b_cached = false, int pass0 = -2)
{
    const int lim = v.rhs - 1, T = blockDim.x;
    constexpr int U = 4;
    const bool cn = !is_f64<S>::value && v.st->noncanon == 0;      // Rational: the canonical quotient (rat_ops.hip.h)
    for (int pass = pass0 == -1 ? 1 : 0; pass < 2; pass++) {
        Cand<S> best; best.q = zero<S>(); best.idx = INT_MAX;
        bool weird = false;
        // every load of a row is issued before the first test (two dependent rounds per U rows
        // instead of four per row): this workgroup is latency-bound, not bandwidth-bound
        for (int i0 = threadIdx.x; i0 < v.m; i0 += U * T) {
            S a[U], bb[U]; int b[U], cc[U]; uint32_t w[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = min(i0 + u * T, v.m - 1);
                a[u] = col_cached ? v.nextcol[i] : v.tab[(size_t)i * v.ld + nv];
                bb[u] = b_cached ? v.bcol[i] : v.tab[(size_t)i * v.ld + v.rhs];
                b[u] = v.eq2bv[i];
            }
#pragma unroll
            for (int u = 0; u < U; u++) { w[u] = v.ppt[(size_t)nv * v.pw + (b[u] >> 5)]; cc[u] = v.colcnt[b[u]]; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * T;
                if (i >= v.m) continue;
                if (pass == 0 ? le(a[u], zero<S>()) : eq(a[u], zero<S>())) continue;
                if (((w[u] >> (b[u] & 31)) & 1u) || cc[u] >= lim) continue;
                Cand<S> c; c.q = l_div(cn, bb[u], a[u]); c.idx = i;
                weird |= unordered_value(c.q);
                best = better(best, c);
            }
        }
        best = block_argmin(best, sh);
        if (__builtin_expect(__syncthreads_or(weird ? 1 : 0), 0)) {
            // a candidate quotient with den <= 0: replay the reference's scan of this pass (wave 0, rows in order)
            __shared__ int sh_seq;
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                int sbest = INT_MAX; S sq = zero<S>();
                for (int base = 0; base < v.m; base += 64) {
                    const int i = min(base + lane, v.m - 1);
                    const S a = col_cached ? v.nextcol[i] : v.tab[(size_t)i * v.ld + nv];
                    const S bb = b_cached ? v.bcol[i] : v.tab[(size_t)i * v.ld + v.rhs];
                    const int b = v.eq2bv[i];
                    const uint32_t w = v.ppt[(size_t)nv * v.pw + (b >> 5)];
                    const int cc = v.colcnt[b];
                    bool ok = base + lane < v.m && !(pass == 0 ? le(a, zero<S>()) : eq(a, zero<S>()));
                    ok = ok && !(((w >> (b & 31)) & 1u) || cc >= lim);
                    const S q = ok ? l_div(cn, bb, a) : zero<S>();
                    scan_step_in_order(q, ok, base, sbest, sq);
                }
                if (lane == 0) sh_seq = sbest;
            }
            __syncthreads();
            best.idx = sh_seq;
            __syncthreads();
        }
        if (best.idx != INT_MAX) return v.eq2bv[best.idx];
    }
    return -1;
}

// In-workgroup pricing scan (lpsol.h:1054-1069 without the side effect): lowest
// nonbasic j with c_j > 0 whose pair-table row is still open, and whether any
// nonbasic c_j > 0 exists at all.
// Opt-in NON-PARITY pricing: Dantzig's rule, the largest reduced cost instead of
// the reference's first positive one. A candidate is one 64-bit key -- the fp64 bits of c_j > 0 (which
// order like integers) with the low 20 mantissa bits replaced by ~j, so that an integer max picks
// the largest cost and, among costs equal to 2^-32 relative, the lowest column.
__device__ __forceinline__ unsigned long long dz_key(double c, int j)
{ unsigned long long b; __builtin_memcpy(&b, &c, 8); return (b & ~0xFFFFFull) | (unsigned long long)(0xFFFFF - j); }
__device__ __forceinline__ int dz_col(unsigned long long key) { return key ? 0xFFFFF - (int)(key & 0xFFFFFull) : INT_MAX; }
template <class S> __device__ __forceinline__ unsigned long long dz_key_of(S c, int j) { return 0ull; }
template <> __device__ __forceinline__ unsigned long long dz_key_of<F64>(F64 c, int j) { return dz_key(c.v, j); }
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)k, o), hi = __shfl_xor((unsigned)(k >> 32), o);
        const unsigned long long t = ((unsigned long long)hi << 32) | lo;
        k = t > k ? t : k;
    }
    return k;
}

template <class S> __device__ void price_scan(const LpView<S> & v, int * sh_i, int * sh_flag,
                                              int & first, int & anypos)
{
    const int rhs = v.rhs, lim = rhs - 1;
    int f = INT_MAX, any = 0;
    if (v.st->pricing == 1) {                                  // Dantzig (non-parity mode, fp64 only)
        unsigned long long key = 0;
        for (int j = threadIdx.x; j < rhs; j += blockDim.x)
            if (v.nv[j] && gt(v.obj[j], zero<S>())) {
                any = 1;
                if (v.rowcnt[j] < lim) { const unsigned long long kj = dz_key_of<S>(v.obj[j], j); key = kj > key ? kj : key; }
            }
        key = wave_max_u64(key);
        unsigned long long * shk = (unsigned long long *)sh_i; // 16 ints = 8 keys: waves in two rounds of 8
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 8) shk[threadIdx.x >> 6] = key;
        __syncthreads();
        unsigned long long best = 0;
        const int nw = blockDim.x >> 6;
        for (int k = 0; k < (nw < 8 ? nw : 8); k++) best = shk[k] > best ? shk[k] : best;
        __syncthreads();
        if (nw > 8) {                                          // waves 8..15 (1024-thread pick)
            if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) >= 8) shk[(threadIdx.x >> 6) - 8] = key;
            __syncthreads();
            for (int k = 0; k < nw - 8; k++) best = shk[k] > best ? shk[k] : best;
            __syncthreads();
        }
        first = dz_col(best);
        if (threadIdx.x == 0) *sh_flag = 0;
        __syncthreads();
        if (any) *sh_flag = 1;
        __syncthreads();
        anypos = *sh_flag;
        return;
    }
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
// Where a pick leaves its decision: the serial loop's LoopState fields or a PipeDesc.
struct PickOut {
    int * status; int * row; int * col; int * leave; int * next_first; int * anypos;
    unsigned long long * cnv_bits; unsigned long long * piv_bits;
};
// `concurrent`: a sweep is rewriting the tableau while this runs, so only the contiguous column
// copies may be read; the one branch that needs other tableau columns (findPivotNVandBVPair) is
// deferred to the next iteration (row = -1, look-ahead carried over), which has no sweep.
// `zero_upto` != nullptr (pipelined loop): the zeroing of basic objective entries below the
// entering index (lpsol.h:1055-1060) and the basis swap are NOT done here; the bound is
// returned and the consumer of the decision applies both (k_pipe_prep / pick_commit), so that
// after k iterations exactly k pivots are visible. Returns true when a pivot was chosen.
template <class S> __device__ void pick_commit(const LpView<S> & v, int r, int enter, int leave)
{
    LoopState * st = v.st;
    v.nv[enter] = 0; v.nv[leave] = 1; v.bv[enter] = 1; v.bv[leave] = 0;   // lpsol.h:1504-1510
    v.eq2bv[r] = enter; v.bv2eq[enter] = r; v.bv2eq[leave] = -1;
    XPG_TRACE_PIVOT("hbm-pick", enter, leave, r);
    const unsigned t = st->total_pivots;
    if ((int)t < v.trace_cap) { v.trace[2 * t] = enter; v.trace[2 * t + 1] = leave; }
    st->total_pivots = t + 1;
    st->done += 1;
}

template <class S> __device__ bool pick_body(const LpView<S> & v, int first, int anypos, int cached_col,
                                             bool b_cached, bool concurrent, const PickOut o,
                                             S * colbuf_out, Cand<S> * sh_c, int * sh_i, int * sh_flag,
                                             int * zero_upto = nullptr, int pass0 = -2)
{
    LoopState * st = v.st;
    if (st->done >= st->max_iter) {                    // while (cnt < m_max_iter), lpsol.h:1039
        __syncthreads();
        if (threadIdx.x == 0) { *o.status = 4; *o.row = -1; }
        return false;
    }
    const int rhs = v.rhs, lim = rhs - 1;
    __syncthreads();
    if (first == NF_UNKNOWN) price_scan(v, sh_i, sh_flag, first, anypos);
    const int stop = first == INT_MAX ? rhs : first;
    if (zero_upto) {
        if (threadIdx.x == 0) *zero_upto = stop;
    } else {
        for (int j = threadIdx.x; j < stop; j += blockDim.x)
            if (!v.nv[j]) v.obj[j] = zero<S>();       // lpsol.h:1055-1060
    }
    __syncthreads();
    int enter = -1, leave = -1;
    if (first == INT_MAX) {
        if (!anypos) {                                 // optimum reached: lpsol.h:1089
            if (threadIdx.x == 0) { *o.status = ST_CHECK_OPT; *o.row = -1; }
            return false;
        }
        if (concurrent) {
            if (threadIdx.x == 0) { *o.row = -1; *o.next_first = INT_MAX; *o.anypos = 1; }
            return false;
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
            if (threadIdx.x == 0) { *o.status = 1; *o.row = -1; }   // SIX_UNBOUND
            return false;
        }
    } else {
        leave = ratio_test(v, first, sh_c, first == cached_col, b_cached, pass0);
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
            price_scan(v, sh_i, sh_flag, nf, any);
            if (threadIdx.x == 0) { *o.row = -1; *o.next_first = nf; *o.anypos = any; }
            return false;
        }
        enter = first;
    }
    // ---- pivot (enter, leave) chosen
    const bool col_cached = enter == cached_col;
    const int r = v.bv2eq[leave];
    __syncthreads();                                   // everyone has read bv2eq[leave]
    for (int i = threadIdx.x; i < v.m; i += blockDim.x)
        colbuf_out[i] = neg(col_cached ? v.nextcol[i] : v.tab[(size_t)i * v.ld + enter]);   // :1485
    if (threadIdx.x == 0) {
        if (!ppt_seen(v, enter, leave)) {              // genPair, lpsol.h:100-104
            v.ppt[(size_t)enter * v.pw + (leave >> 5)] |= 1u << (leave & 31);
            v.rowcnt[enter] += 1; v.colcnt[leave] += 1;
        }
        *o.row = r; *o.col = enter; *o.leave = leave;
        *o.cnv_bits = to_bits(v.obj[enter]);
        *o.piv_bits = to_bits(col_cached ? v.nextcol[r] : v.tab[(size_t)r * v.ld + enter]);
        if (!zero_upto) pick_commit(v, r, enter, leave);
        *o.next_first = INT_MAX; *o.anypos = 0;        // the prep kernel's look-ahead fills these
    }
    return true;
}

template <class S> __global__ __launch_bounds__(1024) void k_pick(LpView<S> v)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<S>)];
    __shared__ int sh_i[16];
    __shared__ int sh_flag;
    LoopState * st = v.st;
    if (st->status != ST_RUNNING) return;
    const PickOut o = { &st->status, &st->row, &st->col, &st->leave, &st->next_first, &st->anypos,
                        &st->cnv_bits, &st->piv_bits };
    const int first = st->next_first, anypos = st->anypos, cached_col = st->cached_col;
    const bool b_cached = st->bcol_valid != 0;
    pick_body<S>(v, first, anypos, cached_col, b_cached, false, o, v.colbuf, (Cand<S> *)sh_c_raw, sh_i, &sh_flag);
}

// Row / column staging, objective update and basis swap for the pivot chosen in
// LoopState (lpsol.h:1468-1474, :1485, :1496-1510). Reads the tableau only.
// guarded: only while the loop is running; counted: the pivot counts towards 'cnt'
// (the forced pivots of phase 1 are neither, lpsol.h:906-908, :939).
// canonical-operand forms of the rational operations (scalar.hip.h), selected where every operand is known canonical
template <class S> __device__ __forceinline__ S mul_c(S a, S b, bool) { return mul(a, b); }
template <> __device__ __forceinline__ R32 mul_c<R32>(R32 a, R32 b, bool canon) { return canon ? mul_canon(a, b) : mul_any_ol(a, b); }
template <class S> __device__ __forceinline__ S add_c(S a, S b, bool) { return add(a, b); }
template <> __device__ __forceinline__ R32 add_c<R32>(R32 a, R32 b, bool canon) { return canon ? add_canon(a, b) : add_any_ol(a, b); }
template <class S> __device__ __forceinline__ S scaled_c(S cell, S x, int mode, bool canon)
{ return mode == SCALE_KEEP ? cell : (mode == SCALE_ZERO ? zero<S>() : mul_c(cell, x, canon)); }
// The objective row's update (lpsol.h:1496-1501): t = e * -1 (negated back beyond rhs), t *= c_nv, obj += t. For
// canonical Rational operands e * -1 is the negation (same magnitude: no gcd, no appro), and scale-then-add is the
// sweep's fused operation (mul and add of canonical operands are symmetric in their operands).
template <class S> __device__ __forceinline__ S obj_update_c(S e, bool beyond_rhs, S cnv, int cmode, S oj, bool canon)
{
    S t = mul_c(e, minus_one<S>(), canon);                     // nvexp.mul(-1), :1496
    if (beyond_rhs) t = neg(t);                                // :1497-1499
    t = scaled_c(t, cnv, cmode, canon);                        // nvexp.mul(tgtf(nv)), :1500
    return add_c(t, oj, canon);                                // addRowToRow, :1501
}
template <> __device__ __forceinline__ R32 obj_update_c<R32>(R32 e, bool beyond_rhs, R32 cnv, int cmode, R32 oj, bool canon)
{
    if (!canon) {
        R32 t = mul_any_ol(e, minus_one<R32>());
        if (beyond_rhs) t = neg(t);
        t = cmode == SCALE_KEEP ? t : (cmode == SCALE_ZERO ? zero<R32>() : mul_any_ol(t, cnv));
        return add_any_ol(t, oj);
    }
    const R32 t = beyond_rhs ? e : neg(e);                     // (e.num == 0: 0/1 either way)
    if (cmode == SCALE_MUL) return fma_canon(oj, cnv, t);
    return add_canon(cmode == SCALE_KEEP ? t : zero<R32>(), oj);
}

template <class S> __global__ __launch_bounds__(256)
void k_prep(LpView<S> v, int guarded, int counted, int bookkeeping, int lookahead)
{
    LoopState * st = v.st;
    if ((guarded && st->status != ST_RUNNING) || st->row < 0) return;
    const bool canon = guarded && !is_f64<S>::value && st->noncanon == 0;
    const int r = st->row, c = st->col;
    const S s = div(one<S>(), from_bits<S>(st->piv_bits));   // 1/(eq.get(eqnum, nv)), :1471
    const int smode = scale_mode(s);
    const S cnv = from_bits<S>(st->cnv_bits);
    const int cmode = scale_mode(cnv);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const int lim = v.rhs - 1;
    int nf = INT_MAX, any = 0;
    for (int j = gid; j < v.W; j += gsz) {
        S e = scaled_c(v.tab[(size_t)r * v.ld + j], s, smode, canon);
        v.rowbuf[j] = e;
        S t = mul_c(e, minus_one<S>(), canon);                 // nvexp.mul(-1), :1496
        if (j >= v.rhs) t = neg(t);                            // :1497-1499
        t = scaled_c(t, cnv, cmode, canon);                    // nvexp.mul(tgtf(nv)), :1500
        const S o = add_c(t, v.obj[j], canon);                 // addRowToRow, :1501
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
        XPG_TRACE_PIVOT("hbm-prep", c, leave, r);
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

// K1, rational: integer-ALU bound (Euclid loops), not HBM. Two things make it cheaper without changing a bit
// (scalar.hip.h, fma_canon): while every cell is canonical -- always, unless the caller's input held fractions
// not in lowest terms (LoopState::noncanon) -- a + k*e is one fused operation on 32-bit gcds, and a column whose
// scaled pivot-row entry is 0 is left alone altogether (a + k*0 = a exactly; in the first pivots of a slack-form
// LP that is about half the columns: the slack columns of rows that have not pivoted yet).
// The grid is (row blocks, column blocks): workgroups go to the 8 XCDs round-robin by linear index, and with the
// column block as the fast index a tableau of 8 column blocks pinned each block to one XCD -- the dead slack half
// then left four XCDs idle (2.9 waves per SIMD on average, VALUBusy 41 %). Rows as the fast index spread every
// column block over all XCDs, and a row still meets the same XCD's L2 on every pivot.
template <int ROWS> __global__ __launch_bounds__(256)
void k_update_r32(LpView<R32> v, int guarded)
{
    const LoopState * st = v.st;
    if ((guarded && st->status != ST_RUNNING) || st->row < 0) return;
    const int r = st->row;
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= v.W) return;
    const int xcol = guarded ? st->next_first : -1;
    const bool ex_col = guarded && j == xcol, ex_b = guarded && j == v.rhs;
    if (guarded && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        v.st->cached_col = (xcol >= 0 && xcol < v.W) ? xcol : -1; v.st->bcol_valid = 1;
    }
    const bool canon = guarded && st->noncanon == 0;          // (one-shot pivots on a caller's tableau: generic)
    const R32 e = v.rowbuf[j];
    const int i0 = blockIdx.x * ROWS;
    if (canon && e.num == 0 && !ex_col && !ex_b) {
        if (r >= i0 && r < i0 + ROWS) v.tab[(size_t)r * v.ld + j] = e;      // the pivot row's own cell := e
        return;
    }
    for (int ii = 0; ii < ROWS; ii++) {
        const int i = i0 + ii;
        if (i >= v.m) break;
        R32 * p = v.tab + (size_t)i * v.ld + j;
        const R32 o = (i == r) ? e : l_fma(canon, *p, v.colbuf[i], e);
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

// ---- Pipelined fp64 loop: two launches per pivot ------------------------------------------------
// Iteration t:  k_pipe_prep(slot = t & 1)  ->  k_pipe_sweep(slot).
// k_pipe_prep stages the scaled pivot row, updates the objective row and prices the NEXT
// iteration (atomicMin into pd[slot].next_first). k_pipe_sweep's grid carries up to 16 extra
// workgroups (blockIdx.y == 0, dispatched first) that commit this iteration's basis swap and
// choose the pivot of iteration t + 1 WHILE the other workgroups sweep. What they need of the
// post-sweep tableau they produce themselves with the sweep's own mul-then-add:
//   * the constant column lives in the contiguous bcol[] for the whole loop (b_i += k_i e_b);
//     the sweep updates the tableau's copy as part of its normal work, bit-identically;
//   * the predicted entering column: the sweep leaves that 16-byte column pair untouched and
//     the pick workgroups update exactly that pair, keeping the column in nextcol[].
// Nothing the sweep reads is written by the pick workgroups and vice versa (pd[slot ^ 1], the
// other half of colbuf, the basis arrays, the pair table); the only data crossing workgroups
// inside the launch are the pick workgroups' 32-byte ratio-test records (pipe_pick_f64 below);
// launch boundaries order everything else. pd[].stop defers a final status by one launch so
// that the sweep in flight completes.
// What the first workgroup does when the descriptor has no pivot to stage: nothing (fp64: the sweep launch's pick
// workgroup runs the generic pick) or that generic pick itself (Rational, lp_pipe_r32.hip.h).
template <class S> __device__ inline void prep_idle(const LpView<S> &, int, int, bool) {}
template <> __device__ inline void prep_idle<R32>(const LpView<R32> & v, int slot, int colstride, bool inplace);
template <class S> __global__ __launch_bounds__(256) void k_pipe_prep(LpView<S> v, int slot, int colstride)
{
    LoopState * st = v.st;
    PipeDesc & D = st->pd[slot];
    // every scalar this kernel branches on is loaded before the first branch: one round trip
    const int status = st->status, pricing = st->pricing;
    const bool canon = !is_f64<S>::value && st->noncanon == 0;  // Rational: the canonical forms (scalar.hip.h)
    const int stop = D.stop, r = D.row, enter = D.col, leave = D.leave, zu = D.zero_upto;
    const unsigned long long piv_bits = D.piv_bits, cnv_bits = D.cnv_bits;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    if (status != ST_RUNNING) return;
    if (stop != 0) {
        // workgroup 0 alone, so that no workgroup can observe the promoted status half-way
        if (blockIdx.x == 0) {
            for (int j = threadIdx.x; j < zu; j += blockDim.x)
                if (!v.nv[j]) v.obj[j] = zero<S>();            // lpsol.h:1055-1060, deferred by the pick
            __syncthreads();
            if (threadIdx.x == 0) st->status = stop;
        }
        return;
    }
    const S * __restrict__ tab = v.tab;
    if (r < 0) {
        if (blockIdx.x == 0) prep_idle<S>(v, slot, colstride, false);
        return;
    }
    S * __restrict__ rowbuf = v.rowbuf;
    S * __restrict__ objout = v.obj;
    if (gid == 0) v.pickrec[PICK_CTR_OFF + 16 * slot] = 0ull;  // arrival counter of this iteration's pick
    const S s = div(one<S>(), from_bits<S>(piv_bits));        // 1/(eq.get(eqnum, nv)), :1471
    const int smode = scale_mode(s);
    const S cnv = from_bits<S>(cnv_bits);
    const int cmode = scale_mode(cnv);
    const int lim = v.rhs - 1;
    const bool dantzig = pricing == 1;                         // opt-in non-parity pricing (N4)
    int nf = INT_MAX, any = 0;
    unsigned long long key = 0;
    for (int j = gid; j < v.W; j += gsz) {
        const S a = tab[(size_t)r * v.ld + j];                 // all four loads in flight together
        S oj = v.obj[j];
        const bool nvj = j < v.rhs && v.nv[j] != 0;            // basis BEFORE this pivot's swap
        const int rcj = v.rowcnt[j < v.rhs ? j : 0];
        S e = scaled_c(a, s, smode, canon);
        rowbuf[j] = e;
        if (j < zu && !nvj) oj = zero<S>();                    // lpsol.h:1055-1060, deferred by the pick
        const S o = obj_update_c(e, j >= v.rhs, cnv, cmode, oj, canon);   // lpsol.h:1496-1501
        objout[j] = o;
        // look-ahead pricing of the next iteration, on the basis AFTER the swap
        const bool nv_next = j == enter ? false : (j == leave ? true : nvj);
        if (j < v.rhs && nv_next && gt(o, zero<S>())) {
            any = 1;
            if (rcj < lim) {
                if (dantzig) { const unsigned long long kj = dz_key_of<S>(o, j); key = kj > key ? kj : key; }
                else nf = min(nf, j);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { nf = min(nf, __shfl_xor(nf, o)); any |= __shfl_xor(any, o); }
    if (dantzig) key = wave_max_u64(key);
    if ((threadIdx.x & 63) == 0) {
        if (nf != INT_MAX) atomicMin(&D.next_first, nf);
        if (key) atomicMax(&D.price_key, key);
        if (any) atomicOr(&D.anypos, 1);
    }
}

// The pick workgroups of k_pipe_sweep: the first PICK_WGS(gridDim.x) workgroups of row 0 of the
// grid, 256 threads each. They are latency-bound (HBM is saturated by the sweep around them: a
// dependent round trip costs ~6 us then), so the work is spread over up to 16 workgroups with
// every load of a row in flight at once, and the common case is ONE fused pass per workgroup
// over its rows: new constant column, new entering column (the pair the sweep skips), -column
// for the next sweep, and the ratio test's first pass on those fresh values. Each workgroup
// publishes its best row as four 8-byte agent-scope atomic stores, one lane adds to an
// agent-scope counter, and the lane whose add came last combines the records and writes the
// next pivot (MI355X_MICROARCH visibility table, row 1: sc1 stores, counter add after the
// storing lane's vmcnt(0), sc1 loads by the last adder). Bulk data never crosses workgroups
// inside the launch. Anything but "first pass found a row" (relaxed second pass, disableNV,
// findPivotNVandBVPair) is deferred by one iteration: row = -1 with the look-ahead carried
// over, and the next launch, which has nothing to sweep, runs the generic single-workgroup
// pick_body on a quiescent tableau.
__device__ __forceinline__ int pick_wgs(int strips) { return strips < PICK_MAX_WGS ? strips : PICK_MAX_WGS; }

__device__ inline void write_desc(PipeDesc & O, int row, int col, int leave, int next_first, int anypos, int stop,
                                  int cached_col, int zero_upto, unsigned done_after, unsigned total_after,
                                  unsigned long long cnv_bits, unsigned long long piv_bits)
{
    O.row = row; O.col = col; O.leave = leave; O.next_first = next_first; O.anypos = anypos; O.stop = stop;
    O.cached_col = cached_col; O.bcol_valid = 1; O.zero_upto = zero_upto; O.done_after = done_after;
    O.total_after = total_after; O.cnv_bits = cnv_bits; O.piv_bits = piv_bits; O.price_key = 0ull;
}
// The entering column a descriptor predicts: the Dantzig key when the prep kernel priced that way,
// else next_first (the reference's rule, or whatever a pick carried over).
__device__ __forceinline__ int desc_first(const PipeDesc & D)
{ const unsigned long long key = D.price_key; return key ? dz_col(key) : D.next_first; }

__device__ inline void pipe_pick_f64(const LpView<F64> & v, int slot, int colstride, int p, int N)
{
    __shared__ __attribute__((aligned(8))) unsigned char sh_c_raw[16 * sizeof(Cand<F64>)];
    __shared__ int sh_i[16];
    __shared__ int sh_flag;
    Cand<F64> * sh_c = (Cand<F64> *)sh_c_raw;
    LoopState * st = v.st;
    PipeDesc & I = st->pd[slot];
    PipeDesc & O = st->pd[slot ^ 1];
    const int r = I.row, ienter = I.col, ileave = I.leave, first = desc_first(I), anypos = I.anypos;
    const unsigned done_now = I.done_after, total_now = I.total_after, max_iter = st->max_iter;
    const int W = v.W, rhs = v.rhs, ld = v.ld, m = v.m, lim = v.rhs - 1;
    double * __restrict__ tab = (double *)v.tab;
    double * __restrict__ nextcol = (double *)v.nextcol;
    double * __restrict__ bcol = (double *)v.bcol;
    double * __restrict__ cbo = (double *)v.colbuf + (size_t)(slot ^ 1) * colstride;
    const int tid = threadIdx.x;

    if (r < 0) {
        // ---- nothing is being swept: one workgroup, generic code, tableau quiescent
        if (p != 0) return;
        const int izero = I.zero_upto, cached_col = I.cached_col;
        const bool b_was_cached = I.bcol_valid != 0;
        for (int j = tid; j < izero; j += 256)
            if (!v.nv[j]) v.obj[j] = zero<F64>();                              // deferred lpsol.h:1055-1060
        if (!b_was_cached)
            for (int i = tid; i < m; i += 256) bcol[i] = tab[(size_t)i * ld + rhs];
        if (tid == 0)
            write_desc(O, -1, 0, 0, INT_MAX, 0, 0, cached_col, 0, done_now, total_now, 0ull, 0ull);
        __threadfence_block();
        __syncthreads();
        if (done_now >= max_iter) {                    // while (cnt < m_max_iter), lpsol.h:1039
            if (tid == 0) O.stop = 4;
            return;
        }
        const PickOut o = { &O.stop, &O.row, &O.col, &O.leave, &O.next_first, &O.anypos, &O.cnv_bits, &O.piv_bits };
        const bool chosen = pick_body<F64>(v, first, anypos, cached_col, true, false, o, (F64 *)cbo, sh_c, sh_i,
                                           &sh_flag, &O.zero_upto);
        if (chosen && tid == 0) { O.done_after = done_now + 1; O.total_after = total_now + 1; }
        return;
    }

    // ---- a sweep is running around us
    if (p == 0 && tid == 0) {                          // commit this iteration's pivot (lpsol.h:1504-1510)
        v.nv[ienter] = 0; v.nv[ileave] = 1; v.bv[ienter] = 1; v.bv[ileave] = 0;
        v.eq2bv[r] = ienter; v.bv2eq[ienter] = r; v.bv2eq[ileave] = -1;
        const unsigned t = total_now - 1;
        if ((int)t < v.trace_cap) { v.trace[2 * t] = ienter; v.trace[2 * t + 1] = ileave; }
        st->total_pivots = total_now; st->done = done_now;
    }
    const int xc = (first >= 0 && first < W) ? first : -1;
    const double * __restrict__ cb = (const double *)v.colbuf + (size_t)slot * colstride;
    const double * __restrict__ rb = (const double *)v.rowbuf;
    const double eb = rb[rhs];
    const int stride = 256 * N;                        // rows are dealt to the workgroups in blocks of 256
    if (xc < 0 || done_now >= max_iter) {
        // no ratio test this time: keep the columns current, workgroup 0 records the outcome
        if (xc >= 0) {
            const int pc = xc & ~1;
            const bool wide = pc + 1 < W;
            const double e0 = rb[pc], e1 = wide ? rb[pc + 1] : 0.0;
            for (int i = p * 256 + tid; i < m; i += stride) {
                double * q = tab + (size_t)i * ld + pc;
                const double k = cb[i];
                double n0 = q[0] + k * e0, n1 = wide ? q[1] + k * e1 : 0.0;
                if (i == r) { n0 = e0; n1 = e1; }
                q[0] = n0; if (wide) q[1] = n1;
                nextcol[i] = (xc & 1) ? n1 : n0;
            }
        }
        for (int i = p * 256 + tid; i < m; i += stride) {
            const double q = cb[i] * eb;
            bcol[i] = (i == r) ? eb : (bcol[i] + q);
        }
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

    // ---- fused pass over this workgroup's rows
    const int pc = xc & ~1;
    const bool wide = pc + 1 < W, odd = (xc & 1) != 0;
    const double e0 = rb[pc], e1 = wide ? rb[pc + 1] : 0.0;
    unsigned long long cnv_bits = 0; int rc_enter = 0;
    if (tid == 0) { cnv_bits = to_bits(v.obj[xc]); rc_enter = v.rowcnt[xc]; }   // for the last adder's tail
    constexpr int U = 2;
    Cand<F64> best; best.q = zero<F64>(); best.idx = INT_MAX;
    double best_a = 0.0; int best_b = 0, best_cc = 0; uint32_t best_w = 0;
    bool weird = false;                                // a NaN ratio: only the reference's scan order decides (unordered_value)
    for (int i0 = p * 256 + tid; i0 < m; i0 += stride * U) {
        double k[U], bo[U], c0[U], c1[U];
        int bi[U], cc[U]; uint32_t w[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(i0 + u * stride, m - 1);
            const double * q = tab + (size_t)i * ld + pc;
            k[u] = cb[i]; bo[u] = bcol[i];
            if (wide) { const double2 t = *reinterpret_cast<const double2 *>(q); c0[u] = t.x; c1[u] = t.y; }
            else { c0[u] = q[0]; c1[u] = 0.0; }
            bi[u] = v.eq2bv[i];
            if (i == r) bi[u] = ienter;                // the commit above, seen without waiting for it
        }
#pragma unroll
        for (int u = 0; u < U; u++) { w[u] = v.ppt[(size_t)xc * v.pw + (bi[u] >> 5)]; cc[u] = v.colcnt[bi[u]]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * stride;
            if (i >= m) continue;
            double * q = tab + (size_t)i * ld + pc;
            const double qb = k[u] * eb, s0 = k[u] * e0, s1 = k[u] * e1;
            double nb = bo[u] + qb, n0 = c0[u] + s0, n1 = c1[u] + s1;         // the sweep's a + k*e
            if (i == r) { nb = eb; n0 = e0; n1 = e1; }
            bcol[i] = nb;
            if (wide) { double2 t; t.x = n0; t.y = n1; *reinterpret_cast<double2 *>(q) = t; }
            else q[0] = n0;
            const double a = odd ? n1 : n0;
            nextcol[i] = a;
            cbo[i] = -a;                                                      // -column, lpsol.h:1485
            if (le(F64(a), zero<F64>())) continue;                            // findPivotBV, lpsol.h:553-663
            if (((w[u] >> (bi[u] & 31)) & 1u) || cc[u] >= lim) continue;
            Cand<F64> c; c.q = div(F64(nb), F64(a)); c.idx = i;
            weird |= c.q.v != c.q.v;
            const Cand<F64> nbest = better(best, c);
            if (nbest.idx != best.idx) { best_a = a; best_b = bi[u]; best_cc = cc[u]; best_w = w[u]; }
            best = nbest;
        }
    }
    const Cand<F64> wbest = block_argmin(best, sh_c);
    const int wg_weird = __syncthreads_or(weird ? 1 : 0);
    // one lane publishes this workgroup's record: the owner of the winning row, else lane 0
    const bool publisher = wbest.idx != INT_MAX ? (best.idx == wbest.idx) : (tid == 0);
    __shared__ unsigned long long sh_cnv; __shared__ int sh_rc;
    if (tid == 0) { sh_cnv = cnv_bits; sh_rc = rc_enter; }
    __syncthreads();
    if (!publisher) return;
    unsigned long long * rec = v.pickrec + (size_t)p * PICK_REC_WORDS;
    unsigned long long * ctr = v.pickrec + PICK_CTR_OFF + 16 * slot;
    __hip_atomic_store(rec + 0, to_bits(wbest.q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 1, to_bits(F64(best_a)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 2, ((unsigned long long)(unsigned)wbest.idx << 32) | (unsigned)best_b, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 3, ((unsigned long long)best_w << 32) | (unsigned)best_cc | (wg_weird ? 0x80000000u : 0u), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long arrived = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived != (unsigned long long)(N - 1)) return;
    // ---- last adder: combine the records in workgroup order (ties: lowest row, lpsol.h:604-611)
    Cand<F64> g; g.q = zero<F64>(); g.idx = INT_MAX;
    double g_a = 0.0; int g_b = 0, g_cc = 0; uint32_t g_w = 0;
    bool any_weird = false;
    for (int k = 0; k < N; k++) {
        const unsigned long long * rk = v.pickrec + (size_t)k * PICK_REC_WORDS;
        const unsigned long long w0 = __hip_atomic_load(rk + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w1 = __hip_atomic_load(rk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w2 = __hip_atomic_load(rk + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long w3 = __hip_atomic_load(rk + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Cand<F64> c; c.q = from_bits<F64>(w0); c.idx = (int)(unsigned)(w2 >> 32);
        any_weird |= ((unsigned)w3 & 0x80000000u) != 0u;
        const Cand<F64> ng = better(g, c);
        if (ng.idx != g.idx) { g_a = from_bits<F64>(w1).v; g_b = (int)(unsigned)w2; g_w = (uint32_t)(w3 >> 32); g_cc = (int)((unsigned)w3 & 0x7fffffffu); }
        g = ng;
    }
    if (g.idx == INT_MAX || any_weird) {               // first pass empty (second pass / disableNV), or NaN ratios: next launch, generic pick
        write_desc(O, -1, 0, 0, first, anypos, 0, xc, 0, done_now, total_now, 0ull, 0ull);
        return;
    }
    const int enter = xc, leave = g_b;
    if (!((g_w >> (leave & 31)) & 1u)) {               // genPair, lpsol.h:100-104
        v.ppt[(size_t)enter * v.pw + (leave >> 5)] = g_w | (1u << (leave & 31));
        v.rowcnt[enter] = sh_rc + 1; v.colcnt[leave] = g_cc + 1;
    }
    write_desc(O, g.idx, enter, leave, INT_MAX, 0, 0, xc, first, done_now + 1, total_now + 1, sh_cnv,
               to_bits(F64(g_a)));
}

// A/B aid (XPG_LOOP=split): the pick as a launch of its own after a sweep launched without it.
template <int UNUSED = 0> __global__ __launch_bounds__(256) void k_pipe_pick(LpView<F64> v, int slot, int colstride)
{
    if (v.st->status != ST_RUNNING) return;
    pipe_pick_f64(v, slot, colstride, blockIdx.x, gridDim.x);
}

template <int ROWS, int UNROLL> __global__ __launch_bounds__(256)
void k_pipe_sweep(LpView<F64> v, int slot, int colstride, int with_pick,
                  double * __restrict__ tab, const double * __restrict__ rowbuf,
                  const double * __restrict__ colbuf, int descending)
{
    // descending (XPG_ZIGZAG=1, off by default): tiles visited in the reverse order of the previous
    // sweep, hoping to start on lines still in the 256 MiB Infinity Cache. Measured on MI355X:
    // 79.7 us per sweep against 77.8 us in fixed order, so the cache does not retain them that way.
    // tab / rowbuf / colbuf (this slot's half) repeat v's pointers as restrict PARAMETERS: only then
    // does -a_i,nv arrive through the scalar cache (restrict on a local is not enough; measured
    // 84.5 us against 73 us per sweep).
    LoopState * st = v.st;
    const PipeDesc & D = st->pd[slot];
    const int status = st->status, r = D.row, first = desc_first(D);    // one round trip, then branch
    if (status != ST_RUNNING) return;
    if (blockIdx.y == 0) {
        const int N = pick_wgs(gridDim.x);
        if ((int)blockIdx.x < N && with_pick) pipe_pick_f64(v, slot, colstride, blockIdx.x, N);
        return;
    }
    if (r < 0) return;
    const int W = v.W, ld = v.ld, m = v.m;
    const int bx = descending ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int by = descending ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y - 1;
    const int j = bx * 512 + threadIdx.x * 2;
    if (j >= W) return;
    const int xc = (first >= 0 && first < W) ? first : -1;
    if (xc >= 0 && j == (xc & ~1)) return;                    // the pick workgroup's pair
    const int i0 = by * ROWS;
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
            base += ld;
        }
    } else {                                                  // odd last column
        const double e = rowbuf[j];
        for (int i = i0; i < iend; i++) {
            double * p = tab + (size_t)i * ld + j;
            const double q = colbuf[i] * e;
            *p = (i == r) ? e : (*p + q);
        }
    }
}

// SIX::is_feasible's two tests (lpsol.h:798-816). tol == 0: the reference's own comparisons (Float's
// 1e-17 '=='); tol > 0 (opt-in non-parity mode, fp64 only): relative tolerance.
template <class S> __device__ __forceinline__ bool exceeds(S lhs, S rhs, double) { return gt(lhs, rhs); }
template <> __device__ __forceinline__ bool exceeds<F64>(F64 lhs, F64 rhs, double tol)
{ return tol > 0.0 ? lhs.v > rhs.v + tol * fmax(1.0, fabs(rhs.v)) : gt(lhs, rhs); }
template <class S> __device__ __forceinline__ bool differs(S a, S b, double) { return ne(a, b); }
template <> __device__ __forceinline__ bool differs<F64>(F64 a, F64 b, double tol)
{ return tol > 0.0 ? fabs(a.v - b.v) > tol * fmax(1.0, fabs(b.v)) : ne(a, b); }

// ---- optimum: solution read-out + SIX::is_feasible (lpsol.h:1104-1110, :784-822)
template <class S> __global__ void k_solution(LpView<S> v)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int j = gid; j < v.W; j += gsz) {
        S x = zero<S>();
        if (j < v.rhs && v.bv[j]) x = v.tab[(size_t)v.bv2eq[j] * v.ld + v.rhs];
        v.x[j] = x;
        if (j < v.rhs && exceeds(mul(v.vcd[j], x), v.vcr[j], v.st->feas_tol)) v.st->infeasible = 1;
    }
}
// One thread per row; the sum runs over j ascending exactly as the reference
// does, skipping nonbasic j whose x_j is an exact zero where that term cannot change the sum: a * 0 is a zero for
// a finite Float a (and an inf or NaN a -- what convertEq2Ineq's division by a zero entry leaves in a tableau -- makes
// it NaN, which the reference's sum then carries into "optimal but infeasible": those terms stay); Rational's
// a * 0 is 0/1 whatever a holds, and sum + 0/1 re-squeezes a sum that every earlier step already squeezed.
__device__ __forceinline__ bool rowcheck_term_is_noop(F64 a) { return fabs(a.v) <= 1.7976931348623157e308; }
__device__ __forceinline__ bool rowcheck_term_is_noop(R32) { return true; }
template <class S> __global__ void k_rowcheck(LpView<S> v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.m) return;
    S sum = zero<S>();
    const S * row = v.tab + (size_t)i * v.ld;
    for (int j = 0; j < v.rhs; j++)
        if (v.bv[j] || !rowcheck_term_is_noop(row[j])) sum = add(sum, mul(row[j], v.x[j]));
    reduce(sum);
    S b = row[v.rhs];
    reduce(b);
    v.tab[(size_t)i * v.ld + v.rhs] = b;                       // lc.reduce(i, rhs_idx) writes back
    if (differs(sum, b, v.st->feas_tol)) v.st->infeasible = 1;
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
// (Float: a finite cell. A tableau that starts with an inf or a NaN meets NaN ratios, which only the reference's scan order
// decides -- unordered_value above; the blocked loop's record hand-offs have no such path, so the host keeps an LP whose
// input is not finite on the pipelined loop, whose picks defer those ratio tests to the generic pick.)
__device__ __forceinline__ bool is_canonical_cell(F64 a) { return fabs(a.v) <= 1.7976931348623157e308; }
__device__ __forceinline__ bool is_canonical_cell(R32 a) { return canonical(a); }
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
            if (!is_canonical_cell(val)) v.st->noncanon = 1;
        } else {
            // the caller's objective comes back after phase one (k_rebuild_obj copies tgtf, its constant unreduced,
            // lpsol.h:944-953), so its cells count for the canonical forms whichever objective goes in now
            const S tv = j < n ? tgtf[j] : (j == v.rhs ? tgtf[n] : zero<S>());
            if (with_xa) { if (j == n) val = minus_one<S>(); }
            else val = tv;
            v.obj[j] = val;
            if (!is_canonical_cell(tv)) v.st->noncanon = 1;
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
template <class S> __global__ void k_reset_loop(LpView<S> v, unsigned max_iter, int pricing, double feas_tol)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int i = gid; i < v.rhs; i += gsz) { v.rowcnt[i] = 0; v.colcnt[i] = 0; }
    const size_t words = (size_t)v.rhs * v.pw;
    for (size_t t = gid; t < words; t += gsz) v.ppt[t] = 0u;
    if (v.blkR) for (int t = gid; t < BLK_REC_TOTAL_WORDS; t += gsz) v.blkR[t] = 0ull;        // records and partials: no tag of an earlier solve survives
    if (v.blkP) for (int t = gid; t < BLK_PART_TOTAL_INTS; t += gsz) v.blkP[t] = 0;
    if (gid == 0) {
        LoopState * st = v.st;
        st->status = ST_RUNNING; st->done = 0; st->max_iter = max_iter;
        st->row = -1; st->infeasible = 0;
        st->next_first = NF_UNKNOWN; st->anypos = 0; st->cached_col = -1; st->bcol_valid = 0;
        st->pricing = pricing; st->r32_side = 0; st->r32_idle = 0u; st->feas_tol = feas_tol;
        st->blk.batch = -1; st->blk.n = 0; st->blk.closed = 0; st->blk.generic = 0; st->blk.from_generic = 0;
        st->blk.budget = 0xFFFFFFFFu; st->blk.price_key = 0ull;
        st->blk.want_generic = 0; st->blk.la_from_state = 1; st->blk.la_epoch = 0u;
        st->blk.sweeps_full = 0u; st->blk.sweeps_part = 0u;
        st->blk.ch_epoch = 0u; st->blk.ch_budget = 0u; st->blk.ch_done = 0u; st->blk.ch_tp = 0u;
        for (int x = 0; x < 8; x++) st->blk.ch_arrive[x] = 0u;
        st->blk.ch_aborts = 0u; st->blk.ch_runs = 0u; st->blk.ch_misplaced = 0u; st->blk.ch_decide = 0u;
        st->blk.ch0_ticket = 0u; st->blk.ch0_la_epoch = 0u; st->blk.ch0_budget = 0u; st->blk.ch0_done = 0u; st->blk.ch0_tp = 0u; st->blk.ch_folds = 0u;
        for (int k = 0; k < 8; k++) st->blk.dbg[k] = 0ull;
        for (int k = 0; k < 2; k++) v.pickrec[PICK_CTR_OFF + 16 * k] = 0ull;   // arrival counters
        for (int k = 0; k < 2; k++) {
            unsigned long long * go = v.pickrec + PICK_GO_OFF + PICK_GO_WORDS * k;
            go[GO_ARRIVED] = 0ull; go[GO_NF] = (unsigned long long)(unsigned)INT_MAX; go[GO_ANY] = 0ull;
        }
        for (int k = 0; k < PICK_MAX_WGS * PICK_FREC_WORDS; k++) v.pickrec[PICK_FREC_OFF + k] = 0ull;
        for (int k = 0; k < 2; k++) {                  // pipelined loop: iteration 0 has no pivot yet
            PipeDesc & D = st->pd[k];
            D.row = -1; D.col = 0; D.leave = 0; D.next_first = NF_UNKNOWN; D.anypos = 0; D.stop = 0;
            D.cached_col = -1; D.bcol_valid = 0; D.zero_upto = 0; D.pad_ = 0; D.side = 0; D.staged = 0; D.cnv_bits = 0; D.piv_bits = 0;
            D.done_after = 0; D.total_after = st->total_pivots; D.price_key = 0ull;
        }
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
    bool weird = false;
    for (int i = threadIdx.x; i < v.m; i += blockDim.x) weird |= unordered_value(v.tab[(size_t)i * v.ld + v.rhs]);
    if (__builtin_expect(__syncthreads_or(weird ? 1 : 0), 0)) {  // lpsol.h:894-904 as written: row = 0; if (b[row] > b[i]) row = i
        __shared__ int sh_seq;
        if (threadIdx.x < 64) {
            int sbest = INT_MAX; S sq = zero<S>();
            for (int base = 0; base < v.m; base += 64) {
                const int i = min(base + (int)threadIdx.x, v.m - 1);
                scan_step_in_order(v.tab[(size_t)i * v.ld + v.rhs], base + (int)threadIdx.x < v.m, base, sbest, sq);
            }
            if (threadIdx.x == 0) sh_seq = sbest;
        }
        __syncthreads();
        best.idx = sh_seq;
    }
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
