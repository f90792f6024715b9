// Blocked fp64 simplex loop, stages 1 .. B-1 of a batch in ONE launch.
//
// The launch-per-stage chain of lp_blocked.hip.h (pick(t) -> prep(t) -> pick(t+1) ...) is a latency
// chain: measured at 4096 x 8192 (tools/lab/probe_stamps.py) a pick or prep launch lasts 5.8-6.0 us of
// which ~2 us each are its two dependent memory rounds -- first touches of a fresh launch (cold L2,
// cold TLBs) -- and a launch boundary. The same chain step as a phase of a persistent launch costs
// 2.2-2.8 us in the lab (tools/lab/launch_lab.hip, rows D against B). This kernel is that persistent form.
//
// Workers are one-wave workgroups: worker w owns rows 64w .. 64w+63 (pick role, w < npick) and columns
// 64w .. 64w+63 (prep role, w < nprep); one more worker, the committer, owns the basis arrays. There is
// NO barrier and NO store drain on the path: every hand-off is a set of 16-byte granules {data, tag}, each
// written by one sc1 store and polled by its consumers with sc1 loads (MI355X_MICROARCH.md, "Valid forms":
// a granule needs no ordering), and whatever a stage needs FRESH from the stage before it travels inside
// those granules:
//   partial of prep worker w, stage t   g0 {lowest eligible column of its 64 (or INT_MAX), any c_j > 0}
//                                       g1 {e_t[that column]}   g2 {objective entry of that column}
//                                       g3 {e_t[rhs]}  (only the worker that owns the constant column)
//   record of pick worker w, stage t    g0 {ratio key, row}: the ONLY granule everybody polls (one 16-byte load per
//                                       record and lane; six were measured to cost 2.6 us per hand-off in L2
//                                       contention: 129 workers x 64 records x 6 loads on 64 lines)
//                                       g1 {pivot element, leaving}  g2 {pair word, counter, entering | 1 + last
//                                       stage this row pivoted in}  g3 {c_nv}: fetched of the WINNER only, in the
//                                       same round as the pivot row gather
//   commit granule, stage t             written by the committer behind its own drain (off the path)
// Everything else a stage reads from memory was stored at least one full stage earlier by a wave that has
// since completed a poll (whose s_waitcnt vmcnt(0) also waits for that wave's own stores) and then published
// a granule the reader has seen: e_s[first] for s <= t-2, k_s[r] for s <= t-1, and the basis words behind
// the commit granule. Own data never leaves registers: a lane keeps k_s[i] of its row, e_s[j] of its
// column, its objective entry, its row's basic variable, the replayed constant column of its row (ONE
// update per stage: the constant column is the same column every stage) and the last stage its row
// pivoted in (which replaces the list of pivot rows in both replays).
//   stage t, pick role:  poll the partials of stage t-1 and the commit granule -> entering column `first`;
//                        one round: the column gather, e_s[first] (lane s), the three fresh granules, pair
//                        word, counter; replay (the sweep's own mul-then-add per stage), ratio test
//                        (lpsol.h:553-663) as a 64-bit key whose unsigned order is better()'s, DPP min;
//                        -a_i,nv -> K; the winning lane publishes the record
//   stage t, prep role:  poll the records (lane l polls record l), combine; one round: the pivot row
//                        gather, k_s[r] (lane s), nv[j], rowcnt[j]; replay, scaled row -> E, objective row,
//                        look-ahead pricing (lpsol.h:1054-1069), DPP min; partial out
//   committer:           polls the records like everyone, commits the pivot exactly as blk_prep_body does
// What a worker reads of the committed state at entry comes from fields this launch never writes (the
// ticket blk.ch_* left by stage 0's prep launch), so a worker that starts late sees what the early ones saw.
// Anything but "fast pick found a row" ends the launch for every worker alike: the pick role publishes
// CLOSE records instead (budget spent, iteration limit, no eligible column), an empty first ratio pass is
// seen by everyone in the records. The batch then has fewer than B staged pivots, the sweep applies them,
// and the next batch starts with the launch-per-stage kernels, which own every rare branch.
// The protocol needs every worker RESIDENT at once (they poll each other). That is the rule when the handle has the
// device to itself, and not guaranteed otherwise -- another stream of the host application or another process can hold
// CUs for milliseconds. So a launch starts with a roll call: every worker counts itself in (one atomic add), the
// committer waits for the count to reach the grid size and publishes GO in a decision granule; if the count is not
// complete within CH_ARRIVE_TICKS (0.3 ms; a 200-workgroup grid starts within a microsecond on a free chip) it
// publishes ABORT instead: every worker -- those present now and those that get a CU later -- leaves WITHOUT having
// written any state, the batch is closed with the one pivot stage 0 staged, the sweep applies it, and the host, which
// sees blk.ch_aborts move at its next status read, enqueues this LP's further batches as launch-per-stage kernels
// (which need no co-residency). After GO every worker is resident and stays so, and a poll that still does not
// complete within seconds flags ST_CHAIN_STUCK (XPG_ERR_CHAIN_STUCK for the caller): a preempted queue, not a state
// this code recovers from.
#pragma once
#include "lp_blocked.hip.h"

namespace xpg {

enum { ST_CHAIN_STUCK = XPG_ERR_CHAIN_STUCK, CH_SPIN_LIMIT = 1 << 21, CH_CLOSE = 0x7FFFFFFF,
       CH_ARRIVE_TICKS = 30000,    // roll call: 0.3 ms of the 100 MHz clock
       CH_VETO_TICKS = 100000,     // a worker that has waited 1 ms for the verdict closes the roll call itself
       CH_GO = 1, CH_ABORT = 2 };
typedef unsigned int ch_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long ch_lo64(ch_u32x4 g) { return ((unsigned long long)g.y << 32) | g.x; }

// all six granules of one record in one round (the asm block waits for its own loads -- and, vmcnt being one
// counter, for every store this wave issued before)
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
// four granules from four addresses in one round
__device__ __forceinline__ void ch_load4(const void * p0, const void * p1, const void * p2, const void * p3,
                                         ch_u32x4 & g0, ch_u32x4 & g1, ch_u32x4 & g2, ch_u32x4 & g3)
{
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ void ch_load3(const void * p0, const void * p1, const void * p2, ch_u32x4 & g0, ch_u32x4 & g1, ch_u32x4 & g2)
{
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(g0), "=&v"(g1), "=&v"(g2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}
__device__ __forceinline__ ch_u32x4 ch_load1(const void * p)
{
    ch_u32x4 g;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
    return g;
}
// One 16-byte granule {data, tag} by one store. The s_nop belongs to the store: a VMEM store of more than 8
// bytes reads its data registers after issue, the hardware does not interlock a VALU write to them in the next
// cycles, and the compiler's hazard recogniser cannot see a store inside inline asm (seen without it: the next
// granule's tag move landed in this granule's data).
// LOCAL (every worker of the launch on ONE XCD, see k_blk_chain): a PLAIN store -- it writes through the CU's L1 into the
// XCD's L2 and stays there, where the consumers' sc1 loads (which bypass their own L1 only) find it at L2 latency. An sc1
// store drops the line from the L2 and every reader pays the fabric again (tools/lab/xcd_handoff_lab.hip: an
// all-to-all step of 129 workers 2.90 us spread / sc1, 3.74 us on one XCD / sc1, 1.85 us on one XCD / plain).
template <bool LOCAL> __device__ __forceinline__ void ch_store_u32x4(void * p, ch_u32x4 g)
{
    if (LOCAL) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 3" :: "v"(p), "v"(g) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" :: "v"(p), "v"(g) : "memory");
}
template <bool LOCAL> __device__ __forceinline__ void ch_store_granule(void * p, unsigned long long lo, unsigned long long hi)
{
    ch_u32x4 g;
    g.x = (unsigned)lo; g.y = (unsigned)(lo >> 32); g.z = (unsigned)hi; g.w = (unsigned)(hi >> 32);
    ch_store_u32x4<LOCAL>(p, g);
}
template <bool LOCAL> __device__ __forceinline__ void ch_store_granule3(void * p, unsigned x, unsigned y, unsigned z, unsigned tag)
{
    ch_u32x4 g;
    g.x = x; g.y = y; g.z = z; g.w = tag;
    ch_store_u32x4<LOCAL>(p, g);
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
// LOCAL: a plain store (kept in the XCD's L2, see ch_store_u32x4); the readers are sc1 loads on the same XCD
template <class T> __device__ __forceinline__ void ch_st_plain(T * p, T x)
{
    if constexpr (sizeof(T) == 1) {
        unsigned char u; __builtin_memcpy(&u, &x, 1);
        __hip_atomic_store((unsigned char *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else if constexpr (sizeof(T) == 4) {
        unsigned u; __builtin_memcpy(&u, &x, 4);
        __hip_atomic_store((unsigned *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
        static_assert(sizeof(T) == 8, "1-, 4- or 8-byte objects");
        unsigned long long u; __builtin_memcpy(&u, &x, 8);
        __hip_atomic_store((unsigned long long *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
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
template <bool LOCAL, class T> __device__ __forceinline__ void ch_stl(T * p, T x) { if (LOCAL) ch_st_plain(p, x); else ch_st(p, x); }
__device__ __forceinline__ int ch_xcc_id()                               // which of the 8 XCDs this wave runs on
{
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x & 7;
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
// The ratio as a key whose UNSIGNED order is the order better() puts ratios in: -0 and +0 are one value,
// then the usual monotone map of IEEE doubles. ~0 is "no candidate".
__device__ __forceinline__ unsigned long long ch_ratio_key(double q)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, q + 0.0);   // (-0) + (+0) = +0
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
template <int CTRL> __device__ __forceinline__ unsigned long long ch_min_step(unsigned long long x)
{
    const unsigned lo = (unsigned)dpp_row_shr<CTRL>((int)(unsigned)x), hi = (unsigned)dpp_row_shr<CTRL>((int)(unsigned)(x >> 32));
    const unsigned long long t = ((unsigned long long)hi << 32) | lo;
    return t < x ? t : x;
}
__device__ __forceinline__ unsigned long long ch_wave_min_u64(unsigned long long x)
{
    x = ch_min_step<0x111>(x); x = ch_min_step<0x112>(x); x = ch_min_step<0x114>(x); x = ch_min_step<0x118>(x);
    const unsigned long long a = ch_readlane_u64(x, 15), b = ch_readlane_u64(x, 31), c = ch_readlane_u64(x, 47), d = ch_readlane_u64(x, 63);
    const unsigned long long ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// Waiting without asking. A role that has handed its result over knows that the answer takes the other role's whole turn:
// a gather round trip, a replay, a wave reduction and a store's way to the L2 -- 1.7 us and more at every shape measured
// (tools/lab/probe_chain_ts.py). It sleeps through the first part of that (s_sleep N = 64 N clocks: 32 is ~0.9 us) before
// its first poll instead of loading the same L2 lines every 30 ns next to the loads of the workers that are busy: +1.0 %
// pivots/s at 4096 x 8192 and 4096 x 4096, +0.6 % at 4096 x 12289; 32 / 48 / 64 within noise of each other, the shortest
// kept. With fewer workers there is less to keep out of the way of: nothing at 2048 x 4096 (97 workers), -0.3 % at
// 1024 x 1536 and 512 x 1024 (tools/lab/probe_nap_sizes.py) -- launches of up to CH_NAP_FROM workers poll at once.
// Between polls: s_sleep 1 (0 the same, 2 -0.2 %, 4 -1 %).
enum { CH_NAP_FROM = 96 };
#ifndef CH_POLL_NAP
#define CH_POLL_NAP 1
#endif
#ifndef CH_NAP_PICK
#define CH_NAP_PICK 32
#endif
#ifndef CH_NAP_PREP
#define CH_NAP_PREP 32
#endif

// A replay: a := start, then a := a + h_s * c_s for the stages s = 0 .. t-1 IN ORDER, each product rounded, each sum
// rounded -- the sweep's own two roundings per pending update. h_s is this lane's history, c_s the other operand,
// wave-uniform, held by lane s of `cv`. Round 5 (tools/lab/probe_chain_ts.py: the replays were 47 ns per step and stage,
// a fifth of the chain launch -- a rolled loop with the stage number in an SGPR, an LDS round trip per group of four
// and a select between "replace" and "add" inside the dependent chain):
//   * the history is read from LDS into REGISTERS at the top of the stage (ChHist: all BLK_MAX slots, static indices
//     only), long before the gather it is combined with has arrived -- no LDS wait inside the replay;
//   * the products do not depend on each other, only the sums do: the loop is unrolled (the lane numbers of the
//     broadcasts become immediates) and the ONLY dependent chain is t additions;
//   * a stage that must not count adds -0.0 instead of its product, which leaves every double as it is (x + -0.0 = x for
//     x = +-0, denormals, infinities; NaNs stay NaNs). Beyond t that costs nothing: the history slots at and beyond t
//     hold +0.0 (zeroed at the start of the launch, slot s is written in stage s) and the lanes at and beyond t of `cv`
//     hold -1.0, so the product IS -0.0. At or before `star` -- the last stage in which this element's row was the pivot
//     row and the element was REPLACED; `start` is then that stage's value -- it is a select on the product, off the chain.
// STAR: false no element of this wave was replaced, true `star` counts (per lane or wave-uniform alike).
struct ChHist { double h[BLK_MAX]; };
__device__ __forceinline__ void ch_hist_load(ChHist & H, const double * hist)
{
#pragma unroll
    for (int s = 0; s < BLK_MAX; s++) H.h[s] = hist[64 * s];
}
template <bool STAR> __device__ __forceinline__ double ch_replay(double start, const ChHist & H, double cv, int t, int star)
{
    double a = start;
#pragma unroll
    for (int s = 0; s < BLK_MAX; s += 4) {
        if (s >= t) break;
        double p0 = H.h[s] * ch_readlane_f64(cv, s), p1 = H.h[s + 1] * ch_readlane_f64(cv, s + 1);
        double p2 = H.h[s + 2] * ch_readlane_f64(cv, s + 2), p3 = H.h[s + 3] * ch_readlane_f64(cv, s + 3);
        if (STAR) {
            const double ident = -0.0;
            p0 = s > star ? p0 : ident; p1 = s + 1 > star ? p1 : ident; p2 = s + 2 > star ? p2 : ident; p3 = s + 3 > star ? p3 : ident;
        }
        a = a + p0; a = a + p1; a = a + p2; a = a + p3;
    }
    return a;
}

// The winner of a stage, as every worker derives it from the records.
struct ChWinner { int r, enter, leave, qstar, cc; uint32_t w; double a; unsigned long long cnv; };
enum { CH_CLOSE_ROW = 0x7FFFFFFE, CH_UNORDERED_ROW = 0x7FFFFFFD };

// addresses in the hand-off areas (layout: lp_kernels.hip.h)
__device__ __forceinline__ char * ch_rec_g0(const LpView<F64> & v, int w) { return (char *)(v.blkR + BLK_REC_G0) + (size_t)w * 16; }
__device__ __forceinline__ char * ch_rec_pay(const LpView<F64> & v, int w) { return (char *)(v.blkR + BLK_REC_PAY) + (size_t)w * (BLK_REC_PAY_WORDS * 8); }
__device__ __forceinline__ char * ch_part_g0(const LpView<F64> & v, int w) { return (char *)v.blkP + (size_t)w * 16; }
__device__ __forceinline__ char * ch_part_pay(const LpView<F64> & v, int w) { return (char *)(v.blkP + BLK_PART_PAY) + (size_t)w * (BLK_PART_PAY_INTS * 4); }

// Poll g0 {key, row, tag} of the <= 256 records of stage `tag` (lane l polls l, l + 64, ...: 64 granules = eight lines per
// load) and find the winner. Returns the winner's record index (>= 0), -1 on CLOSE, -2 when the first ratio pass was
// empty everywhere, -3 on a stuck poll; W.r is set.
__device__ __forceinline__ int ch_poll_records(const LpView<F64> & v, int npick, unsigned tag, int lane, ChWinner & W)
{
    const int nu = (npick + 63) >> 6;                           // records per lane, wave-uniform
    const int k0 = lane, k1 = lane + 64, k2 = lane + 128, k3 = lane + 192;
    // (a lane without a record of its own re-reads one it shares with another lane of the same line: no hot spot)
    const void * p0 = ch_rec_g0(v, k0 < npick ? k0 : k0 % npick);
    const void * p1 = ch_rec_g0(v, k1 < npick ? k1 : k0 % npick);
    const void * p2 = ch_rec_g0(v, k2 < npick ? k2 : k0 % npick);
    const void * p3 = ch_rec_g0(v, k3 < npick ? k3 : k0 % npick);
    ch_u32x4 g0, g1, g2, g3;
    unsigned spins = 0;
    for (;;) {
        bool ok;
        if (nu == 1) { g0 = ch_load1(p0); ok = g0.w == tag; }
        else { ch_load4(p0, p1, p2, p3, g0, g1, g2, g3); ok = g0.w == tag && g1.w == tag && g2.w == tag && g3.w == tag; }
        if (__all(ok)) break;
        if (++spins > CH_SPIN_LIMIT) return -3;
        __builtin_amdgcn_s_sleep(CH_POLL_NAP);
    }
    unsigned long long key = ~0ull; int row = INT_MAX, rec = 0;
    if (k0 < npick) { key = ch_lo64(g0); row = (int)g0.z; rec = k0; }
    if (nu > 1) {                                               // strict compares: the lower record (lower rows) wins ties
        if (k1 < npick && ch_lo64(g1) < key) { key = ch_lo64(g1); row = (int)g1.z; rec = k1; }
        if (k2 < npick && ch_lo64(g2) < key) { key = ch_lo64(g2); row = (int)g2.z; rec = k2; }
        if (k3 < npick && ch_lo64(g3) < key) { key = ch_lo64(g3); row = (int)g3.z; rec = k3; }
    }
    // every pick worker takes the same fast / close decision, so record 0 tells which it was
    if (__builtin_amdgcn_readfirstlane((int)g0.z) == CH_CLOSE_ROW) return -1;
    const unsigned long long kmin = ch_wave_min_u64(key);
    // key 0 is no ratio's key (it would be a NaN's): a pick worker met an UNORDERED candidate -- a ratio that is NaN -- and
    // findPivotBV's answer then depends on the order of its scan (lpsol.h:599-611: `minbval > v` is false either way round),
    // which a reduction cannot reproduce. The batch ends here like a CLOSE; the generic pick, which scans in order, takes over.
    if (kmin == 0ull) return -1;
    if (kmin == ~0ull) return -2;
    // the lowest row among the records with the least ratio (lpsol.h:604-611)
    const int myrow = key == kmin ? row : INT_MAX;
    W.r = wave_min_int(myrow);
    const unsigned long long hit = __ballot(myrow == W.r);
    return __builtin_amdgcn_readlane(rec, __ffsll((long long)hit) - 1);
}
// g1..g3 of the winner's record (the asm block's wait also completes whatever loads the caller has in flight)
__device__ __forceinline__ bool ch_load_winner(const LpView<F64> & v, int widx, unsigned tag, ChWinner & W)
{
    const char * p = ch_rec_pay(v, widx);
    ch_u32x4 g1, g2, g3;
    unsigned spins = 0;
    for (;;) {
        ch_load3(p, p + 16, p + 32, g1, g2, g3);
        if (__all(g1.w == tag && g2.w == tag && g3.w == tag)) break;     // (every lane loads the same granules)
        if (++spins > CH_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(CH_POLL_NAP);
    }
    W.a = __builtin_bit_cast(double, ch_lo64(g1)); W.leave = (int)g1.z;
    W.w = g2.x; W.cc = (int)g2.y; W.enter = (int)(g2.z & 0xFFFFFFu); W.qstar = (int)(g2.z >> 24) - 1;
    W.cnv = ch_lo64(g3);
    return true;
}

// ---- the committer: one extra worker that owns the basis arrays --------------------------------------
template <bool LOCAL> __device__ __forceinline__ void ch_commit_loop(const LpView<F64> & v, int batch, int t0, int B, int npick, int nprep,
                                                                     unsigned budget, unsigned done, unsigned tp, bool fold_next)
{
    LoopState * st = v.st;
    const int lane = (int)threadIdx.x;
    for (int t = t0; t < B; t++) {
        const unsigned tag = blk_epoch(batch, t);
        ChWinner g;
        const int widx = ch_poll_records(v, npick, tag, lane, g);
        if (widx == -3 || (widx >= 0 && !ch_load_winner(v, widx, tag, g))) { if (lane == 0) ch_st(&st->status, (int)ST_CHAIN_STUCK); return; }
        if (widx < 0) {
            // CLOSE with budget left, or an empty first ratio pass: the batch takes no more pivots (blk_pick_body /
            // blk_prep_body set `closed` in the same cases); with the budget spent it just ends
            if (lane == 0 && (widx == -2 || budget != 0)) st->blk.closed = 1;
            // at stage 0 nothing is staged: the batch is this launch's, empty, and what the fast path could not do (second
            // ratio pass, disableNV, findPivotNVandBVPair, optimum, limits) is the generic pick's, which the next batch
            // with launches of its own starts with (blk_prep_body leaves the same fields when pick(0)'s first pass is empty)
            if (lane == 0 && t == 0) {
                st->blk.batch = batch; st->blk.n = 0; st->blk.closed = 1; st->blk.generic = 0;
                if (widx == -2) st->blk.want_generic = 1;
            }
            // the pick workers do not read the records: the commit granule tells them that the batch ended here
            ch_drain();
            if (lane == 0)
                ch_store_granule<LOCAL>(ch_part_g0(v, nprep), (unsigned long long)(unsigned)INT_MAX, ((unsigned long long)(unsigned)CH_CLOSE_ROW << 32) | (unsigned long long)tag);
            return;
        }
        if (lane == 0) {
            const int enter = g.enter, leave = g.leave, r = g.r;
            if (!((g.w >> (leave & 31)) & 1u)) {                // genPair, lpsol.h:100-104
                ch_stl<LOCAL>(&v.ppt[(size_t)enter * v.pw + (leave >> 5)], g.w | (1u << (leave & 31)));
                // (this lane is the only writer of the counters while the launch runs; LOCAL: a plain read-modify-write,
                // so that the line stays in the L2 the prep workers read it from)
                if (LOCAL) ch_st_plain(&v.rowcnt[enter], ch_ld(&v.rowcnt[enter]) + 1);
                else __hip_atomic_fetch_add(&v.rowcnt[enter], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ch_stl<LOCAL>(&v.colcnt[leave], g.cc + 1);
            }
            ch_stl<LOCAL>(&v.nv[enter], (uint8_t)0); ch_stl<LOCAL>(&v.nv[leave], (uint8_t)1);          // lpsol.h:1504-1510
            ch_stl<LOCAL>(&v.bv[enter], (uint8_t)1); ch_stl<LOCAL>(&v.bv[leave], (uint8_t)0);
            ch_stl<LOCAL>(&v.eq2bv[r], enter); ch_stl<LOCAL>(&v.bv2eq[enter], r); ch_stl<LOCAL>(&v.bv2eq[leave], -1);
            if ((int)tp < v.trace_cap) { v.trace[2 * tp] = enter; v.trace[2 * tp + 1] = leave; }
            st->total_pivots = tp + 1;
            st->done = done + 1;
            st->blk.budget = budget - 1;
            if (t == 0) { st->blk.batch = batch; st->blk.closed = 0; st->blk.generic = 0; }     // (as blk_prep_body's commit of stage 0)
            st->blk.r[t] = r; st->blk.n = t + 1;
            st->blk.la_from_state = 0;
            st->blk.la_epoch = tag;
            if (t == B - 1 && fold_next) {
                // every stage of this batch is committed: the ticket of a chain launch that does the NEXT batch's stage 0
                // itself (behind this batch's sweep, stream order). The roll call's counters are this launch's no longer.
                for (int x = 0; x < 8; x++) st->blk.ch_arrive[x] = 0u;
                st->blk.ch_decide = 0u;
                st->blk.ch0_la_epoch = tag; st->blk.ch0_budget = budget - 1; st->blk.ch0_done = done + 1; st->blk.ch0_tp = tp + 1;
                st->blk.ch0_ticket = blk_ticket0(batch + 1);
            }
            // (the next batch's pick(0) looks through ceil(W / 64) partial slots for this tag: the chain's nprep <= that many
            // carry it, the rest still hold stage 0's and are ignored)
        }
        // the commit is out: the pick role of stage t+1 may read it. The granule is {INT_MAX -- no column, where a partial
        // holds its first candidate --, 0 -- where a partial holds "any c_j > 0": a launch-per-stage look-ahead that met this
        // slot with a matching tag would OR it in --, tag, the pivot row}: the row is all a pick worker needs of the records of stage t
        ch_drain();
        if (lane == 0)
            ch_store_granule<LOCAL>(ch_part_g0(v, nprep), (unsigned long long)(unsigned)INT_MAX, ((unsigned long long)(unsigned)g.r << 32) | (unsigned long long)tag);
        tp += 1; done += 1; budget -= 1;
    }
}

#ifdef XPG_STAMPS
__device__ unsigned long long g_ch_ts[4][BLK_MAX][8];            // [worker class][stage][point]: raw 100 MHz stamps of the last batch
#define CH_TS(pt_) do { if (lane == 0 && tsw >= 0) g_ch_ts[tsw][t][pt_] = wall_clock64(); } while (0)
#else
#define CH_TS(pt_) do { } while (0)
#endif

// Roles are WORKERS of their own (round 4): npick = ceil(m / 64) pick workers (worker w owns rows 64 w .. 64 w + 63, lane l
// row 64 w + l), nprep = ceil(W / 64) prep workers (columns likewise), one committer; every worker is ONE wave. A worker
// keeps the history its replays need -- a pick worker k_s of its rows, a prep worker e_s of its columns, one double per
// lane and stage -- in 16 KB of LDS: the replay reads it with a run-time stage number (four stages per group, the four
// reads in flight together), the one write per stage is a plain ds_write. Register-resident histories were measured
// first and are worse in every form the compiler offers: 32 `if (s < t)` blocks test all 32 conditions whatever t is and
// writing slot t costs 32 compare-and-select pairs (98.4 k pivots/s at 4096 x 8192); a loop with an early exit, a search
// over t or any other formulation is recognised as a[t] and moves the arrays to scratch memory; 16-wide vector types get
// indexed register moves, but the compiler inserts into both halves and selects, and the half not meant is written with
// an index beyond its vector (a memory fault at t >= 16). Fatter workers (2 rows / 2-4 columns per lane, fewer heads in
// every all-to-all hand-off) were measured too: 1 x 1 91.9 / 71.1 k pivots/s at 4096 x 8192 / 4096 x 12289 in batches of
// 16, 2 x 1 89.1 / 69.6, 1 x 2 86.8 / 69.3, 2 x 2 85.2 / 61.5, 2 x 4 75.3 / 61.4 -- the arithmetic of ONE wave is what a
// stage is made of, so it is split over more waves, not fewer.
// nparts0: how many partials the stage before t0 left (a launch of its own: one per 64 columns, like the chain's).
// force_abort: test hook, compiled under -DXPG_TEST_HOOKS only (XPG_CHAIN_TEST_ABORT=k makes every k-th chain launch fail its
// roll call, -k its placement check); the product build ignores the argument.
// LOCAL: the workers are the workgroups with blockIdx % 8 == 0 of a grid of 8 x workers -- workgroups are dealt round-robin
// over the 8 XCDs, so these all land on ONE XCD, whose L2 then is the point of coherence of every hand-off: plain stores,
// sc1 (L1-bypassing) loads, an L2 round trip per hop instead of a fabric one. The placement is an observation, not a
// contract (MI355X_MICROARCH.md, "Workgroup dispatch"), so the roll call CHECKS it: every worker counts itself in on the
// counter of the XCD it reads from HW_REG_XCC_ID, and the committer says GO only if one counter holds them all; otherwise
// ABORT + blk.ch_misplaced, and the host goes back to the spread (sc1) form of this kernel for the rest of the solve.
// LINE (round 5): the pick workers keep the 128-byte line of their row that holds the entering column -- 16 consecutive
// columns -- in LDS. The first positive reduced cost moves up the columns a step at a time, so the next entering column
// is in the line already held 15 times out of 16 (tools/lab/probe_entering_cache.py); but with a row stride that is a
// multiple of 4 KiB (4096 x 8192: 64 KiB) the 64 lines of a worker's column gather share one or two sets of the compute
// unit's L1, which keeps four of them, and every stage paid the L2 for the rest again: 0.45 us of the gather round
// (tools/lab/gather_lab.hip: the same gather 0.16 us from the L1 at a stride of 64 KiB + 128 B). A stage whose column is in
// the line held now reads nothing of the tableau. The host asks for it where the stride is such a one and the XCD still
// seats every worker with the larger LDS block (ch_lds_bytes); other strides leave the lines to the L1.
// LINE is the number of columns held: 16 (the whole 128-byte line) where the XCD seats every worker with 8 KB more, else 8
// (half of it, 4 KB: a batch of 32 stages has 16 KB of history, and 193 workers of 24 KB are one more than 32 compute units
// of 160 KB hold), else 0.
inline int ch_hist_slots(int B, int line) { return line ? (B + 3) / 4 * 4 : BLK_MAX; }
inline size_t ch_lds_bytes(int B, int line)
{
    const size_t plain = (size_t)BLK_MAX * 64 * 8, with_line = (size_t)ch_hist_slots(B, line) * 64 * 8 + (size_t)line * 64 * 8;
    return line ? (with_line > plain ? with_line : plain) : plain;
}
template <bool LOCAL, int LINE>
__global__ __launch_bounds__(64) void k_blk_chain(LpView<F64> v, int batch, int t0, int B, int npick, int nprep, int nparts0, int force_abort, int fold_next)
{
    extern __shared__ __attribute__((aligned(16))) double ch_hist[];     // [BLK_MAX][64]: this worker's history, stage-major
    if (LOCAL && (blockIdx.x & 7u)) return;
    LoopState * st = v.st;
    const int w = LOCAL ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, lane = (int)threadIdx.x;
    const unsigned nworkers = LOCAL ? gridDim.x >> 3 : gridDim.x;      // npick + nprep + 1
    // ---- the ticket (fields this launch never writes before its last stage is committed). t0 = 1: stage 0 of THIS batch --
    // launches of their own -- staged a pivot. t0 = 0: the chain launch of the batch before committed all its stages and
    // admitted this one (ch_commit_loop); this launch then does stage 0 itself, on the tableau that batch's sweep left: no
    // pending update, so nothing is replayed and nothing fresh is needed but the look-ahead partials of that batch's last stage.
    const bool fold = t0 == 0;
    if ((fold ? st->blk.ch0_ticket != blk_ticket0(batch) : st->blk.ch_epoch != blk_epoch(batch, t0 - 1)) ||
        st->status != ST_RUNNING || st->pricing != 0 || (fold && (st->blk.la_from_state != 0 || st->blk.want_generic != 0))) return;
    unsigned budget = fold ? st->blk.ch0_budget : st->blk.ch_budget, done = fold ? st->blk.ch0_done : st->blk.ch_done;
    const unsigned tp0 = fold ? st->blk.ch0_tp : st->blk.ch_tp;
    const unsigned la_tag = st->blk.ch0_la_epoch;             // (t0 = 0: the tag of the partials stage 0 prices from)
    const unsigned max_iter = st->max_iter;
    // ---- roll call (see the header): count in; the committer decides GO / ABORT for everybody
    char * const decision = ch_part_g0(v, BLK_DECISION_SLOT);
    const unsigned roll_tag = fold ? blk_ticket0(batch) : blk_epoch(batch, t0 - 1);
    if (lane == 0) __hip_atomic_fetch_add(&st->blk.ch_arrive[LOCAL ? ch_xcc_id() : 0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w == (int)nworkers - 1) {
        const unsigned long long t_in = wall_clock64();
        bool go = false, misplaced = false;
        for (;;) {
#ifdef XPG_TEST_HOOKS
            if (force_abort) { misplaced = force_abort == 2; break; }     // fault injection: compiled into the hooks build only
#endif
            // lanes 0..7 read one XCD's counter each
            const unsigned mine = ch_ld(&st->blk.ch_arrive[lane & 7]);
            unsigned sum = 0, top = 0;
#pragma unroll
            for (int x = 0; x < 8; x++) { const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)mine, x); sum += c; top = c > top ? c : top; }
            if (top >= nworkers) { go = true; break; }
            if (sum >= nworkers) { misplaced = true; break; }        // all here, but not on one XCD: the L2 hand-offs would not be coherent
            if (wall_clock64() - t_in > (unsigned long long)CH_ARRIVE_TICKS) break;
            __builtin_amdgcn_s_sleep(CH_POLL_NAP);
        }
        // the verdict is set once: a worker that gave up waiting for this one may have closed the roll call already
        unsigned prev = 0u;
        if (lane == 0) prev = atomicCAS(&st->blk.ch_decide, 0u, go ? (unsigned)CH_GO : (unsigned)CH_ABORT);
        prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
        if (prev != 0u) return;                                 // (vetoed: that worker has done the bookkeeping and published ABORT)
        if (lane == 0) {
            if (go) { st->blk.ch_runs += 1u; if (fold) st->blk.ch_folds += 1u; }
            else { st->blk.closed = 1; st->blk.ch_aborts += 1u; if (misplaced) st->blk.ch_misplaced += 1u; }     // nothing of this launch has touched the state: close the batch at stage 0's pivot
        }
        ch_drain();
        // (the decision is an sc1 store in either form: after a failed placement check its readers may sit on any XCD)
        if (lane == 0) ch_store_granule<false>(decision, (unsigned long long)(go ? CH_GO : CH_ABORT), (unsigned long long)roll_tag);
        if (go) ch_commit_loop<LOCAL>(v, batch, t0, B, npick, nprep, budget, done, tp0, fold_next != 0);
        return;
    }
    const int m = v.m, W = v.W, rhs = v.rhs, ld = v.ld, lim = v.rhs - 1;
    const bool picker = w < npick;
    const double * __restrict__ tab = (const double *)v.tab;
    double * K = (double *)v.blkK;
    double * E = (double *)v.blkE;
    double * const hist = ch_hist + lane;                   // this lane's history: hist[64 s]
    const int hslots = LINE ? (B + 3) / 4 * 4 : BLK_MAX;    // history slots in use (LINE: the lines follow them)
    // The committer answers within CH_ARRIVE_TICKS of its own start. A worker that has waited CH_VETO_TICKS knows that
    // the committer itself is not running (no CU for it while these workers hold theirs): it closes the roll call with
    // ABORT through the same compare-and-swap, does the committer's bookkeeping and tells the others.
    // A worker must have the verdict before its first WRITE to global memory: a pick worker reads it only after the partial
    // poll and the column gather of its first stage (read-only work that hides the hand-off), a prep worker -- which would
    // otherwise wait for records that an aborted launch never publishes -- before its first poll.
    bool decided = false;
    auto wait_decision = [&]() -> bool {
        if (decided) return true;
        unsigned spins = 0;
        const unsigned long long t_wait = wall_clock64();
        bool asked = false;
        for (;;) {
            const ch_u32x4 g = ch_load1(decision);
            if (g.z == roll_tag) { if (g.x != (unsigned)CH_GO) return false; decided = true; return true; }
            if (!asked && wall_clock64() - t_wait > (unsigned long long)CH_VETO_TICKS) {
                asked = true;
                unsigned prev = 0u;
                if (lane == 0) prev = atomicCAS(&st->blk.ch_decide, 0u, (unsigned)CH_ABORT);
                prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
                if (prev == 0u) {                               // this worker closed it
                    if (lane == 0) { st->blk.closed = 1; st->blk.ch_aborts += 1u; }
                    ch_drain();
                    if (lane == 0) ch_store_granule<false>(decision, (unsigned long long)CH_ABORT, (unsigned long long)roll_tag);
                    return false;
                }
                if (prev == (unsigned)CH_ABORT) return false;
                // CH_GO: the committer is about to publish it
            }
            if (++spins > CH_SPIN_LIMIT) { if (lane == 0) ch_st(&st->status, (int)ST_CHAIN_STUCK); return false; }
            __builtin_amdgcn_s_sleep(CH_POLL_NAP);
        }
    };
#ifdef XPG_STAMPS
    const int tsw = w == 0 ? 0 : (w == npick - 1 ? 1 : (w == npick ? 2 : (w == npick + nprep - 1 ? 3 : -1)));
#endif

    const bool crowded = npick + nprep > CH_NAP_FROM;       // (see CH_NAP_PICK)
    if (picker) {
        // =========================== a pick worker ================================================================
        const int i = w * 64 + lane;                        // this lane's row
        const bool has_row = i < m;
        const int ic = has_row ? i : 0;
        // own data of the stages before t0 (written by previous launches); every later slot +0.0 (ch_replay counts on it)
        for (int s = 0; s < hslots; s++) hist[64 * s] = 0.0;
        for (int s = 0; s < t0; s++) hist[64 * s] = has_row ? K[(size_t)ic * BLK_MAX + s] : 0.0;
        double * const lcache = ch_hist + (size_t)hslots * 64 + lane;       // LINE: column q of the line held at lcache[64 q]
        int line_group = -1;
        int bi = v.eq2bv[ic];                               // basic variable of this lane's row (stage 0's commit is in)
        int sstar = -1;                                     // last stage in which this lane's row was the pivot row
        double bcur = tab[(size_t)ic * ld + rhs];           // this row's constant, replayed through the stages before t - 1
        double klast = 0.0;                                 // k of stage t - 1 for this row
        for (int s = 0; s < t0; s++) {
            const int rs = st->blk.r[s];
            if (rs == i) sstar = s;
            const double ks = has_row ? K[(size_t)ic * BLK_MAX + s] : 0.0;
            if (s + 1 < t0) {                               // (t0 = 1: no step here; stage t0 - 1 is applied in the loop)
                const double eb = E[(size_t)s * ld + rhs];
                const double pb = ks * eb;
                bcur = (rs == i) ? eb : (bcur + pb);
            } else klast = ks;
        }
        int r_prev = t0 > 0 ? st->blk.r[t0 - 1] : -1;       // pivot row of stage t - 1 (for the constant's step)
        int first_prev = -1;                                // the entering column of stage t - 1
#pragma unroll 1
        for (int t = t0; t < B; t++) {
            const unsigned want_part = t > 0 ? blk_epoch(batch, t - 1) : la_tag, tag = blk_epoch(batch, t);
            CH_TS(0);
            ChHist H;
            ch_hist_load(H, hist);                          // (in flight while the partials are polled)
            if (CH_NAP_PICK && crowded && t > t0) __builtin_amdgcn_s_sleep(CH_NAP_PICK);
            // ---- poll the commit granule of stage t-1 (index 0; the commit of stage t0 - 1 was a launch of its own: no
            // granule to wait for) and the g0 granules of its partials (index k: slot k - 1): four per lane and round, dense.
            // A lane without an index of its own reads another's again -- every granule that passes is one of this stage's,
            // and a minimum does not mind seeing one twice
            const int cnt = t == t0 ? nparts0 : nprep;      // partials of the stage before
            const int lo = t == t0 ? 1 : 0;
            int nf = INT_MAX;
            for (int base = 0; base <= cnt; base += 256) {
                const int span = cnt + 1 - base;
                auto slot = [&](int k) -> const void * {
                    int kk = k <= cnt ? k : base + (lane % span);
                    kk = kk < lo ? lo : kk;
                    return ch_part_g0(v, kk == 0 ? nprep : kk - 1);
                };
                const void * p0 = slot(base + lane);
                const void * p1 = slot(base + lane + 64);
                const void * p2 = slot(base + lane + 128);
                const void * p3 = slot(base + lane + 192);
                ch_u32x4 g0, g1, g2, g3;
                unsigned spins = 0;
                for (;;) {
                    ch_load4(p0, p1, p2, p3, g0, g1, g2, g3);
                    const bool ok = g0.z == want_part && g1.z == want_part && g2.z == want_part && g3.z == want_part;
                    if (__all(ok)) break;
                    // the batch ended at stage t-1 (a NaN ratio, an empty first pass: see ch_commit_loop) -- no partials will come
                    if (base == 0 && lo == 0 && (unsigned)__builtin_amdgcn_readfirstlane((int)g0.z) == want_part
                        && __builtin_amdgcn_readfirstlane((int)g0.w) == CH_CLOSE_ROW) return;
                    if (++spins > CH_SPIN_LIMIT) { if (lane == 0) ch_st(&st->status, (int)ST_CHAIN_STUCK); return; }
                    __builtin_amdgcn_s_sleep(CH_POLL_NAP);
                }
                nf = min(min(nf, (int)g0.x), min(min((int)g1.x, (int)g2.x), (int)g3.x));
                if (base == 0 && lo == 0) {
                    // the pivot row of stage t-1: its basic variable is the column every pick worker took then (lpsol.h:1508)
                    r_prev = __builtin_amdgcn_readfirstlane((int)g0.w);     // (word 3 of the commit granule; word 1 stays 0: a partial's "any c_j > 0")
                    if (r_prev == CH_CLOSE_ROW) return;
                    if (i == r_prev) { bi = first_prev; sstar = t - 1; }
                }
            }
            CH_TS(1);                                       // partials + commit granule seen
            const int first = wave_min_int(nf);
            // the fast path of blk_pick_body, or the end of the batch for everyone
            const bool fast = first >= 0 && first < rhs && done < max_iter && budget != 0;
            if (!fast) {
                if (!wait_decision()) return;
                if (lane == 0) ch_store_granule3<LOCAL>(ch_rec_g0(v, w), ~0u, ~0u, (unsigned)CH_CLOSE_ROW, tag);
                return;
            }
            // ---- one round: the column gather; e_s[first] for s <= t-2 from memory (lane s); the three fresh
            // granules of stage t-1 (lane 32: e[first], lane 33: c[first], lane 34: e[rhs]); pair word, counter
            // (every load of the round is UNCONDITIONAL, on a clamped address, and selected afterwards: a load under a
            // divergent `if` makes the compiler wait for it where the branch joins -- before the loads behind it are issued;
            // round 4's stage paid two dependent round trips here and in the prep role's round, found in the ISA)
            double x0_mem = 0.0;
            constexpr int LN = LINE ? LINE : 16;
            double2 lnew[LN / 2];
            const bool line_miss = LINE && first / LN != line_group;         // (wave-uniform)
            if (!LINE) x0_mem = tab[(size_t)ic * ld + first];
            else if (line_miss) {
                const double2 * lp2 = (const double2 *)&tab[(size_t)ic * ld + (size_t)(first / LN) * LN];      // ld is a multiple of 16
#pragma unroll
                for (int q = 0; q < LN / 2; q++) lnew[q] = lp2[q];
            }
            const uint32_t pw_word = ch_ld(&v.ppt[(size_t)first * v.pw + (bi >> 5)]);
            const int cc = ch_ld(&v.colcnt[bi]);
            // (a column-major copy of E for this load -- two lines instead of one per stage at the tableau's row stride -- was
            // built and measured in round 5: the preps' extra scattered store per stage costs more than the gather gains, -0.5 %)
            const double ev_mem = ch_ld(&E[(size_t)(lane + 1 < t ? lane : 0) * ld + first]);
            const double c0_mem = v.obj[first].v;
            double ev = 0.0;
            if (t == 0) {                                   // nothing pending: c[first] as the last prep stored it
                if (lane == 33) ev = c0_mem;
            } else {
                const bool fresh = lane >= 32 && lane <= 34;
                const char * gp = ch_part_pay(v, (lane == 34 ? rhs : first) >> 6) + (fresh ? 16 * (lane - 32) : 0);
                unsigned spins = 0;
                for (;;) {
                    const ch_u32x4 g = ch_load1(gp);
                    if (__all(g.z == want_part || !fresh)) { if (fresh) ev = __builtin_bit_cast(double, ch_lo64(g)); break; }
                    if (++spins > CH_SPIN_LIMIT) { if (lane == 0) ch_st(&st->status, (int)ST_CHAIN_STUCK); return; }
                    __builtin_amdgcn_s_sleep(CH_POLL_NAP);
                }
            }
            CH_TS(2);                                       // gather round issued, fresh granules in
            if (lane + 1 < t) ev = ev_mem;                  // (t <= 32: the lanes of the fresh granules are beyond t - 2)
            if (LINE && line_miss) {
#pragma unroll
                for (int q = 0; q < LN / 2; q++) { lcache[64 * (2 * q)] = lnew[q].x; lcache[64 * (2 * q + 1)] = lnew[q].y; }
                line_group = first / LN;
            }
            const double x0 = LINE ? lcache[64 * (first % LN)] : x0_mem;
            const double ec_new = ch_readlane_f64(ev, 32), eb_new = ch_readlane_f64(ev, 34);
            const unsigned long long cnv_bits = __builtin_bit_cast(unsigned long long, ch_readlane_f64(ev, 33));
            if (lane == t - 1) ev = ec_new;                 // lane s of ev now holds e_s[first] for every s < t
            if (lane >= t) ev = -1.0;                       // ... and the others what makes their product -0.0 (ch_replay)
            // the constant column: ONE step, stage t-1 (the arithmetic of every other cell of the sweep)
            if (t > 0) {
                const double pb = klast * eb_new;
                bcur = (i == r_prev) ? eb_new : (bcur + pb);
            }
            // the entering column through the stages 0 .. t-1 (ch_replay). A row that was a pivot row of this batch (at most
            // t of the m) starts from e_sstar[first] -- what the sweep of that stage would have put there -- and counts the
            // stages after it; a wave without such a row takes the form without the per-lane test
            double a;
            if (__any(sstar >= 0)) {
                const double es = __shfl(ev, sstar >= 0 ? sstar : 0);
                a = ch_replay<true>(sstar >= 0 ? es : x0, H, ev, t, sstar);
            } else a = ch_replay<false>(x0, H, ev, t, -1);
            CH_TS(5);                                       // (pick) column replayed
            klast = -a;                                                           // -a_i,nv (lpsol.h:1485)
            hist[64 * t] = klast;
            if (!wait_decision()) return;                                         // (first stage only: nothing global has been written so far)
            if (has_row) ch_stl<LOCAL>(&K[(size_t)i * BLK_MAX + t], klast);
            // findPivotBV's first pass (lpsol.h:553-663)
            unsigned long long key = ~0ull;
            bool unordered = false;
            if (has_row && !le(F64(a), zero<F64>()) && !((pw_word >> (bi & 31)) & 1u) && cc < lim) {
                const double q = bcur / a;
                unordered = q != q;                         // NaN (an overflow's inf - inf, mid-solve): no place in any order
                key = ch_ratio_key(q);
            }
            if (__any(unordered)) {                         // see ch_poll_records: everybody leaves, the generic pick scans in order
                if (lane == 0) ch_store_granule3<LOCAL>(ch_rec_g0(v, w), 0u, 0u, (unsigned)CH_UNORDERED_ROW, tag);
                key = ~0ull;
            }
            const bool told = __any(unordered);
            CH_TS(6);                                       // (pick) ratio and key done
            const unsigned long long kmin = ch_wave_min_u64(key);
            const unsigned long long hit = __ballot(key == kmin);
            const bool publisher = kmin != ~0ull ? (lane == __ffsll((long long)hit) - 1) : (lane == 0);
            CH_TS(7);                                       // (pick) wave minimum done
            if (publisher && !told) {
                // g0 first: it is what every worker of the launch waits for; the payload is read of the winner only, one round
                // later, and carries its own tags
                ch_store_granule3<LOCAL>(ch_rec_g0(v, w), (unsigned)kmin, (unsigned)(kmin >> 32), (unsigned)(kmin != ~0ull ? i : INT_MAX), tag);
                const unsigned long long ab = to_bits(F64(a));
                char * pay = ch_rec_pay(v, w);
                ch_store_granule3<LOCAL>(pay, (unsigned)ab, (unsigned)(ab >> 32), (unsigned)bi, tag);
                ch_store_granule3<LOCAL>(pay + 16, pw_word, (unsigned)cc, (unsigned)first | ((unsigned)(sstar + 1) << 24), tag);
                ch_store_granule3<LOCAL>(pay + 32, (unsigned)cnv_bits, (unsigned)(cnv_bits >> 32), 0u, tag);
            }
            CH_TS(3);                                       // record issued
            // (who won is the committer's and the prep workers' to find out: this role meets the pivot row in the commit
            // granule, at the top of the next stage, and none of its own work before that depends on it)
            first_prev = first;
            done += 1; budget -= 1;
        }
        return;
    }

    // =========================== a prep worker ====================================================================
    const int wp = w - npick;                               // its partial slot
    const int j = wp * 64 + lane;                           // this lane's column
    const bool has_col = j < W;
    const int jc = has_col ? j : 0;
    for (int s = 0; s < hslots; s++) hist[64 * s] = 0.0;   // (slots at and beyond the current stage +0.0: ch_replay)
    for (int s = 0; s < t0; s++) hist[64 * s] = has_col ? E[(size_t)s * ld + jc] : 0.0;
    F64 oj = has_col ? v.obj[jc] : zero<F64>();            // this lane's objective entry
    if (!wait_decision()) return;
#pragma unroll 1
    for (int t = t0; t < B; t++) {
        const unsigned tag = blk_epoch(batch, t);
        CH_TS(0);
        ChHist H;
        ch_hist_load(H, hist);                              // (in flight while the records are polled)
        if (CH_NAP_PREP && crowded && t > t0) __builtin_amdgcn_s_sleep(CH_NAP_PREP);
        ChWinner g;
        const int widx = ch_poll_records(v, npick, tag, lane, g);
        if (widx == -3) { if (lane == 0) ch_st(&st->status, (int)ST_CHAIN_STUCK); return; }
        if (widx < 0) return;                               // CLOSE / empty first pass: the committer records it
        CH_TS(4);                                           // records seen and combined
        const int r = g.r;
        // ---- one round: the pivot row gather, k_q[r] (lane q), this column's basis words (the previous stage's
        // commit is behind the records just read), and the rest of the winner's record (whose wait completes the
        // loads before it as well)
        // (unconditional loads on clamped addresses, selected behind the round: see the pick role's round)
        const double x0 = tab[(size_t)r * ld + jc];
        const bool in_vars = has_col && j < rhs;
        const int nv_mem = (int)ch_ld(&v.nv[in_vars ? jc : 0]), rc_mem = ch_ld(&v.rowcnt[in_vars ? jc : 0]);
        const double kv_mem = ch_ld(&K[(size_t)r * BLK_MAX + (lane < t ? lane : 0)]);
        if (!ch_load_winner(v, widx, tag, g)) { if (lane == 0) ch_st(&st->status, (int)ST_CHAIN_STUCK); return; }
        const int nvj = in_vars ? nv_mem : 0, rcj0 = in_vars ? rc_mem : INT_MAX;
        const double kv = lane < t ? kv_mem : -1.0;         // (lanes at and beyond t: what makes their product -0.0, ch_replay)
        CH_TS(1);                                           // (prep) row gather round and the winner's payload in
        const int enter = g.enter, leave = g.leave;
        const F64 sc = div(one<F64>(), F64(g.a));           // 1/(eq.get(eqnum, nv)), lpsol.h:1471
        const int smode = scale_mode(sc);
        const F64 cnv = from_bits<F64>(g.cnv);
        const int cmode = scale_mode(cnv);
        const int qstar = __builtin_amdgcn_readfirstlane(g.qstar);     // the last stage before t in which row r was the pivot row (every lane read the same record)
        // the pivot row as the pending sweeps would leave it (ch_replay): from e_qstar if row r was a pivot row of this
        // batch before (wave-uniform), else from the tableau
        const double x = qstar >= 0 ? ch_replay<true>(hist[64 * qstar], H, kv, t, qstar) : ch_replay<false>(x0, H, kv, t, -1);
        CH_TS(2);                                           // (prep) row replayed
        const F64 e = scaled(F64(x), sc, smode);
        hist[64 * t] = e.v;
        CH_TS(5);                                           // row gather in, replayed, scaled
        int nf = INT_MAX, any = 0;
        if (has_col) {
            ch_stl<LOCAL>(&E[(size_t)t * ld + j], e.v);
            F64 tt = mul(e, minus_one<F64>());              // nvexp.mul(-1), lpsol.h:1496
            if (j >= rhs) tt = neg(tt);                     // :1497-1499
            tt = scaled(tt, cnv, cmode);                    // nvexp.mul(tgtf(nv)), :1500
            const bool in = j < rhs;
            const bool nv_mem = in && j != enter && j != leave && nvj != 0;
            const bool nv_old = in && (j == enter ? true : (j == leave ? false : nv_mem));
            const bool nv_new = in && (j == enter ? false : (j == leave ? true : nv_mem));
            const int rcj = (in && j != enter) ? rcj0 : INT_MAX;
            if (j < enter && in && !nv_old) oj = zero<F64>();     // lpsol.h:1055-1060
            oj = add(tt, oj);                               // addRowToRow, :1501
            v.obj[j] = oj;                                  // (read again only by later launches)
            if (nv_new && gt(oj, zero<F64>())) { any = 1; if (rcj < lim) nf = j; }
        }
        CH_TS(3);                                           // (prep) objective entry and pricing done
        nf = wave_min_int(nf);
        any = __ballot(any != 0) != 0ull ? 1 : 0;
        // ---- the partial: the lane that owns the worker's candidate column publishes what the next pick needs
        // fresh of it, the lane that owns the constant column its new e, lane 0 the candidate itself
        char * pay = ch_part_pay(v, wp);
        if (has_col && j == nf) {
            ch_store_granule<LOCAL>(pay, to_bits(e), (unsigned long long)tag);
            ch_store_granule<LOCAL>(pay + 16, to_bits(oj), (unsigned long long)tag);
        }
        if (has_col && j == rhs) ch_store_granule<LOCAL>(pay + 32, to_bits(e), (unsigned long long)tag);
        if (lane == 0) ch_store_granule<LOCAL>(ch_part_g0(v, wp), ((unsigned long long)(unsigned)any << 32) | (unsigned)nf, (unsigned long long)tag);
        CH_TS(6);                                           // partial issued
    }
}

} // namespace xpg
