// Rational row elimination for batches of small systems -- the work behind the
// polyhedral dependence tests and loop-bound generation:
//   Lineq::removeIdenRow / reduce / fme     (src/com/linsys.cpp:1209-1268, :359-626, :656-774)
//   Matrix<Rational>::rank / det / inv      (src/com/matt.h:2614-2726, :1621-1736, :1743-1845)
//   row primitives mulOfRow / addRowToRow / mul_and_add_row / interch_row / is_rowequ
//                                            (src/com/matt.h:1353, :1437-1460, :1493, :1097, :2344)
// Systems are tiny (W ~ 9-20 columns, R ~ 10-60 rows) and
// thousands are independent. A system is owned by a GROUP of 16, 32 or 64 lanes of one wavefront (the
// smallest that covers its columns: round 1 gave every system a whole wave and left >= 44 of 64 lanes idle at
// these widths); a wavefront carries 64 / L systems. The workgroup is dim3(L, 64 / L): threadIdx.x is the lane
// within the group, threadIdx.y the group. The matrix is staged in the group's slice of LDS, the reference's
// data-dependent control flow is uniform within a group (groups of one wave diverge from each other: the
// hardware masks), every row operation runs one lane per column, and the only synchronisation a group needs is
// the program order of its own wave -- an LDS fence, no s_barrier (which divergent groups could not share).
// The arithmetic is integer (gcd loops), so the bound is ALU/divergence, not HBM.
#pragma once
#include "scalar.hip.h"
#include "rat_ops.hip.h"
#include "lineq_shared.hip.h"
#include <limits.h>

namespace xpg {

enum { CST_UNK = 1, CST_LT = 2, CST_GT = 3, CST_EQ = 4 };       // linsys.h:55-58

// A small row-major rational matrix in LDS (or global scratch), one wave's property.
struct WMat { R32 * a; int r, c, ld; bool cn; };     // cn: every cell canonical (set by w_load)

__device__ __forceinline__ int lane_id() { return (int)threadIdx.x; }          // lane within the system's group
__device__ __forceinline__ int wave_lanes() { return (int)blockDim.x; }         // lanes per system: 16, 32 or 64
// One wave per workgroup: LDS operations of a wave complete in program order, so what a group needs between a
// store and another lane's load is only that the compiler keeps the order.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
// this group's system index / slice of the dynamic LDS
__device__ __forceinline__ int sys_first() { return (int)(blockIdx.x * blockDim.y + threadIdx.y); }
__device__ __forceinline__ int sys_stride() { return (int)(gridDim.x * blockDim.y); }
// Ballot over the lanes of this system's group, bit l = lane l of the group. Control flow is uniform within a
// group, so all of its lanes are here; lanes of the wave's other groups may or may not be and are masked out.
__device__ __forceinline__ unsigned long long grp_ballot(bool p)
{
    const unsigned long long b = __ballot(p);
    const int L = wave_lanes();
    return L == 64 ? b : (b >> (threadIdx.y * L)) & ((1ull << L) - 1);
}
__device__ __forceinline__ int grp_rank(unsigned long long bal) { return __popcll(bal & ((1ull << lane_id()) - 1)); }

// t / d for the cell loops (t = row * cols + col and the like): gfx950 has no integer divide, the generic sequence is
// ~35 instructions. With M = ceil(2^32 / d), umulhi(t, M) == t / d whenever t * d < 2^32; otherwise the plain one.
struct FastDiv {
    unsigned d, M; bool ok;
    __device__ FastDiv(int d_, int max_t)
        : d((unsigned)d_), M(0xFFFFFFFFu / (unsigned)(d_ > 0 ? d_ : 1) + 1u),
          ok(d_ > 1 && (unsigned long long)(max_t > 0 ? max_t : 0) * (unsigned long long)d_ < 0xFFFFFFFFull) {}
    __device__ __forceinline__ int quot(int t) const { return d == 1 ? t : (ok ? (int)__umulhi((unsigned)t, M) : (int)((unsigned)t / d)); }
};

#ifdef XPG_STAMPS
// diagnostic builds: clock ticks (s_memtime) per phase of k_fme_batch. Lane 0 of the system's wave adds them up in
// LDS (a global atomic per stamp would be waited for by the next phase's fence and be what is measured); the
// sums go to g_lq_ticks once per system, after the last stamp.
__device__ unsigned long long g_lq_ticks[16];
__shared__ unsigned long long lq_acc[16];
#define LQ_T0 unsigned long long lq_t = __builtin_readcyclecounter();
#define LQ_T(k) { const unsigned long long n_ = __builtin_readcyclecounter(); if (lane_id() == 0) lq_acc[k] += n_ - lq_t; lq_t = n_; }
#define LQ_FLUSH { if (lane_id() < 16) { atomicAdd(&g_lq_ticks[lane_id()], lq_acc[lane_id()]); lq_acc[lane_id()] = 0; } }
#define LQ_CLEAR { if (lane_id() < 16) lq_acc[lane_id()] = 0; }
#else
#define LQ_T0
#define LQ_T(k)
#define LQ_FLUSH
#define LQ_CLEAR
#endif

// mulOfRow (matt.h:1353-1368): every lane evaluates the same shortcut tests.
__device__ inline void w_scale_row(WMat & m, int row, R32 x)
{
    const int mode = scale_mode(x);
    if (mode != SCALE_KEEP)
        for (int j = lane_id(); j < m.c; j += wave_lanes()) {
            R32 * p = m.a + row * m.ld + j;
            *p = q_scaled(m.cn, *p, x, mode);
        }
    wave_sync();
}

// Lineq::compareConstIterm (linsys.cpp:204-231, :235-274); constant symbols are
// the columns after rhs.
__device__ inline int w_cmp_value(const WMat & m, int rhs, int row, R32 v)
{
    for (int j = rhs + 1; j < m.c; j++) if (ne(m.a[row * m.ld + j], R32(0, 1))) return CST_UNK;
    const R32 c = m.a[row * m.ld + rhs];
    if (eq(c, v)) return CST_EQ;
    return lt(c, v) ? CST_LT : CST_GT;
}
__device__ inline int w_cmp_rows(const WMat & m, int rhs, int r1, int r2)
{
    bool s1 = false, s2 = false, same = true;
    for (int j = rhs + 1; j < m.c; j++) {
        const R32 a = m.a[r1 * m.ld + j], b = m.a[r2 * m.ld + j];
        if (ne(a, R32(0, 1))) s1 = true;
        if (ne(b, R32(0, 1))) s2 = true;
        if (ne(a, b)) { same = false; break; }
    }
    if ((!s1 && !s2) || same) {
        const R32 a = m.a[r1 * m.ld + rhs], b = m.a[r2 * m.ld + rhs];
        if (eq(a, b)) return CST_EQ;
        return lt(a, b) ? CST_LT : CST_GT;
    }
    return CST_UNK;
}

// Compacts the rows whose flag is 0, keeping their order. flags/scratch in LDS. The destinations come from a
// ballot prefix; the cells move in ascending chunks of one per lane (a destination never lies above its source, so
// a chunk that is read whole before it is written cannot overwrite anything still to be read).
__device__ inline void w_compact(WMat & m, const unsigned char * drop, int * map)
{
    const int L = wave_lanes(), lane = lane_id();
    int keep = 0, first = -1;
    for (int base = 0; base < m.r; base += L) {
        const int i = base + lane;
        const bool stays = i < m.r && !drop[i];
        const unsigned long long bal = grp_ballot(stays), in = grp_ballot(i < m.r);
        if (i < m.r) map[i] = stays ? keep + grp_rank(bal) : -1;
        if (first < 0 && bal != in) first = base + __builtin_ctzll(bal ^ in);
        keep += __popcll(bal);
    }
    wave_sync();
    if (first >= 0) {
        const int total = m.r * m.c;
        const FastDiv by_c(m.c, total + L);
        for (int t0 = first * m.c; t0 < total; t0 += L) {
            const int t = t0 + lane, i = by_c.quot(t), j = t - i * m.c;
            const int to = t < total ? map[i] : -1;
            R32 v;
            if (to >= 0) v = m.a[i * m.ld + j];
            wave_sync();
            if (to >= 0) m.a[to * m.ld + j] = v;
            wave_sync();
        }
    }
    m.r = keep;
}

// Lineq::removeIdenRow (linsys.cpp:1209-1268): a row goes iff an earlier row is
// field-wise identical (the row-sum test there is only a prefilter; ours is a hash of the fields in map[]).
// A lane owns rows kb + lane, kb + L + lane, ... (P of them in registers). The hashes of the rows below go by once,
// in descending order, one broadcast LDS read each: what is left in first[] is the SMALLEST row index with the
// lane's hash, so a row has an earlier look-alike iff first < k, and only such a row is compared field by field.
template <int P>
__device__ inline bool w_iden_rows(const WMat & m, const int * map, unsigned char * drop, int kb)
{
    const int L = wave_lanes(), lane = lane_id();
    int hk[P], first[P];
#pragma unroll
    for (int q = 0; q < P; q++) { const int k = kb + q * L + lane; hk[q] = k < m.r ? map[k] : 0; first[q] = k; }
    const int kend = m.r < kb + P * L ? m.r : kb + P * L;
    // four hashes per trip, read before any is used (the LDS latency is paid once per four); the trip that holds
    // kend - 1 may reach up to three entries past it -- inside the scratch, and harmless: a row's own index is
    // visited after any larger one
    for (int i4 = (kend - 1) & ~3; i4 >= 0; i4 -= 4) {
        const int h3 = map[i4 + 3], h2 = map[i4 + 2], h1 = map[i4 + 1], h0 = map[i4];
#pragma unroll
        for (int q = 0; q < P; q++) {
            first[q] = h3 == hk[q] ? i4 + 3 : first[q];
            first[q] = h2 == hk[q] ? i4 + 2 : first[q];
            first[q] = h1 == hk[q] ? i4 + 1 : first[q];
            first[q] = h0 == hk[q] ? i4 : first[q];
        }
    }
    bool any = false;
#pragma unroll
    for (int q = 0; q < P; q++) {
        const int k = kb + q * L + lane;
        if (k >= m.r) continue;
        unsigned char gone = 0;
        for (int i = first[q]; i < k && !gone; i++) {
            if (map[i] != hk[q]) continue;
            bool same = true;
            for (int j = 0; j < m.c && same; j++) same = eq(m.a[i * m.ld + j], m.a[k * m.ld + j]);
            gone = same ? 1 : 0;
        }
        drop[k] = gone;
        any |= gone != 0;
    }
    return any;
}
__device__ inline void w_remove_iden(WMat & m, unsigned char * drop, int * map)
{
    LQ_T0
    for (int k = lane_id(); k < m.r; k += wave_lanes()) {
        unsigned h = 0;
        for (int j = 0; j < m.c; j++) {
            const R32 v = m.a[k * m.ld + j];
            h = (h ^ (unsigned)v.num) * 0x9E3779B1u; h = (h ^ (unsigned)v.den) * 0x85EBCA77u;
        }
        map[k] = (int)h;
    }
    wave_sync();
    LQ_T(12)
    bool any = false;
    const int L = wave_lanes();
    for (int kb = 0; kb < m.r;) {
        const int left = (m.r - kb + L - 1) / L;
        if (left >= 8) { any |= w_iden_rows<8>(m, map, drop, kb); kb += 8 * L; }
        else if (left > 2) { any |= w_iden_rows<4>(m, map, drop, kb); kb += 4 * L; }
        else if (left == 2) { any |= w_iden_rows<2>(m, map, drop, kb); kb += 2 * L; }
        else { any |= w_iden_rows<1>(m, map, drop, kb); kb += L; }
    }
    LQ_T(13)
    if (grp_ballot(any) == 0) return;          // nothing identical: rows and order stay
    wave_sync();
    w_compact(m, drop, map);
    LQ_T(14)
}

// One side of Lineq::reduce's pairwise tightening (linsys.cpp:433-497, :505-573).
// rows[] lists, in row order, the single-variable rows of `var` with the wanted sign.
__device__ inline void w_tighten(WMat & m, int rhs, int var, const short * rows, int n, bool negative,
                                 bool is_intersect, unsigned char * removed, int * any_removed)
{
    const int last = n - 1;
    for (int k1 = 0; k1 < last; k1++) {
        const int r1 = rows[k1];
        if (removed[r1]) continue;
        R32 c = m.a[r1 * m.ld + var];
        if (negative) c = neg(c);
        if (ne(c, R32(1, 1))) w_scale_row(m, r1, q_div(m.cn, R32(1, 1), c));
        bool r1_gone = false;
        for (int k2 = k1 + 1; k2 <= last; k2++) {
            const int r2 = rows[k2];
            if (removed[r2]) continue;
            c = m.a[r2 * m.ld + var];
            if (negative) c = neg(c);
            if (ne(c, R32(1, 1))) w_scale_row(m, r2, q_div(m.cn, R32(1, 1), c));
            const int cres = w_cmp_rows(m, rhs, r1, r2);
            wave_sync();
            if (is_intersect) {
                if (cres == CST_LT || cres == CST_EQ) { if (lane_id() == 0) { removed[r2] = 1; *any_removed = 1; } }
                else if (cres == CST_GT) { if (lane_id() == 0) { removed[r1] = 1; *any_removed = 1; } r1_gone = true; }
            } else {
                if (cres == CST_LT || cres == CST_EQ) { if (lane_id() == 0) { removed[r1] = 1; *any_removed = 1; } r1_gone = true; }
                else if (cres == CST_GT) { if (lane_id() == 0) { removed[r2] = 1; *any_removed = 1; } }
            }
            wave_sync();
            if (r1_gone) break;
        }
    }
}

// Per-wave scratch that lives next to the matrix in LDS.
struct WScratch {
    unsigned char * drop;     // cap_rows
    int * map;                // cap_rows + 1
    short * pos; short * negs;   // cap_rows each: row lists of the current variable
    int * flags;              // 4 ints
};

// Lineq::reduce (linsys.cpp:359-626). Returns consistency; m rewritten in place.
__device__ inline bool w_reduce(WMat & m, int rhs, bool is_intersect, WScratch & s)
{
    LQ_T0
    w_remove_iden(m, s.drop, s.map);
    LQ_T(8)
    unsigned char * removed = s.drop;
    int * any_removed = &s.flags[0];
    int * bad = &s.flags[1];
    // single-variable classification: kind[i] = var (+1) with sign, 0 = other
    int * kind = s.map;
    if (lane_id() == 0) { *any_removed = 0; *bad = 0; }
    for (int i = lane_id(); i < m.r; i += wave_lanes()) { removed[i] = 0; kind[i] = 0; }
    wave_sync();
    unsigned long long singles = 0;            // variables that own a single-variable row (bit 63: any var >= 63)
    for (int i = lane_id(); i < m.r; i += wave_lanes()) {                       // linsys.cpp:378-421
        int vars = 0, single = -1;
        for (int j = 0; j < rhs; j++) if (ne(m.a[i * m.ld + j], R32(0, 1))) { vars++; single = j; }
        if (vars == 0) {
            const int c = w_cmp_value(m, rhs, i, R32(0, 1));
            if (c == CST_LT) *bad = 1;
            else if (c == CST_EQ || c == CST_GT) { removed[i] = 1; *any_removed = 1; }
        } else if (vars == 1) {
            const R32 c = m.a[i * m.ld + single];
            if (gt(c, R32(0, 1))) kind[i] = single + 1;
            else if (lt(c, R32(0, 1))) kind[i] = -(single + 1);
            singles |= 1ull << (single < 63 ? single : 63);
        }
    }
    for (int o = 1; o < wave_lanes(); o <<= 1) {                                // OR over the group (xor butterfly)
        singles |= ((unsigned long long)(unsigned)__shfl_xor((int)(singles >> 32), o) << 32)
                   | (unsigned)__shfl_xor((int)singles, o);
    }
    wave_sync();
    LQ_T(9)
    if (*bad) {
        // The reference stops at the FIRST inconsistent row, having marked the constant-true
        // rows before it as removed -- but it returns without compacting, so only the flag matters.
        return false;
    }
    for (int var = 0; var < rhs; var++) {
        if (!((singles >> (var < 63 ? var : 63)) & 1)) continue;       // no single-variable row of var: nothing to do
        int np = 0, nn = 0;
        for (int base = 0; base < m.r; base += wave_lanes()) {
            const int i = base + lane_id();
            const int kd = i < m.r ? kind[i] : 0;
            const unsigned long long bp = grp_ballot(kd == var + 1), bn = grp_ballot(kd == -(var + 1));
            if (kd == var + 1) s.pos[np + grp_rank(bp)] = (short)i;
            if (kd == -(var + 1)) s.negs[nn + grp_rank(bn)] = (short)i;
            np += __popcll(bp); nn += __popcll(bn);
        }
        wave_sync();
        if (np) w_tighten(m, rhs, var, s.pos, np, false, is_intersect, removed, any_removed);
        if (nn) w_tighten(m, rhs, var, s.negs, nn, true, is_intersect, removed, any_removed);
        if (is_intersect && np && nn) {                                // linsys.cpp:577-602
            for (int a = 0; a < np; a++) {
                const int pi = s.pos[a];
                R32 c = m.a[pi * m.ld + var];
                if (ne(c, R32(1, 1))) w_scale_row(m, pi, q_div(m.cn, R32(1, 1), c));
                for (int b = 0; b < nn; b++) {
                    const int ni = s.negs[b];
                    c = neg(m.a[ni * m.ld + var]);
                    wave_sync();
                    if (ne(c, R32(1, 1))) w_scale_row(m, ni, q_div(m.cn, R32(-1, 1), c));
                    else w_scale_row(m, ni, R32(-1, 1));
                    const int cres = w_cmp_rows(m, rhs, pi, ni);
                    wave_sync();
                    w_scale_row(m, ni, R32(-1, 1));
                    if (cres == CST_LT) return false;
                }
            }
        }
    }
    LQ_T(10)
    if (*any_removed) w_compact(m, removed, s.map);
    LQ_T(11)
    return true;
}

// ---- batched entry kernels --------------------------------------------------------------------
__device__ inline void w_load(WMat & m, const R32 * src, int rows, int cols)
{
    m.r = rows; m.c = cols;
    bool bad = false;
    for (int t = lane_id(); t < rows * cols; t += wave_lanes()) {
        const R32 v = src[t];
        bad |= !canonical(v);
        m.a[t] = v;                              // ld == cols for every matrix that comes through here
    }
    m.cn = grp_ballot(bad) == 0;
    wave_sync();
}
__device__ inline void w_store(const WMat & m, R32 * dst)
{
    for (int t = lane_id(); t < m.r * m.c; t += wave_lanes()) dst[t] = m.a[t];          // ld == c, as in w_load
}

__device__ inline WScratch carve_scratch(unsigned char * p, int cap)
{
    WScratch s;
    s.map = (int *)p; p += (size_t)(cap + 1) * 4;
    s.flags = (int *)p; p += 16;
    s.pos = (short *)p; p += (size_t)((cap + 1) & ~1) * 2;
    s.negs = (short *)p; p += (size_t)((cap + 1) & ~1) * 2;
    s.drop = p;
    return s;
}
// (lineq_lds_bytes, the LDS bytes of one system with this carve: lineq_shared.hip.h)

// mode 0: removeIdenRow, 1: reduce. One wave per system; in/out [nb][rows][cols] in place.
__global__ __launch_bounds__(64) void k_reduce_batch(int nb, R32 * mats, int rows, int cols, int rhs,
                                                     int mode, int is_intersect, int * out_rows, int * out_ok, int sys_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    unsigned char * lds = lds_all + (size_t)threadIdx.y * sys_lds;
    WMat m; m.a = (R32 *)lds; m.ld = cols;
    WScratch s = carve_scratch(lds + (size_t)rows * cols * 8, rows);
    for (int b = sys_first(); b < nb; b += sys_stride()) {
        R32 * g = mats + (size_t)b * rows * cols;
        w_load(m, g, rows, cols);
        bool ok = true;
        if (mode == 0) w_remove_iden(m, s.drop, s.map);
        else ok = w_reduce(m, rhs, is_intersect != 0, s);
        wave_sync();
        w_store(m, g);
        if (lane_id() == 0) { out_rows[b] = m.r; out_ok[b] = ok ? 1 : 0; }
        wave_sync();
    }
}

// Lineq::fme (linsys.cpp:656-774) for one system per wave. out has cap rows; cap_lds of them have room in LDS
// (see the layout note in the body).
// Inputs are [nb][cap_in][cols] with in_rows[b] live rows each (NULL: all cap_in rows), which
// lets eliminations be chained on the device (Lineq::calcBound). chain_ok (may be NULL) carries
// a system's state through a chain: only systems whose entry is 1 are processed, and the entry
// becomes 0 when this elimination finds the system inconsistent.
__global__ __launch_bounds__(64) void k_fme_batch(int nb, const R32 * mats, int cap_in, const int * in_rows, int cols,
                                                  int rhs, int u, int darkshadow, R32 * outs, int cap, int * out_rows,
                                                  int * out_ok, int cap_lds, int * chain_ok, int sys_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    unsigned char * lds = lds_all + (size_t)threadIdx.y * sys_lds;
    // LDS layout: [result area (cap_lds x cols)] [normalised input (cap_in x cols)] [scratch]. A system whose result
    // (free rows + P x N pair rows, known once its rows are classified) fits cap_lds rows is built and reduced in
    // LDS and stored at the end; a larger one is built and reduced directly in its HBM output slot (flat pointers;
    // L2-resident at these sizes). The host sizes cap_lds for the typical result, not for the caller's worst-case
    // cap, so that more systems share a CU (lineq_host.hip.h, fme_lds).
    R32 * const res_lds = (R32 *)lds;
    WMat res; res.a = res_lds; res.ld = cols; res.c = cols;
    WMat tmp; tmp.a = res_lds + (size_t)cap_lds * cols; tmp.ld = cols;
    WScratch s = carve_scratch((unsigned char *)(tmp.a + (size_t)cap_in * cols), cap > cap_in ? cap : cap_in);
    LQ_CLEAR
    for (int b = sys_first(); b < nb; b += sys_stride()) {
        if (chain_ok && chain_ok[b] != 1) { if (lane_id() == 0) { out_rows[b] = 0; out_ok[b] = 0; } continue; }
        const int rows = in_rows ? in_rows[b] : cap_in;
        const R32 * g = mats + (size_t)b * cap_in * cols;
        R32 * const go = outs + (size_t)b * cap * cols;
        LQ_T0
        w_load(tmp, g, rows, cols);
        LQ_T(0)
        res.r = 0; res.cn = tmp.cn;
        // classify rows: 0 = no u (copy), 1 = positive, 2 = negative; 3 = inconsistent constant row
        int * kind = s.map;
        int * bad_at = &s.flags[1];
        if (lane_id() == 0) *bad_at = INT_MAX;
        wave_sync();
        for (int i = lane_id(); i < rows; i += wave_lanes()) {
            bool have = false;
            for (int j = 0; j < rhs && !have; j++) have = ne(tmp.a[i * cols + j], R32(0, 1));
            if (!have && w_cmp_value(tmp, rhs, i, R32(0, 1)) == CST_LT) atomicMin(bad_at, i);
            const R32 c = tmp.a[i * cols + u];
            kind[i] = ne(c, R32(0, 1)) ? (gt(c, R32(0, 1)) ? 1 : 2) : 0;
        }
        wave_sync();
        const int stop = *bad_at == INT_MAX ? rows : *bad_at;      // rows before the first bad one are processed
        LQ_T(1)
        // the row lists: rows without u go first, in order (linsys.cpp:724-730); positive and negative rows of u
        int nfree = 0, np = 0, nn = 0;
        for (int base = 0; base < stop; base += wave_lanes()) {
            const int i = base + lane_id();
            const int kd = i < stop ? kind[i] : -1;
            const unsigned long long bf = grp_ballot(kd == 0), bp = grp_ballot(kd == 1), bn = grp_ballot(kd == 2);
            if (kd == 0) kind[i] = -(nfree + grp_rank(bf)) - 1;       // destination row, encoded negative
            else if (kd == 1) s.pos[np + grp_rank(bp)] = (short)i;
            else if (kd == 2) s.negs[nn + grp_rank(bn)] = (short)i;
            nfree += __popcll(bf); np += __popcll(bp); nn += __popcll(bn);
        }
        wave_sync();
        // where this system's result lives
        const int need = nfree + (np + nn == 1 ? 1 : np * nn);
        const bool in_lds = need <= cap_lds;
        res.a = in_lds ? res_lds : go;
        // normalise the rows that contain u (linsys.cpp:711-723): the rows are independent, so the factors are
        // found one lane per row (kept in the still empty result matrix) and applied one lane per cell
        R32 * fac = cap_lds ? res_lds : go;                       // (the host keeps cap_lds * cols >= cap_in when cap_lds > 0)
        for (int i = lane_id(); i < stop; i += wave_lanes()) {
            int mode = SCALE_KEEP;
            if (kind[i] > 0) {
                const R32 c = tmp.a[i * cols + u];
                R32 f(1, 1);
                if (kind[i] == 1) { if (ne(c, R32(1, 1))) f = q_div(tmp.cn, R32(1, 1), c); }
                else if (ne(c, R32(-1, 1))) f = q_div(tmp.cn, R32(1, 1), neg(c));
                mode = scale_mode(f);
                fac[i] = f;
            }
            s.drop[i] = (unsigned char)mode;
        }
        wave_sync();
        const FastDiv by_cols(cols, (cap > cap_in ? cap : cap_in) * cols + wave_lanes());
        for (int t = lane_id(); t < stop * cols; t += wave_lanes()) {
            const int i = by_cols.quot(t), mode = s.drop[i];
            if (mode != SCALE_KEEP) tmp.a[t] = q_scaled(tmp.cn, tmp.a[t], fac[i], mode);
        }
        wave_sync();
        if (darkshadow) {
            for (int i = lane_id(); i < stop; i += wave_lanes())
                if (kind[i] == 2) tmp.a[i * cols + rhs] = q_sub(tmp.cn, tmp.a[i * cols + rhs], R32(1, 1));
            wave_sync();
        }
        LQ_T(2)
        // rows without u are copied first
        wave_sync();
        for (int t = lane_id(); t < stop * cols; t += wave_lanes()) {
            const int i = by_cols.quot(t), j = t - i * cols;
            if (kind[i] < 0) res.a[(-kind[i] - 1) * cols + j] = tmp.a[t];
        }
        res.r = nfree;
        LQ_T(3)
        bool ok = true;
        int status_rows = -1;
        if (stop < rows) {                                              // inconsistent constant row
            ok = false;
        } else {
            int extra = 0;
            if (np + nn == 1) extra = 1;
            else if (np + nn > 1) extra = np * nn;
            if (res.r + extra > cap) { status_rows = res.r + extra; ok = false; }
            else if (np + nn == 1) {
                const int pi = np == 1 ? s.pos[0] : s.negs[0];
                for (int j = lane_id(); j < cols; j += wave_lanes()) res.a[res.r * cols + j] = tmp.a[pi * cols + j];
                res.r += 1;
            } else if (np + nn > 1) {                                   // every (pos, neg) pair summed
                const int base = res.r * cols, total = np * nn * cols;
                const FastDiv by_nn(nn, np * nn);
                for (int t = lane_id(); t < total; t += wave_lanes()) {
                    const int pair = by_cols.quot(t), j = t - pair * cols;
                    const int pq = by_nn.quot(pair);
                    const int pi = s.pos[pq], ni = s.negs[pair - pq * nn];
                    res.a[base + t] = q_add(tmp.cn, tmp.a[pi * cols + j], tmp.a[ni * cols + j]);
                }
                res.r += np * nn;
            }
            wave_sync();
            LQ_T(4)
            if (status_rows < 0 && res.r > 0) ok = w_reduce(res, rhs, true, s);
            LQ_T(5)
        }
        wave_sync();
        if (status_rows < 0 && in_lds) w_store(res, go);
        if (lane_id() == 0) {
            out_rows[b] = status_rows < 0 ? res.r : -status_rows;       // negative: did not fit, needs that many rows
            out_ok[b] = ok ? 1 : 0;
            if (chain_ok && !ok) chain_ok[b] = status_rows < 0 ? 0 : -status_rows;
        }
        wave_sync();
        LQ_T(6)
        LQ_FLUSH
    }
}

// ---- Gauss-Jordan family: rank / det / inv, one wave per matrix -----------------------------------
__device__ inline R32 abs_r(R32 v) { return lt(v, R32(0, 1)) ? neg(v) : v; }          // matt.h:206-212

__device__ inline void w_swap_rows(WMat & m, int a, int b)
{
    if (a != b)
        for (int j = lane_id(); j < m.c; j += wave_lanes()) {
            const R32 t = m.a[a * m.ld + j]; m.a[a * m.ld + j] = m.a[b * m.ld + j]; m.a[b * m.ld + j] = t;
        }
    wave_sync();
}
// pivot choice shared by rank/det/inv: first nonzero, a later exact 1 wins at once,
// otherwise the largest magnitude (matt.h:2644-2668, :1675-1695, :1796-1813).
__device__ inline int w_find_pivot_seq(const WMat & m, int col, int from)
{
    int swap_row = -1;
    R32 entry(0, 1);
    for (int k = from; k < m.r; k++) {
        const R32 t = m.a[k * m.ld + col];
        if (eq(t, R32(0, 1))) continue;
        if (swap_row == -1) { swap_row = k; entry = t; if (eq(entry, R32(1, 1))) break; }
        else if (eq(t, R32(1, 1))) { swap_row = k; entry = t; break; }
        else if (lt(abs_r(entry), abs_r(t))) { swap_row = k; entry = t; }
    }
    return swap_row;
}
// The same choice with one lane per row: the scan above ends at the first exact 1 if the column has one, and is
// otherwise a running maximum under strict replacement -- the earliest row of maximal |t|. lt() orders by value
// whenever the denominators are positive (every value the reference can hold); a column that shows any other
// denominator takes the sequential scan.
__device__ inline int w_find_pivot(const WMat & m, int col, int from, bool & unit_break)
{
    unit_break = false;
    const int L = wave_lanes(), lane = lane_id();
    int best = -1; R32 bestv(0, 1);
    for (int base = from; base < m.r; base += L) {
        const int k = base + lane;
        const bool in = k < m.r;
        const R32 t = in ? m.a[k * m.ld + col] : R32(0, 1);
        if (grp_ballot(t.den <= 0)) return w_find_pivot_seq(m, col, from);
        const bool nz = ne(t, R32(0, 1));
        const unsigned long long ones = grp_ballot(nz && eq(t, R32(1, 1)));
        if (ones) return base + __builtin_ctzll(ones);
        if (!grp_ballot(nz)) continue;
        R32 a = abs_r(t); int idx = nz ? k : -1;
        for (int o = 1; o < L; o <<= 1) {
            const R32 b(__shfl_xor(a.num, o), __shfl_xor(a.den, o)); const int bi = __shfl_xor(idx, o);
            const bool take = bi >= 0 && (idx < 0 || lt(a, b) || (!lt(b, a) && bi < idx));
            if (take) { a = b; idx = bi; }
        }
        if (best < 0 || lt(bestv, a)) { best = idx; bestv = a; }
    }
    return best;
}

// One elimination step of rank / det / inv: every row i of [lo, m.r) other than `row` whose entry e in `col` is not
// zero becomes row_i + t_i * row_row, with t_i = -e / pivot spelled the way each caller of mul_and_add_row spells it
// (KIND 0: div(neg(e), piv), matt.h:2689-2700; 1: neg(div(e, piv)), matt.h:1707-1718; 2: mul(-1, e), matt.h:1826-1838).
// The rows do not depend on one another, so instead of the reference's row after row the factors are computed one
// lane per row and the update runs one lane per CELL of the block of rows.
template <int KIND>
__device__ inline void w_eliminate(WMat & m, int row, int col, int lo, R32 * tfac, unsigned char * live)
{
    const R32 piv = m.a[row * m.ld + col];
    for (int i = lo + lane_id(); i < m.r; i += wave_lanes()) {
        const R32 e = m.a[i * m.ld + col];
        const bool on = i != row && ne(e, R32(0, 1));
        live[i] = on ? 1 : 0;
        if (on) tfac[i] = KIND == 0 ? q_div(m.cn, neg(e), piv) : (KIND == 1 ? neg(q_div(m.cn, e, piv)) : q_mul(m.cn, R32(-1, 1), e));
    }
    wave_sync();
    const int n = (m.r - lo) * m.c;
    const FastDiv by_c(m.c, n + wave_lanes());
    for (int x = lane_id(); x < n; x += wave_lanes()) {
        const int q = by_c.quot(x), i = lo + q, j = x - q * m.c;
        if (!live[i]) continue;
        m.a[i * m.ld + j] = q_fma(m.cn, m.a[i * m.ld + j], m.a[row * m.ld + j], tfac[i]);
    }
    wave_sync();
}

// Matrix<Rational>::rank(basis, is_unitarize) (matt.h:2614-2726). basis == NULL in the
// reference forces unitarize; rowpos (LDS, one int per row) tracks the row interchanges.
__device__ inline int w_rank(WMat & p, R32 * tfac, unsigned char * live, bool unitarize = true, int * rowpos = nullptr)
{
    int rankv = 0;
    for (int row = 0, col = 0; row < p.r && col < p.c; row++, col++) {
        int swap_row = -1; bool ub;
        for (int w = col; w < p.c; w++) {
            swap_row = w_find_pivot(p, w, row, ub);
            if (swap_row == -1) continue;
            w_swap_rows(p, swap_row, row);
            if (rowpos && lane_id() == 0) { const int t = rowpos[swap_row]; rowpos[swap_row] = rowpos[row]; rowpos[row] = t; }
            col = w;
            break;
        }
        if (swap_row == -1) break;
        const R32 d = p.a[row * p.ld + col];
        wave_sync();
        if (unitarize && ne(d, R32(1, 1))) w_scale_row(p, row, q_div(p.cn, R32(1, 1), d));
        w_eliminate<0>(p, row, col, unitarize ? 0 : row + 1, tfac, live);
        rankv++;
    }
    return rankv;
}

__device__ inline bool w_tri(const WMat & m, int which)
{
    const int n = m.r;
    bool ok = true;
    if (which == 0) { for (int j = 0; j < n && ok; j++) for (int i = j + 1; i < n && ok; i++) ok = eq(m.a[i * m.ld + j], R32(0, 1)); }
    else if (which == 1) { for (int i = 0; i < n && ok; i++) for (int j = i + 1; j < n && ok; j++) ok = eq(m.a[i * m.ld + j], R32(0, 1)); }
    else if (which == 2) { for (int i = 0; i < n && ok; i++) for (int j = 0; j < n - 1 - i && ok; j++) ok = eq(m.a[i * m.ld + j], R32(0, 1)); }
    else { for (int j = 0; j < n && ok; j++) for (int i = n - 1; i > n - 1 - j && ok; i--) ok = eq(m.a[i * m.ld + j], R32(0, 1)); }
    return ok;
}

// Matrix<Rational>::det (matt.h:1621-1736).
__device__ inline R32 w_det(WMat & a, R32 * tfac, unsigned char * live)
{
    const int n = a.r;
#define M_(i, j) a.a[(i) * a.ld + (j)]
    if (n == 1) return M_(0, 0);
    if (n == 2) return q_sub(a.cn, q_mul(a.cn, M_(0, 0), M_(1, 1)), q_mul(a.cn, M_(0, 1), M_(1, 0)));
    if (n == 3) {
        if (w_tri(a, 0) || w_tri(a, 1)) return q_mul(a.cn, q_mul(a.cn, M_(0, 0), M_(1, 1)), M_(2, 2));
        if (w_tri(a, 2) || w_tri(a, 3)) return q_mul(a.cn, q_mul(a.cn, q_mul(a.cn, M_(2, 0), M_(1, 1)), M_(0, 2)), R32(-1, 1));
        R32 d = q_mul(a.cn, q_mul(a.cn, M_(0, 0), M_(1, 1)), M_(2, 2));
        d = q_add(a.cn, d, q_mul(a.cn, q_mul(a.cn, M_(1, 0), M_(2, 1)), M_(0, 2)));
        d = q_add(a.cn, d, q_mul(a.cn, q_mul(a.cn, M_(0, 1), M_(1, 2)), M_(2, 0)));
        d = q_sub(a.cn, d, q_mul(a.cn, q_mul(a.cn, M_(0, 2), M_(1, 1)), M_(2, 0)));
        d = q_sub(a.cn, d, q_mul(a.cn, q_mul(a.cn, M_(0, 1), M_(1, 0)), M_(2, 2)));
        d = q_sub(a.cn, d, q_mul(a.cn, q_mul(a.cn, M_(2, 1), M_(1, 2)), M_(0, 0)));
        return d;
    }
    R32 d(1, 1);
    if (w_tri(a, 0) || w_tri(a, 1)) { for (int i = 0; i < n; i++) d = q_mul(a.cn, d, M_(i, i)); return d; }
    if (w_tri(a, 2) || w_tri(a, 3)) { for (int i = 0; i < n; i++) d = q_mul(a.cn, d, M_(i, n - 1 - i)); return d; }
    int swaps = 0;
    for (int j = 0; j < n; j++) {
        bool ub;
        const int swap_row = w_find_pivot(a, j, j, ub);
        if (swap_row == -1) return R32(0, 1);
        if (swap_row != j) { w_swap_rows(a, swap_row, j); swaps++; }
        w_eliminate<1>(a, j, j, j + 1, tfac, live);
    }
    for (int j = 0; j < n; j++) d = q_mul(a.cn, d, M_(j, j));
    if (swaps & 1) d = neg(d);
    return d;
#undef M_
}

// Matrix<Rational>::inv (matt.h:1743-1845) on the augmented matrix [p | e] (n x 2n) so
// that the row operations hit both halves in one lane-parallel pass.
__device__ inline bool w_inv(WMat & pe, int n, R32 * tfac, unsigned char * live)
{
#define P_(i, j) pe.a[(i) * pe.ld + (j)]
#define E_(i, j) pe.a[(i) * pe.ld + n + (j)]
    if (n == 1) { if (lane_id() == 0) E_(0, 0) = q_div(pe.cn, R32(1, 1), P_(0, 0)); wave_sync(); return true; }
    if (n == 2) {
        R32 k = q_sub(pe.cn, q_mul(pe.cn, P_(0, 0), P_(1, 1)), q_mul(pe.cn, P_(0, 1), P_(1, 0)));
        if (eq(k, R32(0, 1))) return false;
        k = q_div(pe.cn, R32(1, 1), k);
        wave_sync();
        if (lane_id() == 0) {
            R32 v[4] = { P_(1, 1), q_mul(pe.cn, R32(-1, 1), P_(0, 1)), q_mul(pe.cn, R32(-1, 1), P_(1, 0)), P_(0, 0) };
            const int mode = eq(k, R32(0, 1)) ? SCALE_ZERO : (eq(k, R32(1, 1)) ? SCALE_KEEP : SCALE_MUL);
            for (int t = 0; t < 4; t++) E_(t / 2, t % 2) = q_scaled(pe.cn, v[t], k, mode);
        }
        wave_sync();
        return true;
    }
    for (int j = 0; j < n; j++) {
        bool ub;
        WMat pv = pe; pv.c = n;                       // pivot search looks at the left half only
        const int swap_row = w_find_pivot(pv, j, j, ub);
        if (swap_row == -1) return false;
        w_swap_rows(pe, swap_row, j);
        const R32 d = P_(j, j);
        wave_sync();
        if (ne(d, R32(1, 1))) w_scale_row(pe, j, q_div(pe.cn, R32(1, 1), d));
        w_eliminate<2>(pe, j, j, 0, tfac, live);
    }
    return true;
#undef P_
#undef E_
}

// op 0: rank (rows x cols), 1: det (n x n), 2: inv (n x n -> out n x n), 3: rank with basis
// (flag = is_unitarize; out_mat rows x cols, rows past the basis left zero), 4: null space
// (out_mat cols x cols, matt.h:2546-2584). One wave per matrix.
__global__ __launch_bounds__(64) void k_gauss_batch(int nb, const R32 * mats, int rows, int cols, int op, int flag,
                                                    int * out_int, R32 * out_val, R32 * out_mat, int sys_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    unsigned char * lds = lds_all + (size_t)threadIdx.y * sys_lds;
    WMat m; m.a = (R32 *)lds;
    // LDS of one system: [matrix, twice as wide for inv] [row factors: rows x R32] [rowpos: rows ints] [live: rows bytes]
    R32 * tfac = (R32 *)(lds + (((size_t)rows * cols * 8 * (op == 2 ? 2 : 1) + 15) & ~(size_t)15));
    int * rowpos = (int *)(tfac + rows);
    unsigned char * live = (unsigned char *)(rowpos + rows);
    for (int b = sys_first(); b < nb; b += sys_stride()) {
        const R32 * g = mats + (size_t)b * rows * cols;
        if (op == 2) {
            const int n = rows;
            m.r = n; m.c = 2 * n; m.ld = 2 * n;
            bool bad = false;
            for (int t = lane_id(); t < n * n; t += wave_lanes()) {
                const R32 v = g[t];
                bad |= !canonical(v);
                m.a[(t / n) * m.ld + (t % n)] = v;
                m.a[(t / n) * m.ld + n + (t % n)] = (t / n == t % n && n > 2) ? R32(1, 1) : R32(0, 1);
            }
            m.cn = grp_ballot(bad) == 0;
            wave_sync();
            const bool ok = w_inv(m, n, tfac, live);
            wave_sync();
            if (ok) for (int t = lane_id(); t < n * n; t += wave_lanes()) out_mat[(size_t)b * n * n + t] = m.a[(t / n) * m.ld + n + (t % n)];
            if (lane_id() == 0) out_int[b] = ok ? 1 : 0;
        } else if (op == 3) {
            m.ld = cols;
            w_load(m, g, rows, cols);
            for (int t = lane_id(); t < rows; t += wave_lanes()) rowpos[t] = t;
            wave_sync();
            const bool unit = flag != 0;
            const int rk = w_rank(m, tfac, live, unit, rowpos);
            wave_sync();
            R32 * o = out_mat + (size_t)b * rows * cols;
            if (!unit && rk < rows) {           // the original rows in pivot order (matt.h:2710-2719)
                for (int t = lane_id(); t < rk * cols; t += wave_lanes()) o[t] = g[rowpos[t / cols] * cols + t % cols];
            } else {
                for (int t = lane_id(); t < rows * cols; t += wave_lanes()) o[t] = m.a[t];
            }
            if (lane_id() == 0) out_int[b] = rk;
        } else if (op == 4) {
            m.ld = cols;
            w_load(m, g, rows, cols);
            w_rank(m, tfac, live, true, nullptr);
            wave_sync();
            R32 * ns = out_mat + (size_t)b * cols * cols;
            for (int t = lane_id(); t < cols * cols; t += wave_lanes()) ns[t] = (t / cols == t % cols) ? R32(1, 1) : R32(0, 1);
            wave_sync();
            for (int row = 0; row < rows; row++) {
                int col = row;
                while (col < cols && eq(m.a[row * cols + col], R32(0, 1))) col++;
                if (col >= cols) break;
                for (int k = col + lane_id(); k < cols; k += wave_lanes())
                    ns[col * cols + k] = (k == col) ? R32(0, 1) : neg(m.a[row * cols + k]);
                wave_sync();
            }
        } else {
            m.ld = cols;
            w_load(m, g, rows, cols);
            if (op == 0) { const int r = w_rank(m, tfac, live); if (lane_id() == 0) out_int[b] = r; }
            else { const R32 d = (rows == cols) ? w_det(m, tfac, live) : R32(0, 1); if (lane_id() == 0) out_val[b] = d; }
        }
        wave_sync();
    }
}

// ---- INTMat::hnf / gcd (xmat.cpp:853-1030): INT is the ring Z/2^32 ---------------------------------
__device__ __forceinline__ int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int iabs32(int a) { return a < 0 ? (int)(0u - (unsigned)a) : a; }

// exgcd (comf.cpp:295-321): the reference recurses; here the quotients are kept and unwound.
__device__ inline int d_exgcd(int a, int b, int & x, int & y)
{
    int q[48];
    int depth = 0;
    while (b != 0 && depth < 48) { q[depth++] = a / b; const int t = a % b; a = b; b = t; }
    int g = a;
    x = 1; y = 0;
    while (depth > 0) {
        const int x1 = x, y1 = y;
        depth--;
        x = y1;
        y = (int)((unsigned)x1 - (unsigned)q[depth] * (unsigned)y1);
    }
    if (g < 0) { g = -g; x = -x; y = -y; }
    return g;
}

// INTMat::hnf (xmat.cpp:912-992). h (rows x cols) and u (cols x cols) are stacked into one
// (rows + cols) x cols LDS matrix: every elimination matrix of the reference differs from the
// identity in at most two columns, so `h = h * elim; u = u * elim` is one column operation over
// the stacked rows, one lane per row. In Z/2^32 that equals the full product bit for bit.
// status[b] = 0, or XPG_ERR_REF_UNDEFINED (-7) where the reference divides by zero or reads its
// mis-shaped rows x cols "identity" out of bounds (xmat.cpp:936-941, :956-980).
__global__ __launch_bounds__(64) void k_hnf_batch(int nb, const int * mats, int rows, int cols, int * hs, int * us,
                                                  int * status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int * M = (int *)lds;
    const int tot = rows + cols, lim = rows < cols ? rows : cols, lane = lane_id();
    const int IMIN = (int)0x80000000;
#define H_(i, j) M[(i) * cols + (j)]
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const int * g = mats + (size_t)b * rows * cols;
        for (int t = lane; t < tot * cols; t += wave_lanes()) {
            const int r = t / cols, c = t % cols;
            M[t] = r < rows ? g[t] : (r - rows == c ? 1 : 0);
        }
        wave_sync();
        int st = 0;
        for (int i = 0; i < lim && st == 0; i++) {
            for (int j = i + 1; j < cols; j++) {                      // 1. clear row i right of the diagonal
                const int aii = H_(i, i), aij = H_(i, j);
                wave_sync();
                if (aij == 0) continue;
                if (aii == IMIN || aij == IMIN) { st = -7; break; }
                int x, y;
                const int gg = d_exgcd(aii, aij, x, y);               // gen_elim_mat, xmat.cpp:853-868
                const int p = -aij / gg, q = aii / gg;
                for (int t = lane; t < tot; t += wave_lanes()) {
                    const int a = H_(t, i), c = H_(t, j);
                    H_(t, i) = wadd(wmul(a, x), wmul(c, y));
                    H_(t, j) = wadd(wmul(a, p), wmul(c, q));
                }
                wave_sync();
            }
            if (st) break;
            const int dg = H_(i, i);
            wave_sync();
            if (dg < 0) {                                             // 2. positive diagonal
                if (cols > rows) { st = -7; break; }
                for (int t = lane; t < tot; t += wave_lanes()) H_(t, i) = wmul(H_(t, i), -1);
                wave_sync();
            }
            for (int j = 0; j < i; j++) {                             // 3. non-negative left of the diagonal
                const int hij = H_(i, j), hii = H_(i, i);
                wave_sync();
                if (hij >= 0) continue;
                if (hii == 0 || hij == IMIN) { st = -7; break; }
                const int v = iabs32(hij) <= iabs32(hii) ? 1 : iabs32(hij / hii) + 1;
                for (int t = lane; t < tot; t += wave_lanes()) H_(t, j) = wadd(H_(t, j), wmul(H_(t, i), v));
                wave_sync();
            }
            if (st) break;
            for (int j = 0; j < i; j++) {                             // 4. smaller than the diagonal
                const int hij = H_(i, j), hii = H_(i, i);
                wave_sync();
                if (hij < hii) continue;
                if (hii == 0) { st = -7; break; }
                const int d = hij / hii;
                for (int t = lane; t < tot; t += wave_lanes()) H_(t, j) = wadd(H_(t, j), wmul(H_(t, i), -d));
                wave_sync();
            }
        }
        wave_sync();
        if (st == 0) {
            for (int t = lane; t < rows * cols; t += wave_lanes()) hs[(size_t)b * rows * cols + t] = M[t];
            for (int t = lane; t < cols * cols; t += wave_lanes()) us[(size_t)b * cols * cols + t] = M[rows * cols + t];
        }
        if (lane == 0) status[b] = st;
        wave_sync();
    }
#undef H_
}

// INTMat::gcd (xmat.cpp:996-1030): one thread per row, rows of all matrices side by side.
__global__ __launch_bounds__(256) void k_int_gcd_batch(long long total_rows, int * mats, int cols)
{
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= total_rows || cols == 1) return;
    int * r = mats + row * cols;
    unsigned mn = 0xFFFFFFFFu;
    bool allzero = true;
    for (int j = 0; j < cols; j++) {
        const unsigned x = (unsigned)iabs32(r[j]);
        if (x != 0) { mn = mn < x ? mn : x; allzero = false; }
    }
    if (mn == 1 || mn == 0 || allzero) return;
    unsigned g = mn;
    for (int j = 0; j < cols; j++) {
        const unsigned q = (unsigned)iabs32(r[j]);
        if (q != 0 && q != g) {
            int x = (int)g, y = (int)q;                               // sgcd, comf.cpp:226-243
            if (x < 0) x = -x;
            if (y < 0) y = -y;
            if (x > y) { const int t = x; x = y; y = t; }
            while (x) { const int t = x; x = y % x; y = t; }
            g = (unsigned)y;
            if (g == 1) break;
        }
    }
    if (g == 1) return;
    for (int j = 0; j < cols; j++) r[j] = r[j] / (int)g;
}

// ---- packed results (host-array boundary of fme / calcBound): row offsets, then live rows only -------------------
// One workgroup: off[b] = sum of max(rows[k], 0) for k < b, off[nb] = the total (exclusive scan, 64-bit).
__global__ __launch_bounds__(1024) void k_rows_scan(int nb, const int * __restrict__ rows, long long * __restrict__ off)
{
    __shared__ long long sh[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int b = base + tid;
        const long long x = b < nb ? (rows[b] > 0 ? rows[b] : 0) : 0;
        long long incl = x;                                       // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const long long y = ((long long)__shfl_up((int)(incl >> 32), o) << 32) | (unsigned)__shfl_up((int)(unsigned)incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) sh[wv] = incl;
        __syncthreads();
        long long before = carry_s;
        for (int k = 0; k < wv; k++) before += sh[k];
        if (b < nb) off[b] = before + incl - x;
        __syncthreads();
        if (tid == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (tid == 0) off[nb] = carry_s;
}
// System b's live rows (rows[b] of them, at the front of its cap_rows-row slot) to packed[off[b] ..]: 16 bytes per lane.
__global__ __launch_bounds__(256) void k_pack_rows(int nb, const R32 * __restrict__ slots, int cap_rows, int cols,
                                                   const int * __restrict__ rows, const long long * __restrict__ off,
                                                   R32 * __restrict__ packed)
{
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const int r = rows[b];
        if (r <= 0) continue;
        const size_t cells = (size_t)r * cols;
        const R32 * src = slots + (size_t)b * cap_rows * cols;
        R32 * dst = packed + (size_t)off[b] * cols;
        // slot bases are 16-byte aligned when cap_rows * cols is even, destinations when off[b] * cols is: else by cell
        if ((((size_t)src | (size_t)dst) & 15) == 0) {
            const size_t pairs = cells >> 1;
            const uint4 * s4 = reinterpret_cast<const uint4 *>(src);
            uint4 * d4 = reinterpret_cast<uint4 *>(dst);
            for (size_t t = threadIdx.x; t < pairs; t += blockDim.x) d4[t] = s4[t];
            if ((cells & 1) && threadIdx.x == 0) dst[cells - 1] = src[cells - 1];
        } else {
            for (size_t t = threadIdx.x; t < cells; t += blockDim.x) dst[t] = src[t];
        }
    }
}

} // namespace xpg
