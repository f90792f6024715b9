// Host launch wrappers of lineq_kernels.hip.h: stage the caller's host arrays
// through HBM, one wavefront per system, results back. No arithmetic here.
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include "ctx.hip.h"
#include "lineq_kernels.hip.h"

namespace xpg {


// mode 0: removeIdenRow, 1: reduce -- in place on mats[nb][rows][cols]. The *_dev forms take device arrays and
// only enqueue (the caller synchronises the handle's stream); the host-array forms stage through them.
inline int lineq_reduce_batch_dev(xpg_ctx * ctx, int nb, R32 * d_mats, int rows, int cols, int rhs, int mode,
                                  int is_intersect, int32_t * d_rows, int32_t * d_ok)
{
    if (!ctx || nb < 0 || !d_mats || rows <= 0 || cols <= 0 || !d_rows || !d_ok || (mode == 1 && (rhs < 0 || rhs >= cols)))
        return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const size_t lds = lineq_lds_bytes(rows, cols);
    if (lds > 160 * 1024 || rows > 32767) return XPG_ERR_UNSUPPORTED;
    const LineqGeom q = lineq_geom(nb, rows > cols ? rows : cols, lds);
    XPG_TRY(lds_limit((const void *)k_reduce_batch, ctx->device, q.lds));
    hipLaunchKernelGGL(k_reduce_batch, dim3(q.grid), q.block, q.lds, ctx->stream, nb, d_mats, rows,
                       cols, rhs, mode, is_intersect, (int *)d_rows, (int *)d_ok, q.sys_lds);
    XPG_TRY(hipGetLastError());
    return 0;
}
// Pinned host memory of the packed-result entry points: grows, never shrinks, freed with the handle.
// keep: the first `keep` bytes survive a growth (the chunked pipeline below appends while earlier chunks' rows are in it).
inline int hpack_reserve(xpg_ctx * ctx, size_t bytes, size_t keep = 0)
{
    if (bytes <= ctx->hpack_cap) return 0;
    const size_t cap = bytes + bytes / 4 + 4096;
    void * fresh = 0;
    if (hipHostMalloc(&fresh, cap, hipHostMallocDefault) != hipSuccess) { ctx->err = "hipHostMalloc(packed results)"; return XPG_ERR_ALLOC; }
    if (ctx->hpack && keep) memcpy(fresh, ctx->hpack, keep < ctx->hpack_cap ? keep : ctx->hpack_cap);
    if (ctx->hpack) (void)hipHostFree(ctx->hpack);
    ctx->hpack = fresh; ctx->hpack_cap = cap;
    return 0;
}

// Lineq::reduce / removeIdenRow for nb host systems with a PACKED result (round 6): row_offsets[nb + 1] (in rows) and the
// surviving rows of every system back to back. The systems go up once (small batches through the handle's pinned buffer,
// large ones straight from the caller's pages), are reduced in place in HBM, and k_pack_rows_out writes the survivors
// STRAIGHT INTO the handle's pinned host buffer (hipHostMalloc memory is mapped on the device: coalesced 16-byte stores
// over the link), so the row counts, the verdicts, the offsets and the rows are all there after ONE synchronisation -- no
// device slots come back, nothing is allocated per call beyond the handle's cached blocks. `out` (may be NULL) receives a
// copy of the rows, *view (may be NULL) the pinned buffer itself (valid until the handle's next packed call), out_rows
// (may be NULL) the per-system counts.
inline int lineq_reduce_batch_packed(xpg_ctx * ctx, int nb, const R32 * mats, int rows, int cols, int rhs, int mode, int is_intersect,
                                     R32 * out, long long out_cap_rows, const R32 ** view, long long * row_offsets,
                                     int32_t * out_rows, int32_t * out_ok)
{
    if (view) *view = 0;
    if (!ctx || nb < 0 || !mats || rows <= 0 || cols <= 0 || !row_offsets || (mode == 1 && (rhs < 0 || rhs >= cols || !out_ok)) ||
        (out && out_cap_rows < 0))
        return XPG_ERR_SHAPE;
    if (nb == 0) { row_offsets[0] = 0; return 0; }
    const size_t bi = (size_t)nb * rows * cols * 8;
    const size_t meta = (size_t)(nb + 1) * 8 + (size_t)nb * 8;       // offsets, then ok and rows
    const size_t meta_al = (meta + 255) & ~(size_t)255;
    const bool small_in = bi <= ((size_t)4 << 20);
    DevBuf di, dr, dk, doff;
    XPG_TRY(di.alloc(ctx, bi)); XPG_TRY(dr.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dk.alloc(ctx, (size_t)nb * 4));
    XPG_TRY(doff.alloc(ctx, (size_t)(nb + 1) * 8));
    int rc = hpack_reserve(ctx, meta_al + (small_in ? bi : 0) + bi);
    if (rc) return rc;
    char * hp = (char *)ctx->hpack;
    char * hrows_out = hp + meta_al + (small_in ? bi : 0);           // where the survivors land
    if (small_in) {
        memcpy(hp + meta_al, mats, bi);
        XPG_TRY(hipMemcpyAsync(di.p, hp + meta_al, bi, hipMemcpyHostToDevice, ctx->stream));
    } else {
        XPG_TRY(hipMemcpyAsync(di.p, mats, bi, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = lineq_reduce_batch_dev(ctx, nb, (R32 *)di.p, rows, cols, rhs, mode, is_intersect, (int32_t *)dr.p, (int32_t *)dk.p);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, ctx->stream, nb, (const int *)dr.p, (long long *)doff.p);
    hipLaunchKernelGGL(k_pack_rows, dim3(nb < 4096 ? nb : 4096), dim3(256), 0, ctx->stream, nb, (const R32 *)di.p, rows, cols,
                       (const int *)dr.p, (const long long *)doff.p, (R32 *)hrows_out);
    XPG_TRY(hipGetLastError());
    long long * h_off = (long long *)hp; int32_t * h_ok = (int32_t *)(hp + (size_t)(nb + 1) * 8); int32_t * h_rows = h_ok + nb;
    XPG_TRY(hipMemcpyAsync(h_off, doff.p, (size_t)(nb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(h_rows, dr.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mode == 1) XPG_TRY(hipMemcpyAsync(h_ok, dk.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(row_offsets, h_off, (size_t)(nb + 1) * 8);
    if (out_rows) memcpy(out_rows, h_rows, (size_t)nb * 4);
    if (out_ok && mode == 1) memcpy(out_ok, h_ok, (size_t)nb * 4);
    const long long total = row_offsets[nb];
    if (out && out_cap_rows < total) return XPG_ERR_SHAPE;
    if (out && total > 0) memcpy(out, hrows_out, (size_t)total * cols * 8);
    if (view && total > 0) *view = (const R32 *)hrows_out;
    return 0;
}
// The in-place form of the reference's signature (Lineq::reduce(m, ...) overwrites m): the packed call above, then the
// survivors of system b copied to the front of its slot. Rows of a slot behind out_rows[b] keep the caller's input.
inline int lineq_reduce_batch(xpg_ctx * ctx, int nb, R32 * mats, int rows, int cols, int rhs, int mode,
                              int is_intersect, int32_t * out_rows, int32_t * out_ok)
{
    if (!ctx || nb < 0 || !mats || rows <= 0 || cols <= 0 || !out_rows || (mode == 1 && (rhs < 0 || rhs >= cols)))
        return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    std::vector<long long> off((size_t)nb + 1);
    std::vector<int32_t> ok_tmp;
    if (mode == 1 && !out_ok) { ok_tmp.resize((size_t)nb); out_ok = ok_tmp.data(); }
    const R32 * view = 0;
    const int rc = lineq_reduce_batch_packed(ctx, nb, mats, rows, cols, rhs, mode, is_intersect, (R32 *)0, 0, &view, off.data(), out_rows, out_ok);
    if (rc) return rc;
    const size_t rowb = (size_t)cols * 8, slot = (size_t)rows * rowb;
    for (int b = 0; b < nb; b++)
        if (out_rows[b] > 0) memcpy((char *)mats + (size_t)b * slot, (const char *)view + (size_t)off[(size_t)b] * rowb, (size_t)out_rows[b] * rowb);
    return 0;
}

// LDS of one system in k_fme_batch and the rows of its result area. Occupancy beats residency here: the result of
// a system is read and written a few times per row (pair sums, hash, classification, compaction) and its HBM slot
// is L2-resident, while every KB of LDS per system is fewer systems per CU for a kernel that waits on dependent
// gcd chains. Measured at 40 x 13 (16 384 systems): whole cap-row result in LDS (54 KB, 2 per CU) 8.0 M systems/s;
// a result area for the typical result in 32 KB 14.6 M; in 20 KB 20.2 M; NO result area (9 KB: the normalised
// input and the scratch) 27.9 M. So the result stays in LDS only while the whole layout is small (<= 12 KB: the
// 16 x 9 class), and otherwise the area holds just the row factors of the normalisation.
struct FmeLds { int cap_lds; size_t lds; };
inline FmeLds fme_lds(int cap, int cap_in, int cols)
{
    static const int full_kb = [] { const char * e = xpg_hook("XPG_FME_FULL_KB"); return e ? atoi(e) : 12; }();   // A/B knob
    const int capx = cap > cap_in ? cap : cap_in;
    const size_t scratch = lineq_lds_bytes(capx, cols) - (size_t)capx * cols * 8 + 16;
    const size_t tmp = (size_t)cap_in * cols * 8;
    const size_t full = (size_t)cap * cols * 8 + tmp + scratch;
    FmeLds f;
    if (full <= (size_t)full_kb * 1024) { f.cap_lds = cap; f.lds = full; return f; }
    f.cap_lds = (cap_in + cols - 1) / cols;                       // room for one factor per input row
    if (f.cap_lds > cap) f.cap_lds = cap;
    f.lds = (size_t)f.cap_lds * cols * 8 + tmp + scratch;
    return f;
}
// d_outs must be zeroed by the caller where it wants zeros past a result's last row (the host form does).
inline int lineq_fme_batch_dev(xpg_ctx * ctx, int nb, const R32 * d_mats, int rows, int cols, int rhs, int u,
                               int darkshadow, R32 * d_outs, int cap, int32_t * d_rows, int32_t * d_ok)
{
    if (!ctx || nb < 0 || !d_mats || !d_outs || rows <= 0 || cols <= 1 || rhs < 1 || rhs >= cols || u < 0 || u >= rhs ||
        cap < rows || !d_rows || !d_ok)
        return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const FmeLds fl = fme_lds(cap, rows, cols);
    if (fl.lds > 160 * 1024 || cap > 32767) return XPG_ERR_UNSUPPORTED;
    const LineqGeom q = lineq_geom(nb, 64, fl.lds);
    XPG_TRY(lds_limit((const void *)k_fme_batch, ctx->device, q.lds));
    hipLaunchKernelGGL(k_fme_batch, dim3(q.grid), q.block, q.lds, ctx->stream, nb, d_mats, rows,
                       (const int *)0, cols, rhs, u, darkshadow, d_outs, cap, (int *)d_rows, (int *)d_ok,
                       fl.cap_lds, (int *)0, q.sys_lds);
    XPG_TRY(hipGetLastError());
    return 0;
}
inline int lineq_fme_batch(xpg_ctx * ctx, int nb, const R32 * mats, int rows, int cols, int rhs, int u,
                           int darkshadow, R32 * outs, int cap, int32_t * out_rows, int32_t * out_ok)
{
    if (!ctx || nb < 0 || !mats || !outs || rows <= 0 || cols <= 1 || cap < rows || !out_rows || !out_ok) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const size_t bi = (size_t)nb * rows * cols * 8, bo = (size_t)nb * cap * cols * 8;
    DevBuf di, dout, dr, dk;
    XPG_TRY(di.alloc(ctx, bi)); XPG_TRY(dout.alloc(ctx, bo)); XPG_TRY(dr.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dk.alloc(ctx, (size_t)nb * 4));
    XPG_TRY(hipMemcpyAsync(di.p, mats, bi, hipMemcpyHostToDevice, ctx->stream));
    XPG_TRY(hipMemsetAsync(dout.p, 0, bo, ctx->stream));
    const int rc = lineq_fme_batch_dev(ctx, nb, (const R32 *)di.p, rows, cols, rhs, u, darkshadow, (R32 *)dout.p, cap,
                                       (int32_t *)dr.p, (int32_t *)dk.p);
    if (rc) return rc;
    XPG_TRY(hipMemcpyAsync(outs, dout.p, bo, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(out_rows, dr.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(out_ok, dk.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// The packed call for LARGE batches as a pipeline of chunks on two streams (round 4): chunk k goes up, is eliminated, scanned,
// packed and comes down on stream k & 1, so the upload of chunk k + 1 runs while chunk k's rows come down -- the link is
// full duplex, and a single stream uses it one way at a time (2.24 M systems/s at 40 x 13 against 3.1 M if up and down
// were serialised at the link's rate). Opt-in (XPG_FME_CHUNKS): measured no faster than the single stream, see the caller. Row offsets are global: a chunk's base is the total of the chunks before it, known
// when their (small) offset arrays have come down; the rows land behind each other in the pinned buffer (`view`) or `out`.
inline int lineq_fme_batch_packed_chunked(xpg_ctx * ctx, int nb, const R32 * mats, int rows, int cols, int rhs, int u, int darkshadow,
                                          int cap, R32 * out, long long out_cap_rows, const R32 ** view, long long * row_offsets,
                                          int32_t * out_ok, int nch)
{
    if (ctx->lanes.empty()) {                                      // the second stream: a lane handle on the same device
        xpg_ctx * l = 0;
        const int rc = xpg_create(&l, ctx->device);
        if (rc) return rc;
        ctx->lanes.push_back(l);
    }
    xpg_ctx * cs[2] = { ctx, ctx->lanes[0] };
    // device buffers for the WHOLE batch, as the single-stream path has them (they come from and go back to the handle's cache:
    // a buffer per chunk would thrash its 16 blocks); the packed rows of a chunk pass through one of two buffers, one per stream
    struct Chunk { int lo, n; long long * h_off; int32_t * h_ok; int32_t * h_rows; hipEvent_t ev; };
    std::vector<Chunk> ch((size_t)nch);
    DevBuf di, dout, dr, dk, doff, dpk[2];
    {
        const size_t bi_all = (size_t)nb * rows * cols * 8, bo_all = (size_t)nb * cap * cols * 8;
        XPG_TRY(di.alloc(ctx, bi_all)); XPG_TRY(dout.alloc(ctx, bo_all)); XPG_TRY(dr.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dk.alloc(ctx, (size_t)nb * 4));
        XPG_TRY(doff.alloc(ctx, (size_t)(nb + nch) * 8));
    }
    // pinned meta of all chunks: (n + 1) offsets, n ok, n rows each -- in the batch staging buffer (grow-only, pinned)
    const size_t meta_bytes = (size_t)(nb + nch) * 8 + (size_t)nb * 8 + 64 * (size_t)nch;
    if (meta_bytes > ctx->hstage_cap) {
        if (ctx->hstage) (void)hipHostFree(ctx->hstage);
        ctx->hstage = 0; ctx->hstage_cap = 0;
        if (hipHostMalloc(&ctx->hstage, meta_bytes + 4096, hipHostMallocDefault) != hipSuccess) { ctx->err = "hipHostMalloc(fme meta)"; return XPG_ERR_ALLOC; }
        ctx->hstage_cap = meta_bytes + 4096;
    }
    char * hm = (char *)ctx->hstage;
    const size_t rowb = (size_t)cols * 8;
    int rc = 0;
    for (int k = 0; k < nch; k++) {
        Chunk & c = ch[(size_t)k];
        c.lo = (int)((long long)nb * k / nch); c.n = (int)((long long)nb * (k + 1) / nch) - c.lo;
        c.h_off = (long long *)hm; hm += (size_t)(c.n + 1) * 8;
        c.h_ok = (int32_t *)hm; hm += (size_t)c.n * 4; c.h_rows = (int32_t *)hm; hm += (size_t)c.n * 4;
        hm = (char *)(((uintptr_t)hm + 15) & ~(uintptr_t)15);
        c.ev = 0;
    }
    auto fail = [&](int code) {                                    // (every chunk's stream is idle before its buffers go back to the cache)
        (void)hipStreamSynchronize(cs[0]->stream); (void)hipStreamSynchronize(cs[1]->stream);
        for (auto & c : ch) if (c.ev) (void)hipEventDestroy(c.ev);
        return code;
    };
    auto first_half = [&](int k) -> int {
        Chunk & c = ch[(size_t)k];
        xpg_ctx * x = cs[k & 1];
        const size_t bi = (size_t)c.n * rows * cols * 8;
        R32 * d_in = (R32 *)di.p + (size_t)c.lo * rows * cols; R32 * d_out = (R32 *)dout.p + (size_t)c.lo * cap * cols;
        int32_t * d_r = (int32_t *)dr.p + c.lo; int32_t * d_k = (int32_t *)dk.p + c.lo; long long * d_off = (long long *)doff.p + c.lo + k;
        XPG_TRY(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
        XPG_TRY(hipMemcpyAsync(d_in, mats + (size_t)c.lo * rows * cols, bi, hipMemcpyHostToDevice, x->stream));
        const int r = lineq_fme_batch_dev(x, c.n, d_in, rows, cols, rhs, u, darkshadow, d_out, cap, d_r, d_k);
        if (r) { if (x != ctx) ctx->err = x->err; return r; }
        hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, x->stream, c.n, (const int *)d_r, d_off);
        XPG_TRY(hipMemcpyAsync(c.h_off, d_off, (size_t)(c.n + 1) * 8, hipMemcpyDeviceToHost, x->stream));
        XPG_TRY(hipMemcpyAsync(c.h_ok, d_k, (size_t)c.n * 4, hipMemcpyDeviceToHost, x->stream));
        XPG_TRY(hipMemcpyAsync(c.h_rows, d_r, (size_t)c.n * 4, hipMemcpyDeviceToHost, x->stream));
        XPG_TRY(hipEventRecord(c.ev, x->stream));
        return 0;
    };
    for (int k = 0; k < nch && k < 2; k++) if ((rc = first_half(k))) return fail(rc);
    long long base = 0;
    bool fits = true;
    for (int k = 0; k < nch; k++) {
        Chunk & c = ch[(size_t)k];
        xpg_ctx * x = cs[k & 1];
        if (hipEventSynchronize(c.ev) != hipSuccess) { ctx->err = "hipEventSynchronize(fme chunk)"; return fail(XPG_ERR_HIP); }
        for (int b = 0; b < c.n; b++)
            if (c.h_rows[b] < 0) { ctx->err = "fme: a result needs more rows than cap_rows"; row_offsets[0] = c.h_rows[b]; return fail(XPG_ERR_UNSUPPORTED); }
        for (int b = 0; b < c.n; b++) row_offsets[c.lo + b] = base + c.h_off[b];
        memcpy(out_ok + c.lo, c.h_ok, (size_t)c.n * 4);
        const long long total = c.h_off[c.n];
        if (out && base + total > out_cap_rows) fits = false;
        if (total > 0 && (view || (out && fits))) {
            const size_t bp = (size_t)total * rowb, at = (size_t)base * rowb;
            DevBuf & pk = dpk[k & 1];
            if (bp > pk.cap) {                                     // (the chunk two back used it on this stream: idle before it goes back to the cache)
                (void)hipStreamSynchronize(x->stream);
                DevBuf bigger;
                if ((rc = [&]() -> int { XPG_TRY(bigger.alloc(ctx, bp + bp / 4)); return 0; }())) return fail(rc);
                std::swap(pk.p, bigger.p); std::swap(pk.cap, bigger.cap); std::swap(pk.owner, bigger.owner);
            }
            hipLaunchKernelGGL(k_pack_rows, dim3(c.n < 4096 ? c.n : 4096), dim3(256), 0, x->stream, c.n,
                               (const R32 *)dout.p + (size_t)c.lo * cap * cols, cap, cols,
                               (const int *)((int32_t *)dr.p + c.lo), (const long long *)((long long *)doff.p + c.lo + k), (R32 *)pk.p);
            char * dst;
            if (view) {
                if (at + bp > ctx->hpack_cap) {                    // grow: nothing may be in flight into the old buffer
                    (void)hipStreamSynchronize(cs[0]->stream); (void)hipStreamSynchronize(cs[1]->stream);
                    if ((rc = hpack_reserve(ctx, (at + bp) * (size_t)nch / (size_t)(k + 1), at))) return fail(rc);
                }
                dst = (char *)ctx->hpack + at;
            } else dst = (char *)out + at;
            if (hipMemcpyAsync(dst, pk.p, bp, hipMemcpyDeviceToHost, x->stream) != hipSuccess) { ctx->err = "hipMemcpyAsync(fme rows)"; return fail(XPG_ERR_HIP); }
        }
        base += total;
        if (k + 2 < nch && (rc = first_half(k + 2))) return fail(rc);
    }
    row_offsets[nb] = base;
    if (hipStreamSynchronize(cs[0]->stream) != hipSuccess || hipStreamSynchronize(cs[1]->stream) != hipSuccess) { ctx->err = "hipStreamSynchronize(fme chunks)"; return fail(XPG_ERR_HIP); }
    for (auto & c : ch) if (c.ev) { (void)hipEventDestroy(c.ev); c.ev = 0; }
    if (out && !fits) return XPG_ERR_SHAPE;
    if (view) {
        if (out && base > 0) memcpy(out, ctx->hpack, (size_t)base * rowb);
        *view = base > 0 ? (const R32 *)ctx->hpack : (const R32 *)0;
    }
    return 0;
}

// Lineq::fme for nb systems with a PACKED result: row_offsets[nb + 1] (rows, not bytes) and the
// live rows of every system back to back, instead of nb worst-case slots of cap_rows rows (the typical result is a
// third of the worst case: 751 MB of slots against 267 MB of rows for 16 384 systems of 40 x 13). The systems go up and
// the rows come down through pinned memory; `out` (optional) receives a copy, *view (optional) the pinned buffer itself,
// valid until the next packed call on this handle. Returns 0; XPG_ERR_SHAPE when `out` holds fewer than the
// row_offsets[nb] rows needed (row_offsets is filled: the caller can size its buffer and call again).
inline int lineq_fme_batch_packed(xpg_ctx * ctx, int nb, const R32 * mats, int rows, int cols, int rhs, int u, int darkshadow,
                                  int cap, R32 * out, long long out_cap_rows, const R32 ** view, long long * row_offsets,
                                  int32_t * out_ok)
{
    if (!ctx || nb < 0 || !mats || rows <= 0 || cols <= 1 || !row_offsets || !out_ok || (out && out_cap_rows < 0)) return XPG_ERR_SHAPE;
    if (view) *view = 0;
    if (nb == 0) { row_offsets[0] = 0; return 0; }
    if (cap <= 0) cap = rows * rows / 4 + rows + 1;                  // every (positive, negative) pair + the rows without u
    if (cap < rows) cap = rows;
    const size_t bi = (size_t)nb * rows * cols * 8, bo = (size_t)nb * cap * cols * 8;
    // XPG_FME_CHUNKS=n (n > 1): large batches through the two-stream chunk pipeline above. Measured and NOT the default: with
    // the input in the caller's pageable memory the runtime stages every upload itself and the two directions do not overlap --
    // 16 384 systems of 40 x 13: 2.25 M systems/s on one stream, 1.78 / 1.71 / 2.20 M with 4 / 8 / 16 chunks; 60 x 20: 0.757 M
    // against 0.621 / 0.592 / 0.811 M (tools/lab/run_fme_chunks_ab.sh).
    static const int chunks_env = [] { const char * e = xpg_hook("XPG_FME_CHUNKS"); return e ? atoi(e) : 0; }();
    if ((out || view) && nb >= 2048 && bi + bo / 3 >= ((size_t)96 << 20) && chunks_env > 1) {
        int nch = chunks_env;
        if (nch > 16) nch = 16;
        return lineq_fme_batch_packed_chunked(ctx, nb, mats, rows, cols, rhs, u, darkshadow, cap, out, out_cap_rows, view, row_offsets, out_ok, nch);
    }
    const size_t meta = (size_t)(nb + 1) * 8 + (size_t)nb * 8;       // offsets, then ok and rows
    DevBuf di, dout, dr, dk, doff, dpk;
    XPG_TRY(di.alloc(ctx, bi)); XPG_TRY(dout.alloc(ctx, bo)); XPG_TRY(dr.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dk.alloc(ctx, (size_t)nb * 4));
    XPG_TRY(doff.alloc(ctx, (size_t)(nb + 1) * 8));
    static const bool dbg = xpg_hook("XPG_LINEQ_DEBUG") != 0;
    const auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char * what) { if (dbg) { (void)hipStreamSynchronize(ctx->stream); fprintf(stderr, "  fme_packed %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count()); } };
    const bool small_in = bi <= ((size_t)4 << 20), small_out = bo <= ((size_t)512 << 10);
    const size_t meta_al = (meta + 255) & ~(size_t)255;
    int rc = hpack_reserve(ctx, meta_al + (small_in ? bi : 0) + (small_out ? bo : 0));
    if (rc) return rc;
    char * hp = (char *)ctx->hpack;
    // small inputs through the pinned buffer (one memcpy, then a true asynchronous copy); large ones straight from
    // the caller's pages (the runtime's own staging is faster than a single-threaded memcpy of tens of MB)
    if (small_in) {
        memcpy(hp + meta_al, mats, bi);
        XPG_TRY(hipMemcpyAsync(di.p, hp + meta_al, bi, hipMemcpyHostToDevice, ctx->stream));
    } else {
        XPG_TRY(hipMemcpyAsync(di.p, mats, bi, hipMemcpyHostToDevice, ctx->stream));
    }
    lap("h2d");
    rc = lineq_fme_batch_dev(ctx, nb, (const R32 *)di.p, rows, cols, rhs, u, darkshadow, (R32 *)dout.p, cap, (int32_t *)dr.p, (int32_t *)dk.p);
    if (rc) return rc;
    lap("kernel");
    long long * h_off = (long long *)hp; int32_t * h_ok = (int32_t *)(hp + (size_t)(nb + 1) * 8); int32_t * h_rows = h_ok + nb;
    XPG_TRY(hipMemcpyAsync(h_ok, dk.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(h_rows, dr.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (small_out) {
        // a handful of systems (the drop-in adapter eliminates one per call): the slots themselves come down in the
        // same round and are packed here -- one synchronisation, no scan / pack launches
        char * hs = hp + meta_al + (small_in ? bi : 0);
        XPG_TRY(hipMemcpyAsync(hs, dout.p, bo, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipStreamSynchronize(ctx->stream));
        long long tot = 0;
        for (int b = 0; b < nb; b++) {
            if (h_rows[b] < 0) { ctx->err = "fme: a result needs more rows than cap_rows"; row_offsets[0] = h_rows[b]; return XPG_ERR_UNSUPPORTED; }
            row_offsets[b] = tot; tot += h_rows[b];
        }
        row_offsets[nb] = tot;
        memcpy(out_ok, h_ok, (size_t)nb * 4);
        if (out && out_cap_rows < tot) return XPG_ERR_SHAPE;
        const size_t rowb = (size_t)cols * 8;
        for (int b = 0; b < nb; b++) {                             // forward moves: a destination never lies above its source
            const size_t nbytes = (size_t)(row_offsets[b + 1] - row_offsets[b]) * rowb;
            if (out) memcpy((char *)out + (size_t)row_offsets[b] * rowb, hs + (size_t)b * cap * rowb, nbytes);
            if (view) memmove(hs + (size_t)row_offsets[b] * rowb, hs + (size_t)b * cap * rowb, nbytes);
        }
        if (view) *view = (const R32 *)hs;
        return 0;
    }
    hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, ctx->stream, nb, (const int *)dr.p, (long long *)doff.p);
    XPG_TRY(hipMemcpyAsync(h_off, doff.p, (size_t)(nb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < nb; b++)
        if (h_rows[b] < 0) { ctx->err = "fme: a result needs more rows than cap_rows"; row_offsets[0] = h_rows[b]; return XPG_ERR_UNSUPPORTED; }
    memcpy(row_offsets, h_off, (size_t)(nb + 1) * 8);
    memcpy(out_ok, h_ok, (size_t)nb * 4);
    const long long total = row_offsets[nb];
    if (out && out_cap_rows < total) return XPG_ERR_SHAPE;
    if (total == 0 || (!out && !view)) return 0;                   // (neither: a sizing call)
    const size_t bp = (size_t)total * cols * 8;
    lap("scan+meta");
    XPG_TRY(dpk.alloc(ctx, bp));
    hipLaunchKernelGGL(k_pack_rows, dim3(nb < 4096 ? nb : 4096), dim3(256), 0, ctx->stream, nb, (const R32 *)dout.p, cap, cols,
                       (const int *)dr.p, (const long long *)doff.p, (R32 *)dpk.p);
    XPG_TRY(hipGetLastError());
    lap("pack");
    if (view) {                                                    // into pinned memory; the caller's copy, if any, from there
        rc = hpack_reserve(ctx, bp);
        if (rc) return rc;
        XPG_TRY(hipMemcpyAsync(ctx->hpack, dpk.p, bp, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipStreamSynchronize(ctx->stream));
        if (out) memcpy(out, ctx->hpack, bp);
        *view = (const R32 *)ctx->hpack;
    } else {
        XPG_TRY(hipMemcpyAsync(out, dpk.p, bp, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipStreamSynchronize(ctx->stream));
    }
    lap("d2h");
    return 0;
}

// Lineq::calcBound (linsys.cpp:1047-1078) for nb systems at once: for every variable j the
// other variables are eliminated innermost-first by chained k_fme_batch launches that stay on
// the device (ragged row counts travel in an int array). bounds is [nb][rhs][cap][cols],
// out_rows [nb][rhs]; out_ok[b] = 1, 0 (inconsistent) or -needed_rows (cap too small).
// Packed form (round 5; row_offsets != NULL): the nb * rhs results back to back -- row_offsets[b * rhs + j] is where the bounds
// of variable j of system b start (in rows), row_offsets[nb * rhs] the total -- instead of nb * rhs worst-case slots of `cap` rows:
// the slots stay in HBM, k_rows_scan / k_pack_rows compact them there (as for the packed fme), only live rows cross the link.
// `bounds` (may be NULL) has room for bounds_cap_rows rows; `view` (may be NULL) receives a pointer into the handle's pinned buffer.
inline int lineq_calc_bound_batch(xpg_ctx * ctx, int nb, const R32 * mats, int rows, int cols, int rhs, int cap,
                                  R32 * bounds, int32_t * out_rows, int32_t * out_ok,
                                  long long * row_offsets = nullptr, long long bounds_cap_rows = 0, const R32 ** view = nullptr)
{
    const bool packed = row_offsets != nullptr;
    if (view) *view = 0;
    if (!ctx || nb < 0 || !mats || (!bounds && !packed) || rows <= 0 || cols <= 1 || rhs < 1 || rhs >= cols || cap < rows ||
        (!out_rows && !packed) || !out_ok)
        return XPG_ERR_SHAPE;
    if (nb == 0) { if (packed) row_offsets[0] = 0; return 0; }
    const FmeLds fl = fme_lds(cap, cap, cols);
    if (fl.lds > 160 * 1024 || cap > 32767) return XPG_ERR_UNSUPPORTED;
    const size_t slot = (size_t)cap * cols * 8, bsz = (size_t)nb * slot;
    DevBuf d0, da, db, ra, rb, step_ok, chain, dres, drows;
    XPG_TRY(d0.alloc(ctx, bsz)); XPG_TRY(da.alloc(ctx, bsz)); XPG_TRY(db.alloc(ctx, bsz));
    XPG_TRY(ra.alloc(ctx, (size_t)nb * 4)); XPG_TRY(rb.alloc(ctx, (size_t)nb * 4)); XPG_TRY(step_ok.alloc(ctx, (size_t)nb * 4));
    XPG_TRY(chain.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dres.alloc(ctx, bsz * rhs)); XPG_TRY(drows.alloc(ctx, (size_t)nb * rhs * 4));
    // the input, repacked to the [nb][cap][cols] slot layout
    XPG_TRY(hipMemsetAsync(d0.p, 0, bsz, ctx->stream));
    XPG_TRY(hipMemcpy2DAsync(d0.p, slot, mats, (size_t)rows * cols * 8, (size_t)rows * cols * 8, nb,
                             hipMemcpyHostToDevice, ctx->stream));
    std::vector<int32_t> ones(nb, 1), init_rows(nb, rows);
    XPG_TRY(hipMemcpyAsync(chain.p, ones.data(), (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
    const LineqGeom q = lineq_geom(nb, 64, fl.lds);
    XPG_TRY(lds_limit((const void *)k_fme_batch, ctx->device, q.lds));
    for (int j = 0; j < rhs; j++) {
        void * cur = d0.p; void * cur_rows = ra.p;
        XPG_TRY(hipMemcpyAsync(ra.p, init_rows.data(), (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
        bool flip = false;
        for (int i = rhs - 1; i >= 0; i--) {
            if (i == j) continue;
            void * nxt = flip ? da.p : db.p;
            void * nxt_rows = (cur_rows == ra.p) ? rb.p : ra.p;
            hipLaunchKernelGGL(k_fme_batch, dim3(q.grid), q.block, q.lds, ctx->stream, nb, (const R32 *)cur, cap,
                               (const int *)cur_rows, cols, rhs, i, 0, (R32 *)nxt, cap, (int *)nxt_rows,
                               (int *)step_ok.p, fl.cap_lds, (int *)chain.p, q.sys_lds);
            cur = nxt; cur_rows = nxt_rows; flip = !flip;
        }
        XPG_TRY(hipGetLastError());
        // bounds of variable j: slot j of every system
        XPG_TRY(hipMemcpy2DAsync((char *)dres.p + (size_t)j * slot, slot * rhs, cur, slot, slot, nb,
                                 hipMemcpyDeviceToDevice, ctx->stream));
        XPG_TRY(hipMemcpy2DAsync((char *)drows.p + (size_t)j * 4, (size_t)rhs * 4, cur_rows, 4, 4, nb,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (packed) {
        const int nres = nb * rhs;
        DevBuf doff, dpk;
        XPG_TRY(doff.alloc(ctx, (size_t)(nres + 1) * 8));
        XPG_TRY(hipMemcpyAsync(out_ok, chain.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipStreamSynchronize(ctx->stream));
        bool short_cap = false, some_bad = false;
        for (int b = 0; b < nb; b++) { short_cap |= out_ok[b] < 0; some_bad |= out_ok[b] == 0; }
        if (short_cap) {                                           // a step needs more rows than cap (out_ok[b] = -rows needed): nothing to pack
            for (int k = 0; k <= nres; k++) row_offsets[k] = 0;
            return 0;
        }
        if (some_bad) {
            // an inconsistent system's chains hold whatever its last good step left: they count as empty
            std::vector<int32_t> hrows((size_t)nres);
            XPG_TRY(hipMemcpyAsync(hrows.data(), drows.p, (size_t)nres * 4, hipMemcpyDeviceToHost, ctx->stream));
            XPG_TRY(hipStreamSynchronize(ctx->stream));
            for (int b = 0; b < nb; b++)
                if (out_ok[b] == 0) for (int j = 0; j < rhs; j++) hrows[(size_t)b * rhs + j] = 0;
            XPG_TRY(hipMemcpyAsync(drows.p, hrows.data(), (size_t)nres * 4, hipMemcpyHostToDevice, ctx->stream));
            XPG_TRY(hipStreamSynchronize(ctx->stream));            // (hrows is about to go out of scope)
        }
        hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, ctx->stream, nres, (const int *)drows.p, (long long *)doff.p);
        XPG_TRY(hipMemcpyAsync(row_offsets, doff.p, (size_t)(nres + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        XPG_TRY(hipStreamSynchronize(ctx->stream));
        const long long total = row_offsets[nres];
        if (bounds && bounds_cap_rows < total) return XPG_ERR_SHAPE;
        if (total == 0 || (!bounds && !view)) return 0;
        const size_t bp = (size_t)total * cols * 8;
        XPG_TRY(dpk.alloc(ctx, bp));
        hipLaunchKernelGGL(k_pack_rows, dim3(nres < 4096 ? nres : 4096), dim3(256), 0, ctx->stream, nres, (const R32 *)dres.p, cap, cols,
                           (const int *)drows.p, (const long long *)doff.p, (R32 *)dpk.p);
        XPG_TRY(hipGetLastError());
        if (view) {
            const int rc = hpack_reserve(ctx, bp);
            if (rc) return rc;
            XPG_TRY(hipMemcpyAsync(ctx->hpack, dpk.p, bp, hipMemcpyDeviceToHost, ctx->stream));
            XPG_TRY(hipStreamSynchronize(ctx->stream));
            if (bounds) memcpy(bounds, ctx->hpack, bp);
            *view = (const R32 *)ctx->hpack;
        } else {
            XPG_TRY(hipMemcpyAsync(bounds, dpk.p, bp, hipMemcpyDeviceToHost, ctx->stream));
            XPG_TRY(hipStreamSynchronize(ctx->stream));
        }
        return 0;
    }
    XPG_TRY(hipMemcpyAsync(bounds, dres.p, bsz * rhs, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(out_rows, drows.p, (size_t)nb * rhs * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(out_ok, chain.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// Matrix<Rational>::rank on device arrays, enqueue only: d_rank [nb].
inline int rat_rank_batch_dev(xpg_ctx * ctx, int nb, const R32 * d_mats, int rows, int cols, int32_t * d_rank)
{
    if (!ctx || nb < 0 || !d_mats || !d_rank || rows <= 0 || cols <= 0) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    size_t lds = ((size_t)rows * cols * 8 + 15) & ~(size_t)15;
    lds += ((size_t)rows * 13 + 15) & ~(size_t)15;           // row factors, rowpos, live flags
    if (lds > 160 * 1024) return XPG_ERR_UNSUPPORTED;
    const LineqGeom q = lineq_geom(nb, rows * cols > 256 ? 64 : cols, lds);
    XPG_TRY(lds_limit((const void *)k_gauss_batch, ctx->device, q.lds));
    hipLaunchKernelGGL(k_gauss_batch, dim3(q.grid), q.block, q.lds, ctx->stream, nb, d_mats, rows, cols, 0, 0,
                       (int *)d_rank, (R32 *)0, (R32 *)0, q.sys_lds);
    XPG_TRY(hipGetLastError());
    return 0;
}

// op 0: rank, 1: det, 2: inv, 3: rank with basis (flag = is_unitarize), 4: null space
inline int gauss_batch(xpg_ctx * ctx, int nb, const R32 * mats, int rows, int cols, int op, int32_t * out_int,
                       R32 * out_val, R32 * out_mat, int flag = 0)
{
    if (!ctx || nb < 0 || !mats || rows <= 0 || cols <= 0 || ((op == 1 || op == 2) && rows != cols)) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    size_t lds = ((size_t)rows * cols * 8 * (op == 2 ? 2 : 1) + 15) & ~(size_t)15;
    lds += ((size_t)rows * 13 + 15) & ~(size_t)15;           // row factors, rowpos, live flags
    if (lds > 160 * 1024) return XPG_ERR_UNSUPPORTED;
    const size_t bi = (size_t)nb * rows * cols * 8;
    const size_t bo = op == 2 || op == 3 ? bi : (op == 4 ? (size_t)nb * cols * cols * 8 : 8);
    DevBuf di, dint, dval, dmat;
    XPG_TRY(di.alloc(ctx, bi)); XPG_TRY(dint.alloc(ctx, (size_t)nb * 4)); XPG_TRY(dval.alloc(ctx, (size_t)nb * 8));
    XPG_TRY(dmat.alloc(ctx, bo));
    XPG_TRY(hipMemcpyAsync(di.p, mats, bi, hipMemcpyHostToDevice, ctx->stream));
    if (op >= 2) XPG_TRY(hipMemsetAsync(dmat.p, 0, bo, ctx->stream));
    // small matrices share a wave (one lane per column); from a few hundred cells on the cell-parallel elimination
    // fills the wave by itself and sharing only serialises the groups' branches
    const LineqGeom q = lineq_geom(nb, rows * cols > 256 ? 64 : (op == 2 ? 2 * cols : cols), lds);
    XPG_TRY(lds_limit((const void *)k_gauss_batch, ctx->device, q.lds));
    hipLaunchKernelGGL(k_gauss_batch, dim3(q.grid), q.block, q.lds, ctx->stream, nb, (const R32 *)di.p, rows,
                       cols, op, flag, (int *)dint.p, (R32 *)dval.p, (R32 *)dmat.p, q.sys_lds);
    XPG_TRY(hipGetLastError());
    if (out_int) XPG_TRY(hipMemcpyAsync(out_int, dint.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_val) XPG_TRY(hipMemcpyAsync(out_val, dval.p, (size_t)nb * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_mat && op >= 2) XPG_TRY(hipMemcpyAsync(out_mat, dmat.p, bo, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// INTMat::hnf for nb matrices: h [nb][rows][cols], u [nb][cols][cols], status [nb].
inline int int_hnf_batch(xpg_ctx * ctx, int nb, const int32_t * mats, int rows, int cols, int32_t * h, int32_t * u,
                         int32_t * status)
{
    if (!ctx || nb < 0 || !mats || !h || !u || !status || rows <= 0 || cols <= 0) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const size_t lds = ((size_t)(rows + cols) * cols * 4 + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return XPG_ERR_UNSUPPORTED;
    const size_t bi = (size_t)nb * rows * cols * 4, bu = (size_t)nb * cols * cols * 4;
    DevBuf di, dh, du, ds;
    XPG_TRY(di.alloc(ctx, bi)); XPG_TRY(dh.alloc(ctx, bi)); XPG_TRY(du.alloc(ctx, bu)); XPG_TRY(ds.alloc(ctx, (size_t)nb * 4));
    XPG_TRY(hipMemcpyAsync(di.p, mats, bi, hipMemcpyHostToDevice, ctx->stream));
    XPG_TRY(hipMemsetAsync(dh.p, 0, bi, ctx->stream));
    XPG_TRY(hipMemsetAsync(du.p, 0, bu, ctx->stream));
    XPG_TRY(lds_limit((const void *)k_hnf_batch, ctx->device, lds));
    hipLaunchKernelGGL(k_hnf_batch, dim3(lineq_grid(nb)), dim3(64), lds, ctx->stream, nb, (const int *)di.p, rows, cols,
                       (int *)dh.p, (int *)du.p, (int *)ds.p);
    XPG_TRY(hipGetLastError());
    XPG_TRY(hipMemcpyAsync(h, dh.p, bi, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(u, du.p, bu, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipMemcpyAsync(status, ds.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// INTMat::gcd for nb matrices, in place.
inline int int_gcd_batch(xpg_ctx * ctx, int nb, int32_t * mats, int rows, int cols)
{
    if (!ctx || nb < 0 || !mats || rows <= 0 || cols <= 0) return XPG_ERR_SHAPE;
    if (nb == 0) return 0;
    const size_t bi = (size_t)nb * rows * cols * 4;
    const long long total = (long long)nb * rows;
    DevBuf di;
    XPG_TRY(di.alloc(ctx, bi));
    XPG_TRY(hipMemcpyAsync(di.p, mats, bi, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_int_gcd_batch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, total,
                       (int *)di.p, cols);
    XPG_TRY(hipGetLastError());
    XPG_TRY(hipMemcpyAsync(mats, di.p, bi, hipMemcpyDeviceToHost, ctx->stream));
    XPG_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}



} // namespace xpg
