"""Python face of the row-elimination entry points, shaped after xpoly's Lineq
(src/com/linsys.h:61-186) and Matrix<Rational> (src/com/matt.h). Batched: every
call takes a stack of equally shaped systems [nb, rows, cols, 2] (int32 num/den)
and runs one wavefront per system on the GPU.
"""
import ctypes as C

import numpy as np

from ._capi import lib, vp
from .six import RAT, as_kind


def _stack(mats):
    """One system [rows, cols, 2] or a stack [nb, rows, cols, 2] of (num, den) pairs. Integer systems
    are converted explicitly by the caller (xpoly_amd.six.as_kind(a, RAT, ndim)), never guessed."""
    a = np.asarray(mats)
    if a.ndim not in (3, 4) or a.shape[-1] != 2 or not np.issubdtype(a.dtype, np.integer):
        raise ValueError("Lineq takes (num, den) pairs [rows, cols, 2] or [nb, rows, cols, 2], got %s %s" % (a.dtype, a.shape))
    a = as_kind(a, RAT, a.ndim - 1)
    if a.ndim == 3:
        a = a[None]
    return np.ascontiguousarray(a)


class Lineq:
    def __init__(self, ctx):
        self.ctx = ctx

    def reduce(self, mats, rhs_idx, is_intersect=True):
        """Lineq::reduce (linsys.cpp:359-626). Returns (ok[nb], [system b's surviving rows]) -- through the packed entry
        point; the rows are slices of one array."""
        ok, off, packed = self.reduce_packed(mats, rhs_idx, is_intersect)
        return ok, [packed[off[b]: off[b + 1]] for b in range(len(ok))]

    def reduce_packed(self, mats, rhs_idx, is_intersect=True, copy=True):
        """xpg_lineq_reduce_batch_packed_rat32: (ok[nb], row_offsets[nb + 1], rows[row_offsets[nb], cols, 2]); the input is
        not modified. copy=False returns the rows as a view of the handle's pinned buffer (valid until the handle's next
        packed call): what a C++ caller reads its results from."""
        a = _stack(mats)
        nb, rows, cols = a.shape[:3]
        off = np.zeros(nb + 1, dtype=np.int64); ok = np.zeros(nb, dtype=np.int32)
        view = C.c_void_p()
        self.ctx.check(lib().xpg_lineq_reduce_batch_packed_rat32(
            self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols), C.c_int(rhs_idx), C.c_int(int(is_intersect)),
            None, C.c_longlong(0), C.byref(view), vp(off), None, vp(ok)), "xpg_lineq_reduce_batch_packed_rat32")
        total = int(off[nb])
        if not copy and total:
            buf = (C.c_int32 * (total * cols * 2)).from_address(view.value)
            return ok, off, np.frombuffer(buf, dtype=np.int32).reshape(total, cols, 2)
        packed = np.empty((total, cols, 2), dtype=np.int32)
        if total:
            C.memmove(packed.ctypes.data, view.value, packed.nbytes)
        return ok, off, packed

    def reduce_inplace(self, mats, rhs_idx, is_intersect=True):
        """xpg_lineq_reduce_batch_rat32, the in-place form of the reference's signature: `mats` [nb, rows, cols, 2] (int32,
        contiguous) is overwritten -- system b's surviving rows at the front of its slot. Returns (ok[nb], out_rows[nb])."""
        a = mats
        assert isinstance(a, np.ndarray) and a.dtype == np.int32 and a.ndim == 4 and a.flags.c_contiguous
        nb, rows, cols = a.shape[:3]
        out_rows = np.zeros(nb, dtype=np.int32); ok = np.zeros(nb, dtype=np.int32)
        self.ctx.check(lib().xpg_lineq_reduce_batch_rat32(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols),
                                                          C.c_int(rhs_idx), C.c_int(int(is_intersect)), vp(out_rows), vp(ok)),
                       "xpg_lineq_reduce_batch_rat32")
        return ok, out_rows

    def move2var(self, mats, rhs_idx, first_sym, last_sym):
        """Lineq::move2var (linsys.cpp:1177-1200): constant symbols become variables in front of the constant."""
        a = _stack(mats).copy()
        nb, rows, cols = a.shape[:3]
        self.ctx.check(lib().xpg_lineq_move2var_batch_rat32(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols),
                                                            C.c_int(rhs_idx), C.c_int(first_sym), C.c_int(last_sym)),
                       "xpg_lineq_move2var_batch_rat32")
        return a

    def removeIdenRow(self, mats):
        """Lineq::removeIdenRow (linsys.cpp:1209-1268)."""
        a = _stack(mats).copy()
        nb, rows, cols = a.shape[:3]
        out_rows = np.zeros(nb, dtype=np.int32)
        self.ctx.check(lib().xpg_lineq_remove_iden_batch_rat32(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows),
                                                               C.c_int(cols), vp(out_rows)),
                       "xpg_lineq_remove_iden_batch_rat32")
        return [a[b, : out_rows[b]].copy() for b in range(nb)]

    def fme(self, mats, rhs_idx, u, darkshadow=False, cap_rows=None, slots=False):
        """Lineq::fme (linsys.cpp:656-774). Returns (ok[nb], [result of system b]). Through the packed entry point
        (row offsets + live rows only); slots=True takes the cap_rows-slot entry point instead (A/B, tests)."""
        a = _stack(mats)
        nb, rows, cols = a.shape[:3]
        if slots:
            cap = cap_rows or max(rows, rows * rows // 4 + rows + 1)
            outs = np.zeros((nb, cap, cols, 2), dtype=np.int32)
            out_rows = np.zeros(nb, dtype=np.int32); ok = np.zeros(nb, dtype=np.int32)
            self.ctx.check(lib().xpg_lineq_fme_batch_rat32(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols),
                                                           C.c_int(rhs_idx), C.c_int(u), C.c_int(int(darkshadow)), vp(outs),
                                                           C.c_int(cap), vp(out_rows), vp(ok)),
                           "xpg_lineq_fme_batch_rat32")
            if (out_rows < 0).any():
                raise ValueError("fme result needs %d rows, cap_rows is %d" % (-out_rows.min(), cap))
            return ok, [outs[b, : out_rows[b]].copy() for b in range(nb)]
        ok, off, packed = self.fme_packed(a, rhs_idx, u, darkshadow, cap_rows)
        return ok, [packed[off[b]: off[b + 1]] for b in range(nb)]

    def fme_packed(self, mats, rhs_idx, u, darkshadow=False, cap_rows=None, copy=True):
        """xpg_lineq_fme_batch_packed_rat32: (ok[nb], row_offsets[nb + 1], rows[row_offsets[nb], cols, 2]).
        copy=False returns the rows as a view of the handle's pinned buffer, valid until the next packed call
        on this handle (what a C++ caller reads its results from)."""
        a = _stack(mats)
        nb, rows, cols = a.shape[:3]
        off = np.zeros(nb + 1, dtype=np.int64); ok = np.zeros(nb, dtype=np.int32)
        view = C.c_void_p()
        self.ctx.check(lib().xpg_lineq_fme_batch_packed_rat32(
            self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols), C.c_int(rhs_idx), C.c_int(u),
            C.c_int(int(darkshadow)), C.c_int(cap_rows or 0), None, C.c_longlong(0), C.byref(view), vp(off), vp(ok)),
            "xpg_lineq_fme_batch_packed_rat32")
        total = int(off[nb])
        if not copy and total:
            buf = (C.c_int32 * (total * cols * 2)).from_address(view.value)
            return ok, off, np.frombuffer(buf, dtype=np.int32).reshape(total, cols, 2)
        packed = np.empty((total, cols, 2), dtype=np.int32)
        if total:                                           # out of the handle's pinned buffer, before the next call reuses it
            C.memmove(packed.ctypes.data, view.value, packed.nbytes)
        return ok, off, packed

    def reduce_ragged(self, mats_list, rhs_idx=None, is_intersect=True):
        """Lineq::reduce on systems of different shapes in one call: (ok[nb], [system b's surviving rows])."""
        from .six import _ragged_pack
        flat, off, parts = _ragged_pack(mats_list, RAT, 2)
        nb = len(parts)
        rows = np.array([p.shape[0] for p in parts], dtype=np.int32); cols = np.array([p.shape[1] for p in parts], dtype=np.int32)
        rhs = None if rhs_idx is None else np.ascontiguousarray(rhs_idx, dtype=np.int32)
        out_rows = np.zeros(nb, dtype=np.int32); ok = np.zeros(nb, dtype=np.int32)
        self.ctx.check(lib().xpg_lineq_reduce_batch_ragged_rat32(self.ctx._h, C.c_int(nb), vp(flat), vp(rows), vp(cols), vp(off),
                                                                 vp(rhs), C.c_int(int(is_intersect)), vp(out_rows), vp(ok)),
                       "xpg_lineq_reduce_batch_ragged_rat32")
        return ok, [flat[2 * int(off[b]): 2 * int(off[b]) + 2 * int(out_rows[b]) * int(cols[b])].reshape(int(out_rows[b]), int(cols[b]), 2).copy()
                    for b in range(nb)]

    RAGGED_FME_ONE_CALL_BYTES = 256 << 20

    def fme_ragged(self, mats_list, u, rhs_idx=None, darkshadow=False):
        """Lineq::fme on systems of different shapes, eliminating variable u[b] of system b: (ok[nb], [result b])."""
        from .six import _ragged_pack
        flat, off, parts = _ragged_pack(mats_list, RAT, 2)
        nb = len(parts)
        rows = np.array([p.shape[0] for p in parts], dtype=np.int32); cols = np.array([p.shape[1] for p in parts], dtype=np.int32)
        rhs = None if rhs_idx is None else np.ascontiguousarray(rhs_idx, dtype=np.int32)
        uu = np.ascontiguousarray(u, dtype=np.int32)
        out_rows = np.zeros(nb, dtype=np.int32); ok = np.zeros(nb, dtype=np.int32); ooff = np.zeros(nb + 1, dtype=np.int64)
        # ONE call where the worst case is small: the entry point computes every elimination before it looks at the output
        # buffer, so a sizing call does the whole batch twice. An elimination of R rows leaves at most R*R/4 + R rows (P
        # positive x N negative combinations with P + N <= R, plus the rows without the variable): the buffer is sized for
        # that and trimmed. Above 256 MB of worst case (large batches: several GB of zero pages that raise MemoryError where
        # the real result fits easily) the sizing call is the cheaper evil: the first call returns XPG_ERR_SHAPE with the
        # offsets filled (include/xpoly_amd.h), the second gets a buffer of exactly that size.
        capc = int(sum((int(r) * int(r) // 4 + int(r)) * int(c) for r, c in zip(rows, cols)))

        def call(buf, cap_cells):
            return lib().xpg_lineq_fme_batch_ragged_rat32(
                self.ctx._h, C.c_int(nb), vp(flat), vp(rows), vp(cols), vp(off), vp(rhs), vp(uu), C.c_int(int(darkshadow)), vp(buf),
                C.c_longlong(cap_cells), vp(ooff), vp(out_rows), vp(ok))

        if capc * 8 <= self.RAGGED_FME_ONE_CALL_BYTES:
            outs = np.zeros((max(capc, 1), 2), dtype=np.int32)
            self.ctx.check(call(outs, capc), "xpg_lineq_fme_batch_ragged_rat32")
        else:
            rc = call(None, 0)
            # XPG_ERR_SHAPE from a sizing call means "too small, offsets filled" -- or malformed input (cols > 255, u out of
            # range, a bad rhs_idx), refused before any offset is written: told apart by whether anything was filled in
            if rc != -3 or not (int(ooff[nb]) > 0 or out_rows.any()):
                self.ctx.check(rc, "xpg_lineq_fme_batch_ragged_rat32 (sizing)")
            need = int(ooff[nb])
            outs = np.zeros((max(need, 1), 2), dtype=np.int32)
            if need:
                self.ctx.check(call(outs, need), "xpg_lineq_fme_batch_ragged_rat32")
        return ok, [outs[int(ooff[b]): int(ooff[b + 1])].reshape(int(out_rows[b]), int(cols[b]), 2).copy() for b in range(nb)]

    def calcBound(self, mats, rhs_idx, cap_rows=None):
        """Lineq::calcBound (linsys.cpp:1047-1078): chained eliminations on the device.
        Returns (ok[nb], bounds[b][j] = rows x cols x 2 array bounding variable j alone)."""
        a = _stack(mats)
        nb, rows, cols = a.shape[:3]
        cap = cap_rows or max(16, rows * rows)
        out = np.zeros((nb, rhs_idx, cap, cols, 2), dtype=np.int32)
        out_rows = np.zeros((nb, rhs_idx), dtype=np.int32); ok = np.zeros(nb, dtype=np.int32)
        self.ctx.check(lib().xpg_lineq_calc_bound_batch_rat32(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols),
                                                              C.c_int(rhs_idx), C.c_int(cap), vp(out), vp(out_rows), vp(ok)),
                       "xpg_lineq_calc_bound_batch_rat32")
        return ok, [[out[b, j, : max(out_rows[b, j], 0)].copy() for j in range(rhs_idx)] for b in range(nb)]

    def calcBound_packed(self, mats, rhs_idx, cap_rows=None):
        """The same through the packed entry point (xpg_lineq_calc_bound_batch_packed_rat32): only live rows cross the link.
        Returns (ok[nb], bounds[b][j]) like calcBound; ok[b] < 0 (a step needed -ok[b] rows) leaves every bound empty."""
        a = _stack(mats)
        nb, rows, cols = a.shape[:3]
        off = np.zeros(nb * rhs_idx + 1, dtype=np.int64); ok = np.zeros(nb, dtype=np.int32)
        view = C.c_void_p()
        self.ctx.check(lib().xpg_lineq_calc_bound_batch_packed_rat32(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols),
                                                                     C.c_int(rhs_idx), C.c_int(cap_rows or 0), None, C.c_longlong(0),
                                                                     C.byref(view), vp(off), vp(ok)),
                       "xpg_lineq_calc_bound_batch_packed_rat32")
        total = int(off[-1])
        packed = np.zeros((total, cols, 2), dtype=np.int32)
        if total:                                           # out of the handle's pinned buffer, before the next call reuses it
            C.memmove(packed.ctypes.data, view.value, packed.nbytes)
        return ok, [[packed[int(off[b * rhs_idx + j]): int(off[b * rhs_idx + j + 1])].copy() for j in range(rhs_idx)] for b in range(nb)]

    def rank(self, mats):
        a = _stack(mats)
        nb, rows, cols = a.shape[:3]
        out = np.zeros(nb, dtype=np.int32)
        self.ctx.check(lib().xpg_rat_rank_batch(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols), vp(out)),
                       "xpg_rat_rank_batch")
        return out

    def rankBasis(self, mats, is_unitarize):
        """Matrix<Rational>::rank(&basis, is_unitarize) (matt.h:2614-2726): (rank[nb], [basis b])."""
        a = _stack(mats)
        nb, rows, cols = a.shape[:3]
        rk = np.zeros(nb, dtype=np.int32); brows = np.zeros(nb, dtype=np.int32)
        out = np.zeros((nb, rows, cols, 2), dtype=np.int32)
        self.ctx.check(lib().xpg_rat_rank_basis_batch(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols),
                                                      C.c_int(int(is_unitarize)), vp(rk), vp(out), vp(brows)),
                       "xpg_rat_rank_basis_batch")
        return rk, [out[b, : brows[b]].copy() for b in range(nb)]

    def null(self, mats):
        """Matrix<Rational>::null (matt.h:2546-2584): [nb, cols, cols, 2], column convention."""
        a = _stack(mats)
        nb, rows, cols = a.shape[:3]
        out = np.zeros((nb, cols, cols, 2), dtype=np.int32)
        self.ctx.check(lib().xpg_rat_null_batch(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols), vp(out)),
                       "xpg_rat_null_batch")
        return out

    def hnf(self, imats):
        """INTMat::hnf (xmat.cpp:912-992) of int32 matrices [nb, rows, cols]: (status[nb], h, u)."""
        a = np.ascontiguousarray(imats, dtype=np.int32)
        if a.ndim == 2:
            a = a[None]
        nb, rows, cols = a.shape
        h = np.zeros((nb, rows, cols), dtype=np.int32); u = np.zeros((nb, cols, cols), dtype=np.int32)
        st = np.zeros(nb, dtype=np.int32)
        self.ctx.check(lib().xpg_int_hnf_batch(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols), vp(h), vp(u),
                                               vp(st)), "xpg_int_hnf_batch")
        return st, h, u

    def gcd(self, imats):
        """INTMat::gcd (xmat.cpp:996-1030): rows divided by the gcd of their nonzero magnitudes."""
        a = np.ascontiguousarray(imats, dtype=np.int32).copy()
        if a.ndim == 2:
            a = a[None]
        nb, rows, cols = a.shape
        self.ctx.check(lib().xpg_int_gcd_batch(self.ctx._h, C.c_int(nb), vp(a), C.c_int(rows), C.c_int(cols)),
                       "xpg_int_gcd_batch")
        return a

    def det(self, mats):
        a = _stack(mats)
        nb, n = a.shape[0], a.shape[1]
        out = np.zeros((nb, 2), dtype=np.int32)
        self.ctx.check(lib().xpg_rat_det_batch(self.ctx._h, C.c_int(nb), vp(a), C.c_int(n), vp(out)), "xpg_rat_det_batch")
        return out

    def inv(self, mats):
        a = _stack(mats)
        nb, n = a.shape[0], a.shape[1]
        out = np.zeros((nb, n, n, 2), dtype=np.int32); ok = np.zeros(nb, dtype=np.int32)
        self.ctx.check(lib().xpg_rat_inv_batch(self.ctx._h, C.c_int(nb), vp(a), C.c_int(n), vp(out), vp(ok)),
                       "xpg_rat_inv_batch")
        return ok, out


def reduce_dev(ctx, nb, mats_ptr, rows, cols, rhs_idx, is_intersect, out_rows_ptr, out_ok_ptr):
    """xpg_lineq_reduce_batch_rat32_dev: Lineq::reduce in place on device arrays, enqueue only (ctx.sync() after)."""
    ctx.check(lib().xpg_lineq_reduce_batch_rat32_dev(ctx._h, C.c_int(nb), C.c_void_p(mats_ptr), C.c_int(rows), C.c_int(cols),
                                                     C.c_int(rhs_idx), C.c_int(int(is_intersect)), C.c_void_p(out_rows_ptr),
                                                     C.c_void_p(out_ok_ptr)), "xpg_lineq_reduce_batch_rat32_dev")


def fme_dev(ctx, nb, mats_ptr, rows, cols, rhs_idx, u, darkshadow, outs_ptr, cap_rows, out_rows_ptr, out_ok_ptr):
    """xpg_lineq_fme_batch_rat32_dev: Lineq::fme on device arrays, enqueue only."""
    ctx.check(lib().xpg_lineq_fme_batch_rat32_dev(ctx._h, C.c_int(nb), C.c_void_p(mats_ptr), C.c_int(rows), C.c_int(cols),
                                                  C.c_int(rhs_idx), C.c_int(u), C.c_int(int(darkshadow)), C.c_void_p(outs_ptr),
                                                  C.c_int(cap_rows), C.c_void_p(out_rows_ptr), C.c_void_p(out_ok_ptr)),
              "xpg_lineq_fme_batch_rat32_dev")


def rank_dev(ctx, nb, mats_ptr, rows, cols, out_rank_ptr):
    """xpg_rat_rank_batch_dev: Matrix<Rational>::rank on device arrays, enqueue only."""
    ctx.check(lib().xpg_rat_rank_batch_dev(ctx._h, C.c_int(nb), C.c_void_p(mats_ptr), C.c_int(rows), C.c_int(cols),
                                           C.c_void_p(out_rank_ptr)), "xpg_rat_rank_batch_dev")
