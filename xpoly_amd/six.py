"""Python face of the C ABI, shaped after the reference's solver classes so the
parity tests read like calls into xpoly:

    SIX<FloatMat,Float>  ->  SIX(ctx, F64)      maxm / minm / set_param / TwoStageMethod
    SIX<RMat,Rational>   ->  SIX(ctx, RAT)      (src/com/lpsol.h:204-338)

Arrays: fp64 problems are float64 numpy arrays; rational problems are int32
arrays with a trailing axis of 2 = (num, den) -- the in-memory layout of RMat.
Everything numeric happens in libxpoly_amd.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import XPG_RUNNING, XpgError, lib, vp

F64, RAT = 0, 1
SIX_SUCC, SIX_UNBOUND, SIX_NO_PRI_FEASIBLE_SOL, SIX_OPTIMAL_IS_INFEASIBLE, SIX_TIME_OUT = range(5)


def as_kind(a, kind, ndim):
    """The array as the ABI wants it. `ndim` is the number of LOGICAL axes (1: a row such as tgtf,
    2: a matrix, 3: a stack of matrices). A rational problem is given either as integers of exactly
    that many axes (each becomes n/1) or as (num, den) pairs with ONE more, trailing axis of length 2
    -- the in-memory layout of RMat. Nothing is inferred from the shape alone: an integer matrix that
    happens to have two columns stays an integer matrix."""
    if a is None:
        return None
    if kind == F64:
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.ndim != ndim:
            raise ValueError("expected %d axes, got shape %s" % (ndim, a.shape))
        return a
    a = np.asarray(a)
    if a.dtype != np.int32 and a.size and np.issubdtype(a.dtype, np.number):
        # Rational is int32 / int32 (rational.h:66-67): a wider integer (or a float) that does not fit must not wrap silently
        lim = np.iinfo(np.int32)
        if (np.asarray(a) > lim.max).any() or (np.asarray(a) < lim.min).any():
            raise ValueError("rational input holds values outside the int32 range of xpoly's Rational")
        if not np.issubdtype(a.dtype, np.integer) and (np.asarray(a) != np.floor(a)).any():
            raise ValueError("rational input must be integers, got fractional %s values" % a.dtype)
    if a.ndim == ndim + 1:
        if a.shape[-1] != 2 or not np.issubdtype(a.dtype, np.integer):
            raise ValueError("rational input must be integer (num, den) pairs [..., 2], got %s %s" % (a.dtype, a.shape))
        return np.ascontiguousarray(a, dtype=np.int32)
    if a.ndim != ndim:
        raise ValueError("expected %d axes (integers) or %d (num/den pairs), got shape %s" % (ndim, ndim + 1, a.shape))
    out = np.empty(a.shape + (2,), dtype=np.int32)
    out[..., 0] = a
    out[..., 1] = 1
    return out


def empty_kind(shape, kind):
    if kind == F64:
        return np.zeros(shape, dtype=np.float64)
    return np.zeros(tuple(shape) + (2,), dtype=np.int32)


class Context:
    """xpg_ctx: one GPU, one HIP stream."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().xpg_create(C.byref(self._h), C.c_int(device))
        if rc != 0:
            raise XpgError("xpg_create(device=%d) failed: %s" % (device, _capi.ERRORS.get(rc, rc)))
        self.device = device

    def check(self, rc, what):
        if rc < 0 and rc != XPG_RUNNING:
            raise XpgError("%s: %s (%s)" % (what, _capi.ERRORS.get(rc, rc),
                                            lib().xpg_last_error(self._h).decode()))
        return rc

    @property
    def stream(self):
        return lib().xpg_stream(self._h)

    def sync(self):
        self.check(lib().xpg_sync(self._h), "xpg_sync")

    def profile_begin(self, cap, stride=1):
        self.check(lib().xpg_profile_begin(self._h, C.c_int(cap), C.c_int(stride)), "xpg_profile_begin")

    def profile_end(self):
        n, ms = C.c_int(), C.c_double()
        self.check(lib().xpg_profile_end(self._h, C.byref(n), C.byref(ms)), "xpg_profile_end")
        return n.value, ms.value

    def malloc(self, nbytes):
        p = C.c_void_p()
        self.check(lib().xpg_malloc(self._h, C.byref(p), C.c_size_t(nbytes)), "xpg_malloc")
        return p.value

    def free(self, ptr):
        self.check(lib().xpg_free(self._h, C.c_void_p(ptr)), "xpg_free")

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(lib().xpg_upload(self._h, C.c_void_p(dptr), vp(arr), C.c_size_t(arr.nbytes)), "xpg_upload")

    def download(self, arr, dptr):
        self.check(lib().xpg_download(self._h, vp(arr), C.c_void_p(dptr), C.c_size_t(arr.nbytes)), "xpg_download")
        return arr

    def close(self):
        if self._h:
            for r in list(getattr(self, "_lps", ())):       # device LPs of this context first: they hold its stream and buffers
                lp = r()
                if lp is not None:
                    lp.close()
            lib().xpg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- K1 ---------------------------------------------------------------------------
    def pivot(self, kind, tab, obj, rhs_idx, row, col):
        """SIX::pivot arithmetic (lpsol.h:1471-1501) on host arrays, in place."""
        tab = as_kind(tab, kind, 2); obj = as_kind(obj, kind, 1)
        m, W = tab.shape[0], tab.shape[1]
        fn = lib().xpg_pivot_f64 if kind == F64 else lib().xpg_pivot_rat32
        self.check(fn(self._h, vp(tab), C.c_int(m), C.c_int(W), vp(obj), C.c_int(rhs_idx),
                      C.c_int(row), C.c_int(col)), "xpg_pivot")
        return tab, obj

    def pivot_dev(self, kind, tab_ptr, m, W, ld, obj_ptr, rhs_idx, row, col):
        fn = lib().xpg_pivot_f64_dev if kind == F64 else lib().xpg_pivot_rat32_dev
        self.check(fn(self._h, C.c_void_p(tab_ptr), C.c_int(m), C.c_int(W), C.c_int(ld),
                      C.c_void_p(obj_ptr), C.c_int(rhs_idx), C.c_int(row), C.c_int(col)), "xpg_pivot_dev")

    # ---- batches --------------------------------------------------------------------------
    def six_batch(self, kind, is_max, tgtf, leq, max_iter=0xFFFFFFFF):
        tgtf = as_kind(tgtf, kind, 2); leq = as_kind(leq, kind, 3)
        nb, m, cols = leq.shape[0], leq.shape[1], leq.shape[2]
        status = np.zeros(nb, dtype=np.int32)
        v = empty_kind((nb,), kind)
        sol = empty_kind((nb, cols), kind)
        fn = lib().xpg_six_batch_f64 if kind == F64 else lib().xpg_six_batch_rat32
        self.check(fn(self._h, C.c_int(int(is_max)), C.c_int(nb), vp(tgtf), vp(leq), C.c_int(m),
                      C.c_int(cols), C.c_uint(max_iter), vp(status), vp(v), vp(sol)), "xpg_six_batch")
        return status, v, sol

    def six_batch_dev(self, kind, is_max, nb, tgtf_ptr, leq_ptr, m, cols, status_ptr, v_ptr, sol_ptr,
                      pivots_ptr=None, max_iter=0xFFFFFFFF):
        fn = lib().xpg_six_batch_f64_dev if kind == F64 else lib().xpg_six_batch_rat32_dev
        self.check(fn(self._h, C.c_int(int(is_max)), C.c_int(nb), C.c_void_p(tgtf_ptr), C.c_void_p(leq_ptr),
                      C.c_int(m), C.c_int(cols), C.c_uint(max_iter), C.c_void_p(status_ptr),
                      C.c_void_p(v_ptr), C.c_void_p(sol_ptr),
                      C.c_void_p(pivots_ptr) if pivots_ptr else None), "xpg_six_batch_dev")


def _check_multi(rc, what):
    if rc < 0:
        raise XpgError("%s: %s" % (what, _capi.ERRORS.get(rc, rc)))


def six_batch_multi(devices, kind, is_max, tgtf, leq, max_iter=0xFFFFFFFF):
    """xpg_six_batch_*_multi: the batch cut into contiguous shards, one per entry of `devices`, each on a
    context and host thread of its own inside the library; results land in these host arrays."""
    tgtf = as_kind(tgtf, kind, 2); leq = as_kind(leq, kind, 3)
    nb, m, cols = leq.shape[0], leq.shape[1], leq.shape[2]
    status = np.zeros(nb, dtype=np.int32)
    v = empty_kind((nb,), kind); sol = empty_kind((nb, cols), kind)
    dev = (C.c_int * len(devices))(*devices)
    fn = lib().xpg_six_batch_f64_multi if kind == F64 else lib().xpg_six_batch_rat32_multi
    _check_multi(fn(C.c_int(len(devices)), dev, C.c_int(int(is_max)), C.c_int(nb), vp(tgtf), vp(leq), C.c_int(m),
                    C.c_int(cols), C.c_uint(max_iter), vp(status), vp(v), vp(sol)), "xpg_six_batch_multi")
    return status, v, sol


def mip_batch_multi(devices, is_max, is_bin, tgtf, leq):
    tgtf = as_kind(tgtf, RAT, 2); leq = as_kind(leq, RAT, 3)
    nb, rows, cols = leq.shape[0], leq.shape[1], leq.shape[2]
    st = np.zeros(nb, dtype=np.int32); v = empty_kind((nb,), RAT); sol = empty_kind((nb, cols), RAT)
    nodes = C.c_longlong()
    dev = (C.c_int * len(devices))(*devices)
    _check_multi(lib().xpg_mip_batch_rat32_multi(C.c_int(len(devices)), dev, C.c_int(nb), C.c_int(int(is_max)),
                                                 C.c_int(int(is_bin)), vp(tgtf), vp(leq), C.c_int(rows), C.c_int(cols),
                                                 vp(st), vp(v), vp(sol), C.byref(nodes)), "xpg_mip_batch_rat32_multi")
    return st, v, sol, nodes.value


def dep_is_empty_batch_multi(devices, mats):
    mats = as_kind(mats, RAT, 3)
    nb, rows, cols = mats.shape[0], mats.shape[1], mats.shape[2]
    out = np.zeros(nb, dtype=np.int32)
    nodes = C.c_longlong()
    dev = (C.c_int * len(devices))(*devices)
    _check_multi(lib().xpg_dep_is_empty_batch_rat32_multi(C.c_int(len(devices)), dev, C.c_int(nb), vp(mats), C.c_int(rows),
                                                          C.c_int(cols), vp(out), C.byref(nodes)),
                 "xpg_dep_is_empty_batch_rat32_multi")
    return out, nodes.value


class DeviceLP:
    """xpg_lp: a slack-form LP living in HBM (tableau, objective row, basis)."""

    def __init__(self, ctx, kind, leq, tgtf, vc_diag=None, vc_rhs=None, on_device=False, m=None, cols=None):
        self.ctx, self.kind = ctx, kind
        if on_device:
            leq_p, tg_p = C.c_void_p(leq), C.c_void_p(tgtf)
        else:
            leq = as_kind(leq, kind, 2); tgtf = as_kind(tgtf, kind, 1)
            m, cols = leq.shape[0], leq.shape[1]
            leq_p, tg_p = vp(leq), vp(tgtf)
        self.m, self.cols = m, cols
        vd = as_kind(vc_diag, kind, 1); vr = as_kind(vc_rhs, kind, 1)
        self._h = C.c_void_p()
        ctx.check(lib().xpg_lp_create(ctx._h, C.c_int(kind), leq_p, C.c_int(m), C.c_int(cols), tg_p,
                                      vp(vd), vp(vr), C.c_int(int(on_device)), C.byref(self._h)),
                  "xpg_lp_create")
        import weakref
        if not hasattr(ctx, "_lps"):
            ctx._lps = []
        ctx._lps.append(weakref.ref(self))

    def set_options(self, pricing=0, feas_rel_tol=0.0):
        """Opt-in NON-PARITY modes (xpg_lp_set_options): pricing=1 Dantzig's rule, feas_rel_tol>0 a
        tolerant SIX::is_feasible. (0, 0.0) is the reference's behaviour."""
        return self.ctx.check(lib().xpg_lp_set_options(self._h, C.c_int(pricing), C.c_double(feas_rel_tol)),
                              "xpg_lp_set_options")

    def two_stage(self, max_iter=0xFFFFFFFF):
        return self.ctx.check(lib().xpg_lp_two_stage(self._h, C.c_uint(max_iter)), "xpg_lp_two_stage")

    def begin(self):
        return self.ctx.check(lib().xpg_lp_begin(self._h), "xpg_lp_begin")

    def iterate(self, pivots):
        return self.ctx.check(lib().xpg_lp_iterate(self._h, C.c_uint(pivots)), "xpg_lp_iterate")

    def pivots_done(self):
        n = C.c_uint()
        self.ctx.check(lib().xpg_lp_pivots_done(self._h, C.byref(n)), "xpg_lp_pivots_done")
        return n.value

    def counters(self):
        """(sweeps of a full 16-pivot batch, sweeps of a shorter one) of the blocked loop since begin()."""
        f, p = C.c_uint(), C.c_uint()
        self.ctx.check(lib().xpg_lp_counters(self._h, C.byref(f), C.byref(p)), "xpg_lp_counters")
        return f.value, p.value

    def chain_aborts(self):
        """(chain launches given up at their roll call in this solve, whether the solve switched to launch-per-stage)."""
        n, off, runs = C.c_uint(), C.c_int(), C.c_uint()
        self.ctx.check(lib().xpg_lp_chain_aborts(self._h, C.byref(n), C.byref(off), C.byref(runs)), "xpg_lp_chain_aborts")
        self.chain_runs = runs.value
        return n.value, bool(off.value)

    def chain_folds(self):
        """chain launches of this solve that did their batch's stage 0 themselves."""
        n = C.c_uint()
        self.ctx.check(lib().xpg_lp_chain_folds(self._h, C.byref(n)), "xpg_lp_chain_folds")
        return n.value

    def loop_info(self):
        """xpg_lp_loop_info: which loop and kernel instances run the LP in its current shape."""
        o = (C.c_int32 * 10)()
        self.ctx.check(lib().xpg_lp_loop_info(self._h, o, C.c_int(10)), "xpg_lp_loop_info")
        return dict(loop={0: "pipelined", 1: "serial", 3: "blocked"}.get(o[0], o[0]), pivots_per_pass=o[1],
                    chain={0: "launch per stage", 1: "one launch, one XCD", 2: "one launch, spread"}[o[2]] if o[0] == 3 else None,
                    chain_line=o[3], sweep_rows=o[4], ld=o[5], chain_workers=o[6], pick_workers=o[7], prep_workers=o[8],
                    chain_lds_bytes=o[9])

    def shape(self):
        r, w, rhs = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(lib().xpg_lp_shape(self._h, C.byref(r), C.byref(w), C.byref(rhs)), "xpg_lp_shape")
        return r.value, w.value, rhs.value

    def read(self, want_tab=True):
        r, W, rhs = self.shape()
        k = self.kind
        tab = empty_kind((r, W), k) if want_tab else None
        obj = empty_kind((W,), k)
        nv = np.zeros(rhs, dtype=np.uint8); bv = np.zeros(rhs, dtype=np.uint8)
        bv2eq = np.zeros(rhs, dtype=np.int32); eq2bv = np.zeros(r, dtype=np.int32)
        maxv = empty_kind((1,), k); sol = empty_kind((W,), k)
        self.ctx.check(lib().xpg_lp_read(self._h, vp(tab), vp(obj), vp(nv), vp(bv), vp(bv2eq), vp(eq2bv),
                                         vp(maxv), vp(sol)), "xpg_lp_read")
        return dict(tab=tab, tgtf=obj, nvset=nv, bvset=bv, bv2eq=bv2eq, eq2bv=eq2bv, rhs=rhs,
                    maxv=maxv[0], sol=sol)

    def trace(self, cap=65536):
        pairs = np.zeros((cap, 2), dtype=np.int32)
        n = C.c_int()
        self.ctx.check(lib().xpg_lp_trace(self._h, vp(pairs), C.c_int(cap), C.byref(n)), "xpg_lp_trace")
        return pairs[: min(n.value, cap)].copy()

    def close(self):
        if self._h:
            lib().xpg_lp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MIP:
    """Mirror of MIP<Mat,T> (src/com/lpsol.h:2087-2157): maxm / minm with is_bin and
    rational_indicator; IP_* status codes."""

    def __init__(self, ctx, kind):
        self.ctx, self.kind = ctx, kind

    def _solve(self, is_max, tgtf, vc, eq, leq, is_bin, rational_indicator):
        k = self.kind
        tgtf = as_kind(tgtf, k, 1); vc = as_kind(vc, k, 2)
        cols = tgtf.shape[0]
        eq_rows = 0 if eq is None else len(eq)
        leq_rows = 0 if leq is None else len(leq)
        eq_a = as_kind(eq, k, 2) if eq_rows else None
        leq_a = as_kind(leq, k, 2) if leq_rows else None
        ind = None if rational_indicator is None else np.ascontiguousarray(rational_indicator, dtype=np.uint8)
        v = empty_kind((1,), k); sol = empty_kind((cols,), k)
        name = "xpg_mip_%s_%s" % ("maxm" if is_max else "minm", "f64" if k == F64 else "rat32")
        st = getattr(lib(), name)(self.ctx._h, vp(tgtf), vp(vc), C.c_int(vc.shape[0]), vp(eq_a), C.c_int(eq_rows),
                                  vp(leq_a), C.c_int(leq_rows), C.c_int(cols), C.c_int(int(is_bin)), vp(ind),
                                  vp(v), vp(sol))
        if st < 0 and st != -7:
            self.ctx.check(st, name)
        return st, v[0], sol

    def maxm(self, tgtf, vc, eq, leq, is_bin=False, rational_indicator=None):     # lpsol.h:2636-2657
        return self._solve(True, tgtf, vc, eq, leq, is_bin, rational_indicator)

    def minm(self, tgtf, vc, eq, leq, is_bin=False, rational_indicator=None):     # lpsol.h:2681-2702
        return self._solve(False, tgtf, vc, eq, leq, is_bin, rational_indicator)


def six_last_profile():
    """Where the calling thread's last SIX.maxm / minm call spent its time (xpg_six_last_profile), milliseconds."""
    out = (C.c_double * 9)()
    lib().xpg_six_last_profile(out, C.c_int(9))
    names = ("total_ms", "host_reshape_ms", "create_and_upload_ms", "dual_on_device_ms", "device_solve_ms", "read_back_ms", "release_ms")
    d = {k: round(out[i], 3) for i, k in enumerate(names)}
    d["route"] = {1: "LDS batch kernel", 2: "HBM-resident loop"}.get(int(out[7]), "none")
    d["pivots"] = int(out[8])                        # HBM-resident route: pivots of the loop that ended the solve
    return d


def mip_warm(ctx, is_max, tgtf, leq, is_bin=False):
    """OPT-IN, NON-PARITY (xpg_mip_warm_f64): fp64 branch and bound warm-started from the parent's tableau by the dual
    simplex. Returns (status, v, sol, dict(nodes, dual_pivots, root_pivots, max_depth))."""
    tgtf = as_kind(tgtf, F64, 1); leq = as_kind(leq, F64, 2)
    rows, cols = leq.shape
    v = C.c_double(); sol = np.zeros(cols); stats = (C.c_longlong * 4)()
    st = lib().xpg_mip_warm_f64(ctx._h, C.c_int(int(is_max)), vp(tgtf), vp(leq), C.c_int(rows), C.c_int(cols),
                                C.c_int(int(is_bin)), C.byref(v), vp(sol), stats)
    ctx.check(st, "xpg_mip_warm_f64")
    return st, v.value, sol, dict(nodes=stats[0], dual_pivots=stats[1], root_pivots=stats[2], max_depth=stats[3])


def mip_warm_batch(ctx, is_max, tgtf, leq, is_bin=False):
    """OPT-IN, NON-PARITY (xpg_mip_warm_batch_f64): nb fp64 integer programs of one shape, each tree walked by one workgroup
    with the dual simplex warm-started from the parent's tableau. tgtf [nb, cols], leq [nb, rows, cols].
    Returns (status[nb], v[nb], sol[nb, cols], dict(nodes, dual_pivots, root_pivots, max_depth))."""
    tgtf = as_kind(tgtf, F64, 2); leq = as_kind(leq, F64, 3)
    nb, rows, cols = leq.shape
    st = np.zeros(nb, dtype=np.int32); v = np.zeros(nb); sol = np.zeros((nb, cols)); stats = (C.c_longlong * 4)()
    ctx.check(lib().xpg_mip_warm_batch_f64(ctx._h, C.c_int(nb), C.c_int(int(is_max)), vp(tgtf), vp(leq), C.c_int(rows), C.c_int(cols),
                                           C.c_int(int(is_bin)), vp(st), vp(v), vp(sol), stats), "xpg_mip_warm_batch_f64")
    return st, v, sol, dict(nodes=stats[0], dual_pivots=stats[1], root_pivots=stats[2], max_depth=stats[3])


def mip_batch(ctx, is_max, is_bin, tgtf, leq, kind=RAT):
    """nb independent MIPs (x >= 0, inequalities only), each tree walked on the device by one workgroup.
    tgtf [nb, cols(,2)], leq [nb, rows, cols(,2)]. Returns (status[nb], v[nb(,2)], sol[nb,cols(,2)], nodes)."""
    tgtf = as_kind(tgtf, kind, 2); leq = as_kind(leq, kind, 3)
    nb, rows, cols = leq.shape[0], leq.shape[1], leq.shape[2]
    st = np.zeros(nb, dtype=np.int32); v = empty_kind((nb,), kind); sol = empty_kind((nb, cols), kind)
    nodes = C.c_longlong()
    fn = lib().xpg_mip_batch_rat32 if kind == RAT else lib().xpg_mip_batch_f64
    ctx.check(fn(ctx._h, C.c_int(nb), C.c_int(int(is_max)), C.c_int(int(is_bin)), vp(tgtf),
                 vp(leq), C.c_int(rows), C.c_int(cols), vp(st), vp(v), vp(sol), C.byref(nodes)), "xpg_mip_batch")
    return st, v, sol, nodes.value


def mip_batch_eq(ctx, is_max, is_bin, tgtf, leq, eq, kind=RAT):
    """nb independent MIPs with equalities at the root (x >= 0): tgtf [nb, cols(,2)], leq [nb, rows, cols(,2)] or None,
    eq [nb, eq_rows, cols(,2)]. Returns (status[nb], v[nb(,2)], sol[nb,cols(,2)], nodes)."""
    tgtf = as_kind(tgtf, kind, 2); eq = as_kind(eq, kind, 3)
    leq = None if leq is None else as_kind(leq, kind, 3)
    nb, eq_rows, cols = eq.shape[0], eq.shape[1], eq.shape[2]
    rows = 0 if leq is None else leq.shape[1]
    st = np.zeros(nb, dtype=np.int32); v = empty_kind((nb,), kind); sol = empty_kind((nb, cols), kind)
    nodes = C.c_longlong()
    fn = lib().xpg_mip_batch_eq_rat32 if kind == RAT else lib().xpg_mip_batch_eq_f64
    ctx.check(fn(ctx._h, C.c_int(nb), C.c_int(int(is_max)), C.c_int(int(is_bin)), vp(tgtf), vp(leq), C.c_int(rows), vp(eq),
                 C.c_int(eq_rows), C.c_int(cols), vp(st), vp(v), vp(sol), C.byref(nodes)), "xpg_mip_batch_eq")
    return st, v, sol, nodes.value


def dep_is_empty_batch(ctx, mats, rhs_idx=None, vc=None):
    """DepPoly::is_empty(keepit, vc) (src/eng/poly.cpp:530-573) for a stack of dependence polyhedra
    [nb, rows, cols(,2)]: the constant is column rhs_idx (default: the last), the columns behind it are constant
    symbols (moved to the variables first, Lineq::move2var); vc [rhs_idx, rhs_idx + 1] are the caller's variable
    constraints (default -x_i <= 0). Returns (empty[nb], nodes); -7 where the reference is undefined."""
    mats = as_kind(mats, RAT, 3)
    nb, rows, cols = mats.shape[0], mats.shape[1], mats.shape[2]
    rhs_idx = cols - 1 if rhs_idx is None else rhs_idx
    vc_a = None if vc is None else as_kind(vc, RAT, 2)
    out = np.zeros(nb, dtype=np.int32)
    nodes = C.c_longlong()
    ctx.check(lib().xpg_dep_is_empty_batch_ex_rat32(ctx._h, C.c_int(nb), vp(mats), C.c_int(rows), C.c_int(cols),
                                                    C.c_int(rhs_idx), vp(vc_a), vp(out), C.byref(nodes)),
              "xpg_dep_is_empty_batch_ex_rat32")
    return out, nodes.value


def dep_is_empty_batch_symbols_as_vars(ctx, mats, rhs_idx, vc=None):
    """OPT-IN, NON-PARITY (xpg_dep_is_empty_batch_mode_rat32, XPG_DEP_SYMBOLS_AS_VARS): parametrised dependence polyhedra the
    reference leaves undefined -- the constant symbols behind column rhs_idx become free variables of the widened system.
    Returns (empty[nb], nodes)."""
    mats = as_kind(mats, RAT, 3)
    nb, rows, cols = mats.shape[0], mats.shape[1], mats.shape[2]
    vc_a = None if vc is None else as_kind(vc, RAT, 2)
    out = np.zeros(nb, dtype=np.int32)
    nodes = C.c_longlong()
    ctx.check(lib().xpg_dep_is_empty_batch_mode_rat32(ctx._h, C.c_int(nb), vp(mats), C.c_int(rows), C.c_int(cols), C.c_int(rhs_idx),
                                                      vp(vc_a), C.c_int(1), vp(out), C.byref(nodes)),
              "xpg_dep_is_empty_batch_mode_rat32")
    return out, nodes.value


def has_solution(ctx, leq, eq, vc, rhs_idx, is_int_sol, is_unique_sol):
    """Lineq::has_solution (src/com/linsys.cpp:830-906) on rational systems."""
    vc = as_kind(vc, RAT, 2)
    cols = vc.shape[1]
    leq_rows = 0 if leq is None else len(leq)
    eq_rows = 0 if eq is None else len(eq)
    leq_a = as_kind(leq, RAT, 2) if leq_rows else None
    eq_a = as_kind(eq, RAT, 2) if eq_rows else None
    r = lib().xpg_has_solution_rat32(ctx._h, vp(leq_a), C.c_int(leq_rows), vp(eq_a), C.c_int(eq_rows), vp(vc),
                                     C.c_int(vc.shape[0]), C.c_int(cols), C.c_int(rhs_idx),
                                     C.c_int(int(is_int_sol)), C.c_int(int(is_unique_sol)))
    if r < 0 and r != -7:
        ctx.check(r, "xpg_has_solution_rat32")
    return r


class SIX:
    """Mirror of SIX<Mat,T> (src/com/lpsol.h:204-338): same method names,
    argument order and status codes; `kind` picks the FloatMat or RMat flavour."""

    def __init__(self, ctx, kind):
        self.ctx, self.kind = ctx, kind
        self.max_iter = 0xFFFFFFFF

    def set_param(self, indent, max_iter=0xFFFFFFFF):       # lpsol.h:380-385
        self.max_iter = max_iter

    def _solve(self, is_max, tgtf, vc, eq, leq):
        k = self.kind
        tgtf = as_kind(tgtf, k, 1); vc = as_kind(vc, k, 2)
        cols = tgtf.shape[0]
        eq_rows = 0 if eq is None else len(eq)
        leq_rows = 0 if leq is None else len(leq)
        eq_a = as_kind(eq, k, 2) if eq_rows else None
        leq_a = as_kind(leq, k, 2) if leq_rows else None
        v = empty_kind((1,), k); sol = empty_kind((cols,), k)
        name = "xpg_six_%s_%s" % ("maxm" if is_max else "minm", "f64" if k == F64 else "rat32")
        st = getattr(lib(), name)(self.ctx._h, vp(tgtf), vp(vc), C.c_int(vc.shape[0]), vp(eq_a),
                                  C.c_int(eq_rows), vp(leq_a), C.c_int(leq_rows), C.c_int(cols),
                                  C.c_uint(self.max_iter), vp(v), vp(sol))
        if st < 0 and st != -7:
            self.ctx.check(st, name)
        return st, v[0], sol

    def maxm(self, tgtf, vc, eq, leq):                      # lpsol.h:1993-2033
        return self._solve(True, tgtf, vc, eq, leq)

    def minm(self, tgtf, vc, eq, leq):                      # lpsol.h:1662-1732
        return self._solve(False, tgtf, vc, eq, leq)

    def TwoStageMethod(self, leq, tgtf, vc_diag=None, vc_rhs=None):   # lpsol.h:1907-1930
        lp = DeviceLP(self.ctx, self.kind, leq, tgtf, vc_diag, vc_rhs)
        st = lp.two_stage(self.max_iter)
        out = lp.read()
        out["status"] = st
        out["trace"] = lp.trace()
        lp.close()
        return out


# ---- ragged batches: problems of different shapes in one call (include/xpoly_amd.h, "ragged batches") ----------
def _ragged_pack(arrays, kind, ndim):
    """A list of per-problem arrays -> (flat concatenation, cell offsets[nb + 1]); cells are 8 bytes of either kind."""
    parts = [np.ascontiguousarray(as_kind(a, kind, ndim)) for a in arrays]
    cells = [int(np.prod(p.shape[:ndim])) for p in parts]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    off[1:] = np.cumsum(cells)
    flat = np.concatenate([p.reshape(-1) for p in parts]) if parts else np.zeros(0, dtype=np.float64 if kind == F64 else np.int32)
    return np.ascontiguousarray(flat), off, parts


def six_batch_ragged(ctx, kind, is_max, tgtfs, leqs, max_iter=0xFFFFFFFF):
    """SIX::maxm / minm (x >= 0, inequalities only) on LPs of DIFFERENT shapes in one call: leqs[b] is rows_b x cols_b,
    tgtfs[b] has cols_b entries. Returns (status[nb], [v_b], [sol_b])."""
    nb = len(leqs)
    leq_flat, leq_off, lp = _ragged_pack(leqs, kind, 2)
    tg_flat, tg_off, _ = _ragged_pack(tgtfs, kind, 1)
    rows = np.array([p.shape[0] for p in lp], dtype=np.int32); cols = np.array([p.shape[1] for p in lp], dtype=np.int32)
    assert np.array_equal(np.diff(tg_off), cols), "tgtfs[b] must have cols_b entries"
    status = np.zeros(nb, dtype=np.int32)
    v = empty_kind((nb,), kind)
    sol = np.zeros_like(tg_flat)
    fn = lib().xpg_six_batch_f64_ragged if kind == F64 else lib().xpg_six_batch_rat32_ragged
    ctx.check(fn(ctx._h, C.c_int(int(is_max)), C.c_int(nb), vp(tg_flat), vp(leq_flat), vp(rows), vp(cols), vp(leq_off),
                 vp(tg_off), C.c_uint(max_iter), vp(status), vp(v), vp(sol)), "xpg_six_batch_ragged")
    esz = 1 if kind == F64 else 2
    sols = [sol[int(tg_off[b]) * esz: int(tg_off[b + 1]) * esz].reshape((-1,) if kind == F64 else (-1, 2)) for b in range(nb)]
    return status, v, sols


def ragged_pack_rat(mats_list):
    """(flat cells, rows[nb], cols[nb], cell offsets[nb + 1]) of a list of rational systems: the arrays the ragged C entry points take."""
    flat, off, parts = _ragged_pack(mats_list, RAT, 2)
    rows = np.array([p.shape[0] for p in parts], dtype=np.int32); cols = np.array([p.shape[1] for p in parts], dtype=np.int32)
    return flat, rows, cols, off


def dep_is_empty_ragged(ctx, mats_list=None, packed=None):
    """DepPoly::is_empty (src/eng/poly.cpp:530-573) on dependence polyhedra of mixed shapes (constant in the last
    column of each): (empty[nb], nodes). `packed` = ragged_pack_rat(...) of the list, for callers that keep it."""
    flat, rows, cols, off = packed if packed is not None else ragged_pack_rat(mats_list)
    nb = len(rows)
    out = np.zeros(nb, dtype=np.int32)
    nodes = C.c_longlong()
    ctx.check(lib().xpg_dep_is_empty_batch_ragged_rat32(ctx._h, C.c_int(nb), vp(flat), vp(rows), vp(cols), vp(off), vp(out),
                                                        C.byref(nodes)), "xpg_dep_is_empty_batch_ragged_rat32")
    return out, nodes.value
