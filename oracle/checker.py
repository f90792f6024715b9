"""TEST INFRASTRUCTURE -- ctypes bindings for the two CPU checkers.

* ``Ref``    -> oracle/_ref/libxpoly_ref.so : the real xpoly reference behind
                oracle/ref_driver.cpp (built from /root/reference in the
                authoring container; the prebuilt .so travels to the GPU box).
* ``Port``   -> oracle/_build/libxpoly_oracle.so : our CPU restatement
                (oracle/oracle.cpp), same entry points with the prefix ``orc_``.

Only tests/, tools/gen_golden.py, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module. The product (xpoly_amd/) never does.

Data conventions (shared with include/xpoly_amd.h):
  kind 0 = fp64      : numpy float64 arrays, row-major
  kind 1 = rational  : numpy int32 arrays with a trailing axis of 2 = (num, den)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libxpoly_ref.so")
PORT_SO = os.path.join(HERE, "_build", "libxpoly_oracle.so")

F64, RAT = 0, 1
SIX_STATUS = {0: "SUCC", 1: "UNBOUND", 2: "NO_PRI_FEASIBLE_SOL",
              3: "OPTIMAL_IS_INFEASIBLE", 4: "TIME_OUT"}


def build_port():
    subprocess.check_call(["make", "-s", "-C", HERE, "port"])


def build_ref():
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def as_kind(a, kind, ndim):
    """Coerce to the flat layout of `kind`. `ndim` = logical axes; a rational is either integers with
    that many axes or (num, den) pairs with one more, trailing axis of 2 -- never guessed from shape."""
    if kind == F64:
        return np.ascontiguousarray(a, dtype=np.float64)
    a = np.asarray(a)
    if a.ndim == ndim + 1:
        assert a.shape[-1] == 2, a.shape
        return np.ascontiguousarray(a, dtype=np.int32)
    assert a.ndim == ndim, (a.shape, ndim)
    out = np.empty(a.shape + (2,), dtype=np.int32)
    out[..., 0] = a
    out[..., 1] = 1
    return out


def empty_kind(shape, kind):
    if kind == F64:
        return np.zeros(shape, dtype=np.float64)
    return np.zeros(tuple(shape) + (2,), dtype=np.int32)


class _Lib:
    prefix = ""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        self.path = path

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    # ---- SIX::maxm / minm (lpsol.h:1993 / :1662) -------------------------
    def six_solve(self, kind, is_max, tgtf, vc, eq, leq, max_iter=0xFFFFFFFF):
        tgtf = as_kind(tgtf, kind, 1)
        cols = tgtf.shape[-1] if kind == F64 else tgtf.shape[-2]
        vc = as_kind(vc, kind, 2)
        vc_rows = vc.shape[0]
        eq_rows = 0 if eq is None else len(eq)
        leq_rows = 0 if leq is None else len(leq)
        eq_a = as_kind(eq, kind, 2) if eq_rows else None
        leq_a = as_kind(leq, kind, 2) if leq_rows else None
        v = empty_kind((1,), kind)
        sol = empty_kind((cols,), kind)
        fn = self._f("six_solve")
        fn.restype = C.c_int
        st = fn(C.c_int(kind), C.c_int(int(is_max)), _vp(tgtf), _vp(vc),
                C.c_int(vc_rows), _vp(eq_a), C.c_int(eq_rows), _vp(leq_a),
                C.c_int(leq_rows), C.c_int(cols), C.c_uint(max_iter),
                _vp(v), _vp(sol))
        return st, v[0], sol

    # ---- SIX::TwoStageMethod (lpsol.h:1907) ------------------------------
    def two_stage(self, kind, leq, tgtf, max_iter, vc=None):
        leq = as_kind(leq, kind, 2)
        m, cols = leq.shape[0], leq.shape[1]
        tgtf = as_kind(tgtf, kind, 1)
        if vc is None:
            vc = np.zeros((cols - 1, cols), dtype=np.int64)
            vc[np.arange(cols - 1), np.arange(cols - 1)] = -1
            vc = as_kind(vc.astype(np.float64) if kind == F64 else vc.astype(np.int32), kind, 2)
        else:
            vc = as_kind(vc, kind, 2)
        capc = cols + m + 1
        tab = empty_kind((m, capc), kind)
        otg = empty_kind((capc,), kind)
        nv = np.zeros(capc, dtype=np.uint8)
        bv = np.zeros(capc, dtype=np.uint8)
        bv2eq = np.zeros(capc, dtype=np.int32)
        eq2bv = np.zeros(m, dtype=np.int32)
        orows, ocols, orhs = C.c_int(), C.c_int(), C.c_int()
        maxv = empty_kind((1,), kind)
        sol = empty_kind((capc,), kind)
        fn = self._f("two_stage")
        fn.restype = C.c_int
        st = fn(C.c_int(kind), _vp(leq), C.c_int(m), C.c_int(cols), _vp(vc),
                _vp(tgtf), C.c_uint(max_iter), _vp(tab), C.byref(orows),
                C.byref(ocols), _vp(otg), _vp(nv), _vp(bv), _vp(bv2eq),
                _vp(eq2bv), C.byref(orhs), _vp(maxv), _vp(sol))
        r, c, rhs = orows.value, ocols.value, orhs.value
        esz = 1 if kind == F64 else 2
        flat = tab.reshape(-1)[: r * c * esz]
        tab = flat.reshape((r, c) if kind == F64 else (r, c, 2)).copy()
        return dict(status=st, tab=tab, tgtf=otg[:c].copy(), nvset=nv[:rhs].copy(),
                    bvset=bv[:rhs].copy(), bv2eq=bv2eq[:rhs].copy(),
                    eq2bv=eq2bv[:r].copy(), rhs=rhs, maxv=maxv[0], sol=sol[:c].copy())

    # ---- MIP::maxm / minm (lpsol.h:2636 / :2681) -------------------------
    def mip_solve(self, kind, is_max, is_bin, tgtf, vc, eq, leq, rat_ind=None, stats=None):
        """stats: a dict that receives {"nodes", "max_leq_rows"} of the tree walk."""
        tgtf = as_kind(tgtf, kind, 1)
        cols = tgtf.shape[-1] if kind == F64 else tgtf.shape[-2]
        vc = as_kind(vc, kind, 2)
        eq_rows = 0 if eq is None else len(eq)
        leq_rows = 0 if leq is None else len(leq)
        eq_a = as_kind(eq, kind, 2) if eq_rows else None
        leq_a = as_kind(leq, kind, 2) if leq_rows else None
        ind = None if rat_ind is None else np.ascontiguousarray(rat_ind, dtype=np.uint8)
        v = empty_kind((1,), kind)
        sol = empty_kind((cols,), kind)
        args = [C.c_int(kind), C.c_int(int(is_max)), C.c_int(int(is_bin)),
                _vp(tgtf), _vp(vc), C.c_int(vc.shape[0]), _vp(eq_a),
                C.c_int(eq_rows), _vp(leq_a), C.c_int(leq_rows), C.c_int(cols),
                _vp(ind), _vp(v), _vp(sol)]
        if stats is None:
            fn = self._f("mip_solve")
            fn.restype = C.c_int
            st = fn(*args)
        else:
            fn = self._f("mip_solve_stats")
            fn.restype = C.c_int
            nodes, rows = C.c_long(0), (C.c_int * 3)(0, 0, 0)
            st = fn(*args, C.byref(nodes), rows)
            stats["nodes"], stats["max_leq_rows"], stats["max_depth"], stats["spec_chain"] = nodes.value, rows[0], rows[1], rows[2]
        return st, v[0], sol

    # ---- Rational / Float scalars ----------------------------------------
    def rat_op(self, op, a, b):
        rn, rd = C.c_int32(), C.c_int32()
        self._f("rat_op")(C.c_int(op), C.c_int32(a[0]), C.c_int32(a[1]),
                          C.c_int32(b[0]), C.c_int32(b[1]), C.byref(rn), C.byref(rd))
        return rn.value, rd.value

    def rat_cmp(self, cmp, a, b):
        fn = self._f("rat_cmp")
        fn.restype = C.c_int
        return fn(C.c_int(cmp), C.c_int32(a[0]), C.c_int32(a[1]),
                  C.c_int32(b[0]), C.c_int32(b[1]))

    def flt_cmp(self, cmp, x, y):
        fn = self._f("flt_cmp")
        fn.restype = C.c_int
        return fn(C.c_int(cmp), C.c_double(x), C.c_double(y))

    # ---- Lineq (linsys.cpp) ----------------------------------------------
    def fme(self, mat, rhs_idx, u, darkshadow=False, cap_rows=None):
        mat = as_kind(mat, RAT, 2)
        rows, cols = mat.shape[0], mat.shape[1]
        cap = cap_rows or (rows * rows // 4 + rows + 4)
        out = empty_kind((cap, cols), RAT)
        orows, ocols = C.c_int(), C.c_int()
        fn = self._f("fme")
        fn.restype = C.c_int
        ok = fn(_vp(mat), C.c_int(rows), C.c_int(cols), C.c_int(rhs_idx),
                C.c_int(u), C.c_int(int(darkshadow)), _vp(out), C.c_int(cap),
                C.byref(orows), C.byref(ocols))
        if ok < 0:
            raise RuntimeError("fme result does not fit")
        r, c = orows.value, ocols.value
        res = out.reshape(-1)[: r * c * 2].reshape(r, c, 2).copy()
        return ok, res

    def reduce(self, mat, rhs_idx, is_intersect=True):
        mat = as_kind(mat, RAT, 2).copy()
        rows, cols = mat.shape[0], mat.shape[1]
        orows, ocols = C.c_int(), C.c_int()
        fn = self._f("reduce")
        fn.restype = C.c_int
        ok = fn(_vp(mat), C.c_int(rows), C.c_int(cols), C.c_int(rhs_idx),
                C.c_int(int(is_intersect)), C.byref(orows), C.byref(ocols))
        r, c = orows.value, ocols.value
        res = mat.reshape(-1)[: r * c * 2].reshape(r, c, 2).copy()
        return ok, res

    def move2var(self, mat, rhs_idx, first_sym, last_sym):
        """Lineq::move2var (linsys.cpp:1177-1200)."""
        mat = as_kind(mat, RAT, 2).copy()
        self._f("move2var")(_vp(mat), C.c_int(mat.shape[0]), C.c_int(mat.shape[1]), C.c_int(rhs_idx),
                            C.c_int(first_sym), C.c_int(last_sym))
        return mat

    def remove_iden_row(self, mat):
        mat = as_kind(mat, RAT, 2).copy()
        rows, cols = mat.shape[0], mat.shape[1]
        orows = C.c_int()
        self._f("remove_iden_row")(_vp(mat), C.c_int(rows), C.c_int(cols), C.byref(orows))
        r = orows.value
        return mat.reshape(-1)[: r * cols * 2].reshape(r, cols, 2).copy()

    def has_solution(self, leq, eq, vc, rhs_idx, is_int_sol, is_unique_sol):
        vc = as_kind(vc, RAT, 2)
        cols = vc.shape[1]
        leq_rows = 0 if leq is None else len(leq)
        eq_rows = 0 if eq is None else len(eq)
        leq_a = as_kind(leq, RAT, 2) if leq_rows else None
        eq_a = as_kind(eq, RAT, 2) if eq_rows else None
        fn = self._f("has_solution")
        fn.restype = C.c_int
        return fn(_vp(leq_a), C.c_int(leq_rows), _vp(eq_a), C.c_int(eq_rows),
                  _vp(vc), C.c_int(vc.shape[0]), C.c_int(cols), C.c_int(rhs_idx),
                  C.c_int(int(is_int_sol)), C.c_int(int(is_unique_sol)))

    def calc_bound(self, mat, rhs_idx, cap_rows=None):
        """Lineq::calcBound (linsys.cpp:1047-1078): (ok, [bounds of variable j as rows x cols x 2])."""
        mat = as_kind(mat, RAT, 2)
        rows, cols = mat.shape[0], mat.shape[1]
        cap = cap_rows or max(16, 4 * rows * rows)
        out = empty_kind((rhs_idx, cap, cols), RAT)
        orows = np.zeros(rhs_idx, dtype=np.int32)
        fn = self._f("calc_bound")
        fn.restype = C.c_int
        ok = fn(_vp(mat), C.c_int(rows), C.c_int(cols), C.c_int(rhs_idx), _vp(out), C.c_int(cap), _vp(orows))
        if ok < 0:
            raise RuntimeError("calc_bound result does not fit")
        return ok, [out[j, : orows[j]].copy() for j in range(rhs_idx)]

    def rat_rank(self, mat):
        mat = as_kind(mat, RAT, 2)
        fn = self._f("rat_rank")
        fn.restype = C.c_int
        return fn(_vp(mat), C.c_int(mat.shape[0]), C.c_int(mat.shape[1]))

    def rat_det(self, mat):
        mat = as_kind(mat, RAT, 2)
        n, d = C.c_int32(), C.c_int32()
        self._f("rat_det")(_vp(mat), C.c_int(mat.shape[0]), C.byref(n), C.byref(d))
        return n.value, d.value

    def rat_inv(self, mat):
        mat = as_kind(mat, RAT, 2)
        out = empty_kind((mat.shape[0], mat.shape[0]), RAT)
        fn = self._f("rat_inv")
        fn.restype = C.c_int
        ok = fn(_vp(mat), C.c_int(mat.shape[0]), _vp(out))
        return ok, out

    def rat_rank_basis(self, mat, unitarize):
        """Matrix<Rational>::rank(&basis, is_unitarize) (matt.h:2614-2726): (rank, basis)."""
        mat = as_kind(mat, RAT, 2)
        rows, cols = mat.shape[0], mat.shape[1]
        out = empty_kind((rows, cols), RAT)
        orows = C.c_int()
        fn = self._f("rat_rank_basis")
        fn.restype = C.c_int
        rk = fn(_vp(mat), C.c_int(rows), C.c_int(cols), C.c_int(int(unitarize)), _vp(out), C.byref(orows))
        return rk, out[: orows.value].copy()

    def rat_null(self, mat):
        """Matrix<Rational>::null (matt.h:2546-2584): cols x cols, column convention."""
        mat = as_kind(mat, RAT, 2)
        out = empty_kind((mat.shape[1], mat.shape[1]), RAT)
        self._f("rat_null")(_vp(mat), C.c_int(mat.shape[0]), C.c_int(mat.shape[1]), _vp(out))
        return out

    def int_hnf(self, mat):
        """INTMat::hnf (xmat.cpp:912-992): (status, h, u) with h = mat * u."""
        mat = np.ascontiguousarray(mat, dtype=np.int32)
        rows, cols = mat.shape
        h = np.zeros((rows, cols), dtype=np.int32); u = np.zeros((cols, cols), dtype=np.int32)
        fn = self._f("int_hnf")
        fn.restype = C.c_int
        st = fn(_vp(mat), C.c_int(rows), C.c_int(cols), _vp(h), _vp(u))
        return st, h, u

    def int_gcd(self, mat):
        """INTMat::gcd (xmat.cpp:996-1030)."""
        mat = np.ascontiguousarray(mat, dtype=np.int32).copy()
        self._f("int_gcd")(_vp(mat), C.c_int(mat.shape[0]), C.c_int(mat.shape[1]))
        return mat

    def appro_count(self):
        fn = self._f("appro_count")
        fn.restype = C.c_longlong
        return fn()


class Ref(_Lib):
    prefix = "ref_"

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        super().__init__(REF_SO)

    @staticmethod
    def available():
        return os.path.exists(REF_SO)


class Port(_Lib):
    prefix = "orc_"

    def __init__(self):
        if not os.path.exists(PORT_SO):
            build_port()
        super().__init__(PORT_SO)

    def pivot_count(self):
        """SIX::pivot calls of this process so far (restatement only; the reference keeps no such counter)."""
        fn = self.lib.orc_pivot_count
        fn.restype = C.c_longlong
        return fn()
