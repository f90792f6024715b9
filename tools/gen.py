"""Synthetic problem generators shared by tests/, bench.py and the golden-vector
script. Pure numpy; no reference or oracle code is touched here.

Two sources of randomness:
  * xorshift64: s ^= s<<13; s ^= s>>7; s ^= s<<17,
    seed 88172645463325252, u = (s >> 11) * 2**-53 -- the benchmark workloads.
  * numpy Generator -- fuzz families for the differential tests.
"""
import numpy as np

XS_SEED = 88172645463325252
_M64 = (1 << 64) - 1


class XorShift64:
    def __init__(self, seed=XS_SEED):
        self.s = seed & _M64

    def next_u64(self):
        s = self.s
        s ^= (s << 13) & _M64
        s ^= s >> 7
        s ^= (s << 17) & _M64
        self.s = s
        return s

    def uniform(self):
        return (self.next_u64() >> 11) * 2.0 ** -53

    def uniforms(self, n):
        """n uniforms, vectorised in blocks (bit-identical to repeated uniform())."""
        out = np.empty(n, dtype=np.float64)
        s = self.s
        for i in range(n):
            s ^= (s << 13) & _M64
            s ^= s >> 7
            s ^= (s << 17) & _M64
            out[i] = (s >> 11) * 2.0 ** -53
        self.s = s
        return out


def xs_uniform_block(n, seed=XS_SEED):
    """n xorshift64 uniforms from `seed`, computed with numpy uint64 lanes.

    The stream is split into 4096 independent sub-streams seeded by the scalar
    generator, so large tableaux (4096 x 8192) fill in well under a second; the
    layout is deterministic and documented, not the single sequential stream.
    """
    lanes = 4096
    g = XorShift64(seed)
    s = np.array([g.next_u64() for _ in range(lanes)], dtype=np.uint64)
    per = (n + lanes - 1) // lanes
    out = np.empty((per, lanes), dtype=np.float64)
    for i in range(per):
        s ^= s << np.uint64(13)
        s ^= s >> np.uint64(7)
        s ^= s << np.uint64(17)
        out[i] = (s >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return out.reshape(-1)[:n]


def vc_nonneg(nv, kind_float=True, free=()):
    vc = np.zeros((nv, nv + 1), dtype=np.float64 if kind_float else np.int32)
    for i in range(nv):
        if i not in free:
            vc[i, i] = -1
    return vc


# ---- benchmark workloads -----------------------------------
def dense_lp_f64(m, n, seed=XS_SEED):
    """cfg 2b / cfg 3(i): maximise c.x, A x <= b, x >= 0; A,c ~ U(0.1,1), b = n*U(0.5,1)."""
    u = xs_uniform_block(m * n + m + n, seed)
    A = 0.1 + 0.9 * u[: m * n].reshape(m, n)
    b = n * (0.5 + 0.5 * u[m * n: m * n + m])
    c = 0.1 + 0.9 * u[m * n + m:]
    leq = np.concatenate([A, b[:, None]], axis=1)
    tgtf = np.concatenate([c, [0.0]])
    return leq, tgtf


def tableau_f64(m, W, seed=XS_SEED):
    """cfg 2a: T[m][W] ~ U(0.1,1) and an objective row of W entries."""
    u = xs_uniform_block(m * W + W, seed)
    return (0.1 + 0.9 * u[: m * W]).reshape(m, W).copy(), (0.1 + 0.9 * u[m * W:]).copy()


def small_lp_batch_f64(nb, m=32, cols=64, family=0, seed=XS_SEED):
    """cfg 3: nb independent LPs, leq m x cols (cols-1 variables + constant).

    family 0: dense positive (as cfg 2b scaled); family 1: dependence-test-like,
    entries in {-3..3} at density 0.25, rhs in {-2..17}, objective all ones.
    Returns leq [nb,m,cols], tgtf [nb,cols] as float64.
    """
    n = cols - 1
    per = m * n + m + n
    u = xs_uniform_block(nb * per, seed).reshape(nb, per)
    if family == 0:
        A = 0.1 + 0.9 * u[:, : m * n].reshape(nb, m, n)
        b = n * (0.5 + 0.5 * u[:, m * n: m * n + m])
        c = 0.1 + 0.9 * u[:, m * n + m:]
    else:
        ua = u[:, : m * n].reshape(nb, m, n)
        A = np.where(ua < 0.25, np.floor(ua * 28.0) - 3.0, 0.0)   # ua<0.25 -> ua*28 in [0,7) -> {-3..3}
        b = np.floor(u[:, m * n: m * n + m] * 20.0) - 2.0
        c = np.ones((nb, n))
    leq = np.concatenate([A, b[:, :, None]], axis=2)
    tgtf = np.concatenate([c, np.zeros((nb, 1))], axis=1)
    return np.ascontiguousarray(leq), np.ascontiguousarray(tgtf)


def int_lp_rat(m, n, seed=XS_SEED):
    """cfg 4: integer data A in {1..9}, b = n*{3,4,5}, c in {1..9} as (num,den) int32."""
    u = xs_uniform_block(m * n + m + n, seed)
    A = np.floor(u[: m * n] * 9).astype(np.int32).reshape(m, n) + 1
    b = (n * (3 + np.floor(u[m * n: m * n + m] * 3))).astype(np.int32)
    c = np.floor(u[m * n + m:] * 9).astype(np.int32) + 1
    leq = np.concatenate([A, b[:, None]], axis=1)
    tgtf = np.concatenate([c, [0]]).astype(np.int32)
    return to_rat(leq), to_rat(tgtf)


def to_rat(a):
    a = np.asarray(a, dtype=np.int32)
    out = np.empty(a.shape + (2,), dtype=np.int32)
    out[..., 0] = a
    out[..., 1] = 1
    return out


# ---- fuzz families -----------------------------------------------------------------
def random_problem(rng, kind, fam, m, nv, plain=False):
    """A random (tgtf, vc, eq, leq) in the reference's calling convention.

    fam 0: dense positive (origin feasible); 1: small signed integers (phase 1,
    unbounded, infeasible all occur); 2: sparse signed integers with repeated
    ratios (ties, degeneracy); 3: like 1 plus equalities and free variables.
    plain=True keeps x >= 0 and no equalities (input to TwoStageMethod).
    """
    cols = nv + 1
    if fam == 0:
        A = rng.integers(1, 10, size=(m, nv)).astype(np.float64)
        b = rng.integers(nv, 5 * nv + 1, size=m).astype(np.float64)
        c = rng.integers(1, 10, size=nv).astype(np.float64)
        if kind == 0 and rng.integers(0, 2):
            A = A / 7.0 + rng.random((m, nv)) * 0.25
            c = c / 3.0
    elif fam == 1:
        A = rng.integers(-4, 5, size=(m, nv)).astype(np.float64)
        b = rng.integers(-6, 12, size=m).astype(np.float64)
        c = rng.integers(-3, 4, size=nv).astype(np.float64)
    else:
        A = rng.integers(-2, 3, size=(m, nv)).astype(np.float64)
        A *= rng.random((m, nv)) < 0.6
        b = rng.integers(-2, 5, size=m).astype(np.float64)
        c = rng.integers(-1, 3, size=nv).astype(np.float64)
    leq = np.concatenate([A, b[:, None]], axis=1)
    tgtf = np.concatenate([c, [float(rng.integers(0, 3)) if fam else 0.0]])
    free = ()
    eq = None
    if fam == 3 and not plain:
        free = tuple(i for i in range(nv) if rng.random() < 0.25)
        if rng.random() < 0.6:
            ne = int(rng.integers(1, 3))
            Ae = rng.integers(-2, 3, size=(ne, nv)).astype(np.float64)
            be = rng.integers(-3, 6, size=ne).astype(np.float64)
            eq = np.concatenate([Ae, be[:, None]], axis=1)
    vc = vc_nonneg(nv, True, free)
    prob = dict(tgtf=tgtf, vc=vc, leq=leq)
    if eq is not None:
        prob["eq"] = eq
    if kind == 1:
        for k in list(prob):
            prob[k] = to_rat(prob[k].astype(np.int32))
    assert prob["tgtf"].shape[0] == cols
    return prob


def random_mip(rng, m, nv, is_bin):
    A = rng.integers(0, 7, size=(m, nv))
    if rng.random() < 0.3:
        A = A - rng.integers(0, 3, size=(m, nv))
    b = rng.integers(2, 4 * nv + 3, size=m)
    c = rng.integers(1, 9, size=nv)
    leq = np.concatenate([A, b[:, None]], axis=1).astype(np.int32)
    if is_bin and rng.random() < 0.7:
        ub = np.zeros((nv, nv + 1), dtype=np.int32)
        ub[np.arange(nv), np.arange(nv)] = 1
        ub[:, nv] = 1
        leq = np.concatenate([leq, ub], axis=0)
    tgtf = np.concatenate([c, [0]]).astype(np.int32)
    prob = dict(tgtf=to_rat(tgtf), vc=to_rat(vc_nonneg(nv, False)), leq=to_rat(leq))
    if rng.random() < 0.2:
        prob["ind"] = (rng.random(nv + 1) < 0.3).astype(np.uint8)
    return prob


def random_system(rng, rows, nv):
    A = rng.integers(-3, 4, size=(rows, nv))
    A *= rng.random((rows, nv)) < 0.7
    b = rng.integers(-5, 9, size=rows)
    mat = np.concatenate([A, b[:, None]], axis=1).astype(np.int32)
    if rows > 2 and rng.random() < 0.5:
        mat[rng.integers(0, rows)] = mat[rng.integers(0, rows)]
    out = to_rat(mat)
    if rng.random() < 0.3:
        i, j = rng.integers(0, rows), rng.integers(0, nv + 1)
        out[i, j] = (int(rng.integers(-5, 6)), int(rng.integers(2, 5)))
    return out


def random_square(rng, n):
    return to_rat(rng.integers(-4, 5, size=(n, n)).astype(np.int32))


def random_feas(rng, rows, nv):
    A = rng.integers(-3, 4, size=(rows, nv))
    b = rng.integers(-4, 10, size=rows)
    leq = to_rat(np.concatenate([A, b[:, None]], axis=1).astype(np.int32))
    vc = to_rat(vc_nonneg(nv, False))
    return leq, vc


def hard_lp_f64(m, n, seed=XS_SEED):
    """Dense LP whose optimum has every structural variable basic, so the simplex
    needs >= n pivots from the origin (the cfg-2b recipe of SURVEY 8d converges in a
    few dozen pivots, too few to time): A ~ U(0.1,1), b = A x*, c = A^T y* with
    x*, y* ~ U(0.5,1.5) > 0 (x* primal feasible and tight, y* dual feasible =>
    x* optimal by complementary slackness). Origin feasible: no phase 1."""
    u = xs_uniform_block(m * n + m + n, seed)
    A = 0.1 + 0.9 * u[: m * n].reshape(m, n)
    xs = 0.5 + u[m * n + m:]
    ys = 0.5 + u[m * n: m * n + m]
    b = A @ xs
    c = A.T @ ys
    leq = np.concatenate([A, b[:, None]], axis=1)
    tgtf = np.concatenate([c, [0.0]])
    return np.ascontiguousarray(leq), np.ascontiguousarray(tgtf)


def knapsack_batch_rat(nb, nv, seed=XS_SEED):
    """cfg 5: nb 0-1 multi-constraint knapsacks, integer data as (num, den) int32:
    maximise c.x, 2 capacity rows A x <= floor(sum(A_row) / 2), x_j <= 1 rows, x >= 0;
    A, c in {1..9} from the xorshift64 stream. leq [nb, 2 + nv, nv + 1, 2], tgtf [nb, nv + 1, 2]."""
    m = 2
    per = m * nv + nv
    u = xs_uniform_block(nb * per, seed).reshape(nb, per)
    A = np.floor(u[:, : m * nv] * 9).astype(np.int32).reshape(nb, m, nv) + 1
    c = np.floor(u[:, m * nv:] * 9).astype(np.int32) + 1
    b = A.sum(axis=2) // 2
    cap = np.concatenate([A, b[:, :, None]], axis=2)
    ub = np.zeros((nv, nv + 1), dtype=np.int32)
    ub[np.arange(nv), np.arange(nv)] = 1
    ub[:, nv] = 1
    leq = np.concatenate([cap, np.broadcast_to(ub, (nb, nv, nv + 1))], axis=1)
    tgtf = np.concatenate([c, np.zeros((nb, 1), dtype=np.int32)], axis=1)
    return to_rat(np.ascontiguousarray(leq)), to_rat(tgtf)


def interval_mip_f64(rng, m=25, nv=35, maxlen=6):
    """An integer program whose fp64 arithmetic is exact and whose tree goes deep: consecutive-ones rows (totally
    unimodular), half-integer right-hand sides, so every basic solution is a multiple of 1/2 and the walk branches
    on one variable after the other (up to nv bound rows on top of the m inequalities). Returns (leq, tgtf)."""
    A = np.zeros((m, nv), dtype=np.int64)
    starts = np.linspace(0, nv, 13).astype(int)     # the first 12 rows tile the columns: every variable is bounded
    r = 0
    for k in range(12):
        A[r, starts[k]:starts[k + 1]] = 1
        r += 1
    while r < m:
        a = int(rng.integers(0, nv - 1))
        L = int(rng.integers(2, maxlen + 1))
        A[r, a:min(nv, a + L)] = 1
        r += 1
    b = rng.integers(1, 6, size=m) + 0.5
    c = rng.integers(1, 9, size=nv)
    leq = np.concatenate([A, b[:, None]], axis=1).astype(np.float64)
    tgtf = np.concatenate([c, [0]]).astype(np.float64)
    return leq, tgtf


def overflow_lp_f64(trial, key=2026):
    """A dense LP of FINITE cells whose magnitudes are spread element by element over hundreds of decades: products overflow
    after a few pivots, inf - inf follows, and NaN cells appear MID-SOLVE (row / column scalings alone would not do it: the
    simplex is invariant under them). Returns (leq, tgtf); tests keep the trials whose oracle run does meet a NaN."""
    rng = np.random.default_rng([key, trial])
    m = int(rng.integers(20, 70))
    n = int(rng.integers(20, 90))
    span = int(rng.choice([200, 250, 300]))
    A = rng.uniform(0.1, 1.0, size=(m, n)) * 10.0 ** rng.integers(-span, span + 1, size=(m, n))
    b = rng.uniform(0.5, 1.0, size=m) * 10.0 ** rng.integers(0, span + 1, size=m)
    c = rng.uniform(0.1, 1.0, size=n) * 10.0 ** rng.integers(-span, span + 1, size=n)
    leq = np.ascontiguousarray(np.concatenate([A, b[:, None]], axis=1))
    tgtf = np.ascontiguousarray(np.concatenate([c, [0.0]]))
    return leq, tgtf


def interval_lp_f64(m, n, seed=1, maxlen=24):
    """A dense-stored LP whose fp64 arithmetic is EXACT, so that the reference's final feasibility check (row sums compared
    with a 1e-17 window) can succeed at a size where rounded data never does: consecutive-ones rows (a totally
    unimodular matrix: every basis inverse is integral), integer right-hand sides, rows and columns rescaled by powers of
    two (pivots are not all on +-1, every product and quotient is still exact). maximise c.x, A x <= b, x >= 0; the first
    rows tile the columns so that every variable is bounded. Returns (leq [m, n + 1], tgtf [n + 1])."""
    rng = np.random.default_rng([2718, seed])
    A = np.zeros((m, n))
    tiles = max(1, min(m // 2, n // 8))
    starts = np.linspace(0, n, tiles + 1).astype(int)
    r = 0
    for k in range(tiles):
        A[r, starts[k]:starts[k + 1]] = 1
        r += 1
    while r < m:
        a = int(rng.integers(0, n - 1))
        A[r, a:min(n, a + int(rng.integers(2, maxlen + 1)))] = 1
        r += 1
    b = rng.integers(1, 40, size=m).astype(np.float64)
    c = rng.integers(1, 9, size=n).astype(np.float64)
    rs = 2.0 ** rng.integers(-2, 3, size=m)
    cs = 2.0 ** rng.integers(-2, 3, size=n)
    A = A * rs[:, None] * cs[None, :]
    b = b * rs
    c = c * cs
    leq = np.ascontiguousarray(np.concatenate([A, b[:, None]], axis=1))
    tgtf = np.ascontiguousarray(np.concatenate([c, [0.0]]))
    return leq, tgtf


def lp_block_f64(seed, wide=False):
    """One small block of block_lp_f64: 3..10 rows, 3..12 variables (wide: 8..27), origin feasible; even seeds small
    integers (exact arithmetic), odd seeds U(0.1, 1) data (rounded arithmetic). Returns (A [m, n], b [m], c [n])."""
    rng = np.random.default_rng([31337, seed])
    m = int(rng.integers(3, 11)); n = int(rng.integers(3, 13))
    if wide:
        n = int(rng.integers(8, 28))
    if seed % 2 == 0:
        A = rng.integers(0, 5, size=(m, n)).astype(np.float64)
        A[:, A.sum(axis=0) == 0] = 1.0                       # every variable bounded
        b = rng.integers(4, 40, size=m).astype(np.float64)
        c = rng.integers(1, 7, size=n).astype(np.float64)
    else:
        A = 0.1 + 0.9 * rng.random((m, n))
        b = n * (0.5 + 0.5 * rng.random(m))
        c = 0.1 + 0.9 * rng.random(n)
    return A, b, c


def block_lp_f64(block_seeds, wide=False):
    """A block-diagonal LP stored dense: maximise c.x, A x <= b, x >= 0 with A = diag(A_1 .. A_k), the blocks lp_block_f64(s)
    for s in block_seeds. The reference's pricing takes the first column with a positive cost, so it solves the blocks one after
    the other and the whole ends SIX_SUCC exactly when every block does (cells of other blocks are zeros: they add nothing to
    the row sums of the final feasibility check) -- the way to a LARGE fp64 LP that ends with status 0 and a non-zero optimum
    (dense random LPs of that size never do). The seeds come from a search with the oracle and are part of the
    fixture (tools/gen_golden_end.py). Returns (leq [m, n + 1], tgtf [n + 1])."""
    return _assemble_blocks([lp_block_f64(int(s), wide) for s in block_seeds])


def _assemble_blocks(blocks):
    m = sum(b[0].shape[0] for b in blocks); n = sum(b[0].shape[1] for b in blocks)
    leq = np.zeros((m, n + 1)); tgtf = np.zeros(n + 1)
    r = c = 0
    for A, b, cc in blocks:
        leq[r:r + A.shape[0], c:c + A.shape[1]] = A
        leq[r:r + A.shape[0], n] = b
        tgtf[c:c + A.shape[1]] = cc
        r += A.shape[0]; c += A.shape[1]
    return leq, tgtf


def cover_block_f64(seed):
    """The dual of lp_block_f64(seed) as a covering block: minimise b.x, A^T x >= c, x >= 0, written in the reference's
    form -A^T x <= -c. Returns (A [n, m], b [n], c [m]) of that form."""
    A, b, c = lp_block_f64(seed)
    return -A.T.copy(), -c, b


def cover_lp_f64(block_seeds):
    """Block-diagonal covering LP for SIX::minm (minimise c.x, A x >= b, x >= 0, every right-hand side negative in the
    a.x <= b form: the origin is infeasible for the primal, SIX::minm solves the dual): the blocks cover_block_f64(s)."""
    return _assemble_blocks([cover_block_f64(int(s)) for s in block_seeds])
