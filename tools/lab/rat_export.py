"""Exports states of the cfg-4 rational LP (1024 x 2048) for tools/lab/rat_sweep_lab.hip: for each pivot K the
tableau before it, the scaled pivot row e, the negated pivot column k, the pivot row index and the tableau after it
(the oracle's), as raw little-endian int32 (num, den) pairs under tools/lab/_data/ (git-ignored, travels with gpurun).
PYTHONPATH=. python tools/lab/rat_export.py 4 12 16"""
import os, sys, numpy as np
sys.path.insert(0, ".")
import bench
from tools import gen
from oracle.checker import Port, RAT

def main():
    Ks = [int(a) for a in sys.argv[1:]] or [4, 12, 16]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_data"); os.makedirs(out, exist_ok=True)
    port = Port()
    leq, tgtf = gen.int_lp_rat(bench.RAT_M, bench.RAT_N)
    for K in Ks:
        a = port.two_stage(RAT, leq, tgtf, K - 1); b = port.two_stage(RAT, leq, tgtf, K)
        r = int(np.nonzero(a["eq2bv"] != b["eq2bv"])[0][0]); c = int(b["eq2bv"][r])
        T = a["tab"].astype(np.int32); U = b["tab"].astype(np.int32)
        e = U[r].copy()
        k = T[:, c].copy(); k[:, 0] = -k[:, 0]                        # neg(): -num / den
        hdr = np.array([T.shape[0], T.shape[1], r, c], np.int32)
        with open(os.path.join(out, "pivot%02d.bin" % K), "wb") as f:
            f.write(hdr.tobytes()); f.write(T.tobytes()); f.write(e.tobytes()); f.write(k.tobytes()); f.write(U.tobytes())
        print("pivot", K, "row", r, "col", c, T.shape, "changed cells", int((T != U).any(axis=2).sum()))

if __name__ == "__main__":
    main()
