#!/usr/bin/env python3
"""Searches, with the restatement (oracle/_build), the block seeds of the large LPs that END in SIX_SUCC with a non-zero optimum
(tools/gen.py block_lp_f64 / cover_lp_f64; fixtures tests/golden/g12_end_states.json via tools/gen_golden_end.py):

    python tools/find_blocks.py max ROWS [COLS] OUT.json [wide]      blocks whose own SIX::maxm ends 0, until >= ROWS rows (>= COLS variables)
    python tools/find_blocks.py min ROWS OUT.json                    covering blocks whose own SIX::minm ends 0

The reference's pricing takes the first column with a positive cost, so a block-diagonal LP is solved block after block and
ends SIX_SUCC when every block does; the whole LP is solved once more here to confirm it."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.checker import F64, Port  # noqa: E402
from tools import gen  # noqa: E402


def main():
    port = Port()
    mode, rows = sys.argv[1], int(sys.argv[2])
    args = sys.argv[3:]
    wide = "wide" in args
    args = [a for a in args if a != "wide"]
    cols = int(args[0]) if len(args) > 1 else 0
    out = args[-1]
    good, r, c, s, hist = [], 0, 0, 0, {}
    while r < rows or c < cols:
        A, b, cc = gen.lp_block_f64(s, wide) if mode == "max" else gen.cover_block_f64(s)
        leq = np.concatenate([A, b[:, None]], axis=1); tg = np.concatenate([cc, [0.0]])
        st, v, _ = port.six_solve(F64, mode == "max", tg, gen.vc_nonneg(A.shape[1]), None, leq)
        hist[st] = hist.get(st, 0) + 1
        if st == 0 and float(v) != 0.0:
            good.append(s); r += A.shape[0]; c += A.shape[1]
        s += 1
    leq, tg = gen.block_lp_f64(good, wide) if mode == "max" else gen.cover_lp_f64(good)
    print("tried", s, "kept", len(good), "statuses", hist, "shape", leq.shape, flush=True)
    t0 = time.time(); p0 = port.pivot_count()
    st, v, sol = port.six_solve(F64, mode == "max", tg, gen.vc_nonneg(leq.shape[1] - 1), None, leq)
    print("whole: status", st, "v", float(v), "pivots", port.pivot_count() - p0, "%.1f s" % (time.time() - t0))
    assert st == 0
    json.dump(good, open(out, "w"))


if __name__ == "__main__":
    main()
