"""The randomized GPU-vs-oracle cross-checks (tests/crosscheck_*.py) as part of the driver-run suite.

Each script is a program of its own (module-level loops, a summary line with its mismatch count); here every one runs
as a child process at a fixed seed and a size that finishes in seconds, and the test asserts that it compared
something and that every mismatch count it printed is zero. XPG_CROSSCHECK_FULL=1 runs the by-hand sizes instead
(minutes). The scripts found three real divergences in round 3 (commits 569d75d, 9820111, 0e98101) while nobody but
the author ran them.
"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FULL = os.environ.get("XPG_CROSSCHECK_FULL") == "1"

# script, quick arguments, by-hand arguments
CASES = [
    ("crosscheck_six_eq.py", ["60"], ["400"]),              # SIX::maxm / minm + has_solution with equalities, both scalars
    ("crosscheck_mip_eq.py", ["120"], ["2000"]),            # MIP with root equalities, both scalars
    ("crosscheck_unordered.py", ["1"], ["1"]),              # n/0 cells: unordered comparisons in the ratio tests
    ("crosscheck_nonfinite_f64.py", [], []),                # inf / NaN cells in fp64 LPs
    ("crosscheck_batch_weird.py", ["1"], ["1"]),            # LDS batch kernel on non-canonical / n/0 input
    ("crosscheck_lineq_weird.py", ["7", "64"], ["7", "256"]),   # reduce / fme / gauss on non-canonical input
    ("crosscheck_mip_weird.py", ["5"], ["5"]),              # device tree walk + dep front end on non-canonical input
    ("crosscheck_pivot_weird.py", ["9"], ["9"]),            # K1 on non-canonical / non-finite cells
    ("crosscheck_batch_rat.py", ["5", "24"], ["5", "512"]),     # rational LDS batches, several shapes
    ("crosscheck_dep.py", ["1", "192"], ["1", "1024"]),     # DepPoly::is_empty front end
    ("crosscheck_lineq.py", ["2026", "128"], ["2026", "768"]),  # reduce / fme / calcBound / gauss / hnf
    ("crosscheck_mip.py", ["3", "48"], ["3", "256"]),       # MIP batches (integer and 0-1)
]

# "TOTAL mismatches: 0", "compared 382 mismatches 0 ...", "mismatching LPs 0"  |  "..., 0 mismatches so far", "..., 0 mismatches"
COUNT = re.compile(r"mismatch(?:es|ing LPs):?[ \t]+(\d+)|(\d+)[ \t]+mismatches(?![ \t]*:?[ \t]*\d)")


@pytest.mark.parametrize("script,quick,full", CASES, ids=[c[0][len("crosscheck_"):-3] for c in CASES])
def test_crosscheck(script, quick, full):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(HERE, script)] + (full if FULL else quick), cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800 if FULL else 300)
    out = r.stdout
    assert r.returncode == 0, out[-3000:]
    counts = [int(a or b) for a, b in COUNT.findall(out)]
    assert counts, "no summary line:\n" + out[-2000:]
    assert all(c == 0 for c in counts), out[-3000:]
    assert "MISMATCH " not in out, out[-3000:]
    assert re.search(r"(compared|checked|polyhedra|matrices)\D{0,40}[1-9]\d*|[1-9]\d*\s+(compared|LPs|polyhedra|matrices|checked)", out), out[-2000:]
