"""Which kernels serve MIPs with root equalities (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import xpoly_amd
from oracle.checker import Port
from mip_eq_cases import run
ctx = xpoly_amd.Context(0)
print(run(ctx, Port(), 1, 4243, 120))
