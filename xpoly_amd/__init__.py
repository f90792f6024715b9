"""xpoly_amd -- MI355X-native simplex / row-elimination kernels behind xpoly's
SIX / MIP / Lineq interfaces. See DESIGN.md; the C ABI is include/xpoly_amd.h."""
from .six import (F64, RAT, SIX, Context, DeviceLP, SIX_SUCC, SIX_UNBOUND,  # noqa: F401
                  SIX_NO_PRI_FEASIBLE_SOL, SIX_OPTIMAL_IS_INFEASIBLE, SIX_TIME_OUT, XpgError)
