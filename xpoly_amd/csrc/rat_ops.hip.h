// Arithmetic of the small-problem kernels (LDS-resident LPs, row elimination), selected per PROBLEM.
// Every RESULT of the reference's Rational arithmetic is canonical (lowest terms, den > 0) -- but not every value it
// can hold: Rational(INT, INT) stores num / den as given (rational.cpp:60-64: 2/4, 0/3, 1/-2 stay), so inputs are
// checked. A problem whose input cells all are canonical -- checked when it is loaded, the objective's constant
// included -- stays so under these operations; it takes the 32-bit
// cross-cancelling forms of scalar.hip.h, which equal the reference's operations bit for bit on such operands
// (tests/cxx/fma_canon_fuzz.cpp). Any other problem takes the generic forms -- valid for any (num, den), out of line.
// For Float the flag is ignored.
#pragma once
#include "scalar.hip.h"

namespace xpg {

// (the generic forms without a divide: scalar.hip.h, equal to add / mul / div for every operand pair). Inlined since
// round 3: out of line every call saved and restored registers through scratch, and the MIP tree walk -- whose substituted
// nodes are never canonical -- ran 10-15 % slower (1024 knapsacks 200 k -> 229 k MIPs/s, 8192: 781 k -> 859 k) for a library
// that builds in 75 s instead of 170 (-DXPG_ANY_INLINE=__noinline__ for quick builds).
#ifndef XPG_ANY_INLINE
#define XPG_ANY_INLINE __forceinline__
#endif
__device__ XPG_ANY_INLINE R32 add_any(R32 a, R32 b) { return add_any_fast(a, b); }
__device__ XPG_ANY_INLINE R32 mul_any(R32 a, R32 b) { return mul_any_fast(a, b); }
__device__ XPG_ANY_INLINE R32 div_any(R32 a, R32 b) { return div_any_fast(a, b); }
// Out-of-line twins for the HBM-resident LP loop (lp_kernels.hip.h, lp_pipe_r32.hip.h): its launches are small and
// start with a cold instruction cache, its problems are canonical unless the caller's input was not, and the sweep
// launch lives on a register budget (k_pipe_sweep_r32) -- inlining the generic forms there cost 8 % of the loop.
__device__ __noinline__ R32 add_any_ol(R32 a, R32 b) { return add_any_fast(a, b); }
__device__ __noinline__ R32 mul_any_ol(R32 a, R32 b) { return mul_any_fast(a, b); }
__device__ __noinline__ R32 div_any_ol(R32 a, R32 b) { return div_any_fast(a, b); }
__device__ __forceinline__ R32 l_fma(bool cn, R32 a, R32 k, R32 e) { return cn ? fma_canon(a, k, e) : add_any_ol(a, mul_any_ol(k, e)); }
__device__ __forceinline__ R32 l_div(bool cn, R32 a, R32 b) { return cn && b.num != 0 ? div_canon(a, b) : div_any_ol(a, b); }
__device__ __forceinline__ F64 l_div(bool, F64 a, F64 b) { return div(a, b); }
__device__ __forceinline__ R32 q_add(bool cn, R32 a, R32 b) { return cn ? add_canon(a, b) : add_any(a, b); }
__device__ __forceinline__ R32 q_sub(bool cn, R32 a, R32 b) { return q_add(cn, a, neg(b)); }
__device__ __forceinline__ R32 q_mul(bool cn, R32 a, R32 b) { return cn ? mul_canon(a, b) : mul_any(a, b); }
__device__ __forceinline__ R32 q_div(bool cn, R32 a, R32 b) { return cn && b.num != 0 ? div_canon(a, b) : div_any(a, b); }
// add(a, mul(k, e)): the cell of a pivot sweep (lpsol.h:1481-1490) and of mul_and_add_row (matt.h:1493-1501)
__device__ __forceinline__ R32 q_fma(bool cn, R32 a, R32 k, R32 e) { return cn ? fma_canon(a, k, e) : add_any(a, mul_any(k, e)); }
__device__ __forceinline__ R32 q_scaled(bool cn, R32 cell, R32 x, int mode)
{
    return mode == SCALE_KEEP ? cell : (mode == SCALE_ZERO ? R32(0, 1) : q_mul(cn, cell, x));
}
__device__ __forceinline__ bool q_canonical(R32 a) { return canonical(a); }

__device__ __forceinline__ F64 q_add(bool, F64 a, F64 b) { return add(a, b); }
__device__ __forceinline__ F64 q_sub(bool, F64 a, F64 b) { return sub(a, b); }
__device__ __forceinline__ F64 q_mul(bool, F64 a, F64 b) { return mul(a, b); }
__device__ __forceinline__ F64 q_div(bool, F64 a, F64 b) { return div(a, b); }
__device__ __forceinline__ F64 q_fma(bool, F64 a, F64 k, F64 e) { return add(a, mul(k, e)); }
__device__ __forceinline__ F64 q_scaled(bool, F64 cell, F64 x, int mode) { return scaled(cell, x, mode); }
__device__ __forceinline__ bool q_canonical(F64) { return true; }

} // namespace xpg
