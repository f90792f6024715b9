// The two pieces of the row-elimination code that the dependence-test front end (mip_host.hip.h) shares with it and
// that reference no kernel: the LDS footprint of one system (does the on-device reduce fit?) and Lineq::move2var on
// one host matrix. Kept apart from lineq_kernels.hip.h so that the MIP translation unit does not compile those kernels.
#pragma once
#include "scalar.hip.h"

namespace xpg {

// LDS bytes of one system of at most cap rows (the carve of carve_scratch, lineq_kernels.hip.h)
__host__ __device__ inline size_t lineq_lds_bytes(int cap, int cols)
{
    size_t b = (size_t)cap * cols * 8;
    b += (size_t)(cap + 1) * 4 + 16 + (size_t)((cap + 1) & ~1) * 4 + (size_t)((cap + 3) & ~3);
    return (b + 15) & ~(size_t)15;
}

// Lineq::move2var (src/com/linsys.cpp:1177-1200) for one system: the constant symbols first_sym..last_sym become
// variables in front of the constant column rhs_idx -- taken out, multiplied by -1 with the scalar's own '*'
// (Matrix::mul, matt.h:1331-1348: 2/4 comes back as -1/2) and inserted before column rhs_idx. Shape unchanged.
inline void move2var_one(const R32 * in, R32 * out, int rows, int cols, int rhs_idx, int first_sym, int last_sym)
{
    for (int i = 0; i < rows; i++) {
        const R32 * src = in + (size_t)i * cols;
        R32 * dst = out + (size_t)i * cols;
        int c = 0;
        for (int j = 0; j < rhs_idx; j++) dst[c++] = src[j];
        for (int j = first_sym; j <= last_sym; j++) dst[c++] = mul(src[j], R32(-1, 1));
        for (int j = rhs_idx; j < cols; j++)
            if (j < first_sym || j > last_sym) dst[c++] = src[j];
    }
}

} // namespace xpg
