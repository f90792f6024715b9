/* Compiled as C99 by tests/test_abi.py: the boundary header must be plain C. */
#include "xpoly_amd.h"
#include <stdio.h>

int main(void)
{
    xpg_ctx * ctx = 0;
    int rc = xpg_create(&ctx, 0);
    if (rc != 0) { printf("no device: %d\n", rc); return rc == XPG_ERR_NO_DEVICE ? 0 : 1; }
    /* src/example/example.cpp:54-93 through the C ABI */
    double tgtf[3] = {2, -1, 0}, vc[6] = {-1, 0, 0, 0, -1, 0}, leq[6] = {2, -1, 2, 1, -5, -4}, v, sol[3];
    int st = xpg_six_maxm_f64(ctx, tgtf, vc, 2, 0, 0, leq, 2, 3, 0xFFFFFFFFu, &v, sol);
    printf("status %d max %.17g at (%.17g, %.17g)\n", st, v, sol[0], sol[1]);
    xpg_destroy(ctx);
    return (st == XPG_SIX_SUCC && v == 2.0) ? 0 : 1;
}
