/* TEST INFRASTRUCTURE -- C ABI of the CPU restatement (oracle/oracle.cpp).
 * Mirrors oracle/ref_driver.cpp with the prefix orc_. See oracle/README.md. */
#ifndef XPOLY_ORACLE_H
#define XPOLY_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int orc_six_solve(int kind, int is_max, const void * tgtf, const void * vc, int vc_rows,
                  const void * eq, int eq_rows, const void * leq, int leq_rows, int cols,
                  unsigned max_iter, void * out_v, void * out_sol);
int orc_two_stage(int kind, const void * leq, int m, int cols, const void * vc,
                  const void * tgtf, unsigned max_iter, void * out_tab, int * out_rows,
                  int * out_cols, void * out_tgtf, uint8_t * out_nvset, uint8_t * out_bvset,
                  int32_t * out_bv2eq, int32_t * out_eq2bv, int * out_rhs, void * out_maxv,
                  void * out_sol);
int orc_two_stage_trace(int kind, const void * leq, int m, int cols, const void * vc,
                        const void * tgtf, unsigned max_iter, void * out_tab, int * out_rows,
                        int * out_cols, void * out_tgtf, uint8_t * out_nvset,
                        uint8_t * out_bvset, int32_t * out_bv2eq, int32_t * out_eq2bv,
                        int * out_rhs, void * out_maxv, void * out_sol, int32_t * out_trace,
                        int trace_cap, int * out_trace_len);
int orc_mip_solve(int kind, int is_max, int is_bin, const void * tgtf, const void * vc,
                  int vc_rows, const void * eq, int eq_rows, const void * leq, int leq_rows,
                  int cols, const uint8_t * rat_ind, void * out_v, void * out_sol);
int orc_mip_solve_stats(int kind, int is_max, int is_bin, const void * tgtf, const void * vc,
                        int vc_rows, const void * eq, int eq_rows, const void * leq, int leq_rows,
                        int cols, const uint8_t * rat_ind, void * out_v, void * out_sol, long * nodes,
                        int * max_leq_rows);
void orc_rat_op(int op, int32_t an, int32_t ad, int32_t bn, int32_t bd, int32_t * rn,
                int32_t * rd);
int orc_rat_cmp(int cmp, int32_t an, int32_t ad, int32_t bn, int32_t bd);
int orc_flt_cmp(int cmp, double x, double y);
void orc_pivot_f64(double * tab, int m, int W, double * obj, int rhs_idx, int row, int col);
void orc_pivot_rat32(int32_t * tab, int m, int W, int32_t * obj, int rhs_idx, int row, int col);
void orc_set_strict(int on);
long long orc_appro_count(void);
long long orc_reduce_count(void);
#ifdef __cplusplus
}
#endif
#endif
