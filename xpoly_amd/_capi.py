"""ctypes binding of include/xpoly_amd.h. Loads xpoly_amd/libxpoly_amd.so and
fails loudly if it is missing: there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("XPG_SO_PATH") or os.path.join(HERE, "libxpoly_amd.so")   # override: A/B builds

XPG_RUNNING = -1000
ERRORS = {-1: "XPG_ERR_HIP", -2: "XPG_ERR_ALLOC", -3: "XPG_ERR_SHAPE", -4: "XPG_ERR_UNSUPPORTED",
          -5: "XPG_ERR_NO_DEVICE", -7: "XPG_ERR_REF_UNDEFINED", -8: "XPG_ERR_CHAIN_STUCK"}

# every symbol include/xpoly_amd.h declares
SYMBOLS = [
    "xpg_device_count", "xpg_create", "xpg_destroy", "xpg_last_error", "xpg_version", "xpg_stream",
    "xpg_sync", "xpg_profile_begin", "xpg_profile_end", "xpg_malloc", "xpg_free", "xpg_upload", "xpg_download",
    "xpg_pivot_f64_dev", "xpg_pivot_rat32_dev", "xpg_pivot_f64", "xpg_pivot_rat32",
    "xpg_lp_create", "xpg_lp_destroy", "xpg_lp_two_stage", "xpg_lp_begin", "xpg_lp_iterate",
    "xpg_lp_pivots_done", "xpg_lp_counters", "xpg_lp_chain_folds", "xpg_mip_batch_eq_rat32", "xpg_mip_batch_eq_f64", "xpg_lp_set_options", "xpg_lp_shape", "xpg_lp_read", "xpg_lp_trace",
    "xpg_six_maxm_f64", "xpg_six_minm_f64", "xpg_six_maxm_rat32", "xpg_six_minm_rat32",
    "xpg_six_batch_f64", "xpg_six_batch_rat32", "xpg_six_batch_f64_dev", "xpg_six_batch_rat32_dev",
    "xpg_mip_maxm_rat32", "xpg_mip_minm_rat32", "xpg_mip_maxm_f64", "xpg_mip_minm_f64",
    "xpg_has_solution_rat32", "xpg_mip_batch_rat32", "xpg_mip_batch_f64", "xpg_dep_is_empty_batch_rat32",
    "xpg_lineq_reduce_batch_rat32", "xpg_lineq_remove_iden_batch_rat32", "xpg_lineq_fme_batch_rat32",
    "xpg_lineq_calc_bound_batch_rat32", "xpg_lineq_reduce_batch_rat32_dev", "xpg_lineq_fme_batch_rat32_dev",
    "xpg_rat_rank_batch_dev", "xpg_rat_rank_batch", "xpg_rat_det_batch", "xpg_rat_inv_batch",
    "xpg_rat_rank_basis_batch", "xpg_rat_null_batch", "xpg_int_hnf_batch", "xpg_int_gcd_batch",
    "xpg_six_batch_f64_multi", "xpg_six_batch_rat32_multi", "xpg_mip_batch_rat32_multi",
    "xpg_dep_is_empty_batch_rat32_multi", "xpg_dep_is_empty_batch_ex_rat32", "xpg_lineq_move2var_batch_rat32", "xpg_mip_warm_f64", "xpg_mip_warm_batch_f64", "xpg_six_last_profile", "xpg_lineq_calc_bound_batch_packed_rat32",
    "xpg_lineq_fme_batch_packed_rat32", "xpg_trim", "xpg_lp_chain_aborts", "xpg_test_sweep_tile", "xpg_test_pick_ld", "xpg_test_canon_ops_rat32", "xpg_test_any_ops_rat32",
    "xpg_six_batch_f64_ragged", "xpg_six_batch_rat32_ragged", "xpg_dep_is_empty_batch_ragged_rat32",
    "xpg_lineq_reduce_batch_ragged_rat32", "xpg_lineq_fme_batch_ragged_rat32", "xpg_lineq_reduce_batch_packed_rat32", "xpg_lp_loop_info", "xpg_test_normalize", "xpg_dep_is_empty_batch_mode_rat32",
]

_lib = None


class XpgError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise XpgError(
                "%s is missing: build it with `python -m xpoly_amd.build` (hipcc, gfx950). "
                "xpoly_amd has no CPU fallback." % SO_PATH)
        _lib = C.CDLL(SO_PATH)
        _lib.xpg_last_error.restype = C.c_char_p
        _lib.xpg_version.restype = C.c_char_p
        _lib.xpg_stream.restype = C.c_void_p
        for name in SYMBOLS:
            fn = getattr(_lib, name)
            if fn.restype is C.c_int:
                fn.restype = C.c_int
    return _lib


def vp(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)
