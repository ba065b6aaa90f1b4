"""ctypes binding of the C-ABI in include/tinyopt_amd.h.

Plumbing only: loads ``libtinyopt_amd.so`` (built in-tree by ``__graft_entry__.build()``) and
declares the prototypes.  There is NO fallback: if the HIP library is missing or fails to load the
import raises — the product path never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TINYOPT_AMD_LIB lets experiments load an alternative build of the SAME C-ABI (never a CPU path)
LIB_PATH = os.environ.get("TINYOPT_AMD_LIB", os.path.join(_HERE, "libtinyopt_amd.so"))

F32, F64 = 0, 1
MODEL_DENSE_ROW, MODEL_GAUSSIAN_PRIOR, MODEL_SQRT2, MODEL_SE3_REPROJ, MODEL_CIRCLE_FIT, MODEL_DENSE_ROW_AD6 = 1, 2, 3, 4, 5, 6
MODEL_TESTFN = 7
MODEL_MAHA_PRIOR = 8
MODEL_SE3_PRIOR = 9
MODEL_DENSE_ROW_NATURAL = 10
MODEL_DENSE_ROW_AD = 11

# StopReason, same integers as include/tinyopt/stop_reasons.h:14-43
STOP_NAMES = {
    -4: "kOutOfMemory", -3: "kSolverFailed", -2: "kSystemHasNaNOrInf", -1: "kSkipped", 0: "kNone",
    1: "kMinError", 2: "kMinRelError", 3: "kMinDeltaNorm", 4: "kMinGradNorm", 5: "kMaxIters",
    6: "kMaxNoDecr", 7: "kMaxConsecNoDecr", 8: "kTimedOut", 9: "kUserStopped",
}


class ToaOptions(C.Structure):
    """POD mirror of tinyopt::Options (include/tinyopt/optimizers/options.h:18-156)."""
    _fields_ = [
        ("solver_type", C.c_int32), ("max_iters", C.c_int32),
        ("min_error", C.c_float), ("min_rerr_dec", C.c_float),
        ("min_step_norm2", C.c_float), ("min_grad_norm2", C.c_float),
        ("max_total_failures", C.c_int32), ("max_consec_failures", C.c_int32),
        ("damping_init", C.c_float), ("damping_min", C.c_float), ("damping_max", C.c_float),
        ("good_factor", C.c_float), ("bad_factor", C.c_float),
        ("grad_clipping", C.c_float), ("check_min_H_diag", C.c_float),
        ("check_final_cost", C.c_uint8), ("use_step_quality_approx", C.c_uint8),
        ("use_ldlt", C.c_uint8), ("H_is_full", C.c_uint8), ("save_last", C.c_uint8),
        ("use_squared_norm", C.c_uint8), ("downscale_by_2", C.c_uint8), ("normalize", C.c_uint8),
    ]


class ToaResults(C.Structure):
    """POD mirror of tinyopt::Output (include/tinyopt/output.h:26-145); device pointers."""
    _fields_ = [
        ("stop_reason", C.c_void_p), ("num_iters", C.c_void_p), ("num_failures", C.c_void_p),
        ("num_consec_failures", C.c_void_p), ("final_cost", C.c_void_p),
        ("final_num_residuals", C.c_void_p), ("final_rerr_dec", C.c_void_p),
        ("final_hessian", C.c_void_p), ("errs", C.c_void_p), ("deltas2", C.c_void_p),
        ("successes", C.c_void_p), ("hist_stride", C.c_int32), ("_pad", C.c_int32),
        ("final_inlier_ratio", C.c_void_p),
    ]


# every symbol include/tinyopt_amd.h declares: name -> (restype, argtypes)
ABI_VERSION = 6   # include/tinyopt_amd.h TOA_ABI_VERSION


class ToaTuning(C.Structure):   # include/tinyopt_amd.h toa_tuning
    _fields_ = [(k, C.c_int32) for k in ("memo_off", "coop_off", "coop_chunks", "max_workgroups", "wide_no_autosplit", "wide_multilaunch",
                                         "wide_no_team", "wide_team_max_per_cu", "wide_graph", "large_row_split", "large_pipeline",
                                         "large_library_gram", "large_library_solver", "fail_workspace_alloc", "large_one_lane", "large_chol_no_lookahead", "large_gram_plain_deal", "narrow_mfma_pass", "se3_reproj_header_l2")] + [("reserved", C.c_int32 * 13)]


class ToaJitSpec(C.Structure):   # include/tinyopt_amd.h toa_jit_spec
    _fields_ = [("dtype", C.c_int32), ("num_params", C.c_int32), ("residuals_per_item", C.c_int32), ("scalars_per_item", C.c_int32),
                ("header_scalars", C.c_int32), ("manifold", C.c_int32), ("kind", C.c_int32), ("x_scalars", C.c_int32),
                ("plus_body", C.c_char_p), ("reserved", C.c_int32 * 6)]


_P = C.c_void_p
PROTOTYPES = {
    "toa_options_default": (None, [C.POINTER(ToaOptions)]),
    "toa_options_benchmark": (None, [C.POINTER(ToaOptions)]),
    "toa_create": (C.c_int, [C.POINTER(_P), C.c_int, _P]),
    "toa_destroy": (C.c_int, [_P]),
    "toa_last_error": (C.c_char_p, []),
    "toa_device_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "toa_lm_state_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int64]),
    "toa_lm_begin": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P]),
    "toa_lm_step": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P, _P, _P]),
    "toa_lm_step_info": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "toa_lm_stop": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P, _P, _P]),
    "toa_shard_range": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "toa_comm_unique_id": (C.c_int, [_P]),
    "toa_comm_init_rank": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "toa_comm_destroy": (C.c_int, [_P]),
    "toa_gather": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, _P, C.POINTER(ToaResults), C.c_int, _P, C.POINTER(ToaResults)]),
    "toa_jet_eval": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, _P, _P]),
    "toa_ba_run": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P]),
    "toa_set_loss": (C.c_int, [_P, C.c_int, C.c_double]),
    "toa_robust_norm": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, C.c_double, _P, _P]),
    "toa_hbm_read_probe": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "toa_llc_read_probe": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_double)]),
    "toa_malloc": (C.c_int, [_P, C.POINTER(_P), C.c_size_t]),
    "toa_free": (C.c_int, [_P, _P]),
    "toa_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "toa_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "toa_memset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "toa_synchronize": (C.c_int, [_P]),
    "toa_dense_row_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "toa_dense_row_pack": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, _P]),
    "toa_dense_row_synth": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint64, C.c_int64, _P, _P, _P]),
    "toa_accumulate": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, C.c_int, _P, _P, _P, _P]),
    "toa_solve_damped": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, _P, C.c_double, _P, _P]),
    "toa_inv_cov": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, _P, _P]),
    "toa_lm_run": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions),
                             C.POINTER(ToaResults), _P]),
    "toa_ba_lists_run": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, _P, _P, _P, C.POINTER(ToaOptions),
                                   C.POINTER(ToaResults), _P, C.c_double]),
    "toa_model_compile": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    "toa_abi_version": (C.c_int, []),
    "toa_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "toa_set_tuning": (C.c_int, [_P, C.POINTER(ToaTuning)]),
    "toa_get_tuning": (C.c_int, [_P, C.POINTER(ToaTuning)]),
    "toa_debug_timeline": (C.c_int, [_P, C.c_char_p]),
    "toa_model_compile_ex": (C.c_int, [_P, C.POINTER(ToaJitSpec), C.c_char_p, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    "toa_jit_set_cache_dir": (C.c_int, [C.c_char_p]),
    "toa_lm_step_log": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, _P, _P, _P]),
    "toa_jit_model_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "toa_jit_model_stats": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "toa_model_destroy": (C.c_int, [_P]),
    "toa_jit_lm_run": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P]),
    "toa_jit_accumulate": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, C.c_int, _P, _P, _P, _P]),
    "toa_jit_lm_run_split": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P, C.c_int]),
    "toa_jit_lm_begin": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P]),
    "toa_jit_lm_step": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P, _P, _P]),
    "toa_jit_lm_stop": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions), C.POINTER(ToaResults), _P, _P, _P]),
    "toa_lm_run_split": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, C.POINTER(ToaOptions),
                                   C.POINTER(ToaResults), _P, C.c_int]),
}

_lib = None


def load():
    """Load the shared library once.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "tinyopt_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.toa_abi_version()
    if got != ABI_VERSION:   # (e.g. counters_dev grew from [4] to [8] uint64 with version 4: a stale mirror would be written out of bounds)
        raise RuntimeError(f"{LIB_PATH} has ABI version {got}, this Python mirror expects {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


class ToaError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        msg = load().toa_last_error()
        raise ToaError(f"tinyopt_amd error {rc}: {msg.decode() if msg else ''}")
