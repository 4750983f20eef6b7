"""ctypes binding of libscaml_hip.so (the C ABI declared in include/scaml_gp.h).

The product path has no CPU fallback: if the HIP library is missing or a symbol cannot be
resolved, importing this module raises.  Build it with ``python -c "import __graft_entry__ as
g; g.build()"`` from the repository root (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_uint, c_void_p

# torch must be imported BEFORE the library is loaded: torch ships its own libamdhip64.so.7 and
# libscaml_hip.so links against the same SONAME, so whichever loads first serves both.  Device
# pointers and streams handed across the C ABI come from torch's runtime instance; loading the
# system copy first would give this library a second, unrelated HIP runtime ("no ROCm-capable
# device is detected" at the first launch).
import torch  # noqa: F401  (side effect: HIP runtime loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libscaml_hip.so")

KIND_RBF = 0
KIND_MATERN52 = 1

FIT_STORE_L = 1
FIT_ZERO_UPPER = 2
FIT_NO_RETRY = 4
POST_XQ_PER_TASK = 1
POST_MEAN_ONLY = 2

E_BADARG = -1
E_TOOLARGE = -2
E_LAUNCH = -3

_dp = c_void_p  # device pointers travel as integers


class ScamlLibraryError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ScamlLibraryError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run __graft_entry__.build()); there is no CPU fallback for the GP hot path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.scaml_version.restype = c_int
    lib.scaml_version.argtypes = []
    lib.scaml_last_error.restype = c_char_p
    lib.scaml_last_error.argtypes = []
    lib.scaml_fit_max_n.restype = c_int
    lib.scaml_fit_max_n.argtypes = []
    lib.scaml_fit_max_d.restype = c_int
    lib.scaml_fit_max_d.argtypes = [c_int]
    lib.scaml_fit_blocked_max_n.restype = c_int
    lib.scaml_fit_blocked_max_n.argtypes = []
    lib.scaml_fit_blocked_max_d.restype = c_int
    lib.scaml_fit_blocked_max_d.argtypes = []
    lib.scaml_gp_fit_blocked_workspace_bytes.restype = ctypes.c_longlong
    lib.scaml_gp_fit_blocked_workspace_bytes.argtypes = [c_int, c_int]
    lib.scaml_gp_fit_blocked_f64.restype = c_int
    lib.scaml_gp_fit_blocked_f64.argtypes = [
        _dp, _dp, _dp, _dp, _dp,  # X, y, theta, n_points, jitter_in
        c_int, c_int, c_int, c_int,  # T, N, D, kind
        _dp, _dp, _dp, _dp, _dp,  # L, alpha, quad, logdet, mll
        _dp, _dp, _dp, ctypes.c_uint,  # info, jitter_used, Linv_diag, flags
        _dp, ctypes.c_longlong, _dp,  # workspace, workspace_bytes, stream
    ]
    lib.scaml_gp_fit_fused_f64.restype = c_int
    lib.scaml_gp_fit_fused_f64.argtypes = [
        _dp, _dp, _dp, _dp, _dp,  # X, y, theta, n_points, jitter_in
        c_int, c_int, c_int, c_int,  # T, N, D, kind
        _dp, _dp, _dp, _dp, _dp,  # L, alpha, quad, logdet, mll
        _dp, _dp, _dp, c_uint, c_void_p,  # info, jitter_used, Linv_diag, flags, stream
    ]
    lib.scaml_kernel_matrix_f64.restype = c_int
    lib.scaml_kernel_matrix_f64.argtypes = [_dp, _dp, _dp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _dp, c_void_p]
    lib.scaml_potrf_batched_f64.restype = c_int
    lib.scaml_potrf_batched_f64.argtypes = [_dp, _dp, _dp, _dp, c_int, c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, c_uint, c_void_p]
    lib.scaml_posterior_max_n.restype = c_int
    lib.scaml_posterior_max_n.argtypes = []
    lib.scaml_posterior_batched_f64.restype = c_int
    lib.scaml_posterior_batched_f64.argtypes = [
        _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,  # Xq, X, theta, L, Linv_diag, alpha, y_mean, y_std, n_points
        c_int, c_int, c_int, c_int, c_int,  # T, N, M, D, kind
        _dp, _dp, _dp, ctypes.c_uint, c_void_p,  # mu, var, V, flags, stream
    ]
    lib.scaml_posterior_cov_f64.restype = c_int
    lib.scaml_posterior_cov_f64.argtypes = [_dp, _dp, _dp, _dp, c_int, c_int, c_int, c_int, c_int, c_int, _dp, ctypes.c_uint, c_void_p]
    lib.scaml_linv_batched_f64.restype = c_int
    lib.scaml_linv_batched_f64.argtypes = [_dp, _dp, _dp, c_int, c_int, _dp, c_void_p]
    lib.scaml_linv_batched_lower_f64.restype = c_int
    lib.scaml_linv_batched_lower_f64.argtypes = [_dp, _dp, _dp, c_int, c_int, _dp, c_void_p]
    lib.scaml_posterior_linv_f64.restype = c_int
    lib.scaml_posterior_linv_f64.argtypes = [
        _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,  # Xq, X, theta, Linv, alpha, y_mean, y_std, n_points
        c_int, c_int, c_int, c_int, c_int,  # T, N, M, D, kind
        _dp, _dp, _dp, ctypes.c_uint, c_void_p,  # mu, var, V, flags, stream
    ]
    lib.scaml_cho_solve_batched_f64.restype = c_int
    lib.scaml_cho_solve_batched_f64.argtypes = [_dp, _dp, _dp, _dp, c_int, c_int, c_int, _dp, c_void_p]
    lib.scaml_solve_lt_batched_f64.restype = c_int
    lib.scaml_solve_lt_batched_f64.argtypes = [_dp, _dp, _dp, _dp, c_int, c_int, c_int, _dp, c_void_p]
    lib.scaml_weighted_task_sum_f64.restype = c_int
    lib.scaml_weighted_task_sum_f64.argtypes = [_dp, _dp, _dp, c_int, ctypes.c_longlong, c_int, _dp, c_void_p]
    lib.scaml_weighted_prior_reduce_f64.restype = c_int
    lib.scaml_weighted_prior_reduce_f64.argtypes = [_dp, _dp, _dp, _dp, c_int, c_int, c_int, _dp, _dp, c_void_p]
    lib.scaml_mll_backward_workspace_doubles.restype = ctypes.c_longlong
    lib.scaml_mll_backward_workspace_doubles.argtypes = [c_int, c_int, c_int]
    lib.scaml_mll_backward_f64.restype = c_int
    lib.scaml_mll_backward_f64.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, c_int, c_int, c_int, c_int, _dp, _dp, c_void_p]
    lib.scaml_posterior_linv_cov_f64.restype = c_int
    lib.scaml_posterior_linv_cov_f64.argtypes = [_dp] * 9 + [c_int] * 6 + [_dp, _dp, _dp, ctypes.c_uint, c_void_p]
    c_double = ctypes.c_double
    lib.scaml_target_assemble_f64.restype = c_int
    lib.scaml_target_assemble_f64.argtypes = [_dp] * 6 + [c_double, c_double, c_int, c_int, c_int, c_int] + [_dp] * 5 + [c_void_p]
    lib.scaml_target_finish_f64.restype = c_int
    lib.scaml_target_finish_f64.argtypes = [_dp] * 5 + [c_double, c_double, c_double, _dp, c_int, c_int, _dp, _dp, c_void_p]
    lib.scaml_posterior_linv_grad_f64.restype = c_int
    lib.scaml_posterior_linv_grad_f64.argtypes = [_dp] * 10 + [c_int] * 6 + [_dp, _dp, _dp, ctypes.c_uint, c_void_p]
    lib.scaml_target_posterior_grad_f64.restype = c_int
    lib.scaml_target_posterior_grad_f64.argtypes = [_dp] * 8 + [c_double, _dp, c_int, c_int, c_int, c_int, _dp, _dp, c_void_p]
    lib.scaml_target_fit_max_n.restype = c_int
    lib.scaml_target_fit_max_n.argtypes = [c_int, c_int]
    lib.scaml_target_fit_max_d.restype = c_int
    lib.scaml_target_fit_max_d.argtypes = []
    lib.scaml_target_fit_workspace_doubles.restype = ctypes.c_longlong
    lib.scaml_target_fit_workspace_doubles.argtypes = [c_int, c_int, c_int, c_int]
    host_spec = ctypes.POINTER(c_double)   # (the one host pointer of the ABI: 19 doubles read during the call)
    lib.scaml_target_mll_f64.restype = c_int
    lib.scaml_target_mll_f64.argtypes = [_dp] * 4 + [c_double, c_double, host_spec, _dp] + [c_int] * 5 + [_dp] * 4 + [c_void_p]
    lib.scaml_target_fit_f64.restype = c_int
    lib.scaml_target_fit_f64.argtypes = ([_dp] * 4 + [c_double, c_double, host_spec, _dp] + [c_int] * 7 + [c_double, c_double] + [_dp] * 5
                                         + [ctypes.c_longlong, c_void_p])
    lib.scaml_debug_target_fit_path.restype = c_int
    lib.scaml_debug_target_fit_path.argtypes = [c_int]
    lib.scaml_debug_blocked_fit_path.restype = c_int
    lib.scaml_debug_blocked_fit_path.argtypes = [c_int]
    lib.scaml_debug_coop_far.restype = c_int
    lib.scaml_debug_coop_far.argtypes = [c_int]
    lib.scaml_debug_force_two_launch_grad.restype = c_int
    lib.scaml_debug_force_two_launch_grad.argtypes = [c_int]
    return lib


lib = _load()

# Every symbol include/scaml_gp.h declares; tests check the built library exports them all.
EXPORTED_SYMBOLS = (
    "scaml_version",
    "scaml_last_error",
    "scaml_fit_max_n",
    "scaml_fit_max_d",
    "scaml_gp_fit_fused_f64",
    "scaml_fit_blocked_max_n",
    "scaml_fit_blocked_max_d",
    "scaml_gp_fit_blocked_workspace_bytes",
    "scaml_gp_fit_blocked_f64",
    "scaml_kernel_matrix_f64",
    "scaml_potrf_batched_f64",
    "scaml_posterior_max_n",
    "scaml_posterior_batched_f64",
    "scaml_posterior_cov_f64",
    "scaml_linv_batched_f64",
    "scaml_linv_batched_lower_f64",
    "scaml_posterior_linv_f64",
    "scaml_cho_solve_batched_f64",
    "scaml_solve_lt_batched_f64",
    "scaml_weighted_task_sum_f64",
    "scaml_weighted_prior_reduce_f64",
    "scaml_mll_backward_workspace_doubles",
    "scaml_mll_backward_f64",
    "scaml_posterior_linv_cov_f64",
    "scaml_target_assemble_f64",
    "scaml_target_finish_f64",
    "scaml_posterior_linv_grad_f64",
    "scaml_target_posterior_grad_f64",
    "scaml_target_fit_max_n",
    "scaml_target_fit_max_d",
    "scaml_target_fit_workspace_doubles",
    "scaml_target_mll_f64",
    "scaml_target_fit_f64",
)


def check_rc(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == E_BADARG:
        raise ValueError(f"{what}: bad argument")
    if rc == E_TOOLARGE:
        raise ValueError(f"{what}: problem size exceeds kernel limits")
    msg = lib.scaml_last_error().decode("utf-8", "replace")
    raise RuntimeError(f"{what}: HIP launch failed ({msg})")
