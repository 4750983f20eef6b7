"""Torch-tensor front end of the C ABI (device memory, streams; no arithmetic here).

Every function takes CUDA(=HIP) fp64 tensors, passes raw device pointers plus the current
stream to libscaml_hip.so and returns freshly allocated output tensors.  Nothing is
synchronised; ``info`` tensors stay on the device until the caller inspects them.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from ._lib import KIND_MATERN52, KIND_RBF  # noqa: F401  (re-export)


class NotPSDError(RuntimeError):
    """Cholesky failed for at least one task even after the jitter escalation
    (mirrors linear_operator.utils.errors.NotPSDError, a RuntimeError the reference
    catches at scamlgp/utils.py:180,193)."""


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _check(t: torch.Tensor, name: str, shape=None, dtype=torch.float64) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); the GP hot path has no CPU fallback")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype} (got {t.dtype})")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)} (got {tuple(t.shape)})")
    return t.contiguous()


def _stream_handle() -> int:
    return torch.cuda.current_stream().cuda_stream


def gp_fit_fused(
    X: torch.Tensor,
    y: torch.Tensor,
    theta: torch.Tensor,
    kind: int,
    n_points: Optional[torch.Tensor] = None,
    jitter: Optional[torch.Tensor] = None,
    store_L: bool = True,
    zero_upper: bool = True,
    retry: bool = True,
    want_alpha: bool = True,
    want_linv: bool = False,
    out: Optional[Dict[str, torch.Tensor]] = None,
) -> Dict[str, torch.Tensor]:
    """K + noise, jittered Cholesky, alpha, quad, logdet, MLL for a stack of tasks.

    X (T, N, D), y (T, N), theta (T, D+2) = [lengthscales, outputscale, noise] (constrained).
    Returns dict(L, alpha, quad, logdet, mll, info, jitter, Linv_diag); ``L`` is None when
    store_L=False, ``Linv_diag`` (T, ceil(N/16), 16, 16) only with want_linv=True.
    One launch of ``scaml_gp_fit_fused_f64`` (include/scaml_gp.h).  ``out`` may be the dict a
    previous call returned for the same shapes/flags: its tensors are reused (no allocation).
    """
    if X.dim() != 3:
        raise ValueError("X must be (T, N, D)")
    T, N, D = X.shape
    X = _check(X, "X")
    y = _check(y, "y", (T, N))
    theta = _check(theta, "theta", (T, D + 2))
    if n_points is not None:
        n_points = _check(n_points, "n_points", (T,), torch.int32)
    if jitter is not None:
        jitter = _check(jitter, "jitter", (T,))
    dev = X.device
    with torch.cuda.device(dev):
        if out is not None:
            L, alpha, quad, logdet, mll, info, jit_used = (out[k] for k in ("L", "alpha", "quad", "logdet", "mll", "info", "jitter"))
            linv = out.get("Linv_diag")
            if (L is None) == store_L or (alpha is None) == want_alpha or quad.shape != (T,) or (store_L and L.shape != (T, N, N)):
                raise ValueError("out= does not match this call's shapes/flags")
        else:
            L = torch.empty((T, N, N), dtype=torch.float64, device=dev) if store_L else None
            alpha = torch.empty((T, N), dtype=torch.float64, device=dev) if want_alpha else None
            quad = torch.empty((T,), dtype=torch.float64, device=dev)
            logdet = torch.empty((T,), dtype=torch.float64, device=dev)
            mll = torch.empty((T,), dtype=torch.float64, device=dev)
            info = torch.empty((T,), dtype=torch.int32, device=dev)
            jit_used = torch.empty((T,), dtype=torch.float64, device=dev)
            linv = torch.empty((T, (N + 15) // 16, 16, 16), dtype=torch.float64, device=dev) if want_linv else None
        if n_points is not None and want_alpha:
            alpha.zero_()
        flags = 0
        if store_L:
            flags |= _lib.FIT_STORE_L
            if zero_upper:
                flags |= _lib.FIT_ZERO_UPPER
            if n_points is not None:
                L.zero_()
        if not retry:
            flags |= _lib.FIT_NO_RETRY
        rc = _lib.lib.scaml_gp_fit_fused_f64(
            _ptr(X), _ptr(y), _ptr(theta), _ptr(n_points), _ptr(jitter),
            T, N, D, int(kind),
            _ptr(L), _ptr(alpha), _ptr(quad), _ptr(logdet), _ptr(mll),
            _ptr(info), _ptr(jit_used), _ptr(linv), flags, _stream_handle(),
        )
    _lib.check_rc(rc, "scaml_gp_fit_fused_f64")
    return dict(L=L, alpha=alpha, quad=quad, logdet=logdet, mll=mll, info=info, jitter=jit_used, Linv_diag=linv)


def kernel_matrix(X1: torch.Tensor, theta: torch.Tensor, kind: int, X2: Optional[torch.Tensor] = None,
                  add_noise: bool = False) -> torch.Tensor:
    """K (T, N1, N2) = os k(X1 / l, X2 / l).  X2 None: square training matrix (optionally + noise I);
    X2 (N2, D): one query set shared by all tasks; X2 (T, N2, D): per-task.  scaml_kernel_matrix_f64."""
    T, N1, D = X1.shape
    X1 = _check(X1, "X1")
    theta = _check(theta, "theta", (T, D + 2))
    shared = 0
    N2 = N1
    if X2 is not None:
        shared = 1 if X2.dim() == 2 else 0
        N2 = X2.shape[-2]
        X2 = _check(X2, "X2", (N2, D) if shared else (T, N2, D))
    K = torch.empty((T, N1, N2), dtype=torch.float64, device=X1.device)
    with torch.cuda.device(X1.device):
        rc = _lib.lib.scaml_kernel_matrix_f64(_ptr(X1), _ptr(X2), _ptr(theta), T, N1, N2, D, int(kind), shared,
                                              1 if add_noise else 0, _ptr(K), _stream_handle())
    _lib.check_rc(rc, "scaml_kernel_matrix_f64")
    return K


def potrf_batched(A: torch.Tensor, y: Optional[torch.Tensor] = None, n_points: Optional[torch.Tensor] = None,
                  jitter: Optional[torch.Tensor] = None, zero_upper: bool = True, retry: bool = True,
                  want_linv: bool = False) -> Dict[str, torch.Tensor]:
    """Jittered Cholesky of a stack of given SPD matrices A (T, N, N) (+ optional solve with y (T, N)).
    Returns dict(L, alpha, quad, logdet, info, jitter, Linv_diag).  scaml_potrf_batched_f64."""
    if A.dim() != 3 or A.shape[1] != A.shape[2]:
        raise ValueError("A must be (T, N, N)")
    T, N, _ = A.shape
    A = _check(A, "A")
    if y is not None:
        y = _check(y, "y", (T, N))
    if n_points is not None:
        n_points = _check(n_points, "n_points", (T,), torch.int32)
    if jitter is not None:
        jitter = _check(jitter, "jitter", (T,))
    dev = A.device
    with torch.cuda.device(dev):
        L = torch.zeros((T, N, N), dtype=torch.float64, device=dev) if n_points is not None else torch.empty((T, N, N), dtype=torch.float64, device=dev)
        alpha = torch.zeros((T, N), dtype=torch.float64, device=dev) if y is not None else None
        quad = torch.empty((T,), dtype=torch.float64, device=dev) if y is not None else None
        logdet = torch.empty((T,), dtype=torch.float64, device=dev)
        info = torch.empty((T,), dtype=torch.int32, device=dev)
        jit_used = torch.empty((T,), dtype=torch.float64, device=dev)
        linv = torch.empty((T, (N + 15) // 16, 16, 16), dtype=torch.float64, device=dev) if want_linv else None
        flags = _lib.FIT_STORE_L | (_lib.FIT_ZERO_UPPER if zero_upper else 0) | (0 if retry else _lib.FIT_NO_RETRY)
        rc = _lib.lib.scaml_potrf_batched_f64(_ptr(A), _ptr(y), _ptr(n_points), _ptr(jitter), T, N, _ptr(L), _ptr(alpha),
                                              _ptr(quad), _ptr(logdet), _ptr(info), _ptr(jit_used), _ptr(linv), flags,
                                              _stream_handle())
    _lib.check_rc(rc, "scaml_potrf_batched_f64")
    return dict(L=L, alpha=alpha, quad=quad, logdet=logdet, info=info, jitter=jit_used, Linv_diag=linv)


def source_posteriors(
    Xq: torch.Tensor,
    X: torch.Tensor,
    theta: torch.Tensor,
    kind: int,
    L: torch.Tensor,
    Linv_diag: torch.Tensor,
    alpha: torch.Tensor,
    y_mean: Optional[torch.Tensor] = None,
    y_std: Optional[torch.Tensor] = None,
    n_points: Optional[torch.Tensor] = None,
    want_var: bool = True,
    cov_first: int = 0,
) -> Dict[str, torch.Tensor]:
    """Posteriors of all source GPs at the shared query points Xq (M, D).

    Returns dict(mean (T, M), var (T, M) or None, cov (T, cov_first, M) or None): ``cov`` is the
    posterior covariance between the first ``cov_first`` query points and all of them (put the
    target's training points first to get Sigma_nn and Sigma_nq in one go).  Un-standardised with
    y_mean / y_std (T) when given.  Launches scaml_posterior_batched_f64 (+ scaml_posterior_cov_f64).
    """
    if X.dim() != 3 or Xq.dim() != 2:
        raise ValueError("X must be (T, N, D) and Xq (M, D)")
    T, N, D = X.shape
    M = Xq.shape[0]
    if Xq.shape[1] != D:
        raise ValueError("Xq and X disagree on D")
    X = _check(X, "X")
    Xq = _check(Xq, "Xq")
    theta = _check(theta, "theta", (T, D + 2))
    L = _check(L, "L", (T, N, N))
    Linv_diag = _check(Linv_diag, "Linv_diag", (T, (N + 15) // 16, 16, 16))
    alpha = _check(alpha, "alpha", (T, N))
    if y_mean is not None:
        y_mean = _check(y_mean, "y_mean", (T,))
    if y_std is not None:
        y_std = _check(y_std, "y_std", (T,))
    if n_points is not None:
        n_points = _check(n_points, "n_points", (T,), torch.int32)
    if not 0 <= cov_first <= M:
        raise ValueError("cov_first must be within [0, M]")
    dev = X.device
    with torch.cuda.device(dev):
        mu = torch.empty((T, M), dtype=torch.float64, device=dev)
        var = torch.empty((T, M), dtype=torch.float64, device=dev) if want_var else None
        V = torch.empty((T, N, M), dtype=torch.float64, device=dev) if cov_first > 0 else None
        rc = _lib.lib.scaml_posterior_batched_f64(
            _ptr(Xq), _ptr(X), _ptr(theta), _ptr(L), _ptr(Linv_diag), _ptr(alpha), _ptr(y_mean), _ptr(y_std),
            _ptr(n_points), T, N, M, D, int(kind), _ptr(mu), _ptr(var), _ptr(V), _stream_handle())
        _lib.check_rc(rc, "scaml_posterior_batched_f64")
        cov = None
        if cov_first > 0:
            cov = torch.empty((T, cov_first, M), dtype=torch.float64, device=dev)
            rc = _lib.lib.scaml_posterior_cov_f64(_ptr(Xq), _ptr(theta), _ptr(V), _ptr(y_std), T, N, M, cov_first, D,
                                                  int(kind), _ptr(cov), _stream_handle())
            _lib.check_rc(rc, "scaml_posterior_cov_f64")
    return dict(mean=mu, var=var, cov=cov)


def weighted_task_sum(values: torch.Tensor, weights: torch.Tensor, power: int = 1,
                      active: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum_t w_t^power * values[t] over the task axis (dim 0), optionally only over active tasks
    (bool mask, scamlgp/model.py:368-372).  Launches scaml_weighted_task_sum_f64."""
    T = values.shape[0]
    values = _check(values, "values")
    weights = _check(weights, "weights", (T,))
    if active is not None:
        if not active.is_cuda or active.shape != (T,):
            raise ValueError("active must be a (T,) GPU tensor")
        active = active.to(torch.uint8).contiguous()
    out = torch.empty(values.shape[1:], dtype=torch.float64, device=values.device)
    with torch.cuda.device(values.device):
        rc = _lib.lib.scaml_weighted_task_sum_f64(_ptr(values), _ptr(weights), _ptr(active), T, out.numel(), int(power),
                                                  _ptr(out), _stream_handle())
    _lib.check_rc(rc, "scaml_weighted_task_sum_f64")
    return out


def mll_backward(
    X: torch.Tensor,
    theta: torch.Tensor,
    kind: int,
    L: torch.Tensor,
    Linv_diag: torch.Tensor,
    alpha: torch.Tensor,
    n_points: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """d mll[t] / d theta[t] (T, D+2) for the constrained hyper-parameters, from the outputs of
    ``gp_fit_fused(..., want_linv=True)``.  Launches scaml_mll_backward_f64; the final sum over the
    per-tile partials and the 1 / (2 n_t) scaling are two tiny torch ops."""
    T, N, D = X.shape
    X = _check(X, "X")
    theta = _check(theta, "theta", (T, D + 2))
    L = _check(L, "L", (T, N, N))
    Linv_diag = _check(Linv_diag, "Linv_diag", (T, (N + 15) // 16, 16, 16))
    alpha = _check(alpha, "alpha", (T, N))
    if n_points is not None:
        n_points = _check(n_points, "n_points", (T,), torch.int32)
    nb = (N + 15) // 16
    nt = nb * (nb + 1) // 2
    dev = X.device
    with torch.cuda.device(dev):
        work = torch.empty((T * N * N,), dtype=torch.float64, device=dev)
        partials = torch.empty((T, nt, D + 2), dtype=torch.float64, device=dev)
        rc = _lib.lib.scaml_mll_backward_f64(_ptr(X), _ptr(theta), _ptr(L), _ptr(Linv_diag), _ptr(alpha), _ptr(n_points),
                                             T, N, D, int(kind), _ptr(work), _ptr(partials), _stream_handle())
    _lib.check_rc(rc, "scaml_mll_backward_f64")
    n = n_points.to(torch.float64) if n_points is not None else torch.full((T,), float(N), dtype=torch.float64, device=dev)
    return partials.sum(1) / (2.0 * n.clamp_min(1.0)).unsqueeze(-1)


def raise_if_not_psd(info: torch.Tensor) -> None:
    """Host-side check of the per-task status (one device->host sync)."""
    bad = torch.nonzero(info > 0).flatten()
    if bad.numel():
        raise NotPSDError(
            f"Matrix not positive definite after repeatedly adding jitter up to 1.0e-06 "
            f"(tasks {bad.tolist()[:8]}{'...' if bad.numel() > 8 else ''})."
        )
