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


def fit_max_n() -> int:
    """Largest N the single-launch fused fit takes (scaml_fit_max_n)."""
    return int(_lib.lib.scaml_fit_max_n())


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
    if N > _lib.lib.scaml_fit_max_n():
        if out is not None:
            raise ValueError("out= is not supported beyond scaml_fit_max_n()")
        if N <= _lib.lib.scaml_fit_blocked_max_n() and N % 16 == 0 and D <= _lib.lib.scaml_fit_blocked_max_d() and not _FORCE_COMPOSED_TWO_BLOCK:
            return _gp_fit_blocked(X, y, theta, kind, n_points, jitter, zero_upper, retry)
        return _gp_fit_two_block(X, y, theta, kind, n_points, jitter, store_L, retry)
    dev = X.device
    with torch.cuda.device(dev):
        if out is not None:
            L, alpha, quad, logdet, mll, info, jit_used = (out[k] for k in ("L", "alpha", "quad", "logdet", "mll", "info", "jitter"))
            linv = out.get("Linv_diag")
            if want_linv and linv is None:
                raise ValueError("out= has no Linv_diag buffer but want_linv=True")
            if not want_linv:
                linv = None
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


_FORCE_COMPOSED_TWO_BLOCK = False   # tests / A-B timing: route 256 < N <= 512 through _gp_fit_two_block


def _gp_fit_blocked(X, y, theta, kind, n_points, jitter, zero_upper, retry) -> Dict[str, torch.Tensor]:
    """Fused fit for scaml_fit_max_n() < N <= scaml_fit_blocked_max_n() (the N = 512 source tasks of BASELINE
    configs[4]): ``scaml_gp_fit_blocked_f64`` -- by shape, ONE launch with several CUs per task (csrc/gp_fit_coop.hip; stacks that
    leave CUs idle) or a 2 x 2 block factorisation enqueued as one sequence of the library's launches (csrc/gp_fit_blocked.hip);
    jitter ladder included, without a host synchronisation or a framework op in between.  L and Linv_diag are always produced."""
    T, N, D = X.shape
    dev = X.device
    with torch.cuda.device(dev):
        # ragged stacks: rows past n_t are never written but READ by the strip solve -> zeros
        L = (torch.zeros if n_points is not None else torch.empty)((T, N, N), dtype=torch.float64, device=dev)
        alpha = (torch.zeros if n_points is not None else torch.empty)((T, N), dtype=torch.float64, device=dev)
        quad, logdet, mll, jit_used = (torch.empty((T,), dtype=torch.float64, device=dev) for _ in range(4))
        info = torch.empty((T,), dtype=torch.int32, device=dev)
        linv = torch.empty((T, N // 16, 16, 16), dtype=torch.float64, device=dev)
        nbytes = int(_lib.lib.scaml_gp_fit_blocked_workspace_bytes(T, N))
        ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=dev)
        flags = _lib.FIT_STORE_L | (_lib.FIT_ZERO_UPPER if zero_upper else 0) | (0 if retry else _lib.FIT_NO_RETRY)
        rc = _lib.lib.scaml_gp_fit_blocked_f64(
            _ptr(X), _ptr(y), _ptr(theta), _ptr(n_points), _ptr(jitter), T, N, D, int(kind),
            _ptr(L), _ptr(alpha), _ptr(quad), _ptr(logdet), _ptr(mll), _ptr(info), _ptr(jit_used), _ptr(linv), flags,
            _ptr(ws), nbytes, _stream_handle())
        ws.record_stream(torch.cuda.current_stream(dev))
    _lib.check_rc(rc, "scaml_gp_fit_blocked_f64")
    return dict(L=L, alpha=alpha, quad=quad, logdet=logdet, mll=mll, info=info, jitter=jit_used, Linv_diag=linv)


_JITTER_LADDER = (1e-8, 1e-7, 1e-6)  # psd_safe_cholesky's escalation


def _gp_fit_two_block(X, y, theta, kind, n_points, jitter, store_L, retry) -> Dict[str, torch.Tensor]:
    """Fused fit for scaml_fit_max_n() < N <= 2 * scaml_fit_max_n() (the N = 512 source tasks of
    BASELINE config 5) as a 2 x 2 block factorisation composed from the library's own launches:

        [K11 K12]   [L11      ] [L11^T  V  ]      L11 = fused fit of block 1 (register-resident kernel)
        [K21 K22] = [V^T   L22] [      L22^T]     V   = L11^-1 K12          (scaml_posterior_batched_f64)
                                                  S   = K22 - V^T V          (scaml_posterior_cov_f64)
                                                  L22 = chol(S + noise I)    (scaml_potrf_batched_f64)

    The posterior of block 1 at the points of block 2 is exactly the conditional the Schur complement
    needs: alpha_2 = S^-1 (y_2 - K21 K11^-1 y_1) falls out of the POTRF's solve, and
    alpha_1 = alpha_1' - K11^-1 K12 alpha_2 = alpha_1' - L11^-T (V alpha_2) (scaml_solve_lt_batched_f64).
    quad and logdet add over the blocks.  The jitter ladder is driven from the host here (one
    status read per attempt; every attempt is single-shot in the kernels) so that, like
    psd_safe_cholesky, one jitter value applies to the whole matrix of a failing task only.
    Torch is used for slicing/concatenation only.
    """
    T, N, D = X.shape
    N1 = _lib.lib.scaml_fit_max_n()
    if N > 2 * N1:
        raise ValueError(f"N = {N} exceeds the supported {2 * N1} points per task")
    N2 = N - N1
    dev = X.device
    X1, X2 = X[:, :N1].contiguous(), X[:, N1:].contiguous()
    y1, y2 = y[:, :N1].contiguous(), y[:, N1:].contiguous()
    n1 = n2 = None
    if n_points is not None:
        n1 = n_points.clamp(max=N1).to(torch.int32)
        n2 = (n_points - N1).clamp(min=0).to(torch.int32)
    noise = theta[:, D + 1].contiguous()
    jit = torch.zeros((T,), dtype=torch.float64, device=dev)
    base = jitter if jitter is not None else torch.zeros((T,), dtype=torch.float64, device=dev)
    for step in range(len(_JITTER_LADDER) + 1):
        f1 = gp_fit_fused(X1, y1, theta, kind, n_points=n1, jitter=base + jit, retry=False, want_linv=True)
        p12 = source_posteriors(X2, X1, theta, kind, f1["L"], f1["Linv_diag"], f1["alpha"], n_points=n1,
                                want_var=False, cov_first=N2, keep_V=True)
        f2 = potrf_batched(p12["cov"], y2 - p12["mean"], n_points=n2, jitter=noise + base + jit, retry=False,
                           want_linv=True)
        info = torch.where(f1["info"] > 0, f1["info"], torch.where(f2["info"] > 0, f2["info"] + N1, f2["info"]))
        failed = info > 0
        if not retry or step == len(_JITTER_LADDER) or not bool(failed.any()):
            break
        jit = torch.where(failed, torch.full_like(jit, _JITTER_LADDER[step]), jit)
    # K11^-1 K12 alpha_2 = L11^-T (V alpha_2) with the V = L11^-1 K12 the Schur complement was formed from: a
    # mat-vec and the backward half of a Cholesky solve (alpha_2 is zero past n2, so V's padded columns drop out)
    u = cho_solve(f1["L"], f1["Linv_diag"], torch.bmm(p12["V"], f2["alpha"].unsqueeze(-1)), n_points=n1, backward_only=True).squeeze(-1)
    alpha = torch.cat([f1["alpha"] - u, f2["alpha"]], 1)
    quad = f1["quad"] + f2["quad"]
    logdet = f1["logdet"] + f2["logdet"]
    n = n_points.to(torch.float64) if n_points is not None else torch.full((T,), float(N), dtype=torch.float64, device=dev)
    mll = -(quad + logdet + n * 1.8378770664093453) / (2.0 * n.clamp_min(1.0))
    L = None
    if store_L:
        L = torch.zeros((T, N, N), dtype=torch.float64, device=dev)
        L[:, :N1, :N1] = f1["L"]
        Vt = p12["V"].transpose(1, 2)
        if n2 is not None:
            Vt = Vt * (torch.arange(N2, device=dev)[None, :, None] < n2[:, None, None])
        L[:, N1:, :N1] = Vt
        L[:, N1:, N1:] = f2["L"]
    linv = torch.cat([f1["Linv_diag"], f2["Linv_diag"]], 1)
    return dict(L=L, alpha=alpha, quad=quad, logdet=logdet, mll=mll, info=info.to(torch.int32), jitter=jit, Linv_diag=linv)


def linv_batched(L: torch.Tensor, Linv_diag: torch.Tensor, n_points: Optional[torch.Tensor] = None, lower_only: bool = False) -> torch.Tensor:
    """L^-1 (T, N, N) of the factors of a fused fit (zero above the diagonal).  scaml_linv_batched_f64; ``lower_only``:
    scaml_linv_batched_lower_f64 -- the block rows above each strip's diagonal block are left unwritten (no kernel of the library
    reads them; the result is then NOT a dense matrix to multiply with)."""
    T, N, _ = L.shape
    L = _check(L, "L", (T, N, N))
    Linv_diag = _check(Linv_diag, "Linv_diag", (T, (N + 15) // 16, 16, 16))
    if n_points is not None:
        n_points = _check(n_points, "n_points", (T,), torch.int32)
    out = torch.empty_like(L)
    with torch.cuda.device(L.device):
        fn = _lib.lib.scaml_linv_batched_lower_f64 if lower_only else _lib.lib.scaml_linv_batched_f64
        rc = fn(_ptr(L), _ptr(Linv_diag), _ptr(n_points), T, N, _ptr(out), _stream_handle())
    _lib.check_rc(rc, "scaml_linv_batched_f64")
    return out


def cho_solve(L: torch.Tensor, Linv_diag: torch.Tensor, B: torch.Tensor, n_points: Optional[torch.Tensor] = None,
              backward_only: bool = False) -> torch.Tensor:
    """(L L^T)^-1 B for B (T, N, R) with the factors of a fused fit (scaml_cho_solve_batched_f64), or, with
    backward_only, L^-T B (scaml_solve_lt_batched_f64)."""
    T, N, _ = L.shape
    R = B.shape[-1]
    L = _check(L, "L", (T, N, N))
    Linv_diag = _check(Linv_diag, "Linv_diag", (T, (N + 15) // 16, 16, 16))
    B = _check(B, "B", (T, N, R))
    if n_points is not None:
        n_points = _check(n_points, "n_points", (T,), torch.int32)
    out = torch.empty_like(B)
    with torch.cuda.device(L.device):
        fn = _lib.lib.scaml_solve_lt_batched_f64 if backward_only else _lib.lib.scaml_cho_solve_batched_f64
        rc = fn(_ptr(L), _ptr(Linv_diag), _ptr(B), _ptr(n_points), T, N, R, _ptr(out), _stream_handle())
    _lib.check_rc(rc, "scaml_solve_lt_batched_f64" if backward_only else "scaml_cho_solve_batched_f64")
    return out


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
    keep_V: bool = False,
    mean_only: bool = False,
    Linv: Optional[torch.Tensor] = None,
    VA: Optional[torch.Tensor] = None,
) -> Dict[str, torch.Tensor]:
    """Posteriors of all source GPs at the shared query points Xq (M, D), or at per-task query sets
    Xq (T, M, D).  ``mean_only`` skips the triangular solve (mu = m + s K_* alpha only; L and
    Linv_diag may be None); ``keep_V`` also returns V = L^-1 K_*^T (T, N, M).  With ``Linv`` (T, N, N) from
    ``linv_batched`` the posteriors come from the explicit inverse factor (scaml_posterior_linv_f64: a
    triangular matrix product instead of a substitution -- the fast path when one fit serves many queries).
    ``VA`` (T, N, cov_first): the V of the leading ``cov_first`` query points from an earlier call (``keep_V=True`` on those points
    alone) -- the fused covariance pass then skips recomputing it (the leading points are a model's fixed training inputs).

    Returns dict(mean (T, M), var (T, M) or None, cov (T, cov_first, M) or None): ``cov`` is the
    posterior covariance between the first ``cov_first`` query points and all of them (put the
    target's training points first to get Sigma_nn and Sigma_nq in one go).  Un-standardised with
    y_mean / y_std (T) when given.  Launches scaml_posterior_batched_f64 (+ scaml_posterior_cov_f64).
    """
    if X.dim() != 3 or Xq.dim() not in (2, 3):
        raise ValueError("X must be (T, N, D) and Xq (M, D) or (T, M, D)")
    T, N, D = X.shape
    M = Xq.shape[-2]
    if Xq.shape[-1] != D:
        raise ValueError("Xq and X disagree on D")
    flags = 0
    X = _check(X, "X")
    if Xq.dim() == 3:
        Xq = _check(Xq, "Xq", (T, M, D))
        flags |= _lib.POST_XQ_PER_TASK
    else:
        Xq = _check(Xq, "Xq")
    theta = _check(theta, "theta", (T, D + 2))
    if mean_only:
        flags |= _lib.POST_MEAN_ONLY
        want_var, cov_first, keep_V, L, Linv_diag = False, 0, False, None, None
    elif Linv is None:
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
        fused_cov = (Linv is not None and not mean_only and not keep_V and 0 < cov_first <= min(96, N, M))
        V = torch.empty((T, N, M), dtype=torch.float64, device=dev) if ((cov_first > 0 and not fused_cov) or keep_V) else None
        cov = None
        if fused_cov:
            # the covariance block comes out of the posterior pass itself: V of the leading cov_first points first
            # (T, N, cov_first -- small), then one pass over all M points gives mean, var and cov; V (T, N, M) is never stored
            Linv = _check(Linv, "Linv", (T, N, N))
            if VA is not None:
                VA = _check(VA, "VA", (T, N, cov_first))
            else:
                Xa = Xq[:, :cov_first].contiguous() if Xq.dim() == 3 else Xq[:cov_first].contiguous()
                VA = torch.empty((T, N, cov_first), dtype=torch.float64, device=dev)
                mu_a = torch.empty((T, cov_first), dtype=torch.float64, device=dev)
                rc = _lib.lib.scaml_posterior_linv_f64(
                    _ptr(Xa), _ptr(X), _ptr(theta), _ptr(Linv), _ptr(alpha), _ptr(y_mean), _ptr(y_std), _ptr(n_points),
                    T, N, cov_first, D, int(kind), _ptr(mu_a), None, _ptr(VA), flags, _stream_handle())
                _lib.check_rc(rc, "scaml_posterior_linv_f64")
            cov = torch.empty((T, cov_first, M), dtype=torch.float64, device=dev)
            if var is None:
                var = torch.empty((T, M), dtype=torch.float64, device=dev)
            rc = _lib.lib.scaml_posterior_linv_cov_f64(
                _ptr(Xq), _ptr(X), _ptr(theta), _ptr(Linv), _ptr(alpha), _ptr(y_mean), _ptr(y_std), _ptr(n_points), _ptr(VA),
                T, N, M, cov_first, D, int(kind), _ptr(mu), _ptr(var), _ptr(cov), flags, _stream_handle())
            _lib.check_rc(rc, "scaml_posterior_linv_cov_f64")
            return dict(mean=mu, var=var if want_var else None, cov=cov, V=None)
        if Linv is not None and not mean_only:
            Linv = _check(Linv, "Linv", (T, N, N))
            rc = _lib.lib.scaml_posterior_linv_f64(
                _ptr(Xq), _ptr(X), _ptr(theta), _ptr(Linv), _ptr(alpha), _ptr(y_mean), _ptr(y_std), _ptr(n_points),
                T, N, M, D, int(kind), _ptr(mu), _ptr(var), _ptr(V), flags, _stream_handle())
            _lib.check_rc(rc, "scaml_posterior_linv_f64")
        else:
            rc = _lib.lib.scaml_posterior_batched_f64(
            _ptr(Xq), _ptr(X), _ptr(theta), _ptr(L), _ptr(Linv_diag), _ptr(alpha), _ptr(y_mean), _ptr(y_std),
                _ptr(n_points), T, N, M, D, int(kind), _ptr(mu), _ptr(var), _ptr(V), flags, _stream_handle())
            _lib.check_rc(rc, "scaml_posterior_batched_f64")
        if cov_first > 0:
            cov = torch.empty((T, cov_first, M), dtype=torch.float64, device=dev)
            rc = _lib.lib.scaml_posterior_cov_f64(_ptr(Xq), _ptr(theta), _ptr(V), _ptr(y_std), T, N, M, cov_first, D,
                                                  int(kind), _ptr(cov), flags & _lib.POST_XQ_PER_TASK, _stream_handle())
            _lib.check_rc(rc, "scaml_posterior_cov_f64")
    return dict(mean=mu, var=var, cov=cov, V=V if keep_V else None)


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


def weighted_prior_reduce(mu: Optional[torch.Tensor], cov: Optional[torch.Tensor], weights: torch.Tensor,
                          active: Optional[torch.Tensor] = None):
    """Target prior of scamlgp/model.py:108-135 from stacked source posteriors: (sum_t w_t mu[t], sum_t w_t^2 cov[t])
    over the active tasks.  mu (T, M), cov (T, Ma, M); either may be None.  scaml_weighted_prior_reduce_f64."""
    T = weights.shape[0]
    weights = _check(weights, "weights", (T,))
    M = mu.shape[1] if mu is not None else cov.shape[2]
    Ma = cov.shape[1] if cov is not None else 0
    if mu is not None:
        mu = _check(mu, "mu", (T, M))
    if cov is not None:
        cov = _check(cov, "cov", (T, Ma, M))
    if active is not None:
        active = active.to(torch.uint8).contiguous()
    dev = weights.device
    mu_s = torch.empty((M,), dtype=torch.float64, device=dev) if mu is not None else None
    cov_s = torch.empty((Ma, M), dtype=torch.float64, device=dev) if cov is not None else None
    with torch.cuda.device(dev):
        rc = _lib.lib.scaml_weighted_prior_reduce_f64(_ptr(mu), _ptr(cov), _ptr(weights), _ptr(active), T, M, Ma, _ptr(mu_s),
                                                      _ptr(cov_s), _stream_handle())
    _lib.check_rc(rc, "scaml_weighted_prior_reduce_f64")
    return mu_s, cov_s


def target_posterior(cov_s: torch.Tensor, mean_s: torch.Tensor, var_s: torch.Tensor, Xall: torch.Tensor, theta: torch.Tensor,
                     train_targets: torch.Tensor, m_all: float, s_all: float, kind: int, observation_noise: bool = False,
                     factor: Optional[Dict[str, torch.Tensor]] = None):
    """Posterior mean / variance (original units) of the ScaML-GP target GP at the M query points behind the n training
    points in ``Xall`` (n + M, D), from the weighted source sums at the same points: cov_s (n, n + M), mean_s, var_s
    (n + M).  scaml_target_assemble_f64 -> scaml_potrf_batched_f64 (T = 1, jitter ladder) -> scaml_cho_solve_batched_f64
    -> scaml_target_finish_f64: four launches, no host synchronisation.  A factorisation that fails even with jitter shows as NaN
    in mu / var (the status stays on the device).  Returns (mu (M,), var (M,), info (1,), jitter (1,))."""
    f = target_posterior_full(cov_s, mean_s, var_s, Xall, theta, train_targets, m_all, s_all, kind, observation_noise, factor)
    return f["mu"], f["var"], f["info"], f["jitter"]


def target_posterior_full(cov_s: torch.Tensor, mean_s: torch.Tensor, var_s: torch.Tensor, Xall: torch.Tensor, theta: torch.Tensor,
                          train_targets: torch.Tensor, m_all: float, s_all: float, kind: int, observation_noise: bool = False,
                          factor: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """``target_posterior`` with its intermediates: dict(mu, var, info, jitter, alpha (n,) = Knn^-1 resid, Z (n, M) = Knn^-1 Knq,
    factor) -- alpha and Z are what the input gradient of the posterior contracts against (``target_posterior_grad``).  ``factor``:
    the ``factor`` entry of an earlier call with the SAME training block (cov_s[:, :n], mean_s[:n], theta, targets): the jittered
    Cholesky of Knn is then reused instead of recomputed -- it does not depend on the query points (an acquisition optimisation
    evaluates hundreds of query batches against one factor)."""
    n = int(train_targets.shape[0])
    W, D = Xall.shape
    M = W - n
    if not 1 <= n <= _lib.lib.scaml_fit_max_n():
        raise ValueError(f"target_posterior takes 1 <= n <= {_lib.lib.scaml_fit_max_n()} training points")
    cov_s = _check(cov_s, "cov_s", (n, W))
    mean_s = _check(mean_s, "mean_s", (W,))
    var_s = _check(var_s, "var_s", (W,))
    Xall = _check(Xall, "Xall")
    theta = _check(theta, "theta", (D + 2,))
    train_targets = _check(train_targets, "train_targets", (n,))
    dev = Xall.device
    f64 = dict(dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        Knn, resid = torch.empty((1, n, n), **f64), torch.empty((1, n), **f64)
        Knq, mean_q, var_q = torch.empty((1, n, M), **f64), torch.empty((M,), **f64), torch.empty((M,), **f64)
        rc = _lib.lib.scaml_target_assemble_f64(_ptr(cov_s), _ptr(mean_s), _ptr(var_s), _ptr(Xall), _ptr(theta), _ptr(train_targets),
                                                float(m_all), float(s_all), n, M, D, int(kind), _ptr(Knn), _ptr(resid), _ptr(Knq),
                                                _ptr(mean_q), _ptr(var_q), _stream_handle())
        _lib.check_rc(rc, "scaml_target_assemble_f64")
        f = factor if factor is not None else potrf_batched(Knn, resid, want_linv=True)
        mu, var = torch.empty((M,), **f64), torch.empty((M,), **f64)
        Z = None
        if M > 0:
            Z = cho_solve(f["L"], f["Linv_diag"], Knq)
            rc = _lib.lib.scaml_target_finish_f64(_ptr(Knq), _ptr(Z), _ptr(f["alpha"]), _ptr(mean_q), _ptr(var_q), float(m_all), float(s_all),
                                                  float(theta[D + 1]) if observation_noise else 0.0, _ptr(f["info"]), n, M, _ptr(mu), _ptr(var),
                                                  _stream_handle())
            _lib.check_rc(rc, "scaml_target_finish_f64")
    return dict(mu=mu, var=var, info=f["info"], jitter=f["jitter"], alpha=f["alpha"][0], Z=None if Z is None else Z[0], factor=f)


def source_posteriors_grad(Xq: torch.Tensor, Xa: Optional[torch.Tensor], X: torch.Tensor, theta: torch.Tensor, kind: int, Linv: torch.Tensor,
                           alpha: torch.Tensor, y_mean: Optional[torch.Tensor] = None, y_std: Optional[torch.Tensor] = None,
                           n_points: Optional[torch.Tensor] = None, VA: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Source posteriors at Xq (Mq, D) WITH their input gradients, one launch of scaml_posterior_linv_grad_f64: dict(mu (T, Mq, 16),
    var (T, Mq, 16), cov (T, Ma, Mq * 16) or None); column 0 of each 16-block is the value, columns 1 .. D the derivative w.r.t.
    x_0 .. x_{D-1}.  Xa (Ma, D), VA (T, N, Ma): the leading points of the covariance block (the target's training inputs) and their
    V = L^-1 K(X, Xa) from ``source_posteriors(..., keep_V=True)``."""
    T, N, D = X.shape
    Mq = Xq.shape[0]
    X = _check(X, "X")
    Xq = _check(Xq, "Xq", (Mq, D))
    theta = _check(theta, "theta", (T, D + 2))
    Linv = _check(Linv, "Linv", (T, N, N))
    alpha = _check(alpha, "alpha", (T, N))
    Ma = 0 if VA is None else int(VA.shape[-1])
    if Ma:
        VA = _check(VA, "VA", (T, N, Ma))
        Xa = _check(Xa, "Xa", (Ma, D))
    if y_mean is not None:
        y_mean = _check(y_mean, "y_mean", (T,))
    if y_std is not None:
        y_std = _check(y_std, "y_std", (T,))
    if n_points is not None:
        n_points = _check(n_points, "n_points", (T,), torch.int32)
    dev = X.device
    with torch.cuda.device(dev):
        mu = torch.empty((T, Mq, 16), dtype=torch.float64, device=dev)
        var = torch.empty((T, Mq, 16), dtype=torch.float64, device=dev)
        cov = torch.empty((T, Ma, Mq * 16), dtype=torch.float64, device=dev) if Ma else None
        rc = _lib.lib.scaml_posterior_linv_grad_f64(_ptr(Xq), _ptr(Xa) if Ma else None, _ptr(X), _ptr(theta), _ptr(Linv), _ptr(alpha), _ptr(y_mean),
                                                    _ptr(y_std), _ptr(n_points), _ptr(VA) if Ma else None, T, N, Mq, Ma, D, int(kind), _ptr(mu),
                                                    _ptr(var), _ptr(cov), 0, _stream_handle())
    _lib.check_rc(rc, "scaml_posterior_linv_grad_f64")
    return dict(mu=mu, var=var, cov=cov)


def target_posterior_grad(cov_g: Optional[torch.Tensor], mu_g: torch.Tensor, var_g: torch.Tensor, Xt: torch.Tensor, Xq: torch.Tensor,
                          theta: torch.Tensor, alpha: Optional[torch.Tensor], Z: Optional[torch.Tensor], s_all: float,
                          info: Optional[torch.Tensor], kind: int):
    """d mu* / d x and d var* / d x (Mq, D) of the target posterior from the weighted task sums of ``source_posteriors_grad``
    (cov_g (n, Mq * 16), mu_g / var_g (Mq * 16)) and the value path's alpha (n,), Z (n, Mq).  scaml_target_posterior_grad_f64."""
    Mq, D = Xq.shape
    n = 0 if cov_g is None else int(cov_g.shape[0])
    mu_g = _check(mu_g.reshape(-1), "mu_g", (Mq * 16,))
    var_g = _check(var_g.reshape(-1), "var_g", (Mq * 16,))
    Xq = _check(Xq, "Xq")
    theta = _check(theta, "theta", (D + 2,))
    if n:
        cov_g = _check(cov_g, "cov_g", (n, Mq * 16))
        Xt = _check(Xt, "Xt", (n, D))
        alpha = _check(alpha, "alpha", (n,))
        Z = _check(Z, "Z", (n, Mq))
    dev = Xq.device
    with torch.cuda.device(dev):
        dmu = torch.empty((Mq, D), dtype=torch.float64, device=dev)
        dvar = torch.empty((Mq, D), dtype=torch.float64, device=dev)
        rc = _lib.lib.scaml_target_posterior_grad_f64(_ptr(cov_g) if n else None, _ptr(mu_g), _ptr(var_g), _ptr(Xt) if n else None, _ptr(Xq),
                                                      _ptr(theta), _ptr(alpha) if n else None, _ptr(Z) if n else None, float(s_all), _ptr(info), n, Mq, D,
                                                      int(kind), _ptr(dmu), _ptr(dvar), _stream_handle())
    _lib.check_rc(rc, "scaml_target_posterior_grad_f64")
    return dmu, dvar


def mll_backward_workspace(T: int, N: int, D: int, device) -> Dict[str, torch.Tensor]:
    """Reusable buffers of ``mll_backward`` for a (T, N, D) stack: the explicit inverse factors (T N^2 doubles) and
    the per-tile partial sums.  An optimiser loop allocates them once (scaml_mll_backward_workspace_doubles)."""
    nb = (N + 15) // 16
    nt = nb * (nb + 1) // 2
    return dict(work=torch.empty((T * N * N,), dtype=torch.float64, device=device),
                partials=torch.empty((T, nt, D + 2), dtype=torch.float64, device=device), shape=(T, N, D))


def mll_backward(
    X: torch.Tensor,
    theta: torch.Tensor,
    kind: int,
    L: torch.Tensor,
    Linv_diag: torch.Tensor,
    alpha: torch.Tensor,
    n_points: Optional[torch.Tensor] = None,
    workspace: Optional[Dict[str, torch.Tensor]] = None,
) -> torch.Tensor:
    """d mll[t] / d theta[t] (T, D+2) for the constrained hyper-parameters, from the outputs of
    ``gp_fit_fused(..., want_linv=True)``.  Launches scaml_mll_backward_f64; the final sum over the
    per-tile partials and the 1 / (2 n_t) scaling are two tiny torch ops.  ``workspace`` (from
    ``mll_backward_workspace``) is reused across calls instead of allocating T N^2 doubles each time."""
    T, N, D = X.shape
    X = _check(X, "X")
    theta = _check(theta, "theta", (T, D + 2))
    L = _check(L, "L", (T, N, N))
    Linv_diag = _check(Linv_diag, "Linv_diag", (T, (N + 15) // 16, 16, 16))
    alpha = _check(alpha, "alpha", (T, N))
    if n_points is not None:
        n_points = _check(n_points, "n_points", (T,), torch.int32)
    dev = X.device
    if workspace is None:
        workspace = mll_backward_workspace(T, N, D, dev)
    elif workspace["shape"] != (T, N, D) or workspace["work"].device != dev:
        raise ValueError("workspace does not match this call's (T, N, D) / device")
    work, partials = workspace["work"], workspace["partials"]
    with torch.cuda.device(dev):
        rc = _lib.lib.scaml_mll_backward_f64(_ptr(X), _ptr(theta), _ptr(L), _ptr(Linv_diag), _ptr(alpha), _ptr(n_points),
                                             T, N, D, int(kind), _ptr(work), _ptr(partials), _stream_handle())
    _lib.check_rc(rc, "scaml_mll_backward_f64")
    n = n_points.to(torch.float64) if n_points is not None else torch.full((T,), float(N), dtype=torch.float64, device=dev)
    return partials.sum(1) / (2.0 * n.clamp_min(1.0)).unsqueeze(-1)


class FusedMLL(torch.autograd.Function):
    """mll (T,) = FusedMLL.apply(X, y, theta, kind, n_points): the marginal log-likelihood of every task of the
    stack as a differentiable torch op.  forward = one launch of scaml_gp_fit_fused_f64 (K, jittered Cholesky,
    alpha, MLL), backward = scaml_mll_backward_f64 (analytic d mll / d theta; no gradient flows to X or y).
    Replaces the autograd pass through kernel -> Cholesky -> solves that botorch's fit_gpytorch_mll runs per
    L-BFGS-B iteration (scamlgp/utils.py:175, 190).  Tasks whose factorisation fails even with jitter give
    mll = NaN and a zero gradient; ``FusedMLL.last`` keeps the last forward's outputs (info, jitter, L, alpha)."""

    last: Optional[Dict[str, torch.Tensor]] = None
    workspace: Optional[Dict[str, torch.Tensor]] = None

    @staticmethod
    def forward(ctx, X, y, theta, kind, n_points=None, out=None):
        # (N > scaml_fit_max_n() goes through the two-block composite, which allocates its own outputs)
        if out is not None and X.shape[1] > _lib.lib.scaml_fit_max_n():
            out = None
        fit = gp_fit_fused(X.detach(), y.detach(), theta.detach(), kind, n_points=n_points, want_linv=True, zero_upper=False,
                           out=out)
        ctx.kind, ctx.n_points = int(kind), n_points
        ctx.save_for_backward(X.detach(), theta.detach(), fit["L"], fit["Linv_diag"], fit["alpha"], fit["info"])
        FusedMLL.last = fit
        return fit["mll"].clone()

    @staticmethod
    def backward(ctx, grad_mll):
        X, theta, L, Linv_diag, alpha, info = ctx.saved_tensors
        ws = FusedMLL.workspace
        if ws is not None and (ws["shape"] != tuple(X.shape) or ws["work"].device != X.device):
            ws = None
        if ws is None:
            ws = FusedMLL.workspace = mll_backward_workspace(*X.shape, X.device)
        g = mll_backward(X, theta, ctx.kind, L, Linv_diag, alpha, n_points=ctx.n_points, workspace=ws)
        g = torch.where((info > 0).unsqueeze(-1), torch.zeros_like(g), g)
        return None, None, g * grad_mll.unsqueeze(-1), None, None, None


class GaussianLogProb(torch.autograd.Function):
    """log N(resid | 0, K) for ONE given covariance matrix K (n, n), n <= scaml_fit_max_n(), as a differentiable torch op on the
    library's launches: forward = scaml_potrf_batched_f64 (T = 1; psd_safe_cholesky's jitter ladder in-kernel, alpha, quad, logdet),
    backward = scaml_cho_solve_batched_f64 with the identity as right-hand side (K^-1), d/dK = (alpha alpha^T - K^-1) / 2,
    d/dresid = -alpha.  This is the MultivariateNormal.log_prob of the TARGET GP's marginal likelihood
    (scamlgp/utils.py:171-177 on the ScaMLGP model of scamlgp/model.py:359-384), whose K is not a kernel matrix of points but
    weighted source covariances + target kernel + noise.  Nothing synchronises with the host: a factorisation that fails even
    with jitter gives NaN (value and gradient), and the status of the last call is in ``GaussianLogProb.last_info``."""

    last_info: Optional[torch.Tensor] = None

    @staticmethod
    def forward(ctx, K, resid):
        n = K.shape[-1]
        f = potrf_batched(K.detach().reshape(1, n, n).contiguous(), resid.detach().reshape(1, n).contiguous(), want_linv=True)
        GaussianLogProb.last_info = f["info"]
        ctx.save_for_backward(f["L"], f["Linv_diag"], f["alpha"], f["info"])
        nan = torch.where(f["info"] > 0, float("nan"), 0.0).to(torch.float64)
        return (-0.5 * (f["quad"] + f["logdet"] + n * 1.8378770664093453) + nan).reshape(())

    @staticmethod
    def backward(ctx, g):
        L, W, alpha, info = ctx.saved_tensors
        n = L.shape[-1]
        eye = torch.eye(n, dtype=torch.float64, device=L.device).unsqueeze(0)
        Kinv = cho_solve(L, W, eye)[0]
        a = alpha[0]
        nan = torch.where(info > 0, float("nan"), 0.0).to(torch.float64)
        gK = 0.5 * (torch.outer(a, a) - Kinv) + nan
        return g * gK, g * (-a + nan)


def gaussian_log_prob(K: torch.Tensor, resid: torch.Tensor) -> torch.Tensor:
    """Functional form of ``GaussianLogProb``."""
    return GaussianLogProb.apply(K, resid)


def fused_mll(X: torch.Tensor, y: torch.Tensor, theta: torch.Tensor, kind: int,
              n_points: Optional[torch.Tensor] = None, out: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """Functional form of ``FusedMLL``.  ``out`` (the dict ``FusedMLL.last`` of an earlier call with the same shapes)
    lets an optimiser loop reuse the factor buffers: the caller then runs backward before the next forward."""
    return FusedMLL.apply(X, y, theta, kind, n_points, out)



# ---- (8) target GP: objective + gradient / whole refit in one launch -------------------------------------------------------
class TargetFitProblem:
    """Device-resident inputs of ``scaml_target_mll_f64`` / ``scaml_target_fit_f64`` for ONE ScaMLGP training set: the source
    terms cached at construction (scamlgp/model.py:279-289) in the kernel's layouts -- means (T, n), covariances packed lower
    (T, n (n + 1) / 2) --, the target inputs / standardised targets, the standardiser and the constraint / prior block.
    Built once per model; every objective evaluation and the refit reuse it."""

    def __init__(self, source_means: torch.Tensor, source_covs: torch.Tensor, train_X: torch.Tensor, train_targets: torch.Tensor,
                 m_all: float, s_all: float, spec, weights_prior, weights_lower_bound: float, kind: int):
        import ctypes

        from . import hyper

        n, T = source_means.shape
        D = train_X.shape[-1]
        self.n, self.T, self.D, self.kind = int(n), int(T), int(D), int(kind)
        self.P = D + 2 + T
        self.device = train_X.device
        self.means_t = _check(source_means.transpose(0, 1), "source_means^T", (T, n))
        ia, ib = torch.tril_indices(n, n, device=self.device)
        self.covs_p = source_covs[ia, ib, :].transpose(0, 1).contiguous()          # (T, n (n + 1) / 2): plumbing, once per model
        self.X = _check(train_X, "train_X", (n, D))
        self.y = _check(train_targets, "train_targets", (n,))
        self.m_all, self.s_all = float(m_all), float(s_all)

        def prior(pr):
            if pr is None:
                return [0.0, 0.0, 0.0]
            if isinstance(pr, hyper.GammaPrior):
                return [1.0, pr.concentration, pr.rate]
            if isinstance(pr, hyper.LogNormalPrior):
                return [2.0, pr.loc, pr.scale]
            raise TypeError(f"prior {type(pr).__name__} is not supported by the target-fit kernel")

        vals = [spec.ls_constraint.lower, spec.ls_constraint.upper, spec.os_constraint.lower, spec.os_constraint.upper,
                spec.noise_constraint.lower, spec.noise_constraint.upper, *prior(spec.ls_prior), *prior(spec.os_prior),
                *prior(spec.noise_prior), *prior(weights_prior), float(weights_lower_bound)]
        self.spec_host = (ctypes.c_double * 19)(*[float(v) for v in vals])

    @staticmethod
    def supported(n: int, T: int, D: int) -> bool:
        return D <= _lib.lib.scaml_target_fit_max_d() and 1 <= n <= _lib.lib.scaml_target_fit_max_n(T, D)


def target_mll(prob: TargetFitProblem, z: torch.Tensor) -> Dict[str, torch.Tensor]:
    """mll(z_b) and d mll / d z_b for the rows of z (B, D + 2 + T) = [raw lengthscales, raw outputscale, raw noise, weights]:
    the ScaMLGP training objective of scamlgp/model.py:360-363 + utils.py:171-177 (priors included, divided by n) with its
    analytic gradient, ONE launch of scaml_target_mll_f64.  Returns dict(value (B,), grad (B, P), info (B,), jitter (B,));
    a matrix that is not positive definite even with jitter 1e-6 gives value NaN, info > 0 and a zero gradient."""
    z = _check(z, "z", (z.shape[0], prob.P))
    B = z.shape[0]
    dev = prob.device
    with torch.cuda.device(dev):
        value = torch.empty((B,), dtype=torch.float64, device=dev)
        grad = torch.empty((B, prob.P), dtype=torch.float64, device=dev)
        info = torch.empty((B,), dtype=torch.int32, device=dev)
        jit = torch.empty((B,), dtype=torch.float64, device=dev)
        rc = _lib.lib.scaml_target_mll_f64(_ptr(prob.means_t), _ptr(prob.covs_p), _ptr(prob.X), _ptr(prob.y), prob.m_all, prob.s_all,
                                           prob.spec_host, _ptr(z), B, prob.n, prob.T, prob.D, prob.kind, _ptr(value), _ptr(grad),
                                           _ptr(info), _ptr(jit), _stream_handle())
    _lib.check_rc(rc, "scaml_target_mll_f64")
    return dict(value=value, grad=grad, info=info, jitter=jit)


def target_fit(prob: TargetFitProblem, z0: torch.Tensor, max_iter: int = 200, history: int = 10, gtol: float = 1e-5,
               ftol: float = 2.2e-9) -> Dict[str, torch.Tensor]:
    """Maximise mll from every row of z0 (B, P) on the device: ONE launch of scaml_target_fit_f64 runs all B L-BFGS
    optimisations (warm start + restarts of scamlgp/utils.py:184-199) to convergence -- no host round trip per evaluation.
    Returns dict(z (B, P) optima, value (B,) mll there, info, jitter, stats (B, 4) = [iterations, evaluations, status, 0])."""
    z = _check(z0, "z0", (z0.shape[0], prob.P)).clone()
    B = z.shape[0]
    dev = prob.device
    with torch.cuda.device(dev):
        value = torch.empty((B,), dtype=torch.float64, device=dev)
        info = torch.empty((B,), dtype=torch.int32, device=dev)
        jit = torch.empty((B,), dtype=torch.float64, device=dev)
        stats = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        nws = int(_lib.lib.scaml_target_fit_workspace_doubles(B, prob.T, prob.D, history))
        ws = torch.empty((max(nws, 1),), dtype=torch.float64, device=dev)
        rc = _lib.lib.scaml_target_fit_f64(_ptr(prob.means_t), _ptr(prob.covs_p), _ptr(prob.X), _ptr(prob.y), prob.m_all, prob.s_all,
                                           prob.spec_host, _ptr(z), B, prob.n, prob.T, prob.D, prob.kind, int(max_iter), int(history),
                                           float(gtol), float(ftol), _ptr(value), _ptr(info), _ptr(jit), _ptr(stats), _ptr(ws), nws,
                                           _stream_handle())
        ws.record_stream(torch.cuda.current_stream(dev))
    _lib.check_rc(rc, "scaml_target_fit_f64")
    return dict(z=z, value=value, info=info, jitter=jit, stats=stats)


def raise_if_not_psd(info: torch.Tensor) -> None:
    """Host-side check of the per-task status (one device->host sync)."""
    if bool((info < 0).any()):
        # only the several-CUs-per-task fit reports this (csrc/gp_fit_coop.hip): its workgroups wait for each other inside one launch, with
        # bounded polls; a task given up after ~4 s means its workgroups were not resident together -- another launch (a second process
        # or stream on this GPU) held the CUs while waiting itself.  SCAML_BLOCKED_FIT_PATH=1 keeps to the sequence of launches.
        raise RuntimeError("scaml_gp_fit_blocked_f64: the one-launch fit timed out waiting for its own workgroups (the GPU is shared with "
                           "another long-running launch?); set SCAML_BLOCKED_FIT_PATH=1 to use the sequence of launches")
    bad = torch.nonzero(info > 0).flatten()
    if bad.numel():
        raise NotPSDError(
            f"Matrix not positive definite after repeatedly adding jitter up to 1.0e-06 "
            f"(tasks {bad.tolist()[:8]}{'...' if bad.numel() > 8 else ''})."
        )
