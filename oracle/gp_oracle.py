"""CPU oracle for the ScaML-GP batched GP-inference hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain torch-fp64 CPU restatement of the arithmetic the reference
executes for one source task / the target task.  It is the *checker* for the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``scalable-meta-learning-with-gaussian-processes_amd/`` imports it.

PARITY UNPINNED.  The reference (``/root/reference/scamlgp``) delegates every numeric
step to third-party packages that are not vendored and not installable here
(gpytorch==1.9.0, linear-operator==0.2.0, botorch==0.7.3; ``poetry.lock:703-704,
1020-1021,135-136``), and none of the reference's own tests pins a kernel value, a
Cholesky factor, a posterior or an MLL (``tests/conftest.py:6-8`` seeds randomly,
``tests/optimizer_test.py`` asserts behaviour only).  The oracle therefore restates
the *published* algorithm of those pinned versions, anchored on the reference call
sites cited per function, and is cross-validated in ``tests/test_oracle.py`` against two
independent implementations that are importable here (scikit-learn
``GaussianProcessRegressor`` with a fixed kernel, and ``scipy.linalg``).

Conventions: all tensors fp64 on CPU; ``theta`` for a task is the *constrained*
hyper-parameter vector ``[lengthscale_0..lengthscale_{D-1}, outputscale, noise]``.
Kernel kinds: ``KIND_RBF = 0``, ``KIND_MATERN52 = 1``.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

KIND_RBF = 0
KIND_MATERN52 = 1

_LOG_2PI = math.log(2.0 * math.pi)


# ---------------------------------------------------------------------------
# A1  parameterisation: gpytorch Interval constraint with the default sigmoid transform
#     (reference call sites: scamlgp/model.py:31, 52-56, 64-68, 91-103)
# ---------------------------------------------------------------------------
def interval_transform(raw: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    """theta = lo + (hi - lo) * sigmoid(raw)  (gpytorch.constraints.Interval.transform)."""
    return lo + (hi - lo) * torch.sigmoid(raw)


def interval_inverse_transform(theta: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    """raw = logit((theta - lo) / (hi - lo))  (Interval.inverse_transform)."""
    p = (theta - lo) / (hi - lo)
    return torch.log(p) - torch.log1p(-p)


# ---------------------------------------------------------------------------
# priors (gpytorch.priors.GammaPrior / LogNormalPrior = torch.distributions log_prob)
#     reference call sites: scamlgp/model.py:28, 41-42, 82, 86, 328
# ---------------------------------------------------------------------------
def gamma_log_prob(x: torch.Tensor, concentration: float, rate: float) -> torch.Tensor:
    return (
        concentration * math.log(rate)
        + (concentration - 1.0) * torch.log(x)
        - rate * x
        - math.lgamma(concentration)
    )


def lognormal_log_prob(x: torch.Tensor, loc: float, scale: float) -> torch.Tensor:
    lx = torch.log(x)
    return -lx - math.log(scale) - 0.5 * _LOG_2PI - (lx - loc) ** 2 / (2.0 * scale * scale)


def source_gp_log_prior(theta: torch.Tensor) -> torch.Tensor:
    """Sum of the source-GP hyper-priors evaluated on the constrained values.

    lengthscale ~ Gamma(3, 6), outputscale ~ Gamma(2, 0.15) (scamlgp/model.py:41-42),
    noise ~ LogNormal(-8, 2) (scamlgp/model.py:28).  theta[..., :D] lengthscales,
    theta[..., D] outputscale, theta[..., D+1] noise.
    """
    ls, os_, noise = theta[..., :-2], theta[..., -2], theta[..., -1]
    return (
        gamma_log_prob(ls, 3.0, 6.0).sum(-1)
        + gamma_log_prob(os_, 2.0, 0.15)
        + lognormal_log_prob(noise, -8.0, 2.0)
    )


def target_gp_log_prior(theta: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """Target-GP hyper-priors: lengthscale ~ LogNormal(0.5, 1.5), outputscale ~
    LogNormal(-2, 3) (scamlgp/model.py:82, 86), noise ~ LogNormal(-8, 2) (:28),
    weights ~ Gamma(1, 1) each (:326-331)."""
    ls, os_, noise = theta[..., :-2], theta[..., -2], theta[..., -1]
    return (
        lognormal_log_prob(ls, 0.5, 1.5).sum(-1)
        + lognormal_log_prob(os_, -2.0, 3.0)
        + lognormal_log_prob(noise, -8.0, 2.0)
        + gamma_log_prob(weights, 1.0, 1.0).sum(-1)
    )


# ---------------------------------------------------------------------------
# A2  botorch Standardize(m=1)   (reference call sites: scamlgp/model.py:185, 272-276)
# ---------------------------------------------------------------------------
def standardize_fit(Y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Return (mean, std) over dim -2 with botorch's floor: std < 1e-8 -> 1.

    Y: (..., N, 1).  Unbiased std (N-1); a single point (NaN std) also maps to 1.
    """
    m = Y.mean(dim=-2, keepdim=True)
    if Y.shape[-2] < 2:
        s = torch.ones_like(m)
    else:
        s = Y.std(dim=-2, keepdim=True)
        s = torch.where(s >= 1e-8, s, torch.ones_like(s))
    return m, s


# ---------------------------------------------------------------------------
# A3  gpytorch distance (gpytorch 1.9.0 kernels/kernel.py ``sq_dist`` / ``dist``)
# ---------------------------------------------------------------------------
def sq_dist_gpytorch(x1: torch.Tensor, x2: torch.Tensor, x1_eq_x2: bool) -> torch.Tensor:
    """||a||^2 - 2 a.b + ||b||^2 after centring on x1's mean, diagonal zeroed when
    x1 is x2, clamped at 0 — the expansion gpytorch uses (one matmul)."""
    adjustment = x1.mean(-2, keepdim=True)
    x1 = x1 - adjustment
    x2 = x1 if x1_eq_x2 else x2 - adjustment
    x1_norm = x1.pow(2).sum(dim=-1, keepdim=True)
    x1_pad = torch.ones_like(x1_norm)
    x2_norm = x2.pow(2).sum(dim=-1, keepdim=True)
    x2_pad = torch.ones_like(x2_norm)
    x1_ = torch.cat([-2.0 * x1, x1_norm, x1_pad], dim=-1)
    x2_ = torch.cat([x2, x2_pad, x2_norm], dim=-1)
    res = x1_.matmul(x2_.transpose(-2, -1))
    if x1_eq_x2:
        res.diagonal(dim1=-2, dim2=-1).fill_(0)
    return res.clamp_min_(0)


def sq_dist_direct(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """sum_d (a_d - b_d)^2 — the formulation the HIP kernels use (more accurate; differs
    from ``sq_dist_gpytorch`` by O(eps * ||x||^2))."""
    diff = x1.unsqueeze(-2) - x2.unsqueeze(-3)
    return diff.pow(2).sum(-1)


# ---------------------------------------------------------------------------
# A4  kernels: ScaleKernel(RBFKernel | MaternKernel(nu=2.5), ARD)
#     (reference call sites: scamlgp/model.py:44-70, 87-105)
# ---------------------------------------------------------------------------
def kernel_matrix(
    x1: torch.Tensor,
    x2: Optional[torch.Tensor],
    lengthscale: torch.Tensor,
    outputscale: torch.Tensor,
    kind: int,
    dist: str = "gpytorch",
) -> torch.Tensor:
    """outputscale * k(x1 / l, x2 / l).  ``x2=None`` means x2 is x1 (training block)."""
    same = x2 is None
    if kind == KIND_MATERN52:
        # MaternKernel.forward subtracts the global mean of x1 before scaling.
        mean = x1.reshape(-1, x1.shape[-1]).mean(0)
        x1 = x1 - mean
        x2 = None if same else x2 - mean
    a = x1 / lengthscale
    b = a if same else x2 / lengthscale
    if dist == "gpytorch":
        d2 = sq_dist_gpytorch(a, b, same)
    else:
        d2 = sq_dist_direct(a, b)
    if kind == KIND_RBF:
        k = torch.exp(-0.5 * d2)
    elif kind == KIND_MATERN52:
        r = d2.clamp_min(1e-30).sqrt()
        k = (1.0 + math.sqrt(5.0) * r + (5.0 / 3.0) * r * r) * torch.exp(-math.sqrt(5.0) * r)
    else:
        raise ValueError(f"unknown kernel kind {kind}")
    return outputscale * k


# ---------------------------------------------------------------------------
# A6  linear_operator.utils.cholesky.psd_safe_cholesky  (reached from utils.py:171-177)
# ---------------------------------------------------------------------------
class NotPSDError(RuntimeError):
    pass


def psd_safe_cholesky(A: torch.Tensor, max_tries: int = 3) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """cholesky_ex first; on failure add 1e-8 * 10^i to the diagonal of the failing batch
    members only, at most ``max_tries`` times.  Returns (L, info_first_try, jitter_used).
    """
    L, info = torch.linalg.cholesky_ex(A)
    info0 = info.clone()
    jitter_used = torch.zeros(A.shape[:-2], dtype=A.dtype)
    if not torch.any(info):
        return L, info0, jitter_used
    if torch.isnan(A).any():
        raise NotPSDError("cholesky: input contains NaN")
    Aprime = A.clone()
    jitter_prev = 0.0
    for i in range(max_tries):
        jitter_new = 1e-8 * (10 ** i)
        failing = info > 0
        diag_add = (failing.to(A.dtype) * (jitter_new - jitter_prev))
        Aprime.diagonal(dim1=-2, dim2=-1).add_(diag_add.unsqueeze(-1))
        jitter_used = torch.where(failing, torch.full_like(jitter_used, jitter_new), jitter_used)
        jitter_prev = jitter_new
        L, info = torch.linalg.cholesky_ex(Aprime)
        if not torch.any(info):
            return L, info0, jitter_used
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter_new:.1e}.")


# ---------------------------------------------------------------------------
# A5  one "task-posterior": K + noise, Cholesky, alpha, quad, logdet, MLL
#     (reference chain: utils.py:171-177 -> ExactMarginalLogLikelihood -> MVN.log_prob)
# ---------------------------------------------------------------------------
def gp_fit(
    X: torch.Tensor, y: torch.Tensor, theta: torch.Tensor, kind: int, dist: str = "gpytorch"
) -> dict:
    """X (N, D), y (N,) already standardised, theta (D+2,) constrained.

    Returns K (with noise), L, v = L^-1 y, alpha = K^-1 y, quad, logdet,
    mll (WITHOUT priors, divided by N), info (first-try), jitter.
    """
    N, D = X.shape
    ls, os_, noise = theta[:D], theta[D], theta[D + 1]
    K = kernel_matrix(X, None, ls, os_, kind, dist)
    K = K + noise * torch.eye(N, dtype=X.dtype)
    L, info, jitter = psd_safe_cholesky(K)
    v = torch.linalg.solve_triangular(L, y.unsqueeze(-1), upper=False)
    alpha = torch.linalg.solve_triangular(L.transpose(-1, -2), v, upper=True).squeeze(-1)
    quad = (v * v).sum()
    logdet = 2.0 * torch.log(torch.diagonal(L)).sum()
    mll = -0.5 * (quad + logdet + N * _LOG_2PI) / N
    return dict(K=K, L=L, v=v.squeeze(-1), alpha=alpha, quad=quad, logdet=logdet, mll=mll,
                info=info, jitter=jitter)


def gp_fit_stack_loop(X: torch.Tensor, y: torch.Tensor, theta: torch.Tensor, kind: int,
                      dist: str = "gpytorch") -> dict:
    """The reference's shape of work: a Python loop over tasks (scamlgp/model.py:176-188),
    one dense Cholesky per task.  X (T, N, D), y (T, N), theta (T, D+2)."""
    outs = [gp_fit(X[t], y[t], theta[t], kind, dist) for t in range(X.shape[0])]
    return {k: torch.stack([o[k] for o in outs]) for k in ("L", "alpha", "quad", "logdet", "mll", "info", "jitter")}


def gp_fit_stack_batched(X: torch.Tensor, y: torch.Tensor, theta: torch.Tensor, kind: int) -> dict:
    """Same arithmetic as one batched torch.linalg call (stronger CPU baseline)."""
    T, N, D = X.shape
    ls, os_, noise = theta[:, None, :D], theta[:, D, None, None], theta[:, D + 1]
    K = kernel_matrix(X, None, ls, os_, kind)
    K = K + noise[:, None, None] * torch.eye(N, dtype=X.dtype)
    L, info, jitter = psd_safe_cholesky(K)
    v = torch.linalg.solve_triangular(L, y.unsqueeze(-1), upper=False)
    alpha = torch.linalg.solve_triangular(L.transpose(-1, -2), v, upper=True).squeeze(-1)
    quad = (v * v).sum((-1, -2))
    logdet = 2.0 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
    mll = -0.5 * (quad + logdet + N * _LOG_2PI) / N
    return dict(L=L, alpha=alpha, quad=quad, logdet=logdet, mll=mll, info=info, jitter=jitter)


def mll_with_source_priors(X, y, theta, kind, dist="gpytorch") -> torch.Tensor:
    """Full training objective of a source GP (A5): [log N(y|0,K) + sum log p(theta)] / N."""
    N = X.shape[0]
    out = gp_fit(X, y, theta, kind, dist)
    return out["mll"] + source_gp_log_prior(theta) / N


def mll_value_and_grad_raw(X, y, raw, kind, bounds, with_priors=True):
    """Value and gradient of the MLL w.r.t. the RAW (unconstrained) parameters through the
    sigmoid-Interval transform, by autograd — the oracle for the analytic HIP backward.

    raw (D+2,), bounds = [(lo, hi)] * (D+2) in theta order."""
    raw = raw.clone().requires_grad_(True)
    lo = torch.tensor([b[0] for b in bounds], dtype=raw.dtype)
    hi = torch.tensor([b[1] for b in bounds], dtype=raw.dtype)
    theta = lo + (hi - lo) * torch.sigmoid(raw)
    N, D = X.shape
    K = _kernel_matrix_grad(X, theta, kind)
    K = K + theta[D + 1] * torch.eye(N, dtype=X.dtype)
    L = torch.linalg.cholesky(K)
    v = torch.linalg.solve_triangular(L, y.unsqueeze(-1), upper=False)
    val = -0.5 * ((v * v).sum() + 2.0 * torch.log(torch.diagonal(L)).sum() + N * _LOG_2PI)
    if with_priors:
        val = val + source_gp_log_prior(theta)
    val = val / N
    (g,) = torch.autograd.grad(val, raw)
    return val.detach(), g, theta.detach()


def _kernel_matrix_grad(X, theta, kind):
    """Differentiable kernel matrix (direct differences; no in-place ops)."""
    D = X.shape[-1]
    a = X / theta[:D]
    d2 = (a.unsqueeze(-2) - a.unsqueeze(-3)).pow(2).sum(-1)
    if kind == KIND_RBF:
        k = torch.exp(-0.5 * d2)
    else:
        # sqrt has an infinite derivative at 0: mask the diagonal like gpytorch's clamp does
        r = torch.sqrt(d2.clamp_min(1e-30))
        k = (1.0 + math.sqrt(5.0) * r + (5.0 / 3.0) * d2) * torch.exp(-math.sqrt(5.0) * r)
    return theta[D] * k


# ---------------------------------------------------------------------------
# A7  source posterior at query points (reference call sites: model.py:128, 281)
# ---------------------------------------------------------------------------
def source_posterior(
    xq: torch.Tensor, X: torch.Tensor, theta: torch.Tensor, kind: int, L: torch.Tensor,
    alpha: torch.Tensor, y_mean: float, y_std: float, full_cov: bool = True, dist: str = "gpytorch",
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Posterior of one source GP at xq (M, D), un-standardised (botorch
    ``Standardize.untransform_posterior``): mu = m + s * K_* alpha,
    Sigma = s^2 (k(x,x) - V^T V), V = L^-1 K_*^T.  No observation noise."""
    D = X.shape[-1]
    ls, os_ = theta[:D], theta[D]
    # (MaternKernel centres on the mean of its first argument, xq here: a pure translation)
    Ks = kernel_matrix(xq, X, ls, os_, kind, dist)
    mu = Ks @ alpha
    V = torch.linalg.solve_triangular(L, Ks.transpose(-1, -2), upper=False)
    if full_cov:
        Kss = kernel_matrix(xq, None, ls, os_, kind, dist)
        cov = Kss - V.transpose(-1, -2) @ V
    else:
        cov = os_ - (V * V).sum(0)
    return y_mean + y_std * mu, (y_std ** 2) * cov


# ---------------------------------------------------------------------------
# A8  pruning mask + weighted target prior (reference: model.py:192-215, 108-135)
# ---------------------------------------------------------------------------
def significant_weights_mask(weights: torch.Tensor, std_Y_vals: torch.Tensor, threshold: float) -> torch.Tensor:
    num_weights = len(weights)
    w_times_sigma = weights * std_Y_vals
    norm_weights = w_times_sigma * num_weights / w_times_sigma.sum()
    return norm_weights >= threshold


def target_prior(mus: torch.Tensor, covs: torch.Tensor, weights: torch.Tensor,
                 mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """mu_s = sum_i w_i mu_i, Sigma_s = sum_i w_i^2 Sigma_i over the (masked) tasks.
    mus (T, M), covs (T, M, M) or (T, M) for diagonal-only."""
    w = weights if mask is None else weights * mask.to(weights.dtype)
    mu = (w[:, None] * mus).sum(0)
    if covs.dim() == 3:
        cov = ((w ** 2)[:, None, None] * covs).sum(0)
    else:
        cov = ((w ** 2)[:, None] * covs).sum(0)
    return mu, cov


# ---------------------------------------------------------------------------
# A9/A10  target model: training objective and posterior (reference: model.py:359-384,
#          optimizer.py:176-185; ExactGP eval-mode algebra)
# ---------------------------------------------------------------------------
def target_train_mll(
    Xt: torch.Tensor, yt_std: torch.Tensor, source_means: torch.Tensor, source_covs: torch.Tensor,
    weights: torch.Tensor, theta_t: torch.Tensor, kind: int, m_all: float, s_all: float,
    with_priors: bool = True, dist: str = "gpytorch",
) -> torch.Tensor:
    """ScaMLGP.forward training branch (model.py:360-363, 376-383) + MLL (A5/A9).

    source_means (n, T), source_covs (n, n, T) as cached at model.py:282-289;
    yt_std = (y - m_all) / s_all."""
    n, D = Xt.shape
    mean = (source_means @ weights - m_all) / s_all
    cov = (source_covs @ weights ** 2) / (s_all ** 2)
    cov = cov + kernel_matrix(Xt, None, theta_t[:D], theta_t[D], kind, dist)
    cov = cov + theta_t[D + 1] * torch.eye(n, dtype=Xt.dtype)
    L, _, _ = psd_safe_cholesky(cov)
    v = torch.linalg.solve_triangular(L, (yt_std - mean).unsqueeze(-1), upper=False)
    val = -0.5 * ((v * v).sum() + 2.0 * torch.log(torch.diagonal(L)).sum() + n * _LOG_2PI)
    if with_priors:
        val = val + target_gp_log_prior(theta_t, weights)
    return val / n


def target_posterior(
    xq: torch.Tensor, Xt: torch.Tensor, yt: torch.Tensor,
    prior_mean_joint: torch.Tensor, prior_cov_joint: torch.Tensor,
    theta_t: torch.Tensor, kind: int, m_all: float, s_all: float, dist: str = "gpytorch",
) -> Tuple[torch.Tensor, torch.Tensor]:
    """A10.  prior_mean_joint (n+M,), prior_cov_joint (n+M, n+M): the weighted source prior
    (A8) at cat(Xt, xq) in ORIGINAL units.  Returns the posterior mean / covariance at xq,
    un-standardised with (m_all, s_all)."""
    n = Xt.shape[0]
    D = Xt.shape[-1]
    xall = torch.cat([Xt, xq], 0)
    mean = (prior_mean_joint - m_all) / s_all
    cov = prior_cov_joint / (s_all ** 2) + kernel_matrix(xall, None, theta_t[:D], theta_t[D], kind, dist)
    Knn = cov[:n, :n] + theta_t[D + 1] * torch.eye(n, dtype=Xt.dtype)
    Kxn = cov[n:, :n]
    Kxx = cov[n:, n:]
    L, _, _ = psd_safe_cholesky(Knn)
    resid = ((yt - m_all) / s_all - mean[:n]).unsqueeze(-1)
    a = torch.cholesky_solve(resid, L).squeeze(-1)
    mu = mean[n:] + Kxn @ a
    V = torch.linalg.solve_triangular(L, Kxn.transpose(-1, -2), upper=False)
    S = Kxx - V.transpose(-1, -2) @ V
    return m_all + s_all * mu, (s_all ** 2) * S


# ---------------------------------------------------------------------------
# A11  acquisition functions (reference: utils.py:215-224; optimizer.py:96-98)
# ---------------------------------------------------------------------------
def ucb_minimize(mu: torch.Tensor, var: torch.Tensor, beta: float = 9.0) -> torch.Tensor:
    """botorch UpperConfidenceBound(beta, maximize=False): -mu + sqrt(beta * var)."""
    return -mu + torch.sqrt(beta * var)


def expected_improvement_minimize(mu: torch.Tensor, var: torch.Tensor, best_f: float) -> torch.Tensor:
    """botorch ExpectedImprovement(maximize=False): sigma = sqrt(max(var, 1e-9)),
    u = -(mu - best_f) / sigma, EI = sigma * (phi(u) + u * Phi(u))."""
    sigma = var.clamp_min(1e-9).sqrt()
    u = -(mu - best_f) / sigma
    normal = torch.distributions.Normal(torch.zeros_like(u), torch.ones_like(u))
    return sigma * (torch.exp(normal.log_prob(u)) + u * normal.cdf(u))
