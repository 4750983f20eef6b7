"""Hyper-parameter plumbing of the GP stack: sigmoid-Interval re-parameterisation, priors, prior
sampling and a batched L-BFGS that advances many independent problems with one objective call
per iteration (one fused GPU launch for all tasks x restarts).

Restates, without gpytorch/botorch, the pieces the reference takes from them:
  scamlgp/model.py:25-33, 36-70, 73-105   constraints (Interval, sigmoid transform), initial values, priors
  scamlgp/utils.py:31-69                  sample_all_priors (resample until the constraint's
                                          inverse transform is finite, up to 5 retries)
  scamlgp/utils.py:139-212                warm start + prior-sampled restarts, keep the best
Plain torch only (runs on CPU or GPU tensors alike); no hot-path arithmetic lives here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional, Tuple

import torch

_LOG_2PI = math.log(2.0 * math.pi)


# --- constraints -------------------------------------------------------------------------
@dataclass(frozen=True)
class Interval:
    """gpytorch.constraints.Interval with the default sigmoid transform."""
    lower: float
    upper: float

    def transform(self, raw: torch.Tensor) -> torch.Tensor:
        return self.lower + (self.upper - self.lower) * torch.sigmoid(raw)

    def inverse_transform(self, value: torch.Tensor) -> torch.Tensor:
        p = (value - self.lower) / (self.upper - self.lower)
        return torch.log(p) - torch.log1p(-p)

    def dtransform(self, raw: torch.Tensor) -> torch.Tensor:
        """d theta / d raw."""
        s = torch.sigmoid(raw)
        return (self.upper - self.lower) * s * (1.0 - s)


# --- priors (log-densities on the constrained values, and samplers) -----------------------------
@dataclass(frozen=True)
class GammaPrior:
    concentration: float
    rate: float

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        c, r = self.concentration, self.rate
        return c * math.log(r) + (c - 1.0) * torch.log(x) - r * x - math.lgamma(c)

    def dlog_prob(self, x: torch.Tensor) -> torch.Tensor:
        return (self.concentration - 1.0) / x - self.rate

    def sample(self, shape, generator=None, dtype=torch.float64, device=None) -> torch.Tensor:
        g = torch.distributions.Gamma(torch.tensor(self.concentration, dtype=dtype), torch.tensor(self.rate, dtype=dtype))
        # torch.distributions draws from the global RNG (the reference seeds it: scamlgp/model.py:163-164)
        return g.sample(shape).to(device)


@dataclass(frozen=True)
class LogNormalPrior:
    loc: float
    scale: float

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        lx = torch.log(x)
        return -lx - math.log(self.scale) - 0.5 * _LOG_2PI - (lx - self.loc) ** 2 / (2.0 * self.scale ** 2)

    def dlog_prob(self, x: torch.Tensor) -> torch.Tensor:
        return -(1.0 + (torch.log(x) - self.loc) / self.scale ** 2) / x

    def sample(self, shape, generator=None, dtype=torch.float64, device=None) -> torch.Tensor:
        d = torch.distributions.LogNormal(torch.tensor(self.loc, dtype=dtype), torch.tensor(self.scale, dtype=dtype))
        return d.sample(shape).to(device)


@dataclass(frozen=True)
class HyperSpec:
    """Constraint + prior + initial value of the three hyper-parameter groups of one GP:
    lengthscales (D values), outputscale, noise variance."""
    ls_constraint: Interval
    ls_prior: object
    ls_init: float
    os_constraint: Interval
    os_prior: object
    os_init: float
    noise_constraint: Interval
    noise_prior: object
    noise_init: float

    def constraints(self, D: int):
        return [self.ls_constraint] * D + [self.os_constraint, self.noise_constraint]

    def priors(self, D: int):
        return [self.ls_prior] * D + [self.os_prior, self.noise_prior]

    def bounds(self, D: int, dtype=torch.float64, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
        # memoised per (D, dtype, device): building device tensors from Python lists is a synchronous host-to-device
        # copy, and this is called several times per objective evaluation of the batched optimiser
        key = (D, dtype, str(device), tuple((c.lower, c.upper) for c in (self.ls_constraint, self.os_constraint, self.noise_constraint)))
        cache = self.__dict__.setdefault("_bounds_cache", {})
        if key not in cache:
            lo = torch.tensor([c.lower for c in self.constraints(D)], dtype=dtype, device=device)
            hi = torch.tensor([c.upper for c in self.constraints(D)], dtype=dtype, device=device)
            cache[key] = (lo, hi)
        return cache[key]

    def init_theta(self, D: int, dtype=torch.float64, device=None) -> torch.Tensor:
        return torch.tensor([self.ls_init] * D + [self.os_init, self.noise_init], dtype=dtype, device=device)

    # vectorised over leading dims: theta (..., D+2)
    def to_theta(self, raw: torch.Tensor) -> torch.Tensor:
        lo, hi = self.bounds(raw.shape[-1] - 2, raw.dtype, raw.device)
        return lo + (hi - lo) * torch.sigmoid(raw)

    def to_raw(self, theta: torch.Tensor) -> torch.Tensor:
        lo, hi = self.bounds(theta.shape[-1] - 2, theta.dtype, theta.device)
        p = (theta - lo) / (hi - lo)
        return torch.log(p) - torch.log1p(-p)

    def dtheta_draw(self, raw: torch.Tensor) -> torch.Tensor:
        lo, hi = self.bounds(raw.shape[-1] - 2, raw.dtype, raw.device)
        s = torch.sigmoid(raw)
        return (hi - lo) * s * (1.0 - s)

    def log_prior(self, theta: torch.Tensor) -> torch.Tensor:
        D = theta.shape[-1] - 2
        return (self.ls_prior.log_prob(theta[..., :D]).sum(-1) + self.os_prior.log_prob(theta[..., D])
                + self.noise_prior.log_prob(theta[..., D + 1]))

    def dlog_prior(self, theta: torch.Tensor) -> torch.Tensor:
        D = theta.shape[-1] - 2
        return torch.cat([self.ls_prior.dlog_prob(theta[..., :D]), self.os_prior.dlog_prob(theta[..., D:D + 1]),
                          self.noise_prior.dlog_prob(theta[..., D + 1:])], -1)

    def sample_prior(self, shape, D: int, num_retries: int = 5, dtype=torch.float64, device=None) -> torch.Tensor:
        """scamlgp/utils.py:31-69: draw every hyper-parameter from its prior; a draw whose inverse
        constraint transform is not finite (outside the interval) is redrawn, at most
        ``num_retries`` times, then a RuntimeError is raised."""
        out = torch.empty(*shape, D + 2, dtype=dtype, device=device)
        cons, pris = self.constraints(D), self.priors(D)
        groups = [(slice(0, D), cons[0], pris[0]), (slice(D, D + 1), cons[D], pris[D]), (slice(D + 1, D + 2), cons[D + 1], pris[D + 1])]
        for sl, con, pri in groups:
            width = sl.stop - sl.start
            s = pri.sample((*shape, width), dtype=dtype, device=device)
            for attempt in range(num_retries + 1):
                # a model's draw is valid when every element maps to a finite raw value
                bad = ~torch.isfinite(con.inverse_transform(s)).all(-1, keepdim=True)
                if not bool(bad.any()):
                    break
                if attempt == num_retries:
                    raise RuntimeError(f"Sampling of a hyper-prior failed {num_retries} times. Please check the "
                                       "compatibility between prior support and the constraint.")
                s = torch.where(bad, pri.sample((*shape, width), dtype=dtype, device=device), s)
            out[..., sl] = s
        return out


# reference defaults -----------------------------------------------------------------------------
def source_gp_spec() -> HyperSpec:
    """scamlgp/model.py:25-33 (likelihood) and :36-70 (_get_kernel_source_gp)."""
    return HyperSpec(Interval(1e-4, 1e2), GammaPrior(3.0, 6.0), 0.5,
                     Interval(1e-4, 1e2), GammaPrior(2.0, 0.15), 1.0,
                     Interval(1e-8, 1e-2), LogNormalPrior(-8.0, 2.0), 1e-3)


def target_gp_spec() -> HyperSpec:
    """scamlgp/model.py:25-33 (likelihood) and :73-105 (_get_default_kernel)."""
    return HyperSpec(Interval(1e-4, 1e2), LogNormalPrior(0.5, 1.5), 1.0,
                     Interval(1e-4, 1e2), LogNormalPrior(-2.0, 3.0), 0.1,
                     Interval(1e-8, 1e-2), LogNormalPrior(-8.0, 2.0), 1e-3)


# --- module-like parameter holders (what the reference hands around as `likelihood` / `covar_module`) ---------
# The reference passes gpytorch modules: GaussianLikelihood(noise_prior, noise_constraint) and
# ScaleKernel(RBFKernel | MaternKernel(ard_num_dims, lengthscale_prior, lengthscale_constraint), outputscale_prior,
# outputscale_constraint) (scamlgp/model.py:25-105), and re-uses the fitted ones of the previous model when it
# rebuilds ScaMLGP (scamlgp/optimizer.py:176-183).  These classes carry the same state -- constraint, prior and
# the current (raw) value -- under the same attribute names, without gpytorch; the arithmetic stays in the kernels.
class _Param:
    """One constrained parameter group: raw tensor + Interval + prior."""

    def __init__(self, constraint: Interval, prior, value):
        self.constraint, self.prior = constraint, prior
        self.raw = constraint.inverse_transform(torch.as_tensor(value, dtype=torch.float64))

    @property
    def value(self) -> torch.Tensor:
        return self.constraint.transform(self.raw)

    def set(self, value) -> None:
        self.raw = self.constraint.inverse_transform(torch.as_tensor(value, dtype=torch.float64).to(self.raw.device))

    def to(self, device) -> None:
        self.raw = self.raw.to(device)


class GaussianLikelihood:
    """Homoskedastic noise sigma^2 (scamlgp/model.py:25-33): ``.noise``, ``.raw_noise``, ``.noise_prior``,
    ``.raw_noise_constraint``."""

    def __init__(self, noise_prior=None, noise_constraint: Optional[Interval] = None, initial_value: float = 1e-3):
        self._p = _Param(noise_constraint or Interval(1e-8, 1e-2), noise_prior or LogNormalPrior(-8.0, 2.0), initial_value)

    noise = property(lambda self: self._p.value, lambda self, v: self._p.set(v))
    raw_noise = property(lambda self: self._p.raw)
    noise_prior = property(lambda self: self._p.prior)
    raw_noise_constraint = property(lambda self: self._p.constraint)

    def to(self, device):
        self._p.to(device)
        return self


class _BaseKernel:
    kind = 0

    def __init__(self, ard_num_dims: int = 1, lengthscale_prior=None, lengthscale_constraint: Optional[Interval] = None,
                 initial_value=0.5):
        self.ard_num_dims = int(ard_num_dims)
        init = torch.as_tensor(initial_value, dtype=torch.float64).expand(self.ard_num_dims).clone()
        self._p = _Param(lengthscale_constraint or Interval(1e-4, 1e2), lengthscale_prior or GammaPrior(3.0, 6.0), init)

    lengthscale = property(lambda self: self._p.value, lambda self, v: self._p.set(v))
    raw_lengthscale = property(lambda self: self._p.raw)
    lengthscale_prior = property(lambda self: self._p.prior)
    raw_lengthscale_constraint = property(lambda self: self._p.constraint)


class RBFKernel(_BaseKernel):
    """k = exp(-r^2 / 2), ARD (the reference's default base kernel, scamlgp/model.py:168-171, 292-297)."""
    kind = 0


class MaternKernel(_BaseKernel):
    """Matern nu = 5/2, ARD (gpytorch MaternKernel's default nu)."""
    kind = 1
    nu = 2.5


class ScaleKernel:
    """outputscale * base_kernel (scamlgp/model.py:44-70, 87-105)."""

    def __init__(self, base_kernel: _BaseKernel, outputscale_prior=None, outputscale_constraint: Optional[Interval] = None,
                 initial_value: float = 1.0):
        self.base_kernel = base_kernel
        self._p = _Param(outputscale_constraint or Interval(1e-4, 1e2), outputscale_prior or GammaPrior(2.0, 0.15), initial_value)

    outputscale = property(lambda self: self._p.value, lambda self, v: self._p.set(v))
    raw_outputscale = property(lambda self: self._p.raw)
    outputscale_prior = property(lambda self: self._p.prior)
    raw_outputscale_constraint = property(lambda self: self._p.constraint)
    kind = property(lambda self: self.base_kernel.kind)

    def to(self, device):
        self._p.to(device)
        self.base_kernel._p.to(device)
        return self


def get_default_likelihood() -> GaussianLikelihood:
    """scamlgp/model.py:25-33 ``_get_default_likelihood``."""
    return GaussianLikelihood(LogNormalPrior(-8.0, 2.0), Interval(1e-8, 1e-2), 1e-3)


def get_kernel_source_gp(base_kernel=RBFKernel, ard_num_dims: int = 1) -> ScaleKernel:
    """scamlgp/model.py:36-70 ``_get_kernel_source_gp``: the priors of botorch's SingleTaskGP."""
    return ScaleKernel(base_kernel(ard_num_dims, GammaPrior(3.0, 6.0), Interval(1e-4, 1e2), 0.5),
                       GammaPrior(2.0, 0.15), Interval(1e-4, 1e2), 1.0)


def get_default_kernel(base_kernel=RBFKernel, ard_num_dims: int = 1) -> ScaleKernel:
    """scamlgp/model.py:73-105 ``_get_default_kernel`` (target GP: broad log-normal priors)."""
    return ScaleKernel(base_kernel(ard_num_dims, LogNormalPrior(0.5, 1.5), Interval(1e-4, 1e2), 1.0),
                       LogNormalPrior(-2.0, 3.0), Interval(1e-4, 1e2), 0.1)


def spec_from_modules(likelihood: GaussianLikelihood, covar_module: ScaleKernel) -> HyperSpec:
    """The constraints / priors of a (likelihood, covar_module) pair as a HyperSpec; the init values are the
    modules' CURRENT values (lengthscales: their mean -- HyperSpec inits are per group)."""
    bk = covar_module.base_kernel
    return HyperSpec(bk.raw_lengthscale_constraint, bk.lengthscale_prior, float(bk.lengthscale.mean()),
                     covar_module.raw_outputscale_constraint, covar_module.outputscale_prior, float(covar_module.outputscale),
                     likelihood.raw_noise_constraint, likelihood.noise_prior, float(likelihood.noise))


def modules_from_spec(spec: HyperSpec, kind: int, D: int):
    bk = (MaternKernel if kind == 1 else RBFKernel)(D, spec.ls_prior, spec.ls_constraint, spec.ls_init)
    return (GaussianLikelihood(spec.noise_prior, spec.noise_constraint, spec.noise_init),
            ScaleKernel(bk, spec.os_prior, spec.os_constraint, spec.os_init))


# --- batched L-BFGS -----------------------------------------------------------------------------
@dataclass
class LBFGSResult:
    x: torch.Tensor          # (B, P) final points
    f: torch.Tensor          # (B,) final objective values (minimised)
    n_iter: int
    n_eval: int
    converged: torch.Tensor  # (B,) bool
    failed: torch.Tensor     # (B,) bool: objective returned non-finite at the start


def batched_lbfgs(fun: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]], x0: torch.Tensor,
                  max_iter: int = 200, history: int = 10, gtol: float = 1e-5, ftol: float = 2.2e-9,
                  c1: float = 1e-4, max_ls: int = 20) -> LBFGSResult:
    """Minimise B independent functions of P variables at once.  ``fun(x)`` -> (f (B,), g (B, P))
    evaluates all of them in one call (for the GP stack: one fused launch).  Two-loop L-BFGS with a
    per-problem backtracking (Armijo) line search; problems whose trial value is not finite shrink
    their step; stopping rules follow scipy's L-BFGS-B (projected-gradient ``gtol``, relative
    decrease ``ftol``), which botorch's fit_gpytorch_mll drives in the reference
    (scamlgp/utils.py:175).  No bounds: the raw parameters are unconstrained."""
    x = x0.clone()
    B, P = x.shape
    f, g = fun(x)
    n_eval = 1
    failed = ~torch.isfinite(f) | ~torch.isfinite(g).all(-1)
    f = torch.where(failed, torch.full_like(f, float("inf")), f)
    g = torch.where(failed.unsqueeze(-1), torch.zeros_like(g), g)
    S = torch.zeros(B, history, P, dtype=x.dtype, device=x.device)
    Y = torch.zeros_like(S)
    rho = torch.zeros(B, history, dtype=x.dtype, device=x.device)
    valid = torch.zeros(B, history, dtype=torch.bool, device=x.device)
    done = failed.clone()
    it = 0
    for it in range(1, max_iter + 1):
        # two-loop recursion (newest pair at index 0)
        q = g.clone()
        alphas = []
        for i in range(history):
            a = torch.where(valid[:, i], rho[:, i] * (S[:, i] * q).sum(-1), torch.zeros_like(f))
            q = q - a.unsqueeze(-1) * Y[:, i]
            alphas.append(a)
        ys = (S[:, 0] * Y[:, 0]).sum(-1)
        yy = (Y[:, 0] * Y[:, 0]).sum(-1)
        gamma = torch.where(valid[:, 0] & (yy > 0), ys / yy.clamp_min(1e-300), torch.ones_like(f))
        r = gamma.unsqueeze(-1) * q
        for i in reversed(range(history)):
            b = torch.where(valid[:, i], rho[:, i] * (Y[:, i] * r).sum(-1), torch.zeros_like(f))
            r = r + (alphas[i] - b).unsqueeze(-1) * S[:, i]
        d = -r
        gd = (g * d).sum(-1)
        # not a descent direction (or first iteration): steepest descent, scaled like scipy's first step
        bad_dir = ~(gd < 0)
        d = torch.where(bad_dir.unsqueeze(-1), -g, d)
        gd = torch.where(bad_dir, -(g * g).sum(-1), gd)
        t = torch.where(valid[:, 0] & ~bad_dir, torch.ones_like(f), (1.0 / g.norm(dim=-1).clamp_min(1e-12)).clamp_max(1.0))
        # backtracking line search, all problems in lockstep
        accepted = done.clone()
        x_new, f_new, g_new = x.clone(), f.clone(), g.clone()
        for _ in range(max_ls):
            trial = x + t.unsqueeze(-1) * d
            ft, gt = fun(torch.where(accepted.unsqueeze(-1), x, trial))
            n_eval += 1
            ok = torch.isfinite(ft) & torch.isfinite(gt).all(-1) & (ft <= f + c1 * t * gd) & ~accepted
            x_new = torch.where(ok.unsqueeze(-1), trial, x_new)
            f_new = torch.where(ok, ft, f_new)
            g_new = torch.where(ok.unsqueeze(-1), gt, g_new)
            accepted = accepted | ok
            if bool(accepted.all()):
                break
            t = torch.where(accepted, t, 0.5 * t)
        stalled = ~accepted  # line search failed: stop this problem where it is
        s_vec = x_new - x
        y_vec = g_new - g
        sy = (s_vec * y_vec).sum(-1)
        upd = (sy > 1e-10 * y_vec.norm(dim=-1) * s_vec.norm(dim=-1)) & ~done & ~stalled
        S = torch.where(upd[:, None, None], torch.cat([s_vec.unsqueeze(1), S[:, :-1]], 1), S)
        Y = torch.where(upd[:, None, None], torch.cat([y_vec.unsqueeze(1), Y[:, :-1]], 1), Y)
        rho = torch.where(upd[:, None], torch.cat([(1.0 / sy.clamp_min(1e-300)).unsqueeze(1), rho[:, :-1]], 1), rho)
        valid = torch.where(upd[:, None], torch.cat([torch.ones_like(upd).unsqueeze(1), valid[:, :-1]], 1), valid)
        rel = (f - f_new) / torch.maximum(torch.maximum(f.abs(), f_new.abs()), torch.ones_like(f))
        x, f_prev, f, g = x_new, f, f_new, g_new
        done = done | stalled | (g.abs().amax(-1) <= gtol) | ((rel <= ftol) & ~stalled & (it > 1))
        if bool(done.all()):
            break
    return LBFGSResult(x=x, f=f, n_iter=it, n_eval=n_eval, converged=done & ~failed, failed=failed)
