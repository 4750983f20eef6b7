"""Host-side mirror of ``scamlgp/utils.py``: marginal-likelihood fitting with restarts and the
acquisition functions, on top of the batched GPU path.

  optimize_marginal_likelihood   scamlgp/utils.py:139-212  (warm start + ``num_restarts`` prior
                                 samples, keep the best state, ModelFittingError if all failed)
  UpperConfidenceBound           scamlgp/utils.py:215-224  (beta = 9, maximize=False)
  ExpectedImprovement            scamlgp/optimizer.py:96-98 (botorch analytic EI, minimisation)
"""
from __future__ import annotations

import logging
import math
from typing import Dict, Union

import numpy as np
import scipy.optimize
import torch

from . import dist as sdist
from . import hyper
from . import ops
from .model import ScaMLGP, SourceGP, SourceGPStack

logger = logging.getLogger("scamlgp_amd")


class ModelFittingError(RuntimeError):
    """All optimisation attempts failed (mirrors botorch.exceptions.ModelFittingError)."""


class _GraphedBatchObjective:
    """``fun(x) -> (f (B,), g (B, P))`` of the batched L-BFGS (one fused fit + one gradient launch + constraint transform, priors
    and their autograd backward: ~40 launches around ~0.1 ms of GPU work for a small stack) captured once into a HIP graph on a
    static input buffer and replayed per evaluation.  ``ok`` is False if capture is not possible; the caller keeps the eager one."""

    def __init__(self, fun, x0: torch.Tensor):
        self.ok = False
        dev = x0.device
        try:
            self.x = x0.detach().clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):   # (also allocates the factor buffers and the gradient workspace outside the graph's pool)
                    fun(self.x)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.f, self.g = fun(self.x)
            self.ok = True
        except Exception as e:
            logger.warning("stack objective: HIP graph capture failed (%s); evaluating eagerly", e)
            torch.cuda.synchronize(dev)

    def __call__(self, x: torch.Tensor):
        self.x.copy_(x)
        self.graph.replay()
        return self.f.clone(), self.g.clone()


def _fit_stack(stack: SourceGPStack, num_restarts: int, max_iter: int = 200, use_graph: bool = True) -> None:
    """All T tasks x (1 + num_restarts) starts as ONE batch: every L-BFGS iteration is one fused-fit
    launch + one gradient launch over (1 + R) * T problems."""
    T, D = stack.T, stack.D
    reps = 1 + num_restarts
    starts = [stack.raw.clone()]
    for _ in range(num_restarts):
        starts.append(stack.spec.to_raw(stack.spec.sample_prior((T,), D, device=stack.device)))
    x0 = torch.cat(starts, 0)  # problem b = rep * T + task
    fun = lambda r: stack.objective(r, reps)   # noqa: E731
    if use_graph and stack.device.type == "cuda":
        gfun = _GraphedBatchObjective(fun, x0)
        if gfun.ok:
            fun = gfun
    res = hyper.batched_lbfgs(fun, x0, max_iter=max_iter)
    f = torch.where(res.failed | ~torch.isfinite(res.f), torch.full_like(res.f, float("inf")), res.f).reshape(reps, T)
    best = f.argmin(0)
    if bool(torch.isinf(f.min(0).values).any()):
        bad = torch.nonzero(torch.isinf(f.min(0).values)).flatten().tolist()
        raise ModelFittingError(
            "Hyperparameter optimization failed for all attempts. Usually this indicates a problem with model's "
            f"input data or hyperparameter priors definitions. (tasks {[stack.task_ids[i] for i in bad][:8]})")
    n_failed = int(torch.isinf(f).sum())
    if n_failed:
        logger.warning("%d of %d hyper-parameter optimisation attempts failed and were skipped.", n_failed, f.numel())
    stack.raw = res.x.reshape(reps, T, D + 2)[best, torch.arange(T, device=stack.device)].contiguous()
    stack.refresh()
    obj = -f.min(0).values
    # (a sharded stack also reports the objective summed over every rank's tasks: the fit's one collective)
    total = sdist.fused_allreduce([obj.sum().reshape(1)], stack.shard)[0]
    stack.last_fit_info = dict(n_iter=res.n_iter, n_eval=res.n_eval, objective=obj, objective_sum=total.squeeze(0))


class _GraphedObjective:
    """-mll(z) and its gradient for z = [raw_theta || raw_weights] of a ScaMLGP, captured once (torch.cuda.graph: autograd's
    backward included) and replayed.  ``ok`` is False if capture is not possible on this build; the caller then evaluates eagerly."""

    def __init__(self, model: ScaMLGP, D2: int):
        self.ok = False
        dev = model.device
        try:
            self.z = torch.zeros(D2 + model.T, dtype=torch.float64, device=dev, requires_grad=True)
            self.z.data.copy_(torch.cat([model.raw_theta, model.raw_weights]))
            self.host = torch.empty(1 + D2 + model.T, dtype=torch.float64).pin_memory()

            def body():
                val = -model.mll(self.z[:D2], self.z[D2:])
                (g,) = torch.autograd.grad(val, self.z)
                return torch.cat([val.detach().reshape(1), g])

            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    body()
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = body()
            self.ok = True
        except Exception as e:  # capture unsupported for some op on this build: eager evaluation is always available
            logger.warning("target objective: HIP graph capture failed (%s); evaluating eagerly", e)
            torch.cuda.synchronize(dev)

    def __call__(self, z: np.ndarray) -> np.ndarray:
        self.z.data.copy_(torch.from_numpy(z), non_blocking=False)
        self.graph.replay()
        self.host.copy_(self.out, non_blocking=False)
        return self.host.numpy()


def _fit_target_scipy(model: ScaMLGP, starts: torch.Tensor, maxiter: int, use_graph: bool):
    """The refit where the library's target-fit kernel does not take the shape (n beyond its LDS, D > 16): scipy L-BFGS-B per
    start point, objective and gradient from torch autograd on the device (``ScaMLGP.mll``), replayed from a HIP graph.
    Returns (z (B, P), f (B,)) with f = -mll at the end points (inf for a failed run)."""
    D2, T = model.raw_theta.numel(), model.T
    bounds = [(None, None)] * D2 + [(model.weights_lower_bound, None)] * T

    def fun_eager(z: np.ndarray):
        zt = torch.tensor(z, dtype=torch.float64, device=model.device, requires_grad=True)
        try:
            val = -model.mll(zt[:D2], zt[D2:])
            (g,) = torch.autograd.grad(val, zt)
        except RuntimeError:  # Cholesky failure somewhere in the line search
            return float("inf"), np.zeros_like(z)
        v = float(val.detach())
        if not math.isfinite(v):
            return float("inf"), np.zeros_like(z)
        return v, g.cpu().numpy()

    # ~100 small launches per evaluation: forward AND backward are captured once into a HIP graph on static buffers and replayed per
    # evaluation; one device -> host copy of [value || gradient] per evaluation is the only synchronisation.
    graphed = _GraphedObjective(model, D2) if use_graph and model.device.type == "cuda" else None

    def fun(z: np.ndarray):
        if graphed is None or not graphed.ok:
            return fun_eager(z)
        out = graphed(z)
        v = float(out[0])
        if not math.isfinite(v):
            return float("inf"), np.zeros_like(z)
        return v, out[1:].copy()

    zs, fs = [], []
    for z0 in starts.cpu().numpy():
        r = scipy.optimize.minimize(fun, z0, jac=True, method="L-BFGS-B", bounds=bounds, options=dict(maxiter=maxiter))
        zs.append(torch.from_numpy(np.asarray(r.x, dtype=np.float64)))
        fs.append(float(r.fun))
    return torch.stack(zs), torch.tensor(fs, dtype=torch.float64)


def _fit_target(model: ScaMLGP, num_restarts: int, maxiter: int = 200, use_graph: bool = True, use_kernel: bool = True) -> None:
    """Target GP: weights + kernel hyper-parameters (scamlgp/optimizer.py:176-185 -> scamlgp/utils.py:139-212).  The warm start
    and the ``num_restarts`` prior-sampled starts are ONE batch; ``scaml_target_fit_f64`` runs all their L-BFGS optimisations
    (box bound w >= 1e-10, scamlgp/model.py:334) to convergence in one launch on the device; the best end point is kept.
    On a sharded model rank 0 fits and broadcasts the result: every rank must hold the same weights and hyper-parameters (the
    weighted sums the ranks all-reduce in ``_source_prior`` are built from them)."""
    if model.n == 0:
        return
    shard = model._shard
    D2, T = model.raw_theta.numel(), model.T
    best = torch.zeros(D2 + T + 1, dtype=torch.float64, device=model.device)   # [state || ok]
    if shard is None or shard.rank == 0:
        starts = [torch.cat([model.raw_theta, model.raw_weights])]
        for _ in range(num_restarts):
            th = model.spec.sample_prior((), D2 - 2, device=model.device)
            w = model.weights_prior.sample((T,), device=model.device).clamp_min(model.weights_lower_bound)
            starts.append(torch.cat([model.spec.to_raw(th), w]))
        z0 = torch.stack(starts)
        prob = model.target_problem() if (use_kernel and model.device.type == "cuda") else None
        if prob is not None:
            res = ops.target_fit(prob, z0, max_iter=maxiter)
            z, f = res["z"], -res["value"]
            f = torch.where(torch.isfinite(f) & (res["stats"][:, 2] != 4), f, torch.full_like(f, float("inf")))
            model.last_fit_info = dict(stats=res["stats"], objective=res["value"])
        else:
            z, f = _fit_target_scipy(model, z0, maxiter, use_graph)
            z, f = z.to(model.device), f.to(model.device)
        n_failed = int(torch.isinf(f).sum())     # (the one host synchronisation of the refit)
        if n_failed and n_failed < f.numel():
            logger.warning("Error occurred while optimizing the model hyperparameters; %d restart(s) will be skipped.", n_failed)
        if n_failed < f.numel():
            best[:-1] = z[int(f.argmin())]
            best[-1] = 1.0
    if shard is not None and shard.world > 1:
        import torch.distributed as dist

        dist.broadcast(best, src=dist.get_global_rank(shard.group, 0) if shard.group is not None else 0, group=shard.group)
    if float(best[-1]) != 1.0:
        raise ModelFittingError("Hyperparameter optimization failed for all attempts. Usually this indicates a problem with "
                                "model's input data or hyperparameter priors definitions.")
    model.load_state_dict({"raw_theta": best[:D2].clone(), "raw_weights": best[D2:-1].clone()})


def optimize_marginal_likelihood(model: Union[SourceGPStack, ScaMLGP, Dict, SourceGP], num_restarts: int = 0, **fit_options):
    """Refit the model's hyper-parameters by maximising the marginal log likelihood (+ priors) on its
    training data: once warm-started from the current parameters, ``num_restarts`` times from prior
    samples; the best state is kept (scamlgp/utils.py:139-212)."""
    if isinstance(model, dict):
        model = list(model.values())[0]
    if isinstance(model, SourceGP):
        model = model._stack
    if isinstance(model, SourceGPStack):
        return _fit_stack(model, num_restarts, **fit_options)
    if isinstance(model, ScaMLGP):
        return _fit_target(model, num_restarts, **fit_options)
    raise TypeError(f"cannot fit a {type(model).__name__}")


# --- acquisition functions (to be MAXIMISED, for minimising the objective) -------------------------
def _single_q(X: torch.Tensor) -> torch.Tensor:
    """botorch's analytic acquisition functions take ``batch_shape x 1 x d`` (t_batch_mode_transform(expected_q=1)) and return
    ``batch_shape`` values; a q > 1 batch is an error there and here.  X (M, D) -- this package's flat list of M points -- passes."""
    if X.dim() > 2 and X.shape[-2] != 1:
        raise ValueError(f"analytic acquisition functions take q = 1 (X of shape batch_shape x 1 x d), got q = {X.shape[-2]}")
    return X


def _drop_q(X: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    return v.squeeze(-1) if X.dim() > 2 else v



class UpperConfidenceBound:
    """botorch UpperConfidenceBound(model, beta=9.0, maximize=False) (scamlgp/utils.py:215-224):
    value = -mu + sqrt(beta * var) per query point; X (M, D) -> (M,)."""

    def __init__(self, model: ScaMLGP, beta: float = 9.0):
        self.model, self.beta = model, beta

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        mvn = self.model.posterior(_single_q(X)).mvn
        return _drop_q(X, -mvn.mean + torch.sqrt(self.beta * mvn.variance.clamp_min(0.0)))

    def value_and_grad(self, X: torch.Tensor):
        """(value (M,), d value / d X (M, D)) from the model's analytic posterior gradients (ScaMLGP.posterior_with_grad)."""
        mu, var, dmu, dvar = self.model.posterior_with_grad(X)
        sd = torch.sqrt(self.beta * var.clamp_min(0.0))
        # d sqrt(beta var) = beta dvar / (2 sqrt(beta var)); zero where the variance is clamped
        coef = torch.where(var > 0, 0.5 * self.beta / sd.clamp_min(1e-300), torch.zeros_like(sd))
        return -mu + sd, -dmu + coef.unsqueeze(-1) * dvar


class ExpectedImprovement:
    """botorch analytic ExpectedImprovement(model, best_f, maximize=False) (scamlgp/optimizer.py:96-98):
    sigma = sqrt(max(var, 1e-9)), u = -(mu - best_f) / sigma, EI = sigma (phi(u) + u Phi(u))."""

    def __init__(self, model: ScaMLGP, best_f: float):
        self.model, self.best_f = model, best_f

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        mvn = self.model.posterior(_single_q(X)).mvn
        sigma = mvn.variance.clamp_min(1e-9).sqrt()
        u = -(mvn.mean - self.best_f) / sigma
        pdf = torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)
        cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
        return _drop_q(X, sigma * (pdf + u * cdf))

    def value_and_grad(self, X: torch.Tensor):
        """(EI (M,), d EI / d X (M, D)): d EI = -Phi(u) d mu + phi(u) d sigma, d sigma = d var / (2 sigma) (zero where the variance
        sits on the 1e-9 floor), from the model's analytic posterior gradients."""
        mu, var, dmu, dvar = self.model.posterior_with_grad(X)
        sigma = var.clamp_min(1e-9).sqrt()
        u = -(mu - self.best_f) / sigma
        pdf = torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)
        cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
        dsig = torch.where(var > 1e-9, 0.5 / sigma, torch.zeros_like(sigma)).unsqueeze(-1) * dvar
        return sigma * (pdf + u * cdf), -cdf.unsqueeze(-1) * dmu + pdf.unsqueeze(-1) * dsig
