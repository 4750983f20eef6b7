"""A thin Bayesian-optimisation driver around the GPU path -- the caller of the hot path in the
reference is blackboxopt's ``SingleObjectiveBOTorchOptimizer`` (absent here); this module restates the
two things ``scamlgp/optimizer.py`` does with the model:

  report():  rebuild ScaMLGP on all target evaluations, handing the previous model's likelihood and kernel
             modules back in (warm start), and refit weights + hyper-parameters (optimizer.py:156-185)
  generate_evaluation_specification():  maximise the acquisition function (UCB with beta = 9 by
             default, utils.py:215-224) over the unit-cube search space.

The acquisition optimiser follows botorch's ``optimize_acqf`` recipe -- ``raw_samples`` random candidates, ``num_restarts``
initial conditions drawn from them (the best one plus a Boltzmann sample of the rest), box-constrained L-BFGS-B over all
starts jointly, best end point wins -- with exact gradients from the posterior's input-gradient kernels (round 3;
central differences where they do not apply)."""
from __future__ import annotations

from typing import Callable, Dict, Hashable, Optional, Tuple

import math

import numpy as np
import scipy.optimize
import torch

from . import hyper
from .model import ScaMLGP, SourceGP
from .utils import ExpectedImprovement, UpperConfidenceBound, optimize_marginal_likelihood


class GraphedAcquisition:
    """An acquisition function for ONE batch shape, captured once into a HIP graph and replayed per evaluation.

    An evaluation of ``UpperConfidenceBound(model)(X)`` is ~15 launches of libscaml_hip.so (source posteriors with the
    fused covariance block, the weighted task sums, the target GP's assemble / Cholesky / solve / finish) plus a few torch
    element-wise kernels; at the batch sizes of the quasi-Newton phase (num_restarts x (2 dim + 1) points) the GPU work is
    shorter than the Python time to enqueue it.  The library never synchronises and keeps every status on the device, so
    the whole evaluation can be stream-captured: a replay is one host call.  The graph holds the ADDRESSES of the model's
    parameter tensors: build a new one after the model is refitted (ScaMLGPBOLoop.suggest does, per BO step)."""

    def __init__(self, af: Callable[[torch.Tensor], torch.Tensor], batch: int, dim: int, device: torch.device):
        self.af, self.batch = af, batch
        self.x = torch.zeros(batch, dim, dtype=torch.float64, device=device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):          # warm-up off the default stream (allocator state, lazy module load)
            for _ in range(2):
                af(self.x)
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = af(self.x)   # a tensor, or a tuple of tensors (value, gradient)

    def __call__(self, X: torch.Tensor):
        if X.shape[0] != self.batch:
            return self.af(X)
        self.x.copy_(X, non_blocking=True)
        self.graph.replay()
        return tuple(o.clone() for o in self.out) if isinstance(self.out, tuple) else self.out.clone()


def optimize_acqf(af: Callable[[torch.Tensor], torch.Tensor], dim: int, raw_samples: int = 1024, num_restarts: int = 10,
                  max_iter: int = 50, generator: Optional[torch.Generator] = None, fd_step: float = 1e-4,
                  eta: float = 2.0, graph_device: Optional[torch.device] = None, analytic_grad: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Maximise ``af`` over [0, 1]^dim; returns (x_best (dim,), af(x_best)).  botorch's ``optimize_acqf`` recipe:
    ``raw_samples`` random candidates -> ``num_restarts`` initial conditions (the best candidate plus a Boltzmann sample
    of the rest, ``initialize_q_batch``) -> ONE box-constrained L-BFGS-B run over all starts jointly (the summed
    acquisition value, which is separable over the starts: botorch's ``gen_candidates_scipy`` does the same) -> the
    best end point.  Gradients: analytic when the acquisition function offers ``value_and_grad`` and its model the posterior
    input-gradient kernels (``ScaMLGP.posterior_with_grad``: an evaluation then scores just the R starts); otherwise every
    objective evaluation is one batched posterior call over the starts and their central-difference stencils (one-sided at the
    box faces, 2 dim + 1 points per start).  With ``graph_device`` (the model's GPU) the evaluation is captured into a HIP graph
    once and replayed per L-BFGS-B step (``GraphedAcquisition``)."""
    cand = torch.rand(raw_samples, dim, dtype=torch.float64, generator=generator)
    vals = af(cand).detach().cpu()
    R = min(num_restarts, raw_samples)
    best0 = int(vals.argmax())
    picks = [best0]
    if R > 1:
        z = (vals - vals.mean()) / vals.std().clamp_min(1e-12)
        pr = torch.exp(eta * (z - z.max()))
        pr[best0] = 0.0
        if float(pr.sum()) > 0:
            k = min(R - 1, int((pr > 0).sum()))
            picks += torch.multinomial(pr, k, replacement=False, generator=generator).tolist()
    x0 = cand[picks]
    R = x0.shape[0]
    eye = torch.eye(dim, dtype=torch.float64)
    analytic = analytic_grad and hasattr(af, "value_and_grad") and getattr(getattr(af, "model", None), "supports_posterior_grad", lambda: False)()
    if analytic:
        # exact gradients from the posterior's input-gradient kernels: an evaluation scores the R starts, nothing else
        vg = GraphedAcquisition(af.value_and_grad, R, dim, graph_device) if graph_device is not None else af.value_and_grad

        def fun(zv: np.ndarray):
            v, g = vg(torch.from_numpy(zv).reshape(R, dim))
            out = torch.cat([v.reshape(-1), g.reshape(-1)]).detach().cpu()   # one device -> host copy per evaluation
            f = float(out[:R].sum())
            if not math.isfinite(f):
                return float("inf"), np.zeros_like(zv)
            return -f, -torch.nan_to_num(out[R:]).numpy()
    else:
        af_step = GraphedAcquisition(af, R * (2 * dim + 1), dim, graph_device) if graph_device is not None else af

    def fun_fd(zv: np.ndarray):
        x = torch.from_numpy(zv).reshape(R, dim)
        xp = (x.unsqueeze(1) + fd_step * eye).clamp(0.0, 1.0)      # (R, dim, dim): start r shifted along dimension d
        xm = (x.unsqueeze(1) - fd_step * eye).clamp(0.0, 1.0)
        v = af_step(torch.cat([x, xp.reshape(-1, dim), xm.reshape(-1, dim)])).detach().cpu()
        dx = (xp - xm).diagonal(dim1=1, dim2=2)
        g = (v[R:R + R * dim].reshape(R, dim) - v[R + R * dim:].reshape(R, dim)) / dx
        f = float(v[:R].sum())
        if not math.isfinite(f):
            return float("inf"), np.zeros_like(zv)
        return -f, -torch.nan_to_num(g).reshape(-1).numpy()

    res = scipy.optimize.minimize(fun if analytic else fun_fd, x0.reshape(-1).numpy(), jac=True, method="L-BFGS-B", bounds=[(0.0, 1.0)] * (R * dim),
                                  options=dict(maxiter=max_iter))
    xs = torch.from_numpy(np.clip(res.x, 0.0, 1.0)).reshape(R, dim)
    fin = torch.nan_to_num(af(xs).detach().cpu(), nan=-float("inf"))
    j = int(fin.argmax())
    if float(fin[j]) >= float(vals[best0]):
        return xs[j], fin[j]
    return cand[best0], vals[best0]


class ScaMLGPBOLoop:
    def __init__(self, source_gps: Dict[Hashable, SourceGP], dim: int, acquisition: str = "ucb", beta: float = 9.0,
                 num_restarts_log_likelihood: int = 5, raw_samples: int = 1024, num_restarts: int = 10, af_max_iter: int = 50,
                 gp_likelihood: Optional[hyper.GaussianLikelihood] = None, gp_kernel: Optional[hyper.ScaleKernel] = None,
                 seed: Optional[int] = None, use_graph: bool = True):
        self.source_gps, self.dim = source_gps, dim
        self.acquisition, self.beta = acquisition, beta
        self.num_restarts_log_likelihood = num_restarts_log_likelihood
        self.raw_samples, self.num_restarts, self.af_max_iter = raw_samples, num_restarts, af_max_iter
        self.use_graph = use_graph
        self.gen = torch.Generator().manual_seed(0 if seed is None else seed)
        self.X = torch.empty(0, dim, dtype=torch.float64)
        self.Y = torch.empty(0, 1, dtype=torch.float64)
        # scamlgp/optimizer.py:142-148: the model before any evaluation (prior only)
        self.model = ScaMLGP(self.X, self.Y, source_gps, likelihood=gp_likelihood, covar_module=gp_kernel)

    def report(self, x: torch.Tensor, y: float) -> None:
        self.X = torch.cat([self.X, torch.as_tensor(x, dtype=torch.float64).reshape(1, -1)], 0)
        self.Y = torch.cat([self.Y, torch.tensor([[float(y)]], dtype=torch.float64)], 0)
        # scamlgp/optimizer.py:176-185, same call sequence: the fitted modules go back in, the weights restart at 1/T
        self.model = ScaMLGP(
            self.X,
            self.Y,
            self.source_gps,
            likelihood=self.model.likelihood,
            covar_module=self.model.covar_module,
        )
        optimize_marginal_likelihood(self.model, self.num_restarts_log_likelihood)

    def acquisition_function(self) -> Callable[[torch.Tensor], torch.Tensor]:
        if self.acquisition == "ei":
            if self.Y.numel() == 0:
                raise ValueError("EI needs at least one evaluation")
            return ExpectedImprovement(self.model, float(self.Y.min()))
        return UpperConfidenceBound(self.model, self.beta)

    def suggest(self) -> torch.Tensor:
        self.model.eval()
        # (the graph path needs training data on the model: the prior-only model takes the torch branch of posterior())
        dev = self.model.device if (self.use_graph and self.model.n >= 1) else None
        x, _ = optimize_acqf(self.acquisition_function(), self.dim, self.raw_samples, self.num_restarts, self.af_max_iter, self.gen,
                             graph_device=dev)
        return x

    def run(self, objective: Callable[[torch.Tensor], float], n_steps: int):
        for _ in range(n_steps):
            x = self.suggest()
            self.report(x, objective(x))
        return self.X, self.Y
