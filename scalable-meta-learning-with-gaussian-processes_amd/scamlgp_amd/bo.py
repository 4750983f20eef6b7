"""A thin Bayesian-optimisation driver around the GPU path — the caller of the hot path in the
reference is blackboxopt's ``SingleObjectiveBOTorchOptimizer`` (absent here); this module restates the
two things ``scamlgp/optimizer.py`` does with the model:

  report():  rebuild ScaMLGP on all target evaluations, warm-started from the previous kernel
             hyper-parameters, and refit weights + hyper-parameters (optimizer.py:156-185)
  generate_evaluation_specification():  maximise the acquisition function (UCB with beta = 9 by
             default, utils.py:215-224) over the unit-cube search space.

The acquisition optimiser evaluates ALL candidates in one batched posterior call (raw Sobol-free
random candidates, then a few rounds of local Gaussian perturbation around the incumbents) rather
than botorch's multi-start L-BFGS-B, which issues one posterior call per gradient step."""
from __future__ import annotations

from typing import Callable, Dict, Hashable, Optional

import torch

from .model import KernelSpec, ScaMLGP, SourceGP
from .utils import ExpectedImprovement, UpperConfidenceBound, optimize_marginal_likelihood


class ScaMLGPBOLoop:
    def __init__(self, source_gps: Dict[Hashable, SourceGP], dim: int, acquisition: str = "ucb", beta: float = 9.0,
                 num_restarts_log_likelihood: int = 5, raw_samples: int = 1024, refine_rounds: int = 2,
                 covar_module: Optional[KernelSpec] = None, seed: Optional[int] = None):
        self.source_gps, self.dim = source_gps, dim
        self.acquisition, self.beta = acquisition, beta
        self.num_restarts = num_restarts_log_likelihood
        self.raw_samples, self.refine_rounds = raw_samples, refine_rounds
        self.covar_module = covar_module
        self.gen = torch.Generator().manual_seed(0 if seed is None else seed)
        self.X = torch.empty(0, dim, dtype=torch.float64)
        self.Y = torch.empty(0, 1, dtype=torch.float64)
        self.model = ScaMLGP(self.X, self.Y, source_gps, covar_module=covar_module)

    def report(self, x: torch.Tensor, y: float) -> None:
        self.X = torch.cat([self.X, torch.as_tensor(x, dtype=torch.float64).reshape(1, -1)], 0)
        self.Y = torch.cat([self.Y, torch.tensor([[float(y)]], dtype=torch.float64)], 0)
        prev = self.model
        self.model = ScaMLGP(self.X, self.Y, self.source_gps, covar_module=self.covar_module)
        self.model.raw_theta = prev.raw_theta.clone()   # kernel / likelihood are re-used (optimizer.py:180-181);
        optimize_marginal_likelihood(self.model, self.num_restarts)  # the weights restart from 1/T (model.py:319-322)

    def _acquisition(self) -> Callable[[torch.Tensor], torch.Tensor]:
        if self.acquisition == "ei":
            if self.Y.numel() == 0:
                raise ValueError("EI needs at least one evaluation")
            return ExpectedImprovement(self.model, float(self.Y.min()))
        return UpperConfidenceBound(self.model, self.beta)

    def suggest(self) -> torch.Tensor:
        af = self._acquisition()
        cand = torch.rand(self.raw_samples, self.dim, dtype=torch.float64, generator=self.gen)
        vals = af(cand).cpu()
        for r in range(self.refine_rounds):
            top = cand[vals.topk(min(16, len(vals))).indices]
            scale = 0.05 / (r + 1)
            local = (top.repeat_interleave(32, 0) + scale * torch.randn(top.shape[0] * 32, self.dim, dtype=torch.float64,
                                                                          generator=self.gen)).clamp(0.0, 1.0)
            lv = af(local).cpu()
            cand, vals = torch.cat([cand, local]), torch.cat([vals, lv])
        return cand[vals.argmax()]

    def run(self, objective: Callable[[torch.Tensor], float], n_steps: int):
        for _ in range(n_steps):
            x = self.suggest()
            self.report(x, objective(x))
        return self.X, self.Y
