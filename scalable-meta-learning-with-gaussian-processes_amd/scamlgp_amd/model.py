"""Host-side mirror of ``scamlgp/model.py`` on top of the HIP hot path.

Same public names and argument meaning as the reference (``meta_fit_scamlgp``, ``ScaMLGP``,
``_compute_target_prior``, ``significant_weights_mask``), but the stack of source GPs is ONE batched
object: every source-side operation (marginal likelihood + gradient for all tasks x restarts,
Cholesky factors, posteriors at shared query points, the weighted prior sum) is a launch of
libscaml_hip.so over the whole ``(T, N, D)`` stack instead of a Python loop over gpytorch models
(``scamlgp/model.py:128, 176-188, 281``).  The target GP's own algebra (n <= ~80 points:
``scamlgp/model.py:359-384`` and gpytorch's exact prediction) stays in torch on the same device.

Deliberate deviations: tasks are fitted simultaneously from the same initial
values, so the reference's sequential warm-start chain (task t starts from task t-1's optimum,
``scamlgp/model.py:177-178``) is not reproduced; restarts are ranked by the training objective, not
by the eval-mode value the reference happens to compute (``scamlgp/utils.py:176-177``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Hashable, List, Optional, Sequence, Tuple

import torch

from . import hyper, ops
from ._lib import KIND_MATERN52, KIND_RBF

_LOG_2PI = math.log(2.0 * math.pi)


# ---------------------------------------------------------------------------------------
# data containers
# ---------------------------------------------------------------------------------------
class _CallableTensor(torch.Tensor):
    """The reference reads datasets both as ``data.X.shape`` (scamlgp/utils.py:117) and as
    ``data.X()`` (scamlgp/model.py:180): a tensor that returns itself when called serves both."""

    def __call__(self):
        return self.as_subclass(torch.Tensor)


class SupervisedDataset:
    """Minimal stand-in for botorch.utils.datasets.SupervisedDataset: X (n, d), Y (n, 1)."""

    def __init__(self, X: torch.Tensor, Y: torch.Tensor):
        self.X = torch.as_tensor(X).as_subclass(_CallableTensor)
        self.Y = torch.as_tensor(Y).as_subclass(_CallableTensor)


def validate_meta_data(meta_data: Dict[Hashable, SupervisedDataset]) -> None:
    """scamlgp/utils.py:112-136 (same checks, same messages)."""
    if len(meta_data) == 0:
        raise ValueError("Empty meta data. Needs at least one source task.")
    task_id_source_0, data_source_0 = list(meta_data.items())[0]
    X_shape, Y_shape = data_source_0.X.shape, data_source_0.Y.shape
    if X_shape[:-2] != Y_shape[:-2]:
        raise ValueError(f"The X and Y batch sizes of task {task_id_source_0} are not equal.")
    for task_id, task_data in meta_data.items():
        if (task_data.X.shape[:-2] != X_shape[:-2] or task_data.Y.shape[:-2] != Y_shape[:-2]
                or task_data.X.shape[-1] != X_shape[-1]):
            raise ValueError(f"Dimensions of tasks {task_id_source_0} and {task_id} do not match.")
        if task_data.Y.shape[-1] != 1:
            raise ValueError(f"The output dimension of task {task_id} is {task_data.Y.shape[-1]} but must be one")


@dataclass(frozen=True)
class KernelSpec:
    """What the reference passes as ``covar_module``: ScaleKernel(RBFKernel | MaternKernel(2.5), ARD)."""
    kind: int = KIND_RBF


def standardize_fit(Y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """botorch Standardize(m=1): mean / unbiased std over dim -2, std < 1e-8 (or n = 1) -> 1."""
    m = Y.mean(dim=-2)
    if Y.shape[-2] < 2:
        s = torch.ones_like(m)
    else:
        s = Y.std(dim=-2)
        s = torch.where(s >= 1e-8, s, torch.ones_like(s))
    return m, s


class _OutcomeTransform:
    def __init__(self, mean: torch.Tensor, std: torch.Tensor):
        self.means, self.stdvs = mean.reshape(1, 1), std.reshape(1, 1)

    def untransform(self, Y: torch.Tensor, Yvar: Optional[torch.Tensor] = None):
        return self.means + self.stdvs * Y, None if Yvar is None else self.stdvs ** 2 * Yvar


class _MVN:
    def __init__(self, mean: torch.Tensor, cov: torch.Tensor):
        self.mean, self.covariance_matrix, self.lazy_covariance_matrix = mean, cov, cov

    @property
    def variance(self):
        return torch.diagonal(self.covariance_matrix, dim1=-2, dim2=-1)


class _Posterior:
    def __init__(self, mean: torch.Tensor, cov: torch.Tensor):
        self.mvn = _MVN(mean, cov)
        self.mean, self.variance = mean.unsqueeze(-1), self.mvn.variance.unsqueeze(-1)


# ---------------------------------------------------------------------------------------
# the stack of source GPs
# ---------------------------------------------------------------------------------------
class SourceGPStack:
    """All source GPs as one padded ``(T, N, D)`` problem on the device.

    Holds the data (per-task standardised targets, botorch ``Standardize`` semantics,
    scamlgp/model.py:185), the raw hyper-parameters ``(T, D+2)`` and, after ``refresh()``, the
    Cholesky factors / alpha / block inverses produced by the fused fit kernel."""

    def __init__(self, task_ids: Sequence[Hashable], X: Sequence[torch.Tensor], Y: Sequence[torch.Tensor],
                 kind: int = KIND_RBF, spec: Optional[hyper.HyperSpec] = None, device: Optional[torch.device] = None):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.task_ids = list(task_ids)
        self.kind = int(kind)
        self.spec = spec or hyper.source_gp_spec()
        T = len(self.task_ids)
        self.D = int(X[0].shape[-1])
        ns = [int(x.shape[-2]) for x in X]
        self.N = max(ns)
        self.ragged = min(ns) != self.N
        Xp = torch.zeros(T, self.N, self.D, dtype=torch.float64)
        yp = torch.zeros(T, self.N, dtype=torch.float64)
        means, stds = torch.zeros(T, dtype=torch.float64), torch.ones(T, dtype=torch.float64)
        for t, (x, y) in enumerate(zip(X, Y)):
            y = torch.as_tensor(y, dtype=torch.float64).reshape(-1, 1).cpu()
            m, s = standardize_fit(y)
            means[t], stds[t] = m.squeeze(), s.squeeze()
            Xp[t, :ns[t]] = torch.as_tensor(x, dtype=torch.float64).cpu()
            yp[t, :ns[t]] = ((y - m) / s).squeeze(-1)
        self.X = Xp.to(self.device)
        self.y = yp.to(self.device)
        self.y_mean, self.y_std = means.to(self.device), stds.to(self.device)
        self.n_list = ns
        self.n_points = torch.tensor(ns, dtype=torch.int32, device=self.device) if self.ragged else None
        self.n_float = torch.tensor(ns, dtype=torch.float64, device=self.device)
        self.raw = self.spec.to_raw(self.spec.init_theta(self.D, device=self.device)).repeat(T, 1)
        self._fit = None

    # -- hyper-parameters -----------------------------------------------------------------
    @property
    def T(self) -> int:
        return len(self.task_ids)

    @property
    def theta(self) -> torch.Tensor:
        return self.spec.to_theta(self.raw)

    def set_theta(self, theta: torch.Tensor) -> None:
        self.raw = self.spec.to_raw(theta.to(self.device, torch.float64))
        self._fit = None

    # -- fused fit ------------------------------------------------------------------------
    def refresh(self) -> dict:
        """Factor all tasks at the current hyper-parameters (one launch) and cache L, alpha, W."""
        fit = ops.gp_fit_fused(self.X, self.y, self.theta, self.kind, n_points=self.n_points, want_linv=True)
        ops.raise_if_not_psd(fit["info"])
        # the explicit inverse factor: computed once per fit, it makes every later posterior a matrix product
        # (the role of gpytorch's prediction-strategy caches; scamlgp/model.py:128, :281 query fixed source GPs)
        fit["Linv"] = ops.linv_batched(fit["L"], fit["Linv_diag"], n_points=self.n_points)
        self._fit = fit
        return fit

    @property
    def fit(self) -> dict:
        return self._fit if self._fit is not None else self.refresh()

    def objective(self, raw: torch.Tensor, reps: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
        """Negative training objective and its gradient w.r.t. the raw parameters for ``reps``
        hyper-parameter sets per task: raw (reps * T, D+2), problem b belongs to task b % T.
        objective = -(mll + sum log p(theta) / n)   (gpytorch ExactMarginalLogLikelihood with priors,
        A5 of SURVEY.md).  One fused-fit launch + one gradient launch for everything."""
        X = self.X.repeat(reps, 1, 1) if reps > 1 else self.X
        y = self.y.repeat(reps, 1) if reps > 1 else self.y
        npts = None if self.n_points is None else self.n_points.repeat(reps)
        nflt = self.n_float.repeat(reps)
        theta = self.spec.to_theta(raw)
        fit = ops.gp_fit_fused(X, y, theta, self.kind, n_points=npts, want_linv=True, zero_upper=False, retry=True)
        g_theta = ops.mll_backward(X, theta, self.kind, fit["L"], fit["Linv_diag"], fit["alpha"], n_points=npts)
        f = -(fit["mll"] + self.spec.log_prior(theta) / nflt)
        g = -(g_theta + self.spec.dlog_prior(theta) / nflt.unsqueeze(-1)) * self.spec.dtheta_draw(raw)
        bad = fit["info"] > 0
        f = torch.where(bad, torch.full_like(f, float("nan")), f)
        return f, g

    # -- posteriors -----------------------------------------------------------------------
    def posterior(self, xq: torch.Tensor, cov_first: int = 0, want_var: bool = True) -> dict:
        """Un-standardised posteriors of all tasks at xq (M, D): mean (T, M), var (T, M),
        cov (T, cov_first, M)."""
        f = self.fit
        return ops.source_posteriors(xq.to(self.device, torch.float64), self.X, self.theta, self.kind, f["L"], f["Linv_diag"],
                                     f["alpha"], self.y_mean, self.y_std, n_points=self.n_points, want_var=want_var,
                                     cov_first=cov_first, Linv=f.get("Linv"))

    def raw_targets(self) -> torch.Tensor:
        """All source observations in original units, concatenated (scamlgp/model.py:264-270)."""
        out = []
        for t, n in enumerate(self.n_list):
            out.append(self.y_mean[t] + self.y_std[t] * self.y[t, :n])
        return torch.cat(out).unsqueeze(-1)


class SourceGP:
    """View of one task of a SourceGPStack with the attributes the reference touches on a
    SingleTaskGP: ``posterior(x).mvn``, ``outcome_transform.stdvs / untransform``, ``train_targets``."""

    def __init__(self, stack: SourceGPStack, index: int):
        self._stack, self._index = stack, index

    @property
    def outcome_transform(self) -> _OutcomeTransform:
        return _OutcomeTransform(self._stack.y_mean[self._index], self._stack.y_std[self._index])

    @property
    def train_targets(self) -> torch.Tensor:
        return self._stack.y[self._index, : self._stack.n_list[self._index]]

    @property
    def train_inputs(self):
        return (self._stack.X[self._index, : self._stack.n_list[self._index]],)

    def posterior(self, x: torch.Tensor) -> _Posterior:
        x2 = x.reshape(-1, x.shape[-1])
        M = x2.shape[0]
        p = self._stack.posterior(x2, cov_first=M)
        return _Posterior(p["mean"][self._index], p["cov"][self._index])


def _stack_of(source_gps: Sequence[SourceGP]) -> Tuple[SourceGPStack, List[int]]:
    stacks = {id(g._stack) for g in source_gps}
    if len(stacks) != 1:
        raise ValueError("all source GPs must come from one meta_fit_scamlgp call (one device-resident stack)")
    return source_gps[0]._stack, [g._index for g in source_gps]


# ---------------------------------------------------------------------------------------
# reference API
# ---------------------------------------------------------------------------------------
def meta_fit_scamlgp(
    meta_data: Dict[Hashable, SupervisedDataset],
    likelihood: Optional[hyper.HyperSpec] = None,
    covar_module: Optional[KernelSpec] = None,
    num_restarts_log_likelihood: int = 5,
    seed: Optional[int] = None,
    device: Optional[torch.device] = None,
) -> Dict[Hashable, SourceGP]:
    """Train the source GPs on the given meta-data (scamlgp/model.py:138-189).

    ``likelihood`` may carry a full HyperSpec (constraints / priors / inits) to override the
    reference defaults, ``covar_module`` a KernelSpec choosing RBF (default, as in the reference)
    or Matern-5/2.  All tasks and all restarts are optimised together on the GPU."""
    from .utils import optimize_marginal_likelihood

    if seed is not None:
        torch.manual_seed(seed=seed)
    validate_meta_data(meta_data)
    first = list(meta_data.values())[0]
    if first.X.dim() != 2:
        raise ValueError("batched (batch_shape x n x d) meta-data is not supported by the stacked GPU path")
    kind = (covar_module or KernelSpec()).kind
    stack = SourceGPStack(list(meta_data.keys()), [d.X() for d in meta_data.values()], [d.Y() for d in meta_data.values()],
                          kind=kind, spec=likelihood, device=device)
    optimize_marginal_likelihood(stack, num_restarts=num_restarts_log_likelihood)
    return {tid: SourceGP(stack, i) for i, tid in enumerate(stack.task_ids)}


def significant_weights_mask(weights: torch.Tensor, std_Y_vals: torch.Tensor, threshold: float) -> torch.Tensor:
    """scamlgp/model.py:192-215: w_i sigma_i n_w / sum_j w_j sigma_j >= threshold."""
    num_weights = len(weights)
    w_times_sigma = weights * std_Y_vals
    norm_weights = w_times_sigma * num_weights / w_times_sigma.sum()
    return norm_weights >= threshold


def _compute_target_prior(x: torch.Tensor, source_gps: List[SourceGP], weights: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """scamlgp/model.py:108-135: mean = sum_i w_i mu_i(x) as (n, 1), cov = sum_i w_i^2 Sigma_i(x, x) (n, n),
    in original units.  One batched posterior launch + two weighted task sums."""
    if len(source_gps) != len(weights):
        raise ValueError(f"The number of source GPs, {len(source_gps)}, does not equal the number of weights, {len(weights)}")
    stack, idx = _stack_of(source_gps)
    x2 = x.reshape(-1, x.shape[-1]).to(stack.device, torch.float64)
    M = x2.shape[0]
    p = stack.posterior(x2, cov_first=M)
    w_full = torch.zeros(stack.T, dtype=torch.float64, device=stack.device)
    w_full[idx] = weights.to(stack.device, torch.float64)
    active = torch.zeros(stack.T, dtype=torch.bool, device=stack.device)
    active[idx] = True
    mean = ops.weighted_task_sum(p["mean"], w_full, 1, active)
    cov = ops.weighted_task_sum(p["cov"], w_full, 2, active)
    return mean.unsqueeze(-1), cov


def _kernel_torch(x1: torch.Tensor, x2: torch.Tensor, theta: torch.Tensor, kind: int) -> torch.Tensor:
    """os * k(x1 / l, x2 / l) in torch (target GP only: n <= ~80 rows)."""
    D = x1.shape[-1]
    a, b = x1 / theta[:D], x2 / theta[:D]
    d2 = (a.unsqueeze(-2) - b.unsqueeze(-3)).pow(2).sum(-1)
    if kind == KIND_RBF:
        k = torch.exp(-0.5 * d2)
    else:
        r = torch.sqrt(d2.clamp_min(1e-30))
        k = (1.0 + math.sqrt(5.0) * r + (5.0 / 3.0) * d2) * torch.exp(-math.sqrt(5.0) * r)
    return theta[D] * k


class ScaMLGP:
    """Scalable meta-learning GP (scamlgp/model.py:218-384): target prior
    N(sum_i w_i mu_i, sum_i w_i^2 Sigma_i + k_t) over the posteriors of the source stack."""

    def __init__(self, train_X: torch.Tensor, train_Y: torch.Tensor, source_gps: Dict[Hashable, SourceGP],
                 likelihood: Optional[hyper.HyperSpec] = None, covar_module: Optional[KernelSpec] = None,
                 weight_pruning_threshold: float = 1e-3) -> None:
        self._weight_pruning_threshold = weight_pruning_threshold
        self.source_gps = source_gps
        gps = list(source_gps.values())
        self._stack, self._idx = _stack_of(gps)
        dev = self._stack.device
        self.device = dev
        self.train_X = torch.as_tensor(train_X, dtype=torch.float64).reshape(-1, self._stack.D).to(dev)
        self.train_Y = torch.as_tensor(train_Y, dtype=torch.float64).reshape(-1, 1).to(dev)
        self.n, self.T = self.train_X.shape[0], len(gps)
        self.kind = (covar_module or KernelSpec()).kind
        self.spec = likelihood or hyper.target_gp_spec()
        # standardise w.r.t. ALL meta + target observations (scamlgp/model.py:264-276)
        Y_all = torch.cat([self._stack.raw_targets(), self.train_Y], dim=-2)
        self.has_transform = self.train_Y.numel() > 0
        m, s = standardize_fit(Y_all)
        self.m_all, self.s_all = (m.squeeze(), s.squeeze()) if self.has_transform else (Y_all.new_zeros(()), Y_all.new_ones(()))
        self.train_targets = ((self.train_Y - self.m_all) / self.s_all).squeeze(-1)
        # cached source posteriors at the target inputs, all tasks, no pruning (scamlgp/model.py:279-289)
        if self.n > 0:
            p = self._stack.posterior(self.train_X, cov_first=self.n)
            self.source_means = p["mean"][self._idx].transpose(0, 1).contiguous()          # (n, T)
            self.source_covs = p["cov"][self._idx].permute(1, 2, 0).contiguous()           # (n, n, T)
        self.raw_theta = self.spec.to_raw(self.spec.init_theta(self._stack.D, device=dev))
        self.raw_weights = torch.full((self.T,), 1.0 / self.T, dtype=torch.float64, device=dev)
        self.weights_prior = hyper.GammaPrior(1.0, 1.0)
        self.weights_lower_bound = 1e-10   # GreaterThan(1e-10, transform=None): a box bound for the optimiser
        self.training = True

    # -- parameters -------------------------------------------------------------------------
    @property
    def weights(self) -> torch.Tensor:
        return self.raw_weights

    @weights.setter
    def weights(self, value) -> None:
        self.raw_weights = torch.as_tensor(value, dtype=torch.float64).to(self.device)

    @property
    def theta(self) -> torch.Tensor:
        return self.spec.to_theta(self.raw_theta)

    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    def state_dict(self) -> dict:
        return {"raw_theta": self.raw_theta.clone(), "raw_weights": self.raw_weights.clone()}

    def load_state_dict(self, sd: dict) -> None:
        self.raw_theta, self.raw_weights = sd["raw_theta"].clone(), sd["raw_weights"].clone()

    # -- model ------------------------------------------------------------------------------
    def _std_source_stds(self) -> torch.Tensor:
        return self._stack.y_std[self._idx]

    def forward(self, x: torch.Tensor) -> _MVN:
        """scamlgp/model.py:359-384.  Training: cached source terms at train_X; eval: pruned weights and a
        fresh batched source posterior at x.  Both in the standardised target space, plus k_t(x, x)."""
        x = torch.as_tensor(x, dtype=torch.float64).reshape(-1, self._stack.D).to(self.device)
        w = self.weights
        if self.training:
            mean = self.source_means @ w
            cov = self.source_covs @ w ** 2
        else:
            mask = significant_weights_mask(w, self._std_source_stds(), self._weight_pruning_threshold)
            gps = [g for g, keep in zip(self.source_gps.values(), mask.tolist()) if keep]
            mean, cov = _compute_target_prior(x, gps, w[mask])
            mean = mean.squeeze(-1)
        if self.has_transform:
            mean = (mean - self.m_all) / self.s_all
            cov = cov / self.s_all ** 2
        cov = cov + _kernel_torch(x, x, self.theta, self.kind)
        return _MVN(mean, cov)

    def mll(self, raw_theta: Optional[torch.Tensor] = None, raw_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Training objective (A9): [log N(y~ | mean, cov + sigma^2 I) + log priors] / n, differentiable in torch."""
        rt = self.raw_theta if raw_theta is None else raw_theta
        w = self.raw_weights if raw_weights is None else raw_weights
        theta = self.spec.to_theta(rt)
        mean = (self.source_means @ w - self.m_all) / self.s_all
        cov = (self.source_covs @ w ** 2) / self.s_all ** 2 + _kernel_torch(self.train_X, self.train_X, theta, self.kind)
        cov = cov + theta[-1] * torch.eye(self.n, dtype=torch.float64, device=self.device)
        Lc = torch.linalg.cholesky(cov)
        v = torch.linalg.solve_triangular(Lc, (self.train_targets - mean).unsqueeze(-1), upper=False)
        val = -0.5 * ((v * v).sum() + 2.0 * torch.log(torch.diagonal(Lc)).sum() + self.n * _LOG_2PI)
        val = val + self.spec.log_prior(theta) + self.weights_prior.log_prob(w).sum()
        return val / self.n

    def posterior(self, X: torch.Tensor, observation_noise: bool = False):
        """Target posterior at X (M, D) in original units (A10): mean (M,), variance (M,).  The source
        prior is evaluated ONCE at cat(train_X, X) — train block, cross block and query diagonal — instead
        of once per query as the reference does."""
        Xq = torch.as_tensor(X, dtype=torch.float64).reshape(-1, self._stack.D).to(self.device)
        M, n = Xq.shape[0], self.n
        w = self.weights
        mask = significant_weights_mask(w, self._std_source_stds(), self._weight_pruning_threshold)
        w_full = torch.zeros(self._stack.T, dtype=torch.float64, device=self.device)
        w_full[self._idx] = w
        active = torch.zeros(self._stack.T, dtype=torch.bool, device=self.device)
        active[torch.as_tensor(self._idx, device=self.device)[mask]] = True
        xall = torch.cat([self.train_X, Xq], 0)
        p = self._stack.posterior(xall, cov_first=n)
        mu_s = ops.weighted_task_sum(p["mean"], w_full, 1, active)
        var_s = ops.weighted_task_sum(p["var"], w_full, 2, active)
        theta = self.theta
        mean = (mu_s - self.m_all) / self.s_all
        var_q = var_s[n:] / self.s_all ** 2 + theta[-2]
        if n == 0:
            mu, var = mean, var_q
        else:
            cov_s = ops.weighted_task_sum(p["cov"], w_full, 2, active) / self.s_all ** 2
            Knn = cov_s[:, :n] + _kernel_torch(self.train_X, self.train_X, theta, self.kind)
            Knn = Knn + theta[-1] * torch.eye(n, dtype=torch.float64, device=self.device)
            Knq = cov_s[:, n:] + _kernel_torch(self.train_X, Xq, theta, self.kind)
            Lc = torch.linalg.cholesky(Knn)
            resid = (self.train_targets - mean[:n]).unsqueeze(-1)
            a = torch.cholesky_solve(resid, Lc).squeeze(-1)
            mu = mean[n:] + Knq.transpose(0, 1) @ a
            Vq = torch.linalg.solve_triangular(Lc, Knq, upper=False)
            var = var_q - (Vq * Vq).sum(0)
        if observation_noise:
            var = var + theta[-1]
        return TargetPosterior(self.m_all + self.s_all * mu, self.s_all ** 2 * var)


@dataclass
class TargetPosterior:
    mean: torch.Tensor      # (M,)
    variance: torch.Tensor  # (M,)
