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
import warnings
from dataclasses import dataclass
from typing import Dict, Hashable, List, Optional, Sequence, Tuple

import torch

from . import dist as sdist
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
    """The checks of scamlgp/utils.py:112-136 with the same ValueError messages (callers match on them): at least
    one task; all tasks share batch shape and input dimension with the first one; one output column."""
    if not meta_data:
        raise ValueError("Empty meta data. Needs at least one source task.")
    tasks = iter(meta_data.items())
    first_id, first = next(tasks)
    batch_x, batch_y, dim = first.X.shape[:-2], first.Y.shape[:-2], first.X.shape[-1]
    if batch_x != batch_y:
        raise ValueError(f"The X and Y batch sizes of task {first_id} are not equal.")
    for tid, data in [(first_id, first), *tasks]:
        same = data.X.shape[:-2] == batch_x and data.Y.shape[:-2] == batch_y and data.X.shape[-1] == dim
        if not same:
            raise ValueError(f"Dimensions of tasks {first_id} and {tid} do not match.")
        n_out = data.Y.shape[-1]
        if n_out != 1:
            raise ValueError(f"The output dimension of task {tid} is {n_out} but must be one")


@dataclass(frozen=True)
class KernelSpec:
    """Shorthand for ``covar_module``: only the kernel family (RBF | Matern-5/2, ARD), reference priors."""
    kind: int = KIND_RBF


def _resolve_modules(likelihood, covar_module, D: int, default_kernel):
    """What the reference accepts as ``likelihood`` / ``covar_module`` (gpytorch modules, scamlgp/model.py:140-141,
    223-224) -> (GaussianLikelihood, ScaleKernel) holders of this package.  Also accepted: a HyperSpec as
    ``likelihood`` (all three parameter groups at once) and a KernelSpec as ``covar_module`` (family only)."""
    kind = covar_module.kind if isinstance(covar_module, (KernelSpec, hyper.ScaleKernel)) else KIND_RBF
    if isinstance(likelihood, hyper.HyperSpec):
        return hyper.modules_from_spec(likelihood, kind, D)
    if likelihood is not None and not isinstance(likelihood, hyper.GaussianLikelihood):
        raise TypeError(f"likelihood must be a GaussianLikelihood (or a HyperSpec), got {type(likelihood).__name__}")
    if covar_module is not None and not isinstance(covar_module, (KernelSpec, hyper.ScaleKernel)):
        raise TypeError(f"covar_module must be a ScaleKernel (or a KernelSpec), got {type(covar_module).__name__}")
    lik = likelihood if likelihood is not None else hyper.get_default_likelihood()
    if isinstance(covar_module, hyper.ScaleKernel):
        cov = covar_module
        if cov.base_kernel.ard_num_dims != D:
            raise ValueError(f"covar_module has ard_num_dims={cov.base_kernel.ard_num_dims} but the inputs have {D} dimensions")
    else:
        cov = default_kernel(hyper.MaternKernel if kind == KIND_MATERN52 else hyper.RBFKernel, D)
    return lik, cov


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
        self._rep_cache = {}
        self.shard: Optional[sdist.TaskShard] = None   # set by meta_fit_scamlgp(shard=True): this stack = tasks [lo, hi) of T_global

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
        # (only the block rows the posterior kernels read are written: lower_only)
        fit["Linv"] = ops.linv_batched(fit["L"], fit["Linv_diag"], n_points=self.n_points, lower_only=True)
        self._fit = fit
        return fit

    @property
    def fit(self) -> dict:
        return self._fit if self._fit is not None else self.refresh()

    def _replicated(self, reps: int):
        """The data of ``reps`` hyper-parameter sets per task, built once per ``reps`` (not per evaluation)."""
        c = self._rep_cache.get(reps)
        if c is None:
            c = dict(X=self.X.repeat(reps, 1, 1) if reps > 1 else self.X, y=self.y.repeat(reps, 1) if reps > 1 else self.y,
                     npts=None if self.n_points is None else self.n_points.repeat(reps), nflt=self.n_float.repeat(reps), out=None)
            self._rep_cache = {reps: c}   # one entry: the buffers of another batch size are released
        return c

    def objective(self, raw: torch.Tensor, reps: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
        """Negative training objective and its gradient w.r.t. the raw parameters for ``reps``
        hyper-parameter sets per task: raw (reps * T, D+2), problem b belongs to task b % T.
        objective = -(mll + sum log p(theta) / n)   (gpytorch ExactMarginalLogLikelihood with priors,
        A5 of SURVEY.md).  The marginal likelihood is the differentiable op ``ops.FusedMLL`` (one fused-fit
        launch forward, the analytic gradient kernels backward); constraint transform and priors are torch
        ops, and torch.autograd chains them -- "autograd stays in PyTorch" (scamlgp/utils.py:175, 190)."""
        c = self._replicated(reps)
        raw_ = raw.detach().requires_grad_(True)
        theta = self.spec.to_theta(raw_)
        mll = ops.fused_mll(c["X"], c["y"], theta, self.kind, c["npts"], c["out"])
        if self.N <= ops.fit_max_n():
            c["out"] = ops.FusedMLL.last   # reuse the factor buffers: backward always runs before the next forward here
        f = -(mll + self.spec.log_prior(theta) / c["nflt"])
        bad = ops.FusedMLL.last["info"] > 0
        (g,) = torch.autograd.grad(torch.where(bad, torch.zeros_like(f), f).sum(), raw_)
        f = torch.where(bad, torch.full_like(f, float("nan")), f.detach())
        return f, g

    def summed_mll_and_grad(self, theta_shared: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """sum_t MLL_t and sum_t dMLL_t/dtheta at ONE hyper-parameter vector shared by all tasks -- of this stack and,
        when the stack is a shard, of every rank's: local sums, then one fused all-reduce [sum MLL || sum grad]
        (BASELINE configs[3]: "RCCL all-reduce of the MLL hyper-gradient")."""
        th = theta_shared.to(self.device, torch.float64).reshape(1, -1).expand(self.T, -1).contiguous().requires_grad_(True)
        mll = ops.fused_mll(self.X, self.y, th, self.kind, self.n_points)
        (g,) = torch.autograd.grad(mll.sum(), th)
        return sdist.reduce_mll_and_grad(mll.detach(), g, self.shard)

    # -- posteriors -----------------------------------------------------------------------
    def posterior(self, xq: torch.Tensor, cov_first: int = 0, want_var: bool = True, VA: Optional[torch.Tensor] = None) -> dict:
        """Un-standardised posteriors of all tasks at xq (M, D): mean (T, M), var (T, M),
        cov (T, cov_first, M).  ``VA``: the cached V of the leading ``cov_first`` points (ops.source_posteriors)."""
        f = self.fit
        return ops.source_posteriors(xq.to(self.device, torch.float64), self.X, self.theta, self.kind, f["L"], f["Linv_diag"],
                                     f["alpha"], self.y_mean, self.y_std, n_points=self.n_points, want_var=want_var,
                                     cov_first=cov_first, Linv=f.get("Linv"), VA=VA)

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
        if x.dim() > 2:
            raise ValueError("SourceGP.posterior takes x (M, D): batched inputs go through ScaMLGP.posterior / forward")
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
def _canonical_order(X: torch.Tensor, Y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """A task's observations in lexicographic order of (x_0, ..., x_{d-1}, y).  The reference sorts the meta
    evaluations before it builds the datasets (scamlgp/utils.py:72-109 ``sort_evaluations``) "to ensure that
    optimization runs are deterministic regardless of the order of input meta evaluations" -- the property its
    ``is_deterministic_with_shuffled_meta_data`` pins (scamlgp/testing.py:38-99); sorting here gives the kernels
    bit-identical stacks for shuffled inputs."""
    import numpy as np

    Xn = torch.as_tensor(X, dtype=torch.float64).reshape(-1, X.shape[-1]).cpu().numpy()
    Yn = torch.as_tensor(Y, dtype=torch.float64).reshape(-1, 1).cpu().numpy()
    order = np.lexsort(tuple(np.concatenate([Xn, Yn], 1).T[::-1]))
    return torch.from_numpy(Xn[order]), torch.from_numpy(Yn[order])


def meta_fit_scamlgp(
    meta_data: Dict[Hashable, SupervisedDataset],
    likelihood: Optional[hyper.GaussianLikelihood] = None,
    covar_module: Optional[hyper.ScaleKernel] = None,
    num_restarts_log_likelihood: int = 5,
    seed: Optional[int] = None,
    device: Optional[torch.device] = None,
    shard: bool = False,
    group=None,
) -> Dict[Hashable, SourceGP]:
    """Train the source GPs on the given meta-data (scamlgp/model.py:138-189).

    ``shard=True`` under an initialised torch.distributed group (one process per GPU): every rank is handed the SAME
    ``meta_data`` and keeps, fits and returns only its contiguous shard of the tasks (``dist.shard_range``); the
    tasks are independent, so the fit needs no collective.  A ScaMLGP built on such a dict all-reduces the one
    coupling of the path -- the weighted sums over tasks (scamlgp/model.py:129-134) -- across the ranks.

    ``likelihood`` / ``covar_module`` are templates, as in the reference (which deep-copies them per task): their
    constraints, priors and current values become every task's starting point.  Defaults: the reference's
    Gaussian likelihood and ScaleKernel(RBFKernel) with SingleTaskGP's priors.  All tasks and all restarts are
    optimised together on the GPU."""
    from .utils import optimize_marginal_likelihood

    if seed is not None:
        torch.manual_seed(seed=seed)
    validate_meta_data(meta_data)
    first = list(meta_data.values())[0]
    if first.X.dim() != 2:
        raise ValueError("batched (batch_shape x n x d) meta-data is not supported by the stacked GPU path")
    lik, cov = _resolve_modules(likelihood, covar_module, int(first.X.shape[-1]), hyper.get_kernel_source_gp)
    task_ids, data = list(meta_data.keys()), list(meta_data.values())
    ts = None
    if shard:
        ts = sdist.TaskShard(len(task_ids), group)
        if ts.hi == ts.lo:
            raise ValueError(f"rank {ts.rank} of {ts.world} would hold no task: shard fewer ranks than tasks")
        task_ids, data = task_ids[ts.local], data[ts.local]
    Xs, Ys = zip(*[_canonical_order(d.X(), d.Y()) for d in data])
    stack = SourceGPStack(task_ids, Xs, Ys, kind=cov.kind, spec=hyper.spec_from_modules(lik, cov), device=device)
    stack.shard = ts if ts is not None and ts.world > 1 else None
    optimize_marginal_likelihood(stack, num_restarts=num_restarts_log_likelihood)
    return {tid: SourceGP(stack, i) for i, tid in enumerate(stack.task_ids)}


def significant_weights_mask(weights: torch.Tensor, std_Y_vals: torch.Tensor, threshold: float) -> torch.Tensor:
    """Pruning criterion of scamlgp/model.py:192-215: task i is kept when its share of the scaled weights,
    w_i sigma_i / mean_j(w_j sigma_j), reaches ``threshold``."""
    scaled = weights * std_Y_vals
    return scaled * scaled.numel() / scaled.sum() >= threshold


def _compute_target_prior(x: torch.Tensor, source_gps: List[SourceGP], weights: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """scamlgp/model.py:108-135: mean = sum_i w_i mu_i(x) as (n, 1), cov = sum_i w_i^2 Sigma_i(x, x) (n, n),
    in original units.  One batched posterior launch + two weighted task sums."""
    if len(source_gps) != len(weights):
        raise ValueError(f"The number of source GPs, {len(source_gps)}, does not equal the number of weights, {len(weights)}")
    stack, idx = _stack_of(source_gps)
    if x.dim() > 2:
        raise ValueError("_compute_target_prior takes x (n, D): batched inputs go through ScaMLGP.posterior / forward")
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


def psd_safe_cholesky(A: torch.Tensor, max_tries: int = 3) -> torch.Tensor:
    """linear_operator's psd_safe_cholesky in torch, for the target GP's n x n matrix (n <= ~80; the reference
    reaches it through scamlgp/utils.py:171-177 for the target fit as for the source fits): try as is; on failure
    add 1e-8, 1e-7, 1e-6 to the diagonal in turn (each try replaces the previous jitter); then NotPSDError.  NaN in
    A raises at once.  Differentiable (the jitter is a constant shift)."""
    if bool(torch.isnan(A).any()):
        raise ops.NotPSDError(f"cholesky_cpu: {int(torch.isnan(A).sum())} of {A.numel()} elements of the matrix are NaN.")
    L, info = torch.linalg.cholesky_ex(A)
    if not bool(info.any()):
        return L
    eye = torch.eye(A.shape[-1], dtype=A.dtype, device=A.device)
    for i in range(max_tries):
        jitter = 1e-8 * 10 ** i
        warnings.warn(f"A not p.d., added jitter of {jitter:.1e} to the diagonal", RuntimeWarning)
        L, info = torch.linalg.cholesky_ex(A + jitter * eye)
        if not bool(info.any()):
            return L
    raise ops.NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter:.1e}.")


def _split_batch(X: torch.Tensor, D: int):
    """X (..., q, D) -> (flat (M, D), batch_shape or None, q): botorch hands models ``batch_shape x q x d`` inputs
    (scamlgp/model.py:359-384 keeps the batch dimensions); the kernels take one flat list of M = prod(batch) * q points."""
    if X.dim() <= 2:
        return X.reshape(-1, D), None, X.reshape(-1, D).shape[0]
    return X.reshape(-1, D), tuple(X.shape[:-2]), int(X.shape[-2])


def _batch_blocks(cov: torch.Tensor, batch, q: int) -> torch.Tensor:
    """The per-batch q x q diagonal blocks of a joint (M, M) covariance, shaped (*batch, q, q)."""
    B = cov.shape[0] // q
    blocks = cov.reshape(B, q, B, q).diagonal(dim1=0, dim2=2).permute(2, 0, 1)
    return blocks.reshape(*batch, q, q)


class _LazyMVN:
    """``posterior(X).mvn``: mean (M,), variance (M,) at once; the joint (M, M) covariance only when asked for
    (an acquisition function over 1 000 candidates never needs it)."""

    def __init__(self, mean: torch.Tensor, variance: torch.Tensor, cov_fn):
        self.mean, self.variance, self._cov_fn, self._cov = mean, variance, cov_fn, None

    @property
    def covariance_matrix(self) -> torch.Tensor:
        if self._cov is None:
            self._cov = self._cov_fn()
        return self._cov

    lazy_covariance_matrix = covariance_matrix

    @property
    def stddev(self) -> torch.Tensor:
        return self.variance.clamp_min(0.0).sqrt()


class TargetPosterior:
    """What botorch's ``model.posterior(X)`` returns, reduced to what its callers read: ``.mean`` / ``.variance``
    as (M, 1) columns, ``.mvn.mean`` (M,), ``.mvn.variance`` (M,), ``.mvn.covariance_matrix`` (M, M)."""

    def __init__(self, mean: torch.Tensor, variance: torch.Tensor, cov_fn):
        self.mvn = _LazyMVN(mean, variance, cov_fn)

    mean = property(lambda self: self.mvn.mean.unsqueeze(-1))
    variance = property(lambda self: self.mvn.variance.unsqueeze(-1))


class ScaMLGP:
    """Scalable meta-learning GP (scamlgp/model.py:218-384): target prior
    N(sum_i w_i mu_i, sum_i w_i^2 Sigma_i + k_t) over the posteriors of the source stack.

    Same constructor arguments and attributes as the reference class: ``likelihood`` / ``covar_module`` are the
    parameter-carrying modules (``hyper.GaussianLikelihood``, ``hyper.ScaleKernel``; the fitted ones of a previous
    model may be passed back in, scamlgp/optimizer.py:176-183), ``train_inputs``, ``train_targets``, ``weights``,
    ``source_gps``, ``source_means``, ``source_covs``, ``forward``, ``posterior``, ``train`` / ``eval``,
    ``state_dict`` / ``load_state_dict``."""

    def __init__(self, train_X: torch.Tensor, train_Y: torch.Tensor, source_gps: Dict[Hashable, SourceGP],
                 likelihood: Optional[hyper.GaussianLikelihood] = None, covar_module: Optional[hyper.ScaleKernel] = None,
                 weight_pruning_threshold: float = 1e-3) -> None:
        self._weight_pruning_threshold = weight_pruning_threshold
        self.source_gps = source_gps
        gps = list(source_gps.values())
        self._stack, self._idx = _stack_of(gps)
        dev = self._stack.device
        self.device = dev
        self.train_X = torch.as_tensor(train_X, dtype=torch.float64).reshape(-1, self._stack.D).to(dev)
        self.train_Y = torch.as_tensor(train_Y, dtype=torch.float64).reshape(-1, 1).to(dev)
        self._idx_dev = torch.as_tensor(self._idx, dtype=torch.int64, device=dev)
        self._shard = self._stack.shard
        if self._shard is not None and len(gps) != self._stack.T:
            raise ValueError("a sharded source stack must be passed whole (every rank its full shard)")
        self.n, self.T = self.train_X.shape[0], (self._shard.n_tasks if self._shard is not None else len(gps))
        lik, cov = _resolve_modules(likelihood, covar_module, self._stack.D, hyper.get_default_kernel)
        self.likelihood, self.covar_module = lik.to(dev), cov.to(dev)
        self.kind = cov.kind
        self.spec = hyper.spec_from_modules(lik, cov)
        # standardise w.r.t. ALL meta + target observations (scamlgp/model.py:264-276)
        self.has_transform = self.train_Y.numel() > 0
        if self._shard is None:
            m, s = standardize_fit(torch.cat([self._stack.raw_targets(), self.train_Y], dim=-2))
        else:   # the same statistics from all-reduced sums: no rank ever holds the other shards' observations
            m, s = sdist.standardize_fit_sharded(self._stack.raw_targets(), self.train_Y, self._shard)
        self.m_all, self.s_all = (m.squeeze(), s.squeeze()) if self.has_transform else (self.train_Y.new_zeros(()), self.train_Y.new_ones(()))
        self._m_all_f, self._s_all_f = float(self.m_all), float(self.s_all)   # (one sync here instead of one per posterior call)
        self.outcome_transform = _OutcomeTransform(self.m_all, self.s_all) if self.has_transform else None
        self.train_inputs = (self.train_X,)
        self.train_targets = ((self.train_Y - self.m_all) / self.s_all).squeeze(-1)
        # cached source posteriors at the target inputs, all tasks, no pruning (scamlgp/model.py:279-289)
        if self.n > 0:
            # (V of the training points: computed once here, reused by every later posterior / gradient call of this model)
            va = self._train_VA() if (self.n <= 96 and self._stack.fit.get("Linv") is not None) else None
            p = self._stack.posterior(self.train_X, cov_first=self.n, VA=va)
            # (sharded: every rank contributes its tasks' columns, one all-reduce each -- n <= ~80, tiny)
            self.source_means = sdist.gather_task_axis(p["mean"][self._idx].transpose(0, 1).contiguous(), self._shard)   # (n, T)
            self.source_covs = sdist.gather_task_axis(p["cov"][self._idx].permute(1, 2, 0).contiguous(), self._shard)    # (n, n, T)
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
    def raw_theta(self) -> torch.Tensor:
        """[raw lengthscales (D), raw outputscale, raw noise]: the modules' parameters as one vector."""
        cov, lik = self.covar_module, self.likelihood
        return torch.cat([cov.base_kernel.raw_lengthscale.reshape(-1), cov.raw_outputscale.reshape(1), lik.raw_noise.reshape(1)])

    @raw_theta.setter
    def raw_theta(self, value: torch.Tensor) -> None:
        v = torch.as_tensor(value, dtype=torch.float64).to(self.device).detach()
        D = self._stack.D
        self.covar_module.base_kernel._p.raw = v[:D].clone()
        self.covar_module._p.raw = v[D].clone()
        self.likelihood._p.raw = v[D + 1].clone()

    def _param_tensors(self):
        cov, lik = self.covar_module, self.likelihood
        return (cov.base_kernel.raw_lengthscale, cov.raw_outputscale, lik.raw_noise)

    @staticmethod
    def _same_tensors(held, now) -> bool:
        """True if `now` are the very tensors `held` was computed from, unmodified since ((tensor, version) pairs: the held
        references keep the objects alive, so identity cannot be faked by a recycled address)."""
        return held is not None and len(held) == len(now) and all(h is t and v == t._version for (h, v), t in zip(held, now))

    @property
    def theta(self) -> torch.Tensor:
        """Constrained target hyper-parameters [lengthscales, outputscale, noise].  Cached per parameter state: an acquisition
        pass asks for them (and for the pruned weights below) on every call, each time a dozen element-wise launches for the
        same numbers -- a quarter of a scoring pass at configs[4] was such glue."""
        now = self._param_tensors()
        c = getattr(self, "_theta_cache", None)
        if c is None or not self._same_tensors(c[0], now):
            c = (tuple((t, t._version) for t in now), self.spec.to_theta(self.raw_theta))
            self._theta_cache = c
        return c[1]

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
        if getattr(self, "_stds_all", None) is None:
            self._stds_all = sdist.gather_task_axis(self._stack.y_std[self._idx], self._shard)
        return self._stds_all

    def _active_tasks(self):
        """Pruned weights as vectors over this rank's stack: w_full (T_stack,), active mask (bool)
        (scamlgp/model.py:365-372; the mask is decided on ALL T weights, each rank then takes its slice)."""
        w = self.weights
        c = getattr(self, "_active_cache", None)
        if c is not None and self._same_tensors(c[0], (w,)):
            return c[1], c[2]      # (same weights as last time: see `theta`)
        held = ((w, w._version),)
        mask = significant_weights_mask(w, self._std_source_stds(), self._weight_pruning_threshold)
        if self._shard is not None:
            w, mask = w[self._shard.local], mask[self._shard.local]
        # (scatter instead of mask indexing: nothing here depends on a value read back by the host, so the whole
        #  acquisition pass can be captured into a HIP graph)
        idx = self._idx_dev
        w_full = torch.zeros(self._stack.T, dtype=torch.float64, device=self.device).scatter(0, idx, w)
        active = torch.zeros(self._stack.T, dtype=torch.bool, device=self.device).scatter(0, idx, mask)
        self._active_cache = (held, w_full, active)
        return w_full, active

    def _source_prior(self, x: torch.Tensor, cov_first: int, train_first: bool = False):
        """sum_i w_i mu_i(x) (M,), sum_i w_i^2 Sigma_i (cov_first, M), sum_i w_i^2 var_i (M,) over the significant tasks
        of ALL ranks, in original units: one batched posterior launch over this rank's stack, the weighted task sums,
        and -- sharded -- ONE all-reduce of the fused buffer [mu_s || Sigma_s || var_s]."""
        w_full, active = self._active_tasks()
        # (the leading points of a joint evaluation are the model's training inputs: their V is computed once per model)
        va = self._train_VA() if (train_first and 0 < cov_first == self.n <= 96 and x.shape[0] > self.n and self._stack.fit.get("Linv") is not None) else None
        p = self._stack.posterior(x, cov_first=cov_first, VA=va)
        mu_s, cov_s = ops.weighted_prior_reduce(p["mean"], p["cov"], w_full, active)
        var_s = ops.weighted_task_sum(p["var"], w_full, 2, active)
        mu_s, cov_s, var_s = sdist.fused_allreduce([mu_s, cov_s, var_s], self._shard)
        return mu_s, cov_s, var_s

    def forward(self, x: torch.Tensor) -> _MVN:
        """scamlgp/model.py:359-384.  Training: cached source terms at train_X; eval: pruned weights and a
        fresh batched source posterior at x.  Both in the standardised target space, plus k_t(x, x)."""
        x, batch, q = _split_batch(torch.as_tensor(x, dtype=torch.float64).to(self.device), self._stack.D)
        w = self.weights
        if self.training:
            mean = self.source_means @ w
            cov = self.source_covs @ w ** 2
        else:
            # = _compute_target_prior over the significant models (scamlgp/model.py:365-375), as stack-wide sums
            mean, cov, _ = self._source_prior(x, x.shape[0])
        if self.has_transform:
            mean = (mean - self.m_all) / self.s_all
            cov = cov / self.s_all ** 2
        cov = cov + _kernel_torch(x, x, self.theta, self.kind)
        if batch is not None and not self.training:   # (batch, q, D) input: independent q-point joints, one per batch element
            return _MVN(mean.reshape(*batch, q), _batch_blocks(cov, batch, q))
        return _MVN(mean, cov)

    __call__ = forward

    def mll(self, raw_theta: Optional[torch.Tensor] = None, raw_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Training objective (A9): [log N(y~ | mean, cov + sigma^2 I) + log priors] / n, differentiable in torch."""
        rt = self.raw_theta if raw_theta is None else raw_theta
        w = self.raw_weights if raw_weights is None else raw_weights
        theta = self.spec.to_theta(rt)
        mean = (self.source_means @ w - self.m_all) / self.s_all
        cov = (self.source_covs @ w ** 2) / self.s_all ** 2 + _kernel_torch(self.train_X, self.train_X, theta, self.kind)
        cov = cov + theta[-1] * torch.eye(self.n, dtype=torch.float64, device=self.device)
        if 1 <= self.n <= ops.fit_max_n():
            # the library's jittered Cholesky + solves as ONE differentiable op: no host synchronisation (psd_safe_cholesky's
            # status check is one), so that the whole objective can be replayed from a HIP graph (utils._fit_target); a matrix
            # that is not positive definite even with jitter gives NaN instead of NotPSDError
            val = ops.gaussian_log_prob(cov, self.train_targets - mean)
        else:
            Lc = psd_safe_cholesky(cov)
            v = torch.linalg.solve_triangular(Lc, (self.train_targets - mean).unsqueeze(-1), upper=False)
            val = -0.5 * ((v * v).sum() + 2.0 * torch.log(torch.diagonal(Lc)).sum() + self.n * _LOG_2PI)
        val = val + self.spec.log_prior(theta) + self.weights_prior.log_prob(w).sum()
        return val / self.n

    def target_problem(self) -> Optional[ops.TargetFitProblem]:
        """The training set in the layouts of the library's target-fit kernel (built once; None if the kernel does not take
        this shape -- more target points than its LDS holds, D > 16 -- and the torch objective ``mll`` is all there is)."""
        if getattr(self, "_tprob", None) is None:
            if self.n < 1 or not ops.TargetFitProblem.supported(self.n, self.T, self._stack.D):
                return None
            self._tprob = ops.TargetFitProblem(self.source_means, self.source_covs, self.train_X, self.train_targets, self._m_all_f,
                                               self._s_all_f, self.spec, self.weights_prior, self.weights_lower_bound, self.kind)
        return self._tprob

    def mll_and_grad(self, z: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """The training objective ``mll`` and its analytic gradient w.r.t. z = [raw_theta || raw_weights] for the rows of z
        (B, D + 2 + T) (default: the model's current parameters, B = 1): ONE launch of scaml_target_mll_f64 instead of torch
        autograd through ~100 small kernels.  Returns (value (B,), grad (B, P))."""
        prob = self.target_problem()
        if prob is None:
            raise ValueError("the target-fit kernel does not take this shape; use ScaMLGP.mll with torch autograd")
        if z is None:
            z = torch.cat([self.raw_theta, self.raw_weights]).unsqueeze(0)
        out = ops.target_mll(prob, z.to(self.device, torch.float64).reshape(-1, prob.P))
        return out["value"], out["grad"]

    def _joint(self, Xq: torch.Tensor, full: bool):
        """Standardised joint prior over cat(train_X, Xq) (A8, A10) from ONE source-posterior launch: mean (n + M,),
        the n x (n + M) covariance block (or the whole (n + M)^2 one with ``full``) and the query diagonal."""
        n, M = self.n, Xq.shape[0]
        xall = torch.cat([self.train_X, Xq], 0)
        first = n + M if full else n
        mu_s, cov_s, var_s = self._source_prior(xall, first)
        theta = self.theta
        mean = (mu_s - self.m_all) / self.s_all
        var_q = var_s[n:] / self.s_all ** 2 + theta[-2]
        cov = None
        if first > 0:
            cov = cov_s / self.s_all ** 2 + _kernel_torch(xall[:first], xall, theta, self.kind)
        return mean, cov, var_q, theta

    # -- posterior with input gradients (what the acquisition optimiser differentiates through) ------------------------------
    def supports_posterior_grad(self) -> bool:
        """The analytic input-gradient path needs training data within the fused covariance block (n <= 96), D <= 15 and the
        explicit inverse factors of the source stack (SourceGPStack.refresh caches them)."""
        return 1 <= self.n <= 96 and self._stack.D <= 15 and self._stack.N <= 512 and self.device.type == "cuda"

    def _train_VA(self) -> torch.Tensor:
        """V = L^-1 K(X_t, train_X) of every source task (T, N, n): fixed for the model's lifetime (sources and training inputs are)."""
        if getattr(self, "_VA", None) is None:
            st, f = self._stack, self._stack.fit
            self._VA = ops.source_posteriors(self.train_X, st.X, st.theta, st.kind, f["L"], f["Linv_diag"], f["alpha"], st.y_mean, st.y_std,
                                             n_points=st.n_points, want_var=False, keep_V=True, Linv=f["Linv"])["V"]
        return self._VA

    def _train_prior(self):
        """The weighted source sums at the training inputs (mu_s (n,), Sigma_s (n, n), var_s (n,)): per weight state."""
        w = self.weights
        c = getattr(self, "_train_prior_cache", None)
        if c is None or not self._same_tensors(c[0], (w,)):
            c = (((w, w._version),), self._source_prior(self.train_X, self.n))
            self._train_prior_cache = c
        return c[1]

    def _target_factor(self):
        """The jittered Cholesky of the target GP's training block Knn (+ alpha) at the current weights / hyper-parameters, or None
        if it has not been computed for this parameter state yet: it does not depend on the query points, and an acquisition
        optimisation scores hundreds of query batches against one parameter state."""
        c = getattr(self, "_factor_cache", None)
        now = (self.weights,) + self._param_tensors()
        return c[1] if c is not None and self._same_tensors(c[0], now) else None

    def _keep_target_factor(self, factor) -> None:
        now = (self.weights,) + self._param_tensors()
        self._factor_cache = (tuple((t, t._version) for t in now), factor)

    def posterior_with_grad(self, X: torch.Tensor):
        """Target posterior mean / variance at X (Mq, D) in original units AND their gradients w.r.t. X: (mu (Mq,), var (Mq,),
        dmu (Mq, D), dvar (Mq, D)).  The reference gets these from torch autograd through ``model.posterior`` inside botorch's
        optimize_acqf (scamlgp/utils.py:215-224, SURVEY 3.3 HOT LOOP #4); here: one source pass with 16 columns per query point
        (value + D derivative right-hand sides through the same L^-1 product, scaml_posterior_linv_grad_f64), the weighted task
        sums, the target GP's value path (assemble / jittered Cholesky / solve / finish) and one contraction kernel
        (scaml_target_posterior_grad_f64) -- an L-BFGS-B evaluation over R starts scores R points instead of the (2 D + 1) R of a
        central-difference stencil, with exact gradients."""
        if not self.supports_posterior_grad():
            raise NotImplementedError("analytic posterior gradients need 1 <= n <= 96 training points and D <= 15")
        Xq = torch.as_tensor(X, dtype=torch.float64).reshape(-1, self._stack.D).to(self.device).contiguous()
        Mq, n = Xq.shape[0], self.n
        st, f = self._stack, self._stack.fit
        w_full, active = self._active_tasks()
        mu_t, cov_tt, var_t = self._train_prior()
        g = ops.source_posteriors_grad(Xq, self.train_X, st.X, st.theta, st.kind, f["Linv"], f["alpha"], st.y_mean, st.y_std, st.n_points,
                                       self._train_VA())
        mu_g, cov_g = ops.weighted_prior_reduce(g["mu"].reshape(st.T, Mq * 16), g["cov"], w_full, active)
        var_g = ops.weighted_task_sum(g["var"].reshape(st.T, Mq * 16), w_full, 2, active)
        mu_g, cov_g, var_g = sdist.fused_allreduce([mu_g, cov_g, var_g], self._shard)
        # the value columns next to the training block: the joint prior the target GP's value path takes
        cov_s = torch.cat([cov_tt, cov_g.reshape(n, Mq, 16)[:, :, 0]], 1)
        mean_s = torch.cat([mu_t, mu_g.reshape(Mq, 16)[:, 0]])
        var_s = torch.cat([var_t, var_g.reshape(Mq, 16)[:, 0]])
        xall = torch.cat([self.train_X, Xq], 0)
        theta = self.theta
        full = ops.target_posterior_full(cov_s, mean_s, var_s, xall, theta, self.train_targets, self._m_all_f, self._s_all_f, self.kind,
                                         factor=self._target_factor())
        self._keep_target_factor(full["factor"])
        dmu, dvar = ops.target_posterior_grad(cov_g, mu_g, var_g, self.train_X, Xq, theta, full["alpha"], full["Z"], self._s_all_f,
                                              full["info"], self.kind)
        return full["mu"], full["var"], dmu, dvar

    def posterior(self, X: torch.Tensor, observation_noise: bool = False) -> TargetPosterior:
        """Target posterior at X (M, D) in original units (A10).  The source prior is evaluated ONCE at
        cat(train_X, X) -- train block, cross block and query diagonal -- instead of once per query as the
        reference does.  ``.mvn.covariance_matrix`` (the joint (M, M) covariance) costs a second
        launch with the full query block and is only computed when read.  X (M, D): M points, ``.mean`` (M, 1), joint (M, M)
        covariance.  X (*batch, q, D): ``.mean`` / ``.variance`` (*batch, q, 1), ``.mvn.covariance_matrix`` (*batch, q, q)."""
        Xq, batch, q = _split_batch(torch.as_tensor(X, dtype=torch.float64).to(self.device), self._stack.D)
        n = self.n
        if 1 <= n <= ops.fit_max_n():
            # the library's target-GP path: weighted source sums at cat(train_X, Xq), then assemble -> jittered Cholesky
            # (T = 1) -> solve -> finish: four launches, no host synchronisation, no torch arithmetic
            xall = torch.cat([self.train_X, Xq], 0)
            mu_s, cov_s, var_s = self._source_prior(xall, n, train_first=True)
            full = ops.target_posterior_full(cov_s, mu_s, var_s, xall, self.theta, self.train_targets, self._m_all_f, self._s_all_f, self.kind,
                                             observation_noise, factor=self._target_factor())
            self._keep_target_factor(full["factor"])
            mu_o, var_o = full["mu"], full["var"]
            # (a factorisation that fails even with jitter -- psd_safe_cholesky would raise NotPSDError -- comes back as NaN: the status
            #  stays on the device so that an acquisition-function evaluation never waits for the host)
        else:
            mean, cov, var_q, theta = self._joint(Xq, full=False)
            if n == 0:
                mu, var = mean, var_q
            else:
                Knn = cov[:, :n] + theta[-1] * torch.eye(n, dtype=torch.float64, device=self.device)
                Knq = cov[:, n:]
                Lc = psd_safe_cholesky(Knn)
                resid = (self.train_targets - mean[:n]).unsqueeze(-1)
                a = torch.cholesky_solve(resid, Lc).squeeze(-1)
                mu = mean[n:] + Knq.transpose(0, 1) @ a
                Vq = torch.linalg.solve_triangular(Lc, Knq, upper=False)
                var = var_q - (Vq * Vq).sum(0)
            if observation_noise:
                var = var + theta[-1]
            mu_o, var_o = self.m_all + self.s_all * mu, self.s_all ** 2 * var

        def full_cov() -> torch.Tensor:
            _, cj, _, th = self._joint(Xq, full=True)
            S = cj[n:, n:]
            if n > 0:
                Lc2 = psd_safe_cholesky(cj[:n, :n] + th[-1] * torch.eye(n, dtype=torch.float64, device=self.device))
                V2 = torch.linalg.solve_triangular(Lc2, cj[:n, n:], upper=False)
                S = S - V2.transpose(0, 1) @ V2
            if observation_noise:
                S = S + th[-1] * torch.eye(S.shape[0], dtype=torch.float64, device=self.device)
            return self.s_all ** 2 * S

        if batch is not None:
            # botorch's batch_shape x q x d convention (scamlgp/model.py:359-384 keeps the batch dimensions): mean / variance
            # (*batch, q, 1), covariance (*batch, q, q) -- one joint per batch element, NOT one joint over all points.  q = 1 (what
            # optimize_acqf sends to an analytic acquisition function) needs no joint at all.
            if q == 1:
                return TargetPosterior(mu_o.reshape(*batch, 1), var_o.reshape(*batch, 1), lambda: var_o.reshape(*batch, 1, 1))
            return TargetPosterior(mu_o.reshape(*batch, q), var_o.reshape(*batch, q), lambda: _batch_blocks(full_cov(), batch, q))
        return TargetPosterior(mu_o, var_o, full_cov)
