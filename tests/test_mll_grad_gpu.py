"""GPU parity test for the analytic MLL gradient (scaml_mll_backward_f64) against torch autograd
through the oracle's op sequence (CPU, fp64)."""
import math

import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import ops, synthetic

pytestmark = pytest.mark.gpu


def _autograd_grad(X, y, theta, kind):
    th = theta.clone().requires_grad_(True)
    N, D = X.shape
    K = O._kernel_matrix_grad(X, th, kind) + th[D + 1] * torch.eye(N, dtype=torch.float64)
    Lc = torch.linalg.cholesky(K)
    v = torch.linalg.solve_triangular(Lc, y.unsqueeze(-1), upper=False)
    mll = -0.5 * ((v * v).sum() + 2 * torch.log(torch.diagonal(Lc)).sum() + N * math.log(2 * math.pi)) / N
    (g,) = torch.autograd.grad(mll, th)
    return g


@pytest.mark.parametrize("T,N,D,kind", [
    (3, 32, 2, O.KIND_RBF),
    (3, 48, 3, O.KIND_MATERN52),
    (2, 100, 5, O.KIND_MATERN52),
    (2, 256, 8, O.KIND_RBF),
    (2, 256, 8, O.KIND_MATERN52),
    (2, 400, 6, O.KIND_MATERN52),   # two-block fit (N > 256) feeding the same gradient kernels
    (2, 512, 6, O.KIND_RBF),
    (2, 72, 40, O.KIND_MATERN52),   # large D: the staged points of a wave's four blocks need > 64 KB of LDS per workgroup
    (3, 16, 1, O.KIND_RBF),         # one block: a super-tile with three absent tiles
    (5, 200, 8, O.KIND_MATERN52),   # N not a multiple of 16 in the largest single-launch size class
    (4, 129, 7, O.KIND_RBF),        # first size past a class boundary: mostly padding
    (3, 64, 8, O.KIND_MATERN52),
    (2, 256, 9, O.KIND_MATERN52),   # D > 8: the two-launch path
])
def test_mll_gradient_matches_autograd(T, N, D, kind, device):
    d = synthetic.smooth_field_task_stack(T, N, D, seed=20 + N)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    rng = np.random.default_rng(N + D)
    theta = torch.from_numpy(np.concatenate([0.5 * (1 + 0.6 * (rng.uniform(size=(T, D)) - 0.5)), 0.8 + 0.4 * rng.uniform(size=(T, 1)),
                                             np.full((T, 1), 2e-3)], 1))
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    # zero_upper=False: what the optimiser loop passes -- the strict upper triangle of L is never read
    fit = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, want_linv=True, zero_upper=N > 256)
    if N <= 256:
        fit["L"] = fit["L"] + torch.triu(torch.full_like(fit["L"], float("nan")), diagonal=1)
    g = ops.mll_backward(X.to(device), theta.to(device), kind, fit["L"], fit["Linv_diag"], fit["alpha"]).cpu()
    for t in range(T):
        ref = _autograd_grad(X[t], y[t], theta[t], kind)
        # gradient entries span orders of magnitude (the noise derivative is ~1e3 larger): compare per entry
        torch.testing.assert_close(g[t], ref, rtol=1e-6, atol=1e-9 * float(ref.abs().max()))
    # the single-launch kernel (N <= 256, D <= 8) and the two-launch path (L^-1 in memory) agree -- both forced: by shape the library
    # sends stacks of at most 64 tasks of more than 64 points through the two launches
    from scamlgp_amd import _lib
    Lc = torch.tril(torch.nan_to_num(fit["L"]))
    was = _lib.lib.scaml_debug_force_two_launch_grad(2)
    try:
        g1 = ops.mll_backward(X.to(device), theta.to(device), kind, fit["L"], fit["Linv_diag"], fit["alpha"]).cpu()
        _lib.lib.scaml_debug_force_two_launch_grad(1)
        g2 = ops.mll_backward(X.to(device), theta.to(device), kind, Lc, fit["Linv_diag"], fit["alpha"]).cpu()
    finally:
        _lib.lib.scaml_debug_force_two_launch_grad(was)
    for t in range(T):
        ref = _autograd_grad(X[t], y[t], theta[t], kind)
        torch.testing.assert_close(g1[t], ref, rtol=1e-6, atol=1e-9 * float(ref.abs().max()))
    torch.testing.assert_close(g1, g2, rtol=1e-8, atol=1e-11 * float(g2.abs().max()))
    torch.testing.assert_close(g, g2, rtol=1e-8, atol=1e-11 * float(g2.abs().max()))


def test_mll_gradient_ragged(device):
    T, N, D, kind = 3, 40, 2, O.KIND_RBF
    d = synthetic.branin_task_stack(T, N, seed=9)
    npts = [40, 17, 33]
    X = torch.from_numpy(d["X"])
    y = torch.zeros(T, N, dtype=torch.float64)
    for t in range(T):
        yy, _, _ = synthetic.standardize_rows(d["Y"][t:t + 1, :npts[t]])
        y[t, :npts[t]] = torch.from_numpy(yy[0])
    theta = torch.tensor([[0.4, 0.6, 1.1, 1e-3]] * T, dtype=torch.float64)
    n_dev = torch.tensor(npts, dtype=torch.int32, device=device)
    fit = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, n_points=n_dev, want_linv=True)
    g = ops.mll_backward(X.to(device), theta.to(device), kind, fit["L"], fit["Linv_diag"], fit["alpha"], n_points=n_dev).cpu()
    for t in range(T):
        ref = _autograd_grad(X[t, :npts[t]], y[t, :npts[t]], theta[t], kind)
        torch.testing.assert_close(g[t], ref, rtol=1e-6, atol=1e-9 * float(ref.abs().max()))
