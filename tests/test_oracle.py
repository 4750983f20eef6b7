"""CPU tests of the oracle (oracle/gp_oracle.py): cross-validation against scikit-learn and
scipy (two independent implementations importable here), restated upstream semantics, and
the committed golden fixtures.  PARITY UNPINNED by the reference itself (see oracle header)."""
import glob
import math
import os

import numpy as np
import pytest
import scipy.linalg
import scipy.stats
import torch

from oracle import gp_oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _problem(N=40, D=3, seed=0, noise=1e-3):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(N, D, dtype=torch.float64, generator=g)
    y = torch.sin(4 * X.sum(-1)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    y = (y - y.mean()) / y.std()
    theta = torch.tensor([0.4, 0.7, 0.55][:D] + [1.3, noise], dtype=torch.float64)
    return X, y, theta


@pytest.mark.parametrize("kind", [O.KIND_RBF, O.KIND_MATERN52])
def test_fit_and_posterior_match_sklearn(kind):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern

    X, y, theta = _problem()
    D = X.shape[1]
    ls, os_, noise = theta[:D].numpy(), float(theta[D]), float(theta[D + 1])
    base = RBF(length_scale=ls) if kind == O.KIND_RBF else Matern(length_scale=ls, nu=2.5)
    gpr = GaussianProcessRegressor(kernel=ConstantKernel(os_) * base, alpha=noise, optimizer=None).fit(X.numpy(), y.numpy())
    out = O.gp_fit(X, y, theta, kind)
    np.testing.assert_allclose(out["L"].numpy(), gpr.L_, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(out["alpha"].numpy(), gpr.alpha_, rtol=1e-6, atol=1e-9)
    N = X.shape[0]
    # sklearn's LML is un-normalised and prior-free
    np.testing.assert_allclose(float(out["mll"]) * N, gpr.log_marginal_likelihood_value_, rtol=1e-9)
    xq = torch.rand(9, D, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    mu, cov = O.source_posterior(xq, X, theta, kind, out["L"], out["alpha"], 0.0, 1.0)
    m_sk, c_sk = gpr.predict(xq.numpy(), return_cov=True)
    np.testing.assert_allclose(mu.numpy(), m_sk, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(cov.numpy(), c_sk, rtol=1e-6, atol=1e-9)
    mu_d, var_d = O.source_posterior(xq, X, theta, kind, out["L"], out["alpha"], 0.0, 1.0, full_cov=False)
    np.testing.assert_allclose(var_d.numpy(), np.diag(c_sk), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("kind", [O.KIND_RBF, O.KIND_MATERN52])
def test_fit_matches_scipy_cholesky(kind):
    X, y, theta = _problem(N=64, D=2, seed=3)
    out = O.gp_fit(X, y, theta, kind, dist="direct")
    K = out["K"].numpy()
    c, low = scipy.linalg.cho_factor(K, lower=True)
    np.testing.assert_allclose(np.tril(c), out["L"].numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(scipy.linalg.cho_solve((c, low), y.numpy()), out["alpha"].numpy(), rtol=1e-7)
    sign, ld = np.linalg.slogdet(K)
    assert sign > 0
    np.testing.assert_allclose(float(out["logdet"]), ld, rtol=1e-10)
    np.testing.assert_allclose(float(out["quad"]), y.numpy() @ np.linalg.solve(K, y.numpy()), rtol=1e-7)


def test_gpytorch_and_direct_distances_agree():
    X, _, theta = _problem(N=50, D=3, seed=7)
    a = X / theta[:3]
    d_g = O.sq_dist_gpytorch(a, a, True)
    d_d = O.sq_dist_direct(a, a)
    assert torch.all(d_g.diagonal() == 0)
    torch.testing.assert_close(d_g, d_d, rtol=0, atol=1e-12)
    b = torch.rand(7, 3, dtype=torch.float64) / theta[:3]
    torch.testing.assert_close(O.sq_dist_gpytorch(b, a, False), O.sq_dist_direct(b, a), rtol=0, atol=1e-12)


def test_matern_diagonal_uses_clamped_distance():
    X, _, theta = _problem(N=8, D=3)
    K = O.kernel_matrix(X, None, theta[:3], theta[3], O.KIND_MATERN52)
    torch.testing.assert_close(K.diagonal(), torch.full((8,), float(theta[3]), dtype=torch.float64), rtol=1e-14, atol=0)


def test_interval_transform_round_trip_and_inits():
    # reference inits: lengthscale 0.5 / outputscale 1.0 in [1e-4, 1e2]; noise 1e-3 in [1e-8, 1e-2]
    for val, lo, hi in [(0.5, 1e-4, 1e2), (1.0, 1e-4, 1e2), (1e-3, 1e-8, 1e-2), (0.1, 1e-4, 1e2)]:
        raw = O.interval_inverse_transform(torch.tensor(val, dtype=torch.float64), lo, hi)
        back = O.interval_transform(raw, lo, hi)
        assert abs(float(back) - val) < 1e-12 * max(1, val)


def test_prior_log_probs_match_torch_distributions():
    x = torch.tensor([0.05, 0.5, 1.7, 12.0], dtype=torch.float64)
    for c, r in [(3.0, 6.0), (2.0, 0.15), (1.0, 1.0)]:
        torch.testing.assert_close(O.gamma_log_prob(x, c, r), torch.distributions.Gamma(torch.tensor(c, dtype=torch.float64), torch.tensor(r, dtype=torch.float64)).log_prob(x))
    for loc, sc in [(-8.0, 2.0), (0.5, 1.5), (-2.0, 3.0)]:
        torch.testing.assert_close(O.lognormal_log_prob(x, loc, sc), torch.distributions.LogNormal(torch.tensor(loc, dtype=torch.float64), torch.tensor(sc, dtype=torch.float64)).log_prob(x))
        np.testing.assert_allclose(O.lognormal_log_prob(x, loc, sc).numpy(), scipy.stats.lognorm(s=sc, scale=math.exp(loc)).logpdf(x.numpy()))


def test_standardize_floor_and_single_point():
    Y = torch.tensor([[1.0], [1.0], [1.0]], dtype=torch.float64)
    m, s = O.standardize_fit(Y)
    assert float(m) == 1.0 and float(s) == 1.0
    m, s = O.standardize_fit(torch.tensor([[2.5]], dtype=torch.float64))
    assert float(m) == 2.5 and float(s) == 1.0
    Y = torch.tensor([[1.0], [2.0], [4.0]], dtype=torch.float64)
    m, s = O.standardize_fit(Y)
    np.testing.assert_allclose(float(s), np.std([1.0, 2.0, 4.0], ddof=1))


def test_psd_safe_cholesky_escalates_only_failing_members():
    g = torch.Generator().manual_seed(1)
    B = torch.rand(3, 6, 3, dtype=torch.float64, generator=g)
    A = B @ B.transpose(-1, -2)  # rank 3 -> singular
    A[0] += 1e-3 * torch.eye(6, dtype=torch.float64)  # member 0 fine
    A[1] -= 1e-9 * torch.eye(6, dtype=torch.float64)  # needs 1e-8
    A[2] -= 5e-8 * torch.eye(6, dtype=torch.float64)  # needs 1e-7
    L, info0, jit = O.psd_safe_cholesky(A)
    assert info0[0] == 0 and info0[1] > 0 and info0[2] > 0
    assert jit.tolist() == [0.0, 1e-8, 1e-7]
    torch.testing.assert_close(L[0] @ L[0].T, A[0])
    with pytest.raises(O.NotPSDError):
        O.psd_safe_cholesky(A - 1e-3 * torch.eye(6, dtype=torch.float64))
    with pytest.raises(O.NotPSDError):
        O.psd_safe_cholesky(torch.full((2, 2), float("nan"), dtype=torch.float64))


def test_pruning_mask_and_target_prior():
    w = torch.tensor([0.5, 1e-6, 0.3, 0.0], dtype=torch.float64)
    s = torch.tensor([1.0, 2.0, 0.5, 1.0], dtype=torch.float64)
    mask = O.significant_weights_mask(w, s, 1e-3)
    assert mask.tolist() == [True, False, True, False]
    mus = torch.arange(12, dtype=torch.float64).reshape(4, 3)
    covs = torch.stack([torch.eye(3, dtype=torch.float64) * (i + 1) for i in range(4)])
    mu, cov = O.target_prior(mus, covs, w, mask)
    torch.testing.assert_close(mu, 0.5 * mus[0] + 0.3 * mus[2])
    torch.testing.assert_close(cov, 0.25 * covs[0] + 0.09 * covs[2])


def test_acquisition_functions():
    mu = torch.tensor([0.1, -0.4, 2.0], dtype=torch.float64)
    var = torch.tensor([0.04, 1.0, 1e-12], dtype=torch.float64)
    torch.testing.assert_close(O.ucb_minimize(mu, var), -mu + 3.0 * var.sqrt())
    ei = O.expected_improvement_minimize(mu, var, best_f=0.0)
    sigma = np.sqrt(np.maximum(var.numpy(), 1e-9))
    u = -(mu.numpy() - 0.0) / sigma
    np.testing.assert_allclose(ei.numpy(), sigma * (scipy.stats.norm.pdf(u) + u * scipy.stats.norm.cdf(u)), rtol=1e-10, atol=1e-300)


def test_mll_gradient_matches_finite_differences():
    X, y, theta = _problem(N=24, D=2, seed=11)
    bounds = [(1e-4, 1e2)] * 3 + [(1e-8, 1e-2)]
    raw = torch.stack([O.interval_inverse_transform(theta[i], *bounds[i]) for i in range(4)])
    for kind in (O.KIND_RBF, O.KIND_MATERN52):
        val, g, _ = O.mll_value_and_grad_raw(X, y, raw, kind, bounds)
        for i in range(4):
            e = torch.zeros(4, dtype=torch.float64)
            e[i] = 1e-5
            vp, _, _ = O.mll_value_and_grad_raw(X, y, raw + e, kind, bounds)
            vm, _, _ = O.mll_value_and_grad_raw(X, y, raw - e, kind, bounds)
            np.testing.assert_allclose(float(g[i]), float(vp - vm) / 2e-5, rtol=2e-5, atol=1e-9)


def test_target_posterior_reduces_to_plain_gp_without_sources():
    # zero source prior -> the target model is an ordinary GP on standardised targets
    X, y, theta = _problem(N=20, D=2, seed=2)
    xq = torch.rand(5, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    n, M = 20, 5
    mu, S = O.target_posterior(xq, X, y, torch.zeros(n + M, dtype=torch.float64), torch.zeros(n + M, n + M, dtype=torch.float64),
                               theta, O.KIND_RBF, 0.0, 1.0)
    out = O.gp_fit(X, y, theta, O.KIND_RBF)
    mu2, S2 = O.source_posterior(xq, X, theta, O.KIND_RBF, out["L"], out["alpha"], 0.0, 1.0)
    torch.testing.assert_close(mu, mu2, rtol=1e-8, atol=1e-10)
    torch.testing.assert_close(S, S2, rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    kind = int(g["kind"])
    X, y, theta = (torch.from_numpy(g[k]) for k in ("X", "y", "theta"))
    xq = torch.from_numpy(g["xq"])
    for t in range(X.shape[0]):
        n = int(g["n_points"][t])
        out = O.gp_fit(X[t, :n], y[t, :n], theta[t], kind)
        np.testing.assert_allclose(out["L"].numpy(), g["L"][t, :n, :n], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(float(out["mll"]), g["mll"][t], rtol=1e-9)
        assert float(out["jitter"]) == g["jitter"][t]
        mu, cov = O.source_posterior(xq, X[t, :n], theta[t], kind, out["L"], out["alpha"], float(g["y_mean"][t]), float(g["y_std"][t]))
        np.testing.assert_allclose(mu.numpy(), g["post_mean"][t], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(cov.numpy(), g["post_cov"][t], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_independent_columns(path):
    """NOT circular: the expected values are the fixture's scikit-learn / scipy columns (tests/golden/make_golden.py,
    ``independent_columns`` -- no oracle code), the checked values come from the oracle now.  Covers RBF and
    Matern-5/2 with ARD, the ragged stack, N = 2 / constant-Y tasks, the reference-held inputs (META_DATA_1D +
    quartic, Forrester family) and, through scipy on scikit-learn's kernel matrix, the jitter-rescued tasks."""
    g = np.load(path)
    kind = int(g["kind"])
    X, y, theta = (torch.from_numpy(g[k]) for k in ("X", "y", "theta"))
    xq = torch.from_numpy(g["xq"])
    for t in range(X.shape[0]):
        n = int(g["n_points"][t])
        out = O.gp_fit(X[t, :n], y[t, :n], theta[t], kind)
        jit = float(g["jitter"][t])
        tight = jit == 0.0
        scale = np.abs(g["sp_L"][t]).max()
        # scipy factor / solve of scikit-learn's kernel matrix (+ the recorded jitter)
        np.testing.assert_allclose(out["L"].numpy(), g["sp_L"][t, :n, :n], rtol=0, atol=(1e-9 if tight else 1e-4) * scale)
        np.testing.assert_allclose(float(out["logdet"]), g["sp_logdet"][t], rtol=1e-9 if tight else 1e-3, atol=1e-10)
        if tight:
            a_scale = np.abs(g["sp_alpha"][t]).max()
            np.testing.assert_allclose(out["alpha"].numpy(), g["sp_alpha"][t, :n], rtol=0, atol=1e-6 * a_scale)
        if np.isfinite(g["sk_lml"][t]):
            # scikit-learn's own factor, weights, log marginal likelihood (un-normalised, prior-free) and predictions
            np.testing.assert_allclose(out["L"].numpy(), g["sk_L"][t, :n, :n], rtol=0, atol=1e-9 * scale)
            np.testing.assert_allclose(out["alpha"].numpy(), g["sk_alpha"][t, :n], rtol=0, atol=1e-6 * np.abs(g["sk_alpha"][t]).max())
            np.testing.assert_allclose(float(out["mll"]) * n, g["sk_lml"][t], rtol=1e-9, atol=1e-9)
            mu, cov = O.source_posterior(xq, X[t, :n], theta[t], kind, out["L"], out["alpha"], float(g["y_mean"][t]), float(g["y_std"][t]))
            np.testing.assert_allclose(mu.numpy(), g["sk_post_mean"][t], rtol=0, atol=1e-7 * max(np.abs(g["sk_post_mean"][t]).max(), 1e-300))
            np.testing.assert_allclose(cov.numpy(), g["sk_post_cov"][t], rtol=0, atol=1e-6 * np.abs(g["sk_post_cov"][t]).max())


def test_loop_and_batched_stack_agree():
    g = np.load([p for p in GOLDEN if "c3r" in p][0])
    X, y, theta = (torch.from_numpy(g[k]) for k in ("X", "y", "theta"))
    a = O.gp_fit_stack_loop(X, y, theta, int(g["kind"]))
    b = O.gp_fit_stack_batched(X, y, theta, int(g["kind"]))
    for k in ("L", "alpha", "mll", "logdet", "quad"):
        torch.testing.assert_close(a[k], b[k], rtol=1e-7, atol=1e-10)
