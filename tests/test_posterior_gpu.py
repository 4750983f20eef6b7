"""GPU parity tests for the batched source posterior (scaml_posterior_batched_f64 /
scaml_posterior_cov_f64) and the weighted task sum (scaml_weighted_task_sum_f64).
Tolerance (north_star): 1e-4 relative on posterior mean / variance."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import _lib, ops, synthetic

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
RTOL = 1e-4


def _fit(X, y, theta, kind, device, n_points=None):
    return ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, n_points=n_points, want_linv=True)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_posterior_matches_golden(path, device):
    g = np.load(path)
    if (g["jitter"] > 0).any():
        pytest.skip("jitter-rescued tasks are ill-conditioned by construction; covered by the fit tests")
    kind = int(g["kind"])
    X, y, theta, xq = (torch.from_numpy(g[k]) for k in ("X", "y", "theta", "xq"))
    ragged = bool((g["n_points"] != X.shape[1]).any())
    npts = torch.from_numpy(g["n_points"]).to(device) if ragged else None
    fit = _fit(X, y, theta, kind, device, npts)
    M = xq.shape[0]
    post = ops.source_posteriors(xq.to(device), X.to(device), theta.to(device), kind, fit["L"], fit["Linv_diag"], fit["alpha"],
                                 torch.from_numpy(g["y_mean"]).to(device), torch.from_numpy(g["y_std"]).to(device),
                                 n_points=npts, cov_first=M)
    mean, var, cov = (post[k].cpu().numpy() for k in ("mean", "var", "cov"))
    scale = np.abs(g["post_cov"]).max(axis=(1, 2), keepdims=True)
    np.testing.assert_allclose(mean, g["post_mean"], rtol=RTOL, atol=RTOL * np.abs(g["post_mean"]).max())
    np.testing.assert_allclose(cov, g["post_cov"], rtol=0, atol=RTOL * scale.max())
    np.testing.assert_allclose(var, np.diagonal(g["post_cov"], axis1=1, axis2=2), rtol=0, atol=RTOL * scale.max())


@pytest.mark.parametrize("T,N,D,M,kind", [
    (3, 32, 2, 7, O.KIND_RBF),
    (4, 100, 5, 33, O.KIND_MATERN52),
    (2, 256, 8, 80, O.KIND_MATERN52),
    (5, 128, 3, 200, O.KIND_RBF),
])
def test_posterior_matches_oracle(T, N, D, M, kind, device):
    d = synthetic.smooth_field_task_stack(T, N, D, seed=3 + N)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    rng = np.random.default_rng(N)
    theta = torch.from_numpy(np.concatenate([0.5 * (1 + 0.4 * (rng.uniform(size=(T, D)) - 0.5)), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1))
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    xq = torch.from_numpy(rng.uniform(size=(M, D)))
    fit = _fit(X, y, theta, kind, device)
    Ma = min(M, 20)
    post = ops.source_posteriors(xq.to(device), X.to(device), theta.to(device), kind, fit["L"], fit["Linv_diag"], fit["alpha"],
                                 torch.from_numpy(m).to(device), torch.from_numpy(s).to(device), cov_first=Ma)
    for t in range(T):
        ref = O.gp_fit(X[t], y[t], theta[t], kind)
        mu, cov = O.source_posterior(xq, X[t], theta[t], kind, ref["L"], ref["alpha"], float(m[t]), float(s[t]))
        scale = float(cov.abs().max())
        torch.testing.assert_close(post["mean"][t].cpu(), mu, rtol=RTOL, atol=RTOL * float(mu.abs().max()))
        torch.testing.assert_close(post["var"][t].cpu(), cov.diagonal(), rtol=0, atol=RTOL * scale)
        torch.testing.assert_close(post["cov"][t].cpu(), cov[:Ma], rtol=0, atol=RTOL * scale)
        # tight check against the same distance formulation
        mu_d, cov_d = O.source_posterior(xq, X[t], theta[t], kind, ref["L"], ref["alpha"], float(m[t]), float(s[t]), dist="direct")
        torch.testing.assert_close(post["mean"][t].cpu(), mu_d, rtol=1e-8, atol=1e-9 * float(mu.abs().max()))


def test_posterior_at_training_points_collapses(device):
    # at the training inputs the posterior variance is bounded by the noise level and the mean
    # reproduces the (standardised) data up to the noise shrinkage
    T, N, D = 2, 64, 2
    d = synthetic.branin_task_stack(T, N, seed=11, noise_std=0.0)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    theta = torch.tensor([[0.3, 0.3, 1.0, 1e-6]] * T, dtype=torch.float64)
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    fit = _fit(X, y, theta, O.KIND_RBF, device)
    post = ops.source_posteriors(X[0].to(device), X.to(device), theta.to(device), O.KIND_RBF, fit["L"], fit["Linv_diag"], fit["alpha"])
    assert float(post["var"][0].max()) < 1e-5 and float(post["var"][0].min()) > -1e-9
    torch.testing.assert_close(post["mean"][0].cpu(), y[0], rtol=0, atol=2e-2)


def test_weighted_task_sum_and_pruning_mask(device):
    T, M = 6, 37
    g = torch.Generator().manual_seed(0)
    mus = torch.randn(T, M, dtype=torch.float64, generator=g)
    covs = torch.randn(T, 5, M, dtype=torch.float64, generator=g)
    w = torch.tensor([0.5, 1e-7, 0.3, 0.0, 0.9, 0.2], dtype=torch.float64)
    stds = torch.tensor([1.0, 2.0, 0.5, 1.0, 1.5, 0.1], dtype=torch.float64)
    mask = O.significant_weights_mask(w, stds, 1e-3)
    mu_ref, cov_ref = O.target_prior(mus, covs.reshape(T, -1), w, mask)
    mu = ops.weighted_task_sum(mus.to(device), w.to(device), 1, mask.to(device))
    cov = ops.weighted_task_sum(covs.to(device), w.to(device), 2, mask.to(device))
    torch.testing.assert_close(mu.cpu(), mu_ref, rtol=1e-14, atol=1e-15)
    torch.testing.assert_close(cov.cpu().reshape(-1), cov_ref, rtol=1e-14, atol=1e-15)
    torch.testing.assert_close(ops.weighted_task_sum(mus.to(device), w.to(device)).cpu(), (w[:, None] * mus).sum(0))


@pytest.mark.parametrize("T,N,D,M,kind", [(3, 48, 3, 21, O.KIND_RBF), (2, 100, 5, 40, O.KIND_MATERN52), (2, 256, 8, 33, O.KIND_MATERN52),
                                           (2, 400, 6, 17, O.KIND_RBF)])
def test_posterior_from_explicit_inverse_matches_substitution_and_oracle(T, N, D, M, kind, device):
    """scaml_linv_batched_f64 + scaml_posterior_linv_f64 against the substitution kernel and the oracle."""
    g = torch.Generator().manual_seed(N + M)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.randn(T, N, dtype=torch.float64, generator=g)
    theta = torch.cat([0.4 + torch.rand(T, D, dtype=torch.float64, generator=g), 0.5 + torch.rand(T, 1, dtype=torch.float64, generator=g),
                       torch.full((T, 1), 2e-3, dtype=torch.float64)], 1)
    xq = torch.rand(M, D, dtype=torch.float64, generator=g)
    n = torch.tensor([N, max(N - 9, 1)] + [N] * (T - 2), dtype=torch.int32)
    ym = torch.randn(T, dtype=torch.float64, generator=g)
    ysd = 0.5 + torch.rand(T, dtype=torch.float64, generator=g)
    fit = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, n_points=n.to(device), want_linv=True)
    Linv = ops.linv_batched(fit["L"], fit["Linv_diag"], n_points=n.to(device))
    args = (xq.to(device), X.to(device), theta.to(device), kind, fit["L"], fit["Linv_diag"], fit["alpha"], ym.to(device), ysd.to(device))
    a = ops.source_posteriors(*args, n_points=n.to(device), cov_first=M, keep_V=True)
    b = ops.source_posteriors(*args, n_points=n.to(device), cov_first=M, keep_V=True, Linv=Linv)
    for k in ("mean", "var", "cov", "V"):
        torch.testing.assert_close(b[k], a[k], rtol=1e-8, atol=1e-10)
    for t in range(T):
        k = int(n[t])
        ref = O.gp_fit(X[t, :k], y[t, :k], theta[t], kind)
        torch.testing.assert_close(torch.tril(Linv[t, :k, :k].cpu()), torch.linalg.inv(ref["L"]), rtol=1e-6, atol=1e-8)
        mu, cov = O.source_posterior(xq, X[t, :k], theta[t], kind, ref["L"], ref["alpha"], float(ym[t]), float(ysd[t]))
        torch.testing.assert_close(b["mean"][t].cpu(), mu, rtol=RTOL, atol=1e-7)
        torch.testing.assert_close(b["cov"][t].cpu(), cov, rtol=RTOL, atol=1e-7)
    # per-task query sets go through the same kernel
    xqt = torch.rand(T, M, D, dtype=torch.float64, generator=g)
    c = ops.source_posteriors(xqt.to(device), *args[1:], n_points=n.to(device))
    d = ops.source_posteriors(xqt.to(device), *args[1:], n_points=n.to(device), Linv=Linv)
    torch.testing.assert_close(d["mean"], c["mean"], rtol=1e-8, atol=1e-10)
    torch.testing.assert_close(d["var"], c["var"], rtol=1e-8, atol=1e-10)


def test_weighted_prior_reduce_matches_oracle(device):
    g = torch.Generator().manual_seed(4)
    T, M, Ma = 7, 19, 5
    mu = torch.randn(T, M, dtype=torch.float64, generator=g)
    cov = torch.randn(T, Ma, M, dtype=torch.float64, generator=g)
    w = torch.rand(T, dtype=torch.float64, generator=g)
    active = torch.tensor([1, 1, 0, 1, 0, 1, 1], dtype=torch.bool)
    mu_s, cov_s = ops.weighted_prior_reduce(mu.to(device), cov.to(device), w.to(device), active.to(device))
    wa = w * active
    torch.testing.assert_close(mu_s.cpu(), (wa[:, None] * mu).sum(0), rtol=1e-12, atol=1e-14)
    torch.testing.assert_close(cov_s.cpu(), ((wa ** 2)[:, None, None] * cov).sum(0), rtol=1e-12, atol=1e-14)
    mu_only, none = ops.weighted_prior_reduce(mu.to(device), None, w.to(device))
    assert none is None
    torch.testing.assert_close(mu_only.cpu(), (w[:, None] * mu).sum(0), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("T,N,D,M,Ma,kind,ragged,per_task", [
    (3, 200, 4, 300, 80, O.KIND_MATERN52, False, False),   # the BO scoring shape: n = 80 training points in front of the candidates
    (4, 96, 2, 40, 33, O.KIND_RBF, True, False),            # ragged stack, Ma not a multiple of 16
    (2, 512, 6, 130, 96, O.KIND_MATERN52, False, False),    # N = 512 sources, the largest fused block (96 leading points)
    (3, 64, 3, 50, 17, O.KIND_RBF, False, True),            # per-task query sets
])
def test_fused_covariance_block_matches_unfused_and_oracle(T, N, D, M, Ma, kind, ragged, per_task, device):
    """scaml_posterior_linv_cov_f64 (the covariance block out of the posterior pass, V never stored) against the
    V-in-memory path (scaml_posterior_linv_f64 + scaml_posterior_cov_f64) and, per task, against the oracle."""
    d = synthetic.smooth_field_task_stack(T, N, D, seed=11 + N)
    rng = np.random.default_rng(N + M)
    npts = [N] * T
    if ragged:
        npts = [N, N - 37, 5, N - 1][:T]
    ys = np.zeros((T, N))
    ym, ysd = np.zeros(T), np.ones(T)
    for t in range(T):
        yy, mm, ss = synthetic.standardize_rows(d["Y"][t:t + 1, :npts[t]])
        ys[t, :npts[t]], ym[t], ysd[t] = yy[0], mm[0], ss[0]
    theta = torch.from_numpy(np.concatenate([0.5 * (1 + 0.4 * (rng.uniform(size=(T, D)) - 0.5)), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1))
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    xq = torch.from_numpy(rng.uniform(size=(T, M, D) if per_task else (M, D)))
    n_dev = torch.tensor(npts, dtype=torch.int32, device=device) if ragged else None
    fit = _fit(X, y, theta, kind, device, n_dev)
    Linv = ops.linv_batched(fit["L"], fit["Linv_diag"], n_points=n_dev)
    args = (xq.to(device), X.to(device), theta.to(device), kind, None, None, fit["alpha"], torch.from_numpy(ym).to(device),
            torch.from_numpy(ysd).to(device))
    fused = ops.source_posteriors(*args, n_points=n_dev, cov_first=Ma, Linv=Linv)
    assert fused["V"] is None
    plain = ops.source_posteriors(*args, n_points=n_dev, cov_first=Ma, Linv=Linv, keep_V=True)   # keep_V: the V-in-memory path
    scale = float(plain["cov"].abs().max())
    torch.testing.assert_close(fused["mean"], plain["mean"], rtol=1e-12, atol=1e-13)
    torch.testing.assert_close(fused["var"], plain["var"], rtol=1e-10, atol=1e-12 * scale)
    torch.testing.assert_close(fused["cov"], plain["cov"], rtol=0, atol=1e-11 * scale)
    for t in range(T):
        n = npts[t]
        ref = O.gp_fit(X[t, :n], y[t, :n], theta[t], kind)
        xt = xq[t] if per_task else xq
        mu, cov = O.source_posterior(xt, X[t, :n], theta[t], kind, ref["L"], ref["alpha"], float(ym[t]), float(ysd[t]))
        torch.testing.assert_close(fused["mean"][t].cpu(), mu, rtol=RTOL, atol=RTOL * float(mu.abs().max()))
        torch.testing.assert_close(fused["cov"][t].cpu(), cov[:Ma], rtol=0, atol=RTOL * float(cov.abs().max()))


def test_linv_lower_only_matches_the_dense_inverse_where_it_is_read(device):
    """scaml_linv_batched_lower_f64 writes the block rows at or below each strip's diagonal block only: the lower triangle equals the
    dense variant's bit for bit, the rest of the buffer is left as it was, and the posteriors computed from it are unchanged."""
    T, N, D, kind = 3, 80, 3, O.KIND_MATERN52
    g = torch.Generator().manual_seed(12)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g).to(device)
    y = torch.randn(T, N, dtype=torch.float64, generator=g).to(device)
    theta = torch.tensor([[0.5] * D + [1.0, 1e-2]] * T, dtype=torch.float64, device=device)
    fit = ops.gp_fit_fused(X, y, theta, kind, want_linv=True)
    dense = ops.linv_batched(fit["L"], fit["Linv_diag"])
    sentinel = 123.456
    out = torch.full_like(dense, sentinel)
    rc = _lib.lib.scaml_linv_batched_lower_f64(fit["L"].data_ptr(), fit["Linv_diag"].data_ptr(), None, T, N, out.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.equal(torch.tril(out), torch.tril(dense))
    blk = torch.arange(N, device=device) // 16
    above = blk[:, None] < blk[None, :]          # block rows above the diagonal block of a column
    assert bool((out[:, above] == sentinel).all())
    xq = torch.rand(20, D, dtype=torch.float64, generator=g).to(device)
    a = ops.source_posteriors(xq, X, theta, kind, fit["L"], fit["Linv_diag"], fit["alpha"], Linv=dense, cov_first=5)
    b = ops.source_posteriors(xq, X, theta, kind, fit["L"], fit["Linv_diag"], fit["alpha"], Linv=out, cov_first=5)
    for k in ("mean", "var", "cov"):   # (the pass adds its wave partials with LDS floating-point atomics: equal to the last bits, not bitwise)
        torch.testing.assert_close(a[k], b[k], rtol=1e-12, atol=1e-14)
