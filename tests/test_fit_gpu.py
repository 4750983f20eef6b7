"""GPU parity tests for the fused fit kernel (scaml_gp_fit_fused_f64) through the C ABI.

Tolerances (BASELINE.json north_star): 1e-4 relative on posterior mean/variance and alpha,
1e-3 relative on the marginal likelihood, against the oracle that restates the reference's
gpytorch op sequence.  Against the oracle evaluated with the kernels' own distance
formulation ("direct") the factorisation itself is held to 1e-9.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import ops, synthetic

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
RTOL_MLL = 1e-3   # north_star tolerance on the marginal likelihood
RTOL_POST = 1e-4  # north_star tolerance on alpha / posterior mean / variance


def _stack(T, N, D, seed, ls=0.5, noise=1e-3, spread=0.4):
    d = synthetic.smooth_field_task_stack(T, N, D, seed=seed)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    rng = np.random.default_rng(seed + 1)
    theta = np.concatenate([ls * (1 + spread * (rng.uniform(size=(T, D)) - 0.5)), np.full((T, 1), 1.0), np.full((T, 1), noise)], 1)
    return torch.from_numpy(d["X"]), torch.from_numpy(ys), torch.from_numpy(theta)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_fit_matches_golden(path, device):
    g = np.load(path)
    kind = int(g["kind"])
    X, y, theta = (torch.from_numpy(g[k]).to(device) for k in ("X", "y", "theta"))
    npts = torch.from_numpy(g["n_points"]).to(device)
    ragged = bool((g["n_points"] != X.shape[1]).any())
    out = ops.gp_fit_fused(X, y, theta, kind, n_points=npts if ragged else None)
    assert not out["info"].cpu().any()
    np.testing.assert_array_equal(out["jitter"].cpu().numpy(), g["jitter"])
    jittered = g["jitter"] > 0
    for t in range(X.shape[0]):
        n = int(g["n_points"][t])
        # a task rescued by jitter has condition number ~1e10: hold it to the north-star
        # tolerances only; well-posed tasks must agree much more tightly
        tight = not jittered[t]
        np.testing.assert_allclose(out["L"][t, :n, :n].cpu().numpy(), g["L"][t, :n, :n], rtol=0,
                                   atol=(1e-9 if tight else 1e-4) * np.abs(g["L"][t]).max())
        np.testing.assert_allclose(out["mll"][t].cpu().numpy(), g["mll"][t], rtol=1e-9 if tight else RTOL_MLL)
        np.testing.assert_allclose(out["logdet"][t].cpu().numpy(), g["logdet"][t], rtol=1e-9 if tight else RTOL_MLL)
        np.testing.assert_allclose(out["quad"][t].cpu().numpy(), g["quad"][t], rtol=1e-7 if tight else RTOL_MLL)
        if tight:
            np.testing.assert_allclose(out["alpha"][t, :n].cpu().numpy(), g["alpha"][t, :n], rtol=0,
                                       atol=RTOL_POST * 1e-2 * np.abs(g["alpha"][t]).max())
    # strict upper triangle is exactly zero, rows/cols beyond n_t untouched (zero-initialised)
    assert float(torch.triu(out["L"], diagonal=1).abs().max()) == 0.0


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_fit_and_posterior_match_independent_columns(path, device):
    """The HIP path against the fixtures' scikit-learn / scipy columns (derived without the oracle,
    tests/golden/make_golden.py): L, alpha, the log marginal likelihood and the predictive mean / covariance on
    c1 / c2r / c3r / c5r / ragged / N2 and the reference-held inputs; the jitter fixture against scipy's factor of
    scikit-learn's kernel matrix plus the recorded jitter.  North-star tolerances: 1e-4 alpha / posterior, 1e-3 MLL."""
    g = np.load(path)
    kind = int(g["kind"])
    X, y, theta = (torch.from_numpy(g[k]).to(device) for k in ("X", "y", "theta"))
    T, N, _ = X.shape
    npts = torch.from_numpy(g["n_points"]).to(device)
    ragged = bool((g["n_points"] != N).any())
    out = ops.gp_fit_fused(X, y, theta, kind, n_points=npts if ragged else None, want_linv=True)
    assert not out["info"].cpu().any()
    np.testing.assert_array_equal(out["jitter"].cpu().numpy(), g["jitter"])
    xq = torch.from_numpy(g["xq"]).to(device)
    M = xq.shape[0]
    post = ops.source_posteriors(xq, X, theta, kind, out["L"], out["Linv_diag"], out["alpha"], torch.from_numpy(g["y_mean"]).to(device),
                                 torch.from_numpy(g["y_std"]).to(device), n_points=npts if ragged else None, cov_first=M)
    for t in range(T):
        n = int(g["n_points"][t])
        tight = g["jitter"][t] == 0.0
        scale = np.abs(g["sp_L"][t]).max()
        np.testing.assert_allclose(out["L"][t, :n, :n].cpu().numpy(), g["sp_L"][t, :n, :n], rtol=0, atol=(1e-8 if tight else 1e-4) * scale)
        np.testing.assert_allclose(float(out["logdet"][t]), g["sp_logdet"][t], rtol=1e-8 if tight else RTOL_MLL, atol=1e-9)
        if tight:
            np.testing.assert_allclose(out["alpha"][t, :n].cpu().numpy(), g["sp_alpha"][t, :n], rtol=0,
                                       atol=RTOL_POST * np.abs(g["sp_alpha"][t]).max())
        if np.isfinite(g["sk_lml"][t]):
            np.testing.assert_allclose(out["L"][t, :n, :n].cpu().numpy(), g["sk_L"][t, :n, :n], rtol=0, atol=1e-8 * scale)
            np.testing.assert_allclose(out["alpha"][t, :n].cpu().numpy(), g["sk_alpha"][t, :n], rtol=0,
                                       atol=RTOL_POST * np.abs(g["sk_alpha"][t]).max())
            np.testing.assert_allclose(float(out["mll"][t]) * n, g["sk_lml"][t], rtol=RTOL_MLL)
            np.testing.assert_allclose(post["mean"][t].cpu().numpy(), g["sk_post_mean"][t], rtol=0,
                                       atol=RTOL_POST * max(np.abs(g["sk_post_mean"][t]).max(), 1e-300))
            np.testing.assert_allclose(post["cov"][t].cpu().numpy(), g["sk_post_cov"][t], rtol=0,
                                       atol=RTOL_POST * np.abs(g["sk_post_cov"][t]).max())
            np.testing.assert_allclose(post["var"][t].cpu().numpy(), np.diag(g["sk_post_cov"][t]), rtol=0,
                                       atol=RTOL_POST * np.abs(g["sk_post_cov"][t]).max())


@pytest.mark.parametrize("T,N,D,kind", [
    (4, 32, 2, O.KIND_RBF),          # BASELINE config 1 shape
    (64, 128, 2, O.KIND_RBF),        # config 2
    (5, 16, 1, O.KIND_MATERN52),
    (3, 17, 3, O.KIND_MATERN52),     # N not a multiple of 16
    (7, 100, 6, O.KIND_RBF),
    (6, 200, 8, O.KIND_MATERN52),
    (16, 256, 8, O.KIND_MATERN52),   # config 3 per-task shape
    (3, 255, 12, O.KIND_RBF),
])
def test_fit_matches_oracle(T, N, D, kind, device):
    X, y, theta = _stack(T, N, D, seed=100 + N + D)
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind)
    assert not out["info"].cpu().any()
    assert float(out["jitter"].abs().max()) == 0.0
    ref_g = O.gp_fit_stack_loop(X, y, theta, kind, dist="gpytorch")   # the reference's formulation
    ref_d = O.gp_fit_stack_loop(X, y, theta, kind, dist="direct")     # the kernel's formulation
    # north-star tolerances vs the reference formulation
    assert _rel(out["mll"].cpu(), ref_g["mll"]) < RTOL_MLL
    assert _rel(out["alpha"].cpu(), ref_g["alpha"]) < RTOL_POST
    # tight agreement vs the same formulation
    assert _rel(out["L"].cpu(), ref_d["L"]) < 1e-9
    assert _rel(out["mll"].cpu(), ref_d["mll"]) < 1e-10
    assert _rel(out["logdet"].cpu(), ref_d["logdet"]) < 1e-10
    assert _rel(out["alpha"].cpu(), ref_d["alpha"]) < 1e-6


def test_ill_conditioned_noise_floor(device):
    # conditioning stress of SURVEY §8(d): lengthscale 1.0, noise 1e-6
    X, y, theta = _stack(8, 128, 4, seed=9, ls=1.0, noise=1e-6, spread=0.0)
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), O.KIND_MATERN52)
    ref = O.gp_fit_stack_loop(X, y, theta, O.KIND_MATERN52)
    assert not out["info"].cpu().any()
    assert _rel(out["mll"].cpu(), ref["mll"]) < RTOL_MLL
    # alpha is ill-conditioned here; the quantity the model uses is K_* alpha
    xq = torch.rand(16, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    for t in range(8):
        Ks = O.kernel_matrix(xq, X[t], theta[t, :4], theta[t, 4], O.KIND_MATERN52)
        mu_gpu = Ks @ out["alpha"][t].cpu()
        mu_ref = Ks @ ref["alpha"][t]
        assert _rel(mu_gpu, mu_ref) < RTOL_POST


def test_mll_only_mode_and_retry_flag(device):
    X, y, theta = _stack(5, 64, 3, seed=4)
    full = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), O.KIND_RBF)
    lite = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), O.KIND_RBF, store_L=False, want_alpha=False, retry=False)
    assert lite["L"] is None and lite["alpha"] is None
    torch.testing.assert_close(full["mll"], lite["mll"], rtol=0, atol=0)
    torch.testing.assert_close(full["quad"], lite["quad"], rtol=0, atol=0)


def test_not_psd_reports_info_and_nan(device):
    X, y, theta = _stack(3, 48, 2, seed=5)
    theta[1, -1] = -0.5  # K - 0.5 I is indefinite: no jitter can rescue it
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), O.KIND_RBF)
    info = out["info"].cpu()
    assert info[0] == 0 and info[2] == 0 and 1 <= int(info[1]) <= 48
    assert torch.isnan(out["mll"][1]).item() and torch.isfinite(out["mll"][[0, 2]]).all().item()
    assert float(out["jitter"][1]) == 1e-6
    with pytest.raises(ops.NotPSDError):
        ops.raise_if_not_psd(out["info"])
    # NaN inputs fail the same way (psd_safe_cholesky raises on NaN)
    Xn = X.clone()
    Xn[2, 3, 0] = float("nan")
    out = ops.gp_fit_fused(Xn.to(device), y.to(device), theta.to(device), O.KIND_RBF)
    assert int(out["info"][2]) > 0


def test_order_invariance_to_point_shuffling(device):
    # property the reference pins at scamlgp/testing.py:99: results do not depend on the order
    # in which a task's points are listed (MLL exactly up to rounding; alpha permutes)
    X, y, theta = _stack(4, 96, 5, seed=6)
    perm = torch.randperm(96, generator=torch.Generator().manual_seed(0))
    a = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), O.KIND_MATERN52)
    b = ops.gp_fit_fused(X[:, perm].contiguous().to(device), y[:, perm].contiguous().to(device), theta.to(device), O.KIND_MATERN52)
    torch.testing.assert_close(a["mll"], b["mll"], rtol=1e-11, atol=0)
    torch.testing.assert_close(a["alpha"][:, perm], b["alpha"], rtol=1e-7, atol=1e-9)


def test_full_size_properties(device):
    # BASELINE headline size (T=256, N=256, D=8, Matern-5/2 ARD): too slow to loop the oracle over
    # every task, so check size-independent properties on all tasks + the oracle on a sample.
    T, N, D = 256, 256, 8
    X, y, theta = _stack(T, N, D, seed=1234)
    Xd, yd, td = X.to(device), y.to(device), theta.to(device)
    out = ops.gp_fit_fused(Xd, yd, td, O.KIND_MATERN52)
    assert not out["info"].cpu().any()
    L, alpha = out["L"], out["alpha"]
    # (1) L L^T reproduces K + noise I (K rebuilt by torch on the GPU, direct differences)
    a = Xd / td[:, None, :D]
    d2 = (a.unsqueeze(-2) - a.unsqueeze(-3)).pow(2).sum(-1)
    r = d2.clamp_min(1e-30).sqrt()
    K = td[:, D, None, None] * (1 + 5 ** 0.5 * r + 5.0 / 3.0 * d2) * torch.exp(-(5 ** 0.5) * r)
    K = K + td[:, D + 1, None, None] * torch.eye(N, dtype=torch.float64, device=device)
    resid = (L @ L.transpose(-1, -2) - K).abs().amax((-1, -2))
    assert float(resid.max()) < 1e-12
    # (2) K alpha = y
    assert float((torch.einsum("tij,tj->ti", K, alpha) - yd).abs().max()) < 1e-8
    # (3) scalars are consistent with L and alpha
    torch.testing.assert_close(out["logdet"], 2 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1), rtol=1e-12, atol=0)
    torch.testing.assert_close(out["quad"], (alpha * yd).sum(-1), rtol=1e-9, atol=0)
    # (4) oracle on a sample of tasks
    idx = [0, 101, 255]
    ref = O.gp_fit_stack_loop(X[idx], y[idx], theta[idx], O.KIND_MATERN52)
    assert _rel(out["mll"][idx].cpu(), ref["mll"]) < RTOL_MLL
    assert _rel(out["alpha"][idx].cpu(), ref["alpha"]) < RTOL_POST
    # (5) launches are deterministic
    out2 = ops.gp_fit_fused(Xd, yd, td, O.KIND_MATERN52)
    assert torch.equal(out["L"], out2["L"]) and torch.equal(out["mll"], out2["mll"])


def test_rejects_cpu_tensors_and_wrong_dtype(device):
    X, y, theta = _stack(2, 16, 2, seed=1)
    with pytest.raises(ValueError):
        ops.gp_fit_fused(X, y, theta, O.KIND_RBF)
    with pytest.raises(ValueError):
        ops.gp_fit_fused(X.to(device).float(), y.to(device), theta.to(device), O.KIND_RBF)
    with pytest.raises(ValueError):
        ops.gp_fit_fused(torch.rand(2, 600, 2, dtype=torch.float64, device=device), torch.rand(2, 600, dtype=torch.float64, device=device),
                         theta.to(device), O.KIND_RBF)   # beyond the two-block limit of 512 points


def test_narrow_and_wide_kernels_for_n_up_to_128_agree(device):
    # 64 < N <= 128 has two kernels: four waves per task (two workgroups share a CU) for stacks that fill the CUs,
    # eight waves per task otherwise (csrc/scaml_host.cpp).  Same stack through both; the first tasks vs the oracle.
    T, N, D = 300, 100, 5
    X, y, theta = _stack(T, N, D, seed=21)
    Xd, yd, td = X.to(device), y.to(device), theta.to(device)
    narrow = ops.gp_fit_fused(Xd, yd, td, O.KIND_MATERN52)                 # T > number of CUs
    wide = ops.gp_fit_fused(Xd[:40].contiguous(), yd[:40].contiguous(), td[:40].contiguous(), O.KIND_MATERN52)
    assert not narrow["info"].cpu().any() and not wide["info"].cpu().any()
    torch.testing.assert_close(narrow["L"][:40], wide["L"], rtol=1e-12, atol=1e-14)
    torch.testing.assert_close(narrow["alpha"][:40], wide["alpha"], rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(narrow["mll"][:40], wide["mll"], rtol=1e-12, atol=0)
    for t in (0, 39):
        ref = O.gp_fit(X[t], y[t], theta[t], O.KIND_MATERN52)
        torch.testing.assert_close(wide["alpha"][t].cpu(), ref["alpha"], rtol=1e-4, atol=1e-8)   # north_star: 1e-4
        torch.testing.assert_close(narrow["mll"][t].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)     # north_star: 1e-3
