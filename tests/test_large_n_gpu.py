"""GPU parity tests beyond the register-resident limit: the two-block fused fit for 256 < N <= 512
(BASELINE config 5: T=32, N=512, D=6), the batched Cholesky solve (scaml_cho_solve_batched_f64) and
per-task query sets / mean-only mode of scaml_posterior_batched_f64.  Tolerances: 1e-4 relative on
alpha / posterior moments, 1e-3 on the MLL (BASELINE.json north_star); the kernels are far inside."""
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import ops

pytestmark = pytest.mark.gpu


def _stack(T, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.sin(3.0 * X.sum(-1)) + 0.1 * torch.randn(T, N, dtype=torch.float64, generator=g)
    y = (y - y.mean(-1, keepdim=True)) / y.std(-1, keepdim=True)
    theta = torch.cat([0.4 + torch.rand(T, D, dtype=torch.float64, generator=g),
                       0.5 + torch.rand(T, 1, dtype=torch.float64, generator=g),
                       1e-3 + 1e-2 * torch.rand(T, 1, dtype=torch.float64, generator=g)], 1)
    return X, y, theta


@pytest.mark.parametrize("kind", [O.KIND_RBF, O.KIND_MATERN52])
@pytest.mark.parametrize("N", [257, 384, 512])
def test_two_block_fit_matches_oracle(N, kind, device):
    T, D = 3, 6
    X, y, theta = _stack(T, N, D, N + kind)
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, want_linv=True)
    ref = O.gp_fit_stack_loop(X, y, theta, kind)
    assert not out["info"].cpu().any()
    assert out["jitter"].cpu().tolist() == ref["jitter"].tolist()
    torch.testing.assert_close(out["L"].cpu(), ref["L"], rtol=1e-7, atol=1e-9)
    torch.testing.assert_close(out["alpha"].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["quad"].cpu(), ref["quad"], rtol=1e-7, atol=1e-8)
    torch.testing.assert_close(out["logdet"].cpu(), ref["logdet"], rtol=1e-9, atol=1e-8)
    torch.testing.assert_close(out["mll"].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)
    # the composite's factors feed the posterior and gradient kernels unchanged
    Xq = torch.rand(33, D, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    post = ops.source_posteriors(Xq.to(device), X.to(device), theta.to(device), kind, out["L"], out["Linv_diag"], out["alpha"])
    for t in range(T):
        mu, cov = O.source_posterior(Xq, X[t], theta[t], kind, ref["L"][t], ref["alpha"][t], 0.0, 1.0)
        torch.testing.assert_close(post["mean"][t].cpu(), mu, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(post["var"][t].cpu(), torch.diagonal(cov), rtol=1e-4, atol=1e-7)


def test_two_block_fit_ragged(device):
    T, N, D, kind = 4, 512, 4, O.KIND_MATERN52
    X, y, theta = _stack(T, N, D, 11)
    n = torch.tensor([512, 300, 256, 100], dtype=torch.int32)
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, n_points=n.to(device))
    assert not out["info"].cpu().any()
    for t in range(T):
        k = int(n[t])
        ref = O.gp_fit(X[t, :k], y[t, :k], theta[t], kind)
        torch.testing.assert_close(out["L"][t, :k, :k].cpu(), ref["L"], rtol=1e-7, atol=1e-9)
        torch.testing.assert_close(out["alpha"][t, :k].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)
        assert float(out["alpha"][t, k:].abs().sum()) == 0.0
        torch.testing.assert_close(out["mll"][t].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)


def test_two_block_fit_jitter_ladder(device):
    # duplicated points + a (slightly) negative "noise": the whole 512-matrix needs the ladder, applied
    # to failing tasks only, with one jitter value for both blocks (psd_safe_cholesky semantics)
    T, N, D, kind = 3, 320, 3, O.KIND_RBF
    X, y, theta = _stack(T, N, D, 3)
    X[1, 300:] = X[1, :20]
    theta[1, D + 1] = -2e-9
    X[2, 10:40] = X[2, 280:310]
    theta[2, D + 1] = -5e-8
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind)
    ref = O.gp_fit_stack_loop(X, y, theta, kind)
    assert out["jitter"].cpu().tolist() == ref["jitter"].tolist()
    assert ref["jitter"].tolist()[0] == 0.0 and min(ref["jitter"].tolist()[1:]) > 0.0
    assert not out["info"].cpu().any()
    torch.testing.assert_close(out["logdet"][0].cpu(), ref["logdet"][0], rtol=1e-9, atol=1e-8)
    one = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, retry=False)
    assert (one["info"].cpu() > 0).tolist() == [False, True, True]


@pytest.mark.parametrize("N,R", [(16, 1), (100, 5), (256, 40), (512, 17)])
def test_cho_solve_matches_torch(N, R, device):
    g = torch.Generator().manual_seed(N + R)
    T = 3
    Bm = torch.randn(T, N, N, dtype=torch.float64, generator=g)
    A = Bm @ Bm.transpose(-1, -2) / N + torch.eye(N, dtype=torch.float64)
    rhs = torch.randn(T, N, R, dtype=torch.float64, generator=g)
    L = torch.linalg.cholesky(A)
    if N <= 256:
        f = ops.potrf_batched(A.to(device), want_linv=True)
        Ld, W = f["L"], f["Linv_diag"]
    else:  # factors from elsewhere: build the diagonal-block inverses on the host
        nb = N // 16
        W = torch.stack([torch.linalg.inv(L[:, 16 * b:16 * b + 16, 16 * b:16 * b + 16]) for b in range(nb)], 1).to(device)
        Ld = L.to(device)
    got = ops.cho_solve(Ld, W, rhs.to(device)).cpu()
    torch.testing.assert_close(got, torch.cholesky_solve(rhs, L), rtol=1e-8, atol=1e-10)
    # the backward half alone (scaml_solve_lt_batched_f64): L^T x = b
    got_t = ops.cho_solve(Ld, W, rhs.to(device), backward_only=True).cpu()
    torch.testing.assert_close(got_t, torch.linalg.solve_triangular(L.transpose(-1, -2), rhs, upper=True), rtol=1e-8, atol=1e-10)
    n = torch.tensor([N, max(N - 7, 1), 1], dtype=torch.int32)
    if N <= 256:
        f = ops.potrf_batched(A.to(device), n_points=n.to(device), want_linv=True)
        got = ops.cho_solve(f["L"], f["Linv_diag"], rhs.to(device), n_points=n.to(device)).cpu()
        for t in range(T):
            k = int(n[t])
            ref = torch.cholesky_solve(rhs[t, :k], torch.linalg.cholesky(A[t, :k, :k]))
            torch.testing.assert_close(got[t, :k], ref, rtol=1e-8, atol=1e-10)
            assert float(got[t, k:].abs().sum()) == 0.0
        got_t = ops.cho_solve(f["L"], f["Linv_diag"], rhs.to(device), n_points=n.to(device), backward_only=True).cpu()
        for t in range(T):
            k = int(n[t])
            Lk = torch.linalg.cholesky(A[t, :k, :k])
            torch.testing.assert_close(got_t[t, :k], torch.linalg.solve_triangular(Lk.T, rhs[t, :k], upper=True), rtol=1e-8, atol=1e-10)
            assert float(got_t[t, k:].abs().sum()) == 0.0


def test_posterior_per_task_queries_and_mean_only(device):
    T, N, D, M, kind = 3, 64, 4, 21, O.KIND_MATERN52
    X, y, theta = _stack(T, N, D, 2)
    Xq = torch.rand(T, M, D, dtype=torch.float64, generator=torch.Generator().manual_seed(9))
    f = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, want_linv=True)
    post = ops.source_posteriors(Xq.to(device), X.to(device), theta.to(device), kind, f["L"], f["Linv_diag"], f["alpha"], cov_first=M)
    mo = ops.source_posteriors(Xq.to(device), X.to(device), theta.to(device), kind, None, None, f["alpha"], mean_only=True)
    ref = O.gp_fit_stack_loop(X, y, theta, kind)
    for t in range(T):
        mu, cov = O.source_posterior(Xq[t], X[t], theta[t], kind, ref["L"][t], ref["alpha"][t], 0.0, 1.0)
        torch.testing.assert_close(post["mean"][t].cpu(), mu, rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(mo["mean"][t].cpu(), mu, rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(post["cov"][t].cpu(), cov, rtol=1e-4, atol=1e-7)
