"""GPU parity tests for the stand-alone kernel matrix (scaml_kernel_matrix_f64) and the batched
jittered Cholesky of given matrices (scaml_potrf_batched_f64)."""
import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", [O.KIND_RBF, O.KIND_MATERN52])
def test_kernel_matrix_matches_oracle(kind, device):
    g = torch.Generator().manual_seed(0)
    T, N1, N2, D = 3, 37, 50, 4
    X1 = torch.rand(T, N1, D, dtype=torch.float64, generator=g)
    X2 = torch.rand(T, N2, D, dtype=torch.float64, generator=g)
    Xs = torch.rand(N2, D, dtype=torch.float64, generator=g)
    theta = torch.cat([0.3 + torch.rand(T, D, dtype=torch.float64, generator=g), 0.5 + torch.rand(T, 1, dtype=torch.float64, generator=g),
                       torch.full((T, 1), 1e-3, dtype=torch.float64)], 1)
    Ksq = ops.kernel_matrix(X1.to(device), theta.to(device), kind, add_noise=True).cpu()
    Kx = ops.kernel_matrix(X1.to(device), theta.to(device), kind, X2=X2.to(device)).cpu()
    Ks = ops.kernel_matrix(X1.to(device), theta.to(device), kind, X2=Xs.to(device)).cpu()
    for t in range(T):
        ls, os_, noise = theta[t, :D], theta[t, D], theta[t, D + 1]
        ref = O.kernel_matrix(X1[t], None, ls, os_, kind) + noise * torch.eye(N1, dtype=torch.float64)
        torch.testing.assert_close(Ksq[t], ref, rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(Kx[t], O.kernel_matrix(X1[t], X2[t], ls, os_, kind), rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(Ks[t], O.kernel_matrix(X1[t], Xs, ls, os_, kind), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("N", [5, 32, 100, 256])
def test_potrf_batched_matches_torch(N, device):
    g = torch.Generator().manual_seed(N)
    T = 4
    B = torch.randn(T, N, N, dtype=torch.float64, generator=g)
    A = B @ B.transpose(-1, -2) / N + torch.eye(N, dtype=torch.float64)
    y = torch.randn(T, N, dtype=torch.float64, generator=g)
    out = ops.potrf_batched(torch.tril(A).to(device), y.to(device))   # only the lower triangle is read
    assert not out["info"].cpu().any() and float(out["jitter"].abs().max()) == 0.0
    L = torch.linalg.cholesky(A)
    torch.testing.assert_close(out["L"].cpu(), L, rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(out["alpha"].cpu(), torch.cholesky_solve(y.unsqueeze(-1), L).squeeze(-1), rtol=1e-8, atol=1e-10)
    torch.testing.assert_close(out["logdet"].cpu(), torch.logdet(A), rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(out["quad"].cpu(), (y * torch.cholesky_solve(y.unsqueeze(-1), L).squeeze(-1)).sum(-1), rtol=1e-8, atol=1e-10)


def test_potrf_batched_jitter_escalation_matches_psd_safe_cholesky(device):
    g = torch.Generator().manual_seed(1)
    B = torch.rand(3, 24, 8, dtype=torch.float64, generator=g)
    A = B @ B.transpose(-1, -2)                       # rank 8 of 24: singular
    A[0] += 1e-3 * torch.eye(24, dtype=torch.float64)  # fine
    A[1] -= 1e-9 * torch.eye(24, dtype=torch.float64)  # needs 1e-8
    A[2] -= 5e-8 * torch.eye(24, dtype=torch.float64)  # needs 1e-7
    out = ops.potrf_batched(A.to(device))
    L_ref, _, jit_ref = O.psd_safe_cholesky(A)
    assert out["jitter"].cpu().tolist() == jit_ref.tolist() == [0.0, 1e-8, 1e-7]
    assert not out["info"].cpu().any()
    torch.testing.assert_close(out["L"][0].cpu(), L_ref[0], rtol=1e-9, atol=1e-11)
    # jitter-rescued members: compare the reconstructed matrix (the factor itself is ill-conditioned)
    for t in (1, 2):
        Lt = out["L"][t].cpu()
        torch.testing.assert_close(Lt @ Lt.T, A[t] + float(jit_ref[t]) * torch.eye(24, dtype=torch.float64), rtol=0, atol=1e-10)
    bad = ops.potrf_batched((A - 1e-2 * torch.eye(24, dtype=torch.float64)).to(device))
    assert bool((bad["info"].cpu() > 0).all())
