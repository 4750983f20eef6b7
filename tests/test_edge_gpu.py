"""GPU parity tests of the fused fit at the edges of its shape space: single points, one task, many small
tasks (more workgroups than CUs), N off the 16-grid, empty and one-point ragged tasks, large D.
Held to 1e-9 on L / MLL and 1e-6 on alpha against the oracle evaluated with direct coordinate differences (the
kernel forms the squared distances in gpytorch's expanded form on the matrix core; north_star asks 1e-4 / 1e-3)."""
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import ops

pytestmark = pytest.mark.gpu

CASES = [
    (1, 16, 1, O.KIND_RBF, None),
    (1, 1, 1, O.KIND_MATERN52, None),
    (2, 17, 20, O.KIND_MATERN52, None),
    (2, 256, 30, O.KIND_RBF, None),
    (3, 255, 7, O.KIND_MATERN52, None),
    (4, 64, 3, O.KIND_RBF, [0, 1, 64, 33]),
    (2, 241, 2, O.KIND_MATERN52, [241, 240]),
    (300, 48, 2, O.KIND_RBF, None),
    (2, 129, 4, O.KIND_MATERN52, None),
]


@pytest.mark.parametrize("T,N,D,kind,npts", CASES)
def test_fit_edge_shapes(T, N, D, kind, npts, device):
    g = torch.Generator().manual_seed(1000 * N + D)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.randn(T, N, dtype=torch.float64, generator=g)
    theta = torch.cat([0.3 + torch.rand(T, D, dtype=torch.float64, generator=g) * D ** 0.5, torch.ones(T, 1, dtype=torch.float64),
                       torch.full((T, 1), 1e-3, dtype=torch.float64)], 1)
    n = None if npts is None else torch.tensor(npts, dtype=torch.int32)
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, n_points=None if n is None else n.to(device))
    assert not out["info"].cpu().any()
    for t in range(min(T, 12)):
        k = N if n is None else int(n[t])
        if k == 0:
            continue
        ref = O.gp_fit(X[t, :k], y[t, :k], theta[t], kind, dist="direct")
        torch.testing.assert_close(out["L"][t, :k, :k].cpu(), ref["L"], rtol=1e-9, atol=1e-11)
        torch.testing.assert_close(out["alpha"][t, :k].cpu(), ref["alpha"], rtol=1e-6, atol=1e-9)
        torch.testing.assert_close(out["mll"][t].cpu(), ref["mll"], rtol=1e-9, atol=1e-12)
        if k < N:   # rows / columns past n_t are never written (the caller's zeros stay)
            assert float(out["L"][t, k:, :].abs().sum()) == 0.0 and float(out["alpha"][t, k:].abs().sum()) == 0.0


@pytest.mark.parametrize("ls,noise,kind", [(1.0, 1e-6, O.KIND_MATERN52), (1.0, 1e-6, O.KIND_RBF), (2.0, 1e-8, O.KIND_RBF)])
def test_fit_conditioning_stress(ls, noise, kind, device):
    """"for a conditioning stress, l = 1.0, sigma^2 = 1e-6" (and harsher).  Against the oracle in the
    reference's own (gpytorch-style) distance formulation; north_star tolerances 1e-4 on alpha, 1e-3 on the MLL."""
    from scamlgp_amd import synthetic
    T, N, D = 4, 256, 8
    d = synthetic.smooth_field_task_stack(T, N, D, seed=3)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    theta = torch.cat([torch.full((T, D), ls), torch.ones(T, 1), torch.full((T, 1), noise)], 1).double()
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind)
    ref = O.gp_fit_stack_loop(X, y, theta, kind, dist="gpytorch")
    assert not out["info"].cpu().any() and out["jitter"].cpu().tolist() == ref["jitter"].tolist()
    torch.testing.assert_close(out["alpha"].cpu(), ref["alpha"], rtol=1e-4, atol=1e-4 * float(ref["alpha"].abs().max()))
    torch.testing.assert_close(out["mll"].cpu(), ref["mll"], rtol=1e-3, atol=0)
    torch.testing.assert_close(out["L"].cpu(), ref["L"], rtol=1e-6, atol=1e-9)


def test_largest_supported_dimension_and_beyond(device):
    # the staged point stack (D rounded up to 4, + 4 tail rows) has to fit the LDS next to the panels
    from scamlgp_amd import _lib
    N = 256
    dmax = _lib.lib.scaml_fit_max_d(N)
    assert dmax >= 32
    g = torch.Generator().manual_seed(11)
    for D, ok in ((dmax, True), (dmax + 1, False)):
        X = torch.rand(2, N, D, dtype=torch.float64, generator=g)
        y = torch.randn(2, N, dtype=torch.float64, generator=g)
        theta = torch.cat([torch.full((2, D), 0.5 * D ** 0.5), torch.ones(2, 1), torch.full((2, 1), 1e-3)], 1).double()
        if ok:
            out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), O.KIND_MATERN52)
            assert not out["info"].cpu().any()
            ref = O.gp_fit(X[1], y[1], theta[1], O.KIND_MATERN52, dist="direct")
            torch.testing.assert_close(out["L"][1].cpu(), ref["L"], rtol=1e-9, atol=1e-11)
            torch.testing.assert_close(out["mll"][1].cpu(), ref["mll"], rtol=1e-9, atol=1e-12)
        else:
            with pytest.raises(Exception):
                ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), O.KIND_MATERN52)


def test_expanded_distance_under_cancellation(device):
    # short lengthscales make |x'|^2 large against the distances of close points: the expanded form
    # |a|^2 + |b|^2 - 2 a.b loses digits there (as gpytorch's does); near-duplicates + exact duplicates included.
    # Tolerances: north_star's 1e-4 on alpha / 1e-3 on the MLL.
    T, N, D = 3, 96, 8
    g = torch.Generator().manual_seed(5)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    X[:, 1] = X[:, 0] + 1e-6      # near-duplicate
    X[:, 3] = X[:, 2]             # exact duplicate
    y = torch.randn(T, N, dtype=torch.float64, generator=g)
    for ls in (0.02, 0.2):
        theta = torch.cat([torch.full((T, D), ls), torch.ones(T, 1), torch.full((T, 1), 1e-3)], 1).double()
        for kind in (O.KIND_MATERN52, O.KIND_RBF):
            out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind)
            assert not out["info"].cpu().any()
            for t in range(T):
                ref = O.gp_fit(X[t], y[t], theta[t], kind, dist="direct")
                torch.testing.assert_close(out["alpha"][t].cpu(), ref["alpha"], rtol=1e-4, atol=1e-7)
                torch.testing.assert_close(out["mll"][t].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)
                torch.testing.assert_close(out["L"][t].cpu(), ref["L"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("N,D", [(32, 200), (64, None), (24, 300)])
def test_more_dimensions_than_threads(device, N, D):
    """D above the workgroup's thread count (128 threads at N <= 32, 256 at N <= 64; 64 .. 256 in the posterior
    kernels): the 1 / lengthscale table has to be filled with a strided loop (round-1 advisor finding: a one-pass fill
    left invl[blockDim .. D-1] uninitialised and the results silently wrong).  Fit + posterior against the oracle."""
    from scamlgp_amd import _lib
    if D is None:
        D = _lib.lib.scaml_fit_max_d(N)
        assert D > 256
    T, M = 3, 20
    g = torch.Generator().manual_seed(N + D)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.randn(T, N, dtype=torch.float64, generator=g)
    # distinct lengthscales per dimension: a wrong table entry changes the result
    ls = (0.5 + torch.rand(T, D, dtype=torch.float64, generator=g)) * D ** 0.5
    theta = torch.cat([ls, torch.ones(T, 1, dtype=torch.float64), torch.full((T, 1), 1e-3, dtype=torch.float64)], 1)
    xq = torch.rand(M, D, dtype=torch.float64, generator=g)
    for kind in (O.KIND_RBF, O.KIND_MATERN52):
        out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, want_linv=True)
        assert not out["info"].cpu().any()
        post = ops.source_posteriors(xq.to(device), X.to(device), theta.to(device), kind, out["L"], out["Linv_diag"], out["alpha"])
        Linv = ops.linv_batched(out["L"], out["Linv_diag"])
        post2 = ops.source_posteriors(xq.to(device), X.to(device), theta.to(device), kind, None, None, out["alpha"], Linv=Linv)
        for t in range(T):
            ref = O.gp_fit(X[t], y[t], theta[t], kind)
            torch.testing.assert_close(out["mll"][t].cpu(), ref["mll"], rtol=1e-3, atol=0)
            torch.testing.assert_close(out["alpha"][t].cpu(), ref["alpha"], rtol=1e-4, atol=1e-4 * float(ref["alpha"].abs().max()))
            torch.testing.assert_close(out["L"][t].cpu(), ref["L"], rtol=1e-6, atol=1e-9)
            mu, cov = O.source_posterior(xq, X[t], theta[t], kind, ref["L"], ref["alpha"], 0.0, 1.0)
            for p in (post, post2):
                torch.testing.assert_close(p["mean"][t].cpu(), mu, rtol=1e-4, atol=1e-4 * float(mu.abs().max()))
                torch.testing.assert_close(p["var"][t].cpu(), torch.diagonal(cov), rtol=1e-4, atol=1e-4 * float(cov.abs().max()))


def test_posterior_expanded_distance_under_cancellation(device):
    """The L^-1 posterior pass takes the squared distances of a block off the matrix core in the expanded form (round 2): short
    lengthscales and query points (almost) on top of training points are where that form loses digits.  Held to north_star's
    1e-4 on mean / variance against the oracle's difference form, with and without the fused covariance block."""
    T, N, D, M = 3, 96, 8, 40
    g = torch.Generator().manual_seed(8)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.randn(T, N, dtype=torch.float64, generator=g)
    Xq = torch.rand(M, D, dtype=torch.float64, generator=g)
    Xq[:10] = X[0, :10] + 1e-6          # near-duplicates of training points of task 0
    Xq[10:14] = X[1, 5:9]               # exact duplicates (task 1)
    Xq[14] = Xq[15] + 1e-7              # two queries almost on top of each other
    for ls in (0.02, 0.2):
        theta = torch.cat([torch.full((T, D), ls), torch.ones(T, 1), torch.full((T, 1), 1e-3)], 1).double()
        for kind in (O.KIND_MATERN52, O.KIND_RBF):
            fit = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, want_linv=True)
            Linv = ops.linv_batched(fit["L"], fit["Linv_diag"])
            post = ops.source_posteriors(Xq.to(device), X.to(device), theta.to(device), kind, None, None, fit["alpha"], Linv=Linv, cov_first=16)
            for t in range(T):
                ref = O.gp_fit(X[t], y[t], theta[t], kind, dist="direct")
                mu, cov = O.source_posterior(Xq, X[t], theta[t], kind, ref["L"], ref["alpha"], 0.0, 1.0)
                var = torch.diagonal(cov)
                torch.testing.assert_close(post["mean"][t].cpu(), mu, rtol=1e-4, atol=1e-4 * float(mu.abs().max()))
                torch.testing.assert_close(post["var"][t].cpu(), var, rtol=1e-4, atol=1e-4 * float(var.abs().max()))
                torch.testing.assert_close(post["cov"][t].cpu(), cov[:16], rtol=1e-4, atol=1e-4 * float(cov.abs().max()))
