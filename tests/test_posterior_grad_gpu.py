"""Input gradients of the posterior (scaml_posterior_linv_grad_f64, scaml_target_posterior_grad_f64) on the MI355X, through the C ABI,
against torch autograd through the ORACLE's source / target posterior (scamlgp/model.py:128-134, 359-384 + gpytorch exact prediction;
what botorch's optimize_acqf differentiates through, scamlgp/utils.py:215-224) -- rel. 1e-4 (north_star's bound on mean / variance)."""
import math

import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import model as M, ops, synthetic, utils
from scamlgp_amd.bo import optimize_acqf

pytestmark = pytest.mark.gpu


def _close(got, ref, rtol=1e-4):
    torch.testing.assert_close(got, ref, rtol=rtol, atol=rtol * float(ref.abs().max()) + 1e-300)


@pytest.mark.parametrize("kind", [O.KIND_RBF, O.KIND_MATERN52])
@pytest.mark.parametrize("T,N,D,Ma,Mq,ragged", [(3, 48, 3, 5, 4, False), (2, 100, 6, 21, 3, True), (2, 64, 1, 1, 2, False), (2, 40, 15, 3, 2, False)])
def test_source_pass_values_and_input_gradients(device, kind, T, N, D, Ma, Mq, ragged):
    g = torch.Generator().manual_seed(N + D + kind)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.sin(3.0 * X.sum(-1)) + 0.1 * torch.randn(T, N, dtype=torch.float64, generator=g)
    theta = torch.cat([0.5 + torch.rand(T, D, dtype=torch.float64, generator=g), 0.5 + torch.rand(T, 1, dtype=torch.float64, generator=g),
                       1e-3 + 1e-2 * torch.rand(T, 1, dtype=torch.float64, generator=g)], 1)
    ym, ys = torch.randn(T, dtype=torch.float64, generator=g), 0.5 + torch.rand(T, dtype=torch.float64, generator=g)
    n = torch.tensor([N, N - 13][:T] + [N] * (T - 2), dtype=torch.int32) if ragged else None
    Xa = torch.rand(Ma, D, dtype=torch.float64, generator=g)
    Xq = torch.rand(Mq, D, dtype=torch.float64, generator=g)
    dev = lambda t: None if t is None else t.to(device)   # noqa: E731
    fit = ops.gp_fit_fused(dev(X), dev(y), dev(theta), kind, n_points=dev(n), want_linv=True)
    Linv = ops.linv_batched(fit["L"], fit["Linv_diag"], n_points=dev(n))
    VA = ops.source_posteriors(dev(Xa), dev(X), dev(theta), kind, fit["L"], fit["Linv_diag"], fit["alpha"], dev(ym), dev(ys), n_points=dev(n),
                               want_var=False, keep_V=True, Linv=Linv)["V"]
    out = ops.source_posteriors_grad(dev(Xq), dev(Xa), dev(X), dev(theta), kind, Linv, fit["alpha"], dev(ym), dev(ys), dev(n), VA)
    mu, var, cov = out["mu"].cpu(), out["var"].cpu(), out["cov"].cpu().reshape(T, Ma, Mq, 16)
    for t in range(T):
        k = int(n[t]) if ragged else N
        ref = O.gp_fit(X[t, :k], y[t, :k], theta[t], kind)
        xq = Xq.clone().requires_grad_(True)
        m_ref, S_ref = O.source_posterior(torch.cat([Xa, xq]), X[t, :k], theta[t], kind, ref["L"], ref["alpha"], float(ym[t]), float(ys[t]))
        mq, vq, cq = m_ref[Ma:], S_ref.diagonal()[Ma:], S_ref[:Ma, Ma:]
        _close(mu[t, :, 0], mq.detach())
        _close(var[t, :, 0], vq.detach())
        _close(cov[t, :, :, 0], cq.detach())
        # every query point's outputs depend on that point alone: summing over q and differentiating gives the per-point gradients
        (gm,) = torch.autograd.grad(mq.sum(), xq, retain_graph=True)
        (gv,) = torch.autograd.grad(vq.sum(), xq, retain_graph=True)
        _close(mu[t, :, 1:1 + D], gm)
        _close(var[t, :, 1:1 + D], gv)
        for a in range(min(Ma, 3)):
            (gc,) = torch.autograd.grad(cq[a].sum(), xq, retain_graph=True)
            _close(cov[t, a, :, 1:1 + D], gc)
        assert not bool(mu[t, :, 1 + D:].any()) and not bool(var[t, :, 1 + D:].any())


def _hartmann_model(device, T, N, n, seed=3):
    kind = O.KIND_MATERN52
    d = synthetic.hartmann6_task_stack(T, N, seed=seed, noise_std=0.1)
    stack = M.SourceGPStack([f"h{t}" for t in range(T)], [torch.from_numpy(d["X"][t]) for t in range(T)],
                            [torch.from_numpy(d["Y"][t]).unsqueeze(-1) for t in range(T)], kind=kind, device=device)
    rng = np.random.default_rng(seed)
    stack.set_theta(torch.from_numpy(np.concatenate([0.6 + 0.8 * rng.uniform(size=(T, 6)), 0.5 + rng.uniform(size=(T, 1)),
                                                     1e-3 + 5e-3 * rng.uniform(size=(T, 1))], 1)))
    stack.refresh()
    gps = {tid: M.SourceGP(stack, i) for i, tid in enumerate(stack.task_ids)}
    g = torch.Generator().manual_seed(seed + 1)
    Xt = torch.rand(n, 6, dtype=torch.float64, generator=g)
    yt = torch.from_numpy(synthetic.hartmann6(Xt.numpy(), alpha=np.array([1.01, 1.19, 2.9, 3.3]))).unsqueeze(-1)
    model = M.ScaMLGP(Xt, yt, gps).eval()
    w = torch.from_numpy(0.01 + 0.1 * rng.uniform(size=T))
    w[1] = 1e-9   # pruned
    model.weights = w
    return stack, model, g


def _oracle_posterior(model, fits, xq):
    stack = model._stack
    w = model.weights.cpu()
    mask = O.significant_weights_mask(w, stack.y_std.cpu(), 1e-3)
    xall = torch.cat([model.train_X.cpu(), xq])
    mus, covs = [], []
    for t in range(stack.T):
        if not bool(mask[t]):
            continue
        mu, cov = O.source_posterior(xall, stack.X[t].cpu(), stack.theta[t].cpu(), stack.kind, fits[t]["L"], fits[t]["alpha"],
                                     float(stack.y_mean[t]), float(stack.y_std[t]))
        mus.append(mu)
        covs.append(cov)
    mu_j, cov_j = O.target_prior(torch.stack(mus), torch.stack(covs), w[mask])
    mu, S = O.target_posterior(xq, model.train_X.cpu(), model.train_Y.cpu().squeeze(-1), mu_j, cov_j, model.theta.cpu(), model.kind,
                               float(model.m_all), float(model.s_all))
    return mu, S.diagonal()


@pytest.mark.parametrize("T,N,n,R", [(4, 64, 7, 5), (32, 512, 80, 10)])
def test_target_posterior_gradients_match_oracle_autograd(device, T, N, n, R):
    """(32, 512, 80, 10): BASELINE configs[4] -- 32 sources of 512 points, 80 target points, the 10 starts of one L-BFGS-B evaluation."""
    stack, model, g = _hartmann_model(device, T, N, n)
    assert model.supports_posterior_grad()
    fits = [O.gp_fit(stack.X[t].cpu(), stack.y[t].cpu(), stack.theta[t].cpu(), stack.kind) for t in range(T)]
    Xq = torch.rand(R, 6, dtype=torch.float64, generator=g)
    mu, var, dmu, dvar = (t.cpu() for t in model.posterior_with_grad(Xq))
    xq = Xq.clone().requires_grad_(True)
    mu_ref, var_ref = _oracle_posterior(model, fits, xq)
    _close(mu, mu_ref.detach())
    _close(var, var_ref.detach())
    # (the posterior at one query point does not depend on the others: the joint's diagonal is each point's own marginal)
    (gm,) = torch.autograd.grad(mu_ref.sum(), xq, retain_graph=True)
    (gv,) = torch.autograd.grad(var_ref.sum(), xq, retain_graph=True)
    _close(dmu, gm)
    _close(dvar, gv)
    # the values are those of the plain posterior call
    post = model.posterior(Xq)
    torch.testing.assert_close(post.mvn.mean.cpu(), mu, rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(post.mvn.variance.cpu(), var, rtol=1e-7, atol=1e-12)
    # acquisition functions on top: UCB and EI against autograd through the oracle's formulas on the oracle's posterior
    best_f = float(model.train_Y.min())
    for af, ref_fn in ((utils.UpperConfidenceBound(model), lambda m, v: O.ucb_minimize(m, v)),
                       (utils.ExpectedImprovement(model, best_f), lambda m, v: O.expected_improvement_minimize(m, v, best_f))):
        val, grad = af.value_and_grad(Xq)
        ref = ref_fn(mu_ref, var_ref)
        (gref,) = torch.autograd.grad(ref.sum(), xq, retain_graph=True)
        _close(val.cpu(), ref.detach())
        _close(grad.cpu(), gref)
        torch.testing.assert_close(val, af(Xq), rtol=1e-9, atol=1e-300)


def test_acquisition_optimiser_uses_analytic_gradients(device):
    """optimize_acqf with the analytic path: every L-BFGS-B evaluation scores R points (not (2 D + 1) R), and its end point is at
    least as good as the central-difference path's from the same starts."""
    stack, model, g = _hartmann_model(device, 4, 64, 9, seed=5)
    af = utils.UpperConfidenceBound(model)
    seen = []
    orig = model.posterior_with_grad

    def counting(X):
        seen.append(int(X.shape[0]))
        return orig(X)

    model.posterior_with_grad = counting
    x1, v1 = optimize_acqf(af, 6, raw_samples=128, num_restarts=6, max_iter=30, generator=torch.Generator().manual_seed(0))
    assert seen and set(seen) == {6}
    x2, v2 = optimize_acqf(af, 6, raw_samples=128, num_restarts=6, max_iter=30, generator=torch.Generator().manual_seed(0), analytic_grad=False)
    assert float(v1) >= float(v2) - 1e-6 * max(1.0, abs(float(v2)))
    assert bool(((x1 >= 0) & (x1 <= 1)).all())
    # graph replay of value + gradient equals the eager evaluation
    model.posterior_with_grad = orig
    x3, v3 = optimize_acqf(af, 6, raw_samples=128, num_restarts=6, max_iter=30, generator=torch.Generator().manual_seed(0), graph_device=device)
    torch.testing.assert_close(v3, v1, rtol=1e-9, atol=1e-12)
