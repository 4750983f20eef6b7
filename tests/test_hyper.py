"""CPU tests of the hyper-parameter plumbing: constraints, priors, prior sampling and the batched
L-BFGS (all plain torch; the oracle is the checker for the restated densities)."""
import math

import numpy as np
import scipy.optimize
import torch

from oracle import gp_oracle as O
from scamlgp_amd import hyper as H


def test_specs_match_reference_inits_and_oracle_priors():
    src, tgt = H.source_gp_spec(), H.target_gp_spec()
    D = 3
    th = src.init_theta(D)
    assert th.tolist() == [0.5, 0.5, 0.5, 1.0, 1e-3]            # scamlgp/model.py:55, 67, 31
    assert tgt.init_theta(D).tolist() == [1.0, 1.0, 1.0, 0.1, 1e-3]  # scamlgp/model.py:94, 102, 31
    raw = src.to_raw(th)
    torch.testing.assert_close(src.to_theta(raw), th, rtol=1e-12, atol=0)
    theta = torch.tensor([[0.3, 0.7, 1.2, 0.9, 2e-3], [0.5, 0.5, 0.5, 1.0, 1e-3]], dtype=torch.float64)
    torch.testing.assert_close(src.log_prior(theta), O.source_gp_log_prior(theta))
    w = torch.tensor([0.2, 0.5], dtype=torch.float64)
    torch.testing.assert_close(tgt.log_prior(theta[0]) + H.GammaPrior(1.0, 1.0).log_prob(w).sum(), O.target_gp_log_prior(theta[0], w))
    # analytic derivatives of transform and priors vs autograd
    r = raw.clone().requires_grad_(True)
    t = src.to_theta(r)
    (gr,) = torch.autograd.grad(src.log_prior(t), r)
    torch.testing.assert_close(gr, src.dlog_prior(t.detach()) * src.dtheta_draw(raw))


def test_prior_sampling_respects_constraints():
    torch.manual_seed(0)
    src = H.source_gp_spec()
    s = src.sample_prior((64,), 4)
    lo, hi = src.bounds(4)
    assert s.shape == (64, 6) and bool(((s > lo) & (s < hi)).all())
    assert bool(torch.isfinite(src.to_raw(s)).all())


def test_batched_lbfgs_matches_scipy_on_rosenbrock_family():
    # B independent 4-d Rosenbrock-like problems with different scales
    B, P = 7, 4
    scale = torch.linspace(1.0, 20.0, B, dtype=torch.float64)

    def fun(x):
        x = x.clone().requires_grad_(True)
        f = (scale[:, None] * (x[:, 1:] - x[:, :-1] ** 2) ** 2 + (1 - x[:, :-1]) ** 2).sum(-1)
        (g,) = torch.autograd.grad(f.sum(), x)
        return f.detach(), g

    x0 = torch.full((B, P), -0.5, dtype=torch.float64)
    res = H.batched_lbfgs(fun, x0, max_iter=500, gtol=1e-8, ftol=0.0)
    assert bool(res.converged.all())
    torch.testing.assert_close(res.x, torch.ones(B, P, dtype=torch.float64), rtol=0, atol=1e-5)
    assert float(res.f.max()) < 1e-10
    # sanity versus scipy on one member
    sp = scipy.optimize.minimize(lambda v: tuple(map(lambda a: a.numpy()[0], fun(torch.tensor(v[None])))), x0[0].numpy(), jac=True, method="L-BFGS-B")
    np.testing.assert_allclose(res.x[0].numpy(), sp.x, atol=1e-4)


def test_batched_lbfgs_handles_nonfinite_regions_and_flags_failures():
    def fun(x):
        f = torch.where(x[:, 0] > 2.0, torch.full_like(x[:, 0], float("nan")), ((x - 1.0) ** 2).sum(-1))
        return f, 2 * (x - 1.0)

    x0 = torch.tensor([[0.0, 0.0], [1.9, 5.0], [3.0, 0.0]], dtype=torch.float64)
    res = H.batched_lbfgs(fun, x0)
    assert res.failed.tolist() == [False, False, True]
    torch.testing.assert_close(res.x[:2], torch.ones(2, 2, dtype=torch.float64), rtol=0, atol=1e-4)
