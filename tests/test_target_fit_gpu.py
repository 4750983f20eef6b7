"""The target GP's fused objective / gradient and the on-device refit (scaml_target_mll_f64, scaml_target_fit_f64) on the MI355X,
through the C ABI, against torch autograd through the oracle's target_train_mll (scamlgp/model.py:360-363, 376-383 +
scamlgp/utils.py:171-177) and against scipy L-BFGS-B on the oracle objective (the optimiser the reference drives,
scamlgp/utils.py:175).  Tolerances: 1e-3 relative on the objective (north_star's MLL bound; measured ~1e-12), 1e-4 on the gradient."""
import numpy as np
import pytest
import scipy.optimize
import torch

from scamlgp_amd import _lib, hyper, ops
from tests._target_problem import TARGET_SPEC, make_target_problem, oracle_mll_and_grad, raw_start

pytestmark = pytest.mark.gpu


def _problem_on_device(prob, device):
    spec = hyper.target_gp_spec()
    return ops.TargetFitProblem(prob["source_means"].to(device), prob["source_covs"].to(device), prob["X"].to(device), prob["y"].to(device),
                                prob["m_all"], prob["s_all"], spec, hyper.GammaPrior(1.0, 1.0), 1e-10, prob["kind"])


def test_spec_block_matches_reference_defaults(device):
    prob = make_target_problem(3, 2, 2, 0)
    tp = _problem_on_device(prob, device)
    np.testing.assert_allclose(list(tp.spec_host), TARGET_SPEC)


@pytest.fixture(params=["matrix-core", "column-by-column"])
def fit_path(request):
    """Both factorisations of the kernel: blocked Cholesky on the matrix cores (n <= 112, the default there) and the column-by-column
    elimination (n up to 128; forced through the developer switch for the shapes the matrix-core path would take)."""
    was = _lib.lib.scaml_debug_target_fit_path(1 if request.param == "column-by-column" else 0)
    yield request.param
    _lib.lib.scaml_debug_target_fit_path(was)


@pytest.mark.parametrize("n,T,D,kind", [(1, 3, 2, 0), (7, 3, 2, 1), (7, 32, 6, 1), (17, 4, 16, 0), (80, 3, 6, 0), (80, 32, 6, 1), (112, 6, 3, 1), (128, 5, 3, 1)])
def test_objective_and_gradient_match_oracle_autograd(device, fit_path, n, T, D, kind):
    prob = make_target_problem(n, T, D, kind, seed=n + T, n_src=16)
    B = 3
    z = raw_start(D, T, seed=n, B=B)
    out = ops.target_mll(_problem_on_device(prob, device), z.to(device))
    assert not bool(out["info"].any())
    for b in range(B):
        val, g = oracle_mll_and_grad(prob, z[b])
        np.testing.assert_allclose(out["value"][b].item(), float(val), rtol=1e-3)
        assert abs(out["value"][b].item() - float(val)) <= 1e-8 * max(1.0, abs(float(val)))   # (measured; the bound above is north_star's)
        np.testing.assert_allclose(out["grad"][b].cpu().numpy(), g.numpy(), rtol=1e-4, atol=1e-7)


def test_repeated_launches_are_bit_identical(device, fit_path):
    """Fixed-order reductions, no atomics: the optimiser's trajectory is reproducible."""
    prob = make_target_problem(40, 8, 4, 1, seed=2)
    tp = _problem_on_device(prob, device)
    z = raw_start(4, 8, seed=1, B=4).to(device)
    a = ops.target_mll(tp, z)
    for _ in range(3):
        b = ops.target_mll(tp, z)
        assert torch.equal(a["value"], b["value"]) and torch.equal(a["grad"], b["grad"])


def test_jitter_ladder_and_hopeless_matrix(device, fit_path):
    """psd_safe_cholesky's ladder in-kernel: duplicated target points (singular kernel part) under a diagonal of 1e-8 noise minus
    6e-8 from the source term fail without jitter and with 1e-8 and pass with 1e-7; a NaN weight cannot be saved and comes back as
    NaN / info > 0 / zero gradient."""
    import warnings

    prob = make_target_problem(12, 3, 2, 0, seed=4)
    prob["X"][5] = prob["X"][4]
    prob["X"][7] = prob["X"][4]
    w = 0.1
    c = 6e-8 * prob["s_all"] ** 2 / (3 * w * w)
    prob["source_covs"] = -c * torch.eye(12, dtype=torch.float64).unsqueeze(-1).repeat(1, 1, 3)
    tp = _problem_on_device(prob, device)
    z = raw_start(2, 3, seed=0, B=2)
    z[:, 3] = -40.0    # raw noise -> 1e-8
    z[:, 2] = 5.0      # outputscale ~ 99
    z[:, 4:] = w
    z[1, 4] = float("nan")
    out = ops.target_mll(tp, z.to(device))
    assert out["info"][0].item() == 0 and out["jitter"][0].item() == 1e-7
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        val, g = oracle_mll_and_grad(prob, z[0])
    np.testing.assert_allclose(out["value"][0].item(), float(val), rtol=1e-6)
    assert out["info"][1].item() > 0 and np.isnan(out["value"][1].item()) and not bool(out["grad"][1].any())


@pytest.mark.parametrize("n,T,D,kind", [(12, 4, 2, 1), (80, 32, 6, 1), (16, 70, 2, 0)])   # (P = 74 > 64: the bookkeeping by all waves)
def test_device_refit_reaches_scipy_optimum(device, fit_path, n, T, D, kind):
    prob = make_target_problem(n, T, D, kind, seed=5, n_src=16)
    tp = _problem_on_device(prob, device)
    B = 3
    z0 = raw_start(D, T, seed=5, B=B)
    res = ops.target_fit(tp, z0.to(device))
    stats = res["stats"].cpu().numpy()
    assert (stats[:, 2] != 4).all() and (stats[:, 0] >= 1).all()

    def fun(zv):
        val, g = oracle_mll_and_grad(prob, torch.from_numpy(zv))
        return -float(val), -g.numpy()

    bounds = [(None, None)] * (D + 2) + [(1e-10, None)] * T
    zs = res["z"].cpu()
    for b in range(B if n < 15 else 1):
        ref = scipy.optimize.minimize(fun, z0[b].numpy(), jac=True, method="L-BFGS-B", bounds=bounds, options=dict(maxiter=200))
        got = -res["value"][b].item()
        assert got <= ref.fun + 1e-3 * max(1.0, abs(ref.fun)), (got, ref.fun, stats[b])
    for b in range(B):   # the reported value is the oracle's objective at the returned point, inside the box
        val, _ = oracle_mll_and_grad(prob, zs[b])
        np.testing.assert_allclose(res["value"][b].item(), float(val), rtol=1e-8)
        assert bool((zs[b][D + 2:] >= 1e-10).all())


def test_limits_and_argument_contract(device):
    lib = _lib.lib
    assert lib.scaml_target_fit_max_n(32, 6) >= 128 and lib.scaml_target_fit_max_d() >= 8
    assert lib.scaml_target_fit_max_n(32, 99) == 0
    prob = make_target_problem(5, 2, 2, 0)
    tp = _problem_on_device(prob, device)
    with pytest.raises(ValueError):
        ops.target_mll(tp, torch.zeros(1, tp.P + 1, dtype=torch.float64, device=device))
    assert not ops.TargetFitProblem.supported(4000, 3, 2)
