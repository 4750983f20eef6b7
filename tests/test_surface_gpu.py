"""The reference's Python surface on the GPU path: the call sequence of scamlgp/optimizer.py:128-148,
176-185 replayed against scamlgp_amd names, the attributes botorch callers read (``likelihood``, ``covar_module``,
``train_inputs``, ``train_targets``, ``posterior(X).mvn``), the differentiable fused MLL op, and the invariance the
reference pins at scamlgp/testing.py:99 (shuffled meta-data give the same model)."""
import math

import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import hyper, model as M, ops, synthetic, utils

pytestmark = pytest.mark.gpu


def _forrester(x, a, b, c):
    # tests/meta_data_examples.py:141-144 (reference-held objective; data only)
    return a * ((6 * x - 2) ** 2 * np.sin(12 * x - 4)) + b * x + c


@pytest.fixture(scope="module")
def forrester_gps(device):
    rng = np.random.default_rng(5)
    meta = {}
    for i, (a, b, c) in enumerate([(0.95, 0.02, 1.0), (1.1, -0.5, 0.3), (0.8, 1.0, -1.0)]):
        x = rng.uniform(size=(32, 1))
        meta[i] = M.SupervisedDataset(torch.from_numpy(x), torch.from_numpy(_forrester(x, a, b, c)))
    return meta, M.meta_fit_scamlgp(meta, num_restarts_log_likelihood=1, seed=0)


def test_optimizer_call_sequence_replayed(forrester_gps):
    """scamlgp/optimizer.py:128-148 then :176-185, verbatim but for the package name."""
    meta, source_gps = forrester_gps
    n_features, batch_shape, torch_dtype = 1, torch.Size(), torch.float64
    gp_likelihood, gp_kernel = None, None
    model = M.ScaMLGP(
        train_X=torch.empty((*batch_shape, 0, n_features), dtype=torch_dtype),
        train_Y=torch.empty((*batch_shape, 0, 1), dtype=torch_dtype),
        source_gps=source_gps,
        likelihood=gp_likelihood,
        covar_module=gp_kernel,
    )
    # prior-only model: the posterior is the weighted source prior + k_t
    p0 = model.posterior(torch.rand(4, 1, dtype=torch.float64))
    assert p0.mean.shape == (4, 1) and p0.variance.shape == (4, 1) and bool((p0.variance > 0).all())
    x_filtered = torch.tensor([[0.1], [0.45], [0.8], [0.62]], dtype=torch_dtype)
    y_filtered = torch.from_numpy(_forrester(x_filtered.numpy(), 1.0, 0.0, 0.0))
    for n in (2, 3, 4):
        prev = model
        model = M.ScaMLGP(
            x_filtered[:n],
            y_filtered[:n],
            source_gps,
            likelihood=model.likelihood,
            covar_module=model.covar_module,
        )
        # the modules are handed over, not copied: the new model starts from the previous fit (warm start)
        assert model.likelihood is prev.likelihood and model.covar_module is prev.covar_module
        torch.testing.assert_close(model.raw_theta, prev.raw_theta)
        assert model.weights.tolist() == [1.0 / 3] * 3    # the weights restart (scamlgp/model.py:319-322)
        utils.optimize_marginal_likelihood(model, 1)
        # what tests/optimizer_test.py:100-103 reads
        assert model.train_inputs[0].numel() == n and model.train_targets.numel() == n
    assert isinstance(model.likelihood, hyper.GaussianLikelihood) and isinstance(model.covar_module, hyper.ScaleKernel)
    th = model.theta
    torch.testing.assert_close(model.covar_module.base_kernel.lengthscale.reshape(-1), th[:1])
    torch.testing.assert_close(model.covar_module.outputscale.reshape(()), th[1])
    torch.testing.assert_close(model.likelihood.noise.reshape(()), th[2])
    assert 1e-8 < float(model.likelihood.noise) < 1e-2
    # a user-supplied kernel module selects the family and the priors (scamlgp/optimizer.py:33-35 gp_kernel)
    mk = hyper.get_default_kernel(hyper.MaternKernel, 1)
    m2 = M.ScaMLGP(x_filtered, y_filtered, source_gps, covar_module=mk)
    assert m2.kind == O.KIND_MATERN52 and m2.covar_module is mk
    with pytest.raises(ValueError, match="ard_num_dims"):
        M.ScaMLGP(x_filtered, y_filtered, source_gps, covar_module=hyper.get_default_kernel(hyper.RBFKernel, 3))


def _oracle_joint(stack, w, x):
    mus, covs = [], []
    for t in range(stack.T):
        n = stack.n_list[t]
        X, y, th = stack.X[t, :n].cpu(), stack.y[t, :n].cpu(), stack.theta[t].cpu()
        fit = O.gp_fit(X, y, th, stack.kind)
        mu, cov = O.source_posterior(x, X, th, stack.kind, fit["L"], fit["alpha"], float(stack.y_mean[t]), float(stack.y_std[t]))
        mus.append(mu)
        covs.append(cov)
    return O.target_prior(torch.stack(mus), torch.stack(covs), w)


def test_posterior_object_matches_oracle_joint(forrester_gps):
    meta, gps = forrester_gps
    stack = gps[0]._stack
    Xt = torch.tensor([[0.15], [0.5], [0.9]], dtype=torch.float64)
    yt = torch.from_numpy(_forrester(Xt.numpy(), 1.0, 0.0, 0.0))
    model = M.ScaMLGP(Xt, yt, gps)
    w = torch.tensor([0.5, 0.3, 0.8], dtype=torch.float64)
    model.weights = w
    xq = torch.linspace(0.02, 0.98, 13, dtype=torch.float64).unsqueeze(-1)
    post = model.eval().posterior(xq)
    mu_j, cov_j = _oracle_joint(stack, w, torch.cat([Xt, xq]))
    mu_ref, S_ref = O.target_posterior(xq, Xt, yt.squeeze(-1), mu_j, cov_j, model.theta.cpu(), O.KIND_RBF, float(model.m_all), float(model.s_all))
    assert post.mean.shape == (13, 1) and post.variance.shape == (13, 1)
    scale = float(S_ref.abs().max())
    torch.testing.assert_close(post.mvn.mean.cpu(), mu_ref, rtol=1e-4, atol=1e-4 * float(mu_ref.abs().max()))
    torch.testing.assert_close(post.mvn.variance.cpu(), S_ref.diagonal(), rtol=1e-4, atol=1e-4 * scale)
    cov = post.mvn.covariance_matrix
    assert cov.shape == (13, 13) and post.mvn.lazy_covariance_matrix is cov
    torch.testing.assert_close(cov.cpu(), S_ref, rtol=1e-4, atol=1e-4 * scale)
    torch.testing.assert_close(torch.diagonal(cov), post.mvn.variance, rtol=1e-8, atol=1e-10 * scale)
    # acquisition values on the HIP posterior vs the oracle's formulas on the oracle's posterior (A11)
    ucb = utils.UpperConfidenceBound(model)(xq).cpu()
    ei = utils.ExpectedImprovement(model, float(yt.min()))(xq).cpu()
    ucb_ref = O.ucb_minimize(mu_ref, S_ref.diagonal())
    ei_ref = O.expected_improvement_minimize(mu_ref, S_ref.diagonal(), float(yt.min()))
    torch.testing.assert_close(ucb, ucb_ref, rtol=1e-4, atol=1e-4 * float(ucb_ref.abs().max()))
    torch.testing.assert_close(ei, ei_ref, rtol=1e-4, atol=1e-4 * float(ei_ref.abs().max()))


@pytest.mark.parametrize("kind,T,N,D", [(O.KIND_RBF, 3, 24, 2), (O.KIND_MATERN52, 2, 96, 5), (O.KIND_MATERN52, 2, 300, 3)])
def test_fused_mll_autograd_function(device, kind, T, N, D):
    """FusedMLL.apply is differentiable in theta and chains with torch ops: value and gradient w.r.t. the RAW
    parameters (through the sigmoid-Interval transform and the priors, all torch autograd) equal the oracle's
    autograd through its own op sequence."""
    g = torch.Generator().manual_seed(N)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.randn(T, N, dtype=torch.float64, generator=g)
    y = (y - y.mean(-1, keepdim=True)) / y.std(-1, keepdim=True)
    spec = hyper.source_gp_spec()
    raw = (spec.to_raw(spec.init_theta(D)).repeat(T, 1) + 0.3 * torch.randn(T, D + 2, dtype=torch.float64, generator=g))
    raw_d = raw.to(device).requires_grad_(True)
    theta = spec.to_theta(raw_d)
    mll = ops.FusedMLL.apply(X.to(device), y.to(device), theta, kind)
    assert mll.shape == (T,) and mll.requires_grad
    obj = mll + spec.log_prior(theta) / N
    wts = torch.linspace(0.5, 1.5, T, dtype=torch.float64, device=device)    # a non-trivial upstream gradient
    (grad,) = torch.autograd.grad((wts * obj).sum(), raw_d)
    bounds = [(1e-4, 1e2)] * (D + 1) + [(1e-8, 1e-2)]
    for t in range(T):
        f, gref, _ = O.mll_value_and_grad_raw(X[t], y[t], raw[t], kind, bounds)
        np.testing.assert_allclose(float(obj[t].detach()), float(f), rtol=1e-3)   # north_star: 1e-3 on the marginal likelihood
        np.testing.assert_allclose(grad[t].cpu().numpy() / float(wts[t]), gref.numpy(), rtol=1e-4, atol=1e-7)
    # numerical gradcheck of the op itself on one hyper-parameter (central differences through the fused fit)
    th0 = spec.to_theta(raw.to(device)).detach()
    th_g = th0.clone().requires_grad_(True)
    (g_th,) = torch.autograd.grad(ops.fused_mll(X.to(device), y.to(device), th_g, kind).sum(), th_g)
    h = 1e-6
    for col in (0, D, D + 1):
        e = torch.zeros_like(th0)
        e[:, col] = h * th0[:, col]
        fd = (ops.fused_mll(X.to(device), y.to(device), th0 + e, kind) - ops.fused_mll(X.to(device), y.to(device), th0 - e, kind)) / (2 * e[:, col])
        np.testing.assert_allclose(g_th[:, col].cpu().numpy(), fd.detach().cpu().numpy(), rtol=2e-4, atol=1e-7)


def _meta_1d():
    # scamlgp/testing.py:18-28 META_DATA_1D (reference-held fixture, data only), search space x0 in [0.5, 3]
    x = np.array([0.8, 1.49, 1.56, 2.5, 3.0, 1.2, 2.7])
    y = np.array([-6.07, -18.6, -19.9, -33.2, -29.2, -31.1, -30.2])
    return (x - 0.5) / 2.5, y


def _quartic(x0):
    # scamlgp/testing.py:31-35: polyval([0.75, 0, -10, 0, 0], x0) on the native scale
    return float(np.polyval(np.array([0.75, 0.0, -10.0, 0.0, 0.0]), x0))


def test_shuffled_meta_data_give_the_same_model(device):
    """scamlgp/testing.py:38-99: the same meta evaluations in another order (the reference sorts them,
    scamlgp/utils.py:72-109; here the batched path must not depend on the order at all) -- and, with several tasks,
    the tasks in another order -- give the same posterior; different meta-data give a different one."""
    xs, ys = _meta_1d()
    rng = np.random.default_rng(0)
    tasks = {"task_1": (xs, ys), "task_2": (xs[::-1].copy() * 0.9 + 0.05, ys[::-1].copy() + 3.0)}
    xt = torch.tensor([[0.2], [0.7]], dtype=torch.float64)
    yt = torch.tensor([[_quartic(0.5 + 2.5 * 0.2)], [_quartic(0.5 + 2.5 * 0.7)]], dtype=torch.float64)
    xq = torch.linspace(0.0, 1.0, 17, dtype=torch.float64).unsqueeze(-1)

    def build(task_order, perm_seed):
        meta = {}
        for name in task_order:
            x, y = tasks[name]
            perm = np.random.default_rng(perm_seed).permutation(len(x)) if perm_seed is not None else np.arange(len(x))
            meta[name] = M.SupervisedDataset(torch.from_numpy(x[perm]).unsqueeze(-1), torch.from_numpy(y[perm]).unsqueeze(-1))
        gps = M.meta_fit_scamlgp(meta, num_restarts_log_likelihood=0, seed=1)
        model = M.ScaMLGP(xt, yt, gps)
        model.weights = torch.tensor([0.7, 0.4] if task_order[0] == "task_1" else [0.4, 0.7], dtype=torch.float64)
        p = model.eval().posterior(xq)
        return p.mvn.mean.cpu(), p.mvn.variance.cpu()

    m0, v0 = build(["task_1", "task_2"], None)
    m1, v1 = build(["task_1", "task_2"], 3)          # shuffled evaluations
    m2, v2 = build(["task_2", "task_1"], 4)          # shuffled evaluations and task order
    for m, v in ((m1, v1), (m2, v2)):
        torch.testing.assert_close(m, m0, rtol=1e-6, atol=1e-6 * float(m0.abs().max()))
        torch.testing.assert_close(v, v0, rtol=1e-5, atol=1e-6 * float(v0.abs().max()))
    # totally different meta-data (testing.py:91-97) must change the answer
    other = {"task_1": M.SupervisedDataset(torch.tensor([[0.02]], dtype=torch.float64), torch.tensor([[-4.07]], dtype=torch.float64))}
    model_o = M.ScaMLGP(xt, yt, M.meta_fit_scamlgp(other, num_restarts_log_likelihood=0, seed=1))
    mo = model_o.eval().posterior(xq).mvn.mean.cpu()
    assert float((mo - m0).abs().max()) > 1e-2 * float(m0.abs().max())


def test_gaussian_log_prob_op_matches_torch_and_gradcheck(device):
    """ops.GaussianLogProb (the target GP's MultivariateNormal.log_prob on the library's factorisation): value and both
    gradients against torch.distributions, finite-difference gradcheck, NaN for a matrix no jitter repairs."""
    from scamlgp_amd import ops
    g = torch.Generator().manual_seed(3)
    for n in (1, 7, 33, 80, 200):
        A = torch.randn(n, n, dtype=torch.float64, generator=g)
        K0 = (A @ A.T / n + 0.5 * torch.eye(n, dtype=torch.float64))
        r0 = torch.randn(n, dtype=torch.float64, generator=g)
        K = K0.to(device).requires_grad_(True)
        r = r0.to(device).requires_grad_(True)
        val = ops.gaussian_log_prob(K, r)
        gK, gr = torch.autograd.grad(val, (K, r))
        Kc, rc = K0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        ref = torch.distributions.MultivariateNormal(torch.zeros(n, dtype=torch.float64), covariance_matrix=Kc).log_prob(rc)
        hK, hr = torch.autograd.grad(ref, (Kc, rc))
        torch.testing.assert_close(val.cpu(), ref.detach(), rtol=1e-10, atol=1e-10)
        torch.testing.assert_close(gr.cpu(), hr, rtol=1e-8, atol=1e-10)
        # torch differentiates through the lower triangle only: compare the symmetrised gradients
        torch.testing.assert_close(0.5 * (gK + gK.T).cpu(), 0.5 * (hK + hK.T), rtol=1e-7, atol=1e-9)
    n = 6
    A = torch.randn(n, n, dtype=torch.float64, generator=g)
    K = (A @ A.T + torch.eye(n, dtype=torch.float64)).to(device).requires_grad_(True)
    r = torch.randn(n, dtype=torch.float64, generator=g).to(device).requires_grad_(True)
    sym = lambda K_, r_: ops.gaussian_log_prob(0.5 * (K_ + K_.T), r_)
    assert torch.autograd.gradcheck(sym, (K, r), eps=1e-6, atol=1e-6, rtol=1e-5)
    bad = -torch.eye(4, dtype=torch.float64, device=device).requires_grad_(True)
    v = ops.gaussian_log_prob(bad, torch.ones(4, dtype=torch.float64, device=device))
    (gb,) = torch.autograd.grad(v, bad)
    assert bool(torch.isnan(v)) and bool(torch.isnan(gb).all()) and int(ops.GaussianLogProb.last_info[0]) > 0


def test_target_objective_replayed_from_a_hip_graph_equals_the_eager_one(device):
    """utils._GraphedObjective: forward + autograd backward of -mll(z) captured once; every replay gives the eager numbers bit for
    bit, and the fit through it ends where the eager fit ends."""
    from scamlgp_amd.utils import _GraphedObjective, _fit_target
    d = synthetic.branin_task_stack(4, 24, seed=2, noise_std=1.0)
    meta = {f"t{t}": M.SupervisedDataset(torch.from_numpy(d["X"][t]), torch.from_numpy(d["Y"][t]).unsqueeze(-1)) for t in range(4)}
    gps = M.meta_fit_scamlgp(meta, num_restarts_log_likelihood=0, seed=3)
    g = torch.Generator().manual_seed(11)
    Xt = torch.rand(9, 2, dtype=torch.float64, generator=g)
    Yt = (torch.sin(5.0 * Xt[:, :1]) + Xt[:, 1:]) * 20.0
    model = M.ScaMLGP(Xt, Yt, gps)
    D2 = model.raw_theta.numel()
    gobj = _GraphedObjective(model, D2)
    assert gobj.ok
    z0 = torch.cat([model.raw_theta, model.raw_weights]).cpu().numpy()
    rng = np.random.default_rng(0)
    for _ in range(5):
        z = z0 + 0.2 * rng.standard_normal(z0.shape)
        z[D2:] = np.abs(z[D2:]) + 1e-3
        zt = torch.tensor(z, dtype=torch.float64, device=device, requires_grad=True)
        val = -model.mll(zt[:D2], zt[D2:])
        (gr,) = torch.autograd.grad(val, zt)
        out = gobj(z).copy()
        assert out[0] == float(val.detach()) and np.array_equal(out[1:], gr.cpu().numpy())
    a, b = M.ScaMLGP(Xt, Yt, gps), M.ScaMLGP(Xt, Yt, gps)
    torch.manual_seed(5); _fit_target(a, num_restarts=1, use_graph=True)
    torch.manual_seed(5); _fit_target(b, num_restarts=1, use_graph=False)
    torch.testing.assert_close(a.raw_theta, b.raw_theta, rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(a.raw_weights, b.raw_weights, rtol=1e-9, atol=1e-9)


def test_stack_objective_replayed_from_a_hip_graph_gives_the_eager_fit(device):
    """utils._GraphedBatchObjective: the batched L-BFGS of the source stack sees the same objective values and gradients whether the
    evaluation is launched op by op or replayed from a HIP graph; the fits end at the same optimum."""
    from scamlgp_amd.utils import _GraphedBatchObjective, _fit_stack
    T, N = 6, 40
    d = synthetic.branin_task_stack(T, N, seed=4, noise_std=1.0)
    mk = lambda: M.SourceGPStack(list(range(T)), [torch.from_numpy(d["X"][t]) for t in range(T)],   # noqa: E731
                                 [torch.from_numpy(d["Y"][t]).unsqueeze(-1) for t in range(T)], kind=O.KIND_MATERN52, device=device)
    st = mk()
    fun = lambda r: st.objective(r, 2)   # noqa: E731
    x0 = torch.cat([st.raw, st.raw + 0.3], 0)
    gfun = _GraphedBatchObjective(fun, x0)
    assert gfun.ok
    g = torch.Generator().manual_seed(0)
    for _ in range(4):
        x = x0 + 0.2 * torch.randn(x0.shape, dtype=torch.float64, generator=g).to(device)
        f1, g1 = fun(x)
        f2, g2 = gfun(x)
        torch.testing.assert_close(f2, f1, rtol=1e-12, atol=0)       # (alpha's last bits are free in the fused fit's tail: not bit-exact)
        torch.testing.assert_close(g2, g1, rtol=1e-9, atol=1e-12)
    a, b = mk(), mk()
    torch.manual_seed(3); _fit_stack(a, num_restarts=2, use_graph=True)
    torch.manual_seed(3); _fit_stack(b, num_restarts=2, use_graph=False)
    torch.testing.assert_close(a.last_fit_info["objective"], b.last_fit_info["objective"], rtol=1e-7, atol=1e-9)
    torch.testing.assert_close(a.theta, b.theta, rtol=1e-4, atol=1e-9)
