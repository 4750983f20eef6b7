"""GPU tests of the host model layer (scamlgp_amd.model / utils / bo) against the oracle."""
import math

import numpy as np
import pytest
import scipy.optimize
import torch

from oracle import gp_oracle as O
from scamlgp_amd import hyper, model as M, synthetic, utils
from scamlgp_amd.bo import ScaMLGPBOLoop

pytestmark = pytest.mark.gpu


def _meta_data(T, N, seed=0):
    d = synthetic.branin_task_stack(T, N, seed=seed, noise_std=1.0)
    return {f"task{t}": M.SupervisedDataset(torch.from_numpy(d["X"][t]), torch.from_numpy(d["Y"][t]).unsqueeze(-1)) for t in range(T)}, d


@pytest.fixture(scope="module")
def fitted(device):
    meta, d = _meta_data(4, 32, seed=0)
    gps = M.meta_fit_scamlgp(meta, num_restarts_log_likelihood=2, seed=1234)
    return meta, d, gps


def test_meta_fit_improves_objective_and_matches_scipy(fitted):
    meta, d, gps = fitted
    stack = gps["task0"]._stack
    assert list(gps.keys()) == list(meta.keys())
    theta = stack.theta.cpu()
    lo, hi = stack.spec.bounds(2)
    assert bool(((theta > lo) & (theta < hi)).all())
    bounds = [(1e-4, 1e2)] * 3 + [(1e-8, 1e-2)]
    raw0 = stack.spec.to_raw(stack.spec.init_theta(2))
    for t in range(stack.T):
        n = stack.n_list[t]
        X, y = stack.X[t, :n].cpu(), stack.y[t, :n].cpu()
        f_init, _, _ = O.mll_value_and_grad_raw(X, y, raw0, O.KIND_RBF, bounds)
        f_ours, _, _ = O.mll_value_and_grad_raw(X, y, stack.raw[t].cpu(), O.KIND_RBF, bounds)
        assert float(f_ours) >= float(f_init)
        # an independent optimiser (scipy L-BFGS-B on the oracle) from the same init does not beat the batched fit
        res = scipy.optimize.minimize(lambda z: tuple(map(lambda a: -a.numpy(), O.mll_value_and_grad_raw(X, y, torch.tensor(z), O.KIND_RBF, bounds)[:2])),
                                      raw0.numpy(), jac=True, method="L-BFGS-B")
        assert float(f_ours) >= -res.fun - 1e-4
        # the cached objective equals the oracle's at the fitted hyper-parameters
        np.testing.assert_allclose(float(stack.last_fit_info["objective"][t]), float(f_ours), rtol=1e-6)


def test_validate_meta_data_errors():
    with pytest.raises(ValueError, match="Empty meta data"):
        M.validate_meta_data({})
    a = M.SupervisedDataset(torch.rand(4, 2), torch.rand(4, 1))
    b = M.SupervisedDataset(torch.rand(4, 3), torch.rand(4, 1))
    with pytest.raises(ValueError, match="do not match"):
        M.validate_meta_data({0: a, 1: b})
    c = M.SupervisedDataset(torch.rand(4, 2), torch.rand(4, 2))
    with pytest.raises(ValueError, match="must be one"):
        M.validate_meta_data({0: a, 1: c})
    assert a.X().shape == a.X.shape == (4, 2)


def _oracle_prior(stack, idx, w, x):
    mus, covs = [], []
    for t in idx:
        n = stack.n_list[t]
        X, y, th = stack.X[t, :n].cpu(), stack.y[t, :n].cpu(), stack.theta[t].cpu()
        fit = O.gp_fit(X, y, th, stack.kind)
        mu, cov = O.source_posterior(x, X, th, stack.kind, fit["L"], fit["alpha"], float(stack.y_mean[t]), float(stack.y_std[t]))
        mus.append(mu)
        covs.append(cov)
    return O.target_prior(torch.stack(mus), torch.stack(covs), w)


def test_compute_target_prior_and_source_gp_views(fitted):
    meta, d, gps = fitted
    stack = gps["task0"]._stack
    x = torch.rand(9, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    w = torch.tensor([0.3, 0.1, 0.7, 0.2], dtype=torch.float64)
    mean, cov = M._compute_target_prior(x, list(gps.values()), w)
    mu_ref, cov_ref = _oracle_prior(stack, range(4), w, x)
    torch.testing.assert_close(mean.squeeze(-1).cpu(), mu_ref, rtol=1e-4, atol=1e-4 * float(mu_ref.abs().max()))
    torch.testing.assert_close(cov.cpu(), cov_ref, rtol=0, atol=1e-4 * float(cov_ref.abs().max()))
    with pytest.raises(ValueError, match="does not equal the number of weights"):
        M._compute_target_prior(x, list(gps.values()), w[:2])
    # subset of tasks
    sub = [gps["task2"], gps["task0"]]
    mean2, _ = M._compute_target_prior(x, sub, torch.tensor([0.5, 0.25], dtype=torch.float64))
    mu2, _ = _oracle_prior(stack, [2, 0], torch.tensor([0.5, 0.25], dtype=torch.float64), x)
    torch.testing.assert_close(mean2.squeeze(-1).cpu(), mu2, rtol=1e-4, atol=1e-4 * float(mu2.abs().max()))
    # single-task view: posterior + outcome transform as the reference uses them (model.py:266, 281-288)
    g = gps["task1"]
    p = g.posterior(x)
    assert p.mvn.mean.shape == (9,) and p.mvn.covariance_matrix.shape == (9, 9)
    y_raw = g.outcome_transform.untransform(g.train_targets.unsqueeze(-1))[0]
    # (meta_fit_scamlgp keeps a task's observations in canonical -- sorted -- order, as the reference's sort_evaluations does)
    order = np.lexsort((d["Y"][1], d["X"][1][:, 1], d["X"][1][:, 0]))
    np.testing.assert_allclose(y_raw.squeeze(-1).cpu().numpy(), d["Y"][1][order], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(g.train_inputs[0].cpu().numpy(), d["X"][1][order], rtol=0)


def test_scamlgp_train_forward_mll_and_posterior_match_oracle(fitted):
    meta, d, gps = fitted
    stack = gps["task0"]._stack
    g = torch.Generator().manual_seed(5)
    n = 7
    Xt = torch.rand(n, 2, dtype=torch.float64, generator=g)
    yt = torch.tensor(synthetic.branin(-5 + 15 * Xt[:, 0].numpy(), 15 * Xt[:, 1].numpy()), dtype=torch.float64).unsqueeze(-1)
    model = M.ScaMLGP(Xt, yt, gps)
    assert model.source_means.shape == (n, 4) and model.source_covs.shape == (n, n, 4)
    assert model.weights.tolist() == [0.25] * 4
    # Standardize over all meta + target data
    Y_all = np.concatenate([d["Y"].reshape(-1), yt.squeeze(-1).numpy()])
    np.testing.assert_allclose(float(model.m_all), Y_all.mean(), rtol=1e-10)
    np.testing.assert_allclose(float(model.s_all), Y_all.std(ddof=1), rtol=1e-10)
    w = torch.tensor([0.4, 0.05, 0.3, 0.6], dtype=torch.float64)
    model.weights = w
    theta_t = model.theta.cpu()
    ref = O.target_train_mll(Xt, model.train_targets.cpu(), model.source_means.cpu(), model.source_covs.cpu(), w, theta_t, O.KIND_RBF,
                             float(model.m_all), float(model.s_all))
    np.testing.assert_allclose(float(model.mll()), float(ref), rtol=1e-6)
    mvn = model.train().forward(Xt)
    mean_ref = (model.source_means.cpu() @ w - float(model.m_all)) / float(model.s_all)
    torch.testing.assert_close(mvn.mean.cpu(), mean_ref, rtol=1e-9, atol=1e-12)
    # posterior vs the oracle's joint formulation (no task pruned with these weights)
    xq = torch.rand(11, 2, dtype=torch.float64, generator=g)
    post = model.eval().posterior(xq)
    mu_j, cov_j = _oracle_prior(stack, range(4), w, torch.cat([Xt, xq]))
    mu_ref, S_ref = O.target_posterior(xq, Xt, yt.squeeze(-1), mu_j, cov_j, theta_t, O.KIND_RBF, float(model.m_all), float(model.s_all))
    torch.testing.assert_close(post.mean.squeeze(-1).cpu(), mu_ref, rtol=1e-4, atol=1e-4 * float(mu_ref.abs().max()))
    torch.testing.assert_close(post.variance.squeeze(-1).cpu(), S_ref.diagonal(), rtol=1e-4, atol=1e-4 * float(S_ref.abs().max()))
    # eval-mode forward prunes insignificant weights (model.py:364-375)
    model.weights = torch.tensor([0.5, 1e-9, 0.5, 0.5], dtype=torch.float64)
    ev = model.eval().forward(xq)
    mu_p, cov_p = _oracle_prior(stack, [0, 2, 3], torch.tensor([0.5, 0.5, 0.5], dtype=torch.float64), xq)
    torch.testing.assert_close(ev.mean.cpu(), (mu_p - float(model.m_all)) / float(model.s_all), rtol=1e-4, atol=1e-6)


def test_target_fit_and_bo_loop_progress(fitted):
    meta, d, gps = fitted
    # target = a Branin family member on the unit square
    def objective(x):
        x = torch.as_tensor(x).reshape(-1)
        return float(synthetic.branin(-5 + 15 * float(x[0]), 15 * float(x[1]), a=1.1, b=0.12, c=1.5, r=6.2, s=9.0, t=0.04))

    loop = ScaMLGPBOLoop(gps, dim=2, num_restarts_log_likelihood=1, raw_samples=512, num_restarts=8, af_max_iter=40, seed=0)
    x0 = loop.suggest()          # works on an empty target data set (prior only)
    assert x0.shape == (2,) and bool(((x0 >= 0) & (x0 <= 1)).all())
    # the multi-start optimiser's answer is not worse than the best of an independent random batch half the size of its
    # own raw batch (up to 1 % of the acquisition function's range over that batch)
    af0 = loop.acquisition_function()
    pv = af0(torch.rand(256, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(9))).cpu()
    assert float(af0(x0.unsqueeze(0))) >= float(pv.max()) - 0.01 * float(pv.max() - pv.min())
    X, Y = loop.run(objective, 6)
    assert X.shape == (6, 2) and Y.shape == (6, 1)
    before = float(loop.model.mll())
    utils.optimize_marginal_likelihood(loop.model, 1)
    assert float(loop.model.mll()) >= before - 1e-8
    assert bool((loop.model.weights >= 1e-10).all())
    # the loop's evaluations beat the median of random search with the same budget (meta-learned prior + UCB)
    rng = np.random.default_rng(0)
    random_best = np.median([min(objective(torch.from_numpy(rng.uniform(size=2))) for _ in range(6)) for _ in range(21)])
    assert float(Y.min()) <= random_best
    # AF values against the oracle's formulas on this model's own posterior moments (the model-vs-oracle posterior
    # comparison is test_scamlgp_train_forward_mll_and_posterior_match_oracle / test_surface_gpu / test_configs_gpu)
    xc = torch.rand(32, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(4))
    mvn = loop.model.eval().posterior(xc).mvn
    best_f = float(Y.min())
    ei = utils.ExpectedImprovement(loop.model, best_f)(xc).cpu()
    ucb = utils.UpperConfidenceBound(loop.model)(xc).cpu()
    torch.testing.assert_close(ei, O.expected_improvement_minimize(mvn.mean.cpu(), mvn.variance.cpu(), best_f), rtol=1e-9, atol=1e-300)
    torch.testing.assert_close(ucb, O.ucb_minimize(mvn.mean.cpu(), mvn.variance.cpu()), rtol=1e-12, atol=0)


def test_meta_fit_large_source_tasks_hartmann6(device):
    """BASELINE config 5 shape (N = 512 source points, D = 6), reduced T and iterations: the stack
    goes through the two-block fit + the gradient kernels inside the batched L-BFGS."""
    T, N = 3, 512
    d = synthetic.hartmann6_task_stack(T, N, seed=3)
    meta = {t: M.SupervisedDataset(torch.from_numpy(d["X"][t]), torch.from_numpy(d["Y"][t]).unsqueeze(-1)) for t in range(T)}
    stack = M.SourceGPStack(list(meta.keys()), [v.X() for v in meta.values()], [v.Y() for v in meta.values()], kind=O.KIND_MATERN52)
    raw0 = stack.raw.clone()
    f0, g0 = stack.objective(raw0, 1)
    utils._fit_stack(stack, num_restarts=0, max_iter=15)
    f1, _ = stack.objective(stack.raw, 1)
    assert bool((f1 <= f0 + 1e-9).all()) and bool((f1 < f0 - 1e-3).any())
    bounds = [(1e-4, 1e2)] * 7 + [(1e-8, 1e-2)]
    for t in range(T):
        fo, go, _ = O.mll_value_and_grad_raw(stack.X[t].cpu(), stack.y[t].cpu(), raw0[t].cpu(), O.KIND_MATERN52, bounds)
        np.testing.assert_allclose(-float(f0[t]), float(fo), rtol=1e-3)
        np.testing.assert_allclose(-g0[t].cpu().numpy(), go.numpy(), rtol=1e-3, atol=1e-6)
    # posteriors of the refitted stack against the oracle at the fitted hyper-parameters
    xq = torch.rand(9, 6, dtype=torch.float64)
    post = stack.posterior(xq)
    for t in range(T):
        ref = O.gp_fit(stack.X[t].cpu(), stack.y[t].cpu(), stack.theta[t].cpu(), O.KIND_MATERN52)
        mu, cov = O.source_posterior(xq, stack.X[t].cpu(), stack.theta[t].cpu(), O.KIND_MATERN52, ref["L"], ref["alpha"],
                                     float(stack.y_mean[t]), float(stack.y_std[t]))
        torch.testing.assert_close(post["mean"][t].cpu(), mu, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(post["var"][t].cpu(), torch.diagonal(cov), rtol=1e-4, atol=1e-8)


def test_graphed_acquisition_replays_match_eager(fitted, device):
    """The acquisition pass captured into a HIP graph (bo.GraphedAcquisition) gives the eager values for new inputs,
    replay after replay -- the library never synchronises, so the whole evaluation is stream-capturable."""
    from scamlgp_amd.bo import GraphedAcquisition
    meta, d, gps = fitted
    g = torch.Generator().manual_seed(12)
    Xt = torch.rand(6, 2, dtype=torch.float64, generator=g)
    yt = torch.tensor(synthetic.branin(-5 + 15 * Xt[:, 0].numpy(), 15 * Xt[:, 1].numpy()), dtype=torch.float64).unsqueeze(-1)
    model = M.ScaMLGP(Xt, yt, gps).eval()
    model.weights = torch.tensor([0.4, 0.1, 0.3, 0.2], dtype=torch.float64)
    for af in (utils.UpperConfidenceBound(model), utils.ExpectedImprovement(model, float(yt.min()))):
        ga = GraphedAcquisition(af, 50, 2, device)
        for _ in range(3):
            X = torch.rand(50, 2, dtype=torch.float64, generator=g)
            torch.testing.assert_close(ga(X), af(X), rtol=1e-12, atol=1e-14)
        assert ga(X[:7]).shape == (7,)   # another batch size falls back to the eager path


def test_parameter_caches_follow_every_kind_of_parameter_change(fitted, device):
    """`theta` and the pruned weights are cached per parameter state (an acquisition pass asks for them on every call): a new
    tensor, an in-place update and a load_state_dict each have to show in the next posterior."""
    meta, d, gps = fitted
    g = torch.Generator().manual_seed(5)
    Xt = torch.rand(6, 2, dtype=torch.float64, generator=g)
    Yt = (torch.sin(4.0 * Xt[:, :1]) + Xt[:, 1:] ** 2) * 30.0 + torch.rand(6, 1, dtype=torch.float64, generator=g)
    model = M.ScaMLGP(Xt, Yt, gps).eval()
    xq = torch.rand(9, 2, dtype=torch.float64, generator=g)

    def fresh(m):   # the same state in a model that has never cached anything
        twin = M.ScaMLGP(Xt, Yt, gps).eval()
        twin.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
        return twin.posterior(xq)

    def check():
        a, b = model.posterior(xq), fresh(model)
        torch.testing.assert_close(a.mean, b.mean, rtol=1e-11, atol=1e-12)
        torch.testing.assert_close(a.variance, b.variance, rtol=1e-10, atol=1e-13)

    p0 = model.posterior(xq)
    th0 = model.theta.clone()
    assert model.theta is model.theta                       # served from the cache
    check()
    model.raw_theta = model.raw_theta + 0.3                 # new tensors
    assert float((model.theta - th0).abs().max()) > 1e-3
    check()
    model.likelihood.raw_noise.add_(0.5)                    # in place
    model.covar_module.base_kernel.raw_lengthscale.mul_(0.9)
    check()
    model.raw_weights.mul_(0.5)                             # in place: the pruning mask and the scattered weights
    check()
    model.weights = torch.full_like(model.raw_weights, 1e-6)   # everything pruned but for the threshold logic
    check()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.raw_theta = model.raw_theta - 1.0
    model.load_state_dict(sd)
    check()
    assert float((model.posterior(xq).mean - p0.mean).abs().max()) > 0.0


def test_target_posterior_marks_a_failed_factorisation_with_nan_on_the_device(device):
    """The target GP's Cholesky failing even with jitter (psd_safe_cholesky would raise NotPSDError) must not go unnoticed and
    must not cost a host synchronisation: scaml_target_finish_f64 turns the outputs into NaN from the status word."""
    from scamlgp_amd import ops
    g = torch.Generator().manual_seed(0)
    n, Mq, D = 12, 7, 2
    xall = torch.rand(n + Mq, D, dtype=torch.float64, generator=g)
    xall[1] = xall[0]                                           # duplicate training points
    zeros = torch.zeros(n, n + Mq, dtype=torch.float64)
    vec = torch.zeros(n + Mq, dtype=torch.float64)
    y = torch.randn(n, dtype=torch.float64, generator=g)
    for noise, ok in ((1e-3, True), (-1e-3, False)):            # a negative "noise" no jitter of the ladder can repair
        theta = torch.tensor([0.5, 0.5, 1.0, noise], dtype=torch.float64)
        mu, var, info, _ = ops.target_posterior(zeros.to(device), vec.to(device), vec.to(device), xall.to(device), theta.to(device), y.to(device),
                                                0.0, 1.0, O.KIND_RBF)
        assert (int(info[0]) == 0) == ok
        assert bool(torch.isfinite(mu).all()) == ok and bool(torch.isfinite(var).all()) == ok
        if not ok:
            assert bool(torch.isnan(mu).all()) and bool(torch.isnan(var).all())


def test_posterior_keeps_botorch_batch_dimensions(fitted):
    """scamlgp/model.py:359-384 keeps x[..., m, D]'s batch dimensions; botorch's optimize_acqf sends (b, 1, D) to analytic
    acquisition functions and (b, q, D) to joint ones.  (b, q, D) must be b independent q-point joints, NOT one joint over b q
    points; (M, D) stays this package's flat list of M points."""
    meta, d, gps = fitted
    stack = gps["task0"]._stack
    g = torch.Generator().manual_seed(15)
    Xt = torch.rand(5, 2, dtype=torch.float64, generator=g)
    yt = torch.tensor(synthetic.branin(-5 + 15 * Xt[:, 0].numpy(), 15 * Xt[:, 1].numpy()), dtype=torch.float64).unsqueeze(-1)
    model = M.ScaMLGP(Xt, yt, gps).eval()
    w = torch.tensor([0.4, 0.05, 0.3, 0.6], dtype=torch.float64)
    model.weights = w
    b, q = 4, 3
    Xb = torch.rand(b, q, 2, dtype=torch.float64, generator=g)
    post = model.posterior(Xb)
    assert post.mean.shape == (b, q, 1) and post.variance.shape == (b, q, 1)
    assert post.mvn.mean.shape == (b, q) and post.mvn.covariance_matrix.shape == (b, q, q)
    theta_t = model.theta.cpu()
    for i in range(b):   # each batch element against the oracle's joint posterior at ITS q points
        mu_j, cov_j = _oracle_prior(stack, range(4), w, torch.cat([Xt, Xb[i]]))
        mu_ref, S_ref = O.target_posterior(Xb[i], Xt, yt.squeeze(-1), mu_j, cov_j, theta_t, O.KIND_RBF, float(model.m_all), float(model.s_all))
        torch.testing.assert_close(post.mvn.mean[i].cpu(), mu_ref, rtol=1e-4, atol=1e-4 * float(mu_ref.abs().max()))
        torch.testing.assert_close(post.mvn.covariance_matrix[i].cpu(), S_ref, rtol=1e-4, atol=1e-4 * float(S_ref.abs().max()))
        torch.testing.assert_close(post.variance[i, :, 0].cpu(), S_ref.diagonal(), rtol=1e-4, atol=1e-4 * float(S_ref.abs().max()))
    # q = 1: shapes of botorch's analytic-acquisition call, values of the flat call
    X1 = Xb[:, :1]
    p1, pf = model.posterior(X1), model.posterior(X1.reshape(b, 2))
    assert p1.mean.shape == (b, 1, 1) and p1.mvn.covariance_matrix.shape == (b, 1, 1)
    torch.testing.assert_close(p1.mean.reshape(-1), pf.mean.reshape(-1), rtol=0, atol=0)
    ucb = utils.UpperConfidenceBound(model)
    assert ucb(X1).shape == (b,) and torch.equal(ucb(X1), ucb(X1.reshape(b, 2)))
    assert utils.ExpectedImprovement(model, 0.0)(X1).shape == (b,)
    with pytest.raises(ValueError, match="q = 1"):
        ucb(Xb)
    # eval-mode forward: the prior of each batch element's q points
    fw = model.forward(Xb)
    assert fw.mean.shape == (b, q) and fw.covariance_matrix.shape == (b, q, q)
    f0 = model.forward(Xb[2])
    torch.testing.assert_close(fw.mean[2], f0.mean, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(fw.covariance_matrix[2], f0.covariance_matrix, rtol=1e-10, atol=1e-12)
    with pytest.raises(ValueError):
        gps["task0"].posterior(Xb)


def test_model_mll_and_grad_kernel_matches_torch_objective(fitted):
    """ScaMLGP.mll_and_grad (ONE launch of scaml_target_mll_f64) against the torch-autograd objective ScaMLGP.mll it replaces in
    the refit, and against the oracle."""
    meta, d, gps = fitted
    g = torch.Generator().manual_seed(25)
    n = 9
    Xt = torch.rand(n, 2, dtype=torch.float64, generator=g)
    yt = torch.tensor(synthetic.branin(-5 + 15 * Xt[:, 0].numpy(), 15 * Xt[:, 1].numpy()), dtype=torch.float64).unsqueeze(-1)
    model = M.ScaMLGP(Xt, yt, gps)
    model.weights = torch.tensor([0.4, 0.05, 0.3, 0.6], dtype=torch.float64)
    z = torch.cat([model.raw_theta, model.raw_weights]).clone().requires_grad_(True)
    val_t = model.mll(z[:4], z[4:])
    (g_t,) = torch.autograd.grad(val_t, z)
    val_k, g_k = model.mll_and_grad()
    np.testing.assert_allclose(float(val_k[0]), float(val_t), rtol=1e-9)
    torch.testing.assert_close(g_k[0], g_t, rtol=1e-6, atol=1e-9)
    ref = O.target_train_mll(Xt, model.train_targets.cpu(), model.source_means.cpu(), model.source_covs.cpu(), model.weights.cpu(),
                             model.theta.cpu(), O.KIND_RBF, float(model.m_all), float(model.s_all))
    np.testing.assert_allclose(float(val_k[0]), float(ref), rtol=1e-6)
    # the refit through the kernel does not end below the torch / scipy refit it replaces
    state = model.state_dict()
    torch.manual_seed(11)
    utils.optimize_marginal_likelihood(model, 1)
    got = float(model.mll())
    model.load_state_dict(state)
    torch.manual_seed(11)
    utils._fit_target(model, 1, use_kernel=False)
    assert got >= float(model.mll()) - 1e-3 * max(1.0, abs(got))
