"""The model-level multi-GPU path on real kernels: two ranks (both on cuda:0, gloo -- RCCL refuses two ranks on one
device, and this box has one GPU) each fit and hold a shard of the source tasks (`meta_fit_scamlgp(shard=True)`); the
ScaMLGP built on the shards all-reduces the weighted task sums of scamlgp/model.py:129-134.  Every rank's results must
equal the single-process model's.  Also: the summed MLL / hyper-gradient exchange of BASELINE configs[3]."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T, N, D, NT, MQ = 6, 40, 2, 5, 9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from scamlgp_amd import model as M, synthetic

    d = synthetic.branin_task_stack(T, N, seed=21, noise_std=1.0)
    meta = {f"t{t}": M.SupervisedDataset(torch.from_numpy(d["X"][t]), torch.from_numpy(d["Y"][t]).unsqueeze(-1)) for t in range(T)}
    g = torch.Generator().manual_seed(8)
    Xt = torch.rand(NT, D, dtype=torch.float64, generator=g)
    yt = torch.tensor(synthetic.branin(-5 + 15 * Xt[:, 0].numpy(), 15 * Xt[:, 1].numpy(), a=1.2, r=6.5), dtype=torch.float64).unsqueeze(-1)
    xq = torch.rand(MQ, D, dtype=torch.float64, generator=g)
    w = torch.tensor([0.4, 0.05, 0.3, 1e-9, 0.6, 0.2], dtype=torch.float64)   # one task gets pruned
    return meta, Xt, yt, xq, w


def _run_model(shard: bool):
    from scamlgp_amd import model as M

    meta, Xt, yt, xq, w = _problem()
    gps = M.meta_fit_scamlgp(meta, num_restarts_log_likelihood=0, seed=3, shard=shard)
    stack = list(gps.values())[0]._stack
    model = M.ScaMLGP(Xt, yt, gps)
    model.weights = w
    post = model.eval().posterior(xq)
    fwd = model.eval().forward(xq)
    th = torch.tensor([0.4, 0.6, 1.2, 2e-3], dtype=torch.float64)
    s_mll, s_grad = stack.summed_mll_and_grad(th)
    pg = model.posterior_with_grad(xq)   # (sharded: one more fused all-reduce of the value + gradient columns)
    return dict(n_local=np.int64(stack.T), theta=stack.theta.cpu().numpy(), objective_sum=float(stack.last_fit_info["objective_sum"]),
                m_all=float(model.m_all), s_all=float(model.s_all), source_means=model.source_means.cpu().numpy(),
                source_covs=model.source_covs.cpu().numpy(), mll=float(model.mll()), mean=post.mvn.mean.cpu().numpy(),
                var=post.mvn.variance.cpu().numpy(), cov=post.mvn.covariance_matrix.cpu().numpy(), fwd_mean=fwd.mean.cpu().numpy(),
                fwd_cov=fwd.covariance_matrix.cpu().numpy(), s_mll=float(s_mll), s_grad=s_grad.cpu().numpy(),
                pg_mu=pg[0].cpu().numpy(), pg_var=pg[1].cpu().numpy(), pg_dmu=pg[2].cpu().numpy(), pg_dvar=pg[3].cpu().numpy())


def _worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "scalable-meta-learning-with-gaussian-processes_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _run_model(shard=True)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_model_matches_single_process(device, tmp_path):
    ref = _run_model(shard=False)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [dict(np.load(str(tmp_path / f"rank{r}.npz"))) for r in range(world)]
    assert [int(g["n_local"]) for g in got] == [3, 3] and int(ref["n_local"]) == T
    # the shards' fits are the unsharded fit's rows (independent tasks, no restarts -> no RNG)
    np.testing.assert_allclose(np.concatenate([g["theta"] for g in got]), ref["theta"], rtol=1e-8)
    for g in got:   # every rank ends up with the same, complete answer
        np.testing.assert_allclose(g["objective_sum"], ref["objective_sum"], rtol=1e-9)
        np.testing.assert_allclose(g["m_all"], ref["m_all"], rtol=1e-12)
        np.testing.assert_allclose(g["s_all"], ref["s_all"], rtol=1e-12)
        np.testing.assert_allclose(g["source_means"], ref["source_means"], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(g["source_covs"], ref["source_covs"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(g["mll"], ref["mll"], rtol=1e-8)
        np.testing.assert_allclose(g["mean"], ref["mean"], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(g["var"], ref["var"], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(g["cov"], ref["cov"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(g["fwd_mean"], ref["fwd_mean"], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(g["fwd_cov"], ref["fwd_cov"], rtol=1e-6, atol=1e-9)
        # configs[3]'s exchange step: [sum MLL || sum dMLL/dtheta] over all ranks' tasks
        np.testing.assert_allclose(g["s_mll"], ref["s_mll"], rtol=1e-11)
        np.testing.assert_allclose(g["s_grad"], ref["s_grad"], rtol=1e-9, atol=1e-12)
        # the posterior with its input gradients: the shards' value / derivative columns add up to the unsharded model's
        np.testing.assert_allclose(g["pg_mu"], ref["mean"], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(g["pg_var"], ref["var"], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(g["pg_dmu"], ref["pg_dmu"], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(g["pg_dvar"], ref["pg_dvar"], rtol=1e-6, atol=1e-8)


# ---- the target refit on a sharded model: every rank must end with the SAME weights and hyper-parameters -------------------
T5 = 5   # odd: the shards differ in size (3 + 2), so the ranks' generators have consumed different amounts by the time of the refit


def _run_target_refit(shard: bool):
    from scamlgp_amd import model as M, synthetic, utils

    d = synthetic.branin_task_stack(T5, N, seed=23, noise_std=1.0)
    meta = {f"t{t}": M.SupervisedDataset(torch.from_numpy(d["X"][t]), torch.from_numpy(d["Y"][t]).unsqueeze(-1)) for t in range(T5)}
    g = torch.Generator().manual_seed(18)
    Xt = torch.rand(8, D, dtype=torch.float64, generator=g)
    yt = torch.tensor(synthetic.branin(-5 + 15 * Xt[:, 0].numpy(), 15 * Xt[:, 1].numpy(), a=0.9, r=5.5), dtype=torch.float64).unsqueeze(-1)
    xq = torch.rand(MQ, D, dtype=torch.float64, generator=g)
    gps = M.meta_fit_scamlgp(meta, num_restarts_log_likelihood=1, seed=3, shard=shard)   # (restarts: the fit draws from the RNG)
    model = M.ScaMLGP(Xt, yt, gps)
    utils.optimize_marginal_likelihood(model, num_restarts=2)
    post = model.eval().posterior(xq)
    return dict(weights=model.weights.cpu().numpy(), raw_theta=model.raw_theta.cpu().numpy(), mll=float(model.mll()),
                mean=post.mvn.mean.cpu().numpy(), var=post.mvn.variance.cpu().numpy())


def _refit_worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "scalable-meta-learning-with-gaussian-processes_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _run_target_refit(shard=True)
    np.savez(os.path.join(out_dir, f"refit{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_target_refit_is_identical_on_every_rank(device, tmp_path):
    """optimize_marginal_likelihood on a ScaMLGP over sharded sources: rank 0 fits (restart points from ITS generator), the result
    is broadcast; weights and theta are bit-identical on all ranks, so the weighted sums they all-reduce afterwards are consistent
    and the ranks make the same number of collective calls."""
    world = 2
    mp.spawn(_refit_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (dict(np.load(str(tmp_path / f"refit{r}.npz"))) for r in range(world))
    assert np.array_equal(a["weights"], b["weights"]) and np.array_equal(a["raw_theta"], b["raw_theta"])
    assert a["weights"].shape == (T5,) and (a["weights"] >= 1e-10).all()
    np.testing.assert_allclose(a["mean"], b["mean"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(a["var"], b["var"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(a["mll"], b["mll"], rtol=1e-12)
    assert np.isfinite(a["mll"])
