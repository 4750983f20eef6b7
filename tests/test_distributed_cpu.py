"""World-size-2 gloo test of the multi-GPU host logic (sharding + the summed-MLL / weighted
prior all-reduce).  Runs on CPU; per-task values come from the oracle, so the test checks that
shard -> local sums -> all-reduce reproduces the single-process result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gp_oracle as O
from scamlgp_amd import dist as sdist
from scamlgp_amd import synthetic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    d = synthetic.branin_task_stack(T, 16, seed=7)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    theta = torch.tensor([[0.5, 0.5, 1.0, 1e-3]] * T, dtype=torch.float64)
    w = torch.linspace(0.1, 0.9, T, dtype=torch.float64)
    xq = torch.rand(5, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    lo, hi = sdist.shard_range(T, world, rank)
    buf = torch.zeros(1 + 5 + 25, dtype=torch.float64)  # [sum MLL | mu_s | Sigma_s]
    for t in range(lo, hi):
        fit = O.gp_fit(X[t], y[t], theta[t], O.KIND_RBF)
        mu, cov = O.source_posterior(xq, X[t], theta[t], O.KIND_RBF, fit["L"], fit["alpha"], float(m[t]), float(s[t]))
        buf[0] += fit["mll"]
        buf[1:6] += w[t] * mu
        buf[6:] += (w[t] ** 2 * cov).flatten()
    sdist.allreduce_sum_(buf)
    if rank == 0:
        np.save(out_path, buf.numpy())
    dist.destroy_process_group()


def test_sharded_sums_match_single_process(tmp_path):
    T, world = 5, 2
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(world, _free_port(), T, out), nprocs=world, join=True)
    got = np.load(out)
    # single-process reference
    d = synthetic.branin_task_stack(T, 16, seed=7)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    theta = torch.tensor([[0.5, 0.5, 1.0, 1e-3]] * T, dtype=torch.float64)
    w = torch.linspace(0.1, 0.9, T, dtype=torch.float64)
    xq = torch.rand(5, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    mus, covs, mll = [], [], 0.0
    for t in range(T):
        fit = O.gp_fit(X[t], y[t], theta[t], O.KIND_RBF)
        mu, cov = O.source_posterior(xq, X[t], theta[t], O.KIND_RBF, fit["L"], fit["alpha"], float(m[t]), float(s[t]))
        mus.append(mu)
        covs.append(cov)
        mll += float(fit["mll"])
    mu_s, cov_s = O.target_prior(torch.stack(mus), torch.stack(covs), w)
    np.testing.assert_allclose(got[0], mll, rtol=1e-12)
    np.testing.assert_allclose(got[1:6], mu_s.numpy(), rtol=1e-12)
    np.testing.assert_allclose(got[6:].reshape(5, 5), cov_s.numpy(), rtol=1e-12, atol=1e-15)


def test_shard_ranges_cover_everything():
    for T in (1, 5, 256, 1024, 1023):
        for world in (1, 2, 3, 8):
            r = [sdist.shard_range(T, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == T
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    b = sdist.shard_ranges_balanced([256] * 4 + [32] * 60, 4)
    assert b[0][0] == 0 and b[-1][1] == 64 and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert all(hi > lo for lo, hi in b)


# ---- the model-level shard helpers (scamlgp_amd.dist) under gloo, world size 2 -------------------------------------
def _helpers_worker(rank, world, port, T, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    d = synthetic.branin_task_stack(T, 12, seed=11)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    raw = torch.zeros(4, dtype=torch.float64)
    bounds = [(1e-4, 1e2)] * 3 + [(1e-8, 1e-2)]
    shard = sdist.TaskShard(T)
    assert (shard.lo, shard.hi) == sdist.shard_range(T, world, rank)
    mll, grad = [], []
    for t in range(shard.lo, shard.hi):
        f, g, _ = O.mll_value_and_grad_raw(X[t], y[t], raw, O.KIND_RBF, bounds, with_priors=False)
        mll.append(f)
        grad.append(g)
    s_mll, s_grad = sdist.reduce_mll_and_grad(torch.stack(mll), torch.stack(grad), shard)
    stds = sdist.gather_task_axis(torch.from_numpy(s[shard.lo:shard.hi].copy()), shard)
    cols = sdist.gather_task_axis(torch.from_numpy(d["Y"][shard.lo:shard.hi, :3].T.copy()), shard)   # (3, T_local) -> (3, T)
    extra = torch.tensor([[0.5], [1.5], [-2.0]], dtype=torch.float64)
    y_local = torch.from_numpy(d["Y"][shard.lo:shard.hi].reshape(-1, 1).copy())
    mean, std = sdist.standardize_fit_sharded(y_local, extra, shard)
    a, none, b = sdist.fused_allreduce([torch.full((2, 2), float(rank + 1), dtype=torch.float64), None,
                                        torch.arange(3, dtype=torch.float64)], shard)
    assert none is None
    if rank == 0:
        np.savez(out_path, s_mll=s_mll.numpy(), s_grad=s_grad.numpy(), stds=stds.numpy(), cols=cols.numpy(), mean=mean.numpy(),
                 std=std.numpy(), a=a.numpy(), b=b.numpy())
    dist.destroy_process_group()


def test_shard_helpers_match_single_process(tmp_path):
    T, world = 5, 2
    out = str(tmp_path / "helpers.npz")
    mp.spawn(_helpers_worker, args=(world, _free_port(), T, out), nprocs=world, join=True)
    got = np.load(out)
    d = synthetic.branin_task_stack(T, 12, seed=11)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    X, y = torch.from_numpy(d["X"]), torch.from_numpy(ys)
    raw = torch.zeros(4, dtype=torch.float64)
    bounds = [(1e-4, 1e2)] * 3 + [(1e-8, 1e-2)]
    fs, gs = zip(*[O.mll_value_and_grad_raw(X[t], y[t], raw, O.KIND_RBF, bounds, with_priors=False)[:2] for t in range(T)])
    np.testing.assert_allclose(got["s_mll"], float(torch.stack(fs).sum()), rtol=1e-12)
    np.testing.assert_allclose(got["s_grad"], torch.stack(gs).sum(0).numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(got["stds"], s, rtol=0)
    np.testing.assert_allclose(got["cols"], d["Y"][:, :3].T, rtol=0)
    y_all = np.concatenate([d["Y"].reshape(-1), [0.5, 1.5, -2.0]])
    np.testing.assert_allclose(got["mean"], y_all.mean(), rtol=1e-13)
    np.testing.assert_allclose(got["std"], y_all.std(ddof=1), rtol=1e-13)
    np.testing.assert_allclose(got["a"], np.full((2, 2), 3.0))
    np.testing.assert_allclose(got["b"], 2 * np.arange(3.0))


def test_bench_parent_refuses_missing_gpus(monkeypatch, capsys):
    """`python bench.py --gpus N` outside torchrun starts the ranks itself; without N GPUs (and without the
    rehearsal switch) it must say so and exit non-zero instead of printing a number."""
    import argparse
    import bench

    monkeypatch.delenv("SCAML_BENCH_REHEARSAL", raising=False)
    args = argparse.Namespace(gpus=64, steps=1, warmup=0, step="fit", no_cpu_baseline=True, prewarm=0, total_tasks=0)
    assert bench.launch_ranks(args) == 2
    assert "needs 64 GPUs" in capsys.readouterr().err
