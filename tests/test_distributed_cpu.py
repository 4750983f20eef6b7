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
