"""Shared by the target-fit tests (CPU host-emulation and GPU): a target-GP training problem built with the ORACLE --
T source GPs on a smooth synthetic family, their posteriors at n target inputs (source_means (n, T), source_covs (n, n, T) as
scamlgp/model.py:279-289 caches them), target observations, the joint standardiser -- and the oracle's objective
(oracle.target_train_mll on raw parameters, differentiated by torch autograd)."""
import math

import numpy as np
import torch

from oracle import gp_oracle as O

LS_LO, LS_HI, OS_LO, OS_HI, NZ_LO, NZ_HI = 1e-4, 1e2, 1e-4, 1e2, 1e-8, 1e-2
# spec vector of the C ABI: bounds, then (kind, p1, p2) for lengthscale / outputscale / noise / weights priors, then w_lower
TARGET_SPEC = [LS_LO, LS_HI, OS_LO, OS_HI, NZ_LO, NZ_HI, 2, 0.5, 1.5, 2, -2.0, 3.0, 2, -8.0, 2.0, 1, 1.0, 1.0, 1e-10]


def make_target_problem(n: int, T: int, D: int, kind: int, seed: int = 0, n_src: int = 24):
    g = torch.Generator().manual_seed(seed)
    Xt = torch.rand(n, D, dtype=torch.float64, generator=g)
    freq = torch.randn(T + 1, D, dtype=torch.float64, generator=g) * 2.0
    phase = torch.rand(T + 1, dtype=torch.float64, generator=g) * 6.28

    def f(t, X):
        return torch.sin(X @ freq[t] + phase[t]) + 0.3 * torch.cos(2.0 * X @ freq[0] + phase[0]) + 0.1 * t

    means, covs, raw_y = [], [], []
    for t in range(T):
        Xs = torch.rand(n_src, D, dtype=torch.float64, generator=g)
        ys = f(t + 1, Xs) + 0.05 * torch.randn(n_src, dtype=torch.float64, generator=g)
        m, s = ys.mean(), ys.std()
        theta = torch.tensor([0.4 + 0.05 * (t % 3)] * D + [1.0, 1e-3], dtype=torch.float64)
        fit = O.gp_fit(Xs, (ys - m) / s, theta, kind)
        mu, cov = O.source_posterior(Xt, Xs, theta, kind, fit["L"], fit["alpha"], float(m), float(s))
        means.append(mu)
        covs.append(0.5 * (cov + cov.T))
        raw_y.append(ys)
    yt = f(0, Xt) + 0.4 * f(1, Xt) + 0.05 * torch.randn(n, dtype=torch.float64, generator=g)
    y_all = torch.cat(raw_y + [yt])
    m_all, s_all = float(y_all.mean()), float(y_all.std())
    source_means = torch.stack(means, 1)      # (n, T)
    source_covs = torch.stack(covs, 2)        # (n, n, T)
    return dict(X=Xt, y=(yt - m_all) / s_all, source_means=source_means, source_covs=source_covs, m_all=m_all, s_all=s_all,
                n=n, T=T, D=D, kind=kind)


def pack_lower(source_covs: torch.Tensor) -> torch.Tensor:
    """(n, n, T) -> (T, n (n + 1) / 2): lower triangle, element (a, b) at a (a + 1) / 2 + b."""
    n = source_covs.shape[0]
    ia, ib = torch.tril_indices(n, n)
    return source_covs[ia, ib, :].transpose(0, 1).contiguous()


def raw_start(D: int, T: int, seed: int = 0, B: int = 1) -> torch.Tensor:
    g = torch.Generator().manual_seed(100 + seed)
    th = torch.cat([0.3 + torch.rand(B, D, dtype=torch.float64, generator=g), 0.05 + 0.3 * torch.rand(B, 1, dtype=torch.float64, generator=g),
                    1e-4 + 2e-3 * torch.rand(B, 1, dtype=torch.float64, generator=g)], 1)
    lo = torch.tensor([LS_LO] * D + [OS_LO, NZ_LO], dtype=torch.float64)
    hi = torch.tensor([LS_HI] * D + [OS_HI, NZ_HI], dtype=torch.float64)
    raw = O.interval_inverse_transform(th, lo, hi)
    w = 0.02 + torch.rand(B, T, dtype=torch.float64, generator=g) / T * 2.0
    return torch.cat([raw, w], 1)


def oracle_mll_and_grad(prob: dict, z: torch.Tensor):
    """mll (scalar) and d mll / d z at z = [raw theta (D + 2) || weights (T)] by autograd through the oracle."""
    D = prob["D"]
    z = z.clone().requires_grad_(True)
    lo = torch.tensor([LS_LO] * D + [OS_LO, NZ_LO], dtype=torch.float64)
    hi = torch.tensor([LS_HI] * D + [OS_HI, NZ_HI], dtype=torch.float64)
    theta = O.interval_transform(z[:D + 2], lo, hi)
    val = O.target_train_mll(prob["X"], prob["y"], prob["source_means"], prob["source_covs"], z[D + 2:], theta, prob["kind"],
                             prob["m_all"], prob["s_all"])
    (g,) = torch.autograd.grad(val, z)
    return val.detach(), g
