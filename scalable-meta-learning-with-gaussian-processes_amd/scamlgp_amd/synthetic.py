"""Synthetic objective families used to generate (T, N, D) task stacks.

Restates, without parameterspace/blackboxopt, the closed-form objectives of
``scamlgp/benchmarking/functions/{branin,hartmann}.py`` and the task-family parameter
ranges of ``scamlgp/benchmarking/benchmarks/{branin,hartmann_3d,hartmann_6d}.py``.  Inputs
are the unit cube (the reference GPs see unit-cube inputs, ``scamlgp/model.py:50-51``);
``X`` is mapped to each function's native search space before evaluation.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

# --- Branin (functions/branin.py:41; search space benchmarks/branin.py:45-47) -------------
BRANIN_BOUNDS = np.array([[-5.0, 10.0], [0.0, 15.0]])
# family ranges for [a, b, c, r, s, t] (benchmarks/branin.py:33-43)
BRANIN_PARAM_RANGES = dict(a=(0.5, 1.5), b=(0.1, 0.15), c=(1.0, 2.0), r=(5.0, 7.0), s=(8.0, 12.0), t=(0.03, 0.05))


def branin(x1, x2, a=1.0, b=5.1 / (4 * math.pi ** 2), c=5 / math.pi, r=6.0, s=10.0, t=1 / (8 * math.pi)):
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    return a * (x2 - b * x1 ** 2 + c * x1 - r) ** 2 + s * (1 - t) * np.cos(x1) + s


# --- Hartmann (functions/hartmann.py:9-36, 96-104, 170-185) ------------------------------
HARTMANN3_A = np.array([[3.0, 10, 30], [0.1, 10, 35], [3.0, 10, 30], [0.1, 10, 35]])
HARTMANN3_P = 1e-4 * np.array([[3689, 1170, 2673], [4699, 4387, 7470], [1091, 8732, 5547], [381, 5743, 8828]])
HARTMANN6_A = np.array(
    [[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14], [3, 3.5, 1.7, 10, 17, 8], [17, 8, 0.05, 10, 0.1, 14]]
)
HARTMANN6_P = 1e-4 * np.array(
    [
        [1312, 1696, 5569, 124, 8283, 5886],
        [2329, 4135, 8307, 3736, 1004, 9991],
        [2348, 1451, 3522, 2883, 3047, 6650],
        [4047, 8828, 8732, 5743, 1091, 381],
    ]
)
# family ranges for alpha1..alpha4 (benchmarks/hartmann_3d.py:31-34, shared by hartmann_6d.py)
HARTMANN_ALPHA_RANGES = ((1.0, 1.02), (1.18, 1.2), (2.8, 3.0), (3.2, 3.4))
HARTMANN_ALPHA_DEFAULT = np.array([1.0, 1.2, 3.0, 3.2])


def hartmann(x: np.ndarray, alpha: np.ndarray, A: np.ndarray, P: np.ndarray) -> np.ndarray:
    """f(x) = -sum_i alpha_i exp(-sum_j A_ij (x_j - P_ij)^2);  x (n, d) -> (n,)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    expo = np.exp(-np.sum(A[None, :, :] * (x[:, None, :] - P[None, :, :]) ** 2, axis=-1))
    return -(expo @ np.asarray(alpha, dtype=np.float64))


def hartmann3(x, alpha=HARTMANN_ALPHA_DEFAULT):
    return hartmann(x, alpha, HARTMANN3_A, HARTMANN3_P)


def hartmann6(x, alpha=HARTMANN_ALPHA_DEFAULT):
    return hartmann(x, alpha, HARTMANN6_A, HARTMANN6_P)


# --- Quadratic (functions/quadratic.py:10-31; family ranges benchmarks/quadratic.py:23-31) ------------------------------
QUADRATIC_PARAM_RANGES = dict(a=(0.5, 1.5), b=(-0.9, 0.9), c=(-1.0, 1.0))
QUADRATIC_BOUNDS = (-1.0, 1.0)


def quadratic(x, a=1.0, b=0.0, c=0.0):
    """f(x) = (a (x + b))^2 + c on x in [-1, 1]."""
    return (np.asarray(a, dtype=np.float64) * (np.asarray(x, dtype=np.float64) + b)) ** 2 + c


# --- designs ------------------------------------------------------------------------------------------------------
def unit_cube_design(T: int, N: int, D: int, rng: np.random.Generator, design: str = "random") -> np.ndarray:
    """(T, N, D) points in the unit cube: i.i.d. uniform ("random", benchmarks/base.py:119-150 samples the search space
    at random) or one scrambled Sobol sequence per task ("sobol": the space-filling initial designs of the BO literature)."""
    if design == "random":
        return rng.uniform(size=(T, N, D))
    if design == "sobol":
        from scipy.stats import qmc

        return np.stack([qmc.Sobol(d=D, scramble=True, seed=int(rng.integers(2 ** 31))).random(N) for _ in range(T)], 0)
    raise ValueError(f"unknown design {design!r}")


# --- task stacks ---------------------------------------------------------------------
def standardize_rows(Y: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Per-task Standardize(m=1) (botorch semantics, scamlgp/model.py:185): unbiased std,
    floor std < 1e-8 -> 1.  Y (T, N) -> (Y_std, mean (T,), std (T,))."""
    m = Y.mean(axis=1)
    s = Y.std(axis=1, ddof=1) if Y.shape[1] > 1 else np.ones_like(m)
    s = np.where(s >= 1e-8, s, 1.0)
    return (Y - m[:, None]) / s[:, None], m, s


def branin_task_stack(T: int, N: int, seed: int = 0, noise_std: float = 1.0) -> Dict[str, np.ndarray]:
    """T Branin-family tasks with N uniform-random unit-cube points each (+ Gaussian noise,
    sigma = 1.0 as in configurations/branin.py:54)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(T, N, 2))
    params = {k: rng.uniform(lo, hi, size=T) for k, (lo, hi) in BRANIN_PARAM_RANGES.items()}
    x1 = BRANIN_BOUNDS[0, 0] + X[..., 0] * (BRANIN_BOUNDS[0, 1] - BRANIN_BOUNDS[0, 0])
    x2 = BRANIN_BOUNDS[1, 0] + X[..., 1] * (BRANIN_BOUNDS[1, 1] - BRANIN_BOUNDS[1, 0])
    Y = branin(x1, x2, *(params[k][:, None] for k in ("a", "b", "c", "r", "s", "t")))
    Y = Y + noise_std * rng.standard_normal(Y.shape)
    return dict(X=X, Y=Y, params=np.stack([params[k] for k in ("a", "b", "c", "r", "s", "t")], 1))


def quadratic_task_stack(T: int, N: int, seed: int = 0, noise_std: float = 0.0, design: str = "random") -> Dict[str, np.ndarray]:
    """T tasks of the one-dimensional Quadratic family (benchmarks/quadratic.py:23-31), x in [-1, 1] mapped to [0, 1]."""
    rng = np.random.default_rng(seed)
    X = unit_cube_design(T, N, 1, rng, design)
    params = {k: rng.uniform(lo, hi, size=T) for k, (lo, hi) in QUADRATIC_PARAM_RANGES.items()}
    x = QUADRATIC_BOUNDS[0] + X[..., 0] * (QUADRATIC_BOUNDS[1] - QUADRATIC_BOUNDS[0])
    Y = quadratic(x, params["a"][:, None], params["b"][:, None], params["c"][:, None])
    Y = Y + noise_std * rng.standard_normal(Y.shape)
    return dict(X=X, Y=Y, params=np.stack([params[k] for k in ("a", "b", "c")], 1))


def hartmann3_task_stack(T: int, N: int, seed: int = 0, noise_std: float = 0.1, design: str = "random") -> Dict[str, np.ndarray]:
    """T Hartmann-3 family tasks (alpha ranges benchmarks/hartmann_3d.py:31-34)."""
    rng = np.random.default_rng(seed)
    X = unit_cube_design(T, N, 3, rng, design)
    alphas = np.stack([rng.uniform(lo, hi, size=T) for lo, hi in HARTMANN_ALPHA_RANGES], 1)
    Y = np.stack([hartmann3(X[t], alphas[t]) for t in range(T)], 0)
    Y = Y + noise_std * rng.standard_normal(Y.shape)
    return dict(X=X, Y=Y, params=alphas)


def hartmann6_task_stack(T: int, N: int, seed: int = 0, noise_std: float = 0.1) -> Dict[str, np.ndarray]:
    """T Hartmann-6 family tasks (alpha ranges hartmann_3d.py:31-34; noise 0.1 as in
    configurations/hartmann6.py:54)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(T, N, 6))
    alphas = np.stack([rng.uniform(lo, hi, size=T) for lo, hi in HARTMANN_ALPHA_RANGES], 1)
    Y = np.stack([hartmann6(X[t], alphas[t]) for t in range(T)], 0)
    Y = Y + noise_std * rng.standard_normal(Y.shape)
    return dict(X=X, Y=Y, params=alphas)


def smooth_field_task_stack(T: int, N: int, D: int, seed: int = 1234, n_features: int = 64,
                            noise_std: float = 0.1) -> Dict[str, np.ndarray]:
    """Seed-fixed smooth random fields (random Fourier features of an RBF prior with
    lengthscales ~ U[0.3, 1]) — the synthetic input for the D=8 throughput configs."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(T, N, D))
    ls = rng.uniform(0.3, 1.0, size=(T, 1, D))
    Wf = rng.standard_normal((T, n_features, D)) / ls
    b = rng.uniform(0, 2 * math.pi, size=(T, n_features))
    amp = rng.standard_normal((T, n_features)) * math.sqrt(2.0 / n_features)
    Y = np.einsum("tnf,tf->tn", np.cos(np.einsum("tnd,tfd->tnf", X, Wf) + b[:, None, :]), amp)
    Y = Y + noise_std * rng.standard_normal(Y.shape)
    return dict(X=X, Y=Y, params=ls[:, 0, :])
