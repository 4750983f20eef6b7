"""Generate the committed golden fixtures (tests/golden/*.npz).

PARITY UNPINNED: the reference's numeric substrate (gpytorch/botorch/linear_operator) is not
installable here and the reference ships no numeric fixtures for this path.  Every fixture therefore
carries TWO sets of expected outputs for the same inputs:
  * ``L, alpha, quad, logdet, mll, jitter, post_mean, post_cov`` from the CPU oracle (oracle/gp_oracle.py, the
    restatement of the reference's op sequence), and
  * ``sk_*`` / ``sp_*`` columns derived WITHOUT the oracle: scikit-learn 1.7 ``GaussianProcessRegressor``
    (``ConstantKernel * RBF|Matern(2.5) + WhiteKernel``, ``optimizer=None``, ``alpha=0``: ``L_``, ``alpha_``,
    ``log_marginal_likelihood_value_``, ``predict(return_cov=True)``) and scipy ``cho_factor`` / ``cho_solve`` on
    scikit-learn's kernel matrix.  The oracle (CPU test) and the HIP path (GPU test) are both checked against
    THESE columns, so a misreading of the published algorithm in the oracle cannot certify itself.
The jitter fixture has no scikit-learn counterpart (scikit-learn has no jitter ladder): its ``sp_*`` columns factor
scikit-learn's kernel matrix plus the recorded jitter with scipy.  Inputs follow SURVEY.md §8(c)/(d): Branin / Hartmann-6
task families (scamlgp/benchmarking/functions, benchmarks/*.py ranges), unit-cube designs,
per-task standardised targets, hyper-parameters at the reference inits (scamlgp/model.py:31,
55, 67) with per-task ARD perturbations.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd"))

from oracle import gp_oracle as O  # noqa: E402
# the generators are plain numpy; import the module file directly so no HIP library is needed
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "synthetic", os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "scamlgp_amd", "synthetic.py"))
synthetic = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synthetic)


def theta_for(T, D, rng, ls=0.5, os_=1.0, noise=1e-3, spread=0.3):
    ls_t = ls * (1.0 + spread * (rng.uniform(size=(T, D)) - 0.5))
    return np.concatenate([ls_t, np.full((T, 1), os_), np.full((T, 1), noise)], 1)


def independent_columns(Xn, ysn, m, s, th, kind, xq, jitter):
    """Expected outputs for ONE task from scikit-learn / scipy only (no oracle code)."""
    import scipy.linalg as sla
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel

    D = Xn.shape[1]
    ls, os_, noise = th[:D], th[D], th[D + 1]
    base = RBF(length_scale=ls) if kind == O.KIND_RBF else Matern(length_scale=ls, nu=2.5)
    sig = ConstantKernel(os_) * base
    n = Xn.shape[0]
    out = {}
    Kn = sig(Xn) + (noise + jitter) * np.eye(n)
    c, low = sla.cho_factor(Kn, lower=True)
    out["sp_L"] = np.tril(c)
    out["sp_alpha"] = sla.cho_solve((c, low), ysn)
    out["sp_logdet"] = 2.0 * np.log(np.diag(c)).sum()
    if noise > 0 and jitter == 0.0:
        gpr = GaussianProcessRegressor(kernel=sig + WhiteKernel(noise), optimizer=None, alpha=0.0).fit(Xn, ysn)
        out["sk_L"], out["sk_alpha"], out["sk_lml"] = gpr.L_, gpr.alpha_, gpr.log_marginal_likelihood_value_
        pm, pc = gpr.predict(xq, return_cov=True)
        out["sk_post_mean"] = m + s * pm
        out["sk_post_cov"] = s ** 2 * (pc - noise * np.eye(xq.shape[0]))   # WhiteKernel adds the noise to the predictive diagonal
    return out


def run_case(name, X, Y, theta, kind, M=12, seed=0, n_points=None):
    T, N, D = X.shape
    Ys, m, s = synthetic.standardize_rows(Y) if n_points is None else (None, None, None)
    if n_points is not None:
        Ys = np.zeros_like(Y)
        m = np.zeros(T)
        s = np.ones(T)
        for t in range(T):
            n = n_points[t]
            yy, mm, ss = synthetic.standardize_rows(Y[t:t + 1, :n])
            Ys[t, :n], m[t], s[t] = yy[0], mm[0], ss[0]
    rng = np.random.default_rng(seed + 99)
    xq = rng.uniform(size=(M, D))
    Xt, yt, tt, xqt = (torch.from_numpy(a) for a in (X, Ys, theta, xq))
    L = np.zeros((T, N, N))
    alpha = np.zeros((T, N))
    quad, logdet, mll, jit = (np.zeros(T) for _ in range(4))
    mu = np.zeros((T, M))
    cov = np.zeros((T, M, M))
    for t in range(T):
        n = N if n_points is None else int(n_points[t])
        out = O.gp_fit(Xt[t, :n], yt[t, :n], tt[t], kind)
        L[t, :n, :n] = out["L"].numpy()
        alpha[t, :n] = out["alpha"].numpy()
        quad[t], logdet[t], mll[t], jit[t] = (float(out[k]) for k in ("quad", "logdet", "mll", "jitter"))
        mu_t, cov_t = O.source_posterior(xqt, Xt[t, :n], tt[t], kind, out["L"], out["alpha"], float(m[t]), float(s[t]))
        mu[t], cov[t] = mu_t.numpy(), cov_t.numpy()
    # oracle-independent columns (padded like the oracle's)
    ind = dict(sp_L=np.zeros((T, N, N)), sp_alpha=np.zeros((T, N)), sp_logdet=np.zeros(T), sk_L=np.zeros((T, N, N)),
               sk_alpha=np.zeros((T, N)), sk_lml=np.full(T, np.nan), sk_post_mean=np.full((T, M), np.nan), sk_post_cov=np.full((T, M, M), np.nan))
    for t in range(T):
        n = N if n_points is None else int(n_points[t])
        cols = independent_columns(X[t, :n], Ys[t, :n], float(m[t]), float(s[t]), theta[t], kind, xq, float(jit[t]))
        for k, v in cols.items():
            if k in ("sp_L", "sk_L"):
                ind[k][t, :n, :n] = v
            elif k in ("sp_alpha", "sk_alpha"):
                ind[k][t, :n] = v
            else:
                ind[k][t] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(
        path, X=X, y=Ys, y_mean=m, y_std=s, theta=theta, kind=np.int32(kind),
        n_points=np.asarray(n_points if n_points is not None else [N] * T, dtype=np.int32),
        L=L, alpha=alpha, quad=quad, logdet=logdet, mll=mll, jitter=jit, xq=xq, post_mean=mu, post_cov=cov, **ind,
    )
    print(f"{name}: T={T} N={N} D={D} kind={kind} jitter={jit.tolist()} -> {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    rng = np.random.default_rng(2024)
    # (i) BASELINE config 1: Branin, 4 meta-tasks x 32 points, RBF
    d = synthetic.branin_task_stack(4, 32, seed=0, noise_std=1.0)
    run_case("c1_branin_T4_N32_rbf", d["X"], d["Y"], theta_for(4, 2, rng), O.KIND_RBF)
    # (ii) reduced config 2 / 3
    d = synthetic.branin_task_stack(3, 64, seed=1, noise_std=1.0)
    run_case("c2r_branin_T3_N64_rbf", d["X"], d["Y"], theta_for(3, 2, rng), O.KIND_RBF)
    d = synthetic.smooth_field_task_stack(3, 64, 8, seed=1234)
    run_case("c3r_field_T3_N64_D8_matern", d["X"], d["Y"], theta_for(3, 8, rng), O.KIND_MATERN52)
    # (iii) Hartmann-6 family (config 5 shape, reduced)
    d = synthetic.hartmann6_task_stack(2, 64, seed=2, noise_std=0.1)
    run_case("c5r_hartmann6_T2_N64_matern", d["X"], d["Y"], theta_for(2, 6, rng, noise=1e-4), O.KIND_MATERN52)
    # (iv) edge cases -------------------------------------------------------------
    # N = 2 tasks and a constant-Y task (tests/meta_data_examples.py:8-53 uses 2 points, equal losses)
    X = rng.uniform(size=(3, 2, 2))
    Y = np.array([[1.0, 2.0], [1.0, 1.0], [-0.3, 0.7]])
    run_case("edge_N2_constY", X, Y, theta_for(3, 2, rng), O.KIND_RBF)
    # ragged stack (n_data_per_task is a list in the reference: benchmarks/base.py:216)
    d = synthetic.branin_task_stack(4, 48, seed=3, noise_std=1.0)
    run_case("edge_ragged_T4_N48_matern", d["X"], d["Y"], theta_for(4, 2, rng), O.KIND_MATERN52, n_points=[48, 17, 1, 33])
    # near-singular: exact duplicate points (K is singular) with a slightly NEGATIVE diagonal
    # shift, so the first Cholesky attempt fails deterministically and the jitter escalation of
    # psd_safe_cholesky decides the result: task 0 succeeds at 1e-8, task 1 at 1e-7, task 2 at once.
    d = synthetic.branin_task_stack(3, 32, seed=4, noise_std=0.0)
    Xs = d["X"].copy()
    Xs[0, 16:] = Xs[0, :16]
    Xs[1, 16:] = Xs[1, :16]
    Ys = d["Y"].copy()
    Ys[0, 16:] = Ys[0, :16]
    Ys[1, 16:] = Ys[1, :16]
    th = theta_for(3, 2, rng, ls=0.5, noise=1e-3, spread=0.0)
    th[0, -1] = -1e-9
    th[1, -1] = -5e-8
    run_case("edge_duplicates_jitter_T3_N32_rbf", Xs, Ys, th, O.KIND_RBF)

    # (v) inputs the reference's own tests hold (data only) ----------------------------------------
    # META_DATA_1D (scamlgp/testing.py:18-28): 7 evaluations of one task on x0 in [0.5, 3], mapped to the unit interval as
    # blackboxopt's to_numerical does; a second task = the test's deterministic objective (testing.py:31-35, the quartic
    # 0.75 x^4 - 10 x^2) at the same inputs.
    x_raw = np.array([0.8, 1.49, 1.56, 2.5, 3.0, 1.2, 2.7])
    y_meta = np.array([-6.07, -18.6, -19.9, -33.2, -29.2, -31.1, -30.2])
    y_quartic = np.polyval(np.array([0.75, 0.0, -10.0, 0.0, 0.0]), x_raw)
    Xr = np.stack([(x_raw - 0.5) / 2.5] * 2)[:, :, None]
    run_case("ref_meta1d_quartic_T2_N7_rbf", Xr, np.stack([y_meta, y_quartic]), theta_for(2, 1, rng, ls=0.3), O.KIND_RBF)
    # Forrester family (tests/meta_data_examples.py:141-175; descriptor of tests/optimizer_test.py:62 first), 32 points
    # per task as in tests/optimizer_test.py:66, designs from a seeded generator on [0, 1]
    xf = np.random.default_rng(62).uniform(size=(3, 32, 1))
    desc = [(0.95, 0.02, 1.0), (1.1, -0.5, 0.3), (0.8, 1.0, -1.0)]
    yf = np.stack([a * ((6 * xf[i, :, 0] - 2) ** 2 * np.sin(12 * xf[i, :, 0] - 4)) + b * xf[i, :, 0] + c for i, (a, b, c) in enumerate(desc)])
    run_case("ref_forrester_T3_N32_matern", xf, yf, theta_for(3, 1, rng, ls=0.2, noise=1e-4), O.KIND_MATERN52)


if __name__ == "__main__":
    main()
