#!/usr/bin/env python3
"""Secondary measurements over the BASELINE.json configurations ("reported also for C2, C4
and the C5 BO-step latency").  Not the headline bench (bench.py): one GPU, HIP-event timing, synthetic data.

  C1  T=4,   N=32,  D=2 RBF        fused fit (launch-latency bound)
  C2  T=64,  N=128, D=2 RBF        fused fit
  C3  T=256, N=256, D=8 Matern     fused fit (= bench.py), + MLL gradient, + posterior at M=256
  C4  T=128, N=256, D=8 Matern     one GPU's shard of the 8-GPU config (128 tasks on 256 CUs)
  C5  T=32,  N=512, D=6 Matern     two-block fit, then the BO step: posteriors of all sources at
                                   n + M candidates (n = 80 target points, M = 1024), weighted prior,
                                   target-GP posterior and UCB -- i.e. one ScaMLGPBOLoop.suggest() scoring pass
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic, model as M, utils

dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def stack(T, N, D, seed=0):
    d = synthetic.smooth_field_task_stack(T, N, D, seed=seed)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
    return (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))


for name, T, N, D, kind in [("C1", 4, 32, 2, 0), ("C2", 64, 128, 2, 0), ("C3", 256, 256, 8, 1), ("C4 shard", 128, 256, 8, 1), ("C5 sources", 32, 512, 6, 1)]:
    X, y, th = stack(T, N, D)
    out = ops.gp_fit_fused(X, y, th, kind, want_linv=True)
    assert not out["info"].any()
    if N <= 256:
        us = timeit(lambda: ops.gp_fit_fused(X, y, th, kind, out=out, zero_upper=False))
    else:
        us = timeit(lambda: ops.gp_fit_fused(X, y, th, kind), reps=5)
    line = f"{name:11s} T={T:4d} N={N:3d} D={D}: fused fit {us:8.1f} us = {T / us * 1e6:10.0f} task-posteriors/s"
    usg = timeit(lambda: ops.mll_backward(X, th, kind, out["L"], out["Linv_diag"], out["alpha"]), reps=10)
    Mq = 256
    xq = torch.rand(Mq, D, dtype=torch.float64, device=dev)
    usp = timeit(lambda: ops.source_posteriors(xq, X, th, kind, out["L"], out["Linv_diag"], out["alpha"]), reps=10)
    usl = timeit(lambda: ops.linv_batched(out["L"], out["Linv_diag"]), reps=10)
    Linv = ops.linv_batched(out["L"], out["Linv_diag"])
    usq = timeit(lambda: ops.source_posteriors(xq, X, th, kind, None, None, out["alpha"], Linv=Linv), reps=10)
    print(line + f"; MLL gradient {usg:8.1f} us; posterior mean+var at M={Mq}: {usp:8.1f} us by substitution, "
          f"{usq:8.1f} us from L^-1 ({usl:6.1f} us once per fit)", flush=True)

# ---- C5: one BO scoring pass on Hartmann-6 with 32 source tasks of 512 points
T, N, D, n, Mc = 32, 512, 6, 80, 1024
d = synthetic.hartmann6_task_stack(T, N, seed=0)
meta = {t: M.SupervisedDataset(torch.from_numpy(d["X"][t]), torch.from_numpy(d["Y"][t]).unsqueeze(-1)) for t in range(T)}
stack_ = M.SourceGPStack(list(meta.keys()), [v.X() for v in meta.values()], [v.Y() for v in meta.values()], kind=1)
t0 = time.perf_counter()
utils._fit_stack(stack_, num_restarts=1, max_iter=30)
torch.cuda.synchronize()
t_fit = time.perf_counter() - t0
gps = {tid: M.SourceGP(stack_, i) for i, tid in enumerate(stack_.task_ids)}
g = torch.Generator().manual_seed(0)
Xt = torch.rand(n, D, dtype=torch.float64, generator=g)
Yt = torch.from_numpy(synthetic.hartmann6(Xt.numpy())).unsqueeze(-1)
t0 = time.perf_counter()
mdl = M.ScaMLGP(Xt, Yt, gps)
torch.cuda.synchronize()
t_model = time.perf_counter() - t0
mdl.eval()
cand = torch.rand(Mc, D, dtype=torch.float64, generator=g)
acq = utils.UpperConfidenceBound(mdl, beta=9.0)
us_score = timeit(lambda: acq(cand), reps=5, warm=2)
print(f"C5 BO      T={T} N={N} D={D}: meta-fit of the source stack (2 starts x 30 L-BFGS iterations, all tasks batched) {t_fit:.2f} s; "
      f"ScaMLGP construction with n={n} target points {t_model * 1e3:.1f} ms; one UCB scoring pass over M={Mc} candidates "
      f"(all source posteriors + weighted prior + target posterior) {us_score / 1e3:.2f} ms", flush=True)
# round 3: the rest of the BO step -- the acquisition optimiser's evaluation with exact gradients and the on-device refit
starts = torch.rand(10, D, dtype=torch.float64, generator=g)
us_vg = timeit(lambda: acq.value_and_grad(starts), reps=10, warm=2)
t0 = time.perf_counter()
utils.optimize_marginal_likelihood(mdl, 2)
torch.cuda.synchronize()
t_refit = time.perf_counter() - t0
print(f"C5 BO      UCB value + exact input gradient at the R=10 starts of an L-BFGS-B evaluation {us_vg / 1e3:.2f} ms (eager; central differences "
      f"score 130 points); refit of the target GP (weights + hyper-parameters, warm start + 2 restarts, one launch on the device) "
      f"{t_refit * 1e3:.1f} ms, stats {mdl.last_fit_info['stats'].cpu().tolist()}", flush=True)
