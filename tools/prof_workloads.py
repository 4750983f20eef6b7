#!/usr/bin/env python3
"""Workloads for the rocprofv3 passes of tools/profile_round.sh (one per argument), HIP-event timings printed too:
  posterior   C3 stack (T=256, N=256, D=8, Matern): L^-1 once, then posteriors (mean + var) at M = 256 shared query points,
              from L^-1 and by substitution
  c5          BASELINE configs[4] shapes (T=32 sources of N=512, D=6): the two-block source fit, L^-1, and one acquisition
              scoring pass of ScaMLGP (n = 80 target points, M = 1024 candidates): source posteriors at n + M points, the
              covariance block, the weighted task sums, the target GP's algebra, UCB
  grad        C3 stack: fit + MLL gradient (what one L-BFGS evaluation of the meta-fit launches)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic, model as M, utils

dev = torch.device("cuda:0")
which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def timeit(fn, reps=reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def stack(T, N, D, seed=0):
    d = synthetic.smooth_field_task_stack(T, N, D, seed=seed)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
    return (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))


if which in ("posterior", "grad"):
    T, N, D, kind = 256, 256, 8, 1
    X, y, th = stack(T, N, D)
    out = ops.gp_fit_fused(X, y, th, kind, want_linv=True)
    if which == "grad":
        ws = ops.mll_backward_workspace(T, N, D, dev)
        us_f = timeit(lambda: ops.gp_fit_fused(X, y, th, kind, out=out, want_linv=True, zero_upper=False))
        us_g = timeit(lambda: ops.mll_backward(X, th, kind, out["L"], out["Linv_diag"], out["alpha"], workspace=ws))
        print(f"grad: C3 T={T} N={N} D={D}: fit {us_f:.1f} us, MLL gradient {us_g:.1f} us")
    else:
        xq = torch.rand(256, D, dtype=torch.float64, device=dev)
        us_l = timeit(lambda: ops.linv_batched(out["L"], out["Linv_diag"]))
        Linv = ops.linv_batched(out["L"], out["Linv_diag"])
        us_q = timeit(lambda: ops.source_posteriors(xq, X, th, kind, None, None, out["alpha"], Linv=Linv))
        us_s = timeit(lambda: ops.source_posteriors(xq, X, th, kind, out["L"], out["Linv_diag"], out["alpha"]), reps=max(reps // 2, 2))
        print(f"posterior: C3 T={T} N={N} D={D} M=256: L^-1 {us_l:.1f} us once per fit; mean+var from L^-1 {us_q:.1f} us, by substitution {us_s:.1f} us")
elif which == "c5":
    T, N, D, n, Mc = 32, 512, 6, 80, 1024
    d = synthetic.hartmann6_task_stack(T, N, seed=0)
    st = M.SourceGPStack(list(range(T)), [torch.from_numpy(d["X"][t]) for t in range(T)], [torch.from_numpy(d["Y"][t]).unsqueeze(-1) for t in range(T)], kind=1)
    us_fit = timeit(lambda: ops.gp_fit_fused(st.X, st.y, st.theta, 1), reps=5)
    st.refresh()
    gps = {tid: M.SourceGP(st, i) for i, tid in enumerate(st.task_ids)}
    g = torch.Generator().manual_seed(0)
    Xt = torch.rand(n, D, dtype=torch.float64, generator=g)
    Yt = torch.from_numpy(synthetic.hartmann6(Xt.numpy())).unsqueeze(-1)
    mdl = M.ScaMLGP(Xt, Yt, gps).eval()
    cand = torch.rand(Mc, D, dtype=torch.float64, generator=g)
    acq = utils.UpperConfidenceBound(mdl, beta=9.0)
    us_score = timeit(lambda: acq(cand), reps=10)
    print(f"c5: T={T} N={N} D={D}: two-block source fit {us_fit:.1f} us; one UCB scoring pass over M={Mc} candidates with n={n} target points {us_score:.1f} us")
    # round 3: the acquisition optimiser's evaluation with exact input gradients (R = 10 starts) and the on-device target refit
    starts = torch.rand(10, D, dtype=torch.float64, generator=g)
    us_vg = timeit(lambda: acq.value_and_grad(starts), reps=10)
    prob = mdl.target_problem()
    z0 = torch.cat([mdl.raw_theta, mdl.raw_weights]).unsqueeze(0).repeat(3, 1)
    z0[1:, :D + 2] += 0.5
    us_obj = timeit(lambda: ops.target_mll(prob, z0), reps=10)
    import time as _t
    ops.target_fit(prob, z0); torch.cuda.synchronize()
    t0 = _t.perf_counter(); res = ops.target_fit(prob, z0); torch.cuda.synchronize(); ms_fit = (_t.perf_counter() - t0) * 1e3
    print(f"c5: UCB value + exact input gradient at R=10 starts {us_vg:.1f} us; target objective + gradient (B=3, n={n}) {us_obj:.1f} us per launch; "
          f"on-device refit of 3 starts {ms_fit:.2f} ms, stats {res['stats'].cpu().tolist()}")
