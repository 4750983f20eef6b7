import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic
dev = torch.device("cuda:0")
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for T in (32, 64, 128, 129, 192, 256, 512):
    N, D = 256, 8
    d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
    X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
    out = ops.gp_fit_fused(X, y, th, 1, want_linv=True)
    us = timeit(lambda: ops.mll_backward(X, th, 1, out["L"], out["Linv_diag"], out["alpha"]))
    print(T, f"{us:.1f} us")
