import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic
dev = torch.device("cuda:0")
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for T, N, D in ((32, 512, 6), (32, 384, 6)):
    d = synthetic.hartmann6_task_stack(T, N, seed=0)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    theta = np.concatenate([np.full((T, D), 0.6), np.ones((T, 1)), np.full((T, 1), 1e-2)], 1)
    X, y, th = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (d["X"], ys, theta))
    out = ops.gp_fit_fused(X, y, th, 1)
    ws = ops.mll_backward_workspace(T, N, D, dev)
    us_g = timeit(lambda: ops.mll_backward(X, th, 1, out["L"], out["Linv_diag"], out["alpha"], workspace=ws))
    us_f = timeit(lambda: ops.gp_fit_fused(X, y, th, 1, zero_upper=False))
    print(f"T={T} N={N}: fit {us_f:.1f} us, MLL gradient {us_g:.1f} us")
