import os, sys, subprocess
ROOT = "/root/repo"
code = r'''
import os, sys
sys.path.insert(0, "/root/repo/scalable-meta-learning-with-gaussian-processes_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scamlgp_amd import ops, synthetic
dev = torch.device("cuda:0")
def timeit(fn, reps=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for T, N, D in ((16,128,2),(32,128,2),(64,128,2),(128,128,2),(256,128,2),(64,128,8),(64,96,4),(32,256,8),(64,256,8)):
    d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
    X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
    out = ops.gp_fit_fused(X, y, th, 0, want_linv=True, zero_upper=False)
    ws = ops.mll_backward_workspace(T, N, D, dev)
    us = timeit(lambda: ops.mll_backward(X, th, 0, out["L"], out["Linv_diag"], out["alpha"], workspace=ws))
    print(f"T={T:4d} N={N:4d} D={D}: {us:7.1f} us")
'''
for name, env in (("default", {}), ("no split", {"SCAML_GRAD_NO_SPLIT": "1"}), ("two-launch", {"SCAML_GRAD_LEGACY": "1"})):
    print("==", name, flush=True)
    subprocess.run([sys.executable, "-c", code], env={**os.environ, **env})
