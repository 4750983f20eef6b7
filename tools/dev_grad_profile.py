"""rocprofv3 target: the MLL gradient path (L^-1 + gradient kernel) at the headline shape only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic
dev = torch.device("cuda:0")
T, N, D = 256, 256, 8
d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
ys, m, s = synthetic.standardize_rows(d["Y"])
theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
out = ops.gp_fit_fused(X, y, th, 1, want_linv=True)
for _ in range(12):
    g = ops.mll_backward(X, th, 1, out["L"], out["Linv_diag"], out["alpha"])
torch.cuda.synchronize()
print("ok")
