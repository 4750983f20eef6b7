import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic
dev = torch.device("cuda:0")
T, N, D = 32, 512, 6
d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
ys, m, s = synthetic.standardize_rows(d["Y"])
theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
for _ in range(12):
    out = ops.gp_fit_fused(X, y, th, 1)
torch.cuda.synchronize()
print("ok", int(out["info"].sum()))
