import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
# developer aid: time the MLL gradient path of a VARIANT build (lib/libscaml_hip_NAME.so, __graft_entry__.py --variant)
# through the package's own ops -- the package loads lib/libscaml_hip.so at import, so the load is redirected here
if len(sys.argv) > 1:
    _real_cdll = ctypes.CDLL
    def _redirect(path, *a, **k):
        if isinstance(path, str) and path.endswith("libscaml_hip.so"):
            path = os.path.join(os.path.dirname(path), sys.argv[1])
        return _real_cdll(path, *a, **k)
    ctypes.CDLL = _redirect
from scamlgp_amd import ops, synthetic
dev = torch.device("cuda:0")
T, N, D = 256, 256, 8
d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
ys, m, s = synthetic.standardize_rows(d["Y"])
theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
out = ops.gp_fit_fused(X, y, th, 1, want_linv=True)
f = lambda: ops.mll_backward(X, th, 1, out["L"], out["Linv_diag"], out["alpha"])
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(sys.argv[1:] , f"{e0.elapsed_time(e1)/20*1e3:.1f} us")
