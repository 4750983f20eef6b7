import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic
dev = torch.device("cuda:0")
def timeit(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (T, N, D, kind) in [(1024, 32, 2, 0), (2048, 64, 2, 0), (1024, 128, 2, 0), (2048, 128, 4, 1), (1024, 256, 8, 1), (2048, 256, 8, 1)]:
    d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
    ys, m, s = synthetic.standardize_rows(d["Y"])
    theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
    X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
    out = ops.gp_fit_fused(X, y, th, kind)
    us = timeit(lambda: ops.gp_fit_fused(X, y, th, kind, out=out, zero_upper=False))
    fl = N**3/3 + N*(N+1)/2*(4*D+(8 if kind else 2)) + 2*N*N + 3*N
    print(f"T={T} N={N} D={D} kind={kind}: {us:.1f} us -> {T/us*1e6:.3e} task-posteriors/s, {T*fl/us/1e6:.2f} TFLOP/s")
