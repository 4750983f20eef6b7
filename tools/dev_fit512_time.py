"""Timing of the fit beyond the single-launch limit (BASELINE configs[4] sources: T = 32, N = 512, D = 6): the blocked fit
(scaml_gp_fit_blocked_f64) against the composition of library launches + torch ops it replaces, HIP events over 30 calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic

dev = torch.device("cuda:0")
shapes = [(32, 512, 6), (32, 384, 6), (128, 512, 6), (8, 512, 6)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]


def timeit(fn, reps=30, warm=3):
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


for T, N, D in shapes:
    d = synthetic.hartmann6_task_stack(T, N, seed=0) if D == 6 else synthetic.smooth_field_task_stack(T, N, D, seed=0)
    Y = d["Y"] if d["Y"].ndim == 2 else d["Y"][..., 0]
    ys, _, _ = synthetic.standardize_rows(Y)
    theta = np.concatenate([np.full((T, D), 0.6), np.ones((T, 1)), np.full((T, 1), 1e-2)], 1)
    X, y, th = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (d["X"], ys, theta))
    res = {}
    for name, composed in (("blocked", False), ("composed", True)):
        ops._FORCE_COMPOSED_TWO_BLOCK = composed
        out = ops.gp_fit_fused(X, y, th, 1)
        us_r = timeit(lambda: ops.gp_fit_fused(X, y, th, 1))
        us_1 = timeit(lambda: ops.gp_fit_fused(X, y, th, 1, retry=False))
        res[name] = out
        print(f"T={T} N={N} D={D} {name:9s}: {us_r:8.1f} us with the jitter ladder enqueued, {us_1:8.1f} us single-shot; failed {int((out['info'] > 0).sum())}", flush=True)
    a, b = res["blocked"], res["composed"]
    print(f"    max |dL| {float((a['L'] - b['L']).abs().max()):.2e}  max |dalpha| / max|alpha| {float((a['alpha'] - b['alpha']).abs().max() / b['alpha'].abs().max()):.2e}"
          f"  max |dmll| {float((a['mll'] - b['mll']).abs().max()):.2e}")
ops._FORCE_COMPOSED_TWO_BLOCK = False
