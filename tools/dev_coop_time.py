"""The fit beyond 256 points per task: several CUs per task in one launch (csrc/gp_fit_coop.hip) against the 2 x 2 sequence of launches
(csrc/gp_fit_blocked.hip), interleaved in one process.  scaml_debug_blocked_fit_path: 1 = sequence of launches, 2 = one launch."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import torch
from scamlgp_amd import _lib, ops, synthetic
dev = torch.device("cuda:0")
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = [(32, 512, 6), (16, 512, 6), (64, 512, 6), (128, 512, 6), (32, 384, 6), (8, 512, 6)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for T, N, D in shapes:
    d = synthetic.smooth_field_task_stack(T, N, D, seed=1)
    import numpy as np
    ys, m, s = synthetic.standardize_rows(d["Y"])
    th = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
    X, y, theta = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, th))
    res = {1: [], 2: []}
    outs = {}
    for rnd in range(3):
        for path in (1, 2):
            _lib.lib.scaml_debug_blocked_fit_path(path)
            out = ops.gp_fit_fused(X, y, theta, 1)
            outs[path] = out
            took = _lib.lib.scaml_debug_blocked_fit_path(-1)
            res[path].append(timeit(lambda: ops.gp_fit_fused(X, y, theta, 1)))
    _lib.lib.scaml_debug_blocked_fit_path(0)
    dl = float((outs[1]["L"] - outs[2]["L"]).abs().max()); da = float((outs[1]["alpha"] - outs[2]["alpha"]).abs().max() / outs[1]["alpha"].abs().max())
    print(f"T={T:4d} N={N} D={D}: sequence of launches {statistics.median(res[1]):7.1f} us, one launch ({took}) {statistics.median(res[2]):7.1f} us; "
          f"max |dL| {dl:.1e}, rel |dalpha| {da:.1e}, info {int(outs[2]['info'].abs().max())}", flush=True)
