"""Time the MLL gradient paths at the BASELINE shapes (HIP events): single-launch fused kernel vs the two-launch path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic, _lib

dev = torch.device("cuda:0")


def timeit(fn, reps=50, warm=5):
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


for name, T, N, D, kind in [("C3", 256, 256, 8, 1), ("C4 shard", 128, 256, 8, 1), ("C2", 64, 128, 2, 0), ("T=512 N=128", 512, 128, 8, 1), ("T=1024 N=64", 1024, 64, 4, 1), ("C1", 4, 32, 2, 0)]:
    d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
    X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
    fit = ops.gp_fit_fused(X, y, th, kind, want_linv=True)
    ws = ops.mll_backward_workspace(T, N, D, dev)
    res = {}
    for mode in (2, 1):   # 2: single launch forced, 1: two launches forced
        _lib.lib.scaml_debug_force_two_launch_grad(mode)
        g = ops.mll_backward(X, th, kind, fit["L"], fit["Linv_diag"], fit["alpha"], workspace=ws)
        res[mode & 1] = (timeit(lambda: _lib.lib.scaml_mll_backward_f64(X.data_ptr(), th.data_ptr(), fit["L"].data_ptr(), fit["Linv_diag"].data_ptr(),
                                                                     fit["alpha"].data_ptr(), None, T, N, D, kind, ws["work"].data_ptr(),
                                                                     ws["partials"].data_ptr(), torch.cuda.current_stream().cuda_stream)), g)
    _lib.lib.scaml_debug_force_two_launch_grad(0)
    err = float((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max())
    us_fit = timeit(lambda: ops.gp_fit_fused(X, y, th, kind, out=fit, want_linv=True, zero_upper=False))
    print(f"{name:12s} T={T:4d} N={N:3d} D={D}: fit {us_fit:7.1f} us; gradient single-launch {res[0][0]:7.1f} us, two-launch {res[1][0]:7.1f} us "
          f"(max rel diff {err:.1e})", flush=True)
