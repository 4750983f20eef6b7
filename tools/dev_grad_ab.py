"""Interleaved A/B timing of the MLL gradient (scaml_mll_backward_f64) of several builds of libscaml_hip in ONE process:
python tools/dev_grad_ab.py [--shape T,N,D] libA.so libB.so ...   (variant builds: python __graft_entry__.py --variant NAME -DFOO)."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic
args = sys.argv[1:]
T, N, D = 256, 256, 8
if args and args[0] == "--shape":
    T, N, D = (int(v) for v in args[1].split(",")); args = args[2:]
vp = ctypes.c_void_p
dev = torch.device("cuda:0")
d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
ys, _, _ = synthetic.standardize_rows(d["Y"])
theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
fit = ops.gp_fit_fused(X, y, th, 1, want_linv=True)
ws = ops.mll_backward_workspace(T, N, D, dev)
ref = ops.mll_backward(X, th, 1, fit["L"], fit["Linv_diag"], fit["alpha"], workspace=ws).clone()
libs = {}
for name in args:
    l = ctypes.CDLL(os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "lib", name))
    l.scaml_mll_backward_f64.argtypes = [vp] * 6 + [ctypes.c_int] * 4 + [vp] * 3
    libs[name] = l
def run(l):
    rc = l.scaml_mll_backward_f64(X.data_ptr(), th.data_ptr(), fit["L"].data_ptr(), fit["Linv_diag"].data_ptr(), fit["alpha"].data_ptr(), None,
                                  T, N, D, 1, ws["work"].data_ptr(), ws["partials"].data_ptr(), None)
    assert rc == 0
for n, l in libs.items():
    run(l); torch.cuda.synchronize()
    g = ws["partials"].sum(1) / (2.0 * N)
    print(f"{n:32s} max rel diff to the shipped library {float((g - ref).abs().max() / ref.abs().max()):.1e}")
res = {n: [] for n in libs}
for rnd in range(6):
    for n, l in libs.items():
        for _ in range(3): run(l)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(l)
        e1.record(); torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 20 * 1e3)
for n, v in res.items():
    print(f"{n:32s} T={T} N={N} D={D}: median {statistics.median(v):7.1f} us  min {min(v):7.1f} us")
