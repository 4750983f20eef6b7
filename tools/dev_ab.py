"""Interleaved A/B timing of several builds of libscaml_hip in ONE process on ONE device
(python tools/dev_ab.py libA.so libB.so ...): headline shape, first-attempt path only."""
import ctypes, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import synthetic
vp = ctypes.c_void_p
libs = {}
for name in sys.argv[1:]:
    l = ctypes.CDLL(os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "lib", name))
    l.scaml_gp_fit_fused_f64.argtypes = [vp]*5 + [ctypes.c_int]*4 + [vp]*8 + [ctypes.c_uint, vp]
    libs[name] = l
T, N, D = 256, 256, 8
dev = torch.device("cuda:0")
d = synthetic.smooth_field_task_stack(T, N, D, seed=1234)
ys, _, _ = synthetic.standardize_rows(d["Y"])
theta = np.concatenate([np.full((T, D), 0.5), np.full((T, 1), 1.0), np.full((T, 1), 1e-3)], 1)
X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
L = torch.zeros(T, N, N, dtype=torch.float64, device=dev); alpha = torch.empty(T, N, dtype=torch.float64, device=dev)
q, ld, mll, jit = (torch.empty(T, dtype=torch.float64, device=dev) for _ in range(4)); info = torch.empty(T, dtype=torch.int32, device=dev)
def run(l, flags):
    rc = l.scaml_gp_fit_fused_f64(X.data_ptr(), y.data_ptr(), th.data_ptr(), None, None, T, N, D, 1, L.data_ptr(), alpha.data_ptr(), q.data_ptr(), ld.data_ptr(), mll.data_ptr(), info.data_ptr(), jit.data_ptr(), None, flags, None)
    assert rc == 0
for n, l in libs.items():   # sanity line per build: a wrong kernel shows up as failed pivots / a different MLL sum
    run(l, 1); torch.cuda.synchronize()
    print(f"{n:28s} failed tasks {int((info != 0).sum())}  sum(mll) {float(mll.sum()):.12f}  max jitter {float(jit.max()):.1e}")
res = {(n, f): [] for n in libs for f in (1, 3)}
for rnd in range(8):
    for n, l in libs.items():
        for flags in (1, 3):
            for _ in range(3): run(l, flags)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run(l, flags)
            e1.record(); torch.cuda.synchronize()
            res[(n, flags)].append(e0.elapsed_time(e1) / 20 * 1e3)
for (n, f), v in res.items():
    print(f"{n:28s} flags={f} (zero_upper={'yes' if f & 2 else 'no '}): median {statistics.median(v):7.1f} us  min {min(v):7.1f} us")
