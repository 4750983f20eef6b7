"""Phase breakdown of the fused fit kernel from the SCAML_STAMPS diagnostic build (GPU only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import synthetic
lib = ctypes.CDLL(os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "lib", "libscaml_hip_stamps.so"))
vp = ctypes.c_void_p
lib.scaml_gp_fit_fused_f64.argtypes = [vp]*5 + [ctypes.c_int]*4 + [vp]*8 + [ctypes.c_uint, vp]
lib.scaml_debug_set_stamp_buffer.argtypes = [vp]
T, N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 8
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
d = synthetic.smooth_field_task_stack(T, N, D, seed=1234)
ys, _, _ = synthetic.standardize_rows(d["Y"])
theta = np.concatenate([np.full((T, D), 0.5), np.full((T, 1), 1.0), np.full((T, 1), 1e-3)], 1)
X, y, th = (torch.from_numpy(a).to(dev) for a in (d["X"], ys, theta))
L = torch.empty(T, N, N, dtype=torch.float64, device=dev); alpha = torch.empty(T, N, dtype=torch.float64, device=dev)
q, ld, mll, jit = (torch.empty(T, dtype=torch.float64, device=dev) for _ in range(4)); info = torch.empty(T, dtype=torch.int32, device=dev)
stamps = torch.zeros(T, 2, 16, dtype=torch.int64, device=dev)
assert lib.scaml_debug_set_stamp_buffer(stamps.data_ptr()) == 0
for _ in range(3):
    rc = lib.scaml_gp_fit_fused_f64(X.data_ptr(), y.data_ptr(), th.data_ptr(), None, None, T, N, D, kind, L.data_ptr(), alpha.data_ptr(), q.data_ptr(), ld.data_ptr(), mll.data_ptr(), info.data_ptr(), jit.data_ptr(), None, 3, None)
    assert rc == 0
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
names = ["load X,y", "K-build", "prologue (spill0+potf2)", "T: trsm+final stores", "Z barrier wait", "U1: col k+1 upd+spill", "X barrier wait", "U2: bulk update", "Y barrier wait (potf2)", "loop exit barrier", "tail: rest", "tail: scalars+copy+bar", "backsub: matvec", "backsub: tiles", "backsub: barrier"]
med = np.median(s[:, 0], 0); medp = np.median(s[:, 1], 0); tot = med[:15].sum()
print(f"{'phase':26s} {'update wave 0':>14s} {'panel wave':>12s}")
for i, nm in enumerate(names):
    print(f"{nm:26s} {med[i]:10.0f} cyc  {100*med[i]/tot:5.1f}%  {medp[i]:10.0f}")
print(f"total {tot:.0f} shader cycles (update wave 0 timeline) ~ {tot/2.4e3:.1f} us at 2.4 GHz")
