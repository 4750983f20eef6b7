"""Cycle stamps of the blocked fit's strip-solve and finish kernels (library built with -DBK_STAMPS:
python __graft_entry__.py --variant blkstamps -DBK_STAMPS); 100 MHz s_memtime ticks -> us."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import synthetic
vp = ctypes.c_void_p
lib = ctypes.CDLL(os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "lib", "libscaml_hip_blkstamps.so"))
lib.scaml_gp_fit_blocked_workspace_bytes.restype = ctypes.c_longlong
lib.scaml_gp_fit_blocked_f64.argtypes = [vp] * 5 + [ctypes.c_int] * 4 + [vp] * 8 + [ctypes.c_uint, vp, ctypes.c_longlong, vp]
T, N, D = 32, 512, 6
dev = torch.device("cuda:0")
d = synthetic.hartmann6_task_stack(T, N, seed=0)
ys, _, _ = synthetic.standardize_rows(d["Y"])
theta = np.concatenate([np.full((T, D), 0.6), np.ones((T, 1)), np.full((T, 1), 1e-2)], 1)
X, y, th = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (d["X"], ys, theta))
L = torch.empty(T, N, N, dtype=torch.float64, device=dev); alpha = torch.empty(T, N, dtype=torch.float64, device=dev)
q, ld, mll, jit = (torch.empty(T, dtype=torch.float64, device=dev) for _ in range(4)); info = torch.empty(T, dtype=torch.int32, device=dev)
W = torch.empty(T, N // 16, 16, 16, dtype=torch.float64, device=dev)
nbytes = lib.scaml_gp_fit_blocked_workspace_bytes(T, N)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
for _ in range(3):
    rc = lib.scaml_gp_fit_blocked_f64(X.data_ptr(), y.data_ptr(), th.data_ptr(), None, None, T, N, D, 1, L.data_ptr(), alpha.data_ptr(), q.data_ptr(), ld.data_ptr(),
                                      mll.data_ptr(), info.data_ptr(), jit.data_ptr(), W.data_ptr(), 1 | 4, ws.data_ptr(), nbytes, None)
    assert rc == 0
torch.cuda.synchronize()
N2 = N - 256
off = T * N2 * N2 * 8 + T * 65536 * 8
r2 = ws[off:off + T * N * 8].view(torch.float64).view(T, N).cpu().numpy()
tick = 1.0   # s_memtime counts shader-clock cycles here
st = np.median(r2[:, :128], axis=0) * tick
print(f"strip solve (cycles): prologue {st[0]:.0f}, end of loop {st[4 + 4 * 15]:.0f}, strip stored {st[70]:.0f}")
prev = st[0]
for kb in range(16):
    b, e = st[2 + 4 * kb], st[4 + 4 * kb]
    print(f"  kb {kb:2d}: owner wave: block read + chain {b - prev:6.0f} cycles, barrier (helper wave: loads, kernel block) {e - b:6.0f}")
    prev = e
print(f"finish (cycles): mat-vec done at {st[100]:.0f}, chain done at {st[101]:.0f}")
