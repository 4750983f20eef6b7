"""Phase stamps of the several-CUs-per-task fit (csrc/gp_fit_coop.hip; library built with
python __graft_entry__.py --variant coopstamps -DCF_STAMPS): 100 MHz wall clock, task 0, thread 0 of every part."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import synthetic
vp = ctypes.c_void_p
lib = ctypes.CDLL(os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "lib", "libscaml_hip_coopstamps.so"))
lib.scaml_gp_fit_blocked_workspace_bytes.restype = ctypes.c_longlong
lib.scaml_gp_fit_blocked_f64.argtypes = [vp] * 5 + [ctypes.c_int] * 4 + [vp] * 8 + [ctypes.c_uint, vp, ctypes.c_longlong, vp]
T, N, D = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 512, 6)))
dev = torch.device("cuda:0")
d = synthetic.smooth_field_task_stack(T, N, D, seed=1)
ys, _, _ = synthetic.standardize_rows(d["Y"])
theta = np.concatenate([np.full((T, D), 0.5), np.ones((T, 1)), np.full((T, 1), 1e-3)], 1)
X, y, th = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (d["X"], ys, theta))
L = torch.empty(T, N, N, dtype=torch.float64, device=dev); alpha = torch.empty(T, N, dtype=torch.float64, device=dev)
q, ld, mll, jit = (torch.empty(T, dtype=torch.float64, device=dev) for _ in range(4)); info = torch.empty(T, dtype=torch.int32, device=dev)
W = torch.empty(T, N // 16, 16, 16, dtype=torch.float64, device=dev)
nbytes = lib.scaml_gp_fit_blocked_workspace_bytes(T, N)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
lib.scaml_debug_blocked_fit_path(2)
for _ in range(3):
    rc = lib.scaml_gp_fit_blocked_f64(X.data_ptr(), y.data_ptr(), th.data_ptr(), None, None, T, N, D, 1, L.data_ptr(), alpha.data_ptr(), q.data_ptr(), ld.data_ptr(),
                                      mll.data_ptr(), info.data_ptr(), jit.data_ptr(), W.data_ptr(), 1 | 2, ws.data_ptr(), nbytes, None)
    assert rc == 0
torch.cuda.synchronize()
assert int(info.abs().max()) == 0
P = min(8, (N + 31) // 32, 256 // T)
NB = (N + 31) // 32
flag_bytes = (T * 44 * 4 + 15) // 16 * 16
off = flag_bytes + (T * N + T * 64) * 8
st = ws[off:off + 8 * 32 * 16 * 8].view(torch.float64).view(8, 32, 16).cpu().numpy() * 0.01   # -> us
t0 = min(st[p, 0, 0] for p in range(P))
print(f"T={T} N={N} D={D}, {P} parts per task; times in us from the first part's start")
print(" col part | start  sums->  barrier potf2 (sweep 1, glue, sweep 2, W21)   barrier  v/W     stores  published | column end")
for j in range(NB):
    p, k = j % P, j // P
    s = st[p, k] - t0
    nxt = (st[p, k + 1, 0] if (j + P < NB) else st[p, 31, 0]) - t0
    print(f" {j:3d} {p:4d} | {s[0]:6.1f} {s[2]:7.1f} {s[3]-s[2]:7.1f} {s[4]-s[3]:7.1f} ({s[8]-s[3]:4.1f} {s[9]-s[8]:4.1f} {s[10]-s[9]:4.1f} {s[4]-s[10]:4.1f}) {s[5]-s[4]:7.1f} {s[6]-s[5]:7.1f} {s[7]-s[6]:7.1f}  {s[7]:7.1f} | {nxt:7.1f}")
lp = (NB - 1) % P
print(f"finish (part {lp}): {st[lp, 31, 0] - t0:.1f} -> {st[lp, 31, 1] - t0:.1f} us; inside: scalars done {st[lp, 30, 0] - t0:.1f}, first rows in {st[lp, 30, 1] - t0:.1f}, every 4 steps: " + " ".join(f"{st[lp, 30, 2 + k] - t0:.1f}" for k in range(8)))
