"""Phase breakdown of the fused fit kernel from the SCAML_STAMPS diagnostic build (GPU only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import synthetic
ONE = [a for a in sys.argv if a.startswith("--one-panel")]
if ONE:
    sys.argv.remove(ONE[0])
PER_PANEL = "--per-panel" in sys.argv
if PER_PANEL:
    sys.argv.remove("--per-panel")
_name = ("libscaml_hip_onepanel%s.so" % ONE[0][len("--one-panel"):]) if ONE else ("libscaml_hip_perpanel.so" if PER_PANEL else "libscaml_hip_stamps%s.so" % os.environ.get("STAMP_WAVE", "0"))
lib = ctypes.CDLL(os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "lib", _name))
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
    rc = lib.scaml_gp_fit_fused_f64(X.data_ptr(), y.data_ptr(), th.data_ptr(), None, None, T, N, D, kind, L.data_ptr(), alpha.data_ptr(), q.data_ptr(), ld.data_ptr(), mll.data_ptr(), info.data_ptr(), jit.data_ptr(), None, 1, None)
    assert rc == 0
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
if ONE:
    med = np.median(s[:, 0], 0); medp = np.median(s[:, 1], 0)
    nu = ["iter start", "cntT[k] complete", "U1 mfma+drain", "parked, arrived cntS", "flagW[k+1] seen", "cntY wait + v done", "cntU[k-1] seen", "TRSM issue+drain",
          "tiles set, panel written, arrived cntT", "fold, arrived cntY", "diag stored", "U2 done, arrived cntU", "column stored"]
    npn = [""] * 7
    for i in range(13):
        print(f"{nu[i]:40s} {med[i]:10.0f} {med[i]-(med[i-1] if i else med[0]):8.0f}   | " + "")
    sys.exit(0)
if PER_PANEL:
    med = np.median(s[:, 0], 0); medp = np.median(s[:, 1], 0)
    print("k   U1(k) done (update wave 0)   delta   | flagW[k] published (panel wave)   delta")
    for k in range(16):
        print(f"{k:2d} {med[k]:12.0f} {med[k]-(med[k-1] if k else 0):8.0f}   | {medp[k]:12.0f} {medp[k]-(medp[k-1] if k else 0):8.0f}")
    sys.exit(0)
# stamp slots: 0-2 common; 3-9 differ per role; 10-15 tail
names_u = {0: "load X,y", 9: "K-build (own tiles)", 1: "K-build: barrier + image loads", 2: "prologue (park tiles)", 3: "wait cntT[k]", 4: "U1: col k+1, D_k+2 + park", 5: "wait flagW[k+1]",
           6: "F: v, trsm, panel write, fold", 7: "U2: bulk update", 8: "column k+1 -> HBM", 15: "loop exit barrier", 11: "tail: M_k products+bar",
           12: "-", 13: "backsub: wait alpha + folds", 14: "backsub: barrier", 10: "tail: rest"}
names_p = {0: "load X,y", 8: "K-build (15 tile images)", 1: "K-build: barrier", 2: "prologue", 3: "wait cntS[j-2] (D_j, R_j)", 4: "own TRSM + diag update", 5: "potf2",
           9: "dl + publish", 15: "loop exit barrier",
           11: "tail: scalars+bar", 12: "backsub: chain steps", 13: "backsub: wait for w_k", 14: "backsub: barrier", 10: "tail: rest"}
med = np.median(s[:, 0], 0); medp = np.median(s[:, 1], 0); tot = med.sum(); totp = medp.sum()
order = [0, 9, 8, 1, 2, 3, 4, 5, 6, 7, 15, 11, 12, 13, 14, 10] if False else [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 11, 12, 13, 14, 10]
print(f"{'update wave 0':30s} {'cycles':>9s} {'%':>6s}   | {'panel wave':30s} {'cycles':>9s} {'%':>6s}")
for i in order:
    print(f"{names_u.get(i, '-'):30s} {med[i]:9.0f} {100*med[i]/tot:5.1f}%   | {names_p.get(i, '-'):30s} {medp[i]:9.0f} {100*medp[i]/totp:5.1f}%")
print(f"total {tot:.0f} / {totp:.0f} s_memtime ticks (update wave 0 / panel wave timeline)")
