"""Developer stress test for the wave hand-off protocol of the fused fit kernel (needs a GPU):
repeats the same launches many times and compares L bit-for-bit (it has no order-dependent sums)
and alpha / scalars to 1e-12 against the first run; also the POTRF entry point.  A mismatch means a race."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import torch
from scamlgp_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for (T, N, D, kind) in [(512, 256, 8, 1), (512, 128, 4, 0), (512, 200, 3, 1), (1024, 64, 2, 0), (1024, 32, 2, 1), (512, 100, 5, 0), (200, 128, 4, 1), (256, 100, 5, 0)]:   # the last two: the wide 8-wave variant for N <= 128 (T <= CUs)
    X = torch.rand(T, N, D, dtype=torch.float64, device=dev)
    y = torch.randn(T, N, dtype=torch.float64, device=dev)
    theta = torch.cat([0.3 + torch.rand(T, D, dtype=torch.float64, device=dev), torch.ones(T, 1, dtype=torch.float64, device=dev),
                       torch.full((T, 1), 1e-3, dtype=torch.float64, device=dev)], 1)
    ref = ops.gp_fit_fused(X, y, theta, kind, want_linv=True)
    K = ops.kernel_matrix(X, theta, kind, add_noise=True)
    refp = ops.potrf_batched(K, y, want_linv=True)
    torch.cuda.synchronize()
    assert not ref["info"].any()
    nbad = 0
    for r in range(reps):
        out = ops.gp_fit_fused(X, y, theta, kind, want_linv=True)
        outp = ops.potrf_batched(K, y, want_linv=True)
        for o, rf in ((out, ref), (outp, refp)):
            if not torch.equal(o["L"], rf["L"]) or not torch.equal(o["Linv_diag"], rf["Linv_diag"]) or not torch.equal(o["logdet"], rf["logdet"]):
                nbad += 1
            if not torch.allclose(o["alpha"], rf["alpha"], rtol=1e-9, atol=1e-12) or not torch.allclose(o["quad"], rf["quad"], rtol=1e-11):
                nbad += 1
    # the two entry points agree with each other, too
    d = (ref["L"] - refp["L"]).abs().max().item()
    print(f"T={T} N={N} D={D} kind={kind}: {nbad} mismatching runs of {2 * reps}; fit vs potrf max |dL| = {d:.2e}", flush=True)
    bad += nbad
print("RACE CHECK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
