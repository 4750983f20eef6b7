"""Randomised comparison of the blocked fit (scaml_gp_fit_blocked_f64) with the composition of library launches it replaces:
random N (multiples of 16 in 272 .. 512), D (1 .. scaml_fit_blocked_max_d()), kernel kind, ragged point counts, zero_upper."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import _lib, ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dmax = _lib.lib.scaml_fit_blocked_max_d()
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    N = int(rng.choice(np.arange(272, 513, 16)))
    D = int(rng.integers(1, dmax + 1))
    T = int(rng.integers(1, 7))
    kind = int(rng.integers(0, 2))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.randn(T, N, dtype=torch.float64, generator=g)
    theta = torch.cat([0.3 * np.sqrt(D) * (0.5 + torch.rand(T, D, dtype=torch.float64, generator=g)), 0.5 + torch.rand(T, 1, dtype=torch.float64, generator=g),
                       1e-3 + 1e-2 * torch.rand(T, 1, dtype=torch.float64, generator=g)], 1)
    npts = None
    if rng.random() < 0.5:
        npts = torch.from_numpy(rng.integers(1, N + 1, size=T).astype(np.int32))
        if rng.random() < 0.5:
            npts[0] = N
    zu = bool(rng.random() < 0.5)
    args = (X.to(dev), y.to(dev), theta.to(dev), kind)
    kw = dict(n_points=None if npts is None else npts.to(dev), zero_upper=zu)
    a = ops.gp_fit_fused(*args, **kw)
    ops._FORCE_COMPOSED_TWO_BLOCK = True
    b = ops.gp_fit_fused(*args, **kw)
    ops._FORCE_COMPOSED_TWO_BLOCK = False
    assert a["info"].cpu().tolist() == b["info"].cpu().tolist(), (N, D, T, kind, npts)
    ok = (a["info"] == 0).cpu()
    La, Lb = torch.tril(a["L"]).cpu()[ok], torch.tril(b["L"]).cpu()[ok]
    errs = [float((La - Lb).abs().max()), float(((a["alpha"] - b["alpha"]).abs().max() / b["alpha"].abs().max()).cpu()),
            float((a["mll"] - b["mll"]).abs().cpu()[ok].max()), float((a["Linv_diag"] - b["Linv_diag"]).abs().max())]
    if zu:
        assert float(torch.triu(a["L"], 1).abs().max()) == 0.0
    worst = max(worst, *errs)
    assert max(errs) < 1e-8, (N, D, T, kind, npts, zu, errs)
print(f"blocked-fit fuzz ok: worst deviation from the composition {worst:.2e}")
