"""Random shapes of scaml_target_mll_f64 (both factorisation paths) against torch autograd through the oracle -- a one-off sweep for
boundary cases: n around multiples of 16 and around the 112 / 128 limits, odd T (the unrolled task loops), D = 1 .. 16."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd"))
import numpy as np, torch
torch.set_num_threads(8)
from scamlgp_amd import _lib, hyper, ops
from tests._target_problem import make_target_problem, oracle_mll_and_grad, raw_start
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = [(1, 1, 1, 0), (15, 2, 1, 1), (16, 3, 16, 0), (17, 5, 2, 1), (31, 7, 3, 0), (32, 9, 4, 1), (33, 17, 5, 0), (47, 19, 6, 1), (48, 33, 7, 0), (63, 1, 8, 1),
         (64, 13, 9, 0), (65, 21, 10, 1), (95, 4, 11, 0), (96, 6, 12, 1), (97, 8, 13, 0), (111, 10, 14, 1), (112, 12, 15, 0), (113, 3, 16, 1), (127, 5, 2, 0), (128, 7, 3, 1)]
for _ in range(12):
    cases.append((int(rng.integers(1, 129)), int(rng.integers(1, 40)), int(rng.integers(1, 17)), int(rng.integers(0, 2))))
worst = 0.0
for n, T, D, kind in cases:
    if not ops.TargetFitProblem.supported(n, T, D):
        print(f"n={n} T={T} D={D}: not supported (LDS)"); continue
    prob = make_target_problem(n, T, D, kind, seed=n * 131 + T, n_src=max(12, 2 * D))
    tp = ops.TargetFitProblem(prob["source_means"].to(dev), prob["source_covs"].to(dev), prob["X"].to(dev), prob["y"].to(dev), prob["m_all"], prob["s_all"],
                              hyper.target_gp_spec(), hyper.GammaPrior(1.0, 1.0), 1e-10, kind)
    z = raw_start(D, T, seed=n + T, B=2)
    res = {}
    for path in (0, 1):
        was = _lib.lib.scaml_debug_target_fit_path(path)
        res[path] = ops.target_mll(tp, z.to(dev))
        _lib.lib.scaml_debug_target_fit_path(was)
    for b in range(2):
        val, g = oracle_mll_and_grad(prob, z[b])
        for path in (0, 1):
            o = res[path]
            ev = abs(o["value"][b].item() - float(val)) / max(1.0, abs(float(val)))
            eg = float((o["grad"][b].cpu() - g).abs().max() / max(1e-6, float(g.abs().max())))
            worst = max(worst, ev, eg)
            flag = "" if (ev < 1e-8 and eg < 1e-5 and int(o["info"][b]) == 0) else "   <-- CHECK"
            if flag or b == 0 and path == 0:
                print(f"n={n:3d} T={T:2d} D={D:2d} kind={kind} path={'elim' if path else 'mfma/auto'} b={b}: value err {ev:.1e} grad err {eg:.1e} info {int(o['info'][b])}{flag}")
print("worst relative error", worst)
