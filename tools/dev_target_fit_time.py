"""Timing of the target-GP kernel (csrc/gp_target_fit.hip) at BASELINE configs[4] shapes: one objective + gradient evaluation
(scaml_target_mll_f64) and the whole refit (scaml_target_fit_f64) for B start points, against the torch / scipy path it replaces."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd"))
import torch

import __graft_entry__ as entry

entry.build()
from scamlgp_amd import hyper, ops
from tests._target_problem import make_target_problem, raw_start

dev = torch.device("cuda:0")
shapes = [(21, 32, 6, 1), (80, 32, 6, 1), (128, 32, 6, 1), (80, 4, 2, 0)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for n, T, D, kind in shapes:
    prob = make_target_problem(n, T, D, kind, seed=1, n_src=16)
    tp = ops.TargetFitProblem(prob["source_means"].to(dev), prob["source_covs"].to(dev), prob["X"].to(dev), prob["y"].to(dev), prob["m_all"],
                              prob["s_all"], hyper.target_gp_spec(), hyper.GammaPrior(1.0, 1.0), 1e-10, kind)
    for B in (1, 3, 6):
        z = raw_start(D, T, seed=3, B=B).to(dev)
        ops.target_mll(tp, z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.target_mll(tp, z)
        e1.record()
        torch.cuda.synchronize()
        t_eval = e0.elapsed_time(e1) / 20 * 1e3
        ops.target_fit(tp, z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ops.target_fit(tp, z)
        torch.cuda.synchronize()
        t_fit = (time.perf_counter() - t0) * 1e3
        st = res["stats"].cpu().tolist()
        ev = max(s[1] for s in st)
        print(f"n={n:4d} T={T:3d} D={D} kind={kind} B={B}: eval+grad {t_eval:7.1f} us/launch; refit {t_fit:7.2f} ms, max evals {ev} -> "
              f"{t_fit * 1e3 / max(ev, 1):6.1f} us/eval; stats {st}")
