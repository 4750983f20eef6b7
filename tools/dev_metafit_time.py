import os, sys, time
torch_seed = 0
sys.path.insert(0, "/root/repo/scalable-meta-learning-with-gaussian-processes_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scamlgp_amd import model as M, synthetic, utils
for T, N in ((8, 32), (32, 32), (32, 128), (256, 128)):
    d = synthetic.branin_task_stack(T, N, seed=0, noise_std=1.0)
    mk = lambda: M.SourceGPStack(list(range(T)), [torch.from_numpy(d["X"][t]) for t in range(T)], [torch.from_numpy(d["Y"][t]).unsqueeze(-1) for t in range(T)], kind=0)
    utils._fit_stack(mk(), num_restarts=1, max_iter=5); torch.cuda.synchronize()
    torch.manual_seed(0); st = mk(); t0 = time.perf_counter(); utils._fit_stack(st, num_restarts=5, max_iter=200, use_graph=os.environ.get('NOGRAPH') is None); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    obj = float(st.last_fit_info["objective_sum"]); print(f"T={T} N={N}: objective {obj:.10f}; meta-fit with 5 restarts {dt:.3f} s, {st.last_fit_info['n_iter']} iterations, {st.last_fit_info['n_eval']} evaluations -> {dt / st.last_fit_info['n_eval'] * 1e3:.2f} ms per evaluation")
