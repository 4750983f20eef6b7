import os, sys, time, cProfile, pstats
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic, model as M, utils
for (T, N, D, name) in [(64, 128, 2, "C2"), (32, 512, 6, "C5")]:
    d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
    meta = {t: M.SupervisedDataset(torch.from_numpy(d["X"][t]), torch.from_numpy(d["Y"][t]).unsqueeze(-1)) for t in range(T)}
    stack = M.SourceGPStack(list(meta.keys()), [v.X() for v in meta.values()], [v.Y() for v in meta.values()], kind=1)
    utils._fit_stack(stack, num_restarts=1, max_iter=5)   # warm-up
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    utils._fit_stack(stack, num_restarts=1, max_iter=30)
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    info = stack.last_fit_info
    print(name, f"{dt*1e3:.1f} ms, iters {info['n_iter']}, evals {info['n_eval']} -> {dt/info['n_eval']*1e3:.2f} ms per evaluation")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
