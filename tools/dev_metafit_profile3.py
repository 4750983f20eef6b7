import cProfile, os, pstats, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import model as M, synthetic, utils
T, N = 32, 512
d = synthetic.hartmann6_task_stack(T, N, seed=0)
def mk():
    return M.SourceGPStack(list(range(T)), [torch.from_numpy(d["X"][t]) for t in range(T)], [torch.from_numpy(d["Y"][t]).unsqueeze(-1) for t in range(T)], kind=1)
for rep in range(3):
    st = mk(); torch.cuda.synchronize()
    t0 = time.perf_counter(); utils._fit_stack(st, num_restarts=1, max_iter=30); torch.cuda.synchronize()
    print(f"meta-fit rep {rep}: {time.perf_counter() - t0:.3f} s, evals {st.last_fit_info['n_eval']}, iters {st.last_fit_info['n_iter']}")
st = mk()
pr = cProfile.Profile(); pr.enable(); utils._fit_stack(st, num_restarts=1, max_iter=30); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
