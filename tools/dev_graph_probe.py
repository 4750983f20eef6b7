"""Which pieces of the acquisition pass can be captured into a HIP graph (torch.cuda.graph)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic, model as M, utils
dev = torch.device("cuda:0")
T, N, D, n, Mc = 8, 128, 3, 20, 64
d = synthetic.smooth_field_task_stack(T, N, D, seed=0)
st = M.SourceGPStack(list(range(T)), [torch.from_numpy(d["X"][t]) for t in range(T)], [torch.from_numpy(d["Y"][t]).unsqueeze(-1) for t in range(T)], kind=1)
f = st.refresh()
gps = {tid: M.SourceGP(st, i) for i, tid in enumerate(st.task_ids)}
Xt = torch.rand(n, D, dtype=torch.float64)
mdl = M.ScaMLGP(Xt, torch.rand(n, 1, dtype=torch.float64), gps).eval()
x = torch.rand(n + Mc, D, dtype=torch.float64, device=dev)
cand = torch.rand(Mc, D, dtype=torch.float64, device=dev)
w = torch.rand(T, dtype=torch.float64, device=dev)
pieces = {
    "gp_fit_fused": lambda: ops.gp_fit_fused(st.X, st.y, st.theta, st.kind),
    "weighted_task_sum": lambda: ops.weighted_task_sum(f["alpha"], w, 1),
    "posterior_linv (no cov)": lambda: ops.source_posteriors(x, st.X, st.theta, st.kind, None, None, f["alpha"], st.y_mean, st.y_std, Linv=f["Linv"]),
    "posterior fused cov": lambda: ops.source_posteriors(x, st.X, st.theta, st.kind, None, None, f["alpha"], st.y_mean, st.y_std, Linv=f["Linv"], cov_first=n),
    "_active_tasks": lambda: mdl._active_tasks(),
    "_source_prior": lambda: mdl._source_prior(x, n),
    "theta": lambda: mdl.theta,
    "posterior": lambda: mdl.posterior(cand).mvn.mean,
}
for name, fn in pieces.items():
    try:
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        print(f"{name:28s} captured and replayed")
    except Exception as e:
        print(f"{name:28s} FAILED: {type(e).__name__}: {str(e).splitlines()[0][:120]}")
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
