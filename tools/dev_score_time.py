"""C5 acquisition scoring pass: source posteriors at n + M points with the covariance block fused into the pass vs through V in
memory (same process, interleaved), and the whole UCB evaluation."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scamlgp_amd import ops, synthetic, model as M, utils
dev = torch.device("cuda:0")
def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
T, N, D, n, Mc = 32, 512, 6, 80, 1024
d = synthetic.hartmann6_task_stack(T, N, seed=0)
st = M.SourceGPStack(list(range(T)), [torch.from_numpy(d["X"][t]) for t in range(T)], [torch.from_numpy(d["Y"][t]).unsqueeze(-1) for t in range(T)], kind=1)
f = st.refresh()
x = torch.rand(n + Mc, D, dtype=torch.float64, device=dev)
args = (x, st.X, st.theta, st.kind, None, None, f["alpha"], st.y_mean, st.y_std)
res = {"fused": [], "V in memory": [], "no cov": []}
for _ in range(4):
    res["fused"].append(timeit(lambda: ops.source_posteriors(*args, cov_first=n, Linv=f["Linv"])))
    res["V in memory"].append(timeit(lambda: ops.source_posteriors(*args, cov_first=n, Linv=f["Linv"], keep_V=True)))
    res["no cov"].append(timeit(lambda: ops.source_posteriors(*args, cov_first=0, Linv=f["Linv"])))
for k, v in res.items():
    print(f"source posteriors T={T} N={N} at {n}+{Mc} points, cov block {k:12s}: median {statistics.median(v):8.1f} us")
gps = {tid: M.SourceGP(st, i) for i, tid in enumerate(st.task_ids)}
g = torch.Generator().manual_seed(0)
Xt = torch.rand(n, D, dtype=torch.float64, generator=g)
mdl = M.ScaMLGP(Xt, torch.from_numpy(synthetic.hartmann6(Xt.numpy())).unsqueeze(-1), gps).eval()
cand = torch.rand(Mc, D, dtype=torch.float64, generator=g)
acq = utils.UpperConfidenceBound(mdl)
print(f"UCB scoring pass: {timeit(lambda: acq(cand)):8.1f} us")
# the same pass captured once into a HIP graph and replayed: what remains is GPU time
try:
    out_static = acq(cand)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            acq(cand)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out_g = acq(cand)
    torch.cuda.synchronize()
    us = timeit(lambda: g.replay())
    print(f"UCB scoring pass replayed from a HIP graph: {us:8.1f} us; max abs diff to eager {float((out_g - out_static).abs().max()):.2e}")
except Exception as e:
    print("graph capture failed:", type(e).__name__, e)
from scamlgp_amd.bo import GraphedAcquisition
for Mb in (130, 1024):
    xb = torch.rand(Mb, D, dtype=torch.float64, device=dev)
    ga = GraphedAcquisition(acq, Mb, D, dev)
    print(f"UCB at {Mb:4d} points: eager {timeit(lambda: acq(xb)):8.1f} us, HIP-graph replay {timeit(lambda: ga(xb)):8.1f} us")
