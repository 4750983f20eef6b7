"""Why does a 20-launch timed region cost 119.8 us per launch when the steady state is 106.6?  Variants of the bench's region on one box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd"))
import torch

import __graft_entry__ as entry

entry.build()
import bench
from scamlgp_amd import ops

dev = torch.device("cuda:0")
_, (X, y, th) = bench.make_inputs(0, dev)
kind = ops.KIND_MATERN52
out = ops.gp_fit_fused(X, y, th, kind)
out_g = ops.gp_fit_fused(X, y, th, kind, want_linv=True)
gws = ops.mll_backward_workspace(256, 256, 8, dev)
mll_rows = torch.zeros(64, 256, dtype=torch.float64, device=dev)


def launch(i=None):
    if i is not None:
        out["mll"] = mll_rows[i]
    ops.gp_fit_fused(X, y, th, kind, out=out, zero_upper=True)


def region(k, pre=None, rows=False, label=""):
    for _ in range(300):
        launch()
    torch.cuda.synchronize()
    if pre is not None:
        pre()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(k):
        launch(i if rows else None)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"{label:48s} k={k:4d}: events {e0.elapsed_time(e1) * 1e3 / k:7.1f} us/launch, wall {wall * 1e6 / k:7.1f}")


def grad_eval():
    ops.gp_fit_fused(X, y, th, kind, out=out_g, zero_upper=True, want_linv=True)
    ops.mll_backward(X, th, kind, out_g["L"], out_g["Linv_diag"], out_g["alpha"], workspace=gws).sum(0)


for rep in range(2):
    region(20, label="plain")
    region(200, label="plain")
    region(20, rows=True, label="mll rows")
    region(20, pre=grad_eval, label="after a gradient evaluation")
    region(20, pre=lambda: torch.sum(mll_rows[:5], 1), label="after a torch.sum")
    region(20, pre=lambda: time.sleep(0.0005), label="after 0.5 ms idle")
# the same 20 launches replayed from a HIP graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    launch()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    for i in range(20):
        launch()
for rep in range(2):
    for _ in range(300):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"graph replay of 20: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us/launch")
