"""Per-launch durations of the fused fit right after a synchronisation (why does a 20-step timed region see 119.8 us per launch and
a 1000-step region 106.6 us on the same box?).  Events between consecutive launches; prints the first 48 and the tail average."""
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
for _ in range(2000):
    ops.gp_fit_fused(X, y, th, kind, out=out, zero_upper=True)
torch.cuda.synchronize()
for idle_ms in (0.0, 1.0, 20.0):
    for rep in range(2):
        for _ in range(500):
            ops.gp_fit_fused(X, y, th, kind, out=out, zero_upper=True)
        torch.cuda.synchronize()
        time.sleep(idle_ms * 1e-3)
        n = 300
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            ops.gp_fit_fused(X, y, th, kind, out=out, zero_upper=True)
            evs[i + 1].record()
        torch.cuda.synchronize()
        d = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]
        print(f"idle {idle_ms} ms rep {rep}: first 24:", " ".join(f"{v:.0f}" for v in d[:24]))
        print(f"   mean[0:20] {sum(d[:20]) / 20:.1f}  mean[20:60] {sum(d[20:60]) / 40:.1f}  mean[100:300] {sum(d[100:]) / 200:.1f} us")
