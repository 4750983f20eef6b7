#!/usr/bin/env python3
"""bench.py — task-posteriors/sec of the fused GP hot path (K + Cholesky + solves + MLL).

One "step" = one pass of the hot path over one batch of synthetic input: a single launch of
scaml_gp_fit_fused_f64 over T = 256 meta-tasks x N = 256 points x D = 8, Matern-5/2 + ARD
(BASELINE.json configs[2], the configuration the metric is quoted on), inputs resident in HBM.
With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL) every rank
owns its own 256-task shard (weak scaling); the summed marginal likelihoods of all steps of the timed
region are all-reduced across ranks in one collective at its end (the path's only exchange step,
batched: a collective per step would take a CU from the next launch's one-per-CU workgroups).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — algorithmic fp64 flops per launch / measured kernel duration vs the fp64 MFMA peak
  cpu_baseline — the CPU oracle (oracle/gp_oracle.py, the reference's op sequence in torch fp64)
                 timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd"))

T_PER_GPU, N_POINTS, DIM = 256, 256, 8
PEAK_FP64_TFLOPS = 78.6  # MI355X fp64 matrix = vector peak: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz
                         # (measured: one v_mfma_f64_16x16x4_f64 per 64 cycles per SIMD, profiles/r01_probe_f64_rates.txt)


def algorithmic_flops_per_task(N: int, D: int, matern: bool) -> float:
    """N^3/3 [POTRF] + N(N+1)/2 (4D + c_k) [kernel, symmetric half]
    + 2 N^2 [two TRSV] + 3 N [quad, logdet]; c_k = 8 Matern-5/2, 2 RBF; FMA = 2."""
    ck = 8 if matern else 2
    return N ** 3 / 3.0 + N * (N + 1) / 2.0 * (4 * D + ck) + 2.0 * N * N + 3.0 * N


def algorithmic_bytes_per_task(N: int, D: int) -> float:
    """SURVEY.md §8(d), "L stored (full)": read X, y, theta; write the dense N x N factor (zeros above the
    diagonal included, every step), alpha, 3 scalars, info, jitter."""
    return 8.0 * (N * D + N + D + 2) + 8.0 * (N * N + N + 3) + 4 + 8


def recorded_pmc_traffic():
    """HBM bytes per launch of the fused-fit kernel from the rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE, one pass each, same command line as this bench) — recorded in profiles/, because
    counters cannot be collected from inside the timed process.  None if no record exists."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path))["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def make_inputs(rank: int, device):
    import numpy as np
    import torch
    from scamlgp_amd import synthetic

    d = synthetic.smooth_field_task_stack(T_PER_GPU, N_POINTS, DIM, seed=1234 + rank)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    rng = np.random.default_rng(4321 + rank)
    # reference inits (scamlgp/model.py:55, 67, 31): lengthscale 0.5 (ARD, +-20 % per dim so the
    # ARD path is exercised), outputscale 1.0, noise 1e-3
    theta = np.concatenate(
        [0.5 * (1 + 0.4 * (rng.uniform(size=(T_PER_GPU, DIM)) - 0.5)), np.full((T_PER_GPU, 1), 1.0), np.full((T_PER_GPU, 1), 1e-3)], 1)
    X = torch.from_numpy(d["X"])
    y = torch.from_numpy(ys)
    th = torch.from_numpy(theta)
    return (X, y, th), (X.to(device), y.to(device), th.to(device))


def cpu_baseline(host_inputs, budget_s: float = 8.0):
    """Time the oracle on the host cores over a bounded sample of the bench workload, two ways:
    the reference's shape of work — a per-task Python loop (scamlgp/model.py:176-188), one dense
    Cholesky per task — and the same arithmetic as one batched torch.linalg call.  The faster of
    the two is reported as the baseline value; both are named in `sample`."""
    import torch
    from oracle import gp_oracle as O

    X, y, th = host_inputs
    cores = torch.get_num_threads()
    O.gp_fit_stack_loop(X[:2], y[:2], th[:2], O.KIND_MATERN52)  # warm-up
    done = 0
    t0 = time.perf_counter()
    while True:
        lo = done % T_PER_GPU
        hi = min(lo + 8, T_PER_GPU)
        O.gp_fit_stack_loop(X[lo:hi], y[lo:hi], th[lo:hi], O.KIND_MATERN52)
        done += hi - lo
        el = time.perf_counter() - t0
        if el > budget_s or done >= 4 * T_PER_GPU:
            break
    loop_rate = done / el
    O.gp_fit_stack_batched(X[:8], y[:8], th[:8], O.KIND_MATERN52)  # warm-up
    chunk, done_b = 64, 0
    t0 = time.perf_counter()
    while True:
        lo = done_b % T_PER_GPU
        O.gp_fit_stack_batched(X[lo:lo + chunk], y[lo:lo + chunk], th[lo:lo + chunk], O.KIND_MATERN52)
        done_b += chunk
        el_b = time.perf_counter() - t0
        if el_b > budget_s or done_b >= 8 * T_PER_GPU:
            break
    batched_rate = done_b / el_b
    return dict(value=max(loop_rate, batched_rate), unit="task-posteriors/s", cores=cores, kind="port",
                sample=(f"torch-fp64 oracle on the bench workload: per-task loop {done} tasks in {el:.1f} s = {loop_rate:.1f}/s; "
                        f"batched torch.linalg (64-task chunks) {done_b} tasks in {el_b:.1f} s = {batched_rate:.1f}/s"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry

    entry.build()
    from scamlgp_amd import ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # SCAML_BENCH_REHEARSAL=1: rehearse the multi-rank control flow on ONE GPU (all ranks on cuda:0, gloo instead of
    # RCCL, which refuses two ranks on one device).  Numbers from such a run mean nothing; it exists so that the
    # barrier / all-reduce / max-over-ranks logic can be exercised on a one-GPU box.
    rehearsal = os.environ.get("SCAML_BENCH_REHEARSAL") == "1"
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(device)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    host_inputs, (X, y, th) = make_inputs(rank, device)
    kind = ops.KIND_MATERN52
    # allocates the output buffers once; every step rewrites them completely
    out = ops.gp_fit_fused(X, y, th, kind)
    total = args.warmup + args.steps
    # Multi-GPU: the only coupling of the shards is the sum of the per-task MLLs (scamlgp/model.py:129-134 sums over
    # tasks).  Every step writes its per-task values into its own row; ONE all-reduce per timed region carries the
    # sums of all its steps ("batch several evaluations per collective").  A collective per step would
    # put an RCCL workgroup on one of the 256 CUs the next launch's 256 one-per-CU workgroups need.
    mll_rows = torch.zeros(total, T_PER_GPU, dtype=torch.float64, device=device) if distributed else None
    works = []

    def step(i):
        if distributed:
            out["mll"] = mll_rows[i]
        ops.gp_fit_fused(X, y, th, kind, out=out, zero_upper=True)   # the full dense L, zeros included, every step

    def reduce_region(lo, hi):
        if distributed:
            sums = mll_rows[lo:hi].sum(1)
            works.append((dist.all_reduce(sums, async_op=True), sums))

    def fence():
        for w, _ in works:
            w.wait()
        works.clear()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    reduce_region(0, args.warmup)
    fence()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record()
    reduce_region(args.warmup, total)   # inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # launch stream only: avg duration per fused-fit launch

    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ok = not bool(out["info"].any().item())

    if rank == 0:
        flops = algorithmic_flops_per_task(N_POINTS, DIM, True) * T_PER_GPU
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        res = {
            "metric": "task-posteriors/sec (K+chol+solve+MLL) at T=256,N=256; 1/2/4/8 GPU",
            "value": world * T_PER_GPU * args.steps / elapsed,
            "unit": "task-posteriors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "configs[2]: 256 meta-tasks x 256 points x d=8, Matern-5/2 + ARD, per GPU "
                            "(fused K + jittered Cholesky + alpha + MLL, L stored)",
                "tasks_per_gpu": T_PER_GPU, "points": N_POINTS, "dim": DIM, "kernel": "matern52",
                "sharding": "task shards, one all-reduce of the per-step MLL sums per timed region" if distributed else "single GPU",
                "all_tasks_psd": ok,
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_TFLOPS, "traffic": recorded_pmc_traffic(),
                "kernel": "gp_fit_fused_kernel<16,7,matern52>",
                "kernel_ms": kernel_ms,
                "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": algorithmic_bytes_per_task(N_POINTS, DIM) * T_PER_GPU,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(host_inputs)
        print(json.dumps(res), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
