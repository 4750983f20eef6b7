#!/usr/bin/env python3
"""bench.py — task-posteriors/sec of the fused GP hot path (K + Cholesky + solves + MLL).

One "step" = one pass of the hot path over one batch of synthetic input: a single launch of
scaml_gp_fit_fused_f64 over T = 256 meta-tasks x N = 256 points x D = 8, Matern-5/2 + ARD
(BASELINE.json configs[2], the configuration the metric is quoted on), inputs resident in HBM.
``--step fit+grad`` adds the analytic hyper-gradient (scaml_mll_backward_f64) to every step: one
L-BFGS evaluation of the meta-fit (the reference's HOT LOOP #2, scamlgp/utils.py:175).

Multi-GPU (BASELINE.json configs[3]: task shards + "RCCL all-reduce of the MLL hyper-gradient"):
``python bench.py --gpus N`` starts N fresh child ranks itself (torch.distributed.run, one rank per
GPU over RCCL); the driver may also launch the ranks directly.  Default: every rank owns its own
256-task shard (weak scaling).  ``--total-tasks 1024`` is configs[3] as BASELINE.json states it: 1024
tasks in all, ``dist.shard_range(1024, N, rank)`` of them per rank (strong scaling; 128 per GPU at N = 8).
The shards' only coupling is a sum over tasks (scamlgp/model.py:129-134): ONE all-reduce per timed
region carries the fused buffer
    [ sum_t MLL_t of every step of the region  ||  sum_t dMLL_t/dtheta (D + 2) ]
The timed region is EXACTLY the K steps + that one reduce; nothing else is enqueued inside it.  With
``--step fit`` the gradient slot of the buffer is filled by one fit + backward evaluation BEFORE the
region (it is not part of the metric's step); with ``--step fit+grad`` every step adds its gradient.
The collective is batched on purpose: a collective per step would put an RCCL workgroup on one of the
256 CUs the next launch's one-per-CU workgroups need.
Clock ramp: a fresh process finds the GPU at idle clocks, and 25 launches do not bring it to its
steady state; a FIXED, untimed prologue of ``--prewarm`` launches of the same kernel (default 2000,
~0.2 s; reported as ``prewarm_launches``) runs before the ``--warmup`` steps at every N.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — algorithmic fp64 flops per launch / measured kernel duration vs the fp64 MFMA peak
  cpu_baseline — the CPU oracle (oracle/gp_oracle.py, the reference's op sequence in torch fp64)
                 timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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


def algorithmic_backward_flops_per_task(N: int, D: int, matern: bool) -> float:
    """SURVEY.md §8(d) "Backward add-on": ~N^3 (L^-1: N^3/3, K^-1 = L^-T L^-1 lower half: N^3/3 ... the survey
    rounds to N^3; counted here as 2 N^3 / 3) + N^2 (4D + c) for G o dK/dtheta."""
    ck = 8 if matern else 2
    return 2.0 * N ** 3 / 3.0 + float(N) * N * (4 * D + ck)


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


def make_inputs(rank: int, device, T: int = T_PER_GPU):
    import numpy as np
    import torch
    from scamlgp_amd import synthetic

    d = synthetic.smooth_field_task_stack(T, N_POINTS, DIM, seed=1234 + rank)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    rng = np.random.default_rng(4321 + rank)
    # reference inits (scamlgp/model.py:55, 67, 31): lengthscale 0.5 (ARD, +-20 % per dim so the
    # ARD path is exercised), outputscale 1.0, noise 1e-3
    theta = np.concatenate(
        [0.5 * (1 + 0.4 * (rng.uniform(size=(T, DIM)) - 0.5)), np.full((T, 1), 1.0), np.full((T, 1), 1e-3)], 1)
    X = torch.from_numpy(d["X"])
    y = torch.from_numpy(ys)
    th = torch.from_numpy(theta)
    return (X, y, th), (X.to(device), y.to(device), th.to(device))


# ---- CPU baseline ------------------------------------------------------------------------------------
def _pool_init():
    # worker of the process-pool leg: a fresh child that never touches the GPU, one thread
    # (the reference's deployment shape: scamlgp/benchmarking/local_runner.py:107-108, 174-181)
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    import torch

    torch.set_num_threads(1)


def _pool_work(args):
    import torch
    from oracle import gp_oracle as O

    X, y, th, reps = args
    X, y, th = torch.from_numpy(X), torch.from_numpy(y), torch.from_numpy(th)
    t0 = time.perf_counter()
    for _ in range(reps):
        O.gp_fit_stack_loop(X, y, th, O.KIND_MATERN52)
    return reps * X.shape[0], time.perf_counter() - t0


def cpu_baseline(host_inputs, budget_s: float = 4.0):
    """Time the oracle on the host cores over a bounded sample of the bench workload (~25 s in all):
      * the reference's shape of work -- a per-task Python loop (scamlgp/model.py:176-188), one dense Cholesky per
        task -- at 1, 8, 32 and all torch threads (256 x 256 matrices do not scale to 128 threads);
      * the same arithmetic as one batched torch.linalg call (64-task chunks) at the best of those settings;
      * the reference's own deployment shape: P = min(64, cores) single-thread worker processes
        (scamlgp/benchmarking/local_runner.py:107-108, 174-181), each looping over its own tasks.
    The best rate is the reported value; every leg is named in `sample`."""
    import multiprocessing as mp

    import torch
    from oracle import gp_oracle as O

    X, y, th = host_inputs
    T_PER_GPU = int(X.shape[0])   # (the sample cycles through the rank's own stack)
    cores = os.cpu_count() or torch.get_num_threads()
    all_threads = torch.get_num_threads()
    O.gp_fit_stack_loop(X[:2], y[:2], th[:2], O.KIND_MATERN52)  # warm-up
    legs = []
    best_threads, best_loop = 1, 0.0
    for nt in sorted({1, min(8, cores), min(32, cores), all_threads}):
        torch.set_num_threads(nt)
        O.gp_fit_stack_loop(X[:2], y[:2], th[:2], O.KIND_MATERN52)
        done, t0 = 0, time.perf_counter()
        while True:
            lo = done % T_PER_GPU
            hi = min(lo + 8, T_PER_GPU)
            O.gp_fit_stack_loop(X[lo:hi], y[lo:hi], th[lo:hi], O.KIND_MATERN52)
            done += hi - lo
            el = time.perf_counter() - t0
            if el > budget_s * 0.6 or done >= 4 * T_PER_GPU:
                break
        rate = done / el
        legs.append((f"per-task loop, {nt} thread{'s' if nt > 1 else ''}", rate, nt, f"{done} tasks in {el:.1f} s"))
        if rate > best_loop:
            best_loop, best_threads = rate, nt
    torch.set_num_threads(best_threads)
    O.gp_fit_stack_batched(X[:8], y[:8], th[:8], O.KIND_MATERN52)  # warm-up
    chunk, done_b, t0 = 64, 0, time.perf_counter()
    while True:
        lo = done_b % T_PER_GPU
        O.gp_fit_stack_batched(X[lo:lo + chunk], y[lo:lo + chunk], th[lo:lo + chunk], O.KIND_MATERN52)
        done_b += chunk
        el_b = time.perf_counter() - t0
        if el_b > budget_s or done_b >= 8 * T_PER_GPU:
            break
    legs.append((f"batched torch.linalg (64-task chunks), {best_threads} threads", done_b / el_b, best_threads, f"{done_b} tasks in {el_b:.1f} s"))
    torch.set_num_threads(all_threads)
    # process pool: fresh children (spawn), no GPU, one thread each
    P = max(1, min(64, cores))
    try:
        ctx = mp.get_context("spawn")
        per = 4  # tasks per worker and repetition
        Xn, yn, thn = X.numpy(), y.numpy(), th.numpy()

        def jobs(reps):
            return [(Xn[(w * per) % T_PER_GPU:(w * per) % T_PER_GPU + per], yn[(w * per) % T_PER_GPU:(w * per) % T_PER_GPU + per],
                     thn[(w * per) % T_PER_GPU:(w * per) % T_PER_GPU + per], reps) for w in range(P)]

        with ctx.Pool(P, initializer=_pool_init) as pool:
            warm = pool.map(_pool_work, jobs(1), chunksize=1)  # imports + first call
            per_rep = max(sum(t for _, t in warm) / len(warm), 1e-4)
            reps = max(1, min(200, int(budget_s / per_rep)))
            t0 = time.perf_counter()
            res = pool.map(_pool_work, jobs(reps), chunksize=1)
            el_p = time.perf_counter() - t0
        done_p = sum(n for n, _ in res)
        legs.append((f"{P} single-thread worker processes (per-task loop each)", done_p / el_p, P, f"{done_p} tasks in {el_p:.1f} s"))
    except Exception as e:  # a box that cannot spawn workers still reports the in-process legs
        legs.append((f"process pool failed: {type(e).__name__}: {e}", 0.0, 0, ""))
    best = max(legs, key=lambda l: l[1])
    sample = "torch-fp64 oracle on the bench workload (T=256 stack, N=256, D=8, Matern-5/2); " + "; ".join(
        f"{name}: {what} = {rate:.1f}/s" for name, rate, _, what in legs) + f"; reported: {best[0]}"
    return dict(value=best[1], unit="task-posteriors/s", cores=best[2], kind="port", sample=sample, host_cores=cores)


# ---- parent: start the ranks ---------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _visible_gpus(env):
    """Number of GPUs a rank will see, asked of a child process that exits at once (None if it cannot tell)."""
    try:
        res = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True, timeout=600)
        return int(res.stdout.strip().splitlines()[-1])
    except Exception:
        return None


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: build once, then start N fresh child ranks
    (no GPU call has happened in this process) and relay their output and exit code."""
    import __graft_entry__ as entry

    # once, here: the ranks then find an up-to-date library and do not race writing it.  This process neither imports torch nor
    # loads the library nor makes a HIP call: a launcher that has opened the GPU must not be the parent of the ranks.
    entry.build(import_package=False)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("SCAML_BENCH_REHEARSAL") != "1":
        have = _visible_gpus(env)   # counted by a throwaway child
        if have is not None and have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, {have} visible "
                             "(SCAML_BENCH_REHEARSAL=1 rehearses the multi-rank control flow on one GPU; its numbers mean nothing)\n")
            return 2
    env["SCAML_BENCH_PARENT_BUILT"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--step", args.step,
           "--prewarm", str(args.prewarm), "--total-tasks", str(args.total_tasks)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in res.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    if res.returncode != 0:
        return res.returncode
    return 0 if line is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--step", choices=("fit", "fit+grad"), default="fit",
                    help="fit: the metric's step (K + Cholesky + solves + MLL); fit+grad: + the hyper-gradient every step")
    ap.add_argument("--prewarm", type=int, default=2000,
                    help="fixed untimed prologue: launches of the step's kernel before --warmup, to bring a fresh GPU to steady clocks")
    ap.add_argument("--total-tasks", type=int, default=0,
                    help="strong scaling: this many tasks IN ALL, sharded over the ranks (1024 = BASELINE configs[3]); "
                         "0 = weak scaling, 256 tasks per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SCAML_BENCH_REHEARSAL=1: rehearse the multi-rank control flow on ONE GPU (all ranks on cuda:0, gloo instead of
    # RCCL, which refuses two ranks on one device).  Numbers from such a run mean nothing; it exists so that the
    # barrier / all-reduce / max-over-ranks logic can be exercised on a one-GPU box.
    rehearsal = os.environ.get("SCAML_BENCH_REHEARSAL") == "1"
    if not rehearsal and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible "
                         "(SCAML_BENCH_REHEARSAL=1 rehearses the multi-rank control flow on one GPU; its numbers mean nothing)")
    device = torch.device("cuda", 0 if rehearsal else local_rank)
    # (SCAML_BENCH_FORCE_DIST=1 under torch.distributed.run with ONE rank: the RCCL code path -- communicator set-up, the async
    #  all-reduce, the barriers -- on a one-GPU box; the collective is then a copy, the number is the N = 1 number)
    distributed = world > 1 or (os.environ.get("SCAML_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank builds (a no-op when the parent launcher or an earlier run already did), the others wait
        dist.init_process_group("gloo" if rehearsal else "nccl", **({} if rehearsal else {"device_id": device}))
        if local_rank == 0:
            entry.build()
        torch.cuda.set_device(device)
        dist.barrier()
    entry.build()
    torch.cuda.set_device(device)
    from scamlgp_amd import ops
    from scamlgp_amd.dist import shard_range

    strong = args.total_tasks > 0
    if strong:
        lo, hi = shard_range(args.total_tasks, world, rank)
        T_loc = hi - lo
        if T_loc < 1:
            raise SystemExit(f"--total-tasks {args.total_tasks} leaves rank {rank} of {world} without a task")
        total_tasks = args.total_tasks
    else:
        T_loc, total_tasks = T_PER_GPU, world * T_PER_GPU
    host_inputs, (X, y, th) = make_inputs(rank, device, T_loc)
    kind = ops.KIND_MATERN52
    with_grad = args.step == "fit+grad"
    # allocates the output buffers once; every step rewrites them completely
    out = ops.gp_fit_fused(X, y, th, kind)
    out_g = ops.gp_fit_fused(X, y, th, kind, want_linv=True)   # the evaluation with a gradient also keeps the diagonal-block inverses
    gws = ops.mll_backward_workspace(T_loc, N_POINTS, DIM, device)
    rows = max(args.warmup, args.steps, 1)
    # Every step writes its per-task MLLs into its own row; the region's ONE all-reduce carries the fused buffer
    # [sum_t MLL_t per step || sum_t dMLL_t/dtheta] (one buffer, several evaluations per collective).
    mll_rows = torch.zeros(rows, T_loc, dtype=torch.float64, device=device)
    fused = torch.zeros(rows + DIM + 2, dtype=torch.float64, device=device)
    grad_acc = fused[rows:]
    works = []

    def step(i):
        o = out_g if with_grad else out
        o["mll"] = mll_rows[i]
        ops.gp_fit_fused(X, y, th, kind, out=o, zero_upper=True, want_linv=with_grad)   # the full dense L, zeros included, every step
        if with_grad:
            grad_acc.add_(ops.mll_backward(X, th, kind, o["L"], o["Linv_diag"], o["alpha"], workspace=gws).sum(0))

    def reduce_region(k):
        """The region's exchange step: the per-step MLL sums go next to the gradient sums, the fused buffer goes through
        the one collective."""
        torch.sum(mll_rows[:k], 1, out=fused[:k])
        if distributed:
            works.append(dist.all_reduce(fused, async_op=True))

    def fence():
        for w in works:
            w.wait()
        works.clear()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    def grad_outside_region():
        # --step fit: the hyper-gradient slot of the buffer comes from ONE evaluation outside the timed region
        ops.gp_fit_fused(X, y, th, kind, out=out_g, zero_upper=True, want_linv=True)
        grad_acc.copy_(ops.mll_backward(X, th, kind, out_g["L"], out_g["Linv_diag"], out_g["alpha"], workspace=gws).sum(0))

    # Untimed prologue, the same at every N and for every --steps / --warmup.  (1) Every code path of the run once: the first
    # launch of a kernel, the first collective, the first event cost the host tens of milliseconds of lazy set-up during which
    # the GPU idles -- and an MI355X that has idled for ~20 ms drops its clocks and needs ~10 ms (60+ launches) of load to get
    # them back (tools/dev_launch_ramp.py: 117 instead of 107 us per launch).  (2) --prewarm launches of the step's kernel.
    # From here to the timed region the GPU never idles for more than a synchronisation.
    step(0)
    grad_outside_region()
    reduce_region(1)
    fence()
    torch.cuda.Event(enable_timing=True).record()
    for _ in range(args.prewarm):
        ops.gp_fit_fused(X, y, th, kind, out=out, zero_upper=True)
    for i in range(args.warmup):
        step(i)
    reduce_region(args.warmup)
    fence()
    grad_acc.zero_()
    if not with_grad:
        grad_outside_region()
    fence()
    # the dominant kernel's launches are timed with HIP events on the stream they are launched on (torch's current
    # stream: ops passes torch.cuda.current_stream() across the C ABI)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    reduce_region(args.steps)   # inside the timed region: one small sum + the one all-reduce
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # launch stream only: avg duration of one step's launches

    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ok = not bool(out["info"].any().item())
    reduced = fused.cpu()

    if rank == 0:
        flops = algorithmic_flops_per_task(N_POINTS, DIM, True) * T_loc
        if with_grad:
            flops += algorithmic_backward_flops_per_task(N_POINTS, DIM, True) * T_loc
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        res = {
            "metric": "task-posteriors/sec (K+chol+solve+MLL) at T=256,N=256; 1/2/4/8 GPU",
            "value": total_tasks * args.steps / elapsed,
            "unit": "task-posteriors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_launches": args.prewarm,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"configs[3]: {total_tasks} meta-tasks x 256 points x d=8, Matern-5/2 + ARD, task-sharded over {world} GPU(s) "
                             f"({T_loc} tasks on rank 0)" if strong else
                             "configs[2]: 256 meta-tasks x 256 points x d=8, Matern-5/2 + ARD, per GPU")
                            + " (fused K + jittered Cholesky + alpha + MLL, L stored)"
                            + (" + analytic MLL hyper-gradient every step" if with_grad else ""),
                "step": args.step,
                "tasks_per_gpu": T_loc, "total_tasks": total_tasks, "points": N_POINTS, "dim": DIM, "kernel": "matern52",
                "sharding": ("task shards, " if distributed else "single GPU, ")
                            + "timed region = the steps + one fused all-reduce [sum MLL per step || sum dMLL/dtheta]"
                            + (" (REHEARSAL: all ranks on one GPU over gloo, numbers meaningless)" if rehearsal and distributed else ""),
                "all_tasks_psd": ok,
                "reduced_mll_sum_last_step": float(reduced[args.steps - 1]),
                "reduced_grad_norm": float(reduced[rows:].norm()),
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_TFLOPS,
                "traffic": recorded_pmc_traffic() if (not with_grad and T_loc == T_PER_GPU) else None,
                "kernel": "gp_fit_fused_kernel<16,7,matern52>" + (" + gp_mll_grad_fused_kernel" if with_grad else ""),
                "kernel_ms": kernel_ms,
                "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": algorithmic_bytes_per_task(N_POINTS, DIM) * T_loc,
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
