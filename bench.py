#!/usr/bin/env python3
"""Benchmark of the solver hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is ONE OUTER ITERATION of the partial-reference solver (20 u inner updates + 20 alpha
inner updates + 1 cost evaluation, demethify/deconvolution.py:206-221) on the headline synthetic
workload of BASELINE.json: 1e6 CpG x 256 samples, 12 known + 4 unknown cell types, inputs already
resident in HBM when the timed region starts.  With N > 1 (launched by torch.distributed.run, one
process per GPU) every rank runs one random restart of the same problem (restart k = seed 1 + k,
data replicated) and the ranks exchange one RCCL all-reduce(min) to pick the best restart: weak
scaling, value = outer iterations of all ranks / wall time.

Prints ONE JSON line on rank 0 (fields: see the driver contract; plus "roofline" and
"cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
WORKLOADS = {
    # name: (N, S, n_c, n_u)
    "headline_1e6x256_12+4": (1_000_000, 256, 12, 4),
    "config2_1e5x64_6+2": (100_000, 64, 6, 2),
}


def algorithmic_bytes(N, S, n_c, n_u):
    """compulsory HBM bytes of one outer iteration: read V and D once,
    read R_trunc, read u and u_ and write u."""
    return N * S * 16 + N * 8 * (n_c + 3 * n_u)


def make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0):
    """Synthetic CpG x sample data, SURVEY.md section 8(d) recipe (Beta(.5,.5) profiles, Dirichlet
    proportions, Poisson(50)+1 depth, Binomial counts); the small factors come from numpy, the two
    N x S draws are made directly in HBM."""
    rs = np.random.RandomState(seed)
    K = n_c + n_u
    Rfull = rs.beta(0.5, 0.5, size=(N, K))
    A = rs.dirichlet(np.ones(K), S).T
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    R_d = torch.from_numpy(Rfull).to(dev)
    P = torch.clamp(R_d @ torch.from_numpy(A).to(dev), 0.0, 1.0)
    D = torch.poisson(torch.full((N, S), 50.0, dtype=torch.float64, device=dev), generator=g) + 1.0
    X = torch.binomial(D, P, generator=g)
    V = (X / D).contiguous()
    Rt = R_d[:, :n_c].contiguous()
    del P, X, R_d
    return V, D.contiguous(), Rt


def cpu_baseline(V_host, D_host, Rt_host, n_u, N_full, budget_iters=2):
    """The oracle (numpy restatement of the reference schedule) timed on this box's host cores on a
    bounded row sample; per-iteration time is linear in N, so the rate is scaled by sample/N."""
    from oracle import solver as osol

    try:
        from threadpoolctl import threadpool_info

        pools = threadpool_info()
        blas = [f"{i.get('internal_api')}:{i.get('num_threads')}" for i in pools]
        blas_threads = max([int(i.get("num_threads") or 1) for i in pools if i.get("user_api") == "blas"] or [1])
    except Exception:  # pragma: no cover
        blas, blas_threads = [], 1
    u0, R, a0 = osol.init_partial("uniform_", V_host, D_host, Rt_host, n_u, seed=1)
    t0 = time.perf_counter()
    osol.solve_partial(u0, R, a0, V_host, D_host, Rt_host, n_u, budget_iters, 20, 0.0,
                       project=osol.simplex_project_columns_fast)
    dt = time.perf_counter() - t0
    n_s = V_host.shape[0]
    rate_sample = budget_iters / dt
    return {
        "value": rate_sample * n_s / N_full,
        "unit": "outer iters/s",
        # threads actually used: numpy's elementwise temporaries are single-threaded, only the skinny dgemms
        # fan out over the BLAS pool (default threading, as the reference would run)
        "cores": blas_threads,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"{budget_iters} outer iterations (T2=20) of oracle/solver.py on the first {n_s} of {N_full} "
                  f"CpG rows x {V_host.shape[1]} samples, {dt:.1f} s wall; rate scaled by {n_s}/{N_full}",
        "blas": blas,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline_1e6x256_12+4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=40_000)
    ap.add_argument("--kernels", type=int, default=0, choices=[0, 1, 2, 3],
                    help="kernel selection level (dmf_context_set_generic): 0 = fused row pass (default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --share-gpu rehearses the N > 1 path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    from demethify_amd import _lib as L
    from demethify_amd.device import Context, Problem, Solver
    from demethify_amd.shard import pick_min_cost

    N, S, n_c, n_u = WORKLOADS[args.workload]
    K = n_c + n_u
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0)
    torch.cuda.synchronize()

    ctx = Context(local_rank)
    ctx.set_generic(args.kernels)
    problem = Problem(ctx, V, D, Rt)
    # restart k uses seed 1 + k; init drawn on the host in the reference's order
    rs = np.random.RandomState(1 + rank)
    u0 = rs.uniform(size=(N, n_u))
    a0 = rs.dirichlet(np.ones(K), S).T
    solver = Solver(problem, u0, a0, L.DMF_MODE_PARTIAL)

    solver.step(args.warmup, 20, 0.0)
    ctx.synchronize()
    # HIP events around the two families that stream V / D (the roofline kernel is one of them); the
    # KB-sized alpha phase is timed after the timed region so that its event records do not sit in it
    ctx.set_profiling(True, families=(L.KERNEL_ROWPASS, L.KERNEL_GRAM))
    ctx.reset_kernel_time()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters, _ = solver.step(args.steps, 20, 0.0)
    ctx.synchronize()
    if world > 1:
        cost, _ = solver.get_cost()
        best_rank, best_cost = pick_min_cost(cost, rank, world, dev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert iters == args.warmup + args.steps, (iters, args.warmup, args.steps)

    fam = {name: ctx.kernel_time(i) for i, name in enumerate(L.KERNEL_FAMILIES)}
    ctx.reset_kernel_time()
    ctx.set_profiling(True, families=(L.KERNEL_ALPHA, L.KERNEL_COST))
    solver.step(2, 20, 0.0)  # untimed: fills in the small families of the per-family table
    ctx.synchronize()
    for i in (L.KERNEL_ALPHA, L.KERNEL_COST):
        fam[L.KERNEL_FAMILIES[i]] = ctx.kernel_time(i)
    ctx.set_profiling(False)

    if rank == 0:
        b_alg = algorithmic_bytes(N, S, n_c, n_u)
        fam_ms = {k: (v[0] / max(v[1], 1), v[1]) for k, v in fam.items()}
        # dominant kernel family of the outer iteration and its algorithmic traffic per launch
        # With the fused row pass (--kernels 0) ONE launch of the "rowpass" family does the whole V / D
        # stream of an outer iteration: B_alg of SURVEY.md 8(d).  The unfused pair (--kernels 3) reads V
        # and D once per kernel.
        per_launch_bytes = {
            "rowpass": N * S * 16 + N * 8 * (n_c + 3 * n_u),   # V, D, R_trunc, u, u_ in; u out
            "gram": N * S * 16 + N * 8 * (n_c + n_u),          # V, D, R_trunc, u in (unfused levels only)
        }
        dom = max(("rowpass", "gram"), key=lambda k: fam[k][0])
        kernel_names = {0: {"rowpass": "k_rowpass_fused", "gram": "k_gram_reduce"},
                        3: {"rowpass": "k_u_phase_mfma", "gram": "k_gram_u"},
                        1: {"rowpass": "k_u_phase_gram", "gram": "k_gram"},
                        2: {"rowpass": "k_u_step_direct", "gram": "k_gram"}}[args.kernels]
        dom_ms = fam_ms[dom][0]
        achieved = per_launch_bytes[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get(args.workload, {}).get(kernel_names[dom], {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "NMF update iters/sec (1e6 CpG x 256 samples x 16 types)" if args.workload.startswith("headline")
                      else f"NMF update iters/sec ({args.workload})",
            "value": world * args.steps / elapsed,
            "unit": "outer iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "N_cpg": N, "S_samples": S, "n_known": n_c, "n_unknown": n_u,
                       "inner_iters": 20, "unit_of_work": "one outer iteration = 20 u + 20 alpha inner updates + cost",
                       "parallelism": f"restart-sharded x{world}" if world > 1 else "single solve"},
            "roofline": {
                "bound": "hbm", "kernel": kernel_names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": per_launch_bytes[dom],
                "avg_launch_ms": dom_ms,
                "whole_iteration": {"algorithmic_bytes": b_alg,
                                    "achieved": b_alg / (ms_per_step * 1e-3) / 1e9,
                                    "frac": b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "family_avg_ms": {k: round(v[0], 4) for k, v in fam_ms.items()},
                "family_launches": {k: v[1] for k, v in fam_ms.items()},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            n_s = min(args.cpu_rows, N)
            out["cpu_baseline"] = cpu_baseline(V[:n_s].cpu().numpy(), D[:n_s].cpu().numpy().astype(np.int64),
                                               Rt[:n_s].cpu().numpy(), n_u, N)
        else:
            out["cpu_baseline"] = None
        if world > 1:
            out["config"]["best_restart_rank"] = int(best_rank)
            out["config"]["best_restart_cost"] = float(best_cost)
        print(json.dumps(out), flush=True)

    solver.close()
    problem.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
