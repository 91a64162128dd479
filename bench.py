#!/usr/bin/env python3
"""Benchmark of the solver hot path on MI355X: BASELINE.json's restart job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--restarts R]

The job (BASELINE.json configs[2]; reference loop demethify/demethify.py:195-203):
R = 64 random restarts of the partial-reference solver on ONE synthetic problem, 1e6 CpG x 256 samples, 12 known + 4
unknown cell types, V / D / R_trunc already resident in HBM on every rank when the timed region starts.  Restart k
uses seed 1 + k and runs on rank k mod N (one process per GPU, torch.distributed, RCCL); the problem is replicated,
nothing of the N x S data path is exchanged.  A "step" is ONE OUTER ITERATION (20 u inner updates + 20 alpha inner
updates + 1 cost evaluation, demethify/deconvolution.py:206-221) of every restart: K steps = the job with
--iterations K 20 --termination 0.  The timed region holds everything the job does after the data is resident:
the host-side init of every restart (legacy MT19937 stream in the reference's order, deconvolution.py:55-56, drawn
by a worker thread while the GPU solves the previous restart), the upload of u0 / alpha0, the solver set-up (initial
cost, :204), the K outer iterations, the per-restart cost_f_w (demethify.py:199), ONE all-reduce(min) over the
R-element cost vector and the broadcast of the winner's (u, alpha).  Total work is fixed as N grows: strong scaling,
value = R * K outer iterations / wall time (max over ranks).

Prints ONE JSON line on rank 0 (fields: see the driver contract; plus "roofline", "cpu_baseline" and "loop_only" =
the rate of the K-iteration loops alone, which is what round 1 reported as value).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6  # FP64 vector = FP64 matrix peak (256 CUs x 4 SIMDs x 16 FMA/clk x 2.4 GHz)
T2 = int(os.environ.get("DMF_BENCH_T2", "20"))  # inner iterations (CLI default 20, demethify.py:64; env: experiments only)
WORKLOADS = {
    # name: (N, S, n_c, n_u)
    "headline_1e6x256_12+4": (1_000_000, 256, 12, 4),
    "config2_1e5x64_6+2": (100_000, 64, 6, 2),
}


def algorithmic_bytes(N, S, n_c, n_u):
    """compulsory HBM bytes of one outer iteration: read V and D once (f64 each),
    read R_trunc, read u and u_ and write u."""
    return N * S * 16 + N * 8 * (n_c + 3 * n_u)


def algorithmic_flops(N, S, n_c, n_u, t2=T2):
    """flops of one outer iteration in the one-pass (fused) form."""
    return N * S * (2 * n_c + 4 * n_u + 2 * n_u * (n_u + 1) + 2 * n_c * n_u + 4) + t2 * N * (2 * n_u * n_u + 8 * n_u)


def make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0, depth=None):
    """Synthetic CpG x sample data, SURVEY.md section 8(d) recipe (Beta(.5,.5) profiles, Dirichlet proportions,
    Poisson(50)+1 depth, Binomial counts).  The small factors (N x K profiles, K x S proportions) come from
    legacy numpy RandomState(seed) exactly as the recipe says; the two N x S draws (Poisson depth, Binomial
    counts) are made directly in HBM with torch's Philox generator: same distributions, different bits than a
    host-side legacy-numpy draw of 2.56e8 values would give."""
    rs = np.random.RandomState(seed)
    K = n_c + n_u
    Rfull = rs.beta(0.5, 0.5, size=(N, K))
    A = rs.dirichlet(np.ones(K), S).T
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    R_d = torch.from_numpy(Rfull).to(dev)
    P = torch.clamp(R_d @ torch.from_numpy(A).to(dev), 0.0, 1.0)
    if depth is None:  # (experiments: depth > ~100 makes some count exceed 127 -> two count digits)
        depth = float(os.environ.get("DMF_BENCH_DEPTH", "50"))
    D = torch.poisson(torch.full((N, S), depth, dtype=torch.float64, device=dev), generator=g) + 1.0
    X = torch.binomial(D, P, generator=g)
    V = (X / D).contiguous()
    Rt = R_d[:, :n_c].contiguous()
    del P, X, R_d
    return V, D.contiguous(), Rt


def restart_init(k, N, S, n_c, n_u):
    """init_BSSMF_md('uniform_', ..., seed=1 + k) of the reference (deconvolution.py:41,55-56): uniform N x n_u
    first, then Dirichlet; a private RandomState(seed) yields the same stream as the global one seeded alike."""
    rs = np.random.RandomState(1 + k)
    u0 = rs.uniform(size=(N, n_u))
    a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    return u0, np.ascontiguousarray(a0)


def restart_init_on_device(k, shape, ctx):
    """The restart's initialisation drawn on the host and uploaded through page-locked memory on the context's copy stream
    (demethify_amd/staging.py, dmf_stage_upload) -- by a worker thread, while the GPU iterates the restart before it."""
    from demethify_amd import staging

    return staging.to_device(restart_init(k, *shape), ctx)


def cpu_baseline(V_dev, D_dev, Rt_dev, n_u, N_full, rows=(250_000, 1_000_000)):
    """The oracle (numpy restatement of the reference schedule, oracle/solver.py) timed on this box's host cores:
    ONE full outer iteration (T2 = 20) of the same synthetic problem at a quarter of the rows and at ALL rows.
    `value` is the full-size measurement (no extrapolation when rows[-1] == N_full); the quarter-size run is
    reported next to it because the time is NOT linear in N on a many-core host (measured on the 256-thread box:
    62 500 rows 7.3 s, 250 000 rows 13.8 s): a sample-and-scale figure would understate the CPU path."""
    from oracle import solver as osol

    try:
        from threadpoolctl import threadpool_info

        pools = threadpool_info()
        blas = [f"{i.get('internal_api')}:{i.get('num_threads')}" for i in pools]
        blas_threads = max([int(i.get("num_threads") or 1) for i in pools if i.get("user_api") == "blas"] or [1])
    except Exception:  # pragma: no cover
        blas, blas_threads = [], 1
    secs = []
    for n_s in rows:
        n_s = min(n_s, N_full)
        V = V_dev[:n_s].cpu().numpy()
        D = D_dev[:n_s].cpu().numpy().astype(np.int64)
        Rt = Rt_dev[:n_s].cpu().numpy()
        u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, n_u, seed=1)
        t0 = time.perf_counter()
        osol.solve_partial(u0, R, a0, V, D, Rt, n_u, 1, T2, 0.0, project=osol.simplex_project_columns_fast)
        secs.append(time.perf_counter() - t0)
        del V, D, Rt, u0, R, a0
    n_big = min(rows[-1], N_full)
    per_row = [s / min(r, N_full) for s, r in zip(secs, rows)]
    return {
        "value": (1.0 / secs[-1]) * n_big / N_full,
        "unit": "outer iters/s",
        # threads actually used: numpy's elementwise temporaries are single-threaded, only the skinny dgemms
        # fan out over the BLAS pool (default threading, as the reference would run)
        "cores": blas_threads,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"1 outer iteration (T2={T2}, incl. the two cost evaluations) of oracle/solver.py on the first {n_big} of "
                  f"{N_full} CpG rows x {V_dev.shape[1]} samples: {secs[-1]:.1f} s wall; rate scaled by {n_big}/{N_full}",
        "linearity": {"rows": [min(r, N_full) for r in rows], "seconds_per_outer_iteration": [round(s, 2) for s in secs],
                      "seconds_per_row_ratio_large_over_small": round(per_row[-1] / per_row[0], 3)},
        "blas": blas,
    }


def git_head():
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True,
                              timeout=5).stdout.strip() or None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--restarts", type=int, default=64, help="restarts of the whole job (BASELINE config 3: 64)")
    ap.add_argument("--workload", default="headline_1e6x256_12+4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernels", type=int, default=0, choices=[0, 1, 2, 3],
                    help="kernel selection level (dmf_context_set_generic): 0 = fastest (default)")
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
    from demethify_amd import shard, staging
    from demethify_amd.device import Context, Problem, Solver

    N, S, n_c, n_u = WORKLOADS[args.workload]
    K = n_c + n_u
    R = args.restarts
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0)
    torch.cuda.synchronize()

    ctx = Context(local_rank)
    ctx.set_generic(args.kernels)
    problem = Problem(ctx, V, D, Rt)

    # ---- warm-up: W untimed steps of one restart per rank (+ the collectives once, so that RCCL's lazy
    # communicator set-up is not in the timed region)
    # (the warm-up restart goes through the staging path too: page-locked buffers and the copy stream exist afterwards)
    staging.reserve(((N, n_u), (K, S)), count=2)
    u0, a0 = staging.to_device(restart_init(rank, N, S, n_c, n_u), ctx)
    with Solver(problem, u0, a0, L.DMF_MODE_PARTIAL) as s:
        kernels = s.describe(T2)
        s.step(args.warmup, T2, 0.0)
        s.direct_cost()
        if world > 1:
            # the job's two exchanges at their real sizes: RCCL connects its rings on the first large message
            shard.allreduce_min_vector({rank: 0.0}, world)
            shard.broadcast_winner(s if rank == 0 else None, (N, n_u), (K, S), 0)
    ctx.synchronize()
    # HIP events, on the stream the kernels are launched on, around EVERY launch of the roofline kernel's family (the row
    # pass) and, during ONE restart in the middle of the timed region, around the other two families that stream V / D (the
    # region's first restarts run while the clocks still ramp: their Gram launches take 0.17 ms against 0.145 later): an event
    # record between two kernels leaves the GPU idle for a few microseconds, which long kernels hide (the command processor
    # works ahead) and short ones do not -- four records per outer iteration were 13 % of config 2's 85 us iteration.  The
    # KB-sized alpha phase is timed after the job.
    ctx.set_profiling(True, families=(L.KERNEL_ROWPASS,))
    ctx.reset_kernel_time()

    mine = shard.my_items(R, rank, world)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # ------------------------------------------------------------------ timed region: the restart job
    feeder = staging.Prefetcher(mine, lambda k: restart_init_on_device(k, (N, S, n_c, n_u), ctx), depth=2, workers=2)
    best, local_costs, loop_s, iters_total = None, {}, 0.0, 0

    def settle(k, s):
        """cost_f_w of restart k (demethify.py:199) has been on its way since the restart's loop ended: take it, keep the
        solver if it is the new minimum (strict '<': the first minimum wins, demethify.py:200)."""
        nonlocal best
        cost = s.cost_end()
        local_costs[k] = cost
        if best is None or cost < best[0]:
            if best is not None:
                best[2].close()
            best = (cost, k, s)
        else:
            s.close()

    waiting = None
    sampled = mine[len(mine) // 2] if len(mine) else None  # the restart whose Gram and cost launches are timed as well
    for k, (u0, a0) in feeder:
        s = Solver(problem, u0, a0, L.DMF_MODE_PARTIAL)  # (set up while the GPU takes the previous restart's cost)
        if waiting is not None:
            settle(*waiting)
        if k == sampled:
            ctx.set_profiling(True, families=(L.KERNEL_ROWPASS, L.KERNEL_GRAM, L.KERNEL_COST))
        tl = time.perf_counter()
        it, _ = s.step(args.steps, T2, 0.0)  # returns after the last iteration's state has been read back
        loop_s += time.perf_counter() - tl
        iters_total += it
        s.cost_begin()
        if k == sampled:
            ctx.set_profiling(True, families=(L.KERNEL_ROWPASS,))
        waiting = (k, s)
    if waiting is not None:
        settle(*waiting)
    costs = shard.allreduce_min_vector(local_costs, R)  # ONE all-reduce(min), 8 R bytes
    best_k = shard.argmin_first(costs)
    owner = best_k % world
    # winner broadcast, device to device: the owner's iterate leaves its solver as a CUDA tensor, every rank receives it
    # in HBM, and only rank 0 -- the rank that writes the output files (demethify.py) -- lands it on the host
    u_w, a_w = shard.broadcast_winner(best[2] if rank == owner else None, (N, n_u), (K, S), owner)
    if best is not None:
        best[2].close()
    ctx.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # ------------------------------------------------------------------ end of the timed region
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert iters_total == len(mine) * args.steps, (iters_total, len(mine), args.steps)
    assert np.isfinite(costs).all() and (rank != 0 or u_w.shape == (N, n_u))

    fam = {name: ctx.kernel_time(i) for i, name in enumerate(L.KERNEL_FAMILIES)}
    ctx.reset_kernel_time()
    ctx.set_profiling(True, families=(L.KERNEL_ALPHA,))
    u0, a0 = restart_init(rank, N, S, n_c, n_u)
    with Solver(problem, u0, a0, L.DMF_MODE_PARTIAL) as s:
        s.step(args.steps, T2, 0.0)  # untimed: fills in the small family of the per-family table
    ctx.synchronize()
    fam[L.KERNEL_FAMILIES[L.KERNEL_ALPHA]] = ctx.kernel_time(L.KERNEL_ALPHA)
    ctx.set_profiling(False)

    if rank == 0:
        b_alg = algorithmic_bytes(N, S, n_c, n_u)
        fam_ms = {k: (v[0] / max(v[1], 1), v[1]) for k, v in fam.items()}
        names = dict(tok.split("=", 1) for tok in kernels.split() if "=" in tok and tok.split("=")[0] in ("rowpass", "gram", "alpha"))
        n_p = n_u * (n_u + 1) // 2
        nd = 2 if "nd=2" in names.get("gram", "") else 1
        # Algorithmic (compulsory) HBM bytes per launch of each streaming kernel, 1 unit = 1 outer iteration per launch.
        # SURVEY.md 8(d) counts V and the counts at 8 bytes each (B_alg = 16 N S + 8 N (n_c + 3 n_u)); the second-
        # generation kernels read the counts as u16 (row pass) and as 8-bit digit planes (integer Gram), an exact
        # re-encoding, so THEIR compulsory bytes are what `achieved` is computed from -- pricing them at 16 bytes per
        # element would report bytes that are never moved.
        if names.get("rowpass", "").startswith("k_rowpass_v2"):
            row_bytes = N * S * (8 + 2) + N * 8 * (n_c + 4 * n_u)       # V f64, counts u16, R_trunc, u + u_ in and out
            gram_bytes = N * S * nd + N * 8 * (n_c + n_u)               # count digit planes, R_trunc, u
            row_flop = N * S * (2 * n_c + 4 * n_u + 2 * n_p + 2) + T2 * N * (2 * n_u * n_u + 8 * n_u)
        else:
            row_bytes = N * S * 16 + N * 8 * (n_c + 4 * n_u)
            gram_bytes = 0 if names.get("gram") == "fused" else N * S * 16 + N * 8 * (n_c + n_u)
            row_flop = algorithmic_flops(N, S, n_c, n_u) if names.get("gram") == "fused" else \
                N * S * (2 * n_c + 4 * n_u + 2 * n_p + 2) + T2 * N * (2 * n_u * n_u + 8 * n_u)
        per_launch_bytes = {"rowpass": row_bytes, "gram": gram_bytes}
        dom = max(("rowpass", "gram"), key=lambda k: fam[k][0])
        dom_kernel = names.get(dom, dom)
        if dom == "gram" and dom_kernel == "fused":
            dom_kernel = "k_gram_reduce"
        dom_ms = fam_ms[dom][0]
        achieved = per_launch_bytes[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # Counter figures of the dominant kernel: NOT measured in this run (counters need rocprofv3 passes of their own,
        # one group per pass) but taken from profiles/counters.json, which tools/collect_counters.py writes from such
        # passes and stamps with the commit they were taken at (profiles/README.md).
        traffic, traffic_source, pmc = None, None, {}
        for name in ("counters.json", "traffic.json"):
            tfile = ROOT / "profiles" / name
            if traffic is not None or not tfile.exists():
                continue
            try:
                tj = json.loads(tfile.read_text())
                entry = tj.get(args.workload, {}).get(dom_kernel.split("<")[0], {})
                traffic = entry.get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = f"profiles/{name}@{tj.get('_commit', 'unknown')} (rocprofv3 --pmc, separate run)"
                    pmc = {k: entry.get(k) for k in ("issue_frac", "wait_frac", "mfma_busy_frac") if entry.get(k) is not None}
            except Exception:
                traffic = None
        # What bounds the dominant kernel, from the data: the roof it sits closest to if it reaches 0.7 of it -- HBM bytes,
        # FP64 rate (this run), issue slots or matrix-pipe time (counters) -- otherwise none of them does: "latency"
        # (dependency and memory stalls that the resident waves do not cover; see wait_frac).
        hbm_frac = achieved / HBM_PEAK_GBS
        fp64_frac = row_flop / (fam_ms["rowpass"][0] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if fam_ms["rowpass"][0] > 0 and dom == "rowpass" else 0.0
        shares = {"hbm": hbm_frac, "fp64": fp64_frac, "issue": pmc.get("issue_frac", 0.0), "mfma": pmc.get("mfma_busy_frac", 0.0)}
        top = max(shares, key=shares.get)
        bound = top if shares[top] >= 0.7 else "latency"
        ms_per_step = elapsed / args.steps * 1e3
        loop_rate = iters_total / loop_s if loop_s > 0 else 0.0
        iter_bytes = row_bytes + gram_bytes
        gram_ms = fam_ms["gram"][0]
        out = {
            "metric": "NMF update iters/sec (1e6 CpG x 256 samples x 16 types)" if args.workload.startswith("headline")
                      else f"NMF update iters/sec ({args.workload})",
            "value": R * args.steps / elapsed,
            "unit": "outer iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (SURVEY 8d recipe; small factors legacy numpy, N x S draws torch Philox on device)",
            "config": {"workload": args.workload, "N_cpg": N, "S_samples": S, "n_known": n_c, "n_unknown": n_u,
                       "inner_iters": T2, "restarts": R, "restart_seed": "1 + k", "restart_placement": "k mod n_gpus",
                       "unit_of_work": "one step = one outer iteration (20 u + 20 alpha inner updates + cost) of each "
                                       "of the R restarts; timed: host init + upload (a worker thread, one restart ahead) + set-up + K iterations + "
                                       "cost_f_w per restart, one all-reduce(min), winner broadcast (device to device; rank 0 lands it on the host)",
                       "parallelism": f"restart-sharded x{world}", "kernels": kernels,
                       "best_restart": int(best_k), "best_restart_cost": float(costs[best_k]), "head": git_head()},
            # rank 0's K-iteration loops alone (what a single solve sustains; round 1 reported this as value)
            "loop_only": {"value": loop_rate, "unit": "outer iters/s per GPU", "ms_per_outer_iteration": 1e3 / loop_rate if loop_rate else None,
                          "restarts_on_rank0": len(mine), "job_seconds": elapsed, "loop_seconds_rank0": loop_s},
            "roofline": {
                # achieved / peak / frac: against the HBM roof (the bytes the kernel must move), whatever `bound` says
                "bound": bound, "bound_shares": {k: round(v, 4) for k, v in shares.items()},
                "kernel": dom_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "issue_frac": pmc.get("issue_frac"), "wait_frac": pmc.get("wait_frac"), "mfma_busy_frac": pmc.get("mfma_busy_frac"),
                "algorithmic_bytes_per_launch": per_launch_bytes[dom],
                "avg_launch_ms": dom_ms,
                # the same launch against the FP64 roof: flops the row pass executes in this formulation (FP64 MFMA and
                # FP64 VALU share one pipe on gfx950)
                "fp64": {"flop_per_launch": row_flop, "achieved": row_flop / (fam_ms["rowpass"][0] * 1e-3) / 1e12 if fam_ms["rowpass"][0] > 0 else 0.0,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": row_flop / (fam_ms["rowpass"][0] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if fam_ms["rowpass"][0] > 0 else 0.0},
                # second kernel of the outer iteration (family "gram": the integer-matrix-core GEMM + its reduction)
                "gram": {"kernel": names.get("gram"), "avg_ms": gram_ms, "algorithmic_bytes": gram_bytes,
                         "achieved": gram_bytes / (gram_ms * 1e-3) / 1e9 if gram_ms > 0 else 0.0, "unit": "GB/s",
                         "int8_ops": 2 * N * S * (n_c * n_u + n_p) * 7 * nd if "i8" in names.get("gram", "") else None},
                "whole_iteration": {"algorithmic_bytes": iter_bytes, "achieved": iter_bytes * loop_rate / 1e9,
                                    "frac": iter_bytes * loop_rate / 1e9 / HBM_PEAK_GBS,
                                    # SURVEY 8(d)'s f64-layout figure (16 B per element) times the same rate: what the
                                    # iteration would have to stream without the integer re-encoding; NOT bytes moved
                                    "sec8d_f64_layout_bytes": b_alg, "sec8d_f64_layout_equivalent": b_alg * loop_rate / 1e9},
                "family_avg_ms": {k: round(v[0], 4) for k, v in fam_ms.items()},
                "family_timed": "HIP events inside the timed region: rowpass every launch, gram and cost one restart's (the middle one); alpha: one restart's launches after it",
                "family_launches": {k: v[1] for k, v in fam_ms.items()},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(V, D, Rt, n_u, N)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    problem.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
