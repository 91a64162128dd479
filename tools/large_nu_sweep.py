#!/usr/bin/env python3
"""Diagnostic: outer-iteration time for large numbers of unknown types (the --ic sweep goes to n_u = 25)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import make_inputs_on_device
from demethify_amd import _lib as L
from demethify_amd.device import Context, Problem, Solver

dev = torch.device("cuda", 0)
ctx = Context(0)
N, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500_000, 128)
for n_c, n_u in [(0, 8), (0, 9), (0, 12), (0, 13), (0, 14), (0, 16), (0, 17), (0, 20), (0, 25), (12, 9), (12, 16)]:
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, max(n_c, 1), n_u, seed=0)
    p = Problem(ctx, V, D, Rt if n_c else None)
    rs = np.random.RandomState(1)
    u0 = rs.uniform(size=(N, n_u)); a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    s = Solver(p, u0, a0, L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED)
    s.step(1, 20, 0.0); ctx.synchronize()
    ctx.set_profiling(True); ctx.reset_kernel_time()
    n = 3
    t0 = time.perf_counter(); s.step(n, 20, 0.0); ctx.synchronize(); dt = (time.perf_counter() - t0) / n
    fam = "  ".join(f"{nm} {ctx.kernel_time(i)[0] / n:.3f}" for i, nm in enumerate(L.KERNEL_FAMILIES))
    ctx.set_profiling(False)
    print(f"N={N} S={S} {n_c}+{n_u}: {dt*1e3:9.3f} ms/iter   [ms: {fam}]", flush=True)
    s.close(); p.close(); del V, D, Rt
