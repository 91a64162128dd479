#!/usr/bin/env python3
"""Diagnostic: outer-iteration time of the purity-constrained solver (CLI default with --purity: 100 x 500)."""
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
ctx.set_generic(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for N, S, n_c, n_u in [(100_000, 64, 6, 2), (1_000_000, 256, 12, 4)]:
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0)
    p = Problem(ctx, V, D, Rt)
    rs = np.random.RandomState(1)
    s = Solver(p, rs.uniform(size=(N, n_u)), rs.dirichlet(np.ones(n_c + n_u), S).T, L.DMF_MODE_PARTIAL)
    s.set_purity(rs.uniform(0.2, 0.9, size=S))
    s.step(1, 500, 0.0); ctx.synchronize()
    ctx.set_profiling(True); ctx.reset_kernel_time()
    n = 3
    t0 = time.perf_counter(); s.step(n, 500, 0.0); ctx.synchronize(); dt = (time.perf_counter() - t0) / n
    fam = "  ".join(f"{nm} {ctx.kernel_time(i)[0] / n:.3f}" for i, nm in enumerate(L.KERNEL_FAMILIES))
    ctx.set_profiling(False)
    print(f"N={N} S={S} {n_c}+{n_u} purity, n_iter2 = 500: {dt*1e3:.3f} ms/iter   [ms: {fam}]", flush=True)
    s.close(); p.close(); del V, D, Rt
