#!/usr/bin/env python3
"""Diagnostic: ms per outer iteration for wide row groups (n_u 5..16) at 5e5 x 128 and 2.5e5 x 256.
   python tools/wide_nu_sweep.py            (DMF_CM_I8_MIN_NU=99 shows the kernels that ran before k_cm_i8 -- in an experiment
   build only: DMF_EXPERIMENT=1 python -m demethify_amd._build --force; the product build has no environment knobs)"""
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
cases = [(500_000, 128, 0, n) for n in (5, 6, 7, 8, 9, 10, 12, 14, 16)] + [(500_000, 128, 12, n) for n in (5, 6, 8, 12)] + [(500_000, 128, 4, 12), (500_000, 128, 8, 8), (500_000, 128, 16, 16)] + \
        [(250_000, 256, 0, 8), (250_000, 256, 12, 6), (250_000, 64, 4, 8)]
if len(sys.argv) > 1:
    cases = [(500_000, 128, 0, n) for n in (5, 6, 7)]
for N, S, n_c, n_u in cases:
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, max(n_c, 1), n_u, seed=0)
    p = Problem(ctx, V, D, Rt if n_c else None)
    rs = np.random.RandomState(1)
    u0 = rs.uniform(size=(N, n_u)); a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    s = Solver(p, u0, a0, L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED)
    s.step(2, 20, 0.0); ctx.synchronize()
    ctx.set_profiling(True); ctx.reset_kernel_time()
    t0 = time.perf_counter(); s.step(10, 20, 0.0); ctx.synchronize(); dt = (time.perf_counter() - t0) / 10
    fam = "  ".join(f"{n} {ctx.kernel_time(i)[0] / 10:.3f}" for i, n in enumerate(L.KERNEL_FAMILIES))
    ctx.set_profiling(False)
    print(f"N={N} S={S} {n_c:2d}+{n_u:2d}: {dt*1e3:7.3f} ms/iter   [{fam}]   {s.describe(20)}", flush=True)
    s.close(); p.close(); del V, D, Rt
