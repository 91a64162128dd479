#!/usr/bin/env python3
"""Diagnostic: odd sample counts on the second-generation kernels (level 0) against the first-generation FP64 kernels
(level 3) that carried them before.   python tools/odd_s_bench.py"""
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
for N, S, n_c, n_u in [(1_000_000, 255, 12, 4), (1_000_000, 256, 12, 4), (500_000, 127, 0, 8), (500_000, 128, 0, 8), (100_000, 63, 6, 2), (500_000, 512, 12, 4), (500_000, 384, 12, 4)]:
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, max(n_c, 1), n_u, seed=0)
    rs = np.random.RandomState(1)
    u0 = rs.uniform(size=(N, n_u)); a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    for level in (0, 3):
        ctx.set_generic(level)
        p = Problem(ctx, V, D, Rt if n_c else None)
        s = Solver(p, u0, a0, L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED)
        s.step(2, 20, 0.0); ctx.synchronize()
        t0 = time.perf_counter(); s.step(10, 20, 0.0); ctx.synchronize(); dt = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter(); s.direct_cost(); tc = time.perf_counter() - t0
        print(f"N={N} S={S} {n_c}+{n_u} level {level}: {dt*1e3:7.3f} ms/iter, cost {tc*1e3:6.3f} ms   {s.describe(20)}", flush=True)
        s.close(); p.close()
    ctx.set_generic(0)
    del V, D, Rt
