#!/usr/bin/env python3
"""Diagnostic: fixed per-problem costs at the headline size (a bootstrap replicate pays them once each)."""
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
N, S, n_c, n_u = 1_000_000, 256, 12, 4
V, D, Rt = make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); p = Problem(ctx, V, D, Rt); ctx.synchronize(); t1 = time.perf_counter()
    idx = np.random.RandomState(rep).randint(0, N, N)
    g = p.gather(idx); ctx.synchronize(); t2 = time.perf_counter()
    rs = np.random.RandomState(1)
    u0 = rs.uniform(size=(N, n_u)); a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    t3 = time.perf_counter(); s = Solver(g, u0, a0, L.DMF_MODE_PARTIAL); ctx.synchronize(); t4 = time.perf_counter()
    print(f"problem from device tensors {1e3*(t1-t0):.1f} ms, row gather + finalize {1e3*(t2-t1):.1f} ms, "
          f"solver create (u upload + initial cost) {1e3*(t4-t3):.1f} ms")
    s.close(); g.close(); p.close()
