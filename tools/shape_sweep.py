#!/usr/bin/env python3
"""Quick throughput check over the BASELINE.json shapes (diagnostic; bench.py is the contract)."""
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
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # dmf_context_set_generic level (0 fused ... 3 unfused MFMA pair)
ctx.set_generic(level)
cases = [("config2", 100_000, 64, 6, 2), ("config5 n_u=4", 500_000, 128, 0, 4), ("config5 n_u=8", 500_000, 128, 0, 8),
         ("config5 n_u=5", 500_000, 128, 0, 5), ("config5 n_u=6", 500_000, 128, 0, 6),
         ("config5 n_u=9", 500_000, 128, 0, 9), ("config5 n_u=12", 500_000, 128, 0, 12), ("12+6", 500_000, 128, 12, 6),
         ("headline", 1_000_000, 256, 12, 4), ("S=100 ragged", 200_000, 100, 5, 3), ("S=512", 500_000, 512, 12, 4)]
for name, N, S, n_c, n_u in cases:
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
    bytes_alg = N * S * 16 + N * 8 * (n_c + 3 * n_u)
    print(f"{name:16s} N={N} S={S} {n_c}+{n_u}: {dt*1e3:8.3f} ms/iter  {1/dt:9.1f} it/s  {bytes_alg/dt/1e9:7.0f} GB/s algorithmic   [ms: {fam}]")
    s.close(); p.close(); del V, D, Rt
