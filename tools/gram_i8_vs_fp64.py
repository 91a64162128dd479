"""Diagnostic: Gram family time per outer iteration with the integer GEMM + b_u kernel (level 0) against the FP64 Gram
kernels (level 3) for shapes whose u phase is a kernel of its own (n_u > 4)."""
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
N, S = 500_000, 128
shapes = [tuple(int(x) for x in a.split('+')) for a in sys.argv[1:]] or [(0, 5), (0, 8), (0, 12), (6, 6)]
for n_c, n_u in shapes:
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, max(n_c, 1), n_u, seed=0)
    rs = np.random.RandomState(1)
    u0 = rs.uniform(size=(N, n_u)); a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    p = Problem(ctx, V, D, Rt if n_c else None)
    for level in (0, 3):
        ctx.set_generic(level)
        s = Solver(p, u0, a0, L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED)
        desc = s.describe(20)
        s.step(1, 20, 0.0); ctx.synchronize()
        ctx.set_profiling(True); ctx.reset_kernel_time()
        n = 3
        t0 = time.perf_counter(); s.step(n, 20, 0.0); ctx.synchronize(); dt = (time.perf_counter() - t0) / n
        fam = "  ".join(f"{nm} {ctx.kernel_time(i)[0] / n:.3f}" for i, nm in enumerate(L.KERNEL_FAMILIES))
        ctx.set_profiling(False)
        print(f"{n_c}+{n_u} level {level}: {dt*1e3:7.3f} ms/iter  [{fam}]  {desc}", flush=True)
        s.close()
    ctx.set_generic(0)
    p.close(); del V, D, Rt
