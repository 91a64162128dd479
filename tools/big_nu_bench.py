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
for N, S, n_c, n_u in [(500_000, 128, 0, 16), (500_000, 128, 0, 17), (500_000, 128, 0, 20), (500_000, 128, 0, 25), (500_000, 128, 6, 20)]:
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, max(n_c, 1), n_u, seed=0)
    rs = np.random.RandomState(1)
    u0 = rs.uniform(size=(N, n_u)); a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    p = Problem(ctx, V, D, Rt if n_c else None)
    s = Solver(p, u0, a0, L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED)
    s.step(2, 20, 0.0); ctx.synchronize()
    ctx.set_profiling(True); ctx.reset_kernel_time()
    t0 = time.perf_counter(); s.step(5, 20, 0.0); ctx.synchronize(); dt = (time.perf_counter() - t0) / 5
    fam = "  ".join(f"{n} {ctx.kernel_time(i)[0] / 5:.3f}" for i, n in enumerate(L.KERNEL_FAMILIES))
    ctx.set_profiling(False)
    print(f"N={N} S={S} {n_c}+{n_u}: {dt*1e3:7.3f} ms/iter [{fam}]  {s.describe(20)}", flush=True)
    s.close(); p.close(); del V, D, Rt
