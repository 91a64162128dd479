"""Where does a bootstrap replicate's time go on the GPU side?  (diagnostic; headline shape, synthetic data)
   python tools/replicate_overheads.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from demethify_amd import _lib as L
from demethify_amd.device import Context, Problem, Solver
from demethify_amd.bootstrap import bootstrap_row_indices
from demethify_amd.staging import indices_to_device

N, S, n_c, n_u = bench.WORKLOADS["headline_1e6x256_12+4"]
dev = torch.device("cuda", 0)
V, D, Rt = bench.make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0)
ctx = Context(0)
full = Problem(ctx, V, D, Rt)
stack = torch.empty((4, N * n_u), dtype=torch.float64, device=dev)
def T(f):
    t = time.perf_counter(); r = f(); ctx.synchronize()
    return r, (time.perf_counter() - t) * 1e3
for rep in range(4):
    idx, t_idx = T(lambda: bootstrap_row_indices(rep + 1, N))
    (u0, a0), t_init = T(lambda: bench.restart_init(rep, N, S, n_c, n_u))
    idx_dev, t_up = T(lambda: indices_to_device(idx, ctx))  # (what the driver's worker thread does, beside the previous solve)
    res, t_gather = T(lambda: full.gather(idx_dev))
    s, t_create = T(lambda: Solver(res, u0, a0, L.DMF_MODE_PARTIAL))
    _, t_step = T(lambda: s.step(20, 20, 0.0))
    _, t_copy = T(lambda: s.copy_u_to(stack[rep]))
    _, t_alpha = T(lambda: s.get_alpha())
    _, t_close = T(lambda: (s.close(), res.close()))
    print(f"rep {rep}: host row draw {t_idx:.1f} ms, host init {t_init:.1f}, index upload {t_up:.1f} | gather + problem set-up {t_gather:.1f} | solver create (host u0) {t_create:.1f} | "
          f"20 iterations {t_step:.1f} | copy u to the stack {t_copy:.2f} | alpha to host {t_alpha:.2f} | close {t_close:.2f}")
