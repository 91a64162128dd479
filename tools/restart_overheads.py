"""Where does a restart's time outside its iteration loop go?  (diagnostic; headline shape, synthetic data)
   python tools/restart_overheads.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from demethify_amd import _lib as L, staging
from demethify_amd.device import Context, Problem, Solver

N, S, n_c, n_u = bench.WORKLOADS["headline_1e6x256_12+4"]
dev = torch.device("cuda", 0)
V, D, Rt = bench.make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0)
ctx = Context(0)
problem = Problem(ctx, V, D, Rt)
def T(f, sync=True):
    t = time.perf_counter(); r = f()
    if sync: ctx.synchronize()
    return r, (time.perf_counter() - t) * 1e3
for rep in range(4):
    (u0, a0), t_init = T(lambda: bench.restart_init(rep, N, S, n_c, n_u), sync=False)
    (du, da), t_up = T(lambda: staging.to_device((u0, a0), ctx))
    s, t_create_host = T(lambda: Solver(problem, u0, a0, L.DMF_MODE_PARTIAL))
    _, t_close = T(lambda: s.close())
    s, t_create_dev = T(lambda: Solver(problem, du, da, L.DMF_MODE_PARTIAL))
    _, t_step = T(lambda: s.step(20, 20, 0.0))
    _, t_cost = T(lambda: s.direct_cost())
    _, t_close2 = T(lambda: s.close())
    _, t_free = T(lambda: (du.close(), da.close()))
    print(f"rep {rep}: host init {t_init:.1f} ms | stage upload {t_up:.1f} | create(host arrays) {t_create_host:.1f} | close {t_close:.2f} | "
          f"create(device arrays) {t_create_dev:.1f} | 20 iterations {t_step:.1f} | direct_cost {t_cost:.2f} | close {t_close2:.2f} | free staged {t_free:.2f}")
