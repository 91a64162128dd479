"""BASELINE config 5 end to end on one GPU: the unsupervised BIC sweep n_u = 2..12 at 5e5 CpG x 128 samples
(ic.evaluate_best_ic: one solve per candidate, cost + criterion on the result), bounded to a fixed number of outer
iterations per candidate so that the run time does not depend on the data's convergence.
   python tools/ic_sweep_bench.py [outer iterations per candidate] [lo hi]      (upstream's own sweep: lo hi = 1 25)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from demethify_amd.ic import evaluate_best_ic

T1 = int(sys.argv[1]) if len(sys.argv) > 1 else 50
LO, HI = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 12)
N, S, K_true = 500_000, 128, 6
rs = np.random.RandomState(0)
R = rs.beta(0.5, 0.5, size=(N, K_true))
A = rs.dirichlet(np.ones(K_true), S).T
D = rs.poisson(50, (N, S)) + 1
V = rs.binomial(D, np.clip(R @ A, 0, 1)) / D
D = D.astype(np.int64)
for rep in range(3):
    t0 = time.perf_counter()
    u, alpha, n_u, scores = evaluate_best_ic(V, None, D, "uniform_", "BIC", 1, T1, 20, 0.0, n_u_values=range(LO, HI + 1))
    dt = time.perf_counter() - t0
    print(f"run {rep}: BIC sweep n_u = {LO}..{HI} at {N} x {S}, {T1} outer iterations per candidate: {dt:.2f} s wall incl. the 1 GB upload; "
          f"picked n_u = {n_u} (data drawn with {K_true} types); {(HI - LO + 1) * T1 / dt:.0f} outer iterations/s over the sweep")
