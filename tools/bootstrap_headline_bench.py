"""Bootstrap replicates per second at the headline size (BASELINE.json configs[3]: 1e6 CpG x 256 samples, 12 + 4 types):
bt_ci's replicate loop (row resample -> device gather -> init -> solve -> profiles kept in HBM) with a fixed number of
outer iterations, then the percentile step.  python tools/bootstrap_headline_bench.py [replicates] [outer iterations]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import solver as osol  # data generator only
from demethify_amd.bootstrap import bt_ci

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T1 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, S, n_c, n_u = 1_000_000, 256, 12, 4
rs = np.random.RandomState(0)
Rfull = rs.beta(0.5, 0.5, size=(N, n_c + n_u))
A = rs.dirichlet(np.ones(n_c + n_u), S).T
D = rs.poisson(50, (N, S)) + 1
V = rs.binomial(D, np.clip(Rfull @ A, 0, 1)) / D
ref = np.ascontiguousarray(Rfull[:, :n_c])
with tempfile.TemporaryDirectory() as out:
    t0 = time.perf_counter()
    bt_ci(95, B, n_u, V, D.astype(np.int64), ref, "uniform_", T1, 20, 0.0, [f"t{k}" for k in range(n_c)], out,
          [f"s{k}" for k in range(S)], None, 1, materialize=False)  # as the CLI calls it
    dt = time.perf_counter() - t0
print(f"{B} replicates x {T1} outer iterations at {N} x {S}, {n_c}+{n_u}: {dt:.2f} s wall incl. upload, percentiles and CSV "
      f"writing -> {B / dt:.2f} replicates/s")
