"""Diagnostic: a few outer iterations of one shape (for rocprofv3 --kernel-trace --stats).
   python tools/one_shape.py N S n_c n_u [iterations]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import make_inputs_on_device
from demethify_amd import _lib as L
from demethify_amd.device import Context, Problem, Solver

N, S, n_c, n_u = (int(x) for x in sys.argv[1:5])
T1 = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda", 0)
ctx = Context(0)
V, D, Rt = make_inputs_on_device(torch, dev, N, S, max(n_c, 1), n_u, seed=0)
p = Problem(ctx, V, D, Rt if n_c else None)
rs = np.random.RandomState(1)
u0 = rs.uniform(size=(N, n_u)); a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
s = Solver(p, u0, a0, L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED)
print(s.describe(20))
s.step(T1, 20, 0.0); ctx.synchronize()
