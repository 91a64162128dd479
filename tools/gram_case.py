#!/usr/bin/env python3
"""Diagnostic: a few outer iterations of one large-n_u case (for rocprofv3 --kernel-trace --stats)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import make_inputs_on_device
from demethify_amd import _lib as L
from demethify_amd.device import Context, Problem, Solver

N, S, n_c, n_u = 500_000, 128, int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
ctx = Context(0)
V, D, Rt = make_inputs_on_device(torch, dev, N, S, max(n_c, 1), n_u, seed=0)
p = Problem(ctx, V, D, Rt if n_c else None)
rs = np.random.RandomState(1)
s = Solver(p, rs.uniform(size=(N, n_u)), rs.dirichlet(np.ones(n_c + n_u), S).T,
           L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED)
s.step(5, 20, 0.0)
ctx.synchronize()
s.close()
p.close()
ctx.close()
print("done", flush=True)
