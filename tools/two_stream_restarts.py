"""Experiment: the 64-restart job of bench.py with the restarts dealt to TWO contexts (streams) of one GPU, a Python thread
each, against one context -- does a second stream fill the gaps between a restart's kernels and the next one's set-up?
   python tools/two_stream_restarts.py [restarts] [steps]"""
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import make_inputs_on_device, restart_init
from demethify_amd import _lib as L
from demethify_amd import staging
from demethify_amd.device import Context, Problem, Solver

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, S, n_c, n_u = 1_000_000, 256, 12, 4
dev = torch.device("cuda", 0)
V, D, Rt = make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0)


def job(n_streams):
    ctxs = [Context(0) for _ in range(n_streams)]
    probs = [Problem(c, V, D, Rt) for c in ctxs]
    costs = {}

    def run(i):
        ctx, prob = ctxs[i], probs[i]
        mine = list(range(i, R, n_streams))
        feed = staging.Prefetcher(mine, lambda k: staging.to_device(restart_init(k, N, S, n_c, n_u), ctx), depth=2, workers=1)
        waiting = None
        for k, (u0, a0) in feed:
            s = Solver(prob, u0, a0, L.DMF_MODE_PARTIAL)
            if waiting is not None:
                costs[waiting[0]] = waiting[1].cost_end()
                waiting[1].close()
            s.step(K, 20, 0.0)
            s.cost_begin()
            waiting = (k, s)
        costs[waiting[0]] = waiting[1].cost_end()
        waiting[1].close()

    for i in range(n_streams):  # warm-up
        with Solver(probs[i], *restart_init(0, N, S, n_c, n_u), L.DMF_MODE_PARTIAL) as s:
            s.step(3, 20, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i,)) for i in range(n_streams)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in ctxs:
        c.synchronize()
    dt = time.perf_counter() - t0
    for p in probs:
        p.close()
    for c in ctxs:
        c.close()
    return dt, costs


for n in (1, 2, 1, 2):
    dt, costs = job(n)
    print(f"{n} stream(s): {R} restarts x {K} iterations in {dt:.3f} s -> {R * K / dt:.0f} outer it/s; min cost {min(costs.values()):.6f}", flush=True)
