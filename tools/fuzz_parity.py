"""Randomised differential run: the device solver against the CPU oracle on random small shapes (kernel selection level 0),
including ragged sample counts, partial last blocks, zero-coverage cells, one and two count digits, wide and narrow
reference blocks.   python tools/fuzz_parity.py [cases] [seed] [wide|many]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import solver as osol
from demethify_amd import _lib as L
from demethify_amd.device import Context, Problem, Solver

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
many = len(sys.argv) > 3 and sys.argv[3] == "many"  # bias towards 17..48 known types
wide = len(sys.argv) > 3 and sys.argv[3] in ("wide", "many")  # bias towards 5..16 unknowns on shapes k_cm_i8 / k_inner_bu take
ctx = Context(0)
worst, t0, paths = 0.0, time.time(), {}
for case in range(n_cases):
    N = int(rng.choice([rng.randint(1, 40), rng.randint(40, 600), rng.randint(600, 6000)]))
    S = int(rng.choice([2 * rng.randint(1, 129), rng.randint(1, 300), 64, 128, 256]))
    n_c = int(rng.choice([0, rng.randint(1, 17)]))
    if many and rng.rand() < 0.7:
        n_c = int(rng.randint(17, 49))  # reference atlases: more than 16 known types
    n_u = int(rng.randint(1, 17)) if (wide and rng.rand() < 0.7) else int(rng.randint(1, 9 if n_c else 13))
    if wide and rng.rand() < 0.25:
        n_u = int(rng.randint(17, 33))  # two DPP rows per CpG row, pair tiles over several producer launches
    if wide and rng.rand() < 0.6:
        S = 4 * int(rng.randint(1, 65))  # the wide-row-group producer takes S % 4 == 0, S <= 256
    n_u = min(n_u, 64 - n_c)
    depth = int(rng.choice([5, 40, 120, 900, 20000]))
    T1 = int(rng.randint(1, 4))
    V, D, Rt = osol.synthetic_problem(N, S, max(n_c, 1), n_u, seed=int(rng.randint(1 << 30)), depth=depth)
    if rng.rand() < 0.5:
        D[:: int(rng.randint(2, 9)), :: int(rng.randint(1, 5))] = 0
        V = np.where(D == 0, 0.0, V)
    Rt = Rt[:, :n_c] if n_c else None
    rs = np.random.RandomState(case)
    u0 = rs.uniform(size=(N, n_u))
    a0 = rs.dirichlet(np.ones(n_c + n_u), S).T.copy()
    if n_c:
        wu, wa = osol.solve_partial(u0.copy(), np.c_[Rt, u0], a0.copy(), V, D, Rt, n_u, T1, 20, 0.0,
                                    project=osol.simplex_project_columns_fast)
        mode = L.DMF_MODE_PARTIAL
    else:
        wu, wa = osol.solve_unsupervised(V, n_u, D, "uniform_", T1, 20, 0.0, init=(u0.copy(), a0.copy()),
                                         project=osol.simplex_project_columns_fast)
        mode = L.DMF_MODE_UNSUPERVISED
    if not (np.isfinite(wu).all() and np.isfinite(wa).all()):
        continue  # the reference itself divides by a vanished Lipschitz bound on this input (e.g. one CpG row)
    with Problem(ctx, V, D, Rt) as p, Solver(p, u0, a0, mode) as s:
        path = s.describe(20)
        s.step(T1, 20, 0.0)
        u, alpha, cost, _ = s.get()
        direct = s.direct_cost()
    err = max(np.abs(alpha - wa).max(), np.abs(u - wu).max())
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    cerr = abs(direct - want) / max(abs(want), 1e-300)
    key = " ".join(tok.split("<")[0] for tok in path.split() if not tok.startswith(("nw=", "grid=", "tail="))) + \
          ("  nd=2" if "nd=2" in path else "") + ("  /w8" if "/w8" in path else "")
    paths[key] = paths.get(key, 0) + 1
    worst = max(worst, err)
    flag = "" if err < 1e-8 and cerr < 1e-10 else "   <<<<<< MISMATCH"
    if flag or case % 20 == 0:
        print(f"case {case}: N={N} S={S} {n_c}+{n_u} depth={depth} T1={T1}  |diff|={err:.2e} cost rel {cerr:.1e}  {path}{flag}", flush=True)
    if flag:
        sys.exit(1)
print(f"{n_cases} cases, worst |diff| {worst:.2e}, {time.time() - t0:.0f} s")
for k, v in sorted(paths.items(), key=lambda kv: -kv[1]):
    print(f"  {v:4d}  {k}")
