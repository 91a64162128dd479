"""Numpy restatement of the loops that call the solver many times — TEST INFRASTRUCTURE ONLY.

Restart pick (demethify/demethify.py:163-177,195-203), bootstrap (demethify/bootstrap.py:10-93,
arithmetic only: no CSV writing) and the AIC/BIC sweep (demethify/ic.py:47-55,169-218).
"""
from __future__ import annotations

import numpy as np

from . import solver as S


def run_one(V, D, Rt, n_u, init_option, seed, iter1, iter2, tol, project=S.simplex_project_columns):
    """demethify/ic.py:47-55 (`run_deconvolution`) -> (u, R, alpha)."""
    if Rt is not None:
        u, R, alpha = S.init_partial(init_option, V, D, Rt, n_u, seed=seed)
        u, alpha = S.solve_partial(u, R, alpha, V, D, Rt, n_u, n_iter1=iter1, n_iter2=iter2, tol=tol,
                                   project=project)
        R = np.hstack((Rt, u.reshape(-1, n_u)))
    else:
        u, alpha = S.solve_unsupervised(V, n_u, D, init_option, n_iter1=iter1, n_iter2=iter2, tol=tol,
                                        seed=seed, project=project)
        R = u
    return u, R, alpha


def restart_pick(V, D, Rt, n_u, init_option, seeds, iter1, iter2, tol):
    """demethify/demethify.py:195-203 with an explicit seed per restart (upstream passes the
    same seed to every restart, so its loop is idempotent); strict '<' keeps the first minimum."""
    best = (float("inf"), None, None, -1)
    costs = []
    for k, seed in enumerate(seeds):
        u, R, alpha = run_one(V, D, Rt, n_u, init_option, seed, iter1, iter2, tol)
        c = S.weighted_cost(V, R, alpha, D)
        costs.append(c)
        if c < best[0]:
            best = (c, u, alpha, k)
    return best[1], best[2], best[3], costs


def ic_sweep(V, Rt, D, init_option, ic, seed, iter1, iter2, tol, n_u_values=range(1, 26),
             project=S.simplex_project_columns):
    """demethify/ic.py:169-218 for ic in {"AIC","BIC"}; upstream hard-codes range(1, 26)."""
    n_cpg, n_samples = V.shape
    n_ct = Rt.shape[1] if Rt is not None else 0
    best_ic, best = float("inf"), (None, None, None)
    scores = []
    for n_u in n_u_values:
        u, R, alpha = run_one(V, D, Rt, n_u, init_option, seed, iter1, iter2, tol, project=project)
        cost = S.weighted_cost(V, R, alpha, D)
        fn = S.bic_as_coded if ic == "BIC" else S.aic_as_coded
        score = fn(cost, n_u, n_cpg, n_ct, n_samples)
        scores.append(score)
        if score < best_ic:
            best_ic, best = score, (u, alpha, n_u)
    return best[0], best[1], best[2], scores


def bootstrap_replicates(n_bootstrap, n_u, V, D, Rt, init_option, iter1, iter2, tol, seed):
    """demethify/bootstrap.py:26-46 (partial-reference branch): returns the per-replicate
    (u, alpha) stacks, shapes (B, N, n_u) and (B, K, S)."""
    us, alphas = [], []
    for s in S.bootstrap_seeds(seed, n_bootstrap):
        idx = S.bootstrap_indices(s, V.shape[0])
        Vb, Db, Rb = V[idx], D[idx], Rt[idx]
        u, R, alpha = S.init_partial(init_option, Vb, Db, Rb, n_u, seed=s)
        u, alpha = S.solve_partial(u, R, alpha, Vb, Db, Rb, n_u, iter1, iter2, tol)
        us.append(u)
        alphas.append(alpha)
    return np.stack(us), np.stack(alphas)


def percentile_bounds(stack, confidence_level):
    """demethify/bootstrap.py:12-14,53-54,77-78 — numpy default (linear) percentiles over axis 0."""
    a = 1 - confidence_level / 100
    lo = np.percentile(stack, 100 * (a / 2), axis=0)
    hi = np.percentile(stack, 100 * (1 - (a / 2)), axis=0)
    return lo, hi
