"""Model selection: how many unknown cell types?  Host-side mirror of demethify/ic.py.

The information criteria are kept exactly as coded upstream (ic.py:11-22), including the BIC
expression that is not the textbook one.  The sweep over candidate n_u (ic.py:169-218, hard-coded
1..25 upstream) is embarrassingly parallel: with torch.distributed initialised the candidates are
dealt to the ranks longest-first and only the per-candidate scores and the winner's factors are
exchanged.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from . import shard
from .deconvolution import _init_unsupervised, cost_f_w, init_BSSMF_md, solve_problem
from .device import Problem, Solver, get_context
from .staging import Prefetcher
from .init_func import wls_intercept

__all__ = ["compute_bic", "compute_aic", "compute_consensus_matrix", "compute_ccc", "run_deconvolution",
           "bicross_validation", "evaluate_best_ic"]


def _n_free(n_u, n_cpg, n_ct, n_samples):
    return n_u * n_cpg + (n_ct + n_u - 1) * n_samples


def compute_bic(cost, n_u, n_cpg, n_ct, n_samples):
    """ic.py:11-15, as coded."""
    l = n_samples * n_cpg
    k = _n_free(n_u, n_cpg, n_ct, n_samples)
    return 2 * np.log(cost) * k * np.log(l) + (k * np.log(l) * (k + 1)) / (l - k - 1)


def compute_aic(cost, n_u, n_cpg, n_ct, n_samples):
    """ic.py:18-22."""
    l = n_samples * n_cpg
    k = _n_free(n_u, n_cpg, n_ct, n_samples)
    return l * np.log(cost / l) + 2 * k + (2 * k * (k + 1)) / (l - k - 1)


def compute_consensus_matrix(alpha_runs):
    """ic.py:24-37: fraction of runs in which two samples share their dominant cell type."""
    labels = np.stack([np.argmax(a, axis=0) for a in alpha_runs])  # (runs, samples)
    same = labels[:, :, None] == labels[:, None, :]
    return same.sum(axis=0) / float(len(alpha_runs))


def compute_ccc(alpha_runs):
    """ic.py:40-45: Brunet's cophenetic correlation coefficient of the consensus matrix."""
    from scipy.cluster.hierarchy import cophenet, linkage
    from scipy.spatial.distance import pdist

    distances = pdist(compute_consensus_matrix(alpha_runs), metric="euclidean")
    ccc, _ = cophenet(linkage(distances, method="average"), distances)
    return ccc


def _solve(problem, meth_f, counts, ref, n_u, init_option, seed, iter1, iter2, tol):
    if ref is not None:
        u0, _, a0 = init_BSSMF_md(init_option, meth_f, counts, ref, n_u, seed=seed, rb_alg=wls_intercept, _stack=False)
        mode = L.DMF_MODE_PARTIAL
    else:
        u0, a0 = _init_unsupervised(init_option, meth_f, n_u, seed)
        mode = L.DMF_MODE_UNSUPERVISED
    return solve_problem(problem, u0, a0, mode, iter1, iter2, tol)


def run_deconvolution(meth_f, counts, ref, n_u, init_option, seed, iter1, iter2, tol, problem=None):
    """ic.py:47-55 -> (u, R, alpha).  ``problem`` lets a sweep reuse one device-resident upload."""
    own = problem is None
    if own:
        problem = Problem(get_context(), meth_f, counts, ref)
    try:
        u, alpha = _solve(problem, meth_f, counts, ref, n_u, init_option, seed, iter1, iter2, tol)
    finally:
        if own:
            problem.close()
    R = np.hstack((ref, u.reshape(-1, n_u))) if ref is not None else u
    return u, R, alpha


def bicross_validation(meth_f, n_u, counts, iter1, iter2, tol, n_folds=10, seed=None, ref=None,
                       init_option="uniform_", fraction=0.3):
    """ic.py:58-89: random-mask hold-out error (returns the SUM over folds, as upstream)."""
    np.random.seed(seed)
    total_press, best_u, best_alpha, min_error = 0, None, None, float("inf")
    for _ in range(n_folds):
        train_mask = np.random.rand(*meth_f.shape) < fraction
        test_mask = ~train_mask
        if np.sum(test_mask) == 0 or np.sum(train_mask) == 0:
            continue
        u, R, alpha = run_deconvolution(meth_f * train_mask, counts * train_mask, ref, n_u, init_option, seed,
                                        iter1, iter2, tol)
        test_error = np.linalg.norm((meth_f - R @ alpha) * test_mask, "fro") ** 2 / np.sum(test_mask)
        total_press += test_error
        if test_error < min_error:
            min_error, best_u, best_alpha = test_error, u, alpha
    return total_press, best_u, best_alpha


def evaluate_best_ic(meth_f, ref, counts, init_option, ic, seed, iter1, iter2, tol, n_restarts=5,
                     n_u_values=None):
    """ic.py:169-218 -> (u, alpha, n_u, list of criterion values).

    ``n_u_values`` defaults to upstream's hard-coded ``range(1, 26)`` (ic.py:171).  AIC / BIC sweeps
    are sharded across the ranks of an initialised torch.distributed job; CCC and BCV run serially.
    """
    default_range = n_u_values is None
    if default_range:
        n_u_values = range(1, 25 + 1)
    n_u_values = list(n_u_values)
    n_cpg, n_samples = meth_f.shape
    n_ct = ref.shape[1] if ref is not None else 0

    if ic == "minka":
        # upstream calls run_deconvolution with 6 of its 9 arguments here (ic.py:189)
        raise TypeError("run_deconvolution() missing 3 required positional arguments: 'iter1', 'iter2', and 'tol'")

    if ic in ("CCC", "BCV"):
        best_ic, best = float("inf"), (None, None, None)
        scores = []
        for n_u in n_u_values:
            if ic == "CCC":
                runs = []
                for restart in range(n_restarts):
                    u, _, alpha = run_deconvolution(meth_f, counts, ref, n_u, init_option, seed + restart, iter1,
                                                    iter2, tol)
                    runs.append(alpha)
                score = -compute_ccc(runs)
            else:
                score, u, alpha = bicross_validation(meth_f, n_u, counts, iter1, iter2, tol, fraction=0.3,
                                                     n_folds=n_restarts, seed=seed, ref=ref,
                                                     init_option=init_option)
            scores.append(score)
            if score < best_ic:
                best_ic, best = score, (u, alpha, n_u)
        return best[0], best[1], best[2], scores

    # AIC / BIC: one solve per candidate; candidates dealt to ranks longest-first (cost grows with n_u)
    rank, world, _ = shard.dist_state()
    # (checked on every rank before any solve, so that all ranks fail -- or trim -- together)
    not_positive = [n for n in n_u_values if n < 1]
    if not_positive:
        raise ValueError(f"candidate n_u values {not_positive}: a number of unknown cell types is at least 1")
    too_many = [n for n in n_u_values if n_ct + n > L.MAX_K]
    if too_many and default_range:
        # upstream's hard-coded 1..25 (ic.py:171) with a reference matrix of 40 or more known types: the kernels take
        # L.MAX_K cell types in total, so the sweep ends where they do -- a deliberate difference, said out loud
        n_u_values = [n for n in n_u_values if n_ct + n <= L.MAX_K]
        if rank == 0:
            print(f"note: with {n_ct} known cell types the sweep stops at {L.MAX_K - n_ct} unknown ones "
                  f"({L.MAX_K} cell types in total is what the kernels are built for; upstream would go on to 25)")
        if not n_u_values:
            raise ValueError(f"{n_ct} known cell types leave no room for an unknown one ({L.MAX_K} in total at most)")
    elif too_many:
        raise ValueError(f"candidate n_u values {too_many} need more than {L.MAX_K} cell types in total "
                         f"({n_ct} known): outside what the kernels are built for")
    order = sorted(range(len(n_u_values)), key=lambda i: -n_u_values[i])
    mine = [order[i] for i in range(rank, len(order), world)]
    formula = compute_bic if ic == "BIC" else compute_aic
    local, keep = [], None  # keep = this rank's best candidate only (the reference keeps the running best, ic.py:212)

    def draw(i):
        # the candidate's initialisation (numpy's global generator: ONE worker thread), drawn while the GPU solves the
        # candidate before it
        n_u = n_u_values[i]
        if ref is not None:
            u0, _, a0 = init_BSSMF_md(init_option, meth_f, counts, ref, n_u, seed=seed, rb_alg=wls_intercept, _stack=False)
        else:
            u0, a0 = _init_unsupervised(init_option, meth_f, n_u, seed)
        return u0, a0

    mode = L.DMF_MODE_PARTIAL if ref is not None else L.DMF_MODE_UNSUPERVISED
    with Problem(get_context(), meth_f, counts, ref) as problem:
        feed = Prefetcher(mine, draw, depth=1, workers=1)
        try:
            for i, (u0, a0) in feed:
                n_u = n_u_values[i]
                with Solver(problem, u0, a0, mode) as s:
                    s.step(iter1, iter2, tol)
                    cost = s.direct_cost()  # cost_f_w(meth_f, R, alpha, counts), ic.py:206, where the iterate lives
                    score = float(formula(cost, n_u, n_cpg, n_ct, n_samples))
                    local.append((i, score))
                    # strict '<' on the score, lowest candidate index among equal scores: what ic.py:212 does serially
                    # (a NaN score never wins, exactly as `ic_result < best_ic` upstream); only a candidate that becomes
                    # this rank's best leaves the device
                    if score < (keep[0] if keep else float("inf")) or (keep and score == keep[0] and i < keep[1]):
                        u, alpha, _, _ = s.get()
                        keep = (score, i, u, alpha)
        finally:
            feed.close()
    scores = [s for _, s in shard.gather_objects(local)]
    best_i, best_score = None, float("inf")
    for i, score in enumerate(scores):  # running strict minimum in candidate order, ic.py:212-216
        if score < best_score:
            best_i, best_score = i, score
    if best_i is None:  # every score NaN / inf: upstream returns its initial Nones
        return None, None, None, scores
    n_best = n_u_values[best_i]
    owner = order.index(best_i) % world
    if rank == owner:
        assert keep is not None and keep[1] == best_i
        payload = (keep[2], keep[3])
    else:
        payload = (np.empty((n_cpg, n_best)), np.empty((n_ct + n_best, n_samples)))
    u, alpha = shard.broadcast_arrays(payload, owner)
    return u, alpha, n_best, scores
