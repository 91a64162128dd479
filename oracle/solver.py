"""Numpy restatement of DeMethify's weighted alternating accelerated projected-gradient solver.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This is the "reference CPU
numpy path": every function keeps the reference's operation order so that it
reproduces the outputs the reference's authors committed under
``/root/reference/test/`` (see tests/test_oracle_golden.py, which pins it to
<=1e-11 on four folders).  Citations are ``file:line`` under /root/reference.

Parity status: PINNED by the reference's own committed outputs
(test/output_partial_ref, test/unsupervised, test/output_ref_based,
test/model_selection, test/purity).  The reference itself is not imported:
its ``numba`` dependency is absent from this image and stays absent.

Notation: V = meth_frequency (N x S, f64), D = d_x / counts (N x S, int64 or
f64), Rt = R_trunc (N x n_c), u (N x n_u), alpha (K x S), K = n_c + n_u.
"""
from __future__ import annotations

import numpy as np
import numpy.random as rd
from scipy.optimize import nnls

__all__ = [
    "set_seed", "weighted_cost", "simplex_project_columns", "nnls_intercept_proportions",
    "momentum_step", "u_phase", "alpha_phase", "init_partial", "solve_partial",
    "solve_unsupervised", "init_unsupervised", "bootstrap_indices", "bootstrap_seeds",
    "bic_as_coded", "aic_as_coded", "synthetic_problem",
]


def set_seed(seed=None):
    """demethify/deconvolution.py:9-11 — seeds the legacy global MT19937 stream.

    ``seed`` may be an int or a 1-element list (CLI ``--seed 5`` yields ``[5]``,
    demethify/demethify.py:43); the two give different streams, as upstream.
    """
    if seed is not None:
        rd.seed(seed)


def weighted_cost(V, R, alpha, D):
    """demethify/deconvolution.py:15-17 — ||sqrt(D) * (V - R alpha)||_F^2 via norm()**2."""
    resid = V - R @ alpha
    return np.linalg.norm(np.sqrt(D) * resid) ** 2


def simplex_project_columns(X, z=1):
    """demethify/deconvolution.py:21-37 — sort-based projection of each column onto the simplex.

    rho is the LAST index j with sorted_j - (cumsum_j - z)/(j+1) > 0.
    """
    p, n = X.shape
    out = np.zeros_like(X)
    for col in range(n):
        srt = np.sort(X[:, col])[::-1]
        shifted = np.cumsum(srt) - z
        rho = -1
        for j in range(p):
            if srt[j] - shifted[j] / (j + 1) > 0:
                rho = j
        theta = shifted[rho] / (rho + 1)
        for j in range(p):
            out[j, col] = max(X[j, col] - theta, 0)
    return out


def simplex_project_columns_fast(X, z=1):
    """Vectorised equivalent of simplex_project_columns (same arithmetic per column).

    Used by the timed CPU baseline so that the un-jitted Python double loop of the
    restatement does not inflate the CPU time (upstream runs it under numba).
    """
    p, n = X.shape
    srt = -np.sort(-X, axis=0)
    shifted = np.cumsum(srt, axis=0) - z
    ranks = np.arange(1, p + 1, dtype=X.dtype)[:, None]
    ok = (srt - shifted / ranks) > 0
    # last True per column; -1 (python wrap-around, as upstream) when none is True
    rho = np.where(ok.any(axis=0), p - 1 - np.argmax(ok[::-1], axis=0), -1)
    theta = shifted[rho, np.arange(n)] / (rho + 1)
    return np.maximum(X - theta[None, :], 0)


def nnls_intercept_proportions(x, d_x, R_full):
    """demethify/init_func.py:8-14 (`wls_intercept`).

    Upstream calls scikit-learn ``LinearRegression(fit_intercept=True, positive=True)
    .fit(R_full, x, d_x.ravel())``.  scikit-learn (pinned 1.2.2 in requirements.txt:5)
    centres X and y by their weighted means, scales rows by sqrt(w) and solves
    ``scipy.optimize.nnls``; restated here without scikit-learn.  Pinned by
    test/output_ref_based/celltypes_proportions.csv.
    """
    w = np.asarray(d_x, dtype=np.float64).ravel()
    X = np.asarray(R_full, dtype=np.float64)
    y = np.asarray(x, dtype=np.float64).reshape(X.shape[0], -1)
    x_mean = np.average(X, axis=0, weights=w)
    y_mean = np.average(y, axis=0, weights=w)
    sw = np.sqrt(w)[:, None]
    Xc = (X - x_mean) * sw
    yc = (y - y_mean) * sw
    coef = np.vstack([nnls(Xc, yc[:, j])[0] for j in range(yc.shape[1])])  # (targets, K)
    temp = coef.T
    return temp / max(temp.sum(), 1e-10)


def momentum_step(a_prev, l_prev, l_cur):
    """Shared scalar recurrence, demethify/deconvolution.py:83-85 and :95-97."""
    a_next = (1 + np.sqrt(1 + 4 * a_prev * a_prev)) / 2
    beta = min((a_prev - 1) / a_next, 0.9999 * np.sqrt(l_prev / l_cur))
    return a_next, beta


def u_phase(u, alpha, n_iter2, a1, l_w_, l_w, u_, V, Rt, n_u, D):
    """demethify/deconvolution.py:81-90 (`update_u`): n_iter2 accelerated projected-gradient
    steps on the unknown profiles with alpha fixed; gradient taken at the extrapolated point."""
    A_known = alpha[:-n_u]
    A_unk = alpha[-n_u:]
    for _ in range(n_iter2):
        a0 = a1
        a1, beta_w = momentum_step(a0, l_w_, l_w)
        u_temp = u + beta_w * (u - u_)
        u_ = u
        u = np.clip((u_temp + (D * ((V - Rt @ A_known - u_temp @ A_unk)) @ A_unk.T) / l_w), 0, 1)
        l_w_ = l_w
    return u, u_, a1, l_w_


def alpha_phase(n_iter2, alpha, a2, l_h_, l_h, alpha_, R, D, V, project=simplex_project_columns):
    """demethify/deconvolution.py:93-102 (`update_alpha`)."""
    for _ in range(n_iter2):
        a0 = a2
        a2, beta_h = momentum_step(a0, l_h_, l_h)
        alpha_temp = alpha + beta_h * (alpha - alpha_)
        alpha_ = alpha
        alpha = project(alpha_temp + (R.T @ (D * (V - R @ alpha_temp))) / l_h)
        l_h_ = l_h
    return alpha, alpha_, a2, l_h_


def _guard_first_unknown_row(alpha, n_u):
    """demethify/deconvolution.py:74-76 — if the first unknown row is all-zero... (as coded:
    ``alpha[-n_u:][0].all() == 0.0`` is True when ANY entry of that row is zero)."""
    if alpha[-n_u:][0].all() == 0.0:
        alpha[-n_u:][0] = 1e-10
        alpha[:-n_u] = (1 - 1e-10) * alpha[:-n_u]
    return alpha


def init_partial(init_option, V, D, Rt, n_u, seed=None):
    """demethify/deconvolution.py:40-78 (`init_BSSMF_md`), options uniform_/beta/uniform.

    RNG order (:55-56): uniform (N x n_u) first, then Dirichlet(ones(K), S).T.
    SVD / ICA initialisers are outside the hot-path scope.
    """
    set_seed(seed)
    S = V.shape[1]
    N, n_c = Rt.shape
    if init_option != "uniform_" and n_u > S:
        init_option = "uniform_"
    if init_option == "uniform":
        u = rd.uniform(size=(N, n_u))
        full = np.c_[Rt, u]
        alpha = np.concatenate(
            [nnls_intercept_proportions(V[:, k:k + 1], D[:, k:k + 1], full) for k in range(S)], axis=1)
    elif init_option == "uniform_":
        u = rd.uniform(size=(N, n_u))
        alpha = rd.dirichlet(np.ones(n_c + n_u), S).T
    elif init_option == "beta":
        half = np.ones((N, n_u)) * 0.5
        u = rd.beta(half, half)
        alpha = rd.dirichlet(np.ones(n_c + n_u), S).T
    else:
        raise NotImplementedError(f"init option {init_option!r} is outside the oracle's scope")
    R = np.c_[Rt, u]
    alpha = _guard_first_unknown_row(alpha, n_u)
    return u, R, alpha


def solve_partial(u, R, alpha, V, D, Rt, n_u, n_iter1=100000, n_iter2=50, tol=1e-3,
                  project=simplex_project_columns, trace=None):
    """demethify/deconvolution.py:190-223 (`mdwbssmf_deconv`).

    Momentum scalars a1/a2 and the previous Lipschitz bounds persist across outer
    iterations.  ``trace`` (a list) receives the cost after every outer iteration.
    """
    a1 = 1.0
    a2 = 1.0
    u_ = u.copy()
    alpha_ = alpha.copy()
    d = D.max() ** 2
    l_w = (np.linalg.norm(alpha[-n_u:]) ** 2) * d
    l_w_ = l_w
    l_h = (np.linalg.norm(R) ** 2) * d
    l_h_ = l_h
    cf = weighted_cost(V, R, alpha, D)
    for _ in range(n_iter1):
        cf_0 = cf
        u, u_, a1, l_w_ = u_phase(u, alpha, n_iter2, a1, l_w_, l_w, u_, V, Rt, n_u, D)
        R = np.hstack((Rt, u.reshape(-1, n_u)))
        l_h = (np.linalg.norm(R) ** 2) * d
        alpha, alpha_, a2, l_h_ = alpha_phase(n_iter2, alpha, a2, l_h_, l_h, alpha_, R, D, V, project)
        l_w = (np.linalg.norm(alpha[-n_u:]) ** 2) * d
        cf = weighted_cost(V, R, alpha, D)
        if trace is not None:
            trace.append(cf)
        if abs(cf - cf_0) < tol:
            break
    return u, alpha


def init_unsupervised(init_option, V, n_u, seed=None):
    """demethify/deconvolution.py:108-127 — inline init of `unsupervised_deconv`.

    ``uniform`` upstream hits an undefined name (:117, NameError); reproduced as NameError.
    """
    set_seed(seed)
    N, S = V.shape
    if init_option != "uniform_" and n_u > S:
        init_option = "uniform_"
    if init_option == "uniform":
        raise NameError("name 'R_trunc' is not defined")  # upstream bug kept, deconvolution.py:117
    if init_option == "uniform_":
        u = rd.uniform(size=(N, n_u))
        alpha = rd.dirichlet(np.ones(n_u), S).T
    elif init_option == "beta":
        half = np.ones((N, n_u)) * 0.5
        u = rd.beta(half, half)
        alpha = rd.dirichlet(np.ones(n_u), S).T
    else:
        raise NotImplementedError(f"init option {init_option!r} is outside the oracle's scope")
    return u, alpha


def solve_unsupervised(V, n_u, D, init_option, n_iter1=100000, n_iter2=20, tol=1e-3, seed=None,
                       project=simplex_project_columns, trace=None, init=None):
    """demethify/deconvolution.py:107-184 (`unsupervised_deconv`).

    Differs from the partial-reference loop in ONE place: the u-gradient is evaluated at
    the previous iterate ``u`` rather than at ``u_temp`` (:163; ``u_ = u`` at :162 only
    aliases).  ``init`` may carry a precomputed (u, alpha) pair (used by tests).
    """
    if init is None:
        u, alpha = init_unsupervised(init_option, V, n_u, seed)
    else:
        u, alpha = init
    a1 = 1.0
    a2 = 1.0
    u_ = u.copy()
    alpha_ = alpha.copy()
    d = D.max() ** 2
    l_w = (np.linalg.norm(alpha[-n_u:]) ** 2) * d
    l_w_ = l_w
    l_h = (np.linalg.norm(u) ** 2) * d
    l_h_ = l_h
    cf = weighted_cost(V, u, alpha, D)
    for _ in range(n_iter1):
        cf_0 = cf
        for _i in range(n_iter2):
            a0 = a1
            a1, beta_w = momentum_step(a0, l_w_, l_w)
            u_temp = u + beta_w * (u - u_)
            u_ = u
            u = np.clip((u_temp + (D * ((V - u @ alpha)) @ alpha.T) / l_w), 0, 1)
            l_w_ = l_w
        l_h = (np.linalg.norm(u) ** 2) * d
        for _j in range(n_iter2):
            a0 = a2
            a2, beta_h = momentum_step(a0, l_h_, l_h)
            alpha_temp = alpha + beta_h * (alpha - alpha_)
            alpha_ = alpha
            alpha = project(alpha_temp + (u.T @ (D * (V - u @ alpha_temp))) / l_h)
            l_h_ = l_h
        l_w = (np.linalg.norm(alpha[-n_u:]) ** 2) * d
        cf = weighted_cost(V, u, alpha, D)
        if trace is not None:
            trace.append(cf)
        if abs(cf - cf_0) < tol:
            break
    return u, alpha


# ---------------------------------------------------------------- drivers' arithmetic

def bootstrap_seeds(seed, n_bootstrap):
    """demethify/bootstrap.py:27 — ``seed = seed + i`` is CUMULATIVE: seed_i = seed_0 + i(i+1)/2."""
    out = []
    s = seed
    for i in range(n_bootstrap):
        s = s + i if s is not None else None
        out.append(s)
    return out


def bootstrap_indices(seed, n_rows):
    """demethify/bootstrap.py:28 — ``sklearn.utils.resample(..., random_state=seed)`` with
    replace=True draws ``RandomState(seed).randint(0, N, size=(N,))`` and applies it to every array."""
    return np.random.RandomState(seed).randint(0, n_rows, size=(n_rows,))


def _n_params(n_u, n_cpg, n_ct, n_samples):
    return n_u * n_cpg + (n_ct + n_u - 1) * n_samples


def bic_as_coded(cost, n_u, n_cpg, n_ct, n_samples):
    """demethify/ic.py:11-15 — the formula exactly as coded upstream."""
    l = n_samples * n_cpg
    k = _n_params(n_u, n_cpg, n_ct, n_samples)
    return 2 * np.log(cost) * k * np.log(l) + (k * np.log(l) * (k + 1)) / (l - k - 1)


def aic_as_coded(cost, n_u, n_cpg, n_ct, n_samples):
    """demethify/ic.py:18-22."""
    l = n_samples * n_cpg
    k = _n_params(n_u, n_cpg, n_ct, n_samples)
    return l * np.log(cost / l) + 2 * k + (2 * k * (k + 1)) / (l - k - 1)


# ---------------------------------------------------------------- synthetic workload

def synthetic_problem(N, S, n_c, n_u, seed=0, depth=50):
    """SURVEY.md section 8(d) generator (recipe of test/gen_data.ipynb cell 5 /
    test/gen_bedmethyl.py:5-20): Beta(.5,.5) profiles, Dirichlet proportions, Poisson depth,
    Binomial counts.  Returns (V f64, D int64, Rt f64)."""
    rs = np.random.RandomState(seed)
    K = n_c + n_u
    Rfull = rs.beta(0.5, 0.5, size=(N, K))
    A = rs.dirichlet(np.ones(K), S).T
    D = rs.poisson(depth, (N, S)) + 1
    X = rs.binomial(D, np.clip(Rfull @ A, 0, 1))
    V = X / D
    return V, D.astype(np.int64), np.ascontiguousarray(Rfull[:, :n_c])


# ---------------------------------------------------------------- purity-constrained variant

def init_partial_purity(init_option, V, D, Rt, n_u, purity, seed=None):
    """demethify/deconvolution.py:228-267 (`init_BSSMF_md_p`), options uniform_/beta/uniform: as
    init_partial but without the first-unknown-row guard (:265-267 returns right after np.c_)."""
    set_seed(seed)
    S = V.shape[1]
    N, n_c = Rt.shape
    if init_option != "uniform" and n_u > S:
        print("The number of unknowns is greater than the number of samples, we'll go with a uniform initialisation. ")
        init_option = "uniform"
    if init_option != "uniform_" and n_u > S:
        init_option = "uniform_"
    if init_option == "uniform":
        u = rd.uniform(size=(N, n_u))
        full = np.c_[Rt, u]
        alpha = np.concatenate(
            [nnls_intercept_proportions(V[:, k:k + 1], D[:, k:k + 1], full) for k in range(S)], axis=1)
    elif init_option == "uniform_":
        u = rd.uniform(size=(N, n_u))
        alpha = rd.dirichlet(np.ones(n_c + n_u), S).T
    elif init_option == "beta":
        half = np.ones((N, n_u)) * 0.5
        u = rd.beta(half, half)
        alpha = rd.dirichlet(np.ones(n_c + n_u), S).T
    else:
        raise NotImplementedError(f"init option {init_option!r} is outside the oracle's scope")
    return u, np.c_[Rt, u], alpha


def frank_wolfe_alpha(W1, W2, V, alpha1, alpha2, purity, max_iter, D):
    """demethify/deconvolution.py:280-302 (`frank_wolfe_nmf`): the known block of every column keeps mass
    purity[col], the unknown block mass 1 - purity[col]; step 2 / (k + 2) towards the best vertex."""
    alpha1 = alpha1.copy()
    alpha2 = alpha2.copy()
    cols = np.arange(alpha1.shape[1])
    for k in range(max_iter):
        resid = D * (V - W1 @ alpha1 - W2 @ alpha2)
        grad1 = -W1.T @ resid
        grad2 = -W2.T @ resid
        s1 = np.zeros_like(alpha1)
        s2 = np.zeros_like(alpha2)
        s1[np.argmin(grad1, axis=0), cols] = purity
        s2[np.argmin(grad2, axis=0), cols] = 1 - purity
        gamma = 2 / (k + 2)
        alpha1 = (1 - gamma) * alpha1 + gamma * s1
        alpha2 = (1 - gamma) * alpha2 + gamma * s2
    return alpha1, alpha2


def solve_partial_purity(u, R, alpha, V, D, Rt, n_u, purity, n_iter1=100, n_iter2=500, tol=1e-3, trace=None):
    """demethify/deconvolution.py:306-337 (`mdwbssmf_deconv_p`)."""
    a1 = 1.0
    u_ = u.copy()
    alpha1, alpha2 = alpha[:-n_u], alpha[-n_u:]
    d = D.max() ** 2
    l_w = (np.linalg.norm(alpha2) ** 2) * d
    l_w_ = l_w
    cf = weighted_cost(V, R, alpha, D)
    for _ in range(n_iter1):
        cf_0 = cf
        u, u_, a1, l_w_ = u_phase(u, alpha, n_iter2, a1, l_w_, l_w, u_, V, Rt, n_u, D)
        R = np.hstack((Rt, u.reshape(-1, n_u)))
        alpha1, alpha2 = frank_wolfe_alpha(Rt, u, V, alpha1, alpha2, purity, n_iter2, D)
        l_w = (np.linalg.norm(alpha2) ** 2) * d
        alpha = np.vstack((alpha1, alpha2))
        cf = weighted_cost(V, R, alpha, D)
        if trace is not None:
            trace.append(cf)
        if abs(cf - cf_0) < tol:
            break
    return u, alpha
