"""Host-side mirror of the reference's solver interface (demethify/deconvolution.py).

Same names, argument order and return values as the reference so that callers written against
``from .deconvolution import *`` (demethify/demethify.py:7, bootstrap.py:6, ic.py:8) keep
working; the arithmetic runs in the HIP kernels behind the C-ABI (include/demethify_hip.h).
Arguments are host numpy arrays and are never mutated; results are fresh arrays.  The random
initialisation stays on the host because parity needs numpy's legacy global MT19937 stream in
the reference's call order (deconvolution.py:55-56).

There is no CPU fallback: every function here raises if the HIP library or the GPU is missing.
"""
from __future__ import annotations

import numpy as np
import numpy.random as rd

from . import _lib as L
from .device import Problem, Solver, get_context
from .init_func import wls_intercept

__all__ = [
    "set_seed", "cost_f_w", "projection_simplex_sort_2d", "init_BSSMF_md", "update_u", "update_alpha",
    "unsupervised_deconv", "mdwbssmf_deconv", "wls_intercept", "solve_problem", "init_BSSMF_md_p",
    "mdwbssmf_deconv_p",
]

_OUT_OF_SCOPE_INITS = ("ICA", "SVD")


def set_seed(seed=None):
    """deconvolution.py:9-11.  ``seed`` may be an int or the 1-element list the CLI produces."""
    if seed is not None:
        rd.seed(seed)


def cost_f_w(y, R, alpha, d_x):
    """deconvolution.py:15-17: ``||sqrt(d_x) * (y - R @ alpha)||_F^2`` on the GPU."""
    with Problem(get_context(), y, d_x, np.asarray(R, dtype=np.float64).reshape(np.shape(y)[0], -1)) as p:
        return p.cost(None, alpha)


def projection_simplex_sort_2d(v, z=1):
    """deconvolution.py:21-37: column-wise projection onto the simplex of mass z."""
    return get_context().project_simplex(v, z)


def _init_guard(alpha, n_u):
    # deconvolution.py:74-76, as coded: triggers when ANY entry of the first unknown row is zero
    if alpha[-n_u:][0].all() == 0.0:
        alpha[-n_u:][0] = 1e-10
        alpha[:-n_u] = (1 - 1e-10) * alpha[:-n_u]
    return alpha


def init_BSSMF_md(init_option, meth_frequency, d_x, R_trunc, n_u, seed=None, rb_alg=wls_intercept, _stack=True):
    """deconvolution.py:40-78 -> (u, R, alpha).  Host-side (RNG stream parity).  ``_stack=False`` (the restart, bootstrap
    and model-selection loops of this package, which only need u and alpha) returns R = None instead of copying
    N x (n_c + n_u) doubles per call -- 25 ms at 1e6 rows, more than the draws themselves."""
    set_seed(seed)
    nb = meth_frequency.shape[1]
    n_rows, n_c = R_trunc.shape
    if init_option != "uniform_" and n_u > nb:
        init_option = "uniform_"
    if init_option in _OUT_OF_SCOPE_INITS:
        raise NotImplementedError(
            f"--init {init_option} (one-shot LAPACK initialiser, demethify/init_func.py) is not part of "
            "this build; use uniform_, uniform or beta")
    if init_option == "uniform":
        u = rd.uniform(size=(n_rows, n_u))
        stacked = np.c_[R_trunc, u]
        alpha = np.concatenate(
            [rb_alg(meth_frequency[:, k:k + 1], d_x[:, k:k + 1], stacked) for k in range(nb)], axis=1)
    elif init_option == "uniform_":
        u = rd.uniform(size=(n_rows, n_u))
        alpha = rd.dirichlet(np.ones(n_c + n_u), nb).T
    elif init_option == "beta":
        shape = np.ones((n_rows, n_u)) * 0.5
        u = rd.beta(shape, shape)
        alpha = rd.dirichlet(np.ones(n_c + n_u), nb).T
    else:
        raise UnboundLocalError(f"unknown init option {init_option!r}")  # upstream: u is never bound
    R = np.c_[R_trunc, u] if _stack else None
    alpha = _init_guard(alpha, n_u)
    return u, R, alpha


def update_u(u, alpha, n_iter2, a1, l_w_, l_w, u_, meth_frequency, R_trunc, n_u, d_x):
    """deconvolution.py:81-90 -> (u, u_, a1, l_w_)."""
    with Problem(get_context(), meth_frequency, d_x, R_trunc) as p:
        return p.update_u(np.asarray(u).reshape(-1, n_u), np.asarray(u_).reshape(-1, n_u), alpha, n_iter2,
                          a1, l_w_, l_w)


def update_alpha(n_iter2, alpha, a2, l_h_, l_h, alpha_, R, d_x, meth_frequency):
    """deconvolution.py:93-102 -> (alpha, alpha_, a2, l_h_).  R is the full N x K profile matrix;
    the C-ABI wants it as [known | unknown], so its last column is handed over as the unknown part
    (the Gram matrices cover every column either way)."""
    R = np.asarray(R, dtype=np.float64)
    known = np.ascontiguousarray(R[:, :-1]) if R.shape[1] > 1 else None
    last = np.ascontiguousarray(R[:, -1:])
    with Problem(get_context(), meth_frequency, d_x, known) as p:
        return p.update_alpha(last, alpha, alpha_, n_iter2, a2, l_h_, l_h)


def solve_problem(problem: Problem, u0, alpha0, mode, n_iter1, n_iter2, tol, return_info=False, purity=None):
    """Run the outer loop on a device-resident problem -> (u, alpha[, cost, iterations]).
    ``purity`` (per-sample mass of the known block) selects the purity-constrained alpha phase."""
    with Solver(problem, u0, alpha0, mode) as s:
        if purity is not None:
            s.set_purity(purity)
        s.step(n_iter1, n_iter2, tol)
        u, alpha, cost, iters = s.get()
    if return_info:
        return u, alpha, cost, iters
    return u, alpha


def mdwbssmf_deconv(u, R, alpha, meth_frequency, d_x, R_trunc, n_u, n_iter1=100000, n_iter2=50, tol=1e-3):
    """deconvolution.py:190-223 -> (u, alpha).  ``R`` is accepted for signature parity; the device
    rebuilds it from R_trunc and u (as the reference does at :210)."""
    del R
    with Problem(get_context(), meth_frequency, d_x, R_trunc) as p:
        return solve_problem(p, np.asarray(u).reshape(-1, n_u), alpha, L.DMF_MODE_PARTIAL, n_iter1, n_iter2,
                             tol)


def init_BSSMF_md_p(init_option, meth_frequency, d_x, R_trunc, n_u, purity, rb_alg=wls_intercept, seed=None, _stack=True):
    """deconvolution.py:228-267 -> (u, R, alpha): as init_BSSMF_md but without the zero guard on the first
    unknown row (the function returns right after building R); ``purity`` only matters to the SVD / ICA
    initialisers, which are outside this build."""
    del purity
    set_seed(seed)
    nb = meth_frequency.shape[1]
    n_rows, n_c = R_trunc.shape
    if init_option != "uniform" and n_u > nb:
        print("The number of unknowns is greater than the number of samples, we'll go with a uniform initialisation. ")
        init_option = "uniform"
    if init_option != "uniform_" and n_u > nb:
        init_option = "uniform_"
    if init_option in _OUT_OF_SCOPE_INITS:
        raise NotImplementedError(
            f"--init {init_option} (one-shot LAPACK initialiser, demethify/init_func.py) is not part of "
            "this build; use uniform_, uniform or beta")
    if init_option == "uniform":
        u = rd.uniform(size=(n_rows, n_u))
        stacked = np.c_[R_trunc, u]
        alpha = np.concatenate(
            [rb_alg(meth_frequency[:, k:k + 1], d_x[:, k:k + 1], stacked) for k in range(nb)], axis=1)
    elif init_option == "uniform_":
        u = rd.uniform(size=(n_rows, n_u))
        alpha = rd.dirichlet(np.ones(n_c + n_u), nb).T
    elif init_option == "beta":
        shape = np.ones((n_rows, n_u)) * 0.5
        u = rd.beta(shape, shape)
        alpha = rd.dirichlet(np.ones(n_c + n_u), nb).T
    else:
        raise UnboundLocalError(f"unknown init option {init_option!r}")
    return u, (np.c_[R_trunc, u] if _stack else None), alpha


def mdwbssmf_deconv_p(u, R, alpha, meth_frequency, d_x, R_trunc, n_u, purity, n_iter1=100, n_iter2=500, tol=1e-3):
    """deconvolution.py:306-337 -> (u, alpha): u phase as mdwbssmf_deconv, alpha phase = Frank-Wolfe with the
    known block of sample s at mass purity[s] and the unknown block at 1 - purity[s] (:280-302)."""
    del R
    with Problem(get_context(), meth_frequency, d_x, R_trunc) as p:
        return solve_problem(p, np.asarray(u).reshape(-1, n_u), alpha, L.DMF_MODE_PARTIAL, n_iter1, n_iter2,
                             tol, purity=purity)


def _init_unsupervised(init_option, meth_frequency, n_u, seed):
    """deconvolution.py:108-137 (host RNG)."""
    set_seed(seed)
    n_rows, nb = meth_frequency.shape
    if init_option != "uniform_" and n_u > nb:
        init_option = "uniform_"
    if init_option == "uniform":
        # upstream references an undefined name here (deconvolution.py:117); kept, not "fixed"
        raise NameError("name 'R_trunc' is not defined")
    if init_option in _OUT_OF_SCOPE_INITS:
        raise NotImplementedError(
            f"--init {init_option} (one-shot LAPACK initialiser, demethify/init_func.py) is not part of "
            "this build; use uniform_ or beta")
    if init_option == "uniform_":
        u = rd.uniform(size=(n_rows, n_u))
        alpha = rd.dirichlet(np.ones(n_u), nb).T
    elif init_option == "beta":
        shape = np.ones((n_rows, n_u)) * 0.5
        u = rd.beta(shape, shape)
        alpha = rd.dirichlet(np.ones(n_u), nb).T
    else:
        raise UnboundLocalError(f"unknown init option {init_option!r}")
    return u, alpha


def unsupervised_deconv(meth_frequency, n_u, d_x, init_option, n_iter1=100000, n_iter2=20, tol=1e-3, seed=None):
    """deconvolution.py:107-184 -> (u, alpha)."""
    u, alpha = _init_unsupervised(init_option, meth_frequency, n_u, seed)
    with Problem(get_context(), meth_frequency, d_x, None) as p:
        return solve_problem(p, u, alpha, L.DMF_MODE_UNSUPERVISED, n_iter1, n_iter2, tol)
