"""Reference-based regression used by the initialisers and by ``--nbunknown 0``.

Host-side only (runs once, milliseconds).  The reference calls
scikit-learn's ``LinearRegression(fit_intercept=True, positive=True)`` with sample weights
(demethify/init_func.py:8-14); that estimator centres by the weighted means, rescales by
sqrt(weight) and solves scipy's NNLS, which is what is done here directly.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import nnls

__all__ = ["wls_intercept"]


def wls_intercept(x, d_x, R_full):
    """Weighted non-negative least squares with intercept, renormalised to proportions.

    Same arguments and return value as the reference's ``wls_intercept``: x (N,) or (N, 1)
    targets, d_x weights, R_full (N, K) profiles -> (K, 1) proportions summing to 1.
    """
    weights = np.asarray(d_x, dtype=np.float64).ravel()
    profiles = np.asarray(R_full, dtype=np.float64)
    target = np.asarray(x, dtype=np.float64)
    one_dim = target.ndim == 1
    target = target.reshape(profiles.shape[0], -1)
    root_w = np.sqrt(weights)[:, None]
    centred_profiles = (profiles - np.average(profiles, axis=0, weights=weights)) * root_w
    centred_target = (target - np.average(target, axis=0, weights=weights)) * root_w
    coef = np.stack([nnls(centred_profiles, centred_target[:, j])[0]
                     for j in range(centred_target.shape[1])])
    if one_dim:
        coef = coef[0]
    temp = coef.T
    return temp / max(temp.sum(), 1e-10)
