"""Pins the CPU oracle to the outputs the reference's authors committed under test/
(SURVEY.md section 4 / 8c).  The oracle is only trusted because these pass."""
import numpy as np
import pytest

from oracle import drivers as odrv
from oracle import solver as osol

from conftest import UPSTREAM, read_profile, read_props
import pandas as pd


def test_partial_reference_folder(toy):
    V, D, ref, _ = toy
    u, R, alpha = osol.init_partial("uniform_", V, D, ref, 1, seed=1)
    trace = []
    u, alpha = osol.solve_partial(u, R, alpha, V, D, ref, 1, n_iter1=10000, n_iter2=20, tol=1e-2, trace=trace)
    assert len(trace) == 54
    assert np.abs(alpha - read_props("output_partial_ref")).max() < 1e-11
    assert np.abs(u - read_profile("output_partial_ref")).max() < 1e-11


def test_unsupervised_folder(toy):
    V, D, _, _ = toy
    trace = []
    u, alpha = osol.solve_unsupervised(V, 4, D, "uniform_", n_iter1=10000, n_iter2=20, tol=1e-2, seed=1,
                                       trace=trace)
    assert len(trace) == 155
    assert np.abs(alpha - read_props("unsupervised")).max() < 1e-11
    assert np.abs(u - read_profile("unsupervised")).max() < 1e-11


def test_purity_folder(toy):
    """README purity command: --purity 60 80 90 20 50 90 100 30 50 10 -> 1 - p/100 (demethify.py:77)."""
    V, D, ref, _ = toy
    purity = 1 - np.array([60, 80, 90, 20, 50, 90, 100, 30, 50, 10]) / 100.0
    u, R, alpha = osol.init_partial_purity("uniform_", V, D, ref, 1, purity, seed=1)
    trace = []
    u, alpha = osol.solve_partial_purity(u, R, alpha, V, D, ref, 1, purity, 100, 500, 1e-2, trace=trace)
    assert len(trace) == 7
    assert np.abs(alpha - read_props("purity")).max() < 1e-12
    assert np.abs(u - read_profile("purity")).max() < 1e-12


def test_reference_based_folder(toy):
    V, D, ref, _ = toy
    alpha = np.concatenate(
        [osol.nnls_intercept_proportions(D[:, k:k + 1] * V[:, k:k + 1], D[:, k:k + 1], ref) for k in range(10)],
        axis=1)
    assert np.abs(alpha - read_props("output_ref_based")).max() < 1e-13


def test_config1_plumbing_kat():
    """BASELINE.json configs[0]: 6 reference types, one sample, --nbunknown 0."""
    r = pd.read_csv(UPSTREAM / "config1" / "bed1_select_ref_intersect.bed", sep="\t").iloc[:, 3:].values[:, :6]
    s = pd.read_csv(UPSTREAM / "config1" / "bed2_intersect.bed", sep="\t")
    v = (s["percent_modified"].values / 100)[:, None]
    c = s["valid_coverage"].values[:, None]
    got = osol.nnls_intercept_proportions(c * v, c, r).ravel()
    want = np.array([0, 0, 0.03548921, 0, 0, 0.96451079])
    assert np.abs(got - want).max() < 5e-9
    assert abs(got.sum() - 1) < 1e-12


def test_model_selection_folder(toy):
    """AIC sweep n_u = 1..25 (ic.py:171) picks 10 unknowns, as test/model_selection/log.log records."""
    V, D, ref, _ = toy
    # the vectorised projection is arithmetically identical per column (test below) and 30x faster
    u, alpha, best, scores = odrv.ic_sweep(V, ref, D, "uniform_", "AIC", 1, 10000, 20, 1e-2,
                                           project=osol.simplex_project_columns_fast)
    assert best == 10 and len(scores) == 25
    assert "AIC : 10" in (UPSTREAM / "model_selection" / "log.log").read_text()
    assert np.abs(alpha - read_props("model_selection")).max() < 1e-10
    assert np.abs(u - read_profile("model_selection")).max() < 1e-10


def test_fast_projection_is_identical():
    rs = np.random.RandomState(3)
    for K, S in ((1, 5), (2, 9), (7, 50), (16, 33), (30, 4)):
        X = rs.randn(K, S) * rs.choice([0.1, 1, 10])
        assert np.array_equal(osol.simplex_project_columns(X), osol.simplex_project_columns_fast(X))
    X = np.zeros((4, 3))
    assert np.array_equal(osol.simplex_project_columns(X), osol.simplex_project_columns_fast(X))


def test_bootstrap_resampling_arithmetic():
    """bootstrap.py:27-28: cumulative seeds; sklearn's resample == RandomState(seed).randint rows."""
    assert osol.bootstrap_seeds(1, 6) == [1, 2, 4, 7, 11, 16]
    sk = pytest.importorskip("sklearn.utils")
    X = np.arange(700).reshape(350, 2)
    y = np.arange(350)
    a, b = sk.resample(X, y, random_state=7)
    idx = osol.bootstrap_indices(7, 350)
    assert np.array_equal(a, X[idx]) and np.array_equal(b, y[idx])


def test_information_criteria_as_coded():
    # hand evaluation of ic.py:11-22 for one point
    cost, n_u, n_cpg, n_ct, n_s = 123.4, 2, 350, 5, 10
    l, k = n_s * n_cpg, n_u * n_cpg + (n_ct + n_u - 1) * n_s
    assert osol.aic_as_coded(cost, n_u, n_cpg, n_ct, n_s) == l * np.log(cost / l) + 2 * k + (2 * k * (k + 1)) / (l - k - 1)
    assert osol.bic_as_coded(cost, n_u, n_cpg, n_ct, n_s) == 2 * np.log(cost) * k * np.log(l) + (k * np.log(l) * (k + 1)) / (l - k - 1)
