"""Edge cases of the device path against the oracle: degenerate and ragged shapes, size limits,
weights that defeat the fast paths' preconditions."""
import numpy as np
import pytest

from oracle import solver as osol

from conftest import rel_err

pytestmark = pytest.mark.gpu
TIGHT = 1e-8


def _run_both(V, D, Rt, n_u, T1, T2, tol, seed=1):
    from demethify_amd import deconvolution as dd

    if Rt is not None:
        u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, n_u, seed=seed)
        wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, n_u, T1, T2, tol,
                                    project=osol.simplex_project_columns_fast)
        gu, ga = dd.mdwbssmf_deconv(u0, R, a0, V, D, Rt, n_u, n_iter1=T1, n_iter2=T2, tol=tol)
    else:
        u0, a0 = osol.init_unsupervised("uniform_", V, n_u, seed=seed)
        wu, wa = osol.solve_unsupervised(V, n_u, D, "uniform_", T1, T2, tol, init=(u0.copy(), a0.copy()),
                                         project=osol.simplex_project_columns_fast)
        gu, ga = dd.unsupervised_deconv(V, n_u, D, "uniform_", n_iter1=T1, n_iter2=T2, tol=tol, seed=seed)
    return (gu, ga), (wu, wa)


@pytest.mark.parametrize("N,S,n_c,n_u", [(40, 1, 3, 1), (5, 4, 2, 1), (15, 8, 2, 2), (17, 8, 2, 2), (33, 2, 0, 2),
                                         (64, 3, 1, 1), (100, 260, 4, 2), (50, 300, 0, 3)])
def test_small_and_ragged_shapes(N, S, n_c, n_u):
    """single sample, fewer than 16 CpG rows, a ragged 16-row tail, S beyond the fused kernel's limit"""
    V, D, Rt = osol.synthetic_problem(N, S, max(n_c, 1), n_u, seed=3, depth=12)
    (gu, ga), (wu, wa) = _run_both(V, D, Rt if n_c else None, n_u, 5, 20, 0.0)
    assert np.abs(ga - wa).max() < TIGHT and np.abs(gu - wu).max() < TIGHT


def test_zero_outer_iterations_returns_the_initial_point():
    V, D, Rt = osol.synthetic_problem(64, 8, 3, 2, seed=4, depth=12)
    (gu, ga), (wu, wa) = _run_both(V, D, Rt, 2, 0, 20, 1e-2)
    u0, _, a0 = osol.init_partial("uniform_", V, D, Rt, 2, seed=1)
    assert np.array_equal(gu, u0) and np.array_equal(ga, a0) and np.array_equal(wu, u0)


@pytest.mark.parametrize("T2", [1, 3, 64, 70])
def test_inner_iteration_counts(T2):
    """T2 = 1; more inner steps than the 64 momentum coefficients one VGPR holds"""
    V, D, Rt = osol.synthetic_problem(160, 16, 4, 2, seed=5, depth=12)
    (gu, ga), (wu, wa) = _run_both(V, D, Rt, 2, 3, T2, 0.0)
    assert np.abs(ga - wa).max() < TIGHT and np.abs(gu - wu).max() < TIGHT


@pytest.mark.parametrize("T2", [7000, 20000])
def test_very_many_inner_iterations(T2):
    """The reference runs any n_iter2; the momentum table of the inner steps passes through LDS in chunks of 6144
    (k_u_inner_rows), so there is no size at which the device path starts to refuse."""
    V, D, Rt = osol.synthetic_problem(48, 6, 3, 2, seed=8, depth=12)
    (gu, ga), (wu, wa) = _run_both(V, D, Rt, 2, 1, T2, 0.0)
    assert np.abs(ga - wa).max() < TIGHT and np.abs(gu - wu).max() < TIGHT


def test_zero_coverage_entries_and_fractional_weights():
    """coverage 0 (what --fillna produces) and non-integer weights (not exact in f32: the fused tile format
    must be refused and the f64 kernels used)"""
    rs = np.random.RandomState(6)
    V, D, Rt = osol.synthetic_problem(200, 16, 4, 2, seed=6, depth=12)
    D0 = D.copy()
    D0[rs.rand(*D.shape) < 0.2] = 0
    V0 = np.where(D0 == 0, 0.0, V)
    (gu, ga), (wu, wa) = _run_both(V0, D0, Rt, 2, 4, 20, 0.0)
    assert np.abs(ga - wa).max() < TIGHT and np.abs(gu - wu).max() < TIGHT
    Df = D.astype(np.float64) + rs.rand(*D.shape) / 3.0
    (gu, ga), (wu, wa) = _run_both(V, Df, Rt, 2, 4, 20, 0.0)
    assert np.abs(ga - wa).max() < TIGHT and np.abs(gu - wu).max() < TIGHT


def test_counts_beyond_f32_exact_range():
    V, D, Rt = osol.synthetic_problem(128, 8, 3, 1, seed=7, depth=12)
    Dbig = D.astype(np.int64) * 3_000_001 + 1  # > 2^24, odd: not representable in f32
    (gu, ga), (wu, wa) = _run_both(V, Dbig, Rt, 1, 3, 20, 0.0)
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT


def test_largest_supported_rank_and_beyond(ctx):
    from demethify_amd._lib import DemethifyHipError
    from demethify_amd.device import Problem, Solver

    V, D, Rt = osol.synthetic_problem(300, 6, 60, 4, seed=8, depth=12)  # K = 64: the alpha kernels' limit
    (gu, ga), (wu, wa) = _run_both(V, D, Rt, 4, 2, 5, 0.0)
    assert np.abs(ga - wa).max() < TIGHT and np.abs(gu - wu).max() < TIGHT
    rs = np.random.RandomState(1)
    with Problem(ctx, V, D, Rt) as p:
        with pytest.raises(DemethifyHipError) as err:
            Solver(p, rs.uniform(size=(300, 5)), rs.dirichlet(np.ones(65), 6).T)
        assert err.value.status == 5  # DMF_ERR_UNSUPPORTED


def test_natural_stop_matches_on_tail_and_wide_problems():
    V, D, Rt = osol.synthetic_problem(1001, 36, 4, 2, seed=9, depth=20)
    u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, 2, seed=2)
    trace = []
    wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, 2, 300, 20, 1e-2, trace=trace,
                                project=osol.simplex_project_columns_fast)
    from demethify_amd import _lib as L
    from demethify_amd.deconvolution import solve_problem
    from demethify_amd.device import Problem, get_context

    with Problem(get_context(), V, D, Rt) as p:
        gu, ga, cost, iters = solve_problem(p, u0, a0, L.DMF_MODE_PARTIAL, 300, 20, 1e-2, return_info=True)
    assert iters == len(trace) < 300
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT
