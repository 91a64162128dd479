"""End-to-end parity of the device-resident outer loop against the committed upstream outputs and
the CPU oracle.  Bar (BASELINE.json north_star): proportions within 1e-5 relative of the CPU path
for a fixed seed; asserted much tighter here because the f64 Gram schedule tracks to ~1e-10."""
import numpy as np
import pytest

from oracle import solver as osol

from conftest import PARITY_RTOL, read_profile, read_props, rel_err

pytestmark = pytest.mark.gpu

TIGHT = 1e-8  # what we hold ourselves to; PARITY_RTOL (1e-5) is the contractual bar
assert TIGHT < PARITY_RTOL


def test_partial_reference_golden_folder(toy):
    """README command: --ref ref_matrix.bed --nbunknown 1 (test/output_partial_ref), seed 1."""
    from demethify_amd import deconvolution as dd

    V, D, ref, _ = toy
    u, R, alpha = dd.init_BSSMF_md("uniform_", V, D, ref, 1, seed=1)
    u, alpha = dd.mdwbssmf_deconv(u, R, alpha, V, D, ref, 1, n_iter1=10000, n_iter2=20, tol=1e-2)
    want = read_props("output_partial_ref")
    assert rel_err(alpha, want) < TIGHT and np.abs(alpha - want).max() < TIGHT
    assert np.abs(u - read_profile("output_partial_ref")).max() < TIGHT


def test_unsupervised_golden_folder(toy):
    """README command: --nbunknown 4 without --ref (test/unsupervised), seed 1."""
    from demethify_amd import deconvolution as dd

    V, D, _, _ = toy
    u, alpha = dd.unsupervised_deconv(V, 4, D, "uniform_", n_iter1=10000, n_iter2=20, tol=1e-2, seed=1)
    want = read_props("unsupervised")
    assert rel_err(alpha, want) < TIGHT and np.abs(alpha - want).max() < TIGHT
    assert np.abs(u - read_profile("unsupervised")).max() < TIGHT


def test_stop_iteration_matches_oracle(toy, ctx):
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem
    from demethify_amd.deconvolution import solve_problem

    V, D, ref, _ = toy
    u0, R, a0 = osol.init_partial("uniform_", V, D, ref, 1, seed=1)
    trace = []
    osol.solve_partial(u0.copy(), R, a0.copy(), V, D, ref, 1, 10000, 20, 1e-2, trace=trace)
    with Problem(ctx, V, D, ref) as p:
        u, alpha, cost, iters = solve_problem(p, u0, a0, L.DMF_MODE_PARTIAL, 10000, 20, 1e-2, return_info=True)
    assert iters == len(trace) == 54
    assert cost == pytest.approx(trace[-1], rel=1e-9)


CASES = [  # (N, S, n_c, n_u, T1)
    (4096, 7, 6, 1, 6), (4096, 64, 6, 2, 6), (4096, 100, 6, 4, 5), (3000, 33, 0, 2, 6), (2500, 64, 0, 4, 5),
    (1111, 130, 12, 4, 4), (900, 20, 3, 8, 4), (700, 12, 0, 12, 3), (600, 16, 2, 16, 3),
]


@pytest.mark.parametrize("generic", [0, 1, 2, 3])
@pytest.mark.parametrize("N,S,n_c,n_u,T1", CASES)
def test_fixed_iteration_parity(ctx, N, S, n_c, n_u, T1, generic):
    """tol = 0 never satisfies the stop test: exactly T1 outer iterations on both sides."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem
    from demethify_amd.deconvolution import solve_problem

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=11, depth=30)
    if n_c:
        u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, n_u, seed=1)
        wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, n_u, T1, 20, 0.0,
                                    project=osol.simplex_project_columns_fast)
        mode = L.DMF_MODE_PARTIAL
    else:
        u0, a0 = osol.init_unsupervised("uniform_", V, n_u, seed=1)
        wu, wa = osol.solve_unsupervised(V, n_u, D, "uniform_", T1, 20, 0.0, init=(u0.copy(), a0.copy()),
                                         project=osol.simplex_project_columns_fast)
        mode = L.DMF_MODE_UNSUPERVISED
    ctx.set_generic(generic)
    try:
        with Problem(ctx, V, D, Rt if n_c else None) as p:
            u, alpha, cost, iters = solve_problem(p, u0, a0, mode, T1, 20, 0.0, return_info=True)
    finally:
        ctx.set_generic(0)
    assert iters == T1
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT
    assert np.abs(u - wu).max() < TIGHT
    R_final = np.c_[Rt, wu] if n_c else wu
    assert cost == pytest.approx(osol.weighted_cost(V, R_final, wa, D), rel=1e-9)


BIG_CASES = [  # (N, S, n_c, n_u, T1): 9 <= n_u <= 26 runs the matrix-core u phase with M_i in LDS + the MFMA Gram
    (650, 24, 0, 9, 3), (1000, 100, 0, 13, 2), (530, 37, 12, 9, 3), (777, 64, 5, 16, 2), (333, 130, 0, 17, 2),
    (600, 48, 3, 20, 2), (512, 20, 0, 25, 2), (500, 128, 16, 26, 2), (400, 30, 1, 27, 2),
]


@pytest.mark.parametrize("generic", [0, 3])
@pytest.mark.parametrize("N,S,n_c,n_u,T1", BIG_CASES)
def test_many_unknown_types_parity(ctx, N, S, n_c, n_u, T1, generic):
    """The --ic sweep goes to n_u = 25 (ic.py:171): same parity bar as the small shapes, ragged N and S included;
    n_u = 27 falls back to the schedule-faithful u steps."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem
    from demethify_amd.deconvolution import solve_problem

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=21, depth=30)
    if n_c:
        u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, n_u, seed=2)
        wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, n_u, T1, 20, 0.0,
                                    project=osol.simplex_project_columns_fast)
        mode = L.DMF_MODE_PARTIAL
    else:
        u0, a0 = osol.init_unsupervised("uniform_", V, n_u, seed=2)
        wu, wa = osol.solve_unsupervised(V, n_u, D, "uniform_", T1, 20, 0.0, init=(u0.copy(), a0.copy()),
                                         project=osol.simplex_project_columns_fast)
        mode = L.DMF_MODE_UNSUPERVISED
    ctx.set_generic(generic)
    try:
        with Problem(ctx, V, D, Rt if n_c else None) as p:
            u, alpha, cost, iters = solve_problem(p, u0, a0, mode, T1, 20, 0.0, return_info=True)
    finally:
        ctx.set_generic(0)
    assert iters == T1
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT
    assert np.abs(u - wu).max() < TIGHT
    R_final = np.c_[Rt, wu] if n_c else wu
    assert cost == pytest.approx(osol.weighted_cost(V, R_final, wa, D), rel=1e-9)


def test_natural_stop_synthetic(ctx):
    from demethify_amd import deconvolution as dd

    V, D, Rt = osol.synthetic_problem(2000, 24, 5, 2, seed=5, depth=25)
    u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, 2, seed=3)
    trace = []
    wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, 2, 400, 20, 1e-2, trace=trace,
                                project=osol.simplex_project_columns_fast)
    u, R2, a = dd.init_BSSMF_md("uniform_", V, D, Rt, 2, seed=3)
    assert np.array_equal(u, u0) and np.array_equal(a, a0)
    gu, ga = dd.mdwbssmf_deconv(u, R2, a, V, D, Rt, 2, n_iter1=400, n_iter2=20, tol=1e-2)
    assert len(trace) < 400
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT


def test_many_unknowns_uses_fallback_path(ctx):
    """n_u > 8 through the reference-named entry point (K = 5 + 12: runtime-K alpha kernel)."""
    from demethify_amd import deconvolution as dd

    V, D, Rt = osol.synthetic_problem(800, 14, 5, 12, seed=9, depth=25)
    u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, 12, seed=2)
    wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, 12, 3, 20, 0.0,
                                project=osol.simplex_project_columns_fast)
    gu, ga = dd.mdwbssmf_deconv(u0, R, a0, V, D, Rt, 12, n_iter1=3, n_iter2=20, tol=0.0)
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT


@pytest.mark.parametrize("generic", [0, 1, 3])
def test_purity_constrained_solver(ctx, generic):
    """mdwbssmf_deconv_p: same u phase, Frank-Wolfe alpha phase; fixed iteration count on synthetic data."""
    from demethify_amd import deconvolution as dd

    V, D, Rt = osol.synthetic_problem(1500, 12, 4, 2, seed=13, depth=25)
    purity = np.linspace(0.2, 0.9, 12)
    u0, R, a0 = osol.init_partial_purity("uniform_", V, D, Rt, 2, purity, seed=5)
    wu, wa = osol.solve_partial_purity(u0.copy(), R, a0.copy(), V, D, Rt, 2, purity, 4, 30, 0.0)
    gu0, gR, ga0 = dd.init_BSSMF_md_p("uniform_", V, D, Rt, 2, purity, seed=5)
    assert np.array_equal(gu0, u0) and np.array_equal(ga0, a0)
    ctx.set_generic(generic)
    try:
        gu, ga = dd.mdwbssmf_deconv_p(gu0, gR, ga0, V, D, Rt, 2, purity, n_iter1=4, n_iter2=30, tol=0.0)
    finally:
        ctx.set_generic(0)
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT
    assert np.allclose(ga[:4].sum(axis=0), purity, atol=1e-12) and np.allclose(ga[4:].sum(axis=0), 1 - purity, atol=1e-12)


def test_inputs_are_not_mutated(toy):
    from demethify_amd import deconvolution as dd

    V, D, ref, _ = toy
    u, R, alpha = dd.init_BSSMF_md("uniform_", V, D, ref, 1, seed=1)
    keep = [x.copy() for x in (u, R, alpha, V, D, ref)]
    dd.mdwbssmf_deconv(u, R, alpha, V, D, ref, 1, n_iter1=3, n_iter2=20, tol=1e-2)
    for a, b in zip((u, R, alpha, V, D, ref), keep):
        assert np.array_equal(a, b)


def test_step_is_resumable(ctx):
    """step(a) then step(b) equals step(a+b): the momentum state lives on the device."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    V, D, Rt = osol.synthetic_problem(1500, 16, 4, 2, seed=21, depth=25)
    u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, 2, seed=1)
    with Problem(ctx, V, D, Rt) as p:
        with Solver(p, u0, a0) as s:
            s.step(3, 20, 0.0)
            it, conv = s.step(4, 20, 0.0)
            u1, a1, c1, _ = s.get()
        with Solver(p, u0, a0) as s:
            s.step(7, 20, 0.0)
            u2, a2, c2, _ = s.get()
    assert it == 7 and not conv
    assert np.array_equal(u1, u2) and np.array_equal(a1, a2) and c1 == c2


def test_staged_initialisation_is_the_same_solve(toy, ctx):
    """A restart's (u0, alpha0) uploaded ahead of time (staging.to_device, the restart loop's worker thread) and handed
    to the solver as device arrays gives bit-identical iterates to the host-array call."""
    from demethify_amd import _lib as L
    from demethify_amd import staging
    from demethify_amd.device import Problem, Solver

    V, D, ref, _ = toy
    u0, R, a0 = osol.init_partial("uniform_", V, D, ref, 2, seed=3)
    with Problem(ctx, V, D, ref) as p:
        with Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
            s.step(7, 20, 0.0)
            c_host = s.direct_cost()
            u_h, a_h, _, _ = s.get()
        u0_d, a0_d = staging.to_device((u0, a0), ctx)
        assert u0_d.is_cuda and a0_d.is_cuda
        with Solver(p, u0_d, a0_d, L.DMF_MODE_PARTIAL) as s:
            s.step(7, 20, 0.0)
            c_dev = s.direct_cost()
            u_d, a_d, _, _ = s.get()
        with pytest.raises(ValueError):
            Solver(p, u0_d, a0, L.DMF_MODE_PARTIAL)
    assert c_host == c_dev
    np.testing.assert_array_equal(u_h, u_d)
    np.testing.assert_array_equal(a_h, a_d)


def test_cost_in_two_halves_equals_direct_cost(ctx, toy):
    """dmf_solver_cost_begin / _end (the restart loops take a restart's cost_f_w while they set up the next one): the same
    bits as dmf_solver_cost, also with another solver created, stepped and destroyed in between, and a solver may be
    destroyed with its cost still on its way."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    V, D, ref, _ = toy
    u0, R, a0 = osol.init_partial("uniform_", V, D, ref, 2, seed=3)
    with Problem(ctx, V, D, ref) as p:
        with Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
            s.step(3, 20, 0.0)
            want = s.direct_cost()
            s.cost_begin()
            with Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as other:
                other.step(2, 20, 0.0)
            assert s.cost_end() == want
            with pytest.raises(L.DemethifyHipError):
                s.cost_end()  # nothing on its way
            s.cost_begin()
            assert s.cost_end() == want
        s2 = Solver(p, u0, a0, L.DMF_MODE_PARTIAL)
        s2.step(1, 20, 0.0)
        s2.cost_begin()
        s2.close()
        with Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s3:
            s3.step(3, 20, 0.0)
            assert s3.direct_cost() == want
