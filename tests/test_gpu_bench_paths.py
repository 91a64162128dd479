"""Oracle parity of exactly the kernel instantiations bench.py times, and of BASELINE.json's config 5.

Every case first asserts (Solver.describe) that it enters the kernel it is meant to cover, so a change of the
shape rules in dmf_solver_create cannot silently move a case onto another path."""
import numpy as np
import pytest

from oracle import drivers as odrv
from oracle import solver as osol

from conftest import rel_err

pytestmark = pytest.mark.gpu

TIGHT = 1e-8

# (N, S, n_c, n_u, T1, what the case is for)
FUSED_CASES = [
    (4096 + 5, 256, 12, 4, 4, "<3,4> nw=4 (the bench's instantiation), one block per workgroup, ragged 5-row tail"),
    (9600 + 5, 256, 12, 4, 3, "<3,4> nw=4, 2-3 blocks per workgroup: the persistent loop and the tile double buffer"),
    (2048, 192, 16, 3, 4, "NKC = 4 (n_c 13..16), NU = 3, nw = 3"),
    (3000, 128, 0, 4, 4, "unsupervised gradient point, no known types, nw = 2 (two workgroups per CU)"),
    (2048, 256, 5, 1, 4, "NU = 1, n_c not a multiple of 4 (padded R_trunc copy)"),
    (8192 + 9, 64, 6, 2, 4, "config 2's instantiation <2,2> nw=1, several blocks per workgroup"),
]


def _oracle(V, D, Rt, n_c, n_u, T1, seed):
    if n_c:
        u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, n_u, seed=seed)
        wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, n_u, T1, 20, 0.0,
                                    project=osol.simplex_project_columns_fast)
    else:
        u0, a0 = osol.init_unsupervised("uniform_", V, n_u, seed=seed)
        wu, wa = osol.solve_unsupervised(V, n_u, D, "uniform_", T1, 20, 0.0, init=(u0.copy(), a0.copy()),
                                         project=osol.simplex_project_columns_fast)
    return u0, a0, wu, wa


@pytest.mark.parametrize("N,S,n_c,n_u,T1,why", FUSED_CASES)
def test_fused_instantiations_against_oracle(ctx, N, S, n_c, n_u, T1, why):
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=31, depth=40)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=1)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    with Problem(ctx, V, D, Rt if n_c else None) as p, Solver(p, u0, a0, mode) as s:
        path = s.describe(20)
        assert f"<{(n_c + 3) // 4},{n_u}>" in path and f"nw={(S + 63) // 64}" in path and f"tail={N % 16}" in path, path
        assert "fused" in path.split("rowpass=")[1].split()[0], path
        it, _ = s.step(T1, 20, 0.0)
        u, alpha, cost, _ = s.get()
        direct = p.cost(u, alpha)
    assert it == T1
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT, why
    assert np.abs(u - wu).max() < TIGHT, why
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)


def test_headline_columns_at_oracle_size(ctx):
    """The bench's shape in everything but the row count: 256 samples, 12 + 4 types, Poisson(50) depth, enough
    rows (40 000) for ten blocks per workgroup, three outer iterations against the oracle."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    N, S, n_c, n_u = 40_000, 256, 12, 4
    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=0, depth=50)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, 3, seed=1)
    with Problem(ctx, V, D, Rt) as p, Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
        assert "<3,4>" in s.describe(20) and "nw=4" in s.describe(20)
        s.step(3, 20, 0.0)
        u, alpha, cost, _ = s.get()
    assert rel_err(alpha, wa) < TIGHT and np.abs(u - wu).max() < TIGHT
    assert cost == pytest.approx(osol.weighted_cost(V, np.c_[Rt, wu], wa, D), rel=1e-10)


# ----------------------------------------------------------------------------------- config 5
def test_unsupervised_bic_sweep_matches_oracle(ctx):
    """BASELINE.json configs[4] at oracle size: fully unsupervised --ic BIC over n_u = 2..12 (ic.py:11-15,169-218,
    with the candidate range the CLI's `--ic BIC n lo hi` passes down)."""
    from demethify_amd.ic import compute_bic, evaluate_best_ic

    V, D, _ = osol.synthetic_problem(2000, 16, 0, 5, seed=4, depth=30)
    values = range(2, 13)
    wu, wa, wn, wscores = odrv.ic_sweep(V, None, D, "uniform_", "BIC", 1, 6, 20, 1e-2, n_u_values=values,
                                        project=osol.simplex_project_columns_fast)
    gu, ga, gn, gscores = evaluate_best_ic(V, None, D, "uniform_", "BIC", 1, 6, 20, 1e-2, n_u_values=values)
    assert gn == wn and len(gscores) == len(wscores) == 11
    assert np.allclose(gscores, wscores, rtol=1e-9, atol=0)
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT
    # the product's own formula on the oracle's cost of the winner (ic.py:11-15 as coded)
    cost = osol.weighted_cost(V, wu, wa, D)
    assert compute_bic(cost, wn, 2000, 0, 16) == pytest.approx(wscores[wn - 2], rel=1e-12)


def test_partial_reference_bic_sweep_matches_oracle(ctx):
    from demethify_amd.ic import evaluate_best_ic

    V, D, Rt = osol.synthetic_problem(1500, 12, 3, 2, seed=6, depth=30)
    values = range(1, 6)
    wu, wa, wn, wscores = odrv.ic_sweep(V, Rt, D, "uniform_", "BIC", 1, 5, 20, 1e-2, n_u_values=values,
                                        project=osol.simplex_project_columns_fast)
    gu, ga, gn, gscores = evaluate_best_ic(V, Rt, D, "uniform_", "BIC", 1, 5, 20, 1e-2, n_u_values=values)
    assert gn == wn and np.allclose(gscores, wscores, rtol=1e-9, atol=0)
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT


def test_config5_shape_properties(ctx):
    """5e5 CpG x 128 samples, no reference, BIC over n_u = 2..12 (BASELINE.json configs[4]) on one GPU with two outer
    iterations per candidate: every score finite, the fast kernels (level 0) and the unfused pair (level 3) agree on
    the scores and on the selected n_u."""
    torch = pytest.importorskip("torch")
    from bench import make_inputs_on_device
    from demethify_amd import _lib as L
    from demethify_amd.deconvolution import _init_unsupervised
    from demethify_amd.device import Problem, Solver
    from demethify_amd.ic import compute_bic

    N, S = 500_000, 128
    V, D, _ = make_inputs_on_device(torch, torch.device("cuda", 0), N, S, 0, 6, seed=3)
    V_host = np.broadcast_to(np.zeros((1, 1)), (N, S))  # the uniform_ init needs the shape only
    scores = {}
    with Problem(ctx, V, D, None) as p:
        for level in (0, 3):
            ctx.set_generic(level)
            try:
                row = []
                for n_u in range(2, 13):
                    u0, a0 = _init_unsupervised("uniform_", V_host, n_u, 1)
                    with Solver(p, u0, a0, L.DMF_MODE_UNSUPERVISED) as s:
                        s.step(2, 20, 0.0)
                        u, alpha, cost, _ = s.get()
                    direct = p.cost(u, alpha)
                    assert direct == pytest.approx(cost, rel=1e-9)
                    row.append(compute_bic(direct, n_u, N, 0, S))
                scores[level] = np.array(row)
            finally:
                ctx.set_generic(0)
    assert np.all(np.isfinite(scores[0]))
    assert np.allclose(scores[0], scores[3], rtol=1e-9)
    assert int(np.argmin(scores[0])) == int(np.argmin(scores[3]))
