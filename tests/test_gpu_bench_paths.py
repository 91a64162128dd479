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

# (N, S, n_c, n_u, T1, what the case is for); every case runs on the second-generation pair (level 0: k_rowpass_v2 on
# u16 counts + the integer-matrix-core Gram) and on the first-generation fused FP64 kernel (level 4: level 0's
# fall-back for counts beyond 32639 or reference profiles outside [0, 1])
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


def _solve_at_level(ctx, level, V, D, Rt, u0, a0, mode, T1, expect):
    """(u, alpha, Gram-form cost, direct cost, path) with the kernel selection `level`; the Problem is created under
    that level because the integer count copies of the second-generation kernels are built at level 0 only."""
    from demethify_amd.device import Problem, Solver

    ctx.set_generic(level)
    try:
        with Problem(ctx, V, D, Rt) as p, Solver(p, u0, a0, mode) as s:
            path = s.describe(20)
            for token in expect:
                assert token in path, (token, path)
            it, _ = s.step(T1, 20, 0.0)
            assert it == T1
            u, alpha, cost, _ = s.get()
            direct = s.direct_cost()
            assert direct == p.cost(u, alpha)  # the device-resident and the host-array entry points agree bit for bit
    finally:
        ctx.set_generic(0)
    return u, alpha, cost, direct, path


@pytest.mark.parametrize("level,kernel", [(0, "k_rowpass_v2"), (4, "k_rowpass_fused")])
@pytest.mark.parametrize("N,S,n_c,n_u,T1,why", FUSED_CASES)
def test_fused_instantiations_against_oracle(ctx, N, S, n_c, n_u, T1, why, level, kernel):
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=31, depth=40)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=1)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    expect = [f"rowpass={kernel}<{(n_c + 3) // 4},{n_u}>", f"nw={(S + 63) // 64}", f"tail={N % 16}"]
    if level == 0:
        expect.append("gram=k_gram_i8<nd=1>")
    u, alpha, cost, direct, _ = _solve_at_level(ctx, level, V, D, Rt if n_c else None, u0, a0, mode, T1, expect)
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT, why
    assert np.abs(u - wu).max() < TIGHT, why
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)


# shapes only the second-generation pair takes (the first-generation kernel needs S % 4 == 0 and N >= 16)
V2_CASES = [
    (5, 6, 2, 1, 5, 40, "fewer than 16 rows: one partial block"),
    (100, 130, 4, 2, 4, 40, "S = 2 mod 4, ragged third column group (clamped V columns, zero-padded counts)"),
    (1000, 64, 3, 2, 4, 3000, "counts up to ~3300: two balanced 8-bit digits (nd=2)"),
    (777, 30, 0, 1, 4, 40, "one feature only (u u), no known types"),
    (2000, 256, 16, 4, 3, 60, "74 features: two launches of the integer Gram (64 + 10)"),
    (1500, 64, 16, 4, 3, 2500, "nd=2 with 74 features and a wide row image: 64 on eight waves (ring of six) + 10 on four"),
    (4096, 256, 12, 4, 3, 60, "eight-wave Gram, 128 row ranges x 2 sample halves: the XCD-aware workgroup order"),
    (8197, 64, 12, 4, 3, 60, "eight-wave Gram, one sample quarter-group per range (nsh = 1), last block 5 rows"),
    (1013, 200, 10, 3, 3, 60, "eight-wave Gram, 36 features (28 lanes idle), ragged samples, natural workgroup order"),
    (3000, 256, 12, 4, 3, 2500, "eight-wave Gram with two count digits (counts above 127), 58 features in one launch"),
    (2005, 130, 8, 3, 3, 1500, "eight-wave Gram, two count digits, 30 features, ragged samples, last block 21 rows"),
]


@pytest.mark.parametrize("N,S,n_c,n_u,T1,depth,why", V2_CASES)
def test_second_generation_shapes_against_oracle(ctx, N, S, n_c, n_u, T1, depth, why):
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=17, depth=depth)
    D[::7, ::3] = 0  # zero coverage (what --fillna produces)
    V = np.where(D == 0, 0.0, V)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=2)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    nd = 1 if D.max() <= 127 else 2
    u, alpha, cost, direct, _ = _solve_at_level(ctx, 0, V, D, Rt if n_c else None, u0, a0, mode, T1,
                                                ["rowpass=k_rowpass_v2<", f"gram=k_gram_i8<nd={nd}>"])
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT, why
    assert np.abs(u - wu).max() < TIGHT, why
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)


def test_second_generation_preconditions_fall_back(ctx):
    """Counts beyond two 8-bit digits, fractional weights or reference profiles outside [0, 1] must not reach the
    integer kernels: the first-generation / unfused FP64 kernels take over and parity holds."""
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(640, 32, 4, 2, seed=23, depth=40)
    cases = {"big counts": (V, D * 1000, Rt), "reference above 1": (V, D, Rt * 1.5)}
    for name, (v, d, rt) in cases.items():
        u0, a0, wu, wa = _oracle(v, d, rt, 4, 2, 3, seed=1)
        u, alpha, _, _, path = _solve_at_level(ctx, 0, v, d, rt, u0, a0, L.DMF_MODE_PARTIAL, 3, ["rowpass="])
        assert "k_rowpass_v2" not in path and "k_gram_i8" not in path, (name, path)
        assert rel_err(alpha, wa) < TIGHT and np.abs(u - wu).max() < TIGHT, name


def test_headline_columns_at_oracle_size(ctx):
    """The bench's shape in everything but the row count: 256 samples, 12 + 4 types, Poisson(50) depth, enough
    rows (40 000) for ten blocks per workgroup, three outer iterations against the oracle."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    N, S, n_c, n_u = 40_000, 256, 12, 4
    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=0, depth=50)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, 3, seed=1)
    with Problem(ctx, V, D, Rt) as p, Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
        assert "k_rowpass_v2<3,4>" in s.describe(20) and "nw=4" in s.describe(20) and "k_gram_i8<nd=1>" in s.describe(20)
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


@pytest.mark.parametrize("N,S,n_c,n_u,depth", [(1500, 64, 16, 4, 2500), (4096, 256, 12, 4, 60), (20000, 64, 6, 2, 40),
                                                (4096, 256, 12, 4, 2500)])
def test_in_launch_hand_overs_are_race_free(ctx, N, S, n_c, n_u, depth):
    """The K <= 16 alpha kernel closes the outer iteration in its last workgroup to arrive (atomics-only hand-over, no
    kernel boundary in between); the Gram reduce did the same for its columns until round 3 and hands over to a launch of
    its own now (k_gram_v2_finish).  The sums involved are exact integers or fixed-order f64 sums, so every repetition of
    a solve must give the SAME BITS; a missed or late contribution (seen once with no-return atomics: 1e-5 relative, one
    run in three) shows up as a difference."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=5, depth=depth)
    rng = np.random.RandomState(4)
    u0 = rng.uniform(size=(N, n_u))
    a0 = rng.dirichlet(np.ones(n_c + n_u), S).T.copy()
    first = None
    with Problem(ctx, V, D, Rt) as p:
        for rep in range(40):
            with Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
                assert "gram=k_gram_i8" in s.describe(20)
                s.step(4, 20, 0.0)
                u, alpha, cost, _ = s.get()
            if first is None:
                first = (u, alpha, cost)
            else:
                assert cost == first[2], rep
                np.testing.assert_array_equal(alpha, first[1], err_msg=f"repetition {rep}")
                np.testing.assert_array_equal(u, first[0], err_msg=f"repetition {rep}")


@pytest.mark.parametrize("N,S,n_c,n_u,depth,expect", [
    (3000, 320, 12, 4, 40, ["rowpass=k_rowpass_v2<3,4>", "nw=5", "gram=k_gram_i8<nd=1>"]),
    (2100, 512, 12, 4, 40, ["rowpass=k_rowpass_v2<3,4>", "nw=8", "gram=k_gram_i8<nd=1>"]),
    (1500, 449, 0, 1, 40, ["rowpass=k_rowpass_v2<0,1>", "nw=8"]),
    (1234, 384, 16, 3, 2500, ["rowpass=k_rowpass_v2<4,3>", "nw=6", "gram=k_gram_i8<nd=2>"]),
    # wider row groups, or more than 512 samples: the producer of the wide-row-group path walks panels of 256 samples
    (2500, 512, 10, 5, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows", "gram=k_bu_cols+k_gram_i8<nd=1>"]),
    (1700, 321, 10, 5, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows", "gram=k_bu_cols+k_gram_i8<nd=1>"]),  # odd S, 65-sample panel
    (1800, 640, 6, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows"]),      # narrow row group beyond the row pass's 512 samples
    (1100, 1024, 12, 4, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows", "gram=k_bu_cols+k_gram_i8<nd=1>"]),  # four full panels
    (900, 769, 0, 9, 2500, ["rowpass=k_cm_i8<nd=2>+k_u_inner_rows"]),     # a one-sample fourth panel, two count digit planes
    (700, 1030, 3, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows"]),      # five panels, the last one of six samples
    (500, 2047, 12, 4, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows", "gram=k_bu_cols+k_gram_i8<nd=1>"]),  # eight panels, odd S
    (400, 2100, 3, 2, 40, ["rowpass=k_u_phase_gram"]),                    # beyond 2048 samples: the any-shape kernels
])
def test_beyond_256_samples(ctx, N, S, n_c, n_u, depth, expect):
    """257..512 samples with up to four unknowns: the second-generation row pass as ONE workgroup of up to eight waves per
    CU.  Wider row groups or up to 2048 samples: k_cm_i8 over panels of 256 samples + the inner-iteration kernel, with the
    integer Gram (b_u stream kernel) and the u16 cost kernel behind them."""
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=11, depth=depth)
    T1 = 3
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=2)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    u, alpha, cost, direct, path = _solve_at_level(ctx, 0, V, D, Rt if n_c else None, u0, a0, mode, T1, expect)
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT
    assert np.abs(u - wu).max() < TIGHT
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)
