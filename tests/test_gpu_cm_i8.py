"""Oracle parity of the wide-row-group u phase (5 <= n_u <= 16 on u16 counts): k_cm_i8 (c_i on the FP64 matrix cores,
M_i exactly on the integer matrix cores, a wave per 32 CpG rows) + k_u_inner_rows16.

Every case asserts (Solver.describe) that it enters that producer; the update_u known-answer cases compare one u phase
with non-initial momentum against the oracle's update_u (deconvolution.py:80-90)."""
import numpy as np
import pytest

from oracle import solver as osol

from conftest import rel_err
from test_gpu_bench_paths import _oracle, _solve_at_level

pytestmark = pytest.mark.gpu

TIGHT = 1e-8

# (N, S, n_c, n_u, T1, depth, why)
CM_CASES = [
    (2000, 128, 0, 8, 3, 40, "config 5's shape class: no known types, 36 pairs = 3 tiles, two column groups"),
    (1500, 128, 0, 12, 3, 40, "78 pairs = 5 tiles (was k_u_phase_big)"),
    (1000, 64, 0, 16, 2, 40, "136 pairs = 9 tiles, one column group, c tile full"),
    (1031, 128, 12, 6, 3, 40, "NKC = 3 known-type chain, last block 7 rows (second half empty)"),
    (33, 4, 2, 5, 3, 40, "S = 4: one strip in range, 60 samples of the column group padded; 33 rows = one full block + 1"),
    (47, 132, 5, 9, 3, 40, "S = 4 mod 64 -> ragged third column group (NCGX = 4 instantiation), n_c not a multiple of 4"),
    (2100, 256, 16, 7, 2, 60, "four column groups (row halves in turn in the M stage), NKC = 4"),
    (1800, 200, 3, 10, 2, 60, "ragged fourth column group, 55 pairs"),
    (1200, 128, 4, 8, 3, 3000, "two count digit planes (counts above 127)"),
    (900, 256, 0, 8, 2, 2500, "two count digit planes, four column groups"),
    (5000, 64, 1, 5, 3, 40, "several blocks per wave? no: 157 blocks over 8-wave workgroups, ragged last block"),
]


@pytest.mark.parametrize("N,S,n_c,n_u,T1,depth,why", CM_CASES)
def test_cm_i8_shapes_against_oracle(ctx, N, S, n_c, n_u, T1, depth, why):
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=23, depth=depth)
    D[::5, ::3] = 0  # zero coverage (what --fillna produces)
    V = np.where(D == 0, 0.0, V)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=3)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    nd = 1 if D.max() <= 127 else 2
    u, alpha, cost, direct, _ = _solve_at_level(ctx, 0, V, D, Rt if n_c else None, u0, a0, mode, T1,
                                                [f"rowpass=k_cm_i8<nd={nd}>+"])
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT, why
    assert np.abs(u - wu).max() < TIGHT, why
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)


# the integer Gram route behind it for many features: more than 64 per launch -> several launches over the 8-bit planes;
# x image (padded known types + unknowns) up to 32 doubles per row; k_bu_cols2 up to 16 unknowns
WIDE_GRAM_CASES = [
    (1500, 128, 0, 12, 3, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "78 features: two launches, no known types"),
    (1000, 128, 0, 16, 2, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "136 features: three launches, 16 unknowns"),
    (1200, 128, 12, 12, 2, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "222 features, x image of 24 doubles"),
    (800, 128, 16, 8, 2, 3000, ["k_cm_i8<nd=2>+k_inner_bu", "gram=k_gram_i8<nd=2>"], "164 features, two count digits, ring of six"),
    (2500, 200, 5, 9, 2, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "two 128-sample groups (eight-wave k_inner_bu), ragged second group, odd n_u"),
    (1100, 256, 0, 8, 2, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "two full 128-sample groups, 1100 rows = 34 chunks + 12 rows"),
    (70, 4, 9, 5, 3, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "S = 4: two lanes of the b_u stream active; 70 rows = 2 chunks + 6 rows"),
    (2000, 128, 0, 6, 3, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "21 features only: behind the fused b_u stream the integer route runs at every width"),
    (1200, 64, 1, 5, 3, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "one known type, five unknowns (20 features)"),
    (600, 130, 3, 13, 2, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "S = 2 mod 4: the last lane of a row holds one pair in range and one out"),
    (300, 2, 2, 6, 3, 40, ["k_cm_i8<nd=1>+k_inner_bu"], "two samples"),
    (450, 6, 0, 9, 3, 900, ["k_cm_i8<nd=2>+k_inner_bu"], "six samples, two count digit planes"),
    (700, 64, 15, 16, 2, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "376 features in six launches, row image of 32 doubles; K = 31"),
    (600, 64, 20, 14, 2, 40, ["k_cm_i8<nd=1>+k_u_inner_rows", "gram=k_gram_mfma"], "row image of 34 doubles: beyond the integer Gram, k_gram_mfma behind k_cm_i8"),
]


@pytest.mark.parametrize("N,S,n_c,n_u,T1,depth,expect,why", WIDE_GRAM_CASES)
def test_wide_gram_routes_against_oracle(ctx, N, S, n_c, n_u, T1, depth, expect, why):
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=29, depth=depth)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=4)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    u, alpha, cost, direct, _ = _solve_at_level(ctx, 0, V, D, Rt if n_c else None, u0, a0, mode, T1, expect)
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT, why
    assert np.abs(u - wu).max() < TIGHT, why
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)


def test_cm_i8_many_blocks_per_wave(ctx):
    """More 32-row blocks than the grid has waves (the persistent loop, the next-block prefetch and the R_trunc row
    hand-over are entered), compared with the first-generation kernels on the same start."""
    from demethify_amd import _lib as L

    N, S, n_c, n_u = 32 * 8 * 256 * 2 + 32 * 5 + 3, 64, 4, 6
    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=5, depth=30)
    rs = np.random.RandomState(7)
    u0 = rs.uniform(size=(N, n_u))
    a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    u_new, a_new, c_new, d_new, _ = _solve_at_level(ctx, 0, V, D, Rt, u0, a0, L.DMF_MODE_PARTIAL, 3, ["rowpass=k_cm_i8<nd=1>"])
    u_old, a_old, c_old, d_old, _ = _solve_at_level(ctx, 3, V, D, Rt, u0, a0, L.DMF_MODE_PARTIAL, 3, ["rowpass=k_u_phase_mfma"])
    assert np.abs(u_new - u_old).max() < 1e-10 and np.abs(a_new - a_old).max() < 1e-10
    assert d_new == pytest.approx(d_old, rel=1e-11)


@pytest.mark.parametrize("n_c,n_u", [(0, 9), (6, 5), (2, 14)])
def test_update_u_known_answer_on_cm_i8(ctx, n_c, n_u):
    """One u phase from a NON-initial momentum state through dmf_update_u against the oracle's u_phase."""
    from demethify_amd.device import Problem

    N, S = 700, 128
    V, D, Rt = osol.synthetic_problem(N, S, max(n_c, 1), n_u, seed=41, depth=50)
    if n_c == 0:
        Rt = np.zeros((N, 0))
    rs = np.random.RandomState(9)
    u, u_prev = rs.uniform(size=(N, n_u)), rs.uniform(size=(N, n_u))
    alpha = rs.dirichlet(np.ones(n_c + n_u), S).T
    d = float(D.max()) ** 2
    l_w = np.linalg.norm(alpha[-n_u:]) ** 2 * d
    a1, l_w_prev = 1.7, 0.9 * l_w
    want = osol.u_phase(u, alpha, 20, a1, l_w_prev, l_w, u_prev, V, Rt, n_u, D)
    with Problem(ctx, V, D, Rt if n_c else None) as p:
        got = p.update_u(u, u_prev, alpha, 20, a1, l_w_prev, l_w)
    assert np.abs(got[0] - want[0]).max() < 1e-11 and np.abs(got[1] - want[1]).max() < 1e-11
    assert got[2] == pytest.approx(want[2], rel=1e-15) and got[3] == want[3]


def test_purity_constrained_solver_on_wide_row_groups(ctx):
    """mdwbssmf_deconv_p with six unknowns: 70 inner steps through k_cm_i8 + k_inner_bu, Frank-Wolfe alpha phase."""
    from demethify_amd import deconvolution as dd

    V, D, Rt = osol.synthetic_problem(900, 128, 3, 6, seed=13, depth=25)
    purity = np.linspace(0.2, 0.9, 128)
    u0, R, a0 = osol.init_partial_purity("uniform_", V, D, Rt, 6, purity, seed=5)
    wu, wa = osol.solve_partial_purity(u0.copy(), R, a0.copy(), V, D, Rt, 6, purity, 3, 70, 0.0)
    gu, ga = dd.mdwbssmf_deconv_p(u0.copy(), R.copy(), a0.copy(), V, D, Rt, 6, purity, n_iter1=3, n_iter2=70, tol=0.0)
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT
    assert np.allclose(ga[:3].sum(axis=0), purity, atol=1e-12) and np.allclose(ga[3:].sum(axis=0), 1 - purity, atol=1e-12)


def test_more_inner_steps_than_the_fused_kernel_holds(ctx):
    """1100 inner steps: beyond k_inner_bu's momentum table -> k_cm_i8 + k_u_inner_rows16 (chunked table) + k_bu_cols."""
    from demethify_amd import deconvolution as dd

    V, D, Rt = osol.synthetic_problem(300, 64, 2, 5, seed=3, depth=25)
    u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, 5, seed=2)
    wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, 5, 2, 1100, 0.0, project=osol.simplex_project_columns_fast)
    gu, ga = dd.mdwbssmf_deconv(u0, R, a0, V, D, Rt, 5, n_iter1=2, n_iter2=1100, tol=0.0)
    assert rel_err(ga, wa) < TIGHT and np.abs(gu - wu).max() < TIGHT


# odd sample counts: rows of V start 8 bytes off a 16-byte boundary every other time, the row's last sample has no partner
# (every integer-count kernel takes samples in pairs or fours): second-generation kernels throughout
ODD_S_CASES = [
    (2000, 255, 12, 4, 3, 40, ["rowpass=k_rowpass_v2<3,4>", "nw=4", "gram=k_gram_i8<nd=1>"], "the headline instantiation, one sample short"),
    (4096 + 5, 129, 6, 3, 3, 40, ["rowpass=k_rowpass_v2<2,3>", "nw=3", "gram=k_gram_i8<nd=1>"], "third column group holds one sample"),
    (1000, 33, 0, 2, 4, 900, ["rowpass=k_rowpass_v2<0,2>", "gram=k_gram_i8<nd=2>"], "unsupervised, two count digit planes"),
    (5, 7, 2, 1, 5, 40, ["rowpass=k_rowpass_v2<1,1>"], "five rows, seven samples"),
    (333, 3, 1, 1, 4, 40, ["rowpass=k_rowpass_v2<1,1>"], "three samples"),
    (1500, 127, 0, 8, 3, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "wide row groups: the last group of four holds three samples"),
    (900, 255, 4, 6, 2, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "four column groups, eight-wave k_inner_bu"),
    (700, 5, 3, 5, 3, 40, ["k_cm_i8<nd=1>+k_inner_bu"], "five samples: one group of four and one lone sample"),
    (650, 67, 0, 12, 2, 40, ["k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "second column group holds three samples"),
    (800, 201, 12, 9, 2, 3000, ["k_cm_i8<nd=2>+k_inner_bu", "gram=k_gram_i8<nd=2>"], "S = 1 mod 4, two count digit planes, known types"),
]


@pytest.mark.parametrize("N,S,n_c,n_u,T1,depth,expect,why", ODD_S_CASES)
def test_odd_sample_counts_against_oracle(ctx, N, S, n_c, n_u, T1, depth, expect, why):
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=37, depth=depth)
    D[::6, ::4] = 0
    V = np.where(D == 0, 0.0, V)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=6)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    u, alpha, cost, direct, _ = _solve_at_level(ctx, 0, V, D, Rt if n_c else None, u0, a0, mode, T1, expect)
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT, why
    assert np.abs(u - wu).max() < TIGHT, why
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)


@pytest.mark.parametrize("S,n_c,n_u", [(131, 5, 3), (77, 0, 7)])
def test_resampled_problem_with_odd_sample_count(ctx, S, n_c, n_u):
    """bootstrap.py:28: a row resample of an odd-S problem (dmf_problem_gather gathers the u16 counts too) solved on the
    second-generation kernels against the oracle on the fancy-indexed arrays."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    N, T1 = 1200, 3
    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=51, depth=45)
    idx = osol.bootstrap_indices(17, N)
    Vg, Dg, Rg = V[idx], D[idx], Rt[idx]
    u0, a0, wu, wa = _oracle(Vg, Dg, Rg, n_c, n_u, T1, seed=8)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    with Problem(ctx, V, D, Rt if n_c else None) as p, p.gather(idx) as g, Solver(g, u0, a0, mode) as s:
        path = s.describe(20)
        assert ("k_rowpass_v2" in path) if n_u <= 4 else ("k_cm_i8" in path), path
        s.step(T1, 20, 0.0)
        u, alpha, _, _ = s.get()
    assert np.abs(alpha - wa).max() < TIGHT and np.abs(u - wu).max() < TIGHT


# more than 16 known cell types (reference atlases): the producer's chain of up to 12 links behind wave-uniform guards, the
# integer Gram while a row's known + unknown values fit 32 doubles, a wave per sample in the alpha phase beyond K = 32
MANY_KNOWN_CASES = [
    (1500, 128, 25, 3, 3, 40, ["rowpass=k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>", "alpha=k_alpha_phase_lanes"], "K = 28"),
    (1200, 64, 28, 4, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_inner_bu", "gram=k_gram_i8<nd=1>"], "row image of exactly 32 doubles, 122 features"),
    (800, 130, 17, 6, 2, 2500, ["rowpass=k_cm_i8<nd=2>+k_inner_bu", "gram=k_gram_i8<nd=2>"], "17 known types: five chain links, one column of the padded copy in use"),
    (900, 200, 39, 2, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows", "gram=k_gram_mfma", "alpha=k_alpha_phase_lanes"], "K = 41: a wave per sample in the alpha phase"),
    (700, 96, 48, 2, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows", "alpha=k_alpha_phase_lanes"], "48 known types: all twelve chain links"),
    (600, 300, 20, 8, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows"], "two panels, wide row group, 20 known types"),
]


@pytest.mark.parametrize("N,S,n_c,n_u,T1,depth,expect,why", MANY_KNOWN_CASES)
def test_many_known_types_against_oracle(ctx, N, S, n_c, n_u, T1, depth, expect, why):
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=43, depth=depth)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=9)
    u, alpha, cost, direct, _ = _solve_at_level(ctx, 0, V, D, Rt, u0, a0, L.DMF_MODE_PARTIAL, T1, expect)
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT, why
    assert np.abs(u - wu).max() < TIGHT, why
    want = osol.weighted_cost(V, np.c_[Rt, wu], wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)


# more than 16 unknown types (upstream's --ic sweep runs to 25): two c tiles, the pair tiles of M_i over several launches of the
# producer (their digit table does not fit the LDS at once), inner iterations with 32 lanes per CpG row
MANY_UNKNOWN_CASES = [
    (1500, 128, 0, 17, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows"], "17 unknowns: second c tile with one live row, 10 pair tiles in two launches"),
    (1200, 128, 0, 20, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows", "gram=k_bu_cols+k_gram_i8<nd=1>"], "210 pairs; integer Gram in four launches"),
    (1000, 128, 0, 25, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows", "gram=k_bu_cols+k_gram_i8<nd=1>"], "the sweep's last candidate: 21 pair tiles in three launches, 325 features in six"),
    (900, 64, 6, 20, 2, 3000, ["rowpass=k_cm_i8<nd=2>+k_u_inner_rows"], "known types, two count digit planes, one column group"),
    (800, 255, 3, 18, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows"], "odd S, four column groups"),
    (600, 300, 0, 32, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows"], "32 unknowns (both c tiles full), two panels"),
    (500, 40, 16, 17, 2, 40, ["rowpass=k_cm_i8<nd=1>+k_u_inner_rows"], "K = 33"),
]


@pytest.mark.parametrize("N,S,n_c,n_u,T1,depth,expect,why", MANY_UNKNOWN_CASES)
def test_many_unknown_types_against_oracle(ctx, N, S, n_c, n_u, T1, depth, expect, why):
    from demethify_amd import _lib as L

    V, D, Rt = osol.synthetic_problem(N, S, n_c, n_u, seed=47, depth=depth)
    u0, a0, wu, wa = _oracle(V, D, Rt, n_c, n_u, T1, seed=10)
    mode = L.DMF_MODE_PARTIAL if n_c else L.DMF_MODE_UNSUPERVISED
    u, alpha, cost, direct, _ = _solve_at_level(ctx, 0, V, D, Rt if n_c else None, u0, a0, mode, T1, expect)
    assert rel_err(alpha, wa) < TIGHT and np.abs(alpha - wa).max() < TIGHT, why
    assert np.abs(u - wu).max() < TIGHT, why
    want = osol.weighted_cost(V, np.c_[Rt, wu] if n_c else wu, wa, D)
    assert cost == pytest.approx(want, rel=1e-9) and direct == pytest.approx(want, rel=1e-11)
