"""The stop test |cf - cf_0| < tol (deconvolution.py:218-220) at deep coverage.

The loop's cost comes from the Gram form v'Dv - 2 a.b + a'Ga, whose cancellation error grows with v'Dv (with the
sequencing depth): where its bound is not far below tol, the library decides a stop on the STREAMING cost of
deconvolution.py:15-17 for this and the previous iterate -- the reference's own formula (dmf_solver_stop_info)."""
import numpy as np
import pytest

from oracle import solver as osol

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [120, 1000, 2500])
def test_stops_at_headline_size_are_decided_on_streaming_costs(ctx, depth):
    """1e6 x 256, 12 + 4 with Poisson(depth) coverage (two count digits): the Gram-form cost drifts from the streaming
    cost by 1e-6 .. 1e-5 here (recorded below; a tenth of the CLI's threshold is 1e-3).  With the confirmation switched
    on, every stop decision inside the band is taken on streaming costs, and the decisions are those of the
    reference's formula evaluated on direct_cost()."""
    torch = pytest.importorskip("torch")
    from bench import make_inputs_on_device
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    N, S, n_c, n_u = 1_000_000, 256, 12, 4
    V, D, Rt = make_inputs_on_device(torch, torch.device("cuda", 0), N, S, n_c, n_u, seed=0, depth=depth)
    rs = np.random.RandomState(1)
    u0 = rs.uniform(size=(N, n_u))
    a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    cli_tol = 1e-2  # demethify.py:35
    ctx.set_stop_confirmation(0)
    with Problem(ctx, V, D, Rt) as p, Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
        assert "k_rowpass_v2<3,4>" in s.describe(20) and "nd=2" in s.describe(20)
        gram_err, stream = [], [s.direct_cost()]
        for _ in range(3):
            s.step(1, 20, 0.0)
            stream.append(s.direct_cost())
            gram_err.append(abs(s.get_cost()[0] - stream[-1]))
        print(f"depth {depth}: |Gram-form cost - streaming cost| over 3 iterations: {gram_err}, cost {stream[-1]:.6g}")
        # the documented rule: stops are confirmed where the Gram form's error BOUND, 1e-15 N S max(counts), reaches
        # tol / 20 -- at the CLI's threshold that is the case from a depth of ~2000 on at this size -- and the bound holds
        bound = 1e-15 * N * S * float(D.max())
        assert max(gram_err) < bound
        s.step(0, 20, cli_tol)
        assert s.stop_info()["confirm_stops"] == (bound >= cli_tol / 20) == (depth == 2500)
        # ... and a threshold a little below the current decrease makes the next iterations pause: each decision must be
        # the reference's |cf - cf_0| < tol on the streaming costs (the first iteration inside the band can only be
        # decided on the Gram form; from the second on the previous streaming cost is known).  The decrease is not
        # monotone in the first iterations (tests below): iterate until it has shrunk three times in a row first.
        shrinking = 0
        for _ in range(200):
            s.step(1, 20, 0.0)
            stream.append(s.direct_cost())
            d1, d0 = abs(stream[-1] - stream[-2]), abs(stream[-2] - stream[-3])
            shrinking = shrinking + 1 if d1 < d0 else 0
            if shrinking >= 3:
                break
        assert shrinking >= 3
        tol = abs(stream[-1] - stream[-2]) * 0.6
        ctx.set_stop_confirmation(1)  # (at this large a threshold the bound would not ask for it: make it)
        stopped_at = None
        for k in range(200):
            it, conv = s.step(1, 20, tol)
            stream.append(s.direct_cost())
            info = s.stop_info()
            if info["n_confirmed"] > 0:
                # the cost the stop test used IS the streaming cost of this iterate: no error against direct_cost()
                assert info["last_stream_cost"] == stream[-1]
                assert conv == (abs(stream[-1] - stream[-2]) < tol)
            if conv:
                stopped_at = it
                break
        print(f"depth {depth}: tol {tol:.4g}, stopped at iteration {stopped_at}, {s.stop_info()}")
        assert stopped_at is not None and s.stop_info()["n_confirmed"] >= 1
        # a further step() is a no-op: the iterate is frozen at the stop iteration
        it2, conv2 = s.step(5, 20, tol)
        assert it2 == stopped_at and conv2
    ctx.set_stop_confirmation(0)
    # (what the Gram form alone would have been off by: recorded by the print above; its BOUND, not the measured value,
    # decides whether stops are confirmed -- dmf_solver_step)
    assert max(gram_err) < 1.0


def test_natural_stop_at_depth_2500_matches_oracle(ctx):
    """2e4 x 64, 6 + 2 at Poisson(2500) coverage (counts up to ~2760: two count digits).  The oracle's cost decrease is
    not monotone there (9.7e5 at iteration 1, a peak of 4.4e6 at 6, ... 1.02e6, 9.3e5, 8.3e5 at 37..39): with tol = 9e5
    its |cf - cf_0| < tol (deconvolution.py:220) first holds at outer iteration 39, and the device must freeze the
    iterate there.  (At this size the Gram form's error bound, ~1e-3, is nine orders below tol: no confirmation.)"""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    V, D, Rt = osol.synthetic_problem(20_000, 64, 6, 2, seed=0, depth=2500)
    u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, 2, seed=1)
    tol = 9e5
    trace = []
    wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, 2, 60, 20, tol, trace=trace,
                                project=osol.simplex_project_columns_fast)
    assert len(trace) == 39
    with Problem(ctx, V, D, Rt) as p, Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
        assert "k_rowpass_v2<2,2>" in s.describe(20) and "nd=2" in s.describe(20)
        it, conv = s.step(60, 20, tol)
        info = s.stop_info()
        gu, ga, cost, iters = s.get()
        direct = s.direct_cost()
    assert conv and it == len(trace) == iters
    assert not info["confirm_stops"]
    assert direct == pytest.approx(trace[-1], rel=1e-10) and cost == pytest.approx(trace[-1], rel=1e-10)
    assert rel_err(ga, wa) < 1e-8 and np.abs(ga - wa).max() < 1e-8 and np.abs(gu - wu).max() < 1e-8


def test_small_problems_stop_on_the_gram_form(ctx, toy):
    """The 350 x 10 example: the Gram form's error bound is orders of magnitude below the CLI's threshold, no streaming
    pass is spent on its stop tests (and it stops at the committed run's iteration 54: test_gpu_solver.py)."""
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    V, D, ref, _ = toy
    u0, R, a0 = osol.init_partial("uniform_", V, D, ref, 1, seed=1)
    with Problem(ctx, V, D, ref) as p, Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
        it, conv = s.step(10000, 20, 1e-2)
        info = s.stop_info()
    assert conv and it == 54
    assert not info["confirm_stops"] and info["n_confirmed"] == 0 and info["n_unconfirmed"] == 0
