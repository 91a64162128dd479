"""BASELINE.json sizes: config 2 against the oracle directly, the headline size through properties that do
not depend on the size (the oracle would need minutes per iteration there)."""
import numpy as np
import pytest

from oracle import solver as osol

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_config2_full_size_against_oracle(ctx):
    """BASELINE.json configs[1]: 1e5 CpG x 64 samples, 6 known + 2 unknown types, two outer iterations."""
    from demethify_amd import _lib as L
    from demethify_amd.deconvolution import solve_problem
    from demethify_amd.device import Problem

    V, D, Rt = osol.synthetic_problem(100_000, 64, 6, 2, seed=0, depth=50)
    u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, 2, seed=1)
    wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, 2, 2, 20, 0.0,
                                project=osol.simplex_project_columns_fast)
    with Problem(ctx, V, D, Rt) as p:
        gu, ga, cost, iters = solve_problem(p, u0, a0, L.DMF_MODE_PARTIAL, 2, 20, 0.0, return_info=True)
        direct = p.cost(gu, ga)
    assert iters == 2
    assert rel_err(ga, wa) < 1e-9 and np.abs(gu - wu).max() < 1e-9
    want = osol.weighted_cost(V, np.c_[Rt, wu], wa, D)
    assert cost == pytest.approx(want, rel=1e-10) and direct == pytest.approx(want, rel=1e-12)


def test_config2_natural_stop_matches_oracle(ctx):
    """Stop-iteration parity at a BASELINE size ("plus natural-stop parity run" for config 2).
    tol = 2e5 makes the oracle's |cf - cf_0| < tol (deconvolution.py:220) fire at outer iteration 11 (its cost
    differences there: ... 4.2e5, 2.6e5, 1.4e5); the device must freeze the iterate at the same iteration."""
    from demethify_amd import _lib as L
    from demethify_amd.deconvolution import solve_problem
    from demethify_amd.device import Problem

    V, D, Rt = osol.synthetic_problem(100_000, 64, 6, 2, seed=0, depth=50)
    u0, R, a0 = osol.init_partial("uniform_", V, D, Rt, 2, seed=1)
    trace = []
    wu, wa = osol.solve_partial(u0.copy(), R, a0.copy(), V, D, Rt, 2, 50, 20, 2e5, trace=trace,
                                project=osol.simplex_project_columns_fast)
    assert len(trace) == 11
    with Problem(ctx, V, D, Rt) as p:
        gu, ga, cost, iters = solve_problem(p, u0, a0, L.DMF_MODE_PARTIAL, 50, 20, 2e5, return_info=True)
        direct = p.cost(gu, ga)
    assert iters == len(trace)
    assert rel_err(ga, wa) < 1e-8 and np.abs(ga - wa).max() < 1e-8 and np.abs(gu - wu).max() < 1e-8
    # the Gram-form cost the stop test uses against the streaming cost of the same iterate, in ABSOLUTE terms:
    # the CLI's default threshold is 1e-2 (demethify.py:35)
    assert abs(cost - direct) < 1e-4 and abs(direct - trace[-1]) < 1e-4 * max(1.0, abs(trace[-1]) * 1e-9)


def test_headline_size_properties(ctx):
    """1e6 x 256, 12 + 4 (the bench workload): invariants of the iteration instead of an oracle run."""
    torch = pytest.importorskip("torch")
    from bench import make_inputs_on_device
    from demethify_amd import _lib as L
    from demethify_amd.device import Problem, Solver

    N, S, n_c, n_u = 1_000_000, 256, 12, 4
    V, D, Rt = make_inputs_on_device(torch, torch.device("cuda", 0), N, S, n_c, n_u, seed=0)
    rs = np.random.RandomState(1)
    u0 = rs.uniform(size=(N, n_u))
    a0 = rs.dirichlet(np.ones(n_c + n_u), S).T
    with Problem(ctx, V, D, Rt) as p:
        costs = []
        with Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
            _, _, c0, _ = s.get()
            costs.append(c0)
            for _ in range(3):
                s.step(1, 20, 0.0)
                costs.append(s.get_cost()[0])
                # the cost the stop test uses (Gram form: v'Dv - 2 a.b + a'Ga, v'Dv ~ 4e9 here) against the
                # streaming cost of the same iterate, in ABSOLUTE terms: the stop threshold of the CLI is 1e-2
                # (demethify.py:35), so the cancellation error must stay orders of magnitude below that
                assert abs(s.direct_cost() - costs[-1]) < 1e-3
            u1, a1, c1, it1 = s.get()
        assert p.cost(u1, a1) == pytest.approx(c1, rel=1e-9) and abs(p.cost(u1, a1) - c1) < 1e-3
        assert it1 == 3 and all(b < a for a, b in zip(costs, costs[1:]))  # monotone decrease from a random start
        # feasibility: proportions on the simplex, profiles in [0, 1]
        assert np.abs(a1.sum(axis=0) - 1).max() < 1e-12 and a1.min() >= 0
        assert u1.min() >= 0 and u1.max() <= 1
        # fixed summation orders everywhere: a second run is bitwise identical
        with Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
            s.step(3, 20, 0.0)
            u2, a2, c2, _ = s.get()
        assert np.array_equal(u1, u2) and np.array_equal(a1, a2) and c1 == c2
        # the unfused kernels (a different summation order) land on the same iterate
        ctx.set_generic(3)
        try:
            with Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
                s.step(3, 20, 0.0)
                u3, a3, c3, _ = s.get()
        finally:
            ctx.set_generic(0)
        assert np.abs(a3 - a1).max() < 1e-9 and np.abs(u3 - u1).max() < 1e-9 and c3 == pytest.approx(c1, rel=1e-10)


def test_bootstrap_building_blocks_at_headline_size(ctx):
    """BASELINE.json configs[3] (1e6 CpG x 256 samples, --confidence 95 500) piece by piece: the device row gather of one
    resample (bootstrap.py:28 = RandomState(seed).randint applied to meth_f, counts, ref alike) against the oracle on
    a 50 000-row slice gathered the same way, a permutation followed by its inverse (bit-for-bit the original
    problem), and the percentile over a 500-replicate stack of 1e6 positions against numpy on sampled columns."""
    torch = pytest.importorskip("torch")
    from bench import make_inputs_on_device
    from demethify_amd.device import Problem

    N, S, n_c, n_u = 1_000_000, 256, 12, 4
    dev = torch.device("cuda", 0)
    V, D, Rt = make_inputs_on_device(torch, dev, N, S, n_c, n_u, seed=0)
    rs = np.random.RandomState(5)
    u = rs.uniform(size=(N, n_u))
    alpha = rs.dirichlet(np.ones(n_c + n_u), S).T
    idx = osol.bootstrap_indices(11, N)  # what sklearn's resample(random_state=11) draws
    with Problem(ctx, V, D, Rt) as p:
        n_s = 50_000
        it = torch.from_numpy(idx[:n_s]).to(dev)
        Vs, Ds, Rs = V[it].cpu().numpy(), D[it].cpu().numpy().astype(np.int64), Rt[it].cpu().numpy()
        want = osol.weighted_cost(Vs, np.c_[Rs, u[:n_s]], alpha, Ds)
        with p.gather(idx[:n_s]) as g:
            assert g.cost(u[:n_s], alpha) == pytest.approx(want, rel=1e-12)
        with p.gather(idx) as g:  # the full resample: 1e6 gathered rows; its first 50 000 rows are the slice above
            full = g.cost(u, alpha)
            assert np.isfinite(full) and full > want
        base = p.cost(u, alpha)
        perm = rs.permutation(N)
        inv = np.argsort(perm)
        with p.gather(perm) as g1, g1.gather(inv) as g2:
            assert g2.cost(u, alpha) == base  # same rows in the same order: bit for bit
            # and a pure reordering of the rows changes the sum only in its rounding
            assert g1.cost(u[perm], alpha) == pytest.approx(base, rel=1e-12)
    del V, D, Rt
    x = torch.rand((500, 1_000_000), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    q = [2.5, 97.5]
    got = ctx.percentile_axis0(x, q)
    cols = np.r_[0:1000, 499_000:500_000, 999_000:1_000_000]
    want_q = np.percentile(x[:, torch.from_numpy(cols).to(dev)].cpu().numpy(), q, axis=0)
    assert np.array_equal(got[:, torch.from_numpy(cols).to(dev)].cpu().numpy(), want_q)  # numpy's "linear" method, bit for bit


def test_config4_bootstrap_end_to_end_at_headline_size(ctx, tmp_path):
    """BASELINE.json configs[3] through the driver itself (demethify_amd.bootstrap.bt_ci = bootstrap.py:10-93) at
    1e6 CpG x 256 samples, 12 + 4: 8 replicates of 5 outer iterations.  The oracle cannot run a 1e6-row replicate in
    test time, so what is asserted needs no oracle solve: the replicates' seeds and row draws are the oracle's
    (bootstrap.py:27-28), every replicate's loop cost agrees with the streaming cost of its iterate, the bounds are
    ordered, and the two CSV files parse back to the percentile arrays bit for bit (bootstrap.py:70,89)."""
    import re
    import zlib

    import pandas as pd

    torch = pytest.importorskip("torch")
    from bench import make_inputs_on_device
    from demethify_amd.bootstrap import bt_ci

    N, S, n_c, n_u, B = 1_000_000, 256, 12, 4, 8
    Vd, Dd, Rd = make_inputs_on_device(torch, torch.device("cuda", 0), N, S, n_c, n_u, seed=0)
    V, D, Rt = Vd.cpu().numpy(), Dd.cpu().numpy().astype(np.int64), Rd.cpu().numpy()
    del Vd, Dd, Rd
    header = [f"type_{k}" for k in range(n_c)]
    samples = [f"s{k}" for k in range(S)]
    seen = {}

    def observe(i, seed_i, idx, solver):
        seen[i] = (seed_i, zlib.crc32(np.ascontiguousarray(idx).tobytes()), solver.get_cost(), solver.direct_cost())

    res = bt_ci(95, B, n_u, V, D, Rt, "uniform_", 5, 20, 0.0, header, str(tmp_path), samples, None, 1,
                materialize=False, _observe=observe)
    seeds = osol.bootstrap_seeds(1, B)
    assert sorted(seen) == list(range(B))
    for i in range(B):
        seed_i, crc, (gram_cost, iters), stream_cost = seen[i]
        assert seed_i == seeds[i] and iters == 5
        assert crc == zlib.crc32(osol.bootstrap_indices(seeds[i], N).tobytes())
        assert abs(gram_cost - stream_cost) < 1e-3 and stream_cost > 0
    props_df, (lower_u, upper_u) = res
    assert lower_u.shape == upper_u.shape == (N, n_u) and (lower_u <= upper_u).all()
    assert (lower_u >= 0).all() and (upper_u <= 1).all() and (upper_u > lower_u).mean() > 0.9
    lo_p = np.array([[props_df.iloc[k, i][0] for i in range(S)] for k in range(n_c + n_u)])
    hi_p = np.array([[props_df.iloc[k, i][1] for i in range(S)] for k in range(n_c + n_u)])
    assert (lo_p <= hi_p).all() and (lo_p >= 0).all() and (hi_p <= 1).all() and (hi_p > lo_p).any()

    number = r"(?:np\.float64\()?([-+0-9.eEinfa]+)\)?"
    pair = re.compile(r"\(" + number + ", " + number + r"\)")

    def parse(frame):
        lo = np.empty(frame.shape)
        hi = np.empty(frame.shape)
        for c, col in enumerate(frame.columns):
            both = frame[col].str.extract(pair).astype(np.float64).to_numpy()
            lo[:, c], hi[:, c] = both[:, 0], both[:, 1]
        return lo, hi

    est = pd.read_csv(tmp_path / "confidence_interval_methylation_estimate.csv")
    assert list(est.columns) == [f"unknown_cell_{k + 1}" for k in range(n_u)] and len(est) == N
    lo, hi = parse(est)
    assert np.array_equal(lo, lower_u) and np.array_equal(hi, upper_u)  # repr round trip: bit for bit
    prop = pd.read_csv(tmp_path / "confidence_interval_celltypes_proportions.csv", index_col=0)
    assert list(prop.index) == header + [f"unknown_cell_{k + 1}" for k in range(n_u)] and list(prop.columns) == samples
    lo, hi = parse(prop)
    assert np.array_equal(lo, lo_p) and np.array_equal(hi, hi_p)
