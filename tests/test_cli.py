"""Command-line drop-in tests: same flags, files and values as the reference's README commands."""
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from oracle import drivers as odrv
from oracle import solver as osol

from conftest import ROOT, UPSTREAM

SAMPLES = [str(UPSTREAM / "output_gen" / f"sample{i}.bed") for i in range(1, 11)]
REF = str(UPSTREAM / "output_gen" / "ref_matrix.bed")


def run_cli(*argv, expect=0):
    proc = subprocess.run([sys.executable, "-m", "demethify_amd", *argv], cwd=ROOT, capture_output=True, text=True)
    assert proc.returncode == expect, proc.stderr[-2000:]
    return proc


def same_csv(got_path, want_path, tol):
    got = pd.read_csv(got_path, index_col=0)
    want = pd.read_csv(want_path, index_col=0)
    assert list(got.columns) == list(want.columns) and list(got.index) == list(want.index)
    assert got.index.name == want.index.name
    assert np.abs(got.values - want.values).max() < tol


def test_reference_based_command(tmp_path):
    """README 'Reference based case' (no --nbunknown): BASELINE.json configs[0] plumbing, CPU only."""
    run_cli("--ref", REF, "--methfreq", *SAMPLES, "--bedmethyl", "--outdir", str(tmp_path), "--noprint")
    same_csv(tmp_path / "celltypes_proportions.csv", UPSTREAM / "output_ref_based" / "celltypes_proportions.csv", 1e-13)
    assert (tmp_path / "log.log").read_text().startswith("Total execution time = ")
    assert not (tmp_path / "methylation_profile_estimate.csv").exists()


def test_config1_single_sample_six_types(tmp_path):
    ref = pd.read_csv(UPSTREAM / "config1" / "bed1_select_ref_intersect.bed", sep="\t")
    six = ref.iloc[:, :9]
    six.to_csv(tmp_path / "ref6.bed", sep="\t", index=False)
    run_cli("--ref", str(tmp_path / "ref6.bed"), "--methfreq", str(UPSTREAM / "config1" / "bed2_intersect.bed"),
            "--bedmethyl", "--nbunknown", "0", "--outdir", str(tmp_path / "out"), "--noprint")
    got = pd.read_csv(tmp_path / "out" / "celltypes_proportions.csv", index_col=0).values.ravel()
    assert np.abs(got - np.array([0, 0, 0.03548921, 0, 0, 0.96451079])).max() < 5e-9


def test_csv_input_without_coverage_column(tmp_path):
    """csv mode: fractions in 'percent_modified'; a single-column file gets valid_coverage = 1."""
    V, D, Rt = osol.synthetic_problem(120, 2, 3, 0, seed=2, depth=10)
    pd.DataFrame(Rt, columns=["a", "b", "c"]).to_csv(tmp_path / "ref.csv", index=False)
    pd.DataFrame({"valid_coverage": D[:, 0], "percent_modified": V[:, 0]}).to_csv(tmp_path / "s1.csv", index=False)
    pd.DataFrame({"percent_modified": V[:, 1]}).to_csv(tmp_path / "s2.csv", index=False)
    run_cli("--ref", str(tmp_path / "ref.csv"), "--methfreq", str(tmp_path / "s1.csv"), str(tmp_path / "s2.csv"),
            "--outdir", str(tmp_path / "out"), "--noprint")
    got = pd.read_csv(tmp_path / "out" / "celltypes_proportions.csv", index_col=0)
    w1 = osol.nnls_intercept_proportions(D[:, :1] * V[:, :1], D[:, :1], Rt).ravel()
    w2 = osol.nnls_intercept_proportions(V[:, 1:2], np.ones((120, 1)), Rt).ravel()
    assert np.abs(got["s1.csv"].values - w1).max() < 1e-12 and np.abs(got["s2.csv"].values - w2).max() < 1e-12
    assert list(got.index) == ["a", "b", "c"]


def test_argument_errors(tmp_path):
    p = run_cli("--ref", REF, "--methfreq", *SAMPLES, "--bedmethyl", "--outdir", str(tmp_path), "--ic", "AIC",
                "--nbunknown", "2", expect=1)
    assert "--ic cannot be used with --nbunknown" in p.stderr
    p = run_cli("--ref", REF, "--methfreq", *SAMPLES, "--bedmethyl", "--outdir", str(tmp_path), "--nbunknown", "1",
                "--init", "SVD", expect=1)
    assert "not part of this build" in p.stderr
    run_cli("--methfreq", *SAMPLES, expect=2)  # --outdir is required
    # --ic NAME [n [lo hi]]: a candidate range needs both ends and 1 <= lo <= hi
    for bad in (["BIC", "5", "3"], ["BIC", "5", "4", "2"], ["BIC", "5", "0", "3"], ["BIC", "5", "2", "3", "4"]):
        p = run_cli("--methfreq", *SAMPLES, "--bedmethyl", "--outdir", str(tmp_path), "--ic", *bad, expect=1)
        assert "--ic" in p.stderr


def test_flag_surface_matches_reference():
    from demethify_amd.demethify import build_parser

    flags = {a.option_strings[0]: a for a in build_parser()._actions if a.option_strings and a.dest != "help"}
    assert set(flags) == {"--methfreq", "--ref", "--iterations", "--nbunknown", "--purity", "--termination", "--init",
                          "--outdir", "--fillna", "--ic", "--confidence", "--plot", "--restart", "--seed",
                          "--noprint", "--bedmethyl"}
    assert flags["--seed"].default == 1 and flags["--seed"].nargs == 1
    assert flags["--termination"].default == 1e-2 and flags["--init"].default == "uniform_"
    ns = build_parser().parse_args(["--methfreq", "a", "--outdir", "o", "--seed", "5"])
    assert ns.seed == [5]  # a list when given on the command line, as upstream


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_partial_reference_command_reproduces_committed_outputs(tmp_path):
    run_cli("--ref", REF, "--methfreq", *SAMPLES, "--nbunknown", "1", "--bedmethyl", "--outdir", str(tmp_path),
            "--noprint")
    same_csv(tmp_path / "celltypes_proportions.csv", UPSTREAM / "output_partial_ref" / "celltypes_proportions.csv", 1e-8)
    got = pd.read_csv(tmp_path / "methylation_profile_estimate.csv")
    want = pd.read_csv(UPSTREAM / "output_partial_ref" / "methylation_profile_estimate.csv")
    assert list(got.columns) == ["unknown_cell_1"] and np.abs(got.values - want.values).max() < 1e-8


@pytest.mark.gpu
def test_purity_command_reproduces_committed_outputs(tmp_path):
    """README 'Partial-reference based case with purity' (test/purity): Frank-Wolfe alpha phase, 100 x 500."""
    run_cli("--ref", REF, "--methfreq", *SAMPLES, "--nbunknown", "1", "--purity", "60", "80", "90", "20", "50", "90",
            "100", "30", "50", "10", "--bedmethyl", "--outdir", str(tmp_path), "--noprint")
    same_csv(tmp_path / "celltypes_proportions.csv", UPSTREAM / "purity" / "celltypes_proportions.csv", 1e-8)
    got = pd.read_csv(tmp_path / "methylation_profile_estimate.csv")
    want = pd.read_csv(UPSTREAM / "purity" / "methylation_profile_estimate.csv")
    assert np.abs(got.values - want.values).max() < 1e-8


@pytest.mark.gpu
def test_unsupervised_command_reproduces_committed_outputs(tmp_path):
    run_cli("--methfreq", *SAMPLES, "--nbunknown", "4", "--bedmethyl", "--outdir", str(tmp_path), "--noprint")
    same_csv(tmp_path / "celltypes_proportions.csv", UPSTREAM / "unsupervised" / "celltypes_proportions.csv", 1e-8)


@pytest.mark.gpu
def test_model_selection_command_reproduces_committed_outputs(tmp_path):
    run_cli("--ref", REF, "--methfreq", *SAMPLES, "--bedmethyl", "--ic", "AIC", "--outdir", str(tmp_path), "--noprint")
    assert (tmp_path / "log.log").read_text().strip().endswith("Number of unknowns that minimises AIC : 10")
    same_csv(tmp_path / "celltypes_proportions.csv", UPSTREAM / "model_selection" / "celltypes_proportions.csv", 1e-7)


@pytest.mark.gpu
def test_unsupervised_bic_range_command(tmp_path, toy):
    """BASELINE.json configs[4] through the CLI: no --ref, `--ic BIC 5 2 4` sweeps n_u = 2..4 only (upstream
    hard-codes 1..25, ic.py:171); scores and winner against the oracle's sweep."""
    V, D, _, _ = toy
    run_cli("--methfreq", *SAMPLES, "--bedmethyl", "--ic", "BIC", "5", "2", "4", "--iterations", "40", "20",
            "--outdir", str(tmp_path), "--noprint")
    wu, wa, wn, _ = odrv.ic_sweep(V, None, D, "uniform_", "BIC", 1, 40, 20, 1e-2, n_u_values=range(2, 5))
    assert (tmp_path / "log.log").read_text().strip().endswith(f"Number of unknowns that minimises BIC : {wn}")
    got = pd.read_csv(tmp_path / "celltypes_proportions.csv", index_col=0)
    assert list(got.index) == [f"unknown_cell_{i + 1}" for i in range(wn)]
    assert np.abs(got.values - wa).max() < 1e-8
    prof = pd.read_csv(tmp_path / "methylation_profile_estimate.csv")
    assert np.abs(prof.values - wu).max() < 1e-8


@pytest.mark.gpu
def test_restarts_pick_the_min_cost_seed(tmp_path, toy):
    V, D, ref, _ = toy
    run_cli("--ref", REF, "--methfreq", *SAMPLES, "--nbunknown", "1", "--bedmethyl", "--outdir", str(tmp_path),
            "--noprint", "--restart", "3", "--iterations", "40", "20")
    u, alpha, best, costs = odrv.restart_pick(V, D, ref, 1, "uniform_", [1, 2, 3], 40, 20, 1e-2)
    got = pd.read_csv(tmp_path / "celltypes_proportions.csv", index_col=0).values
    assert np.abs(got - alpha).max() < 1e-8


@pytest.mark.gpu
def test_confidence_intervals_match_oracle_bootstrap(tmp_path, toy):
    """test/ci in the reference is stale, so the bootstrap is checked against the
    oracle's replicate loop: same seeds, same resampled rows, numpy percentiles."""
    V, D, ref, _ = toy
    run_cli("--ref", REF, "--methfreq", *SAMPLES, "--nbunknown", "1", "--bedmethyl", "--outdir", str(tmp_path),
            "--noprint", "--confidence", "90", "6", "--iterations", "30", "20")
    us, alphas = odrv.bootstrap_replicates(6, 1, V, D, ref, "uniform_", 30, 20, 1e-2, 1)
    lo, hi = odrv.percentile_bounds(alphas, 90)
    table = pd.read_csv(tmp_path / "confidence_interval_celltypes_proportions.csv", index_col=0)
    assert table.index.name == "Cell Type" and list(table.index)[-1] == "unknown_cell_1"
    for s_i, col in enumerate(table.columns):
        for k, cell in enumerate(table[col]):
            a, b = eval(cell, {"np": np})  # "(lo, hi)" tuples, as upstream writes them
            assert abs(a - lo[k, s_i]) < 1e-8 and abs(b - hi[k, s_i]) < 1e-8
    lo_u, hi_u = odrv.percentile_bounds(us, 90)
    prof = pd.read_csv(tmp_path / "confidence_interval_methylation_estimate.csv")
    a, b = eval(prof["unknown_cell_1"][17], {"np": np})
    assert abs(a - lo_u[17, 0]) < 1e-8 and abs(b - hi_u[17, 0]) < 1e-8


def run_cli_ranks(n_ranks, *argv):
    """Two ranks sharing cuda:0 with gloo collectives: the sharded drivers on the real kernels."""
    import os
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, DEMETHIFY_DIST_BACKEND="gloo", DEMETHIFY_DEVICE="0")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "demethify_amd", *argv],
                          cwd=ROOT, capture_output=True, text=True, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]


@pytest.mark.gpu
def test_two_ranks_give_the_single_process_answer(tmp_path):
    """--restart, --confidence and --ic sharded over 2 ranks == the same command on 1 rank."""
    base = ["--ref", REF, "--methfreq", *SAMPLES, "--bedmethyl", "--noprint", "--iterations", "30", "20"]
    cases = {"restart": ["--nbunknown", "1", "--restart", "5"],
             "ci": ["--nbunknown", "1", "--confidence", "90", "5"]}
    for name, extra in cases.items():
        run_cli(*base, *extra, "--outdir", str(tmp_path / f"{name}_1"))
        run_cli_ranks(2, *base, *extra, "--outdir", str(tmp_path / f"{name}_2"))
        files = ["celltypes_proportions.csv", "methylation_profile_estimate.csv"]
        if name == "ci":
            files += ["confidence_interval_celltypes_proportions.csv", "confidence_interval_methylation_estimate.csv"]
        for f in files:
            assert (tmp_path / f"{name}_1" / f).read_text() == (tmp_path / f"{name}_2" / f).read_text(), (name, f)
    # model selection: 25 candidates dealt to 2 ranks (default iterations: the committed folder is the target)
    run_cli_ranks(2, "--ref", REF, "--methfreq", *SAMPLES, "--bedmethyl", "--noprint", "--ic", "AIC", "--outdir",
                  str(tmp_path / "ic_2"))
    assert (tmp_path / "ic_2" / "log.log").read_text().strip().endswith("Number of unknowns that minimises AIC : 10")
    same_csv(tmp_path / "ic_2" / "celltypes_proportions.csv", UPSTREAM / "model_selection" / "celltypes_proportions.csv", 1e-7)
