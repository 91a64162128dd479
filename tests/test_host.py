"""CPU-only tests: host logic of the drop-in layer and the C-ABI's exported symbols."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import solver as osol

from conftest import ROOT, read_props


def test_library_exports_every_declared_symbol():
    """The .so must load on a CPU-only box and export every function include/demethify_hip.h declares."""
    from demethify_amd import _build, _lib

    _build.build()
    header = (ROOT / "include" / "demethify_hip.h").read_text()
    declared = set(re.findall(r"\b(dmf_[a-z_0-9]+)\s*\(", header))
    declared -= {"dmf_status"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().dmf_abi_version() == 1
    assert _lib.load().dmf_status_string(2) == b"shape mismatch"


def test_product_does_not_import_the_oracle():
    for path in (ROOT / "demethify_amd").rglob("*.py"):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path


def test_wls_intercept_matches_reference_based_golden(toy):
    from demethify_amd.init_func import wls_intercept

    V, D, ref, _ = toy
    alpha = np.concatenate([wls_intercept(D[:, k:k + 1] * V[:, k:k + 1], D[:, k:k + 1], ref) for k in range(10)], axis=1)
    assert np.abs(alpha - read_props("output_ref_based")).max() < 1e-13
    one = wls_intercept((D[:, 0] * V[:, 0]), D[:, 0], ref)
    assert one.shape == (5,) and np.allclose(one, alpha[:, 0])


@pytest.mark.parametrize("option", ["uniform_", "beta", "uniform"])
def test_init_matches_oracle_rng_stream(toy, option):
    from demethify_amd.deconvolution import init_BSSMF_md

    V, D, ref, _ = toy
    for seed in (1, [1], 7):
        got = init_BSSMF_md(option, V, D, ref, 2, seed=seed)
        want = osol.init_partial(option, V, D, ref, 2, seed=seed)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    a = init_BSSMF_md(option, V, D, ref, 2, seed=1)[2]
    b = init_BSSMF_md(option, V, D, ref, 2, seed=[1])[2]
    assert not np.array_equal(a, b)  # --seed on the CLI yields a list: a different stream (demethify.py:43)


def test_init_forces_uniform_underscore_when_more_unknowns_than_samples(toy):
    from demethify_amd.deconvolution import init_BSSMF_md

    V, D, ref, _ = toy
    got = init_BSSMF_md("beta", V, D, ref, 11, seed=3)
    want = osol.init_partial("uniform_", V, D, ref, 11, seed=3)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])


def test_unsupervised_uniform_keeps_upstream_nameerror(toy):
    from demethify_amd.deconvolution import _init_unsupervised

    V, _, _, _ = toy
    with pytest.raises(NameError):
        _init_unsupervised("uniform", V, 2, 1)
    u, a = _init_unsupervised("uniform_", V, 4, 1)
    wu, wa = osol.init_unsupervised("uniform_", V, 4, 1)
    assert np.array_equal(u, wu) and np.array_equal(a, wa)


def test_table_reader_equals_upstream_loop(tmp_path, monkeypatch):
    """demethify.py:103-143 restated with usecols + worker processes: same values, dtypes and layout as the
    reference's read_csv / column_stack loop, for bedmethyl and for csv input."""
    import pandas as pd

    from conftest import UPSTREAM
    from demethify_amd import tables

    paths = [str(UPSTREAM / "output_gen" / f"sample{i}.bed") for i in range(1, 11)]
    cols = [pd.read_csv(p, sep="\t") for p in paths]
    want_f = np.column_stack([t["percent_modified"].values / 100 for t in cols])
    want_c = np.column_stack([t["valid_coverage"].values for t in cols])
    for workers in ("1", "4"):
        monkeypatch.setenv("DEMETHIFY_IO_WORKERS", workers)
        got_f, got_c = tables.read_samples(paths, True, False)
        assert np.array_equal(got_f, want_f) and np.array_equal(got_c, want_c)
        assert got_c.dtype == want_c.dtype and got_f.flags["C_CONTIGUOUS"] and got_c.flags["C_CONTIGUOUS"]
    monkeypatch.delenv("DEMETHIFY_IO_WORKERS")
    # csv: a single-column file gets coverage 1 (demethify.py:134-135); NaN handling follows --fillna
    one = tmp_path / "one.csv"
    one.write_text("percent_modified\n0.25\n\n0.5\n".replace("\n\n", "\nNaN\n"))
    two = tmp_path / "two.csv"
    two.write_text("percent_modified,valid_coverage\n0.1,7\n0.2,\n0.3,9\n")
    f, c = tables.read_samples([str(one), str(two)], False, True)
    assert np.array_equal(f, [[0.25, 0.1], [0.0, 0.2], [0.5, 0.3]]) and np.array_equal(c, [[1, 7], [1, 0], [1, 9]])
    f, c = tables.read_samples([str(one), str(two)], False, False)
    assert np.isnan(f[1, 0]) and np.isnan(c[1, 1]) and c.dtype == np.float64
    bad = tmp_path / "bad.csv"
    bad.write_text("a,b\n1,2\n")
    with pytest.raises(KeyError):
        tables.read_samples([str(bad)], False, False)
    short = tmp_path / "short.csv"
    short.write_text("percent_modified,valid_coverage\n0.1,7\n")
    with pytest.raises(ValueError):
        tables.read_samples([str(two), str(short)], False, False)


def _pandas_tables(paths, bedmethyl):
    import os

    from demethify_amd import tables

    os.environ["DEMETHIFY_PANDAS_READER"] = "1"
    try:
        return tables.read_samples(paths, bedmethyl, False)
    finally:
        del os.environ["DEMETHIFY_PANDAS_READER"]


def test_native_table_reader_is_bit_identical_to_pandas(tmp_path):
    """csrc/dmf_tables.hip against the pandas path (demethify.py:103-143) on the reference's own bedmethyl files and on
    synthetic tables with every number format the parser has a branch for."""
    from conftest import UPSTREAM
    from demethify_amd import tables

    golden = [str(UPSTREAM / "output_gen" / f"sample{i}.bed") for i in range(1, 11)]
    golden.append(str(UPSTREAM / "config1" / "bed2_intersect.bed"))
    for group in (golden[:10], golden[10:]):
        got = tables.read_samples_native(group, True)
        want = _pandas_tables(group, True)
        assert got is not None and got[1].dtype == want[1].dtype == np.int64
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert got[0].flags.c_contiguous and got[0].shape == want[0].shape
    rs = np.random.RandomState(3)
    n = 5000
    for k, sep in enumerate(["\t", ","]):
        files = []
        for j in range(3):
            x = rs.uniform(0, 100 if sep == "\t" else 1, n)
            texts = []
            for i, v in enumerate(x):  # 17-digit repr, short decimals, integers, exponent notation, long digit strings
                texts.append([repr(float(v)), f"{v:.2f}", f"{v:.6f}", str(int(v)), f"{v:.10e}", f"{v:.20f}", f"{v / 1e5:.3e}"][i % 7])
            cov = rs.poisson(30, n) + (0 if j else 1)
            path = tmp_path / f"t{k}{j}.txt"
            with open(path, "w") as f:
                f.write(sep.join(["chrom", "valid_coverage", "other", "percent_modified"]) + "\n")
                for i in range(n):
                    f.write(sep.join(["chr1", str(cov[i]), "x y", texts[i]]) + ("\r\n" if j == 1 else "\n"))
            files.append(str(path))
        got = tables.read_samples_native(files, sep == "\t")
        want = _pandas_tables(files, sep == "\t")
        assert got is not None
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # what the parser treats specially is left to pandas: NA in the coverage column, a quoted field
    bad = tmp_path / "na.csv"
    bad.write_text("valid_coverage,percent_modified\n3,0.5\nNA,0.25\n")
    assert tables.read_samples_native([str(bad)], False) is None
    quoted = tmp_path / "q.csv"
    quoted.write_text('valid_coverage,percent_modified\n3,"0.5"\n')
    assert tables.read_samples_native([str(quoted)], False) is None
    # rows with another field count than the header: a trailing separator in every row makes pandas shift the names (an
    # implicit index column), a short row gets NaN -- both are left to pandas, wherever in the file they occur
    trailing = tmp_path / "trailing.bed"
    trailing.write_text("chrom\tvalid_coverage\tpercent_modified\nchr1\t2\t5.0\t\nchr1\t3\t6.0\t\n")
    assert tables.read_samples_native([str(trailing)], True) is None
    late = tmp_path / "late.bed"
    late.write_text("chrom\tvalid_coverage\tpercent_modified\n" + "chr1\t2\t5.0\n" * 50 + "chr1\t3\t6.0\t7\n" + "chr1\t2\t5.0\n" * 5)
    assert tables.read_samples_native([str(late)], True) is None
    short = tmp_path / "short.csv"
    short.write_text("valid_coverage,other,percent_modified\n3,1,0.5\n4,0.25\n")
    assert tables.read_samples_native([str(short)], False) is None
    # a single-column csv gets coverage 1 (demethify.py:137-138)
    single = tmp_path / "single.csv"
    single.write_text("percent_modified\n0.5\n0.125\n")
    got = tables.read_samples_native([str(single)], False)
    want = _pandas_tables([str(single)], False)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[1].tolist() == [[1], [1]]


def test_prefetcher_orders_results_and_propagates_errors():
    """staging.Prefetcher: fn(item) one ahead, results in order, an exception surfaces at its item."""
    from demethify_amd.staging import Prefetcher

    seen = []

    def fn(k):
        seen.append(k)
        if k == 3:
            raise RuntimeError("bad restart")
        return k * k

    out = []
    pf = Prefetcher(range(3), fn, depth=1)
    for k, r in pf:
        out.append((k, r))
    assert out == [(0, 0), (1, 1), (2, 4)]
    pf2 = Prefetcher(range(6), fn, depth=1)
    got = []
    with pytest.raises(RuntimeError, match="bad restart"):
        for k, r in pf2:
            got.append(k)
    assert got == [0, 1, 2]
    pf2.close()
    # a consumer that leaves early lets the worker end
    pf3 = Prefetcher(range(100), lambda k: k, depth=1)
    for k, r in pf3:
        if k == 2:
            break
    pf3.close()
    pf3.join(timeout=5)
    assert not pf3.is_alive()


def test_sharded_restarts_prepare_hook_matches_plain_loop():
    from demethify_amd import shard

    def prepare(k):
        return 10 * k

    def solve_plain(k, best):
        cost = float((k - 2) ** 2)
        return (np.full((2, 1), k), np.full((1, 2), k), cost)

    def solve_prepared(k, best, prepared):
        assert prepared == 10 * k
        return solve_plain(k, best)

    a = shard.sharded_restarts(5, solve_plain, ((2, 1), (1, 2)))
    b = shard.sharded_restarts(5, solve_prepared, ((2, 1), (1, 2)), prepare=prepare)
    assert a[2] == b[2] == 2
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[3], b[3])


def test_interval_csv_writer_is_byte_identical_to_pandas(tmp_path):
    """dmf_write_interval_csv (the profile confidence intervals, bootstrap.py:85-91 upstream) against DataFrame.to_csv of a
    DataFrame of (lower, upper) tuples: same bytes, including the float spellings repr() has for small, large, integral,
    negative-zero and non-finite values."""
    import pandas as pd

    from demethify_amd.bootstrap import _write_interval_csv

    rng = np.random.RandomState(3)
    n_rows, n_u = 5000, 3
    lower = rng.rand(n_rows, n_u)
    upper = lower + rng.rand(n_rows, n_u) * 1e-3
    special = [0.0, 1.0, 1e-5, 1e-4, 1.5e-7, 0.1 + 0.2, 5e-324, 123456789.0, 1e16, 1e15, 0.5, 2.5e-5, 1e22,
               1.7976931348623157e308, 9.999999999999999e-5, 0.00011, 12345678901234567.0, float("nan"), float("inf"),
               -0.0, -3.25, 2.0 ** -1074, 1 / 3, 2 / 3, 1e-310]
    for i, v in enumerate(special):
        lower[i, i % n_u] = v
        upper[(7 * i) % n_rows, (i + 1) % n_u] = v
    cols = [f"unknown_cell_{k + 1}" for k in range(n_u)]
    want = tmp_path / "pandas.csv"
    pd.DataFrame({cols[k]: [(lower[j, k], upper[j, k]) for j in range(n_rows)] for k in range(n_u)}).to_csv(want, index=False)
    got = tmp_path / "native.csv"
    assert _write_interval_csv(str(got), cols, lower, upper)
    assert got.read_bytes() == want.read_bytes()
    assert not _write_interval_csv(str(got), ["a,b", "c", "d"], lower, upper)  # a header that needs quoting: pandas' job


def test_prefetcher_with_several_workers_keeps_item_order():
    """Three workers, results of uneven duration: the consumer still sees the items in order, never more than `depth`
    ahead, and an exception arrives at its item's position."""
    import threading
    import time

    from demethify_amd.staging import Prefetcher

    started, lock = [], threading.Lock()

    def fn(k):
        with lock:
            started.append(k)
        time.sleep(0.02 * ((7 * k) % 3))
        if k == 9:
            raise ValueError("item 9")
        return -k

    pf = Prefetcher(range(12), fn, depth=3, workers=3)
    seen = []
    with pytest.raises(ValueError, match="item 9"):
        for k, r in pf:
            assert r == -k
            with lock:
                assert max(started) <= k + 3  # at most `depth` items beyond the one being consumed
            seen.append(k)
    assert seen == list(range(9))
    pf.close()


def test_kernel_selection_table():
    """Kernel selection is one pure function (csrc/dmf_select.hip): its answers for a grid of shapes, levels, count
    encodings, inner-step counts and flags are pinned by a checked-in table (no GPU involved)."""
    import importlib.util

    from demethify_amd import _lib as L

    spec = importlib.util.spec_from_file_location("make_kernel_selection", ROOT / "tests" / "golden" / "make_kernel_selection.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = L.load()
    want = [line.rstrip("\n").split("\t") for line in (ROOT / "tests" / "golden" / "kernel_selection.tsv").read_text().splitlines()
            if not line.startswith("#")]
    rows = list(mod.grid())
    assert len(rows) == len(want)
    for row, w in zip(rows, want):
        assert [str(x) for x in row] == w[:8]
        assert mod.describe(lib, row) == w[8], row
    # a few rows a reader can check against DESIGN.md section 5 by eye
    table = {tuple(int(x) for x in w[:8]): w[8] for w in want}
    f32 = L.DMF_SELECT_COUNTS_F32_EXACT
    assert table[(1000000, 256, 12, 4, 1, 0, 20, f32)].startswith("rowpass=k_rowpass_v2<3,4> nw=4 grid=512")
    assert "k_rowpass_fused<3,4>" in table[(1000000, 256, 12, 4, 0, 0, 20, f32)]        # counts without integer copies
    assert "k_rowpass_fused<3,4>" in table[(1000000, 256, 12, 4, 1, 0, 0, f32)]         # no inner step: u is not clipped
    assert "k_cm_i8<nd=1>+k_inner_bu" in table[(1000000, 127, 0, 8, 1, 0, 20, f32)]
    assert "k_u_phase_mfma(split)" in table[(1000000, 256, 12, 4, 1, 0, 500, f32)]      # --purity's 500 inner steps
    assert "k_rowpass_v2" not in table[(1000000, 256, 12, 4, 1, 0, 20, f32 | L.DMF_SELECT_ALPHA_OUTSIDE_UNIT)]
    assert "alpha=k_alpha_frank_wolfe_row16" in table[(1000000, 256, 12, 4, 1, 0, 20, f32 | L.DMF_SELECT_PURITY)]


def test_ic_sweep_range_checks_come_before_any_gpu_work():
    """evaluate_best_ic: a candidate below 1 and an EXPLICIT candidate beyond the kernels' 64 cell types raise (with their
    own messages) before a device is touched; only upstream's default 1..25 is trimmed to what fits (GPU test)."""
    from demethify_amd.ic import evaluate_best_ic

    V = np.full((20, 3), 0.5)
    D = np.full((20, 3), 7, dtype=np.int64)
    ref = np.full((20, 40), 0.5)
    with pytest.raises(ValueError, match="at least 1"):
        evaluate_best_ic(V, ref, D, "uniform_", "BIC", 1, 2, 2, 0.0, n_u_values=[0, 2])
    with pytest.raises(ValueError, match="more than 64 cell types"):
        evaluate_best_ic(V, ref, D, "uniform_", "BIC", 1, 2, 2, 0.0, n_u_values=[2, 30])
