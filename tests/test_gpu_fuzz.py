"""A short randomised differential run (tools/fuzz_parity.py): random small shapes -- ragged sample counts, partial last
blocks, zero-coverage cells, one and two count digits, 0..16 known and 1..12 unknown types -- through kernel selection
level 0 against the CPU oracle at 1e-8.  The full campaign (800+ cases per seed) is run by hand when kernels change."""
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_shapes_against_oracle():
    proc = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_parity.py"), "90", "7"], cwd=ROOT,
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0 and "MISMATCH" not in proc.stdout, (proc.stdout[-3000:], proc.stderr[-2000:])
    assert "90 cases" in proc.stdout


def test_random_wide_row_groups_against_oracle():
    """The same with a bias towards 5..16 unknown types on S % 4 == 0 (k_cm_i8 + k_inner_bu and the many-feature integer
    Gram behind them)."""
    proc = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_parity.py"), "70", "5", "wide"], cwd=ROOT,
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0 and "MISMATCH" not in proc.stdout, (proc.stdout[-3000:], proc.stderr[-2000:])
    assert "70 cases" in proc.stdout and "k_cm_i8" in proc.stdout


def test_random_many_known_types_against_oracle():
    """... and towards 17..48 known cell types (the producer's long chain, a wave per sample in the alpha phase)."""
    proc = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_parity.py"), "50", "3", "many"], cwd=ROOT,
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0 and "MISMATCH" not in proc.stdout, (proc.stdout[-3000:], proc.stderr[-2000:])
    assert "50 cases" in proc.stdout and "k_cm_i8" in proc.stdout
