"""Shared fixtures.  GPU tests carry ``@pytest.mark.gpu`` and call through the C-ABI; everything
else runs on CPU.  Nothing here (or in any -m gpu test) reads /root/reference at run time."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"
UPSTREAM = GOLDEN / "upstream"

# parity bar of BASELINE.json's north_star: proportions within 1e-5 (relative) of the CPU path
PARITY_RTOL = 1e-5


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_toy():
    """The reference's synthetic example (test/output_gen): 350 CpG x 10 samples, 5 known types.
    Read exactly as demethify/demethify.py:103-120 reads bedmethyl input."""
    ref = pd.read_csv(UPSTREAM / "output_gen" / "ref_matrix.bed", sep="\t").iloc[:, 3:]
    header = list(ref.columns)
    freqs, counts = [], []
    for i in range(1, 11):
        t = pd.read_csv(UPSTREAM / "output_gen" / f"sample{i}.bed", sep="\t")
        freqs.append(t["percent_modified"].values / 100)
        counts.append(t["valid_coverage"].values)
    return np.column_stack(freqs), np.column_stack(counts), ref.values, header


@pytest.fixture(scope="session")
def toy():
    return load_toy()


def read_props(folder):
    return pd.read_csv(UPSTREAM / folder / "celltypes_proportions.csv", index_col=0).values


def read_profile(folder):
    return pd.read_csv(UPSTREAM / folder / "methylation_profile_estimate.csv").values


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """Some GPU tests use torch tensors as device buffers: torch has to initialise its HIP runtime before the
    library loads its own (see demethify_amd/_lib.py), so the order is settled once per session."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session."""
    from demethify_amd.device import get_context

    return get_context(0)


def rel_err(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
