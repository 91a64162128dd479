#!/usr/bin/env python3
"""Diagnostic: bootstrap replicates per second on the reference's 350 x 10 toy data (upstream: ~46 / s on CPU)."""
import sys, time, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from conftest import load_toy
from demethify_amd.bootstrap import bt_ci

V, D, ref, header = load_toy()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
with tempfile.TemporaryDirectory() as d:
    bt_ci(90, 5, 1, V, D, ref, "uniform_", 10000, 20, 1e-2, header, d, [f"s{i}" for i in range(10)], None, 1)  # warm
    t0 = time.perf_counter()
    bt_ci(90, n, 1, V, D, ref, "uniform_", 10000, 20, 1e-2, header, d, [f"s{i}" for i in range(10)], None, 1)
    dt = time.perf_counter() - t0
print(f"{n} replicates in {dt:.2f} s = {n / dt:.1f} replicates / s")
