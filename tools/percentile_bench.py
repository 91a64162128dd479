"""Diagnostic: throughput of dmf_percentile_axis0 at bootstrap sizes (device-resident stack)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from demethify_amd.device import get_context

ctx = get_context(0)
for n, m, q in [(500, 4_000_000, [2.5, 97.5]), (100, 4_000_000, [2.5, 97.5]), (500, 400_000, [25.0, 75.0])]:
    x = torch.rand((n, m), dtype=torch.float64, device="cuda:0")
    ctx.percentile_axis0(x[:, :1000].contiguous(), q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ctx.percentile_axis0(x, q)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gb = n * m * 8 / 1e9
    # spot check against numpy on a slice
    want = np.percentile(x[:, :2000].cpu().numpy(), q, axis=0)
    ok = np.array_equal(out[:, :2000].cpu().numpy(), want)
    print(f"n={n} m={m} q={q}: {dt * 1e3:.1f} ms  ({gb:.1f} GB read once -> {gb / dt:.0f} GB/s)  bit-exact on a slice: {ok}")
    del x, out
