"""Bootstrap confidence intervals: host-side mirror of demethify/bootstrap.py.

Replicate i resamples the CpG rows with replacement (``sklearn.utils.resample(..., random_state=seed_i)``
upstream, i.e. ``RandomState(seed_i).randint(0, N, N)`` applied to meth_f, counts and ref alike), runs a
full solve on the resampled problem and contributes one column per sample / unknown to the percentile
bounds.  The full data set is uploaded once; each replicate is a row gather on the device; the percentiles
over the replicates are taken on the device too (dmf_percentile_axis0).  With torch.distributed initialised,
replicate i runs on rank i mod world, the KB-sized proportions are gathered and the profile stacks are
re-partitioned by CpG range with one all-to-all.
"""
from __future__ import annotations

import os

import numpy as np
import pandas as pd

from . import _lib as L
from . import shard
from .deconvolution import init_BSSMF_md, init_BSSMF_md_p, solve_problem
from .device import Problem, Solver, get_context
from .init_func import wls_intercept

__all__ = ["bt_ci", "bootstrap_seed_sequence", "bootstrap_row_indices"]


def bootstrap_seed_sequence(seed, n_bootstrap):
    """bootstrap.py:27: ``seed = seed + i`` inside the loop is cumulative (seed_0 + i (i + 1) / 2).
    A list-valued seed (CLI ``--seed s``) fails here exactly as upstream does (TypeError)."""
    out = []
    for i in range(n_bootstrap):
        seed = seed + i if seed is not None else None
        out.append(seed)
    return out


def bootstrap_row_indices(seed, n_rows):
    """Row indices sklearn's ``resample`` draws for ``random_state=seed`` (bootstrap.py:28)."""
    return np.random.RandomState(seed).randint(0, n_rows, size=(n_rows,))


def _device_stack(n_local, width):
    """(n_local, width) float64 buffer on this process's GPU for the replicate profiles (B x N x n_u doubles: 16 GB at
    500 x 1e6 x 4, held in HBM instead of travelling to the host and back for the percentiles), or None when PyTorch
    is not importable (the stack then lives on the host, as upstream's)."""
    try:
        import torch
    except ImportError:  # pragma: no cover - torch is part of the target image
        return None
    if not torch.cuda.is_available() or n_local == 0:
        return None
    return torch.empty((n_local, width), dtype=torch.float64, device=torch.device("cuda", get_context().device))


def bt_ci(confidence_level, n_bootstrap, n_u, meth_f, counts, ref, init_option, n_iter1, n_iter2, tol, header,
          outdir, samples, purity, seed, materialize=True, _observe=None):
    """bootstrap.py:10-93 -> [proportions CI DataFrame, (profile CI DataFrame)]; writes the two CSVs.
    materialize=False (the CLI, which ignores the return value) skips building the N-row DataFrame of tuples for the
    profile intervals: the CSV is written by the library and the second result is the (lower, upper) array pair.
    _observe(i, seed_i, row_indices, solver) is called after replicate i's solve (tests look at the replicate through it)."""
    purity_frac = None
    if purity:
        # upstream quirk kept: bt_ci takes the raw percentages and uses p / 100 (bootstrap.py:18) while main()
        # passes 1 - p / 100 to the point estimate (demethify.py:77)
        purity_frac = np.array(purity) / 100.0
    supervised = n_u == 0
    a = 1 - confidence_level / 100
    lower_percentile = 100 * (a / 2)
    upper_percentile = 100 * (1 - (a / 2))
    n_rows, n_samples = meth_f.shape
    n_ct = ref.shape[1]

    rank, world, _ = shard.dist_state()
    seeds = bootstrap_seed_sequence(seed, n_bootstrap)
    local = []
    if supervised:
        for i in shard.my_items(n_bootstrap, rank, world):
            idx = bootstrap_row_indices(seeds[i], n_rows)
            mf, ct, rf = meth_f[idx], counts[idx], ref[idx]
            props = np.concatenate([wls_intercept(ct[:, k:k + 1] * mf[:, k:k + 1], ct[:, k:k + 1], rf)
                                    for k in range(n_samples)], axis=1)
            local.append((i, (None, props)))
    else:
        # The host work of the replicates ahead (MT19937 index draw of N rows; uniform N x n_u + Dirichlet init: ~10 + ~13 ms
        # at 1e6 rows) runs on worker threads while the GPU gathers and solves replicate i.  The row draw has its own
        # RandomState(seed_i) (what sklearn's resample does); the init re-seeds numpy's GLOBAL stream with seed_i
        # (deconvolution.py:41), so the draws do not depend on the order the threads run in -- but only ONE thread may
        # run initialisers at a time.  When the init does not look at the resampled data ('uniform_' and the other
        # data-free options) the two draws are independent and get a thread each; otherwise one thread does both.
        from .staging import Prefetcher, indices_to_device, reserve, to_device

        ctx = get_context()
        reserve(((n_rows, n_u), (n_ct + n_u, n_samples), (n_rows,)), count=2)

        mine = shard.my_items(n_bootstrap, rank, world)
        needs_data = init_option == "uniform"

        def draw_init(i, idx=None):
            mf = meth_f[idx] if needs_data else np.broadcast_to(meth_f[:1], meth_f.shape)
            ct = counts[idx] if needs_data else np.broadcast_to(counts[:1], counts.shape)
            rf = ref[idx] if needs_data else np.broadcast_to(ref[:1], ref.shape)
            if purity_frac is not None:
                u0, _, a0 = init_BSSMF_md_p(init_option, mf, ct, rf, n_u, purity_frac, seed=seeds[i], rb_alg=wls_intercept, _stack=False)
            else:
                u0, _, a0 = init_BSSMF_md(init_option, mf, ct, rf, n_u, rb_alg=wls_intercept, seed=seeds[i], _stack=False)
            return to_device((u0, a0), ctx)  # page-locked copy + upload on the context's copy stream, here in the worker

        def draw_rows(i):
            # (the row indices travel to HBM from here too: 8 MB from pageable memory took the solving thread ~1 ms per replicate)
            idx = bootstrap_row_indices(seeds[i], n_rows)
            return idx, indices_to_device(idx, ctx)

        def draw_both(i):
            idx, idx_dev = draw_rows(i)
            return (idx, idx_dev) + tuple(draw_init(i, idx))

        if needs_data:
            feeds = (Prefetcher(mine, draw_both, depth=2, workers=1),)
        else:
            feeds = (Prefetcher(mine, draw_rows, depth=2, workers=1), Prefetcher(mine, draw_init, depth=2, workers=1))
        u_stack = _device_stack(len(mine), n_rows * n_u)  # replicate profiles stay in HBM when torch is there
        try:
            with Problem(ctx, meth_f, counts, ref) as full:
                for j, parts in enumerate(zip(*feeds)):
                    if needs_data:
                        (i, (idx, idx_dev, u0, a0)), = parts
                    else:
                        (i, (idx, idx_dev)), (i2, (u0, a0)) = parts
                        assert i == i2
                    with full.gather(idx_dev) as resampled, Solver(resampled, u0, a0, L.DMF_MODE_PARTIAL) as s:
                        if purity_frac is not None:
                            s.set_purity(purity_frac)
                        s.step(n_iter1, n_iter2, tol)
                        if _observe is not None:
                            _observe(i, seeds[i], idx, s)
                        if u_stack is not None:
                            s.copy_u_to(u_stack[j])
                            local.append((i, (None, s.get_alpha())))
                        else:
                            u, alpha, _, _ = s.get()
                            local.append((i, (u, alpha)))
        finally:
            for f in feeds:
                f.close()
    # proportions (K x S per replicate, KB-sized) go to every rank
    merged = shard.gather_objects([(i, pa) for i, (_, pa) in local])
    props_stack = np.stack([pa for _, pa in merged])  # (B, K, S)
    q = [lower_percentile, upper_percentile]

    results = []
    if supervised:
        # (no device in play on this branch: NNLS per sample on the host, as upstream)
        lower_p, upper_p = np.percentile(props_stack, q, axis=0)
    else:
        lower_p, upper_p = get_context().percentile_axis0(props_stack, q)
    unknown_header = [] if supervised else ["unknown_cell_" + str(i + 1) for i in range(n_u)]
    cell_types = header + unknown_header
    table = {f"Sample_{i + 1}": [(lower_p[k, i], upper_p[k, i]) for k in range(n_ct + n_u)]
             for i in range(n_samples)}
    proportions_df = pd.DataFrame(table, index=cell_types)
    proportions_df.columns = samples
    proportions_df.index.name = "Cell Type"
    if rank == 0:
        proportions_df.to_csv(outdir + "/confidence_interval_celltypes_proportions.csv", index=True)
    results.append(proportions_df)

    if not supervised:
        # Profile estimates: B x N x n_u doubles (16 GB at 500 x 1e6 x 4).  Each rank holds the replicates it
        # ran; one all-to-all re-partitions them by CpG range and every rank takes the percentiles of its
        # range on its GPU (shard.percentile_over_replicates); rank 0 writes the CSV.
        if u_stack is not None:
            local_u = u_stack
        else:
            local_u = (np.stack([pu.reshape(-1) for _, (pu, _) in local]) if local else np.empty((0, n_rows * n_u)))
        bounds = shard.percentile_over_replicates(local_u, n_bootstrap, q, get_context().percentile_axis0)
        if bounds is not None:
            lower_u, upper_u = (b.reshape(n_rows, n_u) for b in bounds)  # by resampled position, as upstream
            path = outdir + "/confidence_interval_methylation_estimate.csv"
            if not _write_interval_csv(path, unknown_header, lower_u, upper_u) or materialize:
                # upstream's way (bootstrap.py:85-91): N Python tuples per column through DataFrame.to_csv, ~20 s at 1e6 rows
                ref_estimate_df = pd.DataFrame({unknown_header[k]: [(lower_u[j, k], upper_u[j, k]) for j in range(n_rows)]
                                                for k in range(n_u)})
                if not os.path.exists(path):
                    ref_estimate_df.to_csv(path, index=False)
                results.append(ref_estimate_df)
            else:
                results.append((lower_u, upper_u))
    return results


def _write_interval_csv(path, columns, lower, upper) -> bool:
    """The (lower, upper) table as DataFrame.to_csv writes a DataFrame of tuples, by the library's writer
    (dmf_write_interval_csv: byte-identical, tests/test_host.py).  False if a column name would need CSV quoting."""
    import ctypes as C

    if any(ch in name for name in columns for ch in ',"\r\n'):
        return False
    lib = L.load()
    lower = np.ascontiguousarray(lower, dtype=np.float64)
    upper = np.ascontiguousarray(upper, dtype=np.float64)
    numpy_scalar_repr = int(str((np.float64(0.5),)).startswith("(np.float64("))  # numpy >= 2 prints scalars that way
    try:
        os.remove(path)
    except OSError:
        pass
    rc = lib.dmf_write_interval_csv(path.encode(), ",".join(columns).encode(), lower.ctypes.data_as(C.c_void_p),
                                    upper.ctypes.data_as(C.c_void_p), lower.shape[0], lower.shape[1], numpy_scalar_repr,
                                    min(16, os.cpu_count() or 1))
    return rc == 0
