"""Input tables at scale (demethify/demethify.py:103-143).

Upstream reads one bedmethyl / csv file per sample with ``pd.read_csv`` in a Python loop and stacks the two
columns it needs.  At 1e6 CpG x 256 samples that is ~20 GB of text through a single-threaded parser, which
costs far more wall-clock than the solve.  Here the same parser (so the parsed values are bit-identical) runs
  * on the two needed columns only (``usecols``),
  * on several files at once, in forked worker processes once the input is large (threads do not help: the
    parser's type conversion holds the GIL; measured 1.1x with 8 threads against 5x+ with 8 processes),
  * and, with torch.distributed initialised, on a 1 / world share of the files per rank, the parsed columns being
    exchanged as tensors (one broadcast per rank: RCCL over xGMI with the nccl backend) instead of every rank
    parsing every file.
"""
from __future__ import annotations

import multiprocessing
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pandas as pd

from . import shard

_NEEDED = ("percent_modified", "valid_coverage")


_POOL_MIN_BYTES = 64 << 20  # below this a pool costs more than it saves


def io_workers(paths) -> int:
    """Worker processes for the file loop: DEMETHIFY_IO_WORKERS if set, else 1 for small inputs and the cores
    this process may use (shared between the ranks of a node) for large ones."""
    env = os.environ.get("DEMETHIFY_IO_WORKERS")
    if env:
        return max(1, min(int(env), len(paths)))
    if sum(os.path.getsize(p) for p in paths) < _POOL_MIN_BYTES:
        return 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover - non-Linux
        cores = os.cpu_count() or 1
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    return max(1, min(cores // local_world, len(paths)))


def read_sample(path, bedmethyl: bool, fillna: bool):
    """One sample file -> (methylation fraction, coverage) columns, as demethify.py:112-118 (bedmethyl: tab
    separated, percent / 100) and :133-140 (csv: fractions; a single-column file gets coverage 1)."""
    sep = "\t" if bedmethyl else ","
    columns = list(pd.read_csv(path, sep=sep, nrows=0).columns)
    if "percent_modified" not in columns:
        raise KeyError("percent_modified")
    single = (not bedmethyl) and len(columns) == 1
    if not single and "valid_coverage" not in columns:
        raise KeyError("valid_coverage")
    temp = pd.read_csv(path, sep=sep, usecols=[c for c in _NEEDED if c in columns])
    if single:
        temp["valid_coverage"] = 1
    if fillna:
        temp = temp.fillna(0)
    freq = temp["percent_modified"].values
    return (freq / 100 if bedmethyl else freq), temp["valid_coverage"].values


def _read_one(job):
    return read_sample(*job)


def _read_many(paths, bedmethyl, fillna):
    if not paths:
        return []
    workers = io_workers(paths)
    jobs = [(p, bedmethyl, fillna) for p in paths]
    if workers == 1:
        return [_read_one(j) for j in jobs]
    # fork, not spawn: the workers only parse text and hand two arrays back; they never touch the GPU runtime,
    # and no new program is exec'ed from a process that may already hold a device
    with ProcessPoolExecutor(max_workers=workers, mp_context=multiprocessing.get_context("fork")) as pool:
        return list(pool.map(_read_one, jobs))


def _stack(columns, n_rows, dtype):
    out = np.empty((n_rows, len(columns)), dtype=dtype)  # C order, as np.column_stack gives
    for k, col in enumerate(columns):
        if col.shape[0] != n_rows:
            raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
        out[:, k] = col
    return out


def env_rank_world():
    """(rank, world) of a torch.distributed.run launch, from the environment: known before the process group (and
    with it the GPU runtime) exists."""
    return int(os.environ.get("RANK", "0")), max(1, int(os.environ.get("WORLD_SIZE", "1")))


def parse_share(paths, bedmethyl: bool, fillna: bool):
    """Parse this rank's 1 / world share of the sample files.  Needs RANK / WORLD_SIZE only, so the CLI calls it
    BEFORE the process group and the device are initialised: the worker pool is then always forked from a process
    that has not touched the GPU (a fork of a process that holds a HIP runtime and RCCL threads inherits locked
    mutexes and the KFD file descriptor)."""
    rank, world = env_rank_world()
    mine = shard.my_items(len(paths), rank, world)
    return _read_many([paths[i] for i in mine], bedmethyl, fillna)


def read_samples(paths, bedmethyl: bool, fillna: bool, parsed=None):
    """All sample files -> (meth_f, counts), both (N, S) in C order, equal to upstream's column_stack of the
    per-file columns (same dtypes: counts stay int64 unless a file forces float).  ``parsed`` = the result of an
    earlier parse_share() of the same arguments on this rank (the CLI parses before it initialises the GPU)."""
    rank, world, dev = shard.dist_state()
    if parsed is None:
        mine = shard.my_items(len(paths), rank, world)
        parsed = _read_many([paths[i] for i in mine], bedmethyl, fillna)
    elif world > 1:
        assert (rank, world) == env_rank_world(), "parse_share() ran under a different rank layout"
    if world == 1:
        freqs = [f for f, _ in parsed]
        counts = [c for _, c in parsed]
        n_rows = freqs[0].shape[0]
        return (_stack(freqs, n_rows, np.result_type(*freqs)), _stack(counts, n_rows, np.result_type(*counts)))

    import torch
    import torch.distributed as dist

    # agree on the row count and on whether any coverage column came out as float (NaNs without --fillna)
    n_local = parsed[0][0].shape[0] if parsed else -1
    float_counts = any(c.dtype.kind == "f" for _, c in parsed)
    meta = torch.tensor([n_local, int(float_counts)], dtype=torch.int64, device=dev)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    n_rows, float_counts = int(meta[0]), bool(meta[1])
    if any(f.shape[0] != n_rows for f, _ in parsed):
        raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    count_dtype = np.float64 if float_counts else np.int64
    meth_f = np.empty((n_rows, len(paths)), dtype=np.float64)
    counts = np.empty((n_rows, len(paths)), dtype=count_dtype)
    for src in range(world):
        cols = shard.my_items(len(paths), src, world)
        if not cols:
            continue
        if src == rank:
            f_block = torch.from_numpy(np.stack([np.asarray(f, dtype=np.float64) for f, _ in parsed])).to(dev)
            c_block = torch.from_numpy(np.stack([np.asarray(c, dtype=count_dtype) for _, c in parsed])).to(dev)
        else:
            f_block = torch.empty((len(cols), n_rows), dtype=torch.float64, device=dev)
            c_block = torch.empty((len(cols), n_rows), dtype=torch.from_numpy(np.empty(0, count_dtype)).dtype,
                                  device=dev)
        dist.broadcast(f_block, src=src)
        dist.broadcast(c_block, src=src)
        meth_f[:, cols] = f_block.cpu().numpy().T
        counts[:, cols] = c_block.cpu().numpy().T
    return meth_f, counts
