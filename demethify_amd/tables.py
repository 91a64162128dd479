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
Single-process runs first try the library's own columnar reader (csrc/dmf_tables.hip: memory-mapped, multi-threaded,
the two columns parsed straight into page-locked (N x S) matrices with the arithmetic of pandas' C parser, so the
values are bit-identical); any file that reader declines (NA spellings, quotes, fractional coverage, --fillna) sends
the whole input through the pandas path.
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


# Set once this process has asked the library for page-locked memory (dmf_host_alloc brings up the HIP runtime: its
# threads, its locks, the KFD file descriptor): from then on no worker PROCESS is forked here any more.
_hip_runtime_up = False


def _read_many(paths, bedmethyl, fillna):
    if not paths:
        return []
    workers = io_workers(paths)
    jobs = [(p, bedmethyl, fillna) for p in paths]
    if workers == 1:
        return [_read_one(j) for j in jobs]
    if _hip_runtime_up:
        # the native reader declined a file AFTER it had allocated its page-locked matrices (an NA in mid-file, a
        # fractional coverage): threads instead of forked processes (pandas' C parser releases the GIL)
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=workers) as pool:
            return list(pool.map(_read_one, jobs))
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


def _host_matrix(shape, dtype):
    """(N, S) array in page-locked host memory when the library can provide it (the H2D upload of the problem then
    needs no staging copy), plain numpy memory otherwise."""
    import ctypes as C
    import weakref

    from . import _lib as L

    lib = L.load()
    n_bytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    pinned = C.c_int(0)
    global _hip_runtime_up
    if n_bytes:
        _hip_runtime_up = True  # (whether or not the allocation ends up page-locked: the library has asked the runtime)
    ptr = lib.dmf_host_alloc(n_bytes, C.byref(pinned)) if n_bytes else None
    if not ptr:
        return np.empty(shape, dtype=dtype)
    buf = (C.c_char * n_bytes).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    weakref.finalize(buf, lib.dmf_host_free, ptr, pinned.value)  # released with the last view of the buffer
    return arr


def read_samples_native(paths, bedmethyl: bool):
    """All sample files through the library's columnar reader -> (meth_f, counts) or None if any file needs pandas."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from . import _lib as L

    try:
        lib = L.load()
    except (ImportError, OSError):
        return None
    sep = b"\t" if bedmethyl else b","
    scans = []
    for path in paths:
        n, c_pm, c_cov, n_cols = C.c_int64(), C.c_int(), C.c_int(), C.c_int()
        st = lib.dmf_table_scan(os.fsencode(path), sep, C.byref(n), C.byref(c_pm), C.byref(c_cov), C.byref(n_cols))
        if st != L.DMF_OK or c_pm.value < 0:
            return None
        single = (not bedmethyl) and n_cols.value == 1
        if c_cov.value < 0 and not single:
            return None  # pandas raises KeyError('valid_coverage') there: let it
        scans.append((n.value, c_pm.value, c_cov.value))
    n_rows = scans[0][0]
    if n_rows == 0 or any(s[0] != n_rows for s in scans):
        return None  # pandas' error message for ragged inputs
    S = len(paths)
    meth_f = _host_matrix((n_rows, S), np.float64)
    counts = _host_matrix((n_rows, S), np.int64)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        cores = os.cpu_count() or 1
    files_at_once = max(1, min(S, cores))
    threads_per_file = max(1, cores // files_at_once)

    def one(k):
        _, c_pm, c_cov = scans[k]
        f_ptr = meth_f.ctypes.data + 8 * k
        c_ptr = counts.ctypes.data + 8 * k
        st = lib.dmf_table_read(os.fsencode(paths[k]), sep, c_pm, c_cov, n_rows, C.c_void_p(f_ptr), S,
                                100.0 if bedmethyl else 1.0, C.c_void_p(c_ptr), S, threads_per_file)
        if st == L.DMF_OK and c_cov < 0:
            counts[:, k] = 1  # a single-column csv: coverage 1, demethify.py:137-138
        return st

    with ThreadPoolExecutor(max_workers=files_at_once) as pool:  # ctypes releases the GIL during the call
        if any(st != L.DMF_OK for st in pool.map(one, range(S))):
            return None
    return meth_f, counts


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
    if world == 1 and not fillna and os.environ.get("DEMETHIFY_PANDAS_READER") != "1":
        native = read_samples_native(paths, bedmethyl)
        if native is not None:
            return ("native", native)
    mine = shard.my_items(len(paths), rank, world)
    return _read_many([paths[i] for i in mine], bedmethyl, fillna)


def read_samples(paths, bedmethyl: bool, fillna: bool, parsed=None):
    """All sample files -> (meth_f, counts), both (N, S) in C order, equal to upstream's column_stack of the
    per-file columns (same dtypes: counts stay int64 unless a file forces float).  ``parsed`` = the result of an
    earlier parse_share() of the same arguments on this rank (the CLI parses before it initialises the GPU)."""
    rank, world, dev = shard.dist_state()
    if parsed is None and world == 1:
        parsed = parse_share(paths, bedmethyl, fillna)
    if isinstance(parsed, tuple) and len(parsed) == 2 and parsed[0] == "native":
        return parsed[1]
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
