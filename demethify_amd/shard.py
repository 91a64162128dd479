"""Multi-GPU sharding of the loops that call the solver many times.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  The units (random restarts, bootstrap resamples, model-selection candidates) are
independent, so work item k simply goes to rank k mod world and the exchanges on the solve path are
KB-sized: one all-reduce(min) over the cost vector to pick the winning restart, and gathers of the
per-item results.  No collective touches the N x S data path.  The one sizeable exchange is in the
bootstrap's post-processing: the per-replicate profile estimates are re-partitioned from "by replicate"
to "by CpG range" with one all-to-all, so that every rank can take the percentiles of its own range.
"""
from __future__ import annotations

import numpy as np


def dist_state():
    """(rank, world, device for collectives) — (0, 1, None) when torch.distributed is not initialised."""
    try:
        import torch
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return 0, 1, None
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1, None
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    return dist.get_rank(), dist.get_world_size(), dev


def my_items(n_items, rank=None, world=None):
    """Indices of the work items this rank owns: k = rank, rank + world, ..."""
    if rank is None:
        rank, world, _ = dist_state()
    return list(range(rank, n_items, world))


def allreduce_min_vector(local: dict, n_items: int) -> np.ndarray:
    """Every rank fills its own slots of an n_items float64 vector (+inf elsewhere); one
    all-reduce(min) gives every rank the full vector."""
    rank, world, dev = dist_state()
    vec = np.full(n_items, np.inf, dtype=np.float64)
    for k, c in local.items():
        vec[k] = c
    if world == 1:
        return vec
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(vec).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t.cpu().numpy()


def argmin_first(vec: np.ndarray) -> int:
    """Lowest index among the minima == the reference's strict '<' running minimum
    (demethify/demethify.py:170,200: the first restart wins ties)."""
    return int(np.argmin(vec))


def pick_min_cost(cost: float, rank: int, world: int, dev=None):
    """bench.py helper: one restart per rank -> (winning rank, its cost)."""
    vec = allreduce_min_vector({rank: cost}, world)
    k = argmin_first(vec)
    return k, float(vec[k])


def broadcast_arrays(arrays, src: int):
    """Broadcast a tuple of float64 numpy arrays (shapes known on every rank) from rank src."""
    rank, world, dev = dist_state()
    if world == 1:
        return arrays
    import torch
    import torch.distributed as dist

    out = []
    for a in arrays:
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        else:  # (the receivers' buffers are allocated where the collective runs: nothing to upload)
            t = torch.empty(tuple(np.shape(a)), dtype=torch.float64, device=dev)
        dist.broadcast(t, src=src)
        out.append(t.cpu().numpy())
    return tuple(out)


def broadcast_winner(solver, u_shape, alpha_shape, owner: int, host_ranks=(0,)):
    """The winning restart's iterate from the rank that holds its solver to every rank.  One rank: the solver's
    iterate as host arrays.  More: the profiles leave the owner's solver device to device (Solver.copy_u_to), travel
    as CUDA tensors (RCCL broadcast, no host staging on any rank) and only the ranks in ``host_ranks`` -- the writer of
    the output files -- copy them to the host; the others return (None, None).  ``solver`` is None on non-owners."""
    rank, world, dev = dist_state()
    if world == 1:
        u, alpha, _, _ = solver.get()
        return u, alpha
    import torch
    import torch.distributed as dist

    if dev is None or dev.type != "cuda":  # gloo (CPU tests, the one-GPU rehearsal): host arrays
        payload = ((lambda g: (g[0], g[1]))(solver.get()) if rank == owner
                   else (np.empty(u_shape), np.empty(alpha_shape)))
        return broadcast_arrays(payload, owner)
    return _broadcast_winner_device(solver, u_shape, alpha_shape, owner, rank, dev, host_ranks)


def _broadcast_winner_device(solver, u_shape, alpha_shape, owner, rank, dev, host_ranks):
    import torch
    import torch.distributed as dist

    u_t = torch.empty(u_shape, dtype=torch.float64, device=dev)
    a_t = torch.empty(alpha_shape, dtype=torch.float64, device=dev)
    if rank == owner:
        solver.copy_u_to(u_t)  # (synchronous on the solver's stream)
        a_t.copy_(torch.from_numpy(solver.get_alpha()))
    dist.broadcast(u_t, src=owner)
    dist.broadcast(a_t, src=owner)
    if rank in host_ranks:
        return u_t.cpu().numpy(), a_t.cpu().numpy()
    torch.cuda.current_stream().synchronize()
    return None, None


def gather_objects(local, root_only=False):
    """Gather of picklable per-rank results (lists of (k, payload)); returns the merged, k-sorted list on
    every rank, or with ``root_only`` on rank 0 alone (None elsewhere): the bootstrap's profile stacks are
    n_bootstrap x N x n_u doubles, which only the rank that writes the CSV needs."""
    rank, world, dev = dist_state()
    if world == 1:
        return sorted(local, key=lambda kv: kv[0])
    import torch.distributed as dist

    if root_only:
        bucket = [None] * world if rank == 0 else None
        dist.gather_object(local, bucket, dst=0)
        if rank != 0:
            return None
    else:
        bucket = [None] * world
        dist.all_gather_object(bucket, local)
    merged = [kv for part in bucket for kv in part]
    return sorted(merged, key=lambda kv: kv[0])


def split_positions(m, world):
    """Contiguous, near-equal ranges of m positions, one per rank: [(begin, end)] * world."""
    base, extra = divmod(m, world)
    out, a = [], 0
    for r in range(world):
        b = a + base + (1 if r < extra else 0)
        out.append((a, b))
        a = b
    return out


def percentile_over_replicates(local_stack, n_items, q, percentile_fn):
    """Percentiles over ALL replicates of a stack that is sharded by replicate (bootstrap.py:75-78).

    ``local_stack``: (n_local, m) float64 array, the rows of the replicates this rank ran (rank r owns
    replicates r, r + world, ... of ``n_items``); ``percentile_fn(x, q)`` -> (len(q), m') takes the percentiles
    over axis 0 of a (n, m') array or tensor (the HIP kernel in the product, numpy in the CPU tests).
    One all-to-all turns the "by replicate" partition into a "by position range" one (rank r receives the
    columns of range r from every rank: n_items x m_r values), each rank reduces its range, and rank 0
    collects the (len(q), m) result (None elsewhere)."""
    rank, world, dev = dist_state()
    on_device = type(local_stack).__module__.startswith("torch")  # the bootstrap's HBM-resident stack
    if world == 1:
        out = percentile_fn(local_stack if on_device else np.ascontiguousarray(local_stack), q)
        return out.cpu().numpy() if hasattr(out, "cpu") else np.asarray(out)
    import torch
    import torch.distributed as dist

    if on_device:
        local = local_stack if dev.type == "cuda" else local_stack.cpu()
    else:
        local = torch.as_tensor(np.ascontiguousarray(local_stack, dtype=np.float64)).to(dev)
    n_local, m = local.shape
    counts = [len(my_items(n_items, r, world)) for r in range(world)]
    assert counts[rank] == n_local, (counts, rank, n_local)
    ranges = split_positions(m, world)
    send = torch.cat([local[:, a:b].reshape(-1) for a, b in ranges])
    a, b = ranges[rank]
    recv = torch.empty(n_items * (b - a), dtype=torch.float64, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=[c * (b - a) for c in counts],
                           input_split_sizes=[n_local * (hi - lo) for lo, hi in ranges])
    del send, local
    # rank r's pieces arrive grouped by source rank; replicate i ran on rank i mod world: back to replicate order is
    # not needed for a percentile (a permutation of axis 0)
    mine = percentile_fn(recv.view(n_items, b - a), q) if b > a else np.empty((len(np.atleast_1d(q)), 0))
    mine = mine.cpu().numpy() if hasattr(mine, "cpu") else np.asarray(mine)
    parts = gather_objects([(rank, mine)], root_only=True)
    if parts is None:
        return None
    return np.concatenate([part for _, part in parts], axis=1)


def restart_seed(seed, k):
    """Seed of restart k.  Upstream passes the SAME seed to every restart (demethify.py:168,196),
    which makes ``--restart r`` idempotent; here restart 0 keeps the upstream stream bit-for-bit
    and restart k > 0 uses seed + k (the convention upstream uses for CCC restarts, ic.py:196)."""
    if seed is None or k == 0:
        return seed
    if isinstance(seed, (list, tuple)):
        return [int(seed[0]) + k]
    return seed + k


def sharded_restarts(n_restarts, solve_one, shapes, prepare=None, solve_end=None):
    """Run restarts k = 0..n_restarts-1 across the ranks and return the min-cost one everywhere.

    With ``solve_end`` the restart is taken in two halves: solve_one(k, best_cost[, prepared]) returns a HANDLE (a solver
    whose streaming cost is on its way: Solver.cost_begin) and solve_end(handle, best_cost) -> (u, alpha, cost) is called
    once the NEXT restart has been set up and solved -- the GPU takes restart k's cost while the host prepares k + 1.

    solve_one(k, best_cost) -> (u, alpha, cost) runs restart k on this rank's GPU; it may return (None, None,
    cost) when cost >= best_cost (this rank's running minimum: the iterate of a restart that cannot win need not
    leave the device).  With ``prepare``, prepare(k) -- the restart's initialisation, possibly already uploaded
    (staging.to_device) -- runs in ONE worker thread (the initialisers use numpy's global generator) up to two restarts ahead of the GPU and solve_one is called as
    solve_one(k, best_cost, prepared).  ``shapes`` = (u.shape, alpha.shape) lets non-owner ranks allocate the
    broadcast buffers.  Returns (u, alpha, best_k, cost_vector)."""
    rank, world, _ = dist_state()
    local_costs, keep = {}, {}
    mine = list(my_items(n_restarts, rank, world))
    if prepare is not None:
        from .staging import Prefetcher

        feed = Prefetcher(mine, prepare, depth=2, workers=1)  # one worker: the reference's initialisers seed numpy's GLOBAL generator
        source = iter(feed)
    else:
        feed, source = None, ((k, None) for k in mine)
    def record(k, result):
        nonlocal keep
        u, alpha, cost = result
        local_costs[k] = cost
        # keep only the local best: strict '<' so that the lowest k wins ties locally as well
        if not keep or cost < keep["cost"]:
            keep = {"k": k, "u": u, "alpha": alpha, "cost": cost}

    try:
        waiting = None
        for k, prepared in source:
            best_cost = keep["cost"] if keep else float("inf")
            out = solve_one(k, best_cost) if prepare is None else solve_one(k, best_cost, prepared)
            if solve_end is None:
                record(k, out)
                continue
            if waiting is not None:  # (restart k ran against the minimum BEFORE waiting's cost was known: solve_end decides)
                record(waiting[0], solve_end(waiting[1], keep["cost"] if keep else float("inf")))
            waiting = (k, out)
        if waiting is not None:
            record(waiting[0], solve_end(waiting[1], keep["cost"] if keep else float("inf")))
    finally:
        if feed is not None:
            feed.close()
    costs = allreduce_min_vector(local_costs, n_restarts)
    best_k = argmin_first(costs)
    owner = best_k % world
    if rank == owner:
        assert keep["k"] == best_k and keep["u"] is not None
        payload = (keep["u"], keep["alpha"])
    else:
        payload = (np.empty(shapes[0]), np.empty(shapes[1]))
    u, alpha = broadcast_arrays(payload, owner)
    return u, alpha, best_k, costs
