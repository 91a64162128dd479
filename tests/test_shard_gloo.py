"""Multi-process (gloo, world_size 2) tests of the sharding layer on CPU: the restart pick, the cost
all-reduce and the object gather must give every rank the same answer as a serial run.  The solver
is injected (CPU oracle on a toy problem) because the product solver needs a GPU."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy():
    from oracle import solver as osol

    V, D, Rt = osol.synthetic_problem(200, 6, 3, 1, seed=4, depth=15)
    return V, D, Rt


def _solve_restart(k, best_cost=float("inf")):
    from demethify_amd.shard import restart_seed
    from oracle import drivers as odrv
    from oracle import solver as osol

    V, D, Rt = _toy()
    u, R, alpha = odrv.run_one(V, D, Rt, 1, "uniform_", restart_seed(1, k), 30, 5, 1e-3,
                               project=osol.simplex_project_columns_fast)
    cost = float(osol.weighted_cost(V, R, alpha, D))
    if not cost < best_cost:  # the product leaves such an iterate on the device
        return None, None, cost
    return u, alpha, cost


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from demethify_amd import shard

    assert shard.dist_state()[:2] == (rank, world)
    # 1. cost vector all-reduce + first-minimum pick
    vec = shard.allreduce_min_vector({k: float((k * 7) % 5) for k in shard.my_items(9)}, 9)
    assert np.array_equal(vec, np.array([(k * 7) % 5 for k in range(9)], dtype=float))
    assert shard.argmin_first(vec) == 0
    # 2. sharded restarts == serial pick, identical on every rank
    u, alpha, best_k, costs = shard.sharded_restarts(5, _solve_restart, ((200, 1), (4, 6)))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=u, alpha=alpha, best_k=best_k, costs=costs)
    # 3. object gather keeps item order
    merged = shard.gather_objects([(k, k * k) for k in shard.my_items(7)])
    assert merged == [(k, k * k) for k in range(7)]
    rooted = shard.gather_objects([(k, -k) for k in shard.my_items(5)], root_only=True)
    assert (rooted == [(k, -k) for k in range(5)]) if rank == 0 else rooted is None
    # 4. broadcast from a non-zero owner
    payload = (np.full((3, 2), 5.0),) if rank == 1 else (np.empty((3, 2)),)
    (got,) = shard.broadcast_arrays(payload, 1)
    assert np.array_equal(got, np.full((3, 2), 5.0))
    # 4b. the winner's iterate from the rank that holds its solver (host path of broadcast_winner: gloo has no device)
    class _Held:
        def get(self):
            return np.full((6, 2), 3.5), np.full((4, 5), 0.25), 1.0, 7

    wu, wa = shard.broadcast_winner(_Held() if rank == 1 else None, (6, 2), (4, 5), 1)
    assert np.array_equal(wu, np.full((6, 2), 3.5)) and np.array_equal(wa, np.full((4, 5), 0.25))
    # 5. bootstrap post-processing: the stack sharded by replicate is re-partitioned by position range with one
    #    all-to-all; numpy stands in for the HIP percentile kernel here (no GPU in this suite)
    full = np.random.RandomState(3).uniform(size=(7, 37))  # 7 replicates over 2 ranks: 4 + 3; 37 positions: 19 + 18
    mine = full[shard.my_items(7)]
    got = shard.percentile_over_replicates(mine, 7, [2.5, 97.5], lambda x, q: np.percentile(np.asarray(x), q, axis=0))
    if rank == 0:
        assert np.array_equal(got, np.percentile(full, [2.5, 97.5], axis=0))
    else:
        assert got is None
    # fewer replicates than ranks: a rank with an empty share still takes part in the exchange
    one = full[:1]
    got = shard.percentile_over_replicates(one[shard.my_items(1)], 1, [50.0],
                                           lambda x, q: np.percentile(np.asarray(x), q, axis=0))
    assert (np.array_equal(got, one)) if rank == 0 else got is None
    # 6. input tables: each rank parses its share of the sample files, the columns are exchanged as tensors
    from demethify_amd import tables

    paths = [str(ROOT / "tests" / "golden" / "upstream" / "output_gen" / f"sample{i}.bed") for i in range(1, 8)]
    meth_f, counts = tables.read_samples(paths, True, False)
    np.savez(os.path.join(out_dir, f"tables{rank}.npz"), meth_f=meth_f, counts=counts)
    dist.destroy_process_group()


def test_split_positions_cover_everything():
    from demethify_amd.shard import split_positions

    for m, world in [(37, 2), (5, 8), (16, 4), (0, 3)]:
        r = split_positions(m, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == m
        assert all(a1 == b0 for (_, b0), (a1, _) in zip(r, r[1:]))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_sharded_restarts_match_serial(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    serial = [_solve_restart(k) for k in range(5)]
    costs = np.array([c for _, _, c in serial])
    best = int(np.argmin(costs))
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz")
        assert int(z["best_k"]) == best
        assert np.array_equal(z["costs"], costs)
        assert np.array_equal(z["u"], serial[best][0]) and np.array_equal(z["alpha"], serial[best][1])
    # the sharded table read equals upstream's loop (demethify.py:110-119) on every rank, dtypes included
    import pandas as pd

    cols = [pd.read_csv(ROOT / "tests" / "golden" / "upstream" / "output_gen" / f"sample{i}.bed", sep="\t")
            for i in range(1, 8)]
    want_f = np.column_stack([t["percent_modified"].values / 100 for t in cols])
    want_c = np.column_stack([t["valid_coverage"].values for t in cols])
    for rank in range(world):
        z = np.load(tmp_path / f"tables{rank}.npz")
        assert np.array_equal(z["meth_f"], want_f) and np.array_equal(z["counts"], want_c)
        assert z["counts"].dtype == want_c.dtype and z["meth_f"].flags["C_CONTIGUOUS"]


def test_restart_seed_convention():
    from demethify_amd.shard import restart_seed

    assert restart_seed(1, 0) == 1 and restart_seed(1, 3) == 4
    assert restart_seed([5], 0) == [5] and restart_seed([5], 2) == [7]
    assert restart_seed(None, 4) is None


def test_single_process_fallbacks():
    from demethify_amd import shard

    assert shard.dist_state() == (0, 1, None)
    assert shard.my_items(4) == [0, 1, 2, 3]
    assert np.array_equal(shard.allreduce_min_vector({1: 2.0}, 3), np.array([np.inf, 2.0, np.inf]))
    assert shard.gather_objects([(1, "b"), (0, "a")]) == [(0, "a"), (1, "b")]
