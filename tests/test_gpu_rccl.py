"""The raw collectives demethify_amd/shard.py and tables.py issue, on the nccl (= RCCL) backend with world size 1 on
cuda:0: an op RCCL does not support (dtype, reduce op, split sizes) is found here, before an 8-GPU node is.
shard.py's own helpers return early when world == 1, so the collectives are called directly, with the dtypes,
devices and argument shapes shard.py uses (demethify/demethify.py:195-203 is the loop they shard)."""
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"

# shard.allreduce_min_vector: f64 vector, +inf in the slots of other ranks, ReduceOp.MIN
vec = torch.from_numpy(np.array([3.5, np.inf, -2.0, np.inf])).to(dev)
dist.all_reduce(vec, op=dist.ReduceOp.MIN)
assert vec.cpu().tolist() == [3.5, float("inf"), -2.0, float("inf")]

# bench.py: f64 MAX of the elapsed time; tables.read_samples: int64 MAX of (rows, float flag)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
meta = torch.tensor([350, 1], dtype=torch.int64, device=dev)
dist.all_reduce(meta, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25 and meta.cpu().tolist() == [350, 1]

# shard.broadcast_arrays: f64 tensors of the winner's (u, alpha); tables: f64 and int64 column blocks
for arr in (np.random.RandomState(0).uniform(size=(1000, 4)), np.arange(12, dtype=np.int64).reshape(3, 4)):
    b = torch.from_numpy(arr).to(dev)
    dist.broadcast(b, src=0)
    assert np.array_equal(b.cpu().numpy(), arr)

# shard.percentile_over_replicates: ONE all_to_all_single with explicit split sizes, f64
send = torch.arange(35, dtype=torch.float64, device=dev)
recv = torch.empty(35, dtype=torch.float64, device=dev)
dist.all_to_all_single(recv, send, output_split_sizes=[35], input_split_sizes=[35])
assert torch.equal(recv, send)

# shard.gather_objects: picklable per-rank lists (all ranks, or to rank 0 only)
bucket = [None]
dist.all_gather_object(bucket, [(0, 1.5), (2, -3.0)])
assert bucket == [[(0, 1.5), (2, -3.0)]]
rooted = [None]
dist.gather_object([(1, np.ones(3))], rooted, dst=0)
assert rooted[0][0][0] == 1 and np.array_equal(rooted[0][0][1], np.ones(3))

dist.barrier()

# and the product's helpers on top of the initialised group (world == 1 short cuts must still agree)
sys.path.insert(0, os.environ["DMF_ROOT"])
from demethify_amd import shard
assert shard.dist_state()[:2] == (0, 1) and shard.dist_state()[2].type == "cuda"
assert shard.allreduce_min_vector({0: 2.0, 1: 1.0}, 2).tolist() == [2.0, 1.0]

# shard.broadcast_winner's device path (bench.py, N > 1): the iterate leaves the owner's solver as a CUDA tensor,
# is broadcast in HBM and lands on the host on rank 0 only
from oracle import solver as osol
from demethify_amd import _lib as L
from demethify_amd.device import Context, Problem, Solver
V, D, Rt = osol.synthetic_problem(640, 32, 4, 2, seed=23, depth=40)
rs = np.random.RandomState(1)
u0, a0 = rs.uniform(size=(640, 2)), rs.dirichlet(np.ones(6), 32).T.copy()
ctx = Context(0)
with Problem(ctx, V, D, Rt) as p, Solver(p, u0, a0, L.DMF_MODE_PARTIAL) as s:
    s.step(3, 20, 0.0)
    want_u, want_a, _, _ = s.get()
    got_u, got_a = shard._broadcast_winner_device(s, (640, 2), (6, 32), 0, 0, dev, (0,))
    assert np.array_equal(got_u, want_u) and np.array_equal(got_a, want_a)
    assert shard._broadcast_winner_device(s, (640, 2), (6, 32), 0, 0, dev, ()) == (None, None)
ctx.close()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_collectives_world_size_one(tmp_path):
    import os

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DMF_ROOT=str(ROOT))
    proc = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0 and "RCCL_OK" in proc.stdout, (proc.stdout[-1500:], proc.stderr[-3000:])
