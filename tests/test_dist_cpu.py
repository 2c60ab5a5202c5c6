"""world_size-2 gloo test of the N>1 plumbing (sharding of independent scans, barrier,
sum/max aggregation) -- the same code path bench.py uses with RCCL."""
import importlib
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    d = importlib.import_module("the-cooper-mapper_amd.dist")
    dist = d.init(backend="gloo")
    b, e = d.shard_range(11, rank, world)
    d.barrier(dist)
    # every scan contributes (index+1) "point residuals"; rank r took (r+1) seconds
    sums, tmax = d.aggregate(dist, [sum(i + 1 for i in range(b, e)), e - b], elapsed=rank + 1.0)
    q.put((rank, b, e, sums, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, b0, e0, s0, t0), (r1, b1, e1, s1, t1) = res
    assert (b0, e0, b1, e1) == (0, 6, 6, 11)          # contiguous, balanced, covers all scans
    assert s0 == s1 == [66.0, 11.0] and t0 == t1 == 2.0  # sum over ranks, max over ranks


@pytest.mark.parametrize("n,world", [(0, 4), (3, 8), (8, 8), (115, 7)])
def test_shard_range_partitions(n, world):
    sys.path.insert(0, ROOT)
    d = importlib.import_module("the-cooper-mapper_amd.dist")
    parts = [d.shard_range(n, r, world) for r in range(world)]
    assert parts[0][0] == 0 and parts[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    sizes = [e - b for b, e in parts]
    assert max(sizes) - min(sizes) <= 1
