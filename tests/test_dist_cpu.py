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


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` outside a launcher starts 2 ranks itself (before anything touches the GPU)
    and rank 0 reports n_gpus == 2; under a launcher whose world size differs from --gpus it refuses."""
    import json
    import subprocess
    env = dict(os.environ, LSLAM_BENCH_DRY_RUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["ranks_counted"] == 2
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "--gpus 4" in bad.stderr
