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


def _rs_worker(rank, world, port, q):
    """The exchange pattern of the library's row-sharded PCG (csrc/lslam_posegraph.hip, lslam_pg_set_row_shard) under a real
    process group, in numpy: own rows multiplied / updated / preconditioned locally, one scalar all-reduce (p . A p) and one
    all-reduce of the zero-padded z with r . z, r . r behind it per iteration, the direction recomputed by every rank."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import scipy.sparse as sp
    import torch
    import posegraph_oracle as po
    d = importlib.import_module("the-cooper-mapper_amd.dist")
    dist = d.init(backend="gloo")
    g = po.make_graph(n_kf=230, n_loop=700, laps=3)
    H, b, _ = po.linearize(g["init"], g["ij"], g["meas"], g["info"])
    H = H.tolil(); H[:6, :] = 0; H[:, :6] = 0; H[:6, :6] = np.eye(6); H = H.tocsr()
    b = b.copy(); b[:6] = 0
    n_v = len(g["init"])
    A = (H + 1e-3 * H.diagonal().max() * sp.identity(6 * n_v)).tocsr()
    minv = [np.linalg.inv(A[6 * v:6 * v + 6, 6 * v:6 * v + 6].toarray()) for v in range(n_v)]
    prec = lambda r, v0, v1: np.concatenate([minv[v] @ r[6 * (v - v0):6 * (v - v0) + 6] for v in range(v0, v1)])  # noqa: E731

    def allreduce(x):
        t = torch.from_numpy(x)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    v0, v1 = d.row_shard_range(n_v, rank, world)
    r0, r1 = 6 * v0, 6 * v1
    Aown = A[r0:r1]
    n6 = 6 * n_v
    x = np.zeros(r1 - r0)
    r = b[r0:r1].copy()
    z = prec(b, 0, n_v)                      # replicated start: b is complete on every rank
    rz, rr, bb = float(b @ z), float(b @ b), float(b @ b)
    p = np.zeros(n6)
    rz_old, its = 1.0, 0
    for k in range(500):
        if rr <= 1e-20 * bb:
            break
        p = z + (rz / rz_old if k > 0 else 0.0) * p          # every rank, all rows
        qv = Aown @ p
        pq = float(allreduce(np.array([p[r0:r1] @ qv]))[0])  # exchange 1
        alpha = rz / pq
        x += alpha * p[r0:r1]
        r -= alpha * qv
        zo = prec(r, v0, v1)
        X = np.zeros(n6 + 2)
        X[r0:r1] = zo
        X[n6], X[n6 + 1] = r @ zo, r @ r
        X = allreduce(X)                                      # exchange 2: the padded sum IS the gathered z
        z, rz_old, rz, rr = X[:n6], rz, float(X[n6]), float(X[n6 + 1])
        its += 1
    X = np.zeros(n6)
    X[r0:r1] = x
    xfull = allreduce(X)
    ref = None
    if rank == 0:  # the same PCG in one piece
        xs = np.zeros(n6); rs = b.copy(); zs = prec(b, 0, n_v); ps = np.zeros(n6); rzs = float(rs @ zs); rzo = 1.0
        for k in range(500):
            if rs @ rs <= 1e-20 * bb:
                break
            ps = zs + (rzs / rzo if k > 0 else 0.0) * ps
            qs = A @ ps
            al = rzs / float(ps @ qs)
            xs += al * ps; rs -= al * qs
            zs = prec(rs, 0, n_v)
            rzo, rzs = rzs, float(rs @ zs)
        ref = xs
    q.put((rank, (v0, v1), its, xfull, ref))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_pcg_two_ranks():
    """SURVEY 8e row 3 for graphs that do not fit one GPU's persistent solver: ONE damped solve shared by two ranks.  The rows
    partition in whole 21-vertex blocks, both ranks end with the same bits, and the result is the single-process PCG's."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rs_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, (a0, a1), it0, x0, ref), (_, (b0, b1), it1, x1, _) = res
    assert a0 == 0 and a1 == b0 and b1 == 230 and a1 % 21 == 0 and abs((a1 - a0) - (b1 - b0)) <= 21
    assert it0 == it1 and 5 < it0 < 400
    assert np.array_equal(x0, x1)                          # same reduced numbers on both ranks -> same decisions, same bits
    assert np.abs(x0 - ref).max() <= 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("n,world", [(0, 4), (20, 2), (21, 2), (5000, 8), (50001, 8)])
def test_row_shard_range_partitions(n, world):
    sys.path.insert(0, ROOT)
    d = importlib.import_module("the-cooper-mapper_amd.dist")
    parts = [d.row_shard_range(n, r, world) for r in range(world)]
    assert parts[0][0] == 0 and parts[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert all(a % 21 == 0 for a, _ in parts) and all(e % 21 == 0 or e == n for _, e in parts)
    # and it is the library's partition (no GPU needed for this export)
    import ctypes as C
    lib = importlib.import_module("the-cooper-mapper_amd").load_library()
    for r in range(world):
        b, e = C.c_int32(), C.c_int32()
        lib.lslam_pg_row_shard_range(n, r, world, C.byref(b), C.byref(e))
        assert (b.value, e.value) == parts[r]


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
    # the line the driver parses is the LAST stdout line, whole inside its 8 KB tail, with the contract's keys and what a
    # SCALE run would check (per-rank rates, the RCCL rank count slot); the full report went to bench_report.json + stderr
    last = out.stdout.rstrip("\n").splitlines()[-1]
    assert last == line and len(last.encode()) < 8192 and len(out.stdout.encode()) < 8192
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "value_searching_every_point", "ranks"):
        assert k in rec, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms")) <= set(rec["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(rec["cpu_baseline"])
    assert rec["ranks"]["world"] == 2 and rec["ranks"]["per_rank_value"] == [1.0e9, 2.0e9] and "rccl_ranks" in rec["ranks"]
    assert rec["value"] == 3.0e9 and "workload" in rec["config"]
    full = json.loads([l for l in out.stderr.splitlines() if l.startswith("{")][-1])
    assert full["roofline"]["accounting"] and len(json.dumps(full)) > 8192      # the long form exists, elsewhere
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "--gpus 4" in bad.stderr


def test_compact_line_of_a_real_report_fits_the_driver_tail():
    """bench.compact_line on the full reports kept under profiles/ (real legs, real string lengths): parseable, < 8 KB, and the
    numbers the contract names are the report's own."""
    import glob
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    reports = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_bench.json")))
    assert reports
    for path in reports:
        full = json.load(open(path))
        line = bench.compact_line(full)
        assert len(line.encode()) < 8192 and "\n" not in line
        rec = json.loads(line)
        assert abs(rec["value"] - full["value"]) <= 1e-5 * full["value"] and rec["metric"] == full["metric"]
        assert abs(rec["roofline"]["frac"] - full["roofline"]["frac"]) <= 1e-5 and rec["roofline"]["bound"] == full["roofline"]["bound"]
        assert rec["cpu_baseline"]["kind"] == full["cpu_baseline"]["kind"] and rec["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
        assert rec["config"]["workload"] == full["config"]["workload"]
