#!/usr/bin/env python3
"""The three-node chain (bench.py sweep_pipeline_threads) with per-thread accounting: time in each node's calls against time
blocked on the queues.  python3 tools/node_threads_trace.py [rings] [sweeps]"""
import importlib, os, queue, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
rings = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lo, hi = (-15.0, 15.0) if rings == 16 else (-24.9, 2.0)
world = synth.World(half_extent=175.0)
raws = []
for k in range(sweeps):
    gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
    _, _, _, cloud, _ = synth.make_scan(world, rings, 1800, gt_pose=gt, seed=300 + k, full=True)
    ring = np.floor(cloud[:, 3]).astype(np.int64)
    raws.append(cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))])
ctx_r, ctx_o, ctx_m = pkg.Context(0), pkg.Context(0), pkg.Context(0)
odo = pkg.DeviceLaserOdometry(ctx_o, publish_buffers=8)
mapper = pkg.LaserMapping(ctx_m, cube_dims=(21, 21, 11))
sr = pkg.scan_registration
q1, q2, pool = queue.Queue(maxsize=2), queue.Queue(maxsize=2), queue.Queue()
for _ in range(5):
    pool.put(sr.FeatureSet(ctx_r))
warm = 5
busy = {"registration": [], "odometry": [], "mapping": []}
stamps = {}


def registration():
    for k, raw in enumerate(raws):
        if k == warm:
            stamps["t0"] = time.perf_counter()
        t = time.perf_counter()
        reg, rr = sr.multiscan_register(ctx_r, raw, lo, hi, rings)
        t1 = time.perf_counter()
        f = pool.get()
        t2 = time.perf_counter()
        sr.extract_features_dev(ctx_r, reg, rr, f)
        t3 = time.perf_counter()
        q1.put(f)
        busy["registration"].append((t1 - t) + (t3 - t2))
    q1.put(None)


def odometry():
    while True:
        f = q1.get()
        if f is None:
            break
        t = time.perf_counter()
        T = odo.process(f)
        busy["odometry"].append(time.perf_counter() - t)
        pool.put(f)
        if T is not None:
            q2.put((odo.last_corner, odo.last_surf, T))
    q2.put(None)


def mapping():
    while True:
        item = q2.get()
        if item is None:
            break
        t = time.perf_counter()
        mapper.process(*item)
        busy["mapping"].append(time.perf_counter() - t)
    stamps["t1"] = time.perf_counter()


th = [threading.Thread(target=f) for f in (registration, odometry, mapping)]
for t in th:
    t.start()
for t in th:
    t.join()
n = sweeps - warm
print("%d rings: %.3f ms per sweep over %d sweeps" % (rings, 1e3 * (stamps["t1"] - stamps["t0"]) / n, n))
for k, v in busy.items():
    v = np.array(v[warm:]) * 1e3
    print("  %-12s busy per sweep: mean %.3f ms  p50 %.3f  p90 %.3f  max %.3f" % (k, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.max()))
