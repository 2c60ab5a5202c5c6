#!/usr/bin/env python3
"""Does the batch's second pass (tree searches of the 2.9 % unproven points: 18 % of a sweep at 28 active lanes) hide behind another
half-batch's first pass?  The headline workload (960 cold 64 x 1800 scans against the configs[1] map) as ONE call on one context
against TWO half-batches on two contexts (own streams, own copies of the map) driven from two threads.  No library change: this
is the experiment that decides whether run_batch should deal its chunks to two streams.  GPU box:
    python tools/two_ctx_overlap.py --map-cache build/_mc [--scans 960] [--steps 6]"""
import argparse, importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
ap = argparse.ArgumentParser()
ap.add_argument("--map-cache", default="build/_mc"); ap.add_argument("--scans", type=int, default=960); ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--parts", type=int, default=2)
args = ap.parse_args()
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
import synth_gpu
z = np.load(args.map_cache + ".rank0.npz", allow_pickle=True)
mc, ms = np.ascontiguousarray(z["corner"], np.float32), np.ascontiguousarray(z["surf"], np.float32)
world_model = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
lidar = synth_gpu.GpuLidar(world_model, 0)
rng = np.random.default_rng(4242)
dense = synth_gpu.loop_trajectory(100000)
seg = np.linalg.norm(np.diff(dense[:, 3:5], axis=0), axis=1).mean()
span = int(25.0 / seg)
scans, inits = [], []
for k in range(args.scans):
    g = dense[int(rng.integers(-span, span)) % len(dense)].copy()
    g[3:5] += rng.uniform(-1.0, 1.0, 2)
    g[2] += rng.uniform(-0.2, 0.2)
    scans.append(lidar.scan(g, 64, 1800, seed=900000 + k))
    inits.append(synth.perturb_pose(g, seed=99 + k))
inits = np.stack(inits)


def make(part_scans, part_inits):
    ctx = pkg.Context(0)
    ctx.map_set(mc, ms)
    ctx.scan_set_batch(part_scans)
    o = ctx.default_opts(); o.scans_in_flight = len(part_scans)
    return ctx, o, part_inits


def run(parts, steps):
    res = [None] * len(parts)

    def work(i):
        ctx, o, ini = parts[i]
        n = 0
        for _ in range(steps):
            _, poses, sts = ctx.run_batch(ini, o)
            n += sum(s.point_residuals for s in sts)
        res[i] = (n, poses)
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    return sum(r[0] for r in res) / dt, np.concatenate([r[1] for r in res]), dt / steps


one = [make(scans, inits)]
run(one, 1)
v1, p1, ms1 = run(one, args.steps)
print("one context, %d scans per call          : %.4e point-residuals/s, %.2f ms per step" % (args.scans, v1, 1e3 * ms1), flush=True)
one[0][0].close()
h = (args.scans + args.parts - 1) // args.parts
parts = [make(scans[i:i + h], inits[i:i + h]) for i in range(0, args.scans, h)]
run(parts, 1)
v2, p2, ms2 = run(parts, args.steps)
print("%d contexts on %d threads, %d scans per call: %.4e point-residuals/s, %.2f ms per step (%+.1f %%), same poses: %s" %
      (len(parts), len(parts), h, v2, 1e3 * ms2, 100 * (v2 / v1 - 1), bool(np.array_equal(p1.view(np.uint32), p2.view(np.uint32)))), flush=True)
