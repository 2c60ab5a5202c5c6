#!/usr/bin/env python3
"""Soak of the certificate sweep on the bench workload shape: the same batch of full scans against the voxel map REPS times in
one process; every run must give the same bits (the second pass's work list is filled in the order of atomics -- nothing else
may depend on it).  GPU box:  python tools/soak_cert.py [scans] [reps]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("the-cooper-mapper_amd"); synth = importlib.import_module("the-cooper-mapper_amd.synth")
import synth_gpu
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 96
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
world = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
lidar = synth_gpu.GpuLidar(world, 0)
traj = synth_gpu.loop_trajectory(10000)[-4000::10]
ctx = pkg.Context(0)
fm, stats = synth_gpu.build_voxel_map(pkg, ctx, lidar, traj, rings=16)
fm.update(traj[-120][3:].astype(np.float32))
fm.surround_to_map()
rng = np.random.default_rng(5)
scans, inits = [], []
for k in range(n_scans):
    g = traj[-120 + int(rng.integers(-12, 12))].copy()
    g[3:5] += rng.uniform(-1.0, 1.0, 2)
    scans.append(lidar.scan(g, 64, 1800, seed=9000 + k))
    inits.append(synth.perturb_pose(g, seed=260 + k))
inits = np.stack(inits)
ctx.scan_set_batch(scans)
opts = ctx.default_opts()
opts.scans_in_flight = n_scans
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
ref = None
q0 = ctx.cert_stats()[2]
t0 = time.time()
for r in range(reps):
    _, poses, sts = ctx.run_batch(inits, opts)
    key = (bits(poses).tobytes(), tuple((s.iterations, s.n_rows, s.n_line, s.n_plane, s.status) for s in sts))
    if ref is None:
        ref = key
    elif key != ref:
        print("run %d differs from run 0" % r)
        sys.exit(1)
print("soak: %d runs of %d scans (%d second passes) in %.1f s, every run the same bits" % (reps, n_scans, ctx.cert_stats()[2] - q0, time.time() - t0))
