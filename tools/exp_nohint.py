#!/usr/bin/env python3
"""EXPERIMENT (a -DLSLAM_EXP_COUNT_NOHINT build): of the points the grid sweep lists for the tree search, how many go there without
a bound from the probe (it saw fewer than five candidates), per step of the bench workload.  LSLAM_LIB=build/exp/nohint.so"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
import synth_gpu
z = np.load(sys.argv[1] + ".rank0.npz", allow_pickle=True)
lidar = synth_gpu.GpuLidar(synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5), 0)
ctx = pkg.Context(0)
ctx.map_set(np.ascontiguousarray(z["corner"], np.float32), np.ascontiguousarray(z["surf"], np.float32))
rng = np.random.default_rng(4242)
dense = synth_gpu.loop_trajectory(100000)
seg = np.linalg.norm(np.diff(dense[:, 3:5], axis=0), axis=1).mean(); span = int(25.0 / seg)
scans, inits = [], []
for k in range(64):
    g = dense[int(rng.integers(-span, span)) % len(dense)].copy(); g[3:5] += rng.uniform(-1.0, 1.0, 2); g[2] += rng.uniform(-0.2, 0.2)
    scans.append(lidar.scan(g, 64, 1800, seed=900000 + k)); inits.append(synth.perturb_pose(g, seed=99 + k))
ctx.scan_set_batch(scans)
o = ctx.default_opts(); o.debug_stats = 1; o.scans_in_flight = 64
ctx.run_batch(np.stack(inits), o)
raw = (C.c_uint64 * 32)()
ctx.lib.lslam_debug_grid_stats(ctx.h, raw)
print("listed %d of %d swept; listed WITHOUT a bound from the probe (fewer than five seen): %d (%.1f %% of the listed); listed with a fifth seen beyond the gate: %d"
      % (raw[0], raw[1], raw[2], 100.0 * raw[2] / max(1, raw[0]), raw[3]))
