#!/usr/bin/env python3
"""Soak: the per-sweep chain for many sweeps (the same few raw sweeps over and over, so the map saturates):
device memory in use, time per sweep and pose sanity at intervals."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
rings = int(os.environ.get("RINGS", "16"))
n_sweeps = int(os.environ.get("SWEEPS", "400"))
lo, hi = (-15.0, 15.0) if rings == 16 else (-24.9, 2.0)
world = synth.World(half_extent=175.0)
ctx = pkg.Context(0)
odo = pkg.LaserOdometry(ctx)
mapper = pkg.LaserMapping(ctx, cube_dims=(21, 21, 11))
sr = pkg.scan_registration
raws = []
for k in range(16):
    gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
    _, _, _, cloud, _ = synth.make_scan(world, rings, 1800, gt_pose=gt, seed=300 + k, full=True)
    ring = np.floor(cloud[:, 3]).astype(np.int64)
    raws.append(cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))])
# forward then backward through the 16 poses, again and again: bounded motion, map saturates
order = list(range(16)) + list(range(14, 0, -1))
t_blk = time.perf_counter()
for i in range(n_sweeps):
    raw = raws[order[i % len(order)]]
    reg, rr = sr.multiscan_register(ctx, raw, lo, hi, rings)
    f = sr.extract_features(ctx, reg, rr)
    T = odo.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
    if T is not None:
        M = mapper.process(odo.last_corner, odo.last_surf, T)
        assert np.isfinite(M).all()
    if (i + 1) % 50 == 0:
        free, total = torch.cuda.mem_get_info()
        info = mapper.feature_map.info()
        print("sweep %4d: %.2f ms/sweep, device memory in use %.1f MiB, map %d+%d points, map pose %s" % (
            i + 1, 1e3 * (time.perf_counter() - t_blk) / 50, (total - free) / 2**20, info["n_corner"], info["n_surf"],
            np.round(M[:3, 3], 3)), flush=True)
        t_blk = time.perf_counter()
