#!/usr/bin/env python3
"""ScanMatch::scanMatchScan as the reference calls it (ScanMatch.cpp:54-66: BOTH reference clouds and the scan handed over as
host clouds on every call; the reference rebuilds its kd-trees inside -- quirk Q4): wall time per call through the product's
C ABI, PCIe and map set included, with the kd-trees built per call and with them deferred (lslam_map_defer_trees).

    python tools/call_latency.py --map-cache build/_mc [--calls 20]

The map is the bench's surround (157 k corner + 587 k surf points), the scan one VLP-16 / HDL-64 sweep's features after the
mapping node's VoxelGrid.  Diagnostics for DESIGN 5 ("PCIe-inclusive"); not a bench line."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-cache", default="build/_mc")
    ap.add_argument("--calls", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    pkg = importlib.import_module("the-cooper-mapper_amd")
    synth = importlib.import_module("the-cooper-mapper_amd.synth")
    import synth_gpu
    z = np.load(args.map_cache + ".rank0.npz", allow_pickle=True)
    mc, ms = np.ascontiguousarray(z["corner"], np.float32), np.ascontiguousarray(z["surf"], np.float32)
    world_model = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
    lidar = synth_gpu.GpuLidar(world_model, 0)
    gt = np.asarray(synth_gpu.loop_trajectory(10000)[-1], np.float64)
    init = synth.perturb_pose(gt, seed=77, dt=0.1, dr_deg=0.5)
    for rings in (64, 16):
        qc, qs = lidar.scan(gt, rings, 1800, seed=4321)
        for what in ("full scan", "after VoxelGrid 1.0"):
            for defer, epoch in ((False, 0), (True, 0), (True, 1)):
                ctx = pkg.Context(0)
                ctx.defer_trees(defer)
                sm = pkg.ScanMatch(10, ctx=ctx)
                sm.setReferenceEpoch(epoch)  # 1: the caller promises the reference clouds are unchanged -> uploaded once
                c, s = (qc, qs) if what == "full scan" else (pkg.voxel_grid(ctx, qc, 1.0), pkg.voxel_grid(ctx, qs, 1.0))
                ts = []
                for k in range(args.calls + 2):
                    pose = init.copy()
                    t0 = time.perf_counter()
                    ok, pose = sm.scanMatchScan(mc, ms, c, s, pose)
                    ts.append(time.perf_counter() - t0)
                ts = np.array(ts[2:]) * 1e3
                print("%2d rings, %-20s %6d scan points, trees %-42s: median %.3f ms  min %.3f ms  (ok %s, err %.4f m, lazy %s)"
                      % (rings, what, len(c) + len(s), ("deferred" if defer else "built") + (" + reference epoch (map resident)" if epoch else ""), np.median(ts), ts.min(), ok,
                         float(np.abs(np.asarray(pose)[3:] - gt[3:]).max()), ctx.lazy_trees()))
                ctx.close()


if __name__ == "__main__":
    main()
