#!/usr/bin/env python3
"""One resident 64 x 1800 scan against the bench map: wall time and device time per scanMatchScan loop through each search
(AUTO, LANE, GRID; trees built / deferred) -- which one AUTO should take for a single full scan.  GPU box.
    python tools/single_scan_modes.py --map-cache /tmp/mc"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
ap = argparse.ArgumentParser(); ap.add_argument("--map-cache", default="build/_mc"); ap.add_argument("--calls", type=int, default=200)
args = ap.parse_args()
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
import synth_gpu
z = np.load(args.map_cache + ".rank0.npz", allow_pickle=True)
mc, ms = np.ascontiguousarray(z["corner"], np.float32), np.ascontiguousarray(z["surf"], np.float32)
world_model = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
lidar = synth_gpu.GpuLidar(world_model, 0)
gt = np.asarray(synth_gpu.loop_trajectory(10000)[-1], np.float64)
for rings in (64, 16):
    qc, qs = lidar.scan(gt, rings, 1800, seed=4321)
    init = synth.perturb_pose(gt, seed=99)
    for defer in (False, True):
        ctx = pkg.Context(0)
        ctx.defer_trees(defer)
        ctx.map_set(mc, ms)
        ctx.scan_set(qc, qs)
        for name, mode in (("auto", 0), ("lane", 1), ("grid", 3)):
            o = ctx.default_opts(); o.search_mode = mode
            for _ in range(10): ctx.run(init, o)
            ts = []; g = []
            for _ in range(args.calls):
                t = time.perf_counter(); status, pose, st = ctx.run(init, o); ts.append(time.perf_counter() - t); g.append(st.gpu_ms_total)
            print("%2d rings %6d pts trees %-8s %-5s: wall median %.3f ms  min %.3f  device %.3f ms  iterations %d  err %.4f m  lazy %s"
                  % (rings, len(qc) + len(qs), "deferred" if defer else "built", name, 1e3 * np.median(ts), 1e3 * min(ts), np.median(g), st.iterations,
                     float(np.abs(pose[3:] - gt[3:].astype(np.float32)).max()), ctx.lazy_trees()), flush=True)
        ctx.close()

# A mapping frame's scan (the 16-ring sweep's feature clouds through VoxelGrid 0.2 / 0.4: a few thousand points, about twenty
# workgroups -- the size at which a launch per step is all latency): the kd-tree walk with a launch per sweep and per solve,
# the same loop as ONE persistent launch (LSLAM_AB_PERSISTENT_GN, round 2: measured until now only on a 450-workgroup scan),
# the solve in the sweep's tail (LSLAM_AB_FUSED_SOLVE), and the frame's own path (trees deferred: probe, plan, wide probe,
# queue, solve -- five launches per iteration)
fm_mod = importlib.import_module("the-cooper-mapper_amd.feature_map")
qc, qs = lidar.scan(gt, 16, 1800, seed=4321)
init = synth.perturb_pose(gt, seed=99)
ctx = pkg.Context(0)
fc, fs = fm_mod.voxel_grid(ctx, qc, 0.2), fm_mod.voxel_grid(ctx, qs, 0.4)
ctx.close()
for defer, name, mode, ab in ((False, "lane, a launch per step", 1, 0), (False, "lane, persistent loop", 1, 1), (False, "lane, fused solve", 1, 2),
                              (True, "grid (the frame's path)", 0, 0), (True, "grid, fused solve", 0, 2)):
    ctx = pkg.Context(0)
    ctx.defer_trees(defer)
    ctx.map_set(mc, ms)
    ctx.scan_set(fc, fs)
    o = ctx.default_opts(); o.search_mode = mode; o.ab_switches = ab
    for _ in range(10): ctx.run(init, o)
    ts = []; g = []
    for _ in range(args.calls):
        t = time.perf_counter(); status, pose, st = ctx.run(init, o); ts.append(time.perf_counter() - t); g.append(st.gpu_ms_total)
    print("frame scan %5d pts (%d workgroups) trees %-8s %-26s: wall median %.3f ms  min %.3f  device %.3f ms  iterations %d  err %.4f m"
          % (len(fc) + len(fs), (len(fc) + 255) // 256 + (len(fs) + 255) // 256, "deferred" if defer else "built", name, 1e3 * np.median(ts), 1e3 * min(ts),
             np.median(g), st.iterations, float(np.abs(pose[3:] - gt[3:].astype(np.float32)).max())), flush=True)
    ctx.close()
