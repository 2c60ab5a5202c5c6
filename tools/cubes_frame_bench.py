#!/usr/bin/env python3
"""One mapping frame with the incrementally kept per-cube trees (variant C) next to the whole-surround rebuild
(variant A) on a 3 000-frame map; LSLAM_FMAP_TIMING=1 prints where lslam_fmap_to_cubemap spends its time."""
import importlib, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
pkg = importlib.import_module("the-cooper-mapper_amd"); synth = importlib.import_module("the-cooper-mapper_amd.synth")
import synth_gpu
w = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
lidar = synth_gpu.GpuLidar(w, 0)
traj = synth_gpu.loop_trajectory(3000)
ctx = pkg.Context(0)
fm, st = synth_gpu.build_voxel_map(pkg, ctx, lidar, traj, rings=16, progress=None)
fm.update(traj[-1][3:].astype(np.float32))
sur = fm.get_surround_feature()
opts = ctx.default_opts()
r = bench.mapping_frame_leg(pkg, synth, ctx, sur, lidar, traj[-1], opts, np, False, 64, cubes=True)
print(json.dumps(r["gpu_ms"]), r["cube_trees_built_reused_per_frame"][-1], r["map_points"])
r = bench.mapping_frame_leg(pkg, synth, ctx, sur, lidar, traj[-1], opts, np, False, 64, cubes=False)
print(json.dumps(r["gpu_ms"]))
