#!/usr/bin/env python3
"""Where the odometry node's correspondence search spends its time: per-query profile of one search launch
(lslam_debug_odom_search) over a few sweeps.  python3 tools/odom_search_stats.py [rings]"""
import ctypes as C, importlib, os, sys
os.environ["LSLAM_DEBUG_HOOKS"] = "1"
os.environ["LSLAM_ODOM_SEARCH_TAP"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
rings = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lo, hi = (-15.0, 15.0) if rings == 16 else (-24.9, 2.0)
world = synth.World(half_extent=175.0)
ctx = pkg.Context(0)
sr = pkg.scan_registration
odo = pkg.DeviceLaserOdometry(ctx)
fs = sr.FeatureSet(ctx)
for k in range(6):
    gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
    _, _, _, cloud, _ = synth.make_scan(world, rings, 1800, gt_pose=gt, seed=300 + k, full=True)
    ring = np.floor(cloud[:, 3]).astype(np.int64)
    raw = cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))]
    reg, rr = sr.multiscan_register(ctx, raw, lo, hi, rings)
    cnt = sr.extract_features_dev(ctx, reg, rr, fs)
    odo.process(fs)
    if k < 3:
        continue
    out = np.zeros((cnt["sharp"] + cnt["flat"], 4), np.uint32)
    n = ctx.lib.lslam_debug_odom_search(odo.h, out.ctypes.data_as(C.POINTER(C.c_uint32)), len(out))
    out = out[:n]
    for name, sl in (("sharp", slice(0, cnt["sharp"])), ("flat", slice(cnt["sharp"], n))):
        o = out[sl]
        us = o[:, 0] * 0.01
        print("sweep %d %-5s n=%4d  us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | nn cand mean %.0f max %d, coarse %.1f%% | cat cand mean %.0f max %d, coarse %.1f%%" % (
            k, name, len(o), us.mean(), np.percentile(us, 50), np.percentile(us, 90), np.percentile(us, 99), us.max(), o[:, 1].mean(), o[:, 1].max(),
            100.0 * (o[:, 3] & 1).mean(), o[:, 2].mean(), o[:, 2].max(), 100.0 * ((o[:, 3] >> 1) & 1).mean()))
    slow = out[np.argsort(-out[:, 0].astype(np.int64))[:5]]
    print("   slowest:", [(round(r[0] * 0.01, 1), int(r[1]), int(r[2]), int(r[3])) for r in slow])
