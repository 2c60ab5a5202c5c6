#!/usr/bin/env python3
"""Whole per-sweep chain timing: raw sweep -> registration -> extraction -> odometry -> mapping."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
rings = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lo, hi = (-15.0, 15.0) if rings == 16 else (-24.9, 2.0)
world = synth.World(half_extent=175.0)
ctx = pkg.Context(0)
DEV = os.environ.get("DEV", "1") != "0"  # the odometry node resident on the device (lslam_odom_*); DEV=0: host-pointer calls
odo = pkg.DeviceLaserOdometry(ctx) if DEV else pkg.LaserOdometry(ctx)
fsets = [pkg.scan_registration.FeatureSet(ctx) for _ in range(2)]
dims = tuple(int(v) for v in os.environ.get("DIMS", "21,21,11").split(","))
mapper = pkg.LaserMapping(ctx, cube_dims=dims)
sr = pkg.scan_registration
raws = []
for k in range(int(os.environ.get("SWEEPS", "8"))):
    gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
    c, s, gtp, cloud, ranges = synth.make_scan(world, rings, 1800, gt_pose=gt, seed=300 + k, full=True)
    ring = np.floor(cloud[:, 3]).astype(np.int64)
    raws.append(cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))])
acc = {"register": 0.0, "extract": 0.0, "odometry": 0.0, "mapping": 0.0}
n = 0
# (as bench.py does before its timed regions: a generation-2 pass of the interpreter's garbage collector takes tens of milliseconds
# with torch and numpy loaded and lands in whichever call allocates the triggering object -- here it used to be sweep 5's odometry)
import gc
gc.collect(); gc.freeze(); gc.disable()
for k, raw in enumerate(raws):
    t0 = time.perf_counter(); reg, rr = sr.multiscan_register(ctx, raw, lo, hi, rings)
    if DEV:
        t1 = time.perf_counter(); sr.extract_features_dev(ctx, reg, rr, fsets[k & 1])
        t2 = time.perf_counter(); T = odo.process(fsets[k & 1])
    else:
        t1 = time.perf_counter(); f = sr.extract_features(ctx, reg, rr)
        t2 = time.perf_counter(); T = odo.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
    t3 = time.perf_counter()
    if os.environ.get("DETAIL") and k:
        st = odo.last_stats
        print("  sweep %d: odometry %.3f ms wall, loop %.3f ms on the device, %d iterations, %d rows" % (k, 1e3 * (t3 - t2), st.gpu_ms_total, st.iterations, st.n_rows))
    if T is not None:
        M = mapper.process(odo.last_corner, odo.last_surf, T)
    t4 = time.perf_counter()
    if k >= 2:
        for key, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[key] += d
        n += 1
print("%d rings, %d points/sweep: " % (rings, len(raws[0])) + ", ".join("%s %.2f ms" % (k, 1e3 * v / n) for k, v in acc.items()) +
      " -> %.2f ms per sweep (odometry iterations %d)" % (1e3 * sum(acc.values()) / n, odo.last_stats.iterations))
print("map pose", M[:3, 3])
if os.environ.get("DETAIL") and not DEV:
    # where the odometry stage's wall time goes: the match call (packing, kd-trees of the last clouds,
    # the device loop) and the two transformToEnd calls
    import types
    tm = {"odometry_match": 0.0, "transform_to_end": 0.0, "loop_gpu_ms": 0.0}
    om, te = ctx.odometry_match, ctx.transform_to_end

    def om2(*a, **k):
        t = time.perf_counter(); r = om(*a, **k); tm["odometry_match"] += time.perf_counter() - t
        tm["loop_gpu_ms"] += r[2].gpu_ms_total * 1e-3
        return r

    def te2(*a, **k):
        t = time.perf_counter(); r = te(*a, **k); tm["transform_to_end"] += time.perf_counter() - t
        return r
    ctx.odometry_match, ctx.transform_to_end = om2, te2
    reps = 0
    for k, raw in enumerate(raws[2:]):
        reg, rr = sr.multiscan_register(ctx, raw, lo, hi, rings)
        f = sr.extract_features(ctx, reg, rr)
        odo.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
        reps += 1
    print("odometry detail per sweep:", {k: round(1e3 * v / reps, 3) for k, v in tm.items()},
          "targets", len(odo.tree_corner), len(odo.tree_surf), "queries", len(f["sharp"]), len(f["flat"]))
if os.environ.get("DETAIL"):
    # the mapping stage, step by step (wall time per call, averaged over the sweeps after the first two)
    from collections import defaultdict
    fm_mod = importlib.import_module("the-cooper-mapper_amd.pipeline")
    tm2 = defaultdict(float)

    def wrap(obj, name, label):
        f = getattr(obj, name)

        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            tm2[label] += time.perf_counter() - t
            return r
        setattr(obj, name, g)
    wrap(fm_mod, "voxel_grid2", "voxel_grid2")
    wrap(mapper.feature_map, "update", "update")
    wrap(mapper.feature_map, "surround_counts", "surround_counts")
    wrap(mapper.feature_map, "surround_to_map", "surround_to_map")
    wrap(mapper.feature_map, "add_feature_cloud", "add_feature_cloud")
    wrap(ctx, "scanmatch_scan", "scanmatch_scan")
    reps = 0
    for k, raw in enumerate(raws[2:]):
        reg, rr = sr.multiscan_register(ctx, raw, lo, hi, rings)
        if DEV:
            sr.extract_features_dev(ctx, reg, rr, fsets[k & 1])
            T = odo.process(fsets[k & 1])
        else:
            f = sr.extract_features(ctx, reg, rr)
            T = odo.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
        t = time.perf_counter()
        mapper.process(odo.last_corner, odo.last_surf, T)
        tm2["mapping_total"] += time.perf_counter() - t
        reps += 1
    print("mapping detail per sweep:", {k: round(1e3 * v / reps, 3) for k, v in tm2.items()})
