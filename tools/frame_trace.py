#!/usr/bin/env python3
"""One LaserMapping frame, step by step, many times: where the milliseconds of bench.py's `mapping_frame` leg go.

    python tools/frame_trace.py --map-cache build/_mc [--frames 30] [--rings 64]
        per-step median / min wall time over the frames (host clock around each call of the product API)

    rocprofv3 --kernel-trace -d gpurun_out/ft -o ft --output-format csv -- python3 tools/frame_trace.py --map-cache build/_mc --frames 4 --mark
    python tools/frame_trace.py --timeline gpurun_out/ft/.../ft_kernel_trace.csv
        the kernels of the LAST frame in launch order: start offset, duration, gap to the previous kernel's end [us]

The same steps, inputs and settings as bench.py's mapping_frame_leg (the surround comes from the cache bench.py --map-cache
writes).  Diagnostics only -- not a bench line."""
import argparse
import csv
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def timeline(path, last_marker="grid_bbox_kernel"):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a frame starts with fx_ring_kernel (extract_features); take the last complete one
    starts = [i for i, r in enumerate(rows) if "fx_ring_kernel" in r[2]]
    if len(starts) < 2:
        raise SystemExit("fewer than two frames in the trace")
    a, b = starts[-2], starts[-1]
    t0 = rows[a][0]
    prev_end = t0
    busy = 0
    print("%9s %8s %8s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
    for s, e, name in rows[a:b]:
        short = name.replace("(anonymous namespace)::", "").replace("lslam::", "").replace("void ", "").split("(")[0]
        if "rocprim" in short:
            short = "rocprim:" + ("onesweep" if "onesweep" in name else "block_sort" if "block_sort" in name else "merge" if "merge" in name
                                  else "scan" if "scan" in name else "histogram" if "histogram" in name else "other")
        print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short[:90]))
        busy += e - s
        prev_end = max(prev_end, e)
    print("frame: %.1f us from first kernel to next frame's first kernel, %.1f us of kernels, %d launches"
          % ((rows[b][0] - t0) / 1e3, busy / 1e3, b - a))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-cache", default="build/_mc")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--timeline", default=None)
    ap.add_argument("--mark", action="store_true", help="(profiling run) nothing extra is printed per frame")
    args = ap.parse_args()
    if args.timeline:
        return timeline(args.timeline)
    import numpy as np
    pkg = importlib.import_module("the-cooper-mapper_amd")
    synth = importlib.import_module("the-cooper-mapper_amd.synth")
    import synth_gpu
    z = np.load(args.map_cache + ".rank0.npz", allow_pickle=True)
    world_model = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
    lidar = synth_gpu.GpuLidar(world_model, 0)
    traj = synth_gpu.loop_trajectory(10000)
    gt = np.asarray(traj[-1], np.float64)
    ctx = pkg.Context(0)
    _, _, cloud, ranges = lidar.scan(gt, args.rings, 1800, seed=4321, full=True)
    fm = pkg.FeatureMap(ctx, 21, 21, 11)
    fm.setup_filter_size(0.2, 0.4, 0.6)
    fm.update(gt[3:].astype(np.float32))
    fm.add_feature_cloud(z["corner"], z["surf"], np.eye(4, dtype=np.float32))
    ctx.defer_trees(True)
    R, t = synth.pose_to_Rt(gt)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3], T[:3, 3] = R, t
    init = synth.perturb_pose(gt, seed=77, dt=0.1, dr_deg=0.5)
    opts = ctx.default_opts()
    steps = ("extract_features", "voxel_grid", "update", "surround_to_map", "scan_match", "add_feature_cloud")
    acc = {k: [] for k in steps}
    import gc
    gc.collect()
    gc.freeze()
    for f in range(args.frames + 1):
        t0 = time.perf_counter()
        feat = pkg.scan_registration.extract_features(ctx, cloud, ranges)
        t1 = time.perf_counter()
        dc, ds = pkg.voxel_grid2(ctx, feat["less_sharp"], feat["less_flat"], 1.0)
        t2 = time.perf_counter()
        fm.update(gt[3:].astype(np.float32))
        t3 = time.perf_counter()
        fm.surround_to_map()
        t4 = time.perf_counter()
        status, pose, st = ctx.scanmatch_scan(dc, ds, init, opts)
        t5 = time.perf_counter()
        fm.add_feature_cloud(dc, ds, T)
        t6 = time.perf_counter()
        if f > 0:
            for k, d in zip(steps, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                acc[k].append(d)
    med = {k: 1e3 * float(np.median(v)) for k, v in acc.items()}
    mn = {k: 1e3 * float(np.min(v)) for k, v in acc.items()}
    for k in steps:
        print("%-20s median %.3f ms   min %.3f ms" % (k, med[k], mn[k]))
    tot = np.array([sum(acc[k][i] for k in steps) for i in range(args.frames)]) * 1e3
    print("frames %d: median %.3f  p90 %.3f  p99 %.3f  worst %.3f ms" % (args.frames, np.median(tot), np.percentile(tot, 90),
                                                                        np.percentile(tot, 99), tot.max()))
    for i in np.argsort(-tot)[:8]:
        print("  frame %4d  %.3f ms :" % (i, tot[i]), "  ".join("%s %.3f" % (k[:7], 1e3 * acc[k][i]) for k in steps))
    print("%-20s median %.3f ms   min %.3f ms   (iterations %d, pose err %.4f m, lazy trees %s)"
          % ("frame", sum(med.values()), sum(mn.values()), st.iterations,
             float(np.abs(pose[3:] - gt[3:].astype(np.float32)).max()), ctx.lazy_trees()))


if __name__ == "__main__":
    main()
