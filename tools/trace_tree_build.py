#!/usr/bin/env python3
"""Timeline of one lslam_fmap_surround_to_map (both kd-trees of the surround) from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tb -o t -- python3 tools/trace_tree_build.py
    python3 tools/trace_tree_build.py --report gpurun_out/tb
"""
import csv, glob, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last build = after the last marker kernel (fm_gather_kernel appears once per cloud at the start of a build)
    idx = [i for i, r in enumerate(rows) if "fm_gather_kernel" in r["Kernel_Name"]]
    start = idx[-2]
    t0 = int(rows[start]["Start_Timestamp"])
    busy = 0
    prev_end = t0
    gaps = 0
    names = {}
    for r in rows[start:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        import re
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"<.*", "", n.split("(")[0]).split("::")[-1].split(" ")[-1][:28]
        names.setdefault(n, [0, 0])
        names[n][0] += 1; names[n][1] += e - s
    end = max(int(r["End_Timestamp"]) for r in rows[start:])
    spans = {}
    for r in rows[start:]:
        n = r["Kernel_Name"]
        key = "kd_build_small" if "kd_build_small" in n else ("lv_" if "lv_" in n else "other")
        a, b = spans.get(key, (1 << 62, 0))
        spans[key] = (min(a, int(r["Start_Timestamp"])), max(b, int(r["End_Timestamp"])))
    for k, (a, b) in spans.items():
        print("  phase %-16s from %.1f to %.1f us" % (k, (a - t0) / 1e3, (b - t0) / 1e3))
    print("last build: %d launches, span %.1f us, sum of kernel durations %.1f us" % (len(rows) - start, (end - t0) / 1e3, busy / 1e3))
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print("  %-30s x%3d  %8.1f us  (%.1f us each)" % (n, c, t / 1e3, t / 1e3 / c))
    sys.exit(0)
import numpy as np
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
import synth_gpu
w = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
lidar = synth_gpu.GpuLidar(w, 0)
traj = synth_gpu.loop_trajectory(int(os.environ.get("MAP_FRAMES", "10000")))
ctx = pkg.Context(0)
fm, st = synth_gpu.build_voxel_map(pkg, ctx, lidar, traj, rings=16, progress=None)
fm.update(traj[-1][3:].astype(np.float32))
for k in range(4):
    t0 = time.perf_counter()
    fm.surround_to_map()
    print("surround_to_map %.3f ms" % (1e3 * (time.perf_counter() - t0)), ctx.map_info().n_corner, ctx.map_info().n_surf)
