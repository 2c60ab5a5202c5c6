#!/usr/bin/env python3
"""Quick A/B: average sweep / loop time of the 64-ring workload for the library in LSLAM_LIB."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
pr = synth.make_problem(rings=64, azimuth_steps=1800)
ctx = pkg.Context(0)
ctx.map_set(pr["map_corner"], pr["map_surf"])
ctx.scan_set(pr["corner"], pr["surf"])
opts = ctx.default_opts(); opts.profile = 1
opts.jtj_mode = int(os.environ.get("LSLAM_JTJ_MODE", "0"))
for _ in range(3):
    ctx.run(pr["init_pose"], opts)
sw = []; lp = []
for _ in range(int(os.environ.get("REPS", "20"))):
    status, pose, st = ctx.run(pr["init_pose"], opts)
    sw.append(st.gpu_ms_sweep / max(1, st.sweep_launches)); lp.append(st.gpu_ms_total)
import ctypes as C
clk = (C.c_uint64 * 8)()
ctx.lib.lslam_debug_solve_clocks.argtypes = [C.c_void_p, C.c_uint64 * 8]
ctx.lib.lslam_debug_solve_clocks(ctx.h, clk)
c = [int(x) for x in clk]
print("last solve kernel (us): reduce %.2f  qr %.2f  rest %.2f  total %.2f" % ((c[1]-c[0])/100, (c[2]-c[1])/100, (c[3]-c[2])/100, (c[3]-c[0])/100))
print("%-40s sweep_us median %.1f min %.1f | loop_us median %.1f | iters %d pose %s" % (
    os.path.basename(os.environ.get("LSLAM_LIB", "default")), 1e3 * np.median(sw), 1e3 * min(sw), 1e3 * np.median(lp),
    st.iterations, np.array2string(pose, precision=5)))
# the same loop without per-launch events (a single resident scan then runs as ONE persistent launch unless
# LSLAM_PERSISTENT_GN=0): device time of the whole loop and wall time of the call
import time
opts.profile = 0
for _ in range(3):
    ctx.run(pr["init_pose"], opts)
lp = []; wall = []
for _ in range(int(os.environ.get("REPS", "20"))):
    t0 = time.perf_counter()
    status, pose, st = ctx.run(pr["init_pose"], opts)
    wall.append(time.perf_counter() - t0); lp.append(st.gpu_ms_total)
print("no per-launch events: loop_us median %.1f | wall_us median %.1f | iters %d pose %s" % (
    1e3 * np.median(lp), 1e6 * np.median(wall), st.iterations, np.array2string(pose, precision=5)))
