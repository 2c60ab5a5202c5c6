#!/usr/bin/env python3
"""Where the time of ONE small scan match goes (the mapping node's per-frame match: a few thousand voxel-filtered feature
points against the surround map): lslam_scan_set (pack, H2D, Morton ordering) against lslam_scanmatch_run, wall clock, median
of many calls; and the same for a full 64 x 1800 scan.  GPU box."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
pr = synth.make_problem(rings=64, azimuth_steps=1800)
ctx = pkg.Context(0)
ctx.map_set(pr["map_corner"], pr["map_surf"])
opts = ctx.default_opts()
opts.delta_t_abort = opts.delta_r_abort = 0.1
opts.use_score = 0
def med(f, n=200):
    for _ in range(10): f()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
for name, c, s in (("small (voxel-filtered frame)", pkg.voxel_grid(ctx, pr["corner"], 1.0), pkg.voxel_grid(ctx, pr["surf"], 1.0)),
                   ("full 64 x 1800", pr["corner"], pr["surf"])):
    c = np.ascontiguousarray(c, np.float32); s = np.ascontiguousarray(s, np.float32)
    t_set = med(lambda: ctx.scan_set(c, s))
    st = [None]
    def run():
        st[0] = ctx.run(pr["init_pose"], opts)[2]
    t_run = med(run)
    t_both = med(lambda: ctx.scanmatch_scan(c, s, pr["init_pose"], opts))
    print("%-30s %6d points: scan_set %.3f ms, run %.3f ms (%d iterations, gpu %.3f ms), scanmatch_scan %.3f ms"
          % (name, len(c) + len(s), t_set, t_run, st[0].iterations, st[0].gpu_ms_total, t_both))
