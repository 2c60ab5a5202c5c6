#!/usr/bin/env python3
"""How many points of a Gauss-Newton loop the certificate path (sweep_body) still has to SEARCH: per loop of the 64 x 1800 test
problem and of a 12-scan batch.  LSLAM_DEBUG_CERT_STATS=1 python tools/cert_stats.py"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LSLAM_DEBUG_CERT_STATS"] = "1"
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
pr = synth.make_problem(rings=64, azimuth_steps=1800)
ctx = pkg.Context(0)
ctx.map_set(pr["map_corner"], pr["map_surf"])
ctx.scan_set(pr["corner"], pr["surf"])
out = (C.c_uint64 * 3)()
prev = [0, 0]
for mi in (1, 2, 3, 4, 5, 10):
    o = ctx.default_opts(); o.max_iterations = mi
    status, pose, st = ctx.run(pr["init_pose"], o)
    ctx.lib.lslam_debug_cert_stats(ctx.h, out)
    s, n = out[0] - prev[0], out[1] - prev[1]
    prev = [out[0], out[1]]
    print("max_iterations %2d: %d sweeps, searched %d of %d points swept (%.1f %%)" % (mi, st.sweeps, s, n, 100.0 * s / max(1, n)))
