#!/usr/bin/env python3
"""Feature-extraction timing (row n2): lslam_extract_features on a 64x1800 sweep vs the CPU oracle."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
w = synth.World(half_extent=175.0)
ctx = pkg.Context(0)
for rings in (16, 64):
    c, s, gt, cloud, ranges = synth.make_scan(w, rings, 1800, full=True)
    for _ in range(3):
        out = pkg.scan_registration.extract_features(ctx, cloud, ranges)
    t0 = time.perf_counter()
    for _ in range(10):
        out = pkg.scan_registration.extract_features(ctx, cloud, ranges)
    dt = (time.perf_counter() - t0) / 10
    line = "%d rings, %d points: GPU %.2f ms per sweep (host buffers in and out)" % (rings, len(cloud), 1e3 * dt)
    if "--cpu" in sys.argv:
        from oracle_lib import Oracle
        o = Oracle(native=True)
        t0 = time.perf_counter(); ref = o.extract_features(cloud, ranges); cdt = time.perf_counter() - t0
        line += "; CPU oracle %.1f ms" % (1e3 * cdt)
    print(line, {k: len(v) for k, v in out.items()})
if os.environ.get("FX_CLOCKS"):  # -DLSLAM_FX_CLOCKS build
    import ctypes as C
    clk = (C.c_double * 16)()
    ctx.lib.lslam_debug_fx_clocks.argtypes = [C.c_double * 16]
    ctx.lib.lslam_debug_fx_clocks(clk)
    tot = sum(clk[:6])
    vt = sum(clk[8:13])
    print("fx_ring_voxel_kernel, share of thread 0's time: load+box %.2f, keys %.2f, sort %.2f, heads+scan %.2f, centroids %.2f" % tuple(c / vt for c in clk[8:13]))
    print("points classified per ring: %.0f" % (clk[6] / max(1.0, clk[7])))
    print("fx_ring_kernel, share of thread 0's time: marks %.2f, curvature %.2f, classify %.2f, flat picks %.2f, compactions %.2f, rank sort %.2f" % tuple(c / tot for c in clk[:6]))
