#!/usr/bin/env python3
"""Per-wave phase breakdown of one sweep (shader-clock stamps from the debug tap)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
pr = synth.make_problem(rings=64, azimuth_steps=1800)
ctx = pkg.Context(0)
ctx.map_set(pr["map_corner"], pr["map_surf"])
ctx.scan_set(pr["corner"], pr["surf"])
lib = ctx.lib
lib.lslam_debug_sweep_clocks.restype = C.c_int
lib.lslam_debug_sweep_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_uint64), C.c_size_t]
pose = np.array(pr["init_pose"], np.float32)
buf = np.zeros((4000, 4), np.uint64)
for rep in range(3):
    n = lib.lslam_debug_sweep_clocks(ctx.h, pose.ctypes.data_as(C.POINTER(C.c_float)), 0,
                                     buf.ctypes.data_as(C.POINTER(C.c_uint64)), 4000)
b = buf[:n].astype(np.int64)
t0 = b[:, 0].min()
start = b[:, 0] - t0; end = b[:, 3] - t0
knn = b[:, 1] - b[:, 0]; fit = b[:, 2] - b[:, 1]; red = b[:, 3] - b[:, 2]
print("waves", n, "kernel span cycles", end.max(), "(clock is the 100 MHz-or-shader counter; ratios matter)")
for name, v in (("start", start), ("end", end), ("knn", knn), ("fit", fit), ("reduce", red), ("total", b[:, 3] - b[:, 0])):
    print("%-7s mean %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f" % (name, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.max()))
# which logical blocks are slow? (wave index = logical block*2 + wave; corner blocks first)
tot = b[:, 3] - b[:, 0]
nbc = (len(pr["corner"]) + 127) // 128
print("corner waves: mean total %.0f  knn %.0f fit %.0f" % (tot[:2 * nbc].mean(), knn[:2 * nbc].mean(), fit[:2 * nbc].mean()))
print("surf   waves: mean total %.0f  knn %.0f fit %.0f" % (tot[2 * nbc:].mean(), knn[2 * nbc:].mean(), fit[2 * nbc:].mean()))
order = np.argsort(-tot)[:15]
print("slowest waves (index, total, knn, fit):", [(int(i), int(tot[i]), int(knn[i]), int(fit[i])) for i in order])
# by ring (surf points are ring-major): bucket surf waves into 16 groups
sw = np.arange(2 * nbc, n)
for g in np.array_split(sw, 16):
    print("surf waves %5d-%5d: total %8.0f knn %8.0f fit %7.0f end %8.0f" % (g[0], g[-1], tot[g].mean(), knn[g].mean(), fit[g].mean(), end[g].max()))
