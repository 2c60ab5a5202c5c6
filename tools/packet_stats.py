#!/usr/bin/env python3
"""Per-wavefront counters of the packet search (profiling build: make -C the-cooper-mapper_amd/csrc
EXTRA=-DLSLAM_PACKET_STATS OBJDIR=../../build/obj_stats OUT=../../build/liblslam_stats.so; run with
LSLAM_LIB=build/liblslam_stats.so).  Queries = one 64-ring scan, Morton-ordered, at its perturbed pose."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
pr = synth.make_problem(rings=64, azimuth_steps=1800, world_half=float(os.environ.get("WORLD_HALF", "100")))
ctx = pkg.Context(0)
ctx.map_set(pr["map_corner"], pr["map_surf"])
lib = ctx.lib
lib.lslam_debug_packet_stats.restype = C.c_int
lib.lslam_debug_packet_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
def morton(p):
    def sp(v):
        v = v & 0x3FF; v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    q = np.clip(p[:, :3] * 4.0 + 512.0, 0, 1023).astype(np.int64)
    return sp(q[:, 0]) | (sp(q[:, 1]) << 1) | (sp(q[:, 2]) << 2)
R, t = synth.pose_to_Rt(pr["init_pose"])
for which, sc in ((1, pr["surf"]), (0, pr["corner"])):
    order = np.argsort(morton(sc), kind="stable")
    Q = np.zeros((len(sc), 4), np.float32)
    Q[:, :3] = (sc[order][:, :3].astype(np.float64) @ R.T + t).astype(np.float32)
    nw = (len(Q) + 63) // 64
    out = np.zeros((nw, 8), np.uint32)
    for rep in range(2):
        rc = lib.lslam_debug_packet_stats(ctx.h, which, Q.ctypes.data, len(Q), out.ctypes.data)
        assert rc == 0
    m = out[:, :6].astype(np.float64)
    print("which", which, "waves", nw, "mean nodes %.1f leaves %.1f inserts %.1f pops %.1f ties %.2f cycles %.0f" % tuple(m.mean(0)),
          "| p99 nodes %.0f leaves %.0f cycles %.0f | max cycles %.0f" % (np.percentile(m[:, 0], 99), np.percentile(m[:, 1], 99), np.percentile(m[:, 5], 99), m[:, 5].max()))
