#!/usr/bin/env python3
"""Which scan points are expensive?  Per-lane traversal counts vs. geometry (stats build, LSLAM_NO_MORTON=1)."""
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
B = int(os.environ.get("BLOCK", "256"))
pose = np.array(pr["init_pose"], np.float32)
nbc = (len(pr["corner"]) + B - 1) // B; nbs = (len(pr["surf"]) + B - 1) // B
nb = nbc + nbs; nw = nb * (B // 64)
cap = nw + nb * B * 2 + 16
buf = np.zeros(cap * 4, np.uint64)
n = lib.lslam_debug_sweep_clocks(ctx.h, pose.ctypes.data_as(C.POINTER(C.c_float)), 0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
assert n == nw, (n, nw)
raw = buf[nw * 4: nw * 4 + nb * B * 8].reshape(nb * B, 8)
n_node = raw[:, 3].astype(np.int64); n_leaf = raw[:, 4].astype(np.int64); n_pop = raw[:, 5].astype(np.int64)
surf = pr["surf"]; ns = len(surf)
sn = n_node[nbc * B: nbc * B + ns]; sl = n_leaf[nbc * B: nbc * B + ns]
R, t = synth.pose_to_Rt(pr["init_pose"].astype(np.float64))
pm = surf[:, :3].astype(np.float64) @ R.T + t
rng = np.linalg.norm(surf[:, :3], axis=1)
g = ctx.sweep(pr["init_pose"])
d2 = g["d2"][len(pr["corner"]):]
print("surf: n_node mean %.1f  n_leaf mean %.2f" % (sn.mean(), sl.mean()))
for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 80), (80, 200)):
    m = (rng >= lo) & (rng < hi)
    if m.sum(): print("range %3d-%3d m: n=%6d n_node %.1f n_leaf %.2f  d2[4] median %.3f  height med %.2f" % (lo, hi, m.sum(), sn[m].mean(), sl[m].mean(), np.median(d2[m, 4]), np.median(pm[m, 2])))
gnd = np.abs(pm[:, 2]) < 0.15
print("ground pts: n=%d n_node %.1f n_leaf %.2f | non-ground: n=%d n_node %.1f n_leaf %.2f" % (gnd.sum(), sn[gnd].mean(), sl[gnd].mean(), (~gnd).sum(), sn[~gnd].mean(), sl[~gnd].mean()))
for q in (50, 90, 99, 99.9):
    print("n_leaf p%.1f = %d, n_node p%.1f = %d" % (q, np.percentile(sl, q), q, np.percentile(sn, q)))
hard = sl >= np.percentile(sl, 99)
print("hardest 1%%: range med %.1f height med %.2f d2[4] med %.3f d2[0] med %.4f" % (np.median(rng[hard]), np.median(pm[hard, 2]), np.median(d2[hard, 4]), np.median(d2[hard, 0])))
print("fraction with d2[4] >= 5 (rejected):", (d2[:, 4] >= 5).mean(), " their n_leaf mean %.1f" % (sl[d2[:, 4] >= 5].mean() if (d2[:, 4] >= 5).any() else 0))
