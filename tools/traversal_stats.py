#!/usr/bin/env python3
"""Per-lane traversal statistics (needs a -DLSLAM_TRAVERSAL_STATS build: LSLAM_LIB=build/liblslam_hip_stats.so)."""
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
nb = (len(pr["corner"]) + 255) // 256 + (len(pr["surf"]) + 255) // 256
nw = nb * 4
cap = nw + nb * 256 * 2 + 16
buf = np.zeros(cap * 4, np.uint64)
for rep in range(2):
    n = lib.lslam_debug_sweep_clocks(ctx.h, pose.ctypes.data_as(C.POINTER(C.c_float)), 0,
                                     buf.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
assert n == nw, n
raw = buf[nw * 4: nw * 4 + nb * 256 * 8].reshape(nb * 256, 8)
valid = (raw[:, 7] >> np.uint64(63)) == 1
st = np.zeros((nb * 256, 9), np.int64)
st[:, :6] = raw[:, :6].astype(np.int64)
st[:, 6] = (raw[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
st[:, 7] = (raw[:, 6] >> np.uint64(32)).astype(np.int64)
st[:, 8] = (raw[:, 7] & np.uint64((1 << 63) - 1)).astype(np.int64)
print("lanes", valid.sum())
names = ["t_desc", "t_leaf", "t_pop", "n_node", "n_leaf", "n_pop", "n_take", "n_popit", "t_take"]
for i, nme in enumerate(names):
    v = st[valid, i]
    print("%-7s per-lane mean %9.1f p50 %8.0f p90 %8.0f max %8.0f" % (nme, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.max()))
# per-wave: lanes in a wave run in lockstep, so the wave pays roughly the max over lanes
w = st.reshape(nb * 4, 64, 9)
wm = w.max(axis=1)
print("per-wave max over lanes:")
for i, nme in enumerate(names):
    v = wm[:, i]
    print("%-7s wave mean %9.1f p50 %8.0f p90 %8.0f max %8.0f" % (nme, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.max()))
tot = wm[:, 0] + wm[:, 1] + wm[:, 2] + wm[:, 8]
print("cycles per node step (wave): %.0f ; per leaf: %.0f ; per pop: %.0f" % (
    wm[:, 0].sum() / max(1, wm[:, 3].sum()), wm[:, 1].sum() / max(1, wm[:, 4].sum()), wm[:, 2].sum() / max(1, wm[:, 5].sum())))
for g in np.array_split(np.arange(nb * 4), 12):
    print("waves %5d-%5d  n_node %6.1f n_leaf %5.1f n_pop %6.1f | lane-mean n_node %6.1f n_leaf %5.1f | t %8.0f" % (
        g[0], g[-1], wm[g, 3].mean(), wm[g, 4].mean(), wm[g, 5].mean(),
        w[g][:, :, 3].mean(), w[g][:, :, 4].mean(), tot[g].mean()))
