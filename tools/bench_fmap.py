#!/usr/bin/env python3
"""Map-maintenance timing (row n1): per-frame update + surround->kd-trees + addFeatureCloud on the
64-ring workload's 1.3 M-point map, next to the CPU oracle doing the reference's per-frame work."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
pr = synth.make_problem(rings=64, azimuth_steps=1800)
ctx = pkg.Context(0)
def xyzi(a):
    out = np.zeros((len(a), 4), np.float32); out[:, :3] = a[:, :3]; return out
mc, ms = xyzi(pr["map_corner"]), xyzi(pr["map_surf"])
fm = pkg.FeatureMap(ctx, 21, 11, 21)
fm.setup_filter_size(0.2, 0.4, 0.6)
I = np.eye(4, dtype=np.float32)
gt = pr["gt_pose"]
fm.update(gt[3:])
t0 = time.perf_counter(); fm.add_feature_cloud(mc, ms, I); t_load = time.perf_counter() - t0
print("initial load of %d + %d points: %.1f ms -> %s" % (len(mc), len(ms), 1e3 * t_load, fm.info()["n_corner"] + fm.info()["n_surf"]))
R, t = synth.pose_to_Rt(gt)
T = np.eye(4, dtype=np.float32); T[:3, :3] = R; T[:3, 3] = t
qc, qs = pr["corner"], pr["surf"]
res = {"update": [], "surround_to_map": [], "add": []}
for k in range(6):
    t0 = time.perf_counter(); fm.update(gt[3:] + 0.5 * k); res["update"].append(time.perf_counter() - t0)
    t0 = time.perf_counter(); fm.surround_to_map(); res["surround_to_map"].append(time.perf_counter() - t0)
    t0 = time.perf_counter(); fm.add_feature_cloud(qc, qs, T); res["add"].append(time.perf_counter() - t0)
for k, v in res.items():
    print("%-16s %s ms" % (k, " ".join("%.2f" % (1e3 * x) for x in v)))
print("map now", fm.info()["n_corner"], fm.info()["n_surf"], "surround", fm.surround_counts())
if "--cpu" in sys.argv:
    from oracle_lib import Oracle
    o = Oracle(native=True)
    ofm = o.feature_map(21, 11, 21); ofm.setup_filter_size(0.2, 0.4, 0.6)
    ofm.update(gt[3:]); ofm.add_feature_cloud(mc, ms, I)
    t0 = time.perf_counter(); ofm.update(gt[3:]); c, s = ofm.get_surround_feature(); t1 = time.perf_counter()
    ofm.add_feature_cloud(qc, qs, T); t2 = time.perf_counter()
    print("CPU oracle: update+surround %.1f ms, addFeatureCloud %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
