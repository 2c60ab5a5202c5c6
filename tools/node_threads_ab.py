#!/usr/bin/env python3
"""A/B of the three-thread chain (tools/cpp/node_threads.cpp) built two ways: LaserMapping::process ending with
lslam_fmap_add_feature_cloud_begin (default) or with the waiting call (-DLSLAM_MAPPING_SYNC_ADD), interleaved runs on one box.
    python tools/node_threads_ab.py [rings=16] [runs=5]"""
import importlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
rings = int(sys.argv[1]) if len(sys.argv) > 1 else 16
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lo, hi = (-15.0, 15.0) if rings == 16 else (-24.9, 2.0)
world = synth.World(half_extent=175.0)
tmp = tempfile.mkdtemp(prefix="lslam_ab_")
path = os.path.join(tmp, "sweeps.bin")
with open(path, "wb") as f:
    f.write(np.uint32(rings).tobytes() + np.float32(lo).tobytes() + np.float32(hi).tobytes() + np.uint32(28).tobytes())
    for k in range(28):
        gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
        _, _, _, cloud, _ = synth.make_scan(world, rings, 1800, gt_pose=gt, seed=300 + k, full=True)
        ring = np.floor(cloud[:, 3]).astype(np.int64)
        a = np.ascontiguousarray(cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))][:, :4], np.float32)
        f.write(np.uint32(len(a)).tobytes()); f.write(a.tobytes())
libdir = os.path.dirname(pkg.lib_path())
exes = {}
for name, extra in (("begin", []), ("sync", ["-DLSLAM_MAPPING_SYNC_ADD"])):
    exes[name] = os.path.join(tmp, "nt_" + name)
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-Wall"] + extra + ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "node_threads.cpp"),
                           "-o", exes[name], "-L", libdir, "-llslam_hip", "-Wl,-rpath," + libdir, "-lpthread"], timeout=300)
variants = [("sync", {}), ("begin", {})]  # (add environments to try: ("sync", {"GPU_MAX_HW_QUEUES": "8"}), ("sync", {"HSA_ENABLE_SDMA": "0"}) ...)
for r in range(runs):
    for name, env in variants:
        out = subprocess.run([exes[name], path, "8"], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        line = [l for l in out.stdout.splitlines() if l.startswith("NODE_THREADS")]
        seq = subprocess.run([exes[name], path, "8", "seq"], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        sline = [l for l in seq.stdout.splitlines() if l.startswith("SEQUENTIAL")]
        print("run %d %-5s%-11s %s | %s" % (r, name, " ".join("%s=%s" % kv for kv in env.items())[-11:], line[0] if line else out.stderr[-200:], " ".join(sline[0].split()[:3]) if sline else ""), flush=True)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
