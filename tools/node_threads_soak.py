#!/usr/bin/env python3
"""Soak of the three-nodelet chain (tools/cpp/node_threads.cpp: registration, odometry and mapping on three std::threads, a
context each, the odometry node's persistent five-iteration launches spinning on their exchange slots): N consecutive sweeps of
a sensor driving a circle, one process, no interpreter in the loop.  Checks that it ends, that the mapped pose has followed the
drive, and prints the period.  GPU box:  python tools/node_threads_soak.py [sweeps=600] [rings=16]"""
import importlib, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rings = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lo, hi = (-15.0, 15.0) if rings == 16 else (-24.9, 2.0)
world = synth.World(half_extent=175.0)
R, step = 60.0, 0.4  # a circle of 60 m radius, 0.4 m per sweep
tmp = tempfile.mkdtemp(prefix="lslam_soak_")
path = os.path.join(tmp, "sweeps.bin")
t0 = time.time()
with open(path, "wb") as f:
    f.write(np.uint32(rings).tobytes() + np.float32(lo).tobytes() + np.float32(hi).tobytes() + np.uint32(sweeps).tobytes())
    for k in range(sweeps):
        th = step * k / R
        gt = (0.0, 0.0, th + np.pi / 2, R * np.cos(th) - R, R * np.sin(th), synth.SENSOR_HEIGHT)  # starts at the origin, heading +y
        _, _, _, cloud, _ = synth.make_scan(world, rings, 1800, gt_pose=gt, seed=7000 + k, full=True)
        ring = np.floor(cloud[:, 3]).astype(np.int64)
        a = np.ascontiguousarray(cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))][:, :4], np.float32)
        f.write(np.uint32(len(a)).tobytes())
        f.write(a.tobytes())
print("%d sweeps of %d rings generated in %.0f s" % (sweeps, rings, time.time() - t0), flush=True)
exe = os.path.join(tmp, "node_threads")
libdir = os.path.dirname(pkg.lib_path())
subprocess.check_call(["g++", "-O2", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "node_threads.cpp"),
                       "-o", exe, "-L", libdir, "-llslam_hip", "-Wl,-rpath," + libdir, "-lpthread"], timeout=300)
out = subprocess.run([exe, path, "8"], capture_output=True, text=True, timeout=900)
print(out.stdout.strip())
if out.returncode != 0:
    print("FAILED rc %d: %s" % (out.returncode, out.stderr[-600:]))
    sys.exit(1)
w = [l for l in out.stdout.splitlines() if l.startswith("NODE_THREADS")][0].split()
v = {w[i]: float(w[i + 1]) for i in range(1, len(w) - 1, 2)}
th = step * (sweeps - 1) / R
chord = float(np.hypot(R * np.cos(th) - R, R * np.sin(th)))
print("distance of the last mapped pose from the start %.2f m, of the drive's last pose %.2f m (difference %.2f m over %.0f m driven)" %
      (v["travelled_m"], chord, abs(v["travelled_m"] - chord), step * (sweeps - 1)))
import shutil
shutil.rmtree(tmp, ignore_errors=True)
sys.exit(0 if abs(v["travelled_m"] - chord) < 0.02 * step * sweeps + 1.0 else 2)
