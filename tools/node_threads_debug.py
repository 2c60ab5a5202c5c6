#!/usr/bin/env python3
"""Build and run tools/cpp/node_threads.cpp on a few synthetic sweeps, sequentially and threaded (diagnostics)."""
import importlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
rings = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
lo, hi = (-15.0, 15.0) if rings == 16 else (-24.9, 2.0)
world = synth.World(half_extent=175.0)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "sweeps.bin")
with open(path, "wb") as f:
    f.write(np.uint32(rings).tobytes() + np.float32(lo).tobytes() + np.float32(hi).tobytes() + np.uint32(sweeps).tobytes())
    for k in range(sweeps):
        gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
        _, _, _, cloud, _ = synth.make_scan(world, rings, 1800, gt_pose=gt, seed=300 + k, full=True)
        ring = np.floor(cloud[:, 3]).astype(np.int64)
        a = np.ascontiguousarray(cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))][:, :4], np.float32)
        f.write(np.uint32(len(a)).tobytes()); f.write(a.tobytes())
exe = os.path.join(tmp, "node_threads")
libdir = os.path.dirname(pkg.lib_path())
subprocess.check_call(["g++", "-O1", "-g", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "node_threads.cpp"),
                       "-o", exe, "-L", libdir, "-llslam_hip", "-Wl,-rpath," + libdir, "-lpthread"])
envs = [{}] + [dict(kv.split("=") for kv in e.split(",")) for e in os.environ.get("TRY_ENVS", "").split(";") if e]
for extra in envs:
  for mode in (["seq"], []):
    out = subprocess.run([exe, path, "4"] + mode, capture_output=True, text=True, timeout=300, env=dict(os.environ, **extra))
    print(extra, mode or "threads", "rc", out.returncode, out.stdout.strip(), "|", out.stderr.strip()[-600:])
    if out.returncode < 0 and os.path.exists("/opt/rocm/bin/rocgdb") and not extra:
        dbg = subprocess.run(["/opt/rocm/bin/rocgdb", "-batch", "-ex", "run", "-ex", "bt 25", "-ex", "info threads", "--args", exe, path, "4"] + mode,
                             capture_output=True, text=True, timeout=600)
        print(dbg.stdout[-5000:])
        print(dbg.stderr[-1500:])
