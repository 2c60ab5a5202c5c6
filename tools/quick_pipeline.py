#!/usr/bin/env python3
"""bench.py's sweep_pipeline legs alone (the per-sweep chain: stage times, the three-nodelet chain in C++ and in Python)."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
ctx = pkg.Context(0)
for rings in [int(a) for a in sys.argv[1:]] or [16, 64]:
    r = bench.sweep_pipeline_leg(pkg, synth, ctx, rings, np)
    print(rings, json.dumps({k: r[k] for k in ("ms", "ms_per_sweep", "node_threads", "node_threads_python")}, default=float))
    print("   odometry by sweep", r["odometry"]["iterations_per_sweep"], r["odometry"]["ms_by_sweep"])
