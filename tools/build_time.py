#!/usr/bin/env python3
"""kd-tree build time of lslam_map_set on the bench map (device build vs LSLAM_HOST_TREE=1)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
pr = synth.make_problem(rings=16, azimuth_steps=900)
ctx = pkg.Context(0)
for rep in range(4):
    t0 = time.perf_counter()
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    dt = time.perf_counter() - t0
    i = ctx.map_info()
    print("map_set %.1f ms  (build %.2f ms, pack/upload %.2f ms) device=%d depth %d/%d nodes %d" % (
        1e3 * dt, i.build_ms, i.upload_ms, i.built_on_device, i.depth_corner, i.depth_surf, i.nodes_surf))
