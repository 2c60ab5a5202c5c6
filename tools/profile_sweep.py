#!/usr/bin/env python3
"""Profiling harness: N single sweeps (+ optional full GN loops) of the 64-ring workload.
Run under rocprofv3 (kernel trace or --pmc) on the GPU box."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=10)
ap.add_argument("--loops", type=int, default=0)
ap.add_argument("--rings", type=int, default=64)
ap.add_argument("--jtj-mode", type=int, default=0)
a = ap.parse_args()
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
pr = synth.make_problem(rings=a.rings, azimuth_steps=1800)
ctx = pkg.Context(0)
ctx.map_set(pr["map_corner"], pr["map_surf"])
ctx.scan_set(pr["corner"], pr["surf"])
for _ in range(a.sweeps):
    ctx.sweep(pr["init_pose"], jtj_mode=a.jtj_mode, taps=False)
opts = ctx.default_opts(); opts.jtj_mode = a.jtj_mode
for _ in range(a.loops):
    ctx.run(pr["init_pose"], opts)
print("done")
