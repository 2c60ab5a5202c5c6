#!/usr/bin/env python3
"""Pose-graph LM timing (single GPU) for profiling."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
g = synth.make_pose_graph()
pg = pkg.PoseGraph(0)
pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
pg.optimize(1)
pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
t0 = time.perf_counter()
pg.build()
print("build %.1f ms" % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter()
it = pg.optimize(int(os.environ.get("ITERS", "10")))
dt = time.perf_counter() - t0
st = pg.last_stats
print("LM iters %d in %.1f ms -> %.1f it/s ; trials %d cg %d chi2 %.3e -> %.3e" % (it, 1e3 * dt, it / dt, st.lm_trials, st.cg_iterations, st.chi2_initial, st.chi2_final))
