#!/usr/bin/env python3
"""Soak of the persistent pose-graph solver: the bench graph optimised N times in one process; every run must take the
persistent kernels for every solve (no exchange timeout) and end on the same bits."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("the-cooper-mapper_amd")
synth = importlib.import_module("the-cooper-mapper_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = synth.make_pose_graph()
ref = None
t0 = time.perf_counter()
for k in range(n):
    pg = pkg.PoseGraph(0)
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    its = pg.optimize(1000)
    st = pg.last_stats
    sig = (its, st.lm_trials, st.cg_iterations, st.chi2_final, pg.poses().tobytes())
    assert st.fused_solves == st.lm_trials, (k, st.fused_solves, st.lm_trials)
    if ref is None:
        ref = sig
    assert sig == ref, (k, sig[:4], ref[:4])
    pg.close()
print("pose-graph soak: %d runs, all %d LM iterations / %d PCG iterations, chi2 %.9f, identical bits, %.1f s" % (
    n, ref[0], ref[2], ref[3], time.perf_counter() - t0))
