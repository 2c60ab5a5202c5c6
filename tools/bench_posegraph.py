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
if os.environ.get("PK_CLOCKS"):  # -DLSLAM_PK_CLOCKS build: where the persistent PCG kernel's iterations go
    import ctypes as C
    clk = (C.c_double * 12)()
    pg.lib.lslam_pg_debug_clocks.argtypes = [C.c_void_p, C.c_double * 12]
    pg.lib.lslam_pg_debug_clocks(pg.h, clk)
    names = ["exchange A (poll) + reduce 2", "columns", "items", "row sums + reduce 7 + publish", "exchange B (poll)", "reduce 1", "update + rcs",
             "jacobi + coarse rows", "reduce 8 + publish"]
    n = max(1, st.cg_iterations)
    print("persistent kernel, us per PCG iteration (workgroup 0): " + ", ".join("%s %.2f" % (names[i], clk[i] / 100.0 / n) for i in range(9)) +
          " | total %.2f" % (sum(clk[:9]) / 100.0 / n))
