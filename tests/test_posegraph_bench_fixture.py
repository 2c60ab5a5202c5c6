"""BASELINE config 4 end to end: the committed .g2o fixture IS the bench graph, and the device solver, run until its
own stopping rule ends it, lands on the optimum the numpy/SuperLU oracle reaches on the same graph
(tests/golden/make_posegraph_bench.py wrote both files; g2o itself is not in this image: parity unpinned, the
.g2o file is there for the external cross-check pose_graph/solver_g2o.cpp:79-100 invites)."""
import gzip
import os
import shutil
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _unzipped(tmp_path):
    f = tmp_path / "posegraph_bench.g2o"
    with gzip.open(os.path.join(GOLD, "posegraph_bench.g2o.gz"), "rb") as src, open(f, "wb") as dst:
        shutil.copyfileobj(src, dst)
    return f


def test_fixture_is_the_bench_graph(pkg, synth, tmp_path):
    g = synth.make_pose_graph()
    r = pkg.PoseGraph.read_g2o(_unzipped(tmp_path))  # host-side reader, no GPU
    assert r["fixed"] == 0
    assert np.array_equal(r["ij"], np.asarray(g["ij"], np.int32))
    assert np.array_equal(r["poses"], g["init"]) and np.array_equal(r["meas"], g["meas"])
    assert np.array_equal(r["info"], g["info"])
    z = np.load(os.path.join(GOLD, "posegraph_bench_optimum.npz"))
    assert z["poses"].shape == g["init"].shape and float(z["chi2"]) < 1e-4 * float(z["chi2_history"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("solve_tol", [None, 1e-3])
def test_device_solver_reaches_the_oracle_optimum(pkg, tmp_path, solve_tol):
    """solve_tol None: the default (1e-8: the direct solver's answer as far as LM can tell).  1e-3: an inexact LM
    (lslam_pg_set_solve_tolerance) -- another trajectory, the SAME optimum by the same four criteria."""
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "oracle"))
    import posegraph_oracle as po
    z = np.load(os.path.join(GOLD, "posegraph_bench_optimum.npz"))
    pg = pkg.PoseGraph(0)
    g = pg.load(_unzipped(tmp_path))
    if solve_tol is not None:
        pg.set_solve_tolerance(solve_tol)
    its = pg.optimize(1000)  # SolverG2O::optimize's limit (solver_g2o.cpp:16,90); the LM stopping rule ends it
    st = pg.last_stats
    est = pg.poses()
    assert 0 < its < 1000
    # chi2 within 1e-6 relative of the oracle's converged optimum, keyframe positions within 1e-4 m
    assert abs(st.chi2_final - float(z["chi2"])) <= 1e-6 * float(z["chi2"])
    assert np.abs(est[:, :3] - z["poses"][:, :3]).max() < 1e-4
    q = est[:, 3:] * np.sign((est[:, 3:] * z["poses"][:, 3:]).sum(1, keepdims=True))
    assert np.abs(q - z["poses"][:, 3:]).max() < 1e-6
    # stationarity, judged by the oracle's own linearisation at the device estimate
    H, b, c2 = po.linearize(est, g["ij"], g["meas"], g["info"])
    b[:6] = 0
    b0 = po.linearize(g["poses"], g["ij"], g["meas"], g["info"])[1]
    assert abs(c2 - st.chi2_final) <= 1e-9 * c2
    assert np.abs(b).max() < 1e-9 * np.abs(b0[6:]).max()
    pg.close()
