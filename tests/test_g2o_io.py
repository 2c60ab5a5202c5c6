"""The .g2o text format (SolverG2O::save, pose_graph/solver_g2o.cpp:97-100): host-side reader on
the CPU, save -> read round trip on the GPU."""
import numpy as np
import pytest


def test_read_g2o_file_written_by_g2o_conventions(pkg, tmp_path):
    # ids need not be contiguous; g2o writes 6 significant digits; unknown tags are skipped
    txt = """# a comment-like unknown tag line is ignored
VERTEX_SE3:QUAT 10 0 0 0 0 0 0 1
FIX 10
VERTEX_SE3:QUAT 12 1.5 0.25 -0.125 0 0 0.382683 0.92388
VERTEX_SE3:QUAT 11 3 0 0 0 0 0 1
EDGE_SE3:QUAT 10 12 1.5 0.25 -0.125 0 0 0.382683 0.92388 0.8 0 0 0 0 0 0.4 0 0 0 0 0.8 0 0 0 1 0 0 2 0 1
EDGE_SE3:QUAT 12 11 1 0 0 0 0 0 1 2 0.1 0 0 0 0 2 0 0 0 0 2 0 0 0 2 0 0 2 0 2
"""
    f = tmp_path / "g.g2o"
    f.write_text(txt)
    g = pkg.PoseGraph.read_g2o(f)
    assert g["poses"].shape == (3, 7) and g["ij"].tolist() == [[0, 1], [1, 2]] and g["fixed"] == 0
    assert np.allclose(g["poses"][1], [1.5, 0.25, -0.125, 0, 0, 0.382683, 0.92388])
    assert np.allclose(np.diag(g["info"][0]), [0.8, 0.4, 0.8, 1, 2, 1])
    assert g["info"][1][0, 1] == 0.1 and g["info"][1][1, 0] == 0.1  # upper triangle mirrored
    with pytest.raises(pkg.LslamError):
        pkg.PoseGraph.read_g2o(tmp_path / "missing.g2o")
    bad = tmp_path / "bad.g2o"
    bad.write_text("VERTEX_SE3:QUAT 0 0 0 0 0 0 0 1\nEDGE_SE3:QUAT 0 5 0 0 0 0 0 0 1 " + " ".join(["1"] * 21) + "\n")
    with pytest.raises(pkg.LslamError):
        pkg.PoseGraph.read_g2o(bad)


@pytest.mark.gpu
def test_save_read_round_trip(pkg, synth, tmp_path):
    g = synth.make_pose_graph(n_kf=200, n_loop=400, laps=2, radius=20.0)
    pg = pkg.PoseGraph(0)
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    pg.optimize(3)
    est = pg.poses()
    f = tmp_path / "out.g2o"
    pg.save(f)
    r = pkg.PoseGraph.read_g2o(f)
    assert r["fixed"] == 0 and np.array_equal(r["ij"], np.asarray(g["ij"], np.int32))
    assert np.array_equal(r["poses"], est)        # 17 significant digits: exact
    assert np.array_equal(r["meas"], g["meas"]) and np.array_equal(r["info"], g["info"])
    # a graph loaded from the file optimises to the same estimates
    pg2 = pkg.PoseGraph(0)
    pg2.load(f)
    pg.optimize(2)
    pg2.optimize(2)
    # (to the damped solves' tolerance -- PCG to a relative residual of 1e-8: the first graph may already have switched its
    # second preconditioner level on, the loaded one starts without it)
    assert np.abs(pg2.poses() - pg.poses()).max() < 1e-7
