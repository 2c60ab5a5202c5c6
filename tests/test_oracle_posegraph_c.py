"""oracle/posegraph_oracle.c (compiled LM: analytic Jacobians, RCM-ordered envelope block Cholesky -- the CPU baseline of the
pose-graph leg) against oracle/posegraph_oracle.py (numpy: numeric Jacobians, SuperLU).  CPU only.  Parity of both with the
reference is unpinned: g2o (pose_graph/solver_g2o.cpp:16,79-95) is not available; they restate its published conventions
independently of each other -- different Jacobians, different linear solver."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import posegraph_oracle as po  # noqa: E402
import posegraph_oracle_c as pc  # noqa: E402


def _graph(**kw):
    return po.make_graph(**kw)


def test_linearisation_matches_the_numpy_oracle():
    g = _graph(n_kf=120, n_loop=400)
    H, b, c2 = po.linearize(g["init"], g["ij"], g["meas"], g["info"])
    diag, bc, c2c = pc.linearize(g["init"], g["ij"], g["meas"], g["info"], fixed=0)
    assert abs(c2 - c2c) <= 1e-9 * c2
    Hd = H.toarray()
    scale = np.abs(Hd).max()
    for v in range(1, 120):  # vertex 0 is fixed: identity row in the C oracle
        assert np.abs(diag[v] - Hd[6 * v:6 * v + 6, 6 * v:6 * v + 6]).max() <= 1e-6 * scale, v  # numeric Jacobians: h = 1e-6
    assert np.array_equal(diag[0], np.eye(6)) and not bc[:6].any()
    assert np.abs(bc[6:] - b[6:]).max() <= 1e-6 * np.abs(b).max()


def test_damped_solve_matches_superlu():
    g = _graph(n_kf=150, n_loop=500, seed=3)
    H, b, c2 = po.linearize(g["init"], g["ij"], g["meas"], g["info"])
    for lam in (1e-6 * H.diagonal().max(), 1e-2 * H.diagonal().max()):
        dx = pc.solve(g["init"], g["ij"], g["meas"], g["info"], lam, fixed=0)
        ref = po.solve_damped(H, b, lam, 0)
        assert not dx[:6].any()
        assert np.abs(dx - ref).max() <= 1e-5 * np.abs(ref).max()  # Jacobians differ by the finite-difference error


def test_lm_run_matches_the_numpy_oracle():
    g = _graph(n_kf=200, n_loop=700, seed=11)
    ref, hist = po.optimize(g["init"], g["ij"], g["meas"], g["info"], fixed=0, max_iters=12)
    out, st = pc.optimize(g["init"], g["ij"], g["meas"], g["info"], fixed=0, max_iters=12)
    assert st.status == 0 and st.iterations == len(hist) and st.trials == sum(h["trials"] for h in hist)
    assert abs(st.chi2_final - hist[-1]["chi2"]) <= 1e-6 * hist[-1]["chi2"]
    assert np.abs(out[:, :3] - ref[:, :3]).max() <= 1e-5  # twelve iterations apart through different Jacobians (central differences, h = 1e-6)
    assert st.chi2_final < 1e-3 * st.chi2_initial  # it optimised something


def test_envelope_order_is_a_permutation_and_small():
    """RCM on a chain with loop closures: the envelope is far smaller than the dense lower triangle; disconnected parts and a
    fixed vertex in the middle are handled."""
    g = _graph(n_kf=300, n_loop=1200)
    out, st = pc.optimize(g["init"], g["ij"], g["meas"], g["info"], fixed=0, max_iters=1)
    assert st.env_blocks < 0.35 * 300 * 301 / 2 and st.bandwidth < 300
    # two components: the second one floats (singular without damping) but LM's lambda keeps the factorisation definite
    ij2 = np.concatenate([g["ij"], g["ij"] + 300]).astype(np.int32)
    poses2 = np.concatenate([g["init"], g["init"]])
    out2, st2 = pc.optimize(poses2, ij2, np.concatenate([g["meas"]] * 2), np.concatenate([g["info"]] * 2), fixed=150, max_iters=3)
    assert st2.status == 0 and st2.iterations == 3 and np.array_equal(out2[150], poses2[150])
