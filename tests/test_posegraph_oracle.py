"""Pose-graph oracle (numpy fp64, parity unpinned: g2o is not available) -- internal consistency."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import posegraph_oracle as po  # noqa: E402


def test_error_is_zero_at_consistent_poses_and_update_is_right_multiplication():
    rng = np.random.default_rng(0)
    q = rng.normal(size=(5, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses = np.concatenate([rng.normal(size=(5, 3)), q], 1)
    ij = np.array([[0, 1], [1, 2], [3, 2], [4, 0]], np.int32)
    meas = po.pose_mul(po.pose_inv(poses[ij[:, 0]]), poses[ij[:, 1]])
    assert np.abs(po.edge_error(poses, ij, meas)).max() < 1e-12
    d = rng.normal(0, 0.01, (5, 6))
    moved = po.oplus(poses, d.ravel(), fixed=0)
    assert np.allclose(moved[0], poses[0])
    rel = po.pose_mul(po.pose_inv(poses[1:]), moved[1:])
    assert np.allclose(rel[:, :3], d[1:, :3], atol=1e-12) and np.allclose(rel[:, 3:6], d[1:, 3:], atol=1e-12)


def test_lm_reduces_chi2_and_recovers_loop_consistency():
    g = po.make_graph(n_kf=120, n_loop=400, laps=3, radius=15.0)
    c0 = po.chi2(g["init"], g["ij"], g["meas"], g["info"])
    P, hist = po.optimize(g["init"], g["ij"], g["meas"], g["info"], max_iters=10)
    assert hist[-1]["chi2"] < 1e-2 * c0
    assert all(b["chi2"] <= a["chi2"] + 1e-12 for a, b in zip(hist, hist[1:]))
    assert np.allclose(P[0], g["init"][0])  # first vertex fixed (solver_g2o.cpp:55-59)


def test_sharded_linearization_sums_to_the_full_system():
    """What the RCCL all-reduce adds up: per-shard systems sum to the full one."""
    g = po.make_graph(n_kf=60, n_loop=150, laps=2, radius=12.0)
    H, b, c = po.linearize(g["init"], g["ij"], g["meas"], g["info"])
    ne = len(g["ij"])
    cuts = [0, ne // 3, 2 * ne // 3, ne]
    Hs, bs, cs = 0, 0, 0
    for a, e in zip(cuts, cuts[1:]):
        h_, b_, c_ = po.linearize(g["init"], g["ij"], g["meas"], g["info"], a, e)
        Hs, bs, cs = Hs + h_, bs + b_, cs + c_
    assert abs(Hs - H).max() < 1e-9 and np.abs(bs - b).max() < 1e-9 and abs(cs - c) < 1e-9
