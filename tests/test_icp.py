"""Coarse loop-closure alignment (LoopDetector::corseMatching, pose_graph/loop_detector.hpp:232-255 = PCL's
IterativeClosestPoint with defaults; PARITY UNPINNED: PCL is not available).  CPU: the oracle restatement
recovers known rigid motions.  GPU: lslam_icp_align against the oracle (independent nearest-neighbour search
and SVD) -- same iteration counts, transforms within 1e-4 m / 1e-5."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _room(rng, n=6000):
    """floor + two walls + a pillar: enough structure in every direction"""
    a = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.normal(0, 0.01, n)], 1)
    b = np.stack([np.full(n // 2, 10.0) + rng.normal(0, 0.01, n // 2), rng.uniform(-10, 10, n // 2), rng.uniform(0, 4, n // 2)], 1)
    c = np.stack([rng.uniform(-10, 10, n // 2), np.full(n // 2, -10.0) + rng.normal(0, 0.01, n // 2), rng.uniform(0, 4, n // 2)], 1)
    d = np.stack([2 + rng.normal(0, 0.05, 400), -3 + rng.normal(0, 0.05, 400), rng.uniform(0, 3, 400)], 1)
    p = np.concatenate([a, b, c, d])
    return np.concatenate([p, np.zeros((len(p), 1))], 1).astype(np.float32)


def _rigid(yaw, pitch, t):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = Rz @ Ry, t
    return T


def _problem(seed):
    rng = np.random.default_rng(seed)
    tgt = _room(rng)
    Tgt = _rigid(0.03, -0.01, [0.25, -0.15, 0.05])       # source frame -> target frame (ground truth)
    pick = rng.choice(len(tgt), 5000, replace=False)
    src = tgt[pick].copy()
    inv = np.linalg.inv(Tgt)
    src[:, :3] = (tgt[pick, :3].astype(np.float64) @ inv[:3, :3].T + inv[:3, 3]).astype(np.float32)
    return tgt, src, Tgt


def test_icp_oracle_recovers_a_rigid_motion():
    import icp_oracle
    tgt, src, Tgt = _problem(1)
    T, conv, its, fit = icp_oracle.icp_align(tgt, src, np.eye(4))
    assert conv and 1 <= its <= 10
    assert np.abs(T[:3, 3] - Tgt[:3, 3]).max() < 0.02 and np.abs(T[:3, :3] - Tgt[:3, :3]).max() < 2e-3
    assert fit < 1e-3
    # an empty reference: loop_detector.hpp:233-235
    T2, conv2, its2, _ = icp_oracle.icp_align(np.zeros((0, 4), np.float32), src, np.eye(4))
    assert not conv2 and its2 == 0
    # two source points cannot be aligned (min_number_correspondences_ = 3)
    _, conv3, _, _ = icp_oracle.icp_align(tgt, src[:2], np.eye(4))
    assert not conv3


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_device_icp_matches_oracle(ctx, seed):
    import icp_oracle
    tgt, src, Tgt = _problem(seed)
    guess = _rigid(0.0, 0.0, [0.05, 0.02, 0.0]).astype(np.float32)
    T, conv, its, fit = ctx.icp_align(tgt, src, guess)
    To, convo, itso, fito = icp_oracle.icp_align(tgt, src, guess)
    assert conv == convo and its == itso
    assert np.abs(T[:3, 3] - To[:3, 3]).max() <= 1e-4 and np.abs(T[:3, :3] - To[:3, :3]).max() <= 1e-5
    assert abs(fit - fito) <= 1e-6 + 1e-3 * fito
    assert np.abs(T[:3, 3] - Tgt[:3, 3]).max() < 0.02
    assert np.array_equal(T[3], [0, 0, 0, 1])


@pytest.mark.gpu
def test_device_icp_guards(ctx):
    tgt, src, _ = _problem(3)
    T, conv, its, _ = ctx.icp_align(np.zeros((0, 4), np.float32), src, np.eye(4))
    assert not conv and its == 0 and np.array_equal(T, np.eye(4, dtype=np.float32))
    T, conv, its, _ = ctx.icp_align(tgt, src[:2], np.eye(4))
    assert not conv
    # a correspondence gate that keeps nothing: fewer than 3 correspondences
    far = src.copy()
    far[:, :3] += 500.0
    T, conv, its, _ = ctx.icp_align(tgt, far, np.eye(4), max_correspondence_distance=1.0)
    assert not conv and its == 0
