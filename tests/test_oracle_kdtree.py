"""Pins oracle/lslam_oracle.c's kd-tree (nanoflann v1.2.3 restatement) to the
reference: committed golden vectors (generated from the reference's own
nanoflann.hpp) and, where oracle/_ref was built, the reference library live."""
import numpy as np
import pytest

from oracle_lib import RefNanoflann, have_ref


def test_oracle_knn_matches_reference_goldens(oracle, goldens):
    for name, g in goldens.items():
        tree = oracle.kdtree(g["pts"])
        k = min(5, len(g["pts"]))
        idx, d2 = tree.knn(g["queries"], 5)
        assert np.array_equal(idx[:, :k], g["idx"][:, :k]), name
        assert np.array_equal(d2[:, :k].view(np.int32), g["d2"][:, :k].view(np.int32)), name


def test_goldens_are_sorted_and_exact(goldens):
    """Sanity of the fixtures themselves: ascending d2 and true nearest neighbours (fp64 brute force)."""
    for name, g in goldens.items():
        pts, q = g["pts"].astype(np.float64), g["queries"].astype(np.float64)
        assert np.all(np.diff(g["d2"], axis=1) >= 0), name
        sub = slice(0, 200)
        D = ((q[sub, None, :] - pts[None, :, :]) ** 2).sum(-1)
        kth = np.sort(D, axis=1)[:, 4] if len(pts) >= 5 else None
        if kth is not None:
            assert np.allclose(g["d2"][sub, 4], kth, rtol=1e-5, atol=1e-6), name


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (reference absent)")
@pytest.mark.parametrize("kind", ["uniform", "clustered", "lattice", "stride8"])
def test_oracle_knn_matches_reference_live(oracle, kind):
    rng = np.random.default_rng(hash(kind) % 1000)
    if kind == "uniform":
        pts = rng.uniform(-50, 50, (50000, 3)).astype(np.float32)
        q = rng.uniform(-55, 55, (2000, 3)).astype(np.float32)
    elif kind == "clustered":
        c = rng.uniform(-50, 50, (40, 3))
        pts = (c[rng.integers(0, 40, 30000)] + rng.normal(0, 0.5, (30000, 3))).astype(np.float32)
        q = (c[rng.integers(0, 40, 2000)] + rng.normal(0, 2.0, (2000, 3))).astype(np.float32)
    elif kind == "lattice":
        g = np.arange(-10, 10, 0.2, dtype=np.float32)
        X, Y = np.meshgrid(g, g)
        pts = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size, np.float32)], 1)
        q = pts[rng.integers(0, len(pts), 2000)] + np.float32(0.1)
    else:  # PointXYZI layout: 8 floats per point
        pts = np.zeros((20000, 8), np.float32)
        pts[:, :3] = rng.normal(0, 10, (20000, 3))
        pts[:, 4] = rng.uniform(0, 64, 20000)
        q = np.zeros((1000, 8), np.float32)
        q[:, :3] = rng.normal(0, 12, (1000, 3))
    ref = RefNanoflann(pts)
    tree = oracle.kdtree(pts)
    ri, rd = ref.knn(q, 5)
    oi, od = tree.knn(q, 5)
    assert np.array_equal(oi, ri)
    assert np.array_equal(od.view(np.int32), rd.view(np.int32))


def test_oracle_tree_structure(oracle):
    rng = np.random.default_rng(3)
    pts = rng.normal(0, 5, (5000, 3)).astype(np.float32)
    tree = oracle.kdtree(pts)
    nodes = tree.nodes()
    vind = tree.vind()
    assert sorted(vind.tolist()) == list(range(len(pts)))
    leaf = nodes["kind"] == 0
    # leaves partition [0, n) and hold at most 10 points (nanoflann.hpp:478-483, :939)
    spans = sorted(zip(nodes["a"][leaf].tolist(), nodes["b"][leaf].tolist()))
    assert spans[0][0] == 0 and spans[-1][1] == len(pts)
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert max(b - a for a, b in spans) <= 10
    # inner nodes: divlow <= divhigh, preorder child numbering
    inner = ~leaf
    assert np.all(nodes["divlow"][inner] <= nodes["divhigh"][inner])
    assert np.all(nodes["child2"][inner] > np.nonzero(inner)[0] + 1)
