"""The neighbour certificate of the production sweep (csrc/lslam_kernels.hip, sweep_body, CERT) as a statement about point
sets, checked on the CPU by brute force -- independent of the kd-tree, the kernels and the oracle.

Claim: let q be a query, F its five nearest map points, and lb6 a lower bound of the squared distance from q of every map point
outside F.  Move the query by delta to q'.  If the farthest of F from q' is closer than sqrt(lb6) - delta (with the kernel's
fp32 slack on both sides), then the five nearest map points of q' are the same five.  The bound a search keeps
(knn5_search<TRACK>) is never above the true sixth distance, so every bound between 0 and d6 must be safe; a certified point
keeps q and lb6 of its LAST SEARCH, so chains of small moves measured from the same q must be safe too."""
import numpy as np

SLACK_LO, SLACK_HI, ABS = np.float32(0.99999), np.float32(1.00001), np.float32(1e-6)


def certified(q_prev, lb6, q_new, five_pts):
    """sweep_body's test, in fp32 with its constants."""
    q_prev, q_new, five_pts = q_prev.astype(np.float32), q_new.astype(np.float32), five_pts.astype(np.float32)
    e = q_new - q_prev
    delta = np.sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2], dtype=np.float32)
    d = five_pts - q_new
    dj = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    u = dj.max()
    sl = np.sqrt(np.float32(lb6), dtype=np.float32) * SLACK_LO - delta * SLACK_HI - ABS
    distinct = len(set(dj.tolist())) == 5
    return bool(distinct and np.sqrt(u, dtype=np.float32) * SLACK_HI < sl)


def five_nearest(pts, q):
    d2 = ((pts.astype(np.float64) - q.astype(np.float64)) ** 2).sum(1)
    order = np.argsort(d2, kind="stable")
    return order[:5], d2[order[4]], d2[order[5]]


def test_certificate_implies_the_same_five_neighbours():
    rng = np.random.default_rng(12)
    n_cert = n_changed_uncert = 0
    for trial in range(6):
        # surfaces sampled near a voxel grid's pitch, like the map: points on planes with jitter, plus clutter
        n = 4000
        plane = np.c_[rng.uniform(-20, 20, (n, 2)), rng.normal(0, 0.02, n)]
        wall = np.c_[rng.uniform(-20, 20, n // 2), np.full(n // 2, 7.0) + rng.normal(0, 0.02, n // 2), rng.uniform(0, 5, n // 2)]
        pts = np.r_[plane, wall, rng.uniform(-20, 20, (300, 3))].astype(np.float32)
        for _ in range(400):
            q = (pts[rng.integers(len(pts))] + rng.normal(0, 0.3, 3)).astype(np.float32)
            five, d5, d6 = five_nearest(pts, q)
            # any lower bound up to the true sixth distance (the search's bound is one of them)
            lb6 = np.float32(d6 * rng.choice([1.0, 0.999, 0.9, 0.6, rng.uniform(0, 1)]))
            # a chain of moves, all measured from the position of the last search
            q_new = q.copy()
            for step in range(3):
                q_new = (q_new + rng.normal(0, 1.0, 3) * rng.choice([1e-4, 1e-3, 1e-2, 5e-2, 0.2])).astype(np.float32)
                now, _, _ = five_nearest(pts, q_new)
                same = set(now.tolist()) == set(five.tolist())
                if certified(q, lb6, q_new, pts[five]):
                    n_cert += 1
                    assert same, (trial, q, q_new, lb6, d5, d6)
                elif not same:
                    n_changed_uncert += 1
    assert n_cert > 500 and n_changed_uncert > 50  # both outcomes really occur


def test_certificate_refuses_ties_and_missing_bounds():
    pts = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, 3]], np.float32)
    q = np.zeros(3, np.float32)
    # five equidistant neighbours: the order nanoflann returns depends on its visit order -> never certified
    assert not certified(q, 9.0, q, pts[:5])
    # lb6 = 0 (no bound kept): never certified, however small the move
    five = np.array([[0.1, 0, 0], [0.2, 0, 0], [0.3, 0, 0], [0.4, 0, 0], [0.5, 0, 0]], np.float32)
    assert not certified(q, 0.0, q, five)
    assert certified(q, 4.0, q + np.float32(1e-3), five)
