"""Threshold-adjacent parity fixtures for the accept / reject rules of the fit stage
(/root/reference/L_SLAM/src/util/feature_utils.h): findLine's lambda_2 > 5 lambda_1 (:148), findPlane's
0.2 m inlier bound (:196-200), the weight gates w > 0.1 of the corner (:63-75) and surface (:97-106)
coefficients, and the unguarded d = 0 case of getLinePointDistance (:17-26).

The oracle and the kernel were written from the same reading of that file, so their agreement alone cannot
find a misreading.  Here every rule is located by BISECTION on a one-parameter family of inputs down to two
ADJACENT fp32 values that flip the oracle's decision, and an INDEPENDENT float64 / numpy statement of the rule
(its own eigen-decomposition, least-squares plane, distances) must place the rule's threshold quantity at the
flip point within 1e-4 relative: a wrong constant, a wrong comparison direction or a wrong formula in the oracle
moves the flip point away from where the rule says it is.  The GPU part then feeds the adjacent pairs (one
accepted, one rejected) through the sweep kernel and requires the oracle's flags, bit for bit."""
import numpy as np
import pytest


def _adjacent_flip(f, lo, hi):
    """f(lo) != f(hi): shrink [lo, hi] to two adjacent float32 values with different f."""
    lo, hi = np.float32(lo), np.float32(hi)
    flo = f(lo)
    assert flo != f(hi)
    while True:
        mid = np.float32((np.float64(lo) + np.float64(hi)) / 2)
        if mid == lo or mid == hi:
            break
        if f(mid) == flo:
            lo = mid
        else:
            hi = mid
    assert np.nextafter(lo, hi) == hi
    return lo, hi, flo


# ---- the one-parameter families -----------------------------------------------------------------------------
LINE_BASE = np.array([[-0.8, 0, 0], [-0.4, 0, 0], [0.0, 0, 0], [0.4, 0, 0], [0.8, 0, 0]], np.float64)
LINE_SIDE = np.array([1.0, -1.0, 0.5, -1.0, 0.5])  # zero-mean spread in y


def line_points(s, centre):
    p = LINE_BASE.copy()
    p[:, 1] = LINE_SIDE * float(s)
    return (p + centre).astype(np.float32)


def plane_points(delta, centre):
    p = np.array([[-0.5, -0.5, 0], [0.5, -0.5, 0], [0.5, 0.5, 0], [-0.5, 0.5, 0], [0.07, -0.03, float(delta)]], np.float64)
    return (p + centre).astype(np.float32)


def cloud4(p):
    return np.concatenate([p, np.zeros((len(p), 1), np.float32)], 1)


def test_find_line_rule_flips_where_float64_says(oracle):
    c = np.array([3.0, -2.0, 1.0])
    f = lambda s: oracle.find_line(cloud4(line_points(s, c)), np.arange(5))[0]
    lo, hi, accepted_at_lo = _adjacent_flip(f, 0.05, 0.6)
    assert accepted_at_lo  # a thin cluster is a line, a fat one is not
    for s in (lo, hi):
        p = line_points(s, c).astype(np.float64)
        ev = np.linalg.eigvalsh(np.cov(p.T, bias=True))          # independent: numpy's covariance + LAPACK
        assert abs(ev[2] / ev[1] - 5.0) < 5e-4 * 5.0              # the flip sits at lambda_2 = 5 lambda_1
    # accepted line: A, B = centroid -+ 0.1 * principal direction (:150-152)
    ok, A, B = oracle.find_line(cloud4(line_points(lo, c)), np.arange(5))
    p = line_points(lo, c).astype(np.float64)
    w, v = np.linalg.eigh(np.cov(p.T, bias=True))
    d = v[:, 2] * np.sign(v[:, 2] @ (B - A).astype(np.float64))
    assert np.abs((A + B) / 2 - p.mean(0)).max() < 1e-5 and np.abs((B - A) / 2 - 0.1 * d).max() < 1e-4


def test_find_plane_inlier_bound_flips_where_float64_says(oracle):
    c = np.array([-4.0, 6.0, 0.5])
    f = lambda dl: oracle.find_plane(cloud4(plane_points(dl, c)), np.arange(5))[0]
    lo, hi, accepted_at_lo = _adjacent_flip(f, 0.05, 1.5)
    assert accepted_at_lo
    for dl in (lo, hi):
        p = plane_points(dl, c).astype(np.float64)
        n, *_ = np.linalg.lstsq(p, -np.ones(5), rcond=None)     # independent: LAPACK least squares of [x y z] n = -1
        n /= np.linalg.norm(n)
        D = -n @ p.mean(0)                                        # feature_utils.h:186-190: D from the centroid
        worst = np.abs(p @ n + D).max()
        assert abs(worst - 0.2) < 1e-4 * 0.2                      # the flip sits at the 0.2 m bound (strict '>')


def test_weight_gates_flip_where_float64_says(oracle):
    # corner: w = 1 - 0.9 |d| > 0.1  <=>  |d| < 1
    A, B = np.array([1.0, 2.0, 0.0], np.float32), np.array([1.0, 2.0, 0.2], np.float32)
    f = lambda d: oracle.corner_coeff(A, B, np.array([1.0 + float(d), 2.0, 0.1], np.float32))[0]
    lo, hi, acc = _adjacent_flip(f, 0.5, 1.5)
    assert acc
    for d in (lo, hi):
        assert abs((1.0 - 0.9 * float(d)) - 0.1) < 1e-5
    # the accepted coefficient: unit direction from the line to the point times w, residual w d
    ok, cf = oracle.corner_coeff(A, B, np.array([1.0 + float(lo), 2.0, 0.1], np.float32))
    w = 1.0 - 0.9 * float(lo)
    assert ok and np.abs(cf - np.array([w, 0, 0, w * float(lo)])).max() < 1e-5
    # surface: w = 1 - 0.9 |d| / sqrt(|X|) > 0.1  <=>  |d| < sqrt(|X|)
    plane = np.array([0.0, 0.0, 1.0, 0.0], np.float32)
    g = lambda z: oracle.surf_coeff(plane, np.array([3.0, 4.0, float(z)], np.float32))[0]
    lo, hi, acc = _adjacent_flip(g, 1.0, 4.0)
    assert acc
    for z in (lo, hi):
        X = np.array([3.0, 4.0, float(z)])
        assert abs((1.0 - 0.9 * abs(X[2]) / np.sqrt(np.linalg.norm(X))) - 0.1) < 1e-5


def test_point_on_the_line_gives_nan_direction(oracle):
    """getLinePointDistance does not guard d = 0 (feature_utils.h:17-26): the direction is 0/0, the weight is 1, the
    row is KEPT with NaN coefficients -- restated, not repaired."""
    A, B = np.array([0.0, 0.0, -0.1], np.float32), np.array([0.0, 0.0, 0.1], np.float32)
    ok, cf = oracle.corner_coeff(A, B, np.array([0.0, 0.0, 0.03], np.float32))
    assert ok and np.isnan(cf[:3]).all() and cf[3] == 0.0


CENTRES = [np.array([8.0 * i, 8.0 * j, 1.0]) for i in (-2, -1, 1, 2) for j in (-2, -1, 0, 1, 2)]  # 8 m apart, within 23 m


def _decide(oracle, kind, pts5, q):
    """the oracle's flags for one query against one five-point cluster, through its sweep (neighbours arrive in kNN order,
    which the fits' rounding depends on -- the decision is bisected on exactly the path the comparison takes)"""
    far = cloud4(np.array([[900.0 + i, 900.0, 0.0] for i in range(5)], np.float32))
    mc, ms = (cloud4(pts5), far) if kind == "corner" else (far, cloud4(pts5))
    qc, qs = (cloud4(q[None]), far[:0]) if kind == "corner" else (far[:0], cloud4(q[None]))
    r = oracle.sweep(oracle.kdtree(mc), oracle.kdtree(ms), qc, qs, np.zeros(6, np.float32))
    return int(r["flags"][0])


def _cases(oracle, side):
    """(kind, 5 map points, query) per rule; `side` 0 / 1 picks the member of every adjacent pair (the last value that is
    accepted / the first that is rejected).  Every cluster keeps its place in both variants: the fits depend on where
    the points lie (the plane fit [x y z] n = -1 is not translation invariant), so a pair is bisected where it is used."""
    cases = []
    it = iter(CENTRES)
    c = next(it)
    q1 = (c + [0.1, 0.3, 0.2]).astype(np.float32)
    pair = _adjacent_flip(lambda s: bool(_decide(oracle, "corner", line_points(s, c), q1) & 2), 0.05, 0.6)
    cases.append(("corner", line_points(pair[side], c), q1))
    c2 = next(it)
    q2 = (c2 + [0.1, 0.1, 0.3]).astype(np.float32)
    pair = _adjacent_flip(lambda dl: bool(_decide(oracle, "surf", plane_points(dl, c2), q2) & 2), 0.02, 1.5)
    cases.append(("surf", plane_points(pair[side], c2), q2))
    # weight gates: a fixed thin cluster / flat patch, the query moved across the gate in one-ulp steps
    c3 = next(it)
    pair = _adjacent_flip(lambda d: bool(_decide(oracle, "corner", line_points(0.0, c3), (c3 + [0.0, float(d), 0.0]).astype(np.float32)) & 4),
                          0.5, 1.5)
    cases.append(("corner", line_points(0.0, c3), (c3 + [0.0, float(pair[side]), 0.0]).astype(np.float32)))
    c4 = np.array([1.0, 0.5, 0.2])  # near the origin: the surface weight 1 - 0.9 |d| / sqrt(|X|) only trips within the 5 m^2 gate there
    pair = _adjacent_flip(lambda z: bool(_decide(oracle, "surf", plane_points(0.0, c4), (c4 + [0.0, 0.0, float(z)]).astype(np.float32)) & 4),
                          0.5, 2.0)
    cases.append(("surf", plane_points(0.0, c4), (c4 + [0.0, 0.0, float(pair[side])]).astype(np.float32)))
    c5 = next(it)  # the d = 0 row: a query exactly on the fitted line
    cases.append(("corner", line_points(0.0, c5), c5.astype(np.float32)))
    # plain accepted rows so that every tree holds a few clusters
    for k in range(3):
        c6 = next(it)
        cases.append(("corner", line_points(0.02, c6), (c6 + [0.2, 0.2, 0.1]).astype(np.float32)))
        c7 = next(it)
        cases.append(("surf", plane_points(0.01, c7), (c7 + [0.1, -0.2, 0.25]).astype(np.float32)))
    return cases


@pytest.mark.gpu
def test_device_takes_the_oracles_decisions_at_every_threshold(ctx, oracle):
    seen = []
    for side in (0, 1):
        cases = _cases(oracle, side)
        mc = np.concatenate([m for kind, m, q in cases if kind == "corner"])
        ms = np.concatenate([m for kind, m, q in cases if kind == "surf"])
        qc = np.stack([q for kind, m, q in cases if kind == "corner"])
        qs = np.stack([q for kind, m, q in cases if kind == "surf"])
        ctx.map_set(cloud4(mc), cloud4(ms))
        ctx.scan_set(cloud4(qc), cloud4(qs))
        pose = np.zeros(6, np.float32)
        tc, ts = oracle.kdtree(cloud4(mc)), oracle.kdtree(cloud4(ms))
        o = oracle.sweep(tc, ts, cloud4(qc), cloud4(qs), pose)
        for search in (1, 2):
            g = ctx.sweep(pose, jtj_mode=0, search_mode=search)
            assert np.array_equal(g["idx"], o["idx"]) and np.array_equal(g["flags"], o["flags"])
            nan_o, nan_g = np.isnan(o["coeff"]), np.isnan(g["coeff"])
            assert np.array_equal(nan_o, nan_g) and nan_o.any()            # the d = 0 row is there, NaN in both
            assert np.array_equal(g["coeff"][~nan_g].view(np.uint32), o["coeff"][~nan_o].view(np.uint32))
        seen.append(o["flags"].copy())
    # the four rule cases (line fit, plane fit, corner weight, surf weight) flip between the two variants, nothing else does
    nc = sum(1 for kind, m, q in _cases(oracle, 0) if kind == "corner")
    flips = np.nonzero(seen[0] != seen[1])[0].tolist()
    assert flips == [0, 1, nc, nc + 1], flips
    assert (seen[0][[0, nc]] & 2).all() and not (seen[1][[0, nc]] & 2).any()          # fit accepted -> rejected
    assert (seen[0][[1, nc + 1]] & 4).all() and not (seen[1][[1, nc + 1]] & 4).any()  # row kept -> dropped
