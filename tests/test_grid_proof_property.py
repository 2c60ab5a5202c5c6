"""The grid sweep's completeness proof (csrc/lslam_grid.hpp, knn5_grid) as a statement about point sets, checked on the CPU by
brute force -- independent of the kernels, the kd-tree and the oracle (the style of tests/test_certificate_property.py).

Restated here in numpy with the kernel's constants (cell 0.6 m, GRID_U_SLACK, the 1e-5 pads, GRID_NF_PRUNE_SLACK, the 10-bit
key truncation, the clip margin): place the map points in cells by u(v) = fl(fl(v - org) * inv_c); for a query look only at
the 27 cells around its own, clipped to the box of radius sqrt(bound) + margin when an upper bound of the fifth distance is
known; keep the six smallest TRUNCATED keys; re-measure the six exactly; accept iff the proof obligations of the header hold.

Claim: whenever the probe says PROVEN, the five it returns are the five nearest map points of the query by the reference's own
fp32 distance (util/nanoflann.hpp:364-372 L2_Simple: x, y, z accumulated in order), in ascending order, all distinct, and
every other map point is farther than the fifth by more than nanoflann's own pruning bound can round (GRID_NF_PRUNE_SLACK) --
so nanoflann's traversal (nanoflann.hpp:1433-1497), whatever its visit order, returns the same five.  And GRID_FAR: a finite
query whose cell is not an interior cell is farther than the sqrt(5) m acceptance gate (ScanMatch.cpp:102,120) from every
map point."""
import numpy as np

F = np.float32
CELL = F(0.6)
U_SLACK = F(2.0e-3)
CLIP_MARGIN = F(3.0e-3)
NF_SLACK = F(1.2e-5)
KEY_SLACK = F(1.0e-6)   # the loop's key distance (one rounded square, two fused multiply-adds) against the exact one
ID_BITS = 10
FLT_MAX = np.finfo(np.float32).max


def margin_cells(c):
    return int(F(2.2361) / c) + 3


def dist2(q, p):
    """L2_Simple::evalMetric in fp32, x -> y -> z."""
    dx, dy, dz = F(q[0]) - p[..., 0], F(q[1]) - p[..., 1], F(q[2]) - p[..., 2]
    return ((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + (dz * dz).astype(F)


def key_dist2(q, p):
    """knn5_grid's loop: fl(dx dx), then two fused multiply-adds (exact product, one rounding each)."""
    d = (np.asarray(q, F)[None, :] - p.astype(F)).astype(F).astype(np.float64)
    a = (d[:, 0] * d[:, 0]).astype(F).astype(np.float64)
    a = (d[:, 1] * d[:, 1] + a).astype(F).astype(np.float64)   # double holds the exact product and sum of fp32 operands to < 1/2 ulp(fp32): one rounding
    return (d[:, 2] * d[:, 2] + a).astype(F)


class Grid:
    def __init__(self, pts, c=CELL):
        self.pts = pts.astype(F)
        self.c = F(c)
        self.inv_c = F(1.0) / self.c
        m = margin_cells(self.c)
        lo, hi = self.pts.min(0), self.pts.max(0)
        self.org = (lo - F(m) * self.c).astype(F)
        self.dims = (np.floor(((hi - self.org).astype(F) * self.inv_c).astype(F)).astype(int) + 1 + m)
        u = self.u(self.pts)
        self.cell = np.floor(u).astype(int)
        self.by_cell = {}
        for i, k in enumerate(map(tuple, self.cell)):  # ascending original index inside a cell, as the build's rank pass leaves them
            self.by_cell.setdefault(k, []).append(i)

    def u(self, v):
        return ((v.astype(F) - self.org).astype(F) * self.inv_c).astype(F)

    def probe(self, q, bound=FLT_MAX):
        """-> (verdict, five indices, five distances); verdict 'proven' | 'unproven' | 'far'."""
        q = q.astype(F)
        u = self.u(q)
        n = self.dims
        interior = all(u[a] >= 1.0 and u[a] < F(n[a] - 1) for a in range(3))
        if not interior:
            return ("far" if np.isfinite(u).all() else "unproven"), None, None
        f0 = np.floor(u)
        e = u - f0
        wall = min(min(e[a], F(1.0) - e[a]) for a in range(3))
        rg = self.c * ((F(1.0) - U_SLACK) + wall)
        rg2 = F((rg * rg) * (F(1.0) - (F(1.0e-5) + NF_SLACK)))
        lo, hi = f0 - 1, f0 + 1
        clip_lo2 = FLT_MAX
        if bound < 1e30:
            rb = F(np.sqrt(F(bound)) * (F(1.0) + F(1.0e-5)) + CLIP_MARGIN)
            rbc = F(rb * self.inv_c)
            lo = np.maximum(lo, np.floor(u - rbc))
            hi = np.minimum(hi, np.floor(u + rbc))
            cl = F(rb - U_SLACK * self.c)
            clip_lo2 = F((cl * cl) * (F(1.0) - F(1.0e-5)))
        cand = []   # positions in the order the probe walks them: rows (z outer, y inner), cells of a row x-ascending
        for jz in range(int(f0[2]) - 1, int(f0[2]) + 2):
            for jy in range(int(f0[1]) - 1, int(f0[1]) + 2):
                if not (lo[1] <= jy <= hi[1] and lo[2] <= jz <= hi[2]):
                    continue
                for jx in range(int(lo[0]), int(hi[0]) + 1):
                    cand += self.by_cell.get((jx, jy, jz), [])
        if len(cand) > 64 * 9:
            return "unproven", None, None
        cand = np.array(cand, int)
        d = dist2(q, self.pts[cand]) if len(cand) else np.zeros(0, F)
        dk = key_dist2(q, self.pts[cand]) if len(cand) else np.zeros(0, F)   # what the loop orders by
        keys = (dk.view(np.uint32) & np.uint32(~((1 << ID_BITS) - 1) & 0xFFFFFFFF)).astype(np.uint64) * 1024 + np.arange(len(cand), dtype=np.uint64)
        order = np.argsort(keys, kind="stable")[:6]        # the six smallest truncated keys (ties by place in the walk)
        surv, e6 = cand[order], d[order]
        if len(surv) < 5:
            return "unproven", None, None
        t6 = FLT_MAX if len(surv) < 6 else F(F(np.uint32(keys[order[5]] // 1024).view(F)) * (F(1.0) - KEY_SLACK))
        idx5, d5 = list(surv[:5]), list(e6[:5])
        lb = e6[5] if len(surv) == 6 else FLT_MAX
        if not all(e6[i] <= e6[i + 1] for i in range(len(e6) - 1)):   # two of the six within 2^-13: the search's own sorted insert
            o2 = np.argsort(e6[:5], kind="stable")
            idx5, d5 = [surv[i] for i in o2], [e6[i] for i in o2]
            if len(surv) == 6:
                lb = max(e6[5], d5[4])
                if e6[5] < d5[4]:   # knn_insert_sorted: strictly smaller goes in before
                    pos = int(np.searchsorted(np.array(d5, F), e6[5], side="right"))
                    d5 = d5[:pos] + [e6[5]] + d5[pos:4]
                    idx5 = idx5[:pos] + [surv[5]] + idx5[pos:4]
        if lb < 1e30:
            lb = F(lb * (F(1.0) - NF_SLACK))
        lb6 = min(F(lb), t6, rg2, clip_lo2)
        distinct = all(d5[i] < d5[i + 1] for i in range(4))
        return ("proven" if distinct and d5[4] < lb6 else "unproven"), np.array(idx5), np.array(d5, F)


def carried_bound(q_prev, d5_prev, q_new):
    """grid_carried_bound (csrc/lslam_kernels.hip): the previous sweep's five seen from the new position."""
    e = (q_new.astype(F) - q_prev.astype(F)).astype(F)
    delta = np.sqrt(F((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]), dtype=F)
    r = F((np.sqrt(F(d5_prev), dtype=F) + delta) * (F(1.0) + F(1.0e-5)) + F(1.0e-6))
    return F(r * r)


def voxel_like_map(rng, n_plane=5000):
    """Planes and a wall sampled near a voxel grid's pitch with jitter, poles, clutter -- and a lattice patch (exact ties)."""
    g = np.arange(-14.0, 14.0, 0.4)
    gx, gy = np.meshgrid(g, g)
    ground = np.c_[gx.ravel(), gy.ravel(), np.zeros(gx.size)] + rng.uniform(-0.12, 0.12, (gx.size, 3)) * [1, 1, 0.1]
    h = np.arange(0, 5, 0.4)
    wx, wz = np.meshgrid(g, h)
    wall = np.c_[wx.ravel(), np.full(wx.size, 6.3), wz.ravel()] + rng.uniform(-0.1, 0.1, (wx.size, 3)) * [1, 0.1, 1]
    poles = np.concatenate([np.c_[np.full(len(h), x), np.full(len(h), y), h] for x, y in rng.uniform(-12, 12, (15, 2))])
    lattice = np.c_[np.meshgrid(np.arange(20, 24, 0.4), np.arange(0, 4, 0.4))[0].ravel(),
                    np.meshgrid(np.arange(20, 24, 0.4), np.arange(0, 4, 0.4))[1].ravel(), np.zeros(100)]
    return np.concatenate([ground, wall, poles, lattice, rng.uniform(-14, 14, (200, 3))]).astype(F)


def test_proven_means_nanoflanns_five_with_margin():
    rng = np.random.default_rng(5)
    n_proven = n_unproven = n_bounded_proven = n_clipped_away = 0
    for trial in range(3):
        pts = voxel_like_map(rng)
        G = Grid(pts)
        for _ in range(700):
            base = pts[rng.integers(len(pts))]
            q = (base + rng.normal(0, rng.choice([0.02, 0.15, 0.4]), 3)).astype(F)
            d_all = dist2(q, pts)
            order = np.argsort(d_all, kind="stable")
            truth, d_true = order[:5], d_all[order[:5]]
            d6 = d_all[order[5]]
            for bounded in (False, True):
                bound = FLT_MAX
                if bounded:  # what the next sweep knows: the point was at q_prev, its fifth neighbour d5_prev away (exact there)
                    q_prev = (q + rng.normal(0, rng.choice([1e-3, 1e-2, 0.1]), 3)).astype(F)
                    d_prev = np.sort(dist2(q_prev, pts))[4]
                    bound = min(F(5.0) * (F(1.0) + F(1e-5)), carried_bound(q_prev, d_prev, q))
                    # the carried bound really bounds the fifth distance (triangle inequality) -- or the fifth neighbour is
                    # beyond the sqrt(5) m gate, where the reference looks nothing up (ScanMatch.cpp:102,120)
                    assert bound >= d_true[4] or d_true[4] >= 5.0
                verdict, idx5, d5 = G.probe(q, bound)
                if verdict == "proven":
                    n_proven += 1
                    n_bounded_proven += bounded
                    assert np.array_equal(idx5, truth), (trial, q, idx5, truth)
                    assert np.array_equal(d5.view(np.uint32), d_true.view(np.uint32))
                    assert len(set(d5.tolist())) == 5
                    # every other map point is farther than the fifth by more than nanoflann's pruning bound can round
                    assert F(d6 * (F(1.0) - NF_SLACK)) >= d5[4] or d6 * (1.0 - 1.1e-5) > d5[4], (q, d5[4], d6)
                elif verdict == "unproven":
                    n_unproven += 1
                else:
                    assert d_true[0] > 5.0
    assert n_proven > 1500 and n_unproven > 300 and n_bounded_proven > 700, (n_proven, n_unproven, n_bounded_proven)


def test_truncated_keys_never_hide_a_nearer_point():
    """Many candidates whose squared distances share one 2^-13 bucket (a ring of points at almost the same distance): the six
    kept by truncated key are not the six nearest -- the proof must refuse unless the fifth is clear of the sixth KEY."""
    rng = np.random.default_rng(9)
    refused = accepted = 0
    for trial in range(300):
        q = rng.uniform(-1, 1, 3).astype(F)
        near = q + rng.normal(0, 1, (4, 3)) * 0.05
        dirs = rng.normal(0, 1, (12, 3))
        dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        ring = q + dirs * (0.30 + rng.uniform(0, 2e-6 if trial % 2 else 2e-3, 12))[:, None]
        pts = np.concatenate([near, ring, q + rng.uniform(2, 4, (30, 3)) * rng.choice([-1, 1], (30, 3))]).astype(F)
        G = Grid(pts)
        verdict, idx5, d5 = G.probe(q)
        d_all = dist2(q, pts)
        order = np.argsort(d_all, kind="stable")
        if verdict == "proven":
            accepted += 1
            assert np.array_equal(idx5, order[:5])
            assert d_all[order[5]] * (1.0 - 1.1e-5) > d5[4]
        else:
            refused += 1
    assert refused > 100 and accepted > 20, (refused, accepted)


def test_far_means_beyond_the_gate():
    rng = np.random.default_rng(2)
    pts = voxel_like_map(rng)
    G = Grid(pts)
    lo, hi = pts.min(0), pts.max(0)
    n_far = 0
    for _ in range(3000):
        q = (rng.uniform(lo - 6, hi + 6)).astype(F)
        verdict, _, _ = G.probe(q)
        if verdict == "far":
            n_far += 1
            assert dist2(q, pts).min() > 5.0 * (1 + 1e-5)
    assert n_far > 200
