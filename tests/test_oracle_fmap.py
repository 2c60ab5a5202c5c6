"""CPU checks of the map-maintenance oracle (oracle/fmap_oracle.c) against independent numpy /
pure-Python restatements of the same reference code (util/FeatureMap.h, pcl::VoxelGrid)."""
import numpy as np
import pytest

from oracle_lib import Oracle


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def np_voxel_grid(cloud, leaf):
    """pcl::VoxelGrid::applyFilter in numpy: idx = floor(p*inv) - min_b, linear index, stable
    grouping, fp32 sums in input order / count."""
    c = np.asarray(cloud, np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    cell = np.floor(c[:, :3] * inv).astype(np.int64)
    mn = np.floor(c[:, :3].min(0) * inv).astype(np.int64)
    mx = np.floor(c[:, :3].max(0) * inv).astype(np.int64)
    div = mx - mn + 1
    ijk = cell - mn
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    out = []
    a = 0
    while a < len(order):
        b = a
        s = np.zeros(4, np.float32)
        while b < len(order) and idx[order[b]] == idx[order[a]]:
            s = (s + c[order[b]]).astype(np.float32)
            b += 1
        out.append(s / np.float32(b - a))
        a = b
    return np.array(out, np.float32)


@pytest.mark.parametrize("leaf,n,extent", [(0.2, 3000, 6.0), (1.0, 5000, 80.0), (0.4, 1, 1.0), (0.5, 400, 0.3)])
def test_voxel_grid_matches_numpy(oracle, leaf, n, extent):
    rng = np.random.default_rng(7)
    c = (rng.uniform(-extent, extent, (n, 4))).astype(np.float32)
    c[:, 3] = rng.uniform(0, 64, n).astype(np.float32)
    got = oracle.voxel_grid(c, leaf)
    ref = np_voxel_grid(c, leaf)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # one point per voxel, every centroid inside (or on the rim of) its voxel
    cells = np.floor(got[:, :3] / np.float32(leaf)).astype(np.int64)
    assert len(np.unique(cells, axis=0)) >= len(got) - 2  # a centroid may round onto a face
    # a second pass is (nearly always) the identity: that is what lets untouched cubes be skipped in principle
    again = oracle.voxel_grid(got, leaf)
    assert len(again) <= len(got) and len(again) >= len(got) - 2


def test_voxel_grid_overflow_returns_input(oracle):
    c = np.array([[0, 0, 0, 1], [5000, 5000, 5000, 2], [1, 1, 1, 3]], np.float32)
    out = oracle.voxel_grid(c, 0.01)  # 5e5^3 voxels > INT_MAX: PCL warns and returns the input
    assert np.array_equal(out, c)


class PyFeatureMap:
    """Pure-Python restatement of the cube bookkeeping (FeatureMap.h:232-254,307-377,475-487)."""

    def __init__(self, w, h, d, size, dist):
        self.W, self.H, self.D = w, h, d
        self.origin = [int(round((w - 1) / 2.0)), int(round((h - 1) / 2.0)), int(round((d - 1) / 2.0))]
        self.size, self.dist = np.float32(size), np.float32(dist)
        self.cubes = [[] for _ in range(w * h * d)]  # lists of point tags
        self.valid = []

    def ok(self, i, j, k):
        return 0 <= i < self.W and 0 <= j < self.H and 0 <= k < self.D

    def idx(self, i, j, k):
        return i + j * self.W + k * self.W * self.H

    def cube(self, p):
        return [int(np.float32(np.round(np.float32(p[d]) / self.size)) + np.float32(self.origin[d])) for d in range(3)]

    def push(self, p, tag):
        g = self.cube(p)
        if self.ok(*g):
            self.cubes[self.idx(*g)].append(tag)

    def update(self, pos):
        g = self.cube(pos)
        lim = (self.W, self.H, self.D)
        ng = [min(max(g[d], 3), lim[d] - 4) for d in range(3)]
        dl = [ng[d] - g[d] for d in range(3)]
        if any(dl):
            for i in range(self.W):
                for j in range(self.H):
                    for k in range(self.D):
                        o = (i - dl[0], j - dl[1], k - dl[2])
                        a = self.idx(i, j, k)
                        if self.ok(*o):
                            b = self.idx(*o)
                            self.cubes[a], self.cubes[b] = self.cubes[b], self.cubes[a]
                        else:
                            self.cubes[a] = []
        for d in range(3):
            self.origin[d] += dl[d]
        self.cur = ng


def test_feature_map_bookkeeping_matches_python(oracle):
    """Pushes tagged points, walks the sensor so that update() shifts the grid in both
    directions, and compares per-cube membership with the Python restatement (huge leaf sizes
    would merge points, so the leaves are tiny: VoxelGrid keeps every point)."""
    W, H, D, size, dist = 9, 8, 7, 10.0, 14.0
    fm = oracle.feature_map(W, H, D)
    fm.setup_world_cube_size(size)
    fm.setup_lidar_valid_distance(dist)
    fm.setup_filter_size(1e-3, 1e-3, 1e-3)
    py = PyFeatureMap(W, H, D, size, dist)
    rng = np.random.default_rng(3)
    tag = 0
    T = np.eye(4, dtype=np.float32)
    for step, pos in enumerate([(0, 0, 0), (12, 3, 1), (38, -3, 2), (47, 20, 12), (20, 44, 31), (-30, -50, -2), (-44, 10, 0)]):
        pos = np.array(pos, np.float32)
        fm.update(pos)
        py.update(pos)
        assert list(fm.info()["origin"]) == py.origin, step
        pts = (pos + rng.uniform(-25, 25, (200, 3))).astype(np.float32)
        cloud = np.concatenate([pts, np.arange(tag, tag + 200, dtype=np.float32)[:, None]], 1)
        fm.add_feature_cloud(cloud[:120], cloud[120:], T)
        for p, t in zip(cloud[:120], range(tag, tag + 120)):
            py.push(p, t)
        tag += 200
        # corner membership per cube through the surround of a distance that covers everything
        valid = fm.info()["valid"]
        assert len(valid) > 0
        corner, surf = fm.get_surround_feature()
        want = sorted(t for c in valid for t in py.cubes[c])
        assert sorted(int(v) for v in corner[:, 3]) == want, step
    assert tag == 1400
