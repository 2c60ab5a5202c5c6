"""CPU checks of the feature-extraction oracle (oracle/features_oracle.c) on synthetic ring-sorted
scans: numpy cross-checks of the curvature and of the structural rules of
ScanRegistration::extractFeatures (per-region caps, subset-of-input, stable order)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def scan(synth):
    world = synth.World(half_extent=60.0, wall_half=55.0)
    c, s, gt, cloud, ranges = synth.make_scan(world, 16, 900, full=True)
    return cloud, ranges


def test_curvature_and_marks(oracle, scan):
    cloud, ranges = scan
    out = oracle.extract_features(cloud, ranges)
    curv = out["curvature"]
    xyz = cloud[:, :3]
    rng = np.random.default_rng(0)
    for s, (a, b) in enumerate(ranges):
        if b <= a + 10:
            continue
        for i in rng.integers(a + 5, b - 5, 20):
            d = np.float32(-10.0) * xyz[i]
            for q in range(1, 6):
                d = (d + (xyz[i + q] + xyz[i - q])).astype(np.float32)
            ref = np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])) + np.float32(d[2] * d[2])
            # points outside every region keep 0
            assert curv[i] == 0.0 or curv[i] == np.float32(ref), (s, i)
    # setScanBuffersFor leaves only these marks
    assert set(np.unique(out["picked"])) <= {0, -2, -3, -4}
    assert (out["picked"] == -3).any()  # occlusion boundaries exist in the Manhattan world


def test_outputs_obey_the_extraction_rules(oracle, scan):
    cloud, ranges = scan
    p = oracle.reg_params()
    out = oracle.extract_features(cloud, ranges, p)
    key = {tuple(v) for v in cloud[:, :4].view(np.uint32).tolist()}
    for name in ("sharp", "less_sharp", "flat"):
        pts = out[name]
        assert len(pts) > 0, name
        assert all(tuple(v) in key for v in pts.view(np.uint32).tolist()), name  # copies of input points
    # sharp is a subset of less_sharp; every region contributes at most 4 flat points from loop 1 and
    # at most 4 more one-side-flat ones from loop 3
    ls = {tuple(v) for v in out["less_sharp"].view(np.uint32).tolist()}
    assert all(tuple(v) in ls for v in out["sharp"].view(np.uint32).tolist())
    n_regions = sum(1 for a, b in ranges if b > a + 10) * p.n_feature_regions
    assert len(out["flat"]) <= 2 * p.max_surface_flat * n_regions
    # less_flat is a per-ring VoxelGrid: not more points than flat-curvature points, spacing >= a voxel
    assert 0 < len(out["less_flat"]) < len(cloud)
    # labels: flat picks have low curvature, sharp ones high
    lab, curv = out["label"], out["curvature"]
    assert (curv[lab == 1][curv[lab == 1] > 0] >= 0).all()
    assert ((lab == -1) | (lab == 0) | (lab == 1) | (lab == 5) | (lab == 6)).all()


def test_point_classify_on_constructed_geometry(oracle):
    """A straight run of 11 points is SURFACE_FLAT (both half-lines parallel); a right-angle corner
    is CORNER_SHARP; a line meeting scatter is ONESIDE_FLAT; scatter is MESSY."""
    t = np.arange(-5, 6, dtype=np.float32) * np.float32(0.05)
    line = np.stack([10 + 0 * t, t, 0 * t, 0 * t], 1).astype(np.float32)
    assert oracle.point_classify(line, 5) == -1
    corner = line.copy()
    corner[6:, 0] = 10 + t[6:]
    corner[6:, 1] = 0
    assert oracle.point_classify(corner, 5) == 1
    rng = np.random.default_rng(1)
    half = line.copy()
    half[6:, :3] += rng.normal(0, 0.5, (5, 3)).astype(np.float32)
    assert oracle.point_classify(half, 5) == 5
    mess = line.copy()
    mess[:, :3] += rng.normal(0, 0.5, (11, 3)).astype(np.float32)
    assert oracle.point_classify(mess, 5) == 9


def _raw_sweep(synth, rings=16, steps=900, seed=3):
    """A raw driver cloud: z-up sensor frame, points in arrival order (azimuth sweeping, all rings per
    azimuth step) -- what MultiScanRegistration::process receives."""
    world = synth.World(half_extent=60.0, wall_half=55.0)
    c, s, gt, cloud, ranges = synth.make_scan(world, rings, steps, seed=seed, full=True)
    ring = np.floor(cloud[:, 3]).astype(np.int64)
    rel = cloud[:, 3] - ring
    order = np.lexsort((ring, -rel))  # a Velodyne turns clockwise: azimuth falls along the sweep; then by ring
    return cloud[order], ring[order]


def test_multiscan_register_rings_and_time(oracle, synth):
    raw, ring = _raw_sweep(synth)
    out, ranges = oracle.multiscan_register(raw, -15.0, 15.0, 16)
    assert len(out) == len(raw)  # nothing dropped: all returns are finite, inside the ring table
    for r in range(16):
        a, b = ranges[r]
        seg = out[a:b + 1]
        assert len(seg) == (ring == r).sum()
        assert np.all(np.floor(seg[:, 3]) == r)
        # axes swapped (x', y', z') = (y, z, x); arrival order kept; relTime grows along the sweep
        src = raw[ring == r]
        assert np.array_equal(seg[:, 0], src[:, 1]) and np.array_equal(seg[:, 1], src[:, 2]) and np.array_equal(seg[:, 2], src[:, 0])
        rel = seg[:, 3] - r
        assert rel.min() >= -1e-4 and rel.max() <= 0.1 + 1e-4 and np.all(np.diff(rel) > -1e-4)
    # invalid points are dropped: NaN, near-zero, outside the ring table
    bad = raw[:5].copy()
    bad[0, 0] = np.nan
    bad[1, :3] = 1e-4
    bad[2, :3] = (1.0, 0.0, 5.0)   # 78 degrees up
    out2, ranges2 = oracle.multiscan_register(np.concatenate([raw[:1], bad, raw[1:]]), -15.0, 15.0, 16)
    assert len(out2) == len(raw) + 2
