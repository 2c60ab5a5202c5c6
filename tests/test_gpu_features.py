"""GPU parity for the feature-extraction front end: lslam_extract_features against
oracle/features_oracle.c on the same ring-sorted scans -- bit for bit, including the taps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _compare(got, ref, tag):
    for k in ("curvature",):
        assert np.array_equal(bits(got[k]), bits(ref[k])), (tag, k)
    assert np.array_equal(got["picked"], ref["picked"]), tag
    bad = np.nonzero(got["label"] != ref["label"])[0]
    assert len(bad) == 0, (tag, bad[:10], got["label"][bad[:10]], ref["label"][bad[:10]])
    for k in ("sharp", "less_sharp", "flat", "less_flat"):
        assert got[k].shape == ref[k].shape, (tag, k, got[k].shape, ref[k].shape)
        assert np.array_equal(bits(got[k]), bits(ref[k])), (tag, k)


@pytest.mark.parametrize("rings,steps,seed", [(16, 900, 1), (16, 1800, 2), (64, 1800, 3), (32, 2400, 4)])
def test_extract_features_matches_oracle(pkg, ctx, oracle, synth, rings, steps, seed):
    world = synth.World(half_extent=120.0, wall_half=90.0)
    gt = (0.01, -0.02, 0.3 * seed, 3.0 - seed, -2.0 + 2 * seed, synth.SENSOR_HEIGHT)
    c, s, gt, cloud, ranges = synth.make_scan(world, rings, steps, gt_pose=gt, seed=100 + seed, full=True)
    got = pkg.scan_registration.extract_features(ctx, cloud, ranges, taps=True)
    ref = oracle.extract_features(cloud, ranges)
    assert len(ref["sharp"]) > 10 and len(ref["flat"]) > 50 and len(ref["less_flat"]) > 1000
    _compare(got, ref, (rings, steps))


def test_extract_features_parameters_and_edges(pkg, ctx, oracle, synth):
    world = synth.World(half_extent=60.0, wall_half=55.0)
    c, s, gt, cloud, ranges = synth.make_scan(world, 16, 900, full=True)
    # other parameters (the launch files set some of these)
    p, q = pkg.scan_registration.default_params(ctx), oracle.reg_params()
    for obj in (p, q):
        obj.n_feature_regions = 4
        obj.curvature_region = 3
        obj.max_corner_sharp = 4
        obj.max_surface_flat = 2
        obj.surface_curvature_threshold = 0.1
        obj.less_flat_filter_size = 0.4
    _compare(pkg.scan_registration.extract_features(ctx, cloud, ranges, p, taps=True),
             oracle.extract_features(cloud, ranges, q), "params")
    # rings that are empty or too short are skipped (ScanRegistration.cpp:205-207)
    n0 = int(ranges[3, 1]) + 1
    short = np.concatenate([ranges[:4], [[n0, n0 - 1], [n0, n0 + 9], [n0 + 10, int(ranges[4, 1])]], ranges[5:]]).astype(np.int32)
    _compare(pkg.scan_registration.extract_features(ctx, cloud, short, taps=True),
             oracle.extract_features(cloud, short), "short")
    # PointXYZINormal-like stride: the copied field is at another offset
    wide = np.zeros((len(cloud), 12), np.float32)
    wide[:, :3] = cloud[:, :3]
    wide[:, 8] = cloud[:, 3]
    _compare(pkg.scan_registration.extract_features(ctx, wide, ranges, intensity_field=8, taps=True),
             oracle.extract_features(cloud, ranges), "wide")
    with pytest.raises(pkg.LslamError):
        pkg.scan_registration.extract_features(ctx, cloud, np.array([[0, len(cloud)]], np.int32))
    with pytest.raises(pkg.LslamError):
        pkg.scan_registration.extract_features(ctx, cloud, np.array([[0, 4000]], np.int32))  # > 2560 points


@pytest.mark.parametrize("nf,cr,leaf", [(1, 5, 0.2), (2, 5, 0.2), (3, 4, 0.2), (7, 5, 0.3), (13, 2, 0.2), (40, 5, 0.2), (200, 1, 0.2),
                                         (6, 5, 1.0e-4), (6, 5, 25.0)])
def test_extract_features_region_counts_and_filter_sizes(pkg, ctx, oracle, synth, nf, cr, leaf):
    """Round 5: the regions of a ring are sorted side by side in power-of-two stretches of one LDS array (1 region of ~1 800
    points ... 200 regions of 9), and the per-ring VoxelGrid runs in LDS: leaves so small that PCL's guard hands the ring back
    unfiltered (index volume above INT_MAX), and so large that a ring is one voxel."""
    world = synth.World(half_extent=80.0, wall_half=70.0)
    c, s, gt, cloud, ranges = synth.make_scan(world, 16, 1800, seed=7 + nf, full=True)
    p, q = pkg.scan_registration.default_params(ctx), oracle.reg_params()
    for obj in (p, q):
        obj.n_feature_regions = nf
        obj.curvature_region = cr
        obj.less_flat_filter_size = leaf
    _compare(pkg.scan_registration.extract_features(ctx, cloud, ranges, p, taps=True),
             oracle.extract_features(cloud, ranges, q), (nf, cr, leaf))


def test_extract_features_more_rings_than_the_device_has_cus_for(pkg, ctx, oracle, synth):
    """Round 5: pointClassify runs on helper workgroups (three per ring, dispatched ahead of the rings' own workgroups, which
    pick the verdicts up or classify for themselves).  192 rings x 4 workgroups of 1 024 threads are three times what the
    device holds at once: rings whose helpers are long done, rings whose helpers run beside them, the same lists."""
    world = synth.World(half_extent=120.0, wall_half=90.0)
    c, s, gt, cloud, ranges = synth.make_scan(world, 64, 1800, seed=21, full=True)
    thirds = []
    for a, b in ranges:
        n = int(b) - int(a) + 1
        cuts = [int(a), int(a) + n // 3, int(a) + 2 * n // 3, int(b) + 1]
        thirds += [[cuts[k], cuts[k + 1] - 1] for k in range(3)]
    thirds = np.array(thirds, np.int32)
    assert len(thirds) == 192
    _compare(pkg.scan_registration.extract_features(ctx, cloud, thirds, taps=True), oracle.extract_features(cloud, thirds), "192 rings")


def test_extract_features_many_short_rings(pkg, ctx, oracle, synth):
    """1 200 rings of 24 points (a ring of at most 2 * curvatureRegion + 1 points is skipped, :205-207; these are just above
    it with curvatureRegion = 5): 4 800 workgroups in one launch, the per-list scans over more rings than a workgroup has
    wavefronts for."""
    world = synth.World(half_extent=120.0, wall_half=90.0)
    c, s, gt, cloud, ranges = synth.make_scan(world, 16, 1800, seed=5, full=True)
    n = (len(cloud) // 24) * 24
    short = np.stack([np.arange(0, n, 24), np.arange(0, n, 24) + 23], axis=1).astype(np.int32)[:1200]
    p, q = pkg.scan_registration.default_params(ctx), oracle.reg_params()
    for obj in (p, q):
        obj.n_feature_regions = 2
    _compare(pkg.scan_registration.extract_features(ctx, cloud, short, p, taps=True), oracle.extract_features(cloud, short, q), "short rings")


def test_extract_features_bounds_its_region_count(pkg, ctx, synth):
    world = synth.World(half_extent=60.0, wall_half=55.0)
    c, s, gt, cloud, ranges = synth.make_scan(world, 16, 900, full=True)
    p = pkg.scan_registration.default_params(ctx)
    p.n_feature_regions = 513
    with pytest.raises(pkg.LslamError):
        pkg.scan_registration.extract_features(ctx, cloud, ranges, p)


def test_features_feed_odometry(pkg, ctx, oracle, synth, small_problem):
    """Config 1 end to end on the device: extractFeatures on two consecutive sweeps, then
    LaserOdometry::scanMatch (variant B) of sweep 2's sharp/flat points against sweep 1's
    less-sharp/less-flat clouds -- the same clouds and the same result as the oracle chain."""
    world = small_problem["world"]
    clouds = []
    for k in range(2):
        gt = np.array([0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.25 * k, -2.0 + 0.1 * k, synth.SENSOR_HEIGHT])
        c, s, gt, cloud, ranges = synth.make_scan(world, 16, 900, gt_pose=gt, seed=700 + k, full=True)
        g = pkg.scan_registration.extract_features(ctx, cloud, ranges)
        r = oracle.extract_features(cloud, ranges)
        for key in ("sharp", "less_sharp", "flat", "less_flat"):
            assert np.array_equal(bits(g[key]), bits(r[key]))
        clouds.append(g)
    pose0 = np.zeros(6, np.float32)
    _, pose_g, st_g = ctx.odometry_match(clouds[0]["less_sharp"], clouds[0]["less_flat"], clouds[1]["sharp"],
                                            clouds[1]["flat"], pose0)
    it_o, pose_o, st_o = oracle.odometry_match(clouds[0]["less_sharp"], clouds[0]["less_flat"], clouds[1]["sharp"],
                                               clouds[1]["flat"], pose0)
    assert st_g.iterations == it_o and it_o > 0
    assert np.abs(pose_g[3:] - pose_o[3:]).max() <= 1e-4 and np.abs(pose_g[:3] - pose_o[:3]).max() <= 1e-5


@pytest.mark.parametrize("rings,lo,hi", [(16, -15.0, 15.0), (64, -24.9, 2.0)])
def test_multiscan_register_matches_oracle(pkg, ctx, oracle, synth, rings, lo, hi):
    """MultiScanRegistration::process on the device: same points in the same rings in the same order;
    ring + relTime within 2e-6 (atan / atan2 are the device's), and the chain into extractFeatures."""
    from test_oracle_features import _raw_sweep
    raw, ring = _raw_sweep(synth, rings=rings, steps=900, seed=8)
    got, granges = pkg.scan_registration.multiscan_register(ctx, raw, lo, hi, rings)
    ref, oranges = oracle.multiscan_register(raw, lo, hi, rings)
    assert got.shape == ref.shape and np.array_equal(granges, oranges)
    assert np.array_equal(bits(got[:, :3]), bits(ref[:, :3]))
    assert np.array_equal(np.floor(got[:, 3]), np.floor(ref[:, 3]))
    assert np.abs(got[:, 3] - ref[:, 3]).max() <= 2e-6 * rings
    feat = pkg.scan_registration.extract_features(ctx, got, granges)
    ofeat = oracle.extract_features(ref, oranges)
    for k in ("sharp", "less_sharp", "flat"):
        assert feat[k].shape == ofeat[k].shape and np.array_equal(bits(feat[k][:, :3]), bits(ofeat[k][:, :3])), k
