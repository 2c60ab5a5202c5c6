"""The odometry node resident on the device (include/lslam_c.h lslam_fset_* / lslam_odom_*; csrc/lslam_odom.hip):
LaserOdometry::process (odometry/LaserOdometry.cpp:288-326) with the sweep's feature clouds, the last clouds and their
search grids in HBM.  Held against the host-pointer entry points (lslam_odometry_match + lslam_transform_to_end), against
the kd-tree implementation of the same match (lslam_odometry_match_trees: nanoflann's traversal) and against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _raw(synth, world, k, rings=16, steps=900):
    gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
    _, _, _, cloud, _ = synth.make_scan(world, rings, steps, gt_pose=gt, seed=300 + k, full=True)
    ring = np.floor(cloud[:, 3]).astype(np.int64)
    return cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))]


def test_feature_set_holds_the_lists_bit_for_bit(pkg, ctx, synth, small_problem):
    sr = pkg.scan_registration
    reg, rr = sr.multiscan_register(ctx, _raw(synth, small_problem["world"], 1), -15.0, 15.0, 16)
    f = sr.extract_features(ctx, reg, rr)
    fs = sr.FeatureSet(ctx)
    counts = sr.extract_features_dev(ctx, reg, rr, fs)
    assert counts == {k: len(f[k]) for k in sr.LISTS} == fs.counts()
    for k in sr.LISTS:
        assert np.array_equal(bits(fs.download(k)), bits(f[k])), k
    # host lists up, the same lists down; a smaller sweep into the same set afterwards
    fs2 = sr.FeatureSet(ctx).upload(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
    for k in sr.LISTS:
        assert np.array_equal(bits(fs2.download(k)), bits(f[k])), k
    fs2.upload(f["sharp"][:7], f["less_sharp"][:0], f["flat"][:3], f["less_flat"][:11])
    assert fs2.counts() == dict(sharp=7, less_sharp=0, flat=3, less_flat=11)
    assert np.array_equal(bits(fs2.download("less_flat")), bits(f["less_flat"][:11]))
    fs.close()
    fs2.close()


@pytest.mark.parametrize("rings,steps,lo,hi", [(16, 900, -15.0, 15.0), (64, 600, -24.9, 2.0)])
def test_device_node_equals_the_host_pointer_chain(pkg, ctx, synth, small_problem, rings, steps, lo, hi):
    """Seven sweeps that go out and come back (so that some loops run to the iteration limit and some end at once): the node in
    HBM and LaserOdometry over host pointers give the same _transform, _Tsum and last clouds, bit for bit, every sweep."""
    sr = pkg.scan_registration
    world = small_problem["world"]
    host = pkg.LaserOdometry(ctx)
    dev = pkg.DeviceLaserOdometry(ctx)
    fs = [sr.FeatureSet(ctx), sr.FeatureSet(ctx)]
    iters = []
    for step, k in enumerate((0, 1, 2, 3, 2, 1, 0)):
        reg, rr = sr.multiscan_register(ctx, _raw(synth, world, k, rings, steps), lo, hi, rings)
        f = sr.extract_features(ctx, reg, rr)
        sr.extract_features_dev(ctx, reg, rr, fs[step & 1])
        T_h = host.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
        T_d = dev.process(fs[step & 1])
        if step == 0:
            assert T_h is None and T_d is None
            assert dev.last_ostats.matched == 0
        else:
            assert dev.last_ostats.matched == 1 and dev.last_ostats.tree_fallbacks == 0
            assert np.abs(T_d - T_h).max() <= 1e-6, step  # (numpy's 4x4 product rounds differently from the library's row-times-column loop)
            assert np.array_equal(bits(dev.transform), bits(host.transform)), step
            assert dev.last_stats.iterations == host.last_stats.iterations and dev.last_stats.n_rows == host.last_stats.n_rows
            assert dev.last_ostats.searches == (dev.last_stats.sweeps + 4) // 5
            iters.append(dev.last_stats.iterations)
        assert np.array_equal(bits(dev.last_corner), bits(host.last_corner)), step
        assert np.array_equal(bits(dev.last_surf), bits(host.last_surf)), step
        lc, ls = dev.last_clouds()
        assert np.array_equal(bits(lc), bits(host.last_corner)) and np.array_equal(bits(ls), bits(host.last_surf))
    assert max(iters) == 25 and min(iters) < 25
    dev.close()
    for s in fs:
        s.close()


def test_grid_search_equals_the_kd_tree_search(ctx, oracle, synth, small_problem):
    """The same match with nearestKSearch(., 1) through hashed cell grids and through kd-trees (nanoflann's traversal replayed):
    same iterations, same rows, same pose bits -- on the consecutive-sweep pairs of the parity test, from a far-off initial guess
    (many queries without a neighbour inside the 5 m gate) and with clouds pushed 3 km out (coarse float spacing)."""
    from test_gpu_parity import _odometry_pair
    cases = []
    for k in range(3):
        lc, ls, sharp, flat = _odometry_pair(synth, small_problem["world"], k)
        cases.append((lc, ls, sharp, flat, np.zeros(6, np.float32)))
    lc, ls, sharp, flat = _odometry_pair(synth, small_problem["world"], 0)
    cases.append((lc, ls, sharp, flat, np.array([0.02, -0.03, 0.25, 4.0, -3.0, 0.5], np.float32)))
    far = np.array([3000.0, -2500.0, 0.0, 0.0], np.float32)
    cases.append((lc + far, ls + far, sharp + far, flat + far, np.zeros(6, np.float32)))
    # last clouds that are NOT in ring order (the ring windows are then walked point by point, as the reference does) and ring
    # ids beyond a byte
    rng = np.random.default_rng(5)
    cases.append((lc[rng.permutation(len(lc))], ls[rng.permutation(len(ls))], sharp, flat, np.zeros(6, np.float32)))
    up = np.array([0.0, 0.0, 0.0, 300.0], np.float32)
    cases.append((lc + up, ls + up, sharp + up, flat + up, np.zeros(6, np.float32)))
    for i, (a, b, c, d, p0) in enumerate(cases):
        s_g, pose_g, st_g = ctx.odometry_match(a, b, c, d, p0)
        s_t, pose_t, st_t = ctx.odometry_match(a, b, c, d, p0, trees=True)
        assert (s_g, st_g.iterations, st_g.sweeps, st_g.n_rows, st_g.n_line, st_g.n_plane) == \
               (s_t, st_t.iterations, st_t.sweeps, st_t.n_rows, st_t.n_line, st_t.n_plane), i
        assert np.array_equal(bits(pose_g), bits(pose_t)), i
    ctx.map_set(small_problem["map_corner"], small_problem["map_surf"])


def test_ring_windows_through_the_grids_equal_the_walk(pkg, ctx, synth, small_problem, monkeypatch):
    """The second and third points of a correspondence (:366-403 / :430-477) found as nearest neighbours of a ring category
    through the cell grids, against the same node walking the windows point by point (LSLAM_ODOM_LITERAL_WINDOW=1, read when the
    node is made): same correspondences, hence the same poses bit for bit, over sweeps with long and short loops."""
    sr = pkg.scan_registration
    world = small_problem["world"]
    grid = pkg.DeviceLaserOdometry(ctx)
    monkeypatch.setenv("LSLAM_ODOM_LITERAL_WINDOW", "1")
    walk = pkg.DeviceLaserOdometry(ctx)
    monkeypatch.delenv("LSLAM_ODOM_LITERAL_WINDOW")
    fs = sr.FeatureSet(ctx)
    for step, k in enumerate((0, 2, 4, 3, 3, 1)):
        reg, rr = sr.multiscan_register(ctx, _raw(synth, world, k, 32, 900), -30.67, 10.67, 32)
        sr.extract_features_dev(ctx, reg, rr, fs)
        T_g, T_w = grid.process(fs), walk.process(fs)
        if step:
            assert np.array_equal(bits(grid.transform), bits(walk.transform)), step
            assert (grid.last_stats.iterations, grid.last_stats.n_rows) == (walk.last_stats.iterations, walk.last_stats.n_rows)
            assert np.array_equal(bits(T_g), bits(T_w))
        assert np.array_equal(bits(grid.last_surf), bits(walk.last_surf))
    grid.close()
    walk.close()
    fs.close()


def test_persistent_iterations_equal_the_launch_loop(pkg, ctx, synth, small_problem, monkeypatch):
    """Up to five iterations per launch (odom_gn_kernel: resident workgroups, sentinel-slot exchange, replicated solve) against the
    same node with one launch per step (LSLAM_ODOM_PERSISTENT=0), and against a node whose first exchange gives up
    (LSLAM_ODOM_SPIN_LIMIT=0: the fallback a grid that is not co-resident takes): the same bits from all three."""
    sr = pkg.scan_registration
    world = small_problem["world"]
    fused = pkg.DeviceLaserOdometry(ctx)
    monkeypatch.setenv("LSLAM_ODOM_PERSISTENT", "0")
    loop = pkg.DeviceLaserOdometry(ctx)
    monkeypatch.delenv("LSLAM_ODOM_PERSISTENT")
    monkeypatch.setenv("LSLAM_ODOM_SPIN_LIMIT", "0")
    gives_up = pkg.DeviceLaserOdometry(ctx)
    monkeypatch.delenv("LSLAM_ODOM_SPIN_LIMIT")
    fs = sr.FeatureSet(ctx)
    iters = set()
    for step, k in enumerate((0, 1, 2, 3, 3, 2, 0)):
        reg, rr = sr.multiscan_register(ctx, _raw(synth, world, k, 16, 900), -15.0, 15.0, 16)
        sr.extract_features_dev(ctx, reg, rr, fs)
        Ts = [n.process(fs) for n in (fused, loop, gives_up)]
        if step:
            iters.add(fused.last_stats.iterations)
            for n in (loop, gives_up):
                assert np.array_equal(bits(fused.transform), bits(n.transform)), step
                assert (fused.last_stats.iterations, fused.last_stats.n_rows, fused.last_stats.sweeps, fused.last_stats.converged) == \
                       (n.last_stats.iterations, n.last_stats.n_rows, n.last_stats.sweeps, n.last_stats.converged)
            assert np.array_equal(bits(Ts[0]), bits(Ts[1])) and np.array_equal(bits(Ts[0]), bits(Ts[2]))
        assert np.array_equal(bits(fused.last_surf), bits(loop.last_surf)) and np.array_equal(bits(fused.last_surf), bits(gives_up.last_surf))
    assert len(iters) >= 2 and max(iters) == 25
    for n in (fused, loop, gives_up):
        n.close()
    fs.close()


def test_an_exact_tie_goes_through_the_trees(pkg, ctx, synth, small_problem):
    """Two different points of a last cloud at exactly the same distance from a query: which one nearestKSearch returns is decided
    by nanoflann's visit order, which only the kd-tree search replays -- the node notices, redoes the match through the trees
    and says so."""
    from test_gpu_parity import _odometry_pair
    sr = pkg.scan_registration
    lc, ls, sharp, flat = _odometry_pair(synth, small_problem["world"], 0)
    ls = ls.copy()
    flat = flat.copy()
    # a mirror pair around the first flat query (zero initial transform: the de-skewed query is the query itself)
    flat[0, :3] = np.round(flat[0, :3] * 64.0) / 64.0  # (so that q +- 1/64 are exact)
    q = flat[0, :3].copy()
    ring = np.floor(ls[200, 3])
    ls[200, :3] = q + np.array([1.0 / 64, 0.0, 0.0], np.float32)
    ls[201, :3] = q - np.array([1.0 / 64, 0.0, 0.0], np.float32)
    ls[201, 3] = ls[200, 3]
    d0 = ((ls[200, 0] - q[0]) ** 2 + (ls[200, 1] - q[1]) ** 2) + (ls[200, 2] - q[2]) ** 2
    d1 = ((ls[201, 0] - q[0]) ** 2 + (ls[201, 1] - q[1]) ** 2) + (ls[201, 2] - q[2]) ** 2
    assert d0 == d1 and ring == np.floor(ls[201, 3])
    others = np.delete(ls[:, :3], (200, 201), axis=0)
    assert ((others - q) ** 2).sum(1).min() > d0  # the pair is nearest
    dev = pkg.DeviceLaserOdometry(ctx)
    fs = sr.FeatureSet(ctx)
    fs.upload(sharp[:0], lc, flat[:0], ls)       # first sweep: these become the last clouds as they are
    dev.process(fs)
    fs.upload(sharp, lc, flat, ls)
    dev.process(fs)
    assert dev.last_ostats.matched == 1 and dev.last_ostats.tree_fallbacks == 1
    s_t, pose_t, st_t = ctx.odometry_match(lc, ls, sharp, flat, np.zeros(6, np.float32), trees=True)
    assert np.array_equal(bits(dev.transform), bits(pose_t)) and dev.last_stats.iterations == st_t.iterations
    # ... and the clouds moved on with that pose
    assert np.array_equal(bits(dev.last_clouds()[1]), bits(ctx.transform_to_end(ls, pose_t)))
    # the host-pointer entry takes the same way round
    s_g, pose_g, st_g = ctx.odometry_match(lc, ls, sharp, flat, np.zeros(6, np.float32))
    assert np.array_equal(bits(pose_g), bits(pose_t))
    dev.close()
    fs.close()
    ctx.map_set(small_problem["map_corner"], small_problem["map_surf"])


def test_small_last_clouds_are_not_matched_against(pkg, ctx, synth, small_problem):
    """LaserOdometry.cpp:337: fewer than 11 corner or 101 surface points in the last clouds -- no scan match, _transform stays,
    _Tsum still advances by it and the clouds still move to the sweep end with it."""
    from test_gpu_parity import _odometry_pair
    sr = pkg.scan_registration
    lc, ls, sharp, flat = _odometry_pair(synth, small_problem["world"], 0)
    dev = pkg.DeviceLaserOdometry(ctx)
    fs = sr.FeatureSet(ctx)
    fs.upload(sharp, lc, flat, ls)
    assert dev.process(fs) is None
    fs.upload(sharp, lc[:10], flat, ls)           # matched; leaves a ten-point corner cloud behind
    T1 = dev.process(fs)
    assert dev.last_ostats.matched == 1 and dev.last_ostats.n_last_corner == 10
    tr1 = dev.transform.copy()
    fs.upload(sharp, lc, flat, ls)
    T2 = dev.process(fs)                           # nothing to match against
    assert dev.last_ostats.matched == 0 and dev.last_stats.status == 1
    assert np.array_equal(bits(dev.transform), bits(tr1))
    assert np.allclose(T2, (T1 @ ctx.pose_to_isometry(tr1)).astype(np.float32), atol=1e-6)
    assert np.array_equal(bits(dev.last_corner), bits(ctx.transform_to_end(lc, tr1)))
    T3 = dev.process(fs)                           # and matched again
    assert dev.last_ostats.matched == 1
    dev.close()
    fs.close()


def test_fuzz_grid_search_against_the_tree_search(ctx, small_problem):
    """Random scan pairs the structured scenes do not produce -- clustered and scattered last clouds, queries with no neighbour
    inside the gate, last clouds barely above the guard's sizes, lattice clouds (exact distance ties: the tree path is taken and
    the answer is the tree's by construction), huge rings -- through the hashed-grid search and through kd-trees: the same
    iterations, rows and pose bits every time."""
    rng = np.random.default_rng(20260601)
    n_cases = 0
    for trial in range(36):
        kind = trial % 6
        n_lc, n_ls = int(rng.integers(11, 400)), int(rng.integers(101, 6000))
        rings = int(rng.choice([4, 16, 64, 200]))
        span = float(rng.choice([3.0, 20.0, 80.0]))

        def cloud(n, spread):
            if kind == 1:  # a few tight clusters
                c = rng.uniform(-spread, spread, (8, 3))
                p = c[rng.integers(0, 8, n)] + rng.normal(0, 0.3, (n, 3))
            elif kind == 2:  # a lattice: exact ties
                p = np.round(rng.uniform(-spread, spread, (n, 3)) * 2.0) / 2.0
            elif kind == 3:  # a plane and a wall, as a sweep sees them
                p = rng.uniform(-spread, spread, (n, 3))
                p[: n // 2, 2] = rng.normal(0, 0.02, n // 2)
                p[n // 2:, 0] = spread + rng.normal(0, 0.02, n - n // 2)
            else:
                p = rng.uniform(-spread, spread, (n, 3))
            ring = np.sort(rng.integers(0, rings, n)) if kind != 5 else rng.integers(0, rings, n)  # kind 5: not in ring order
            w = ring + rng.uniform(0.0, 0.0999, n)
            return np.concatenate([p, w[:, None]], axis=1).astype(np.float32)

        lc, ls = cloud(n_lc, span), cloud(n_ls, span)
        motion = np.array([0.01, -0.005, 0.02, 0.2, -0.1, 0.05]) * rng.uniform(0, 2)
        n_sh, n_fl = int(rng.integers(1, 300)), int(rng.integers(1, 900))

        def queries(src, n):
            q = src[rng.integers(0, len(src), n)].copy()
            q[:, :3] += rng.normal(0, 0.05, (n, 3)).astype(np.float32) + motion[3:].astype(np.float32)
            far = rng.random(n) < 0.1
            q[far, :3] += rng.uniform(8, 30, (int(far.sum()), 3)).astype(np.float32)  # nothing within the 5 m gate
            q[:, 3] = np.floor(q[:, 3]) + rng.uniform(0.0, 0.0999, n)
            return q.astype(np.float32)

        sharp, flat = queries(lc, n_sh), queries(ls, n_fl)
        p0 = (motion * rng.uniform(-0.5, 1.5)).astype(np.float32)
        mi = int(rng.choice([1, 6, 25]))
        s_g, pose_g, st_g = ctx.odometry_match(lc, ls, sharp, flat, p0, max_iterations=mi)
        s_t, pose_t, st_t = ctx.odometry_match(lc, ls, sharp, flat, p0, max_iterations=mi, trees=True)
        assert s_g == s_t, (trial, kind)
        n_cases += 1
        assert (st_g.iterations, st_g.sweeps, st_g.n_rows, st_g.n_line, st_g.n_plane, st_g.converged) == \
               (st_t.iterations, st_t.sweeps, st_t.n_rows, st_t.n_line, st_t.n_plane, st_t.converged), (trial, kind)
        assert np.array_equal(bits(pose_g), bits(pose_t)), (trial, kind)
    assert n_cases == 36
    ctx.map_set(small_problem["map_corner"], small_problem["map_surf"])
