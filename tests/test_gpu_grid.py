"""The cell-grid 5-NN (csrc/lslam_grid.hpp, LSLAM_SEARCH_GRID) under the same bars as the tree search: indices and squared
distances bit-exact against the goldens generated from the reference's own nanoflann, against the oracle's kd-tree and
against the lane search of the library itself -- whatever share of the queries the probe can prove; the rest must have gone
through the tree (counted) and match too.  Then the grid SWEEP (sweep_grid_kernel + sweep_queue_kernel) against the tree
sweep and the oracle: taps bit-exact, sums to rounding, Gauss-Newton loops to the pose bar.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LANE, GRID = 1, 3          # LSLAM_SEARCH_LANE / _GRID
POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-5


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def test_grid_knn5_matches_reference_goldens(ctx, goldens):
    """Every golden fixture (lattices and duplicates included: exact distance ties must be REFUSED by the proof and answered by
    nanoflann's traversal) through the grid tap."""
    unproven = total = 0
    for name, g in goldens.items():
        pts = g["pts"]
        if len(pts) < 5:
            continue
        ctx.map_set(pts, pts)
        for which in (0, 1):
            idx, d2, n_un = ctx.knn5(which, g["queries"], search_mode=GRID, want_ties=True)
            assert np.array_equal(idx, g["idx"]), (name, which)
            assert np.array_equal(bits(d2), bits(g["d2"])), (name, which)
            unproven += n_un
            total += len(g["queries"])
    assert 0 < unproven < total  # the tie fixtures were refused, and not everything was


@pytest.mark.parametrize("spacing,jitter", [(0.2, 0.05), (0.4, 0.1), (0.4, 0.0), (1.5, 0.3)])
def test_grid_knn5_equals_lane_search_on_planes_and_lines(ctx, spacing, jitter):
    """Voxel-map-like clouds (a ground plane, two walls, poles) at several densities; queries on, near and far from them,
    outside the map and not-a-number.  spacing 0.4 / jitter 0 is an exact lattice (ties everywhere); spacing 1.5 is sparser
    than the grid's guaranteed radius (most queries unproven)."""
    rng = np.random.default_rng(int(spacing * 1000 + jitter * 100))
    g = np.arange(-20.0, 20.0, spacing, dtype=np.float32)
    gx, gy = np.meshgrid(g, g)
    ground = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size, np.float32)], 1)
    h = np.arange(0.0, 6.0, spacing, dtype=np.float32)
    wx, wz = np.meshgrid(g, h)
    wall1 = np.stack([wx.ravel(), np.full(wx.size, 7.3, np.float32), wz.ravel()], 1)
    wall2 = np.stack([np.full(wx.size, -9.1, np.float32), wx.ravel(), wz.ravel()], 1)
    poles = np.concatenate([np.stack([np.full(len(h), x, np.float32), np.full(len(h), y, np.float32), h], 1)
                            for x, y in rng.uniform(-18, 18, (25, 2))])
    pts = np.concatenate([ground, wall1, wall2, poles]).astype(np.float32)
    pts += rng.uniform(-jitter, jitter, pts.shape).astype(np.float32) if jitter > 0 else 0
    ctx.map_set(pts, pts)
    q = np.concatenate([
        pts[rng.integers(0, len(pts), 6000)] + rng.normal(0, 0.15, (6000, 3)).astype(np.float32),
        pts[rng.integers(0, len(pts), 2000)] + rng.normal(0, 1.0, (2000, 3)).astype(np.float32),
        pts[rng.integers(0, len(pts), 500)],                      # exactly on map points
        rng.uniform(-60, 60, (500, 3)).astype(np.float32),        # mostly outside
        np.array([[np.nan, 0, 0], [0, np.inf, 0], [1e9, 1e9, 1e9]], np.float32)]).astype(np.float32)
    li, ld = ctx.knn5(1, q, search_mode=LANE)
    gi, gd, n_un = ctx.knn5(1, q, search_mode=GRID, want_ties=True)
    fin = np.isfinite(q).all(1)
    assert np.array_equal(gi[fin], li[fin])
    assert np.array_equal(bits(gd[fin]), bits(ld[fin]))
    if spacing == 0.4 and jitter == 0.0:
        assert n_un > 0.2 * len(q)       # a lattice: exact ties wherever a query sits near a symmetry plane of it
    if spacing == 0.2:
        assert n_un < 0.35 * len(q)      # dense and jittered: the probe proves most of them


def test_grid_knn5_matches_oracle_on_synthetic_map(ctx, oracle, synth):
    pr = synth.make_problem(rings=16, azimuth_steps=1800, world_half=100.0)
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    rng = np.random.default_rng(1)
    for which, cloud in ((0, pr["map_corner"]), (1, pr["map_surf"])):
        tree = oracle.kdtree(cloud)
        q = cloud[rng.integers(0, len(cloud), 4000), :3] + rng.normal(0, 0.3, (4000, 3)).astype(np.float32)
        q = np.concatenate([q, rng.uniform(-150, 150, (500, 3)).astype(np.float32)])
        gi, gd = ctx.knn5(which, q, search_mode=GRID)
        oi, od = tree.knn(q, 5)
        assert np.array_equal(gi, oi)
        assert np.array_equal(bits(gd), bits(od))


def test_grid_sweep_taps_equal_tree_sweep(ctx, small_problem):
    """One sweep at the initial pose: per-point neighbours, distances, flags and coefficients of the grid sweep (both passes)
    are the tree sweep's, bit for bit; the sums differ by summation order only."""
    pr = small_problem
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    for jtj in (0, 1):
        a = ctx.sweep(pr["init_pose"], jtj_mode=jtj, search_mode=LANE)
        b = ctx.sweep(pr["init_pose"], jtj_mode=jtj, search_mode=GRID)
        assert np.array_equal(a["flags"], b["flags"])
        assert np.array_equal(a["idx"], b["idx"])
        assert np.array_equal(bits(a["d2"]), bits(b["d2"]))
        assert np.array_equal(bits(a["coeff"]), bits(b["coeff"]))
        assert a["sums"][27] == b["sums"][27] and a["sums"][28] == b["sums"][28]
        assert np.allclose(a["sums"], b["sums"], rtol=2e-5, atol=1e-3)
    assert ctx.grid_launches() >= 2


@pytest.mark.parametrize("cell", [0.0, 0.4, 1.0])
def test_grid_scanmatch_matches_oracle_and_lane(ctx, oracle, small_problem, cell):
    """The whole Gauss-Newton loop through the grid sweep (bounded, clipped probes from the second sweep on): the oracle's
    iteration and row counts, its pose to the bar; the lane search's pose to rounding."""
    pr = small_problem
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    o = ctx.default_opts()
    o.search_mode = GRID
    o.grid_cell = cell
    o.debug_stats = 1
    before = ctx.grid_launches()
    s0 = ctx.cert_stats()
    status, pose, st = ctx.run(pr["init_pose"], o)
    s1 = ctx.cert_stats()
    assert ctx.grid_launches() - before == st.sweeps + (0 if st.sweeps == o.max_iterations else 1) or ctx.grid_launches() > before
    swept, listed = s1[1] - s0[1], s1[0] - s0[0]
    assert swept >= st.sweeps * (len(pr["corner"]) + len(pr["surf"])) and 0 <= listed <= swept
    ok, opose, ost = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])
    assert (status == 0) == ok and st.iterations == ost.iterations
    assert (st.n_rows, st.n_line, st.n_plane) == (ost.n_rows, ost.n_line, ost.n_plane)
    assert np.abs(pose[3:] - opose[3:]).max() <= POSE_TOL_M and np.abs(pose[:3] - opose[:3]).max() <= POSE_TOL_RAD
    o.search_mode = LANE
    status2, pose2, st2 = ctx.run(pr["init_pose"], o)
    assert st2.iterations == st.iterations and np.abs(pose2 - pose).max() <= 2e-6


def test_grid_batch_equals_lane_batch(ctx, synth):
    """A batch of different scans in flight (the bench's shape, small): the same iteration counts, row counts and poses as the
    tree sweeps of the same batch, scan by scan; in chunks too."""
    pr0 = synth.make_problem(rings=16, azimuth_steps=900, world_half=60.0)
    ctx.map_set(pr0["map_corner"], pr0["map_surf"])
    scans, inits = [], []
    for k in range(6):
        pr = synth.make_problem(rings=16, azimuth_steps=900, world_half=60.0, seed=k)
        scans.append((pr["corner"], pr["surf"]))
        inits.append(synth.perturb_pose(pr["gt_pose"], seed=40 + k))
    ctx.scan_set_batch(scans)
    inits = np.stack(inits)
    o = ctx.default_opts()
    o.knn_cert = 0
    o.search_mode = LANE
    _, p_lane, s_lane = ctx.run_batch(inits, o)
    for in_flight in (0, 4):
        o.search_mode = GRID
        o.scans_in_flight = in_flight
        _, p_grid, s_grid = ctx.run_batch(inits, o)
        for a, b in zip(s_lane, s_grid):
            assert (a.iterations, a.n_rows, a.n_line, a.n_plane, a.converged) == (b.iterations, b.n_rows, b.n_line, b.n_plane, b.converged)
        assert np.abs(p_lane - p_grid).max() <= 2e-6
        _, p_again, _ = ctx.run_batch(inits, o)
        assert np.array_equal(p_again, p_grid)  # fixed places, fixed order: the same bits in every run
        # the A/B switch "second probe" (125 cells for the points the 27-cell probe cannot prove, tree search only for what
        # that leaves): another grouping of the same terms
        o.ab_switches = 4
        _, p_2, s_2 = ctx.run_batch(inits, o)
        o.ab_switches = 0
        for a, b in zip(s_lane, s_2):
            assert (a.iterations, a.n_rows, a.n_line, a.n_plane, a.converged) == (b.iterations, b.n_rows, b.n_line, b.n_plane, b.converged)
        assert np.abs(p_lane - p_2).max() <= 2e-6


def test_grid_fine_score_resweep(ctx, small_problem):
    """The _fineScore re-sweep (unbounded, converged scans only) through the grid sweep gives the tree sweep's score2 / percent2."""
    pr = small_problem
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    o = ctx.default_opts()
    o.fine_score = 1
    res = {}
    for mode in (LANE, GRID):
        o.search_mode = mode
        status, pose, st = ctx.run(pr["init_pose"], o)
        res[mode] = (st.converged, st.percent2, st.score2)
    assert res[LANE][0] == res[GRID][0] == 1
    assert res[LANE][1] == res[GRID][1] and abs(res[LANE][2] - res[GRID][2]) <= 1e-6 * max(1.0, abs(res[LANE][2]))


def test_grid_run_does_not_read_what_an_earlier_call_left(ctx, small_problem):
    """Every call starts cold: the grid sweep's second pass runs in a loop's FIRST sweep too, where nothing carried over from
    an earlier call (another pose, another search's bookkeeping) may bound a search.  A certificate-sweep call, whose per-point
    state is of another kind, right before a grid call; the grid call's pose must be the one it gives after itself."""
    pr = small_problem
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    g = ctx.default_opts()
    g.search_mode = GRID
    c = ctx.default_opts()
    c.search_mode = LANE
    c.knn_cert = 2
    _, ref, st_ref = ctx.run(pr["init_pose"], g)
    for other, pose0 in ((c, pr["init_pose"]), (c, pr["gt_pose"]), (g, pr["gt_pose"])):
        ctx.run(pose0, other)
        _, pose, st = ctx.run(pr["init_pose"], g)
        assert np.array_equal(bits(pose), bits(ref)) and (st.iterations, st.n_rows) == (st_ref.iterations, st_ref.n_rows)


def _mapping_frames(pkg, ctx, synth, world, defer, n=4, lattice=False, ab_switches=0):
    """A few LaserMapping frames (the device chain of tests/test_gpu_pipeline.py, mapping half only) -> poses, stats."""
    mapper = pkg.LaserMapping(ctx, cube_dims=(21, 21, 11), defer_trees=defer)
    mapper.opts.ab_switches = ab_switches
    out = []
    for k in range(n):
        gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
        c, s, gtp = synth.make_scan(world, 16, 900, gt_pose=gt, seed=300 + k)
        if lattice:  # centimetre coordinates: exact distance ties between map points
            c, s = np.round(c, 2).astype(np.float32), np.round(s, 2).astype(np.float32)
        odom = ctx.pose_to_isometry(synth.perturb_pose(gtp, seed=900 + k, dt=0.1, dr_deg=0.5))
        M = mapper.process(c, s, odom)
        st = mapper.last_stats
        out.append((M.copy(), None if st is None else (st.iterations, st.n_rows, st.n_line, st.n_plane)))
    mapper.feature_map.close()
    return out


def test_deferred_trees_mapping_frames_equal_eager(pkg, synth, small_problem):
    """The mapping node's frames with the per-frame kd-tree build deferred (lslam_map_defer_trees: grids at map-set time, the
    27-cell probe + the wide probe for what it cannot prove, no tree at all) against the same frames with the trees built
    every frame: the same neighbours, so the same iteration and row counts and poses to the rounding of differently grouped
    sums.  The deferred runs really ran without trees."""
    world = small_problem["world"]
    res = {}
    for defer in (False, True):
        c = pkg.Context(0)
        try:
            res[defer] = _mapping_frames(pkg, c, synth, world, defer)
            sets, builds, pending = c.lazy_trees()
            if defer:
                assert sets >= 3 and builds == 0 and pending, (sets, builds, pending)  # (the first frame's map is empty: nothing to match)
                assert c.grid_launches() > 0
                # ... and a tap on that map builds them on the spot
                q = np.zeros((4, 3), np.float32)
                c.knn5(1, q)
                assert c.lazy_trees() == (sets, 1, False)
            else:
                assert (sets, builds) == (0, 0)
        finally:
            c.close()
    for (Ma, sa), (Mb, sb) in zip(res[False], res[True]):
        assert sa == sb
        assert np.abs(Ma - Mb).max() <= 2e-6
    # the A/B form of the deferred sweep (LSLAM_AB_WIDE_IN_PLACE: unproven points resolved inside the probe's launch, one
    # launch per sweep) -- it forms a workgroup's sums the way the lane search's sweep does, so against the EAGER frames
    # (trees, lane search: what AUTO takes for a single scan) it is the same bits
    c = pkg.Context(0)
    try:
        inplace = _mapping_frames(pkg, c, synth, world, True, ab_switches=8)  # LSLAM_AB_WIDE_IN_PLACE
        sets, builds, pending = c.lazy_trees()
        assert sets >= 3 and builds == 0 and pending
        assert c.grid_wide_launches() > 0 and c.grid_wide_launches() == c.grid_launches()
    finally:
        c.close()
    for (Ma, sa), (Mb, sb) in zip(res[False], inplace):
        assert sa == sb
        assert np.array_equal(bits(Ma), bits(Mb))


def test_deferred_trees_for_a_host_map(pkg, small_problem):
    """lslam_map_set with host clouds under lslam_map_defer_trees (ScanMatch::scanMatchScan's call shape): uploaded, grids
    only; the scan match equals the eager one to the rounding of differently grouped sums, and no tree was built."""
    pr = small_problem
    res = {}
    for defer in (False, True):
        c = pkg.Context(0)
        try:
            c.defer_trees(defer)
            c.map_set(pr["map_corner"], pr["map_surf"])
            c.scan_set(pr["corner"], pr["surf"])
            status, pose, st = c.run(pr["init_pose"])
            res[defer] = (int(status), pose.copy(), st.iterations, st.n_rows)
            assert c.lazy_trees() == ((1, 0, True) if defer else (0, 0, False))
            if defer:
                assert c.grid_launches() > 0 and c.map_info().depth_corner == 0
        finally:
            c.close()
    assert res[False][0] == res[True][0] and res[False][2:] == res[True][2:]
    assert np.abs(res[False][1] - res[True][1]).max() <= 2e-6


def test_deferred_trees_fall_back_to_the_trees_on_exact_ties(pkg, synth):
    """A lattice map (one point per voxel, so the voxel filter keeps the lattice) and scan points that sit at equal distances
    from several of its points: which five nanoflann returns then depends on its visit order -- the one thing the grids
    cannot reproduce.  The deferred call notices (the wide probe raises the scan's flag), builds the trees and runs again
    through them: the eager run's bits."""
    g = np.arange(-20.0, 20.0, 0.5, dtype=np.float32)
    gx, gy = np.meshgrid(g, g)
    ground = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size, np.float32), np.zeros(gx.size, np.float32)], 1)
    h = np.arange(0.5, 6.0, 0.5, dtype=np.float32)
    wx, wz = np.meshgrid(g, h)
    wall = np.stack([wx.ravel(), np.full(wx.size, 8.0, np.float32), wz.ravel(), np.zeros(wx.size, np.float32)], 1)
    surf_map = np.concatenate([ground, wall]).astype(np.float32)
    poles = np.concatenate([np.stack([np.full(len(h), x, np.float32), np.full(len(h), y, np.float32), h, np.zeros(len(h), np.float32)], 1)
                            for x in (-6.0, -2.0, 2.0, 6.0) for y in (-5.0, 0.0, 5.0)])
    rng = np.random.default_rng(5)
    # scan (sensor frame = map frame at the true pose): surf points above cell centres of the lattice (four equidistant
    # neighbours), corner points beside the poles
    qs = np.stack([gx.ravel()[::3] + 0.25, gy.ravel()[::3] + 0.25, np.full(gx.size, 0.0, np.float32)[::3], np.zeros(gx.size, np.float32)[::3]], 1).astype(np.float32)
    qc = (poles + np.array([0.0, 0.05, 0.25, 0.0], np.float32)).astype(np.float32)
    init = np.zeros(6, np.float32)  # the first sweep sees the scan exactly where the ties are
    res = {}
    for defer in (False, True):
        c = pkg.Context(0)
        try:
            c.defer_trees(defer)
            fm = pkg.FeatureMap(c, 21, 21, 11)
            fm.setup_filter_size(0.05, 0.05, 0.05)
            fm.update(np.zeros(3, np.float32))
            fm.add_feature_cloud(poles, surf_map, np.eye(4, dtype=np.float32))
            fm.surround_to_map()
            status, pose, st = c.scanmatch_scan(qc, qs, init)
            res[defer] = (int(status), pose.copy(), st.iterations, st.n_rows, st.n_line, st.n_plane)
            if defer:
                sets, builds, pending = c.lazy_trees()
                assert sets == 1 and builds == 1 and not pending, (sets, builds, pending)  # the trees were needed, and built once
            fm.close()
        finally:
            c.close()
    assert res[False][0] == res[True][0] and res[False][2:] == res[True][2:]
    assert np.array_equal(bits(res[False][1]), bits(res[True][1]))
    assert res[True][3] > 100


def test_grid_limits_fall_back_to_the_tree(pkg, ctx):
    """What the grid cannot take is searched by the tree, silently and exactly: (a) a blob denser than a row's candidate
    counter (more than 64 points in a run of three cells) -- every probe there is 'unproven'; (b) a map whose extent exceeds
    the cell table's limit (2 048 cells per axis): no grid is built at all -- the tap says so, a scan match asked for the grid
    sweep runs the kd-tree walk, and deferring the trees of such a map builds them at once."""
    rng = np.random.default_rng(11)
    blob = rng.uniform(-4, 4, (150000, 3)).astype(np.float32)        # ~ 60 points per 0.6 m cell
    ctx.map_set(blob, blob)
    q = rng.uniform(-3, 3, (3000, 3)).astype(np.float32)
    li, ld = ctx.knn5(1, q, search_mode=LANE)
    gi, gd, n_un = ctx.knn5(1, q, search_mode=GRID, want_ties=True)
    assert np.array_equal(gi, li) and np.array_equal(bits(gd), bits(ld)) and n_un > 0.9 * len(q)
    # (a') cells with more points than the grid build ranks by original index (1 024: their order is whatever the placement's
    # atomics gave) next to ordinary ones: such cells are never proven from, so the answers are the tree's -- in every build
    clump = (rng.normal(0, 0.08, (6000, 3)) + np.array([1.0, 1.0, 1.0])).astype(np.float32)
    sparse = rng.uniform(-6, 6, (40000, 3)).astype(np.float32)
    both = np.concatenate([clump, sparse]).astype(np.float32)
    q2 = np.concatenate([rng.normal(0, 0.3, (500, 3)) + np.array([1.0, 1.0, 1.0]), rng.uniform(-5, 5, (1500, 3))]).astype(np.float32)
    first = None
    for build in range(3):
        ctx.map_set(both, both)
        li2, ld2 = ctx.knn5(1, q2, search_mode=LANE)
        gi2, gd2, n_un2 = ctx.knn5(1, q2, search_mode=GRID, want_ties=True)
        assert np.array_equal(gi2, li2) and np.array_equal(bits(gd2), bits(ld2)) and 0 < n_un2 < len(q2)
        first = first if first is not None else (gi2.copy(), gd2.copy())
        assert np.array_equal(first[0], gi2) and np.array_equal(bits(first[1]), bits(gd2))
    # (b)
    line = np.stack([np.linspace(-900.0, 900.0, 4000), np.zeros(4000), np.zeros(4000)], 1).astype(np.float32)
    wide = np.concatenate([line, line + np.array([0, 0.4, 0], np.float32), line + np.array([0, 0.8, 0.3], np.float32)]).astype(np.float32)
    ctx.map_set(wide, wide)
    with pytest.raises(pkg.LslamError):
        ctx.knn5(1, wide[:10], search_mode=GRID)
    scan = (wide[::7] + np.array([0.02, 0.01, 0.0], np.float32)).astype(np.float32)
    ctx.scan_set(scan[:200], scan)
    o = ctx.default_opts()
    o.search_mode = GRID
    g0 = ctx.grid_launches()
    s_g, p_g, st_g = ctx.run(np.zeros(6, np.float32), o)
    o.search_mode = LANE
    s_l, p_l, st_l = ctx.run(np.zeros(6, np.float32), o)
    assert ctx.grid_launches() == g0 and np.array_equal(bits(p_g), bits(p_l)) and st_g.iterations == st_l.iterations
    c2 = pkg.Context(0)
    try:
        c2.defer_trees(True)
        fm = pkg.FeatureMap(c2, 121, 121, 11)
        fm.setup_filter_size(0.05, 0.05, 0.05)
        fm.setup_lidar_valid_distance(2000.0) if hasattr(fm, "setup_lidar_valid_distance") else None
        fm.update(np.zeros(3, np.float32))
        wide4 = np.c_[wide, np.zeros(len(wide), np.float32)].astype(np.float32)
        fm.add_feature_cloud(wide4, wide4, np.eye(4, dtype=np.float32))
        fm.surround_to_map()
        sets, builds, pending = c2.lazy_trees()
        info = c2.map_info()
        # either the surround is small enough for a grid (deferred) or it is not (built at once): never a map without a search structure
        assert pending == (info.depth_surf == 0) and (pending or info.depth_surf > 0)
        fm.close()
    finally:
        c2.close()


def _seven_in_a_row_problem(with_cluster):
    """A jittered ground patch + wall (surf) and poles (corner) matched at the identity pose, and -- with_cluster -- seven map
    points in ONE cell row, 1.0 ... 1.1 m from a lone scan point: four clearly nearer ones, then three whose squared distances
    share one 2^-13 truncation bucket of the probe's keys, the NEAREST of the three last in index order."""
    rng = np.random.default_rng(77)
    g = np.arange(-16.0, 16.0, 0.4, dtype=np.float32)
    gx, gy = np.meshgrid(g, g)
    ground = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size, np.float32)], 1)
    ground += rng.uniform(-0.1, 0.1, ground.shape).astype(np.float32) * np.array([1, 1, 0.05], np.float32)
    h = np.arange(0.3, 5.0, 0.2, dtype=np.float32)
    poles = np.concatenate([np.stack([np.full(len(h), x, np.float32), np.full(len(h), y, np.float32), h], 1)
                            for x in (-9.0, -3.0, 3.0, 9.0) for y in (-8.0, 0.0, 8.0)]).astype(np.float32)
    poles += rng.uniform(-0.01, 0.01, poles.shape).astype(np.float32)
    qs = (ground[::5] + rng.normal(0, 0.02, ground[::5].shape).astype(np.float32)).astype(np.float32)
    qc = (poles[::2] + np.array([0.03, 0.02, 0.05], np.float32)).astype(np.float32)
    surf = ground
    if with_cluster:
        q = np.array([3.30, 2.10, 30.0], np.float32)              # far above everything else
        xs = [1.00, 1.02, 1.04, 1.06]
        # the three of the bucket: squared distances (the kernel's fp32 arithmetic: only dx is non-zero) with equal upper 22 bits
        base = np.float32(1.10)
        trio = None
        for a in np.arange(0.0, 2e-3, 1e-6, dtype=np.float32):
            cand = [np.float32(q[0] + base + a + np.float32(1e-5)), np.float32(q[0] + base + a + np.float32(2e-5)), np.float32(q[0] + base + a)]
            d2 = [np.float32(np.float32(q[0] - c) * np.float32(q[0] - c)) for c in cand]
            b = [int(np.float32(v).view(np.uint32)) >> 10 for v in d2]
            if b[0] == b[1] == b[2] and d2[2] < d2[0] < d2[1]:
                trio = cand
                break
        assert trio is not None
        cluster = np.array([[q[0] + x, q[1], q[2]] for x in xs] + [[c, q[1], q[2]] for c in trio], np.float32)
        surf = np.concatenate([ground, cluster]).astype(np.float32)   # the cluster last: ascending original index = A, B, C
        qs = np.concatenate([qs, q[None, :]]).astype(np.float32)
    return poles, surf, qc, qs


def _exact_five(q, pts):
    """the five nearest by the reference's own fp32 distance (x -> y -> z), ascending; and the sixth distance"""
    d = (q[None, :].astype(np.float32) - pts[:, :3].astype(np.float32)).astype(np.float32)
    d2 = ((d[:, 0] * d[:, 0]).astype(np.float32) + (d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
    o = np.argsort(d2, kind="stable")
    return o[:5], d2[o[:5]], d2[o[5]]


def test_wide_probe_finds_a_candidate_dropped_behind_six_keys_of_one_lane(pkg):
    """sweep_wide_kernel gives a cell row to ONE lane, and a lane keeps its six smallest TRUNCATED keys.  Seven candidates in a
    row, the fifth to seventh nearest inside one truncation bucket with the nearest of them LAST in index order: the lane
    drops exactly the candidate that is the true fifth neighbour, and all six wave-wide winners are that lane's (round 4's
    kernel returned the wrong fifth and called it proven).  The lane now looks at its rows once more, exactly: the tap returns
    the true five, decided, without a tree; and a whole scan match on that map runs on the grids alone and gives the eager
    call's result."""
    mc, ms, qc, qs = _seven_in_a_row_problem(True)
    q = qs[-1]
    five, d5, d6 = _exact_five(q, ms)
    assert set(five.tolist()) == {len(ms) - 7, len(ms) - 6, len(ms) - 5, len(ms) - 4, len(ms) - 1}  # the four near ones and C
    c = pkg.Context(0)
    try:
        c.defer_trees(True)
        c.map_set(mc, ms)
        idx, d2, und = c.knn5_wide(1, q[None, :3])
        assert und[0] == 0 and np.array_equal(idx[0], five) and np.array_equal(bits(d2[0]), bits(d5))
        assert c.lazy_trees() == (1, 0, True)
    finally:
        c.close()
    res = {}
    for defer in (False, True):
        c = pkg.Context(0)
        try:
            c.defer_trees(defer)
            c.map_set(mc, ms)
            c.scan_set(qc, qs)
            status, pose, st = c.run(np.zeros(6, np.float32))
            res[defer] = (int(status), pose.copy(), st.iterations, st.n_rows, st.n_plane, c.lazy_trees())
        finally:
            c.close()
    assert res[True][5] == (1, 0, True), res[True][5]       # no tree was needed
    assert res[False][0] == res[True][0] and res[False][2:5] == res[True][2:5] and res[True][3] > 100
    assert np.abs(res[False][1] - res[True][1]).max() <= 2e-6


@pytest.mark.parametrize("spacing,jitter", [(0.2, 0.05), (0.4, 0.1), (0.4, 1e-6), (1.5, 0.3)])
def test_wide_probe_tap_equals_tree_search(pkg, spacing, jitter):
    """The search a map without kd-trees is matched through, query by query (lslam_debug_knn5_wide) against nanoflann's
    traversal on the same clouds (another context, trees built): planes, walls and poles at several densities -- 0.4 / 1e-6 is
    a lattice jittered by micrometres (near-ties everywhere).  Wherever the wide probe DECIDES and the reference looks the
    neighbours up at all (d2[4] < 5), indices and distances are nanoflann's bit for bit; undecided only where there is an exact
    tie among the six nearest (checked by brute force)."""
    rng = np.random.default_rng(int(spacing * 1000 + jitter * 100))
    g = np.arange(-12.0, 12.0, spacing, dtype=np.float32)
    gx, gy = np.meshgrid(g, g)
    ground = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size, np.float32)], 1)
    h = np.arange(0.0, 5.0, spacing, dtype=np.float32)
    wx, wz = np.meshgrid(g, h)
    wall = np.stack([wx.ravel(), np.full(wx.size, 5.3, np.float32), wz.ravel()], 1)
    poles = np.concatenate([np.stack([np.full(len(h), x, np.float32), np.full(len(h), y, np.float32), h], 1)
                            for x, y in rng.uniform(-10, 10, (12, 2))])
    pts = np.concatenate([ground, wall, poles]).astype(np.float32)
    pts = (pts + rng.uniform(-jitter, jitter, pts.shape)).astype(np.float32)
    q = np.concatenate([pts[rng.integers(0, len(pts), 1500)] + rng.normal(0, 0.15, (1500, 3)),
                        pts[rng.integers(0, len(pts), 500)] + rng.normal(0, 0.8, (500, 3)),
                        pts[rng.integers(0, len(pts), 100)]]).astype(np.float32)
    a = pkg.Context(0)
    b = pkg.Context(0)
    try:
        a.map_set(pts, pts)
        li, ld = a.knn5(1, q, search_mode=LANE)
        b.defer_trees(True)
        b.map_set(pts, pts)
        wi, wd, und = b.knn5_wide(1, q)
        assert b.lazy_trees() == (1, 0, True)
    finally:
        a.close()
        b.close()
    looked_up = ld[:, 4] < 5.0
    dec = (und == 0) & looked_up
    assert dec.sum() > 0.5 * looked_up.sum()
    assert np.array_equal(wi[dec], li[dec]) and np.array_equal(bits(wd[dec]), bits(ld[dec]))
    for i in np.nonzero((und != 0) & looked_up)[0][:200]:  # undecided => an exact tie among the six nearest
        five, d5, d6 = _exact_five(q[i], pts)
        six = np.concatenate([d5, [d6]])
        assert len(set(six.tolist())) < 6, (i, six)


def test_wide_probe_near_tie_margin_is_an_option(pkg):
    """LSLAM_AB_WIDE_NF_MARGIN (lslam_opts.ab_switches & 16; the tap's nf_margin): a map without kd-trees and a point whose
    fifth and sixth neighbours are two ulps apart.  Off (the default) the pair is ordered by its exact distances -- decided,
    no tree; on, the point counts as undecidable: a scan match builds the trees and repeats the call.  Either way the eager
    call's result."""
    mc, ms, qc, qs = _seven_in_a_row_problem(False)
    q = np.array([3.30, 2.10, 30.0], np.float32)
    near = [[q[0] + dx, q[1], q[2]] for dx in (1.00, 1.02, 1.04, 1.06)]
    xa = np.float32(q[0] + np.float32(1.10))
    pa = np.array([xa, np.float32(q[1] + np.float32(0.01)), q[2]], np.float32)
    pb = np.array([xa, np.float32(q[1] + np.float32(0.010015)), q[2]], np.float32)
    ms2 = np.concatenate([ms, np.array(near + [list(pa), list(pb)], np.float32)]).astype(np.float32)
    five, d5, d6 = _exact_five(q, ms2)
    assert d5[4] < d6 and (d6 - d5[4]) / d6 < 8 * 2.0 ** -23 and five[4] == len(ms2) - 2
    qs2 = np.concatenate([qs, q[None, :]]).astype(np.float32)
    c = pkg.Context(0)
    try:
        c.defer_trees(True)
        c.map_set(mc, ms2)
        idx, d2, und = c.knn5_wide(1, q[None, :3])
        assert und[0] == 0 and np.array_equal(idx[0], five) and np.array_equal(bits(d2[0]), bits(d5))
        idx, d2, und = c.knn5_wide(1, q[None, :3], nf_margin=True)
        assert und[0] == 1
        assert c.lazy_trees() == (1, 0, True)
    finally:
        c.close()
    res = {}
    for name, defer, ab in (("eager", False, 0), ("deferred", True, 0), ("deferred+margin", True, 16)):
        c = pkg.Context(0)
        try:
            c.defer_trees(defer)
            c.map_set(mc, ms2)
            c.scan_set(qc, qs2)
            o = c.default_opts()
            o.ab_switches = ab
            status, pose, st = c.run(np.zeros(6, np.float32), o)
            res[name] = (int(status), pose.copy(), st.iterations, st.n_rows, c.lazy_trees())
        finally:
            c.close()
    assert res["deferred"][4] == (1, 0, True), res["deferred"][4]          # decided by exact distances: no tree
    assert res["deferred+margin"][4] == (1, 1, False), res["deferred+margin"][4]  # refused: trees built, call repeated
    for k in ("deferred", "deferred+margin"):
        assert res[k][0] == res["eager"][0] and res[k][2:4] == res["eager"][2:4]
        assert np.abs(res[k][1] - res["eager"][1]).max() <= 2e-6
    assert np.array_equal(bits(res["deferred+margin"][1]), bits(res["eager"][1]))


@pytest.mark.parametrize("from_sweep", [0, 1, 2])
def test_grid_fit_cache_gives_the_same_bits(pkg, synth, monkeypatch, from_sweep):
    """LSLAM_AB_FIT_CACHE: surf points whose five neighbours did not change reuse the previous sweep's plane, the others are
    compacted through LDS and fitted by as few wavefronts as they fill.  findPlane is a pure function of the five in order and
    every lane still forms its own row, so the loop is the SAME BITS as without the cache -- with the cache used from the
    default sweep (0 = the library's: the fourth) and, forced through the environment, from a loop's second and third sweep
    (where most points' neighbours DID change: both the compacted and the everybody-fits form run)."""
    if from_sweep:
        monkeypatch.setenv("LSLAM_FIT_FROM_SWEEP", str(from_sweep))
    c = pkg.Context(0)
    try:
        pr0 = synth.make_problem(rings=16, azimuth_steps=900, world_half=60.0)
        c.map_set(pr0["map_corner"], pr0["map_surf"])
        scans, inits = [], []
        for k in range(6):
            pr = synth.make_problem(rings=16, azimuth_steps=900, world_half=60.0, seed=k)
            scans.append((pr["corner"], pr["surf"]))
            inits.append(synth.perturb_pose(pr["gt_pose"], seed=40 + k, dt=0.4, dr_deg=2.5))  # far enough for loops of 4+ sweeps
        c.scan_set_batch(scans)
        inits = np.stack(inits)
        o = c.default_opts()
        o.search_mode = GRID
        res = {}
        for ab in (0, 32, 0, 32):
            o.ab_switches = ab
            _, p, st = c.run_batch(inits, o)
            key = (ab, len([k for k in res if k[0] == ab]))
            res[key] = (p.copy(), [(s.status, s.iterations, s.n_rows, s.n_line, s.n_plane, s.converged, s.sweeps) for s in st])
        assert max(s[6] for s in res[(0, 0)][1]) >= 4  # loops long enough for cached fits to be used
        for key in ((32, 0), (0, 1), (32, 1)):
            assert res[key][1] == res[(0, 0)][1], key
            assert np.array_equal(bits(res[key][0]), bits(res[(0, 0)][0])), key
    finally:
        c.close()


def test_grid_batch_over_the_running_scans_only_gives_the_same_bits(ctx, synth):
    """A batch through the grid sweep launches, from a loop's second sweep on, only the workgroups of the scans still running
    (compact_active_kernel; one 4-byte read-back per iteration) instead of every workgroup of every scan.  Every workgroup writes
    a record of its own, so nothing depends on which others were launched: the same bits as with LSLAM_AB_NO_COMPACT, whole
    batch and in chunks, with scans that end at different iterations, one far from the map and one empty."""
    pr0 = synth.make_problem(rings=16, azimuth_steps=900, world_half=60.0)
    ctx.map_set(pr0["map_corner"], pr0["map_surf"])
    scans, inits = [], []
    for k in range(9):
        pr = synth.make_problem(rings=16, azimuth_steps=900, world_half=60.0, seed=k)
        scans.append((pr["corner"], pr["surf"]))
        inits.append(synth.perturb_pose(pr["gt_pose"], seed=40 + k, dt=0.05 + 0.06 * k, dr_deg=0.3 + 0.35 * k))  # loops of different lengths
    inits[4][3] += 400.0                                # far from the map: ends at once (too few matches)
    empty = np.zeros((0, 4), np.float32)
    scans[7] = (empty, empty)
    ctx.scan_set_batch(scans)
    inits = np.stack(inits)
    o = ctx.default_opts()
    o.search_mode = GRID
    ref = None
    for in_flight in (0, 4):
        o.scans_in_flight = in_flight
        for ab in (64, 0, 128):                         # LSLAM_AB_NO_COMPACT first; LSLAM_AB_REFILL: the second pass's two-launch form
            o.ab_switches = ab
            _, p, st = ctx.run_batch(inits, o)
            cur = (p.copy(), [(s.status, s.iterations, s.n_rows, s.n_line, s.n_plane, s.converged, s.sweeps) for s in st])
            if ref is None:
                ref = cur
                assert len({x[1] for x in cur[1]}) >= 3  # iteration counts really differ
            assert cur[1] == ref[1], (in_flight, ab)
            assert np.array_equal(bits(cur[0]), bits(ref[0])), (in_flight, ab)
