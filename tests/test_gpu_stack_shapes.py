"""The kernel the bench times under the oracle (-m gpu, through the C ABI).

`launch_sweep` (csrc/lslam_kernels.hip) has two traversal-stack shapes for the lane search of
ScanMatch.cpp:97-132: the whole 32-level stack in LDS (`sweep_kernel<256,false,false,32>`, what a single
scan's latency-bound launch takes) and 12 levels in LDS with an HBM overflow area
(`sweep_kernel<256,true,false,12>`: five wavefronts per SIMD, two-entry pop rounds), which every launch of
more than 2 048 wavefronts takes -- i.e. every batch, and the whole timed region of bench.py.  These tests put
THAT instantiation against the oracle: forced through the LSLAM_STACK_* bits on small problems (here and in
tests/test_gpu_parity.py), and picked by the library itself on batches large enough -- with
`lslam_debug_sweep_launches` saying which kernel really ran.
"""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4    # BASELINE.json north_star: pose within 1e-4 m of the CPU reference
POSE_TOL_RAD = 1e-5
LANE, DEEP, SHALLOW = 1, 0x100, 0x200  # LSLAM_SEARCH_LANE, LSLAM_STACK_DEEP, LSLAM_STACK_SHALLOW
GRID = 3                                # LSLAM_SEARCH_GRID
CARRIED, FIRST = 0x400, 0x800           # LSLAM_SWEEP_CARRIED / _FIRST: the tap's sweep as a later / the first sweep of the production loop


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _counts_close(a, b):
    """Match counts of the LAST sweep of a loop.  GPU and oracle sum A^T A in different orders, so from the second iteration
    on their poses differ in the last bits (<= 1e-6 m); a point that sits exactly on a gate (d2[4] < 5, the 0.2 m plane
    bound, w > 0.1) can then fall on either side.  At a FIXED pose flags are compared bit for bit (the sweep taps); here a
    handful of threshold-adjacent points out of >= 10^4 may differ."""
    return all(abs(int(x) - int(y)) <= max(2, int(1e-4 * max(x, y))) for x, y in zip(a, b))


def _ran(before, after):
    return {k: after[k] - before[k] for k in after if after[k] != before[k]}


def _deepen(pts, n_clusters, seed):
    """Geometrically spaced far clusters appended to a cloud: nanoflann's mid-split peels them off one or two at a time,
    so the tree over the real points hangs ~n_clusters levels down (deeper than the 33 levels the LDS stack covers)."""
    rng = np.random.default_rng(seed)
    c = 500.0 * 1.6 ** np.arange(n_clusters)
    extra = np.concatenate([np.stack([cc + rng.uniform(0, 0.01 * cc, 12), rng.uniform(-1, 1, 12), rng.uniform(0, 1, 12)], 1)
                            for cc in c]).astype(np.float32)
    out = np.zeros((len(extra), pts.shape[1]), np.float32)
    out[:, :3] = extra
    return np.concatenate([pts, out])


@pytest.mark.parametrize("shape", ["deep", "shallow"])
def test_full_loop_on_a_deep_tree_matches_oracle(ctx, oracle, small_problem, shape):
    """The whole Gauss-Newton loop against trees 40 / 46 levels deep: the overflow area of BOTH shapes is really used
    (levels >= 32 of the deep shape, levels >= 12 of the shallow one), sweeps and final pose against the oracle."""
    pr = small_problem
    mc, ms = _deepen(pr["map_corner"], 34, 1), _deepen(pr["map_surf"], 34, 2)
    tc, ts = oracle.kdtree(mc), oracle.kdtree(ms)
    assert 34 < tc.max_depth() <= 64 and 34 < ts.max_depth() <= 64
    mode = LANE | (DEEP if shape == "deep" else SHALLOW)
    variant = "deep_ovf" if shape == "deep" else "shallow"      # the taps' launches
    loop_variant = variant
    ctx.map_set(mc, ms)
    info = ctx.map_info()
    assert info.depth_corner == tc.max_depth() and info.depth_surf == ts.max_depth()
    ctx.scan_set(pr["corner"], pr["surf"])
    before = ctx.sweep_launches()
    for pose in (pr["init_pose"], pr["gt_pose"]):  # unbounded sweeps: every neighbour list against nanoflann's
        g = ctx.sweep(pose, jtj_mode=1, search_mode=mode)
        o = oracle.sweep(tc, ts, pr["corner"], pr["surf"], pose)
        assert np.array_equal(g["idx"], o["idx"]) and np.array_equal(bits(g["d2"]), bits(o["d2"]))
        assert np.array_equal(g["flags"], o["flags"]) and np.array_equal(bits(g["coeff"]), bits(o["coeff"]))
    opts = ctx.default_opts()
    opts.search_mode = mode
    status, pose, st = ctx.run(pr["init_pose"], opts)  # bounded sweeps, neighbours of the previous sweep as the bound
    ok, opose, ost = oracle.scanmatch_scan(mc, ms, pr["corner"], pr["surf"], pr["init_pose"])
    assert (status == 0) == ok and st.iterations == ost.iterations and st.converged == ost.converged
    assert (st.n_line, st.n_plane, st.n_rows) == (ost.n_line, ost.n_plane, ost.n_rows)
    assert np.abs(pose[3:] - opose[3:]).max() <= POSE_TOL_M and np.abs(pose[:3] - opose[:3]).max() <= POSE_TOL_RAD
    assert set(_ran(before, ctx.sweep_launches())) == {variant, loop_variant}
    # AUTO on a deep tree takes the shallow kernel (bounded loop), whatever the launch size
    before = ctx.sweep_launches()
    status2, pose2, st2 = ctx.run(pr["init_pose"])
    assert set(_ran(before, ctx.sweep_launches())) == {"shallow"}
    assert np.array_equal(bits(pose2), bits(pose)) and st2.iterations == st.iterations


def test_batch_above_512_blocks_runs_the_shallow_kernel_and_matches_oracle(ctx, oracle, synth, small_problem, monkeypatch):
    """Twelve 16 x 900 scans resident together: ~640 blocks = ~2 560 wavefronts, so `launch_sweep` itself picks
    sweep_kernel<256,true,false,12> -- the instantiation bench.py times -- and the loop runs the certificate sweep (two
    passes from the second sweep on).  Every scan against `oracle.scanmatch_scan`, and bit for bit against its own run alone
    through the deep stack (certificate sweep forced there too: a launch that small would search every point, which sums the
    same terms in another grouping); one scan far from the map, one empty."""
    pr = small_problem
    world = pr["world"]
    scans, inits = [], []
    for k in range(12):
        gt = (0.004 * k, -0.006 + 0.001 * k, 0.15 + 0.09 * k, 4.0 - 0.9 * k, -3.0 + 0.55 * k, synth.SENSOR_HEIGHT)
        qc, qs, gt = synth.make_scan(world, 16, 900, gt_pose=gt, seed=300 + k)
        scans.append((qc, qs))
        inits.append(synth.perturb_pose(gt, seed=400 + k))
    inits[5][3] += 400.0                                   # far from the map: ScanMatch.cpp:141-145, pose untouched
    empty = np.zeros((0, 4), np.float32)
    scans[9] = (empty, empty)                              # no points at all
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    deep = ctx.default_opts()
    deep.search_mode = LANE | DEEP
    deep.knn_cert = 2
    single = []
    before = ctx.sweep_launches()
    for (qc, qs), p0 in zip(scans, inits):
        single.append(ctx.scanmatch_scan(qc, qs, p0, deep))
    assert set(_ran(before, ctx.sweep_launches())) == {"deep"}
    ctx.scan_set_batch(scans)
    opts = ctx.default_opts()
    opts.scans_in_flight = 12
    opts.knn_cert = 2
    opts.search_mode = LANE  # (AUTO would take the grid sweep for a batch this size: tests/test_gpu_grid.py)
    before = ctx.sweep_launches()
    worst, poses, stats = ctx.run_batch(np.stack(inits), opts)
    ran = _ran(before, ctx.sweep_launches())
    assert set(ran) == {"shallow"} and ran["shallow"] >= 3, ran  # picked by launch size, not forced
    assert ctx.cert_stats()[2] > 0  # second passes were launched
    n_blocks = sum((len(c) + 255) // 256 + (len(s) + 255) // 256 for c, s in scans)
    assert n_blocks > 512
    for k, (status, pose, st) in enumerate(single):
        assert stats[k].status == st.status and stats[k].iterations == st.iterations, k
        assert (stats[k].n_rows, stats[k].n_line, stats[k].n_plane) == (st.n_rows, st.n_line, st.n_plane), k
        assert np.array_equal(bits(poses[k]), bits(pose)), k  # shallow batch == deep single run, bit for bit
        assert stats[k].point_residuals == st.point_residuals
    for k, ((qc, qs), p0) in enumerate(zip(scans, inits)):
        ok, opose, ost = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], qc, qs, p0)
        # the reference has one `return false` for "too few matches" and "not converged" (ScanMatch.cpp:342-346, oracle: 2);
        # the C ABI tells them apart (LSLAM_TOO_FEW_MATCHES = 5)
        assert (2 if stats[k].status == 5 else stats[k].status) == ost.status, k
        assert stats[k].iterations == ost.iterations and stats[k].converged == ost.converged, k
        assert _counts_close((stats[k].n_line, stats[k].n_plane, stats[k].n_rows), (ost.n_line, ost.n_plane, ost.n_rows)), k
        assert np.abs(poses[k][3:] - opose[3:]).max() <= POSE_TOL_M, k
        assert np.abs(poses[k][:3] - opose[:3]).max() <= POSE_TOL_RAD, k
    assert stats[5].status == 5 and np.array_equal(poses[5], inits[5])
    assert stats[9].status == 5 and stats[9].n_rows == 0
    assert len({s.iterations for s in stats}) > 2  # the scans really leave the launches at different iterations
    # the same batch forced through the deep stack: same bits
    opts.search_mode = LANE | DEEP
    before = ctx.sweep_launches()
    _, poses_d, stats_d = ctx.run_batch(np.stack(inits), opts)
    assert set(_ran(before, ctx.sweep_launches())) == {"deep"}
    assert np.array_equal(bits(poses_d), bits(poses))


@pytest.fixture(scope="module")
def voxel_map_problem(pkg, synth):
    """A reduced BASELINE configs[1] map built the way bench.py builds its 10 000-frame one: VLP-16 frames ray cast along
    the loop through the 600 x 600 m world, voxel-filtered and pushed at their ground-truth poses through the product's
    FeatureMap::addFeatureCloud (util/FeatureMap.h:219-230,289-306; corner leaf 0.2 m, surf leaf 0.4 m) -- 400 frames of the
    last 500 m of the loop -- and full 64 x 1800 scans (configs[2]) taken on that stretch."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    synth_gpu = importlib.import_module("synth_gpu")
    world = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
    lidar = synth_gpu.GpuLidar(world, 0)
    traj = synth_gpu.loop_trajectory(10000)[-4000::10]
    ctx = pkg.Context(0)
    fm, stats = synth_gpu.build_voxel_map(pkg, ctx, lidar, traj, rings=16)
    assert stats["frames"] == 400
    end = traj[-120]  # 150 m before the end of the mapped stretch: the surround is mapped on both sides
    fm.update(end[3:].astype(np.float32))
    surround = fm.get_surround_feature()
    fm.surround_to_map()
    rng = np.random.default_rng(31)
    scans, inits, gts = [], [], []
    for k in range(3):
        g = traj[-120 + int(rng.integers(-12, 12))].copy()
        g[3:5] += rng.uniform(-1.0, 1.0, 2)
        g[2] += rng.uniform(-0.2, 0.2)
        qc, qs = lidar.scan(g, 64, 1800, seed=7000 + k)
        scans.append((qc, qs))
        gts.append(g.astype(np.float32))
        inits.append(synth.perturb_pose(g, seed=60 + k))
    yield dict(ctx=ctx, fm=fm, surround=surround, scans=scans, inits=np.stack(inits), gts=np.stack(gts))
    fm.close()
    ctx.close()


@pytest.mark.parametrize("search", ["auto", "lane"])
def test_voxel_map_batch_of_full_scans_matches_oracle(voxel_map_problem, oracle, monkeypatch, search):
    """configs[1] map (reduced) x configs[2] scans: three full 64 x 1800 scans (~1 350 blocks) in one batch against the
    surround of the addFeatureCloud-built voxel map -- the bench's workload shape, kernel instantiation and code path --
    with every scan's pose, row counts and iteration count against the oracle on the same clouds.  `auto` is what bench.py
    runs: the library picks the grid sweep for a batch this size (sweep_grid_kernel + sweep_queue_kernel); `lane` is the
    kd-tree walk of every point (sweep_kernel<256,true,false,12> + the certificate pass)."""
    vp = voxel_map_problem
    ctx = vp["ctx"]
    mc, ms = vp["surround"]
    assert len(mc) > 20000 and len(ms) > 100000
    n_blocks = sum((len(c) + 255) // 256 + (len(s) + 255) // 256 for c, s in vp["scans"])
    assert n_blocks >= 600 and all(len(c) + len(s) > 100000 for c, s in vp["scans"])
    ctx.scan_set_batch(vp["scans"])
    opts = ctx.default_opts()
    opts.scans_in_flight = 3
    opts.search_mode = LANE if search == "lane" else 0
    before, g0 = ctx.sweep_launches(), ctx.grid_launches()
    worst, poses, stats = ctx.run_batch(vp["inits"], opts)
    ran = _ran(before, ctx.sweep_launches())
    if search == "lane":
        assert set(ran) == {"shallow"} and ctx.grid_launches() == g0, ran
    else:
        assert not ran and ctx.grid_launches() - g0 >= 3, (ran, ctx.grid_launches() - g0)  # picked by launch size, not forced
    for k, (qc, qs) in enumerate(vp["scans"]):
        ok, opose, ost = oracle.scanmatch_scan(mc, ms, qc, qs, vp["inits"][k])
        assert stats[k].status == ost.status and stats[k].converged == ost.converged == 1, k
        assert stats[k].iterations == ost.iterations, k
        assert _counts_close((stats[k].n_line, stats[k].n_plane, stats[k].n_rows), (ost.n_line, ost.n_plane, ost.n_rows)), k
        assert np.abs(poses[k][3:] - opose[3:]).max() <= POSE_TOL_M, k
        assert np.abs(poses[k][:3] - opose[:3]).max() <= POSE_TOL_RAD, k
        assert abs(stats[k].score - ost.score) <= 1e-4 * ost.score and abs(stats[k].percent - ost.percent) <= 1e-4
        # and it is a real match: within centimetres of where the scan was taken
        assert np.abs(poses[k][3:] - vp["gts"][k][3:]).max() < 0.05, k
    # a single full scan (deep stack, latency-bound launch) gives the batch's bits -- in the batch's sweep mode (certificates,
    # which a launch this small would not take by itself)
    forced = ctx.default_opts()
    forced.knn_cert = 2
    ctx.scan_set(*vp["scans"][1])
    if search == "lane":
        before = ctx.sweep_launches()
        status, pose1, st1 = ctx.run(vp["inits"][1], forced)
        assert set(_ran(before, ctx.sweep_launches())) == {"deep"}
        assert np.array_equal(bits(pose1), bits(poses[1])) and st1.iterations == stats[1].iterations
    else:  # the grid sweep of the scan alone (asked for: a launch this small is the lane search's by default): the batch's bits
        forced.search_mode = GRID
        g0 = ctx.grid_launches()
        status, pose1, st1 = ctx.run(vp["inits"][1], forced)
        assert ctx.grid_launches() > g0
        assert np.array_equal(bits(pose1), bits(poses[1])) and st1.iterations == stats[1].iterations


def test_voxel_map_sweep_taps_match_oracle_through_the_shallow_kernel(voxel_map_problem, oracle):
    """One unbounded sweep of a full 64 x 1800 scan over the voxel map through the shallow-stack kernel: neighbour
    indices, distances, flags and coefficients of all ~115 000 points bit for bit against the oracle."""
    vp = voxel_map_problem
    ctx = vp["ctx"]
    mc, ms = vp["surround"]
    qc, qs = vp["scans"][0]
    tc, ts = oracle.kdtree(mc), oracle.kdtree(ms)
    ctx.scan_set(qc, qs)
    before = ctx.sweep_launches()
    g = ctx.sweep(vp["inits"][0], jtj_mode=1, search_mode=LANE | SHALLOW)
    assert set(_ran(before, ctx.sweep_launches())) == {"shallow"}
    o = oracle.sweep(tc, ts, qc, qs, vp["inits"][0])
    assert np.array_equal(g["idx"], o["idx"]) and np.array_equal(bits(g["d2"]), bits(o["d2"]))
    assert np.array_equal(g["flags"], o["flags"]) and np.array_equal(bits(g["coeff"]), bits(o["coeff"]))
    scale = np.abs(o["sums"][:27]).max()
    assert np.abs(g["sums"][:27] - o["sums"][:27]).max() <= 2e-5 * scale
    assert g["sums"][27] == o["sums"][27] and g["sums"][28] == o["sums"][28]
    assert (o["flags"] & 4).sum() > 50000


def _every_sweep_of_the_loop_against_the_oracle(ctx, oracle, tc, ts, qc, qs, init, max_sweeps=8):
    """The grid sweep as the production loop runs it -- the first sweep bounded by the gate, every later one with its probes
    clipped to the bound carried from the sweep before and its second pass's tree searches started from it -- held to the
    oracle's sweep at the same pose, point by point: flags and coefficients of every point bit for bit, neighbour indices and
    squared distances bit for bit wherever the reference looks them up at all (d2[4] < 5, ScanMatch.cpp:102,120).  Sweep k + 1
    is the tap right after a run cut at max_iterations = k (include/lslam_c.h LSLAM_SWEEP_CARRIED).  Returns the number of
    sweeps compared and of points whose neighbours were."""
    ctx.scan_set(qc, qs)
    o = ctx.default_opts()
    o.search_mode = GRID
    n_sweeps = n_pts = 0
    pose = np.asarray(init, np.float32)
    for k in range(max_sweeps):
        if k > 0:
            o.max_iterations = k
            status, pose, st = ctx.run(init, o)
            if st.iterations < k or st.converged:  # the loop ended before sweep k + 1
                break
        g0 = ctx.grid_launches()
        g = ctx.sweep(pose, jtj_mode=1, search_mode=GRID | (CARRIED if k > 0 else FIRST))
        assert ctx.grid_launches() == g0 + 1
        r = oracle.sweep(tc, ts, qc, qs, pose)
        assert np.array_equal(g["flags"], r["flags"]), (k, np.argwhere(g["flags"] != r["flags"])[:5].tolist())
        assert np.array_equal(bits(g["coeff"]), bits(r["coeff"])), k
        looked_up = (r["flags"] & 1) != 0
        assert np.array_equal(g["idx"][looked_up], r["idx"][looked_up]), (k, np.argwhere((g["idx"] != r["idx"]).any(1) & looked_up)[:5].tolist())
        assert np.array_equal(bits(g["d2"])[looked_up], bits(r["d2"])[looked_up]), k
        assert g["sums"][27] == r["sums"][27] and g["sums"][28] == r["sums"][28]
        n_sweeps += 1
        n_pts += int(looked_up.sum())
    return n_sweeps, n_pts


def test_voxel_map_every_sweep_of_the_grid_loop_matches_oracle_index_for_index(voxel_map_problem, oracle):
    """What the headline times from a loop's second sweep on -- the BOUNDED probe (rows and cells clipped to the carried bound)
    and the bounded second pass -- against the oracle index for index, on two full 64 x 1800 scans over the (reduced) voxel map,
    every sweep of their loops."""
    vp = voxel_map_problem
    ctx = vp["ctx"]
    mc, ms = vp["surround"]
    tc, ts = oracle.kdtree(mc), oracle.kdtree(ms)
    for k in (0, 2):
        n_sweeps, n_pts = _every_sweep_of_the_loop_against_the_oracle(ctx, oracle, tc, ts, vp["scans"][k][0], vp["scans"][k][1], vp["inits"][k])
        assert n_sweeps >= 3 and n_pts > 2.5 * 100000, (k, n_sweeps, n_pts)


def test_certificate_sweep_equals_searching_every_point(voxel_map_problem, ctx, oracle, synth, small_problem, monkeypatch):
    """The certificate sweep (sweep_body: neighbour lists of points that hardly moved carried over by proof, the others
    searched in a second pass) against searching every point in every sweep: the same neighbours, hence the same statuses,
    iteration and row counts, and poses equal to the rounding of sums taken in another grouping -- on the bench's workload
    shape (full scans against the voxel map, picked by launch size) and on a small batch with a far and an empty scan
    (forced).  Certificates really are issued and second passes really run."""
    def both(c, inits, opts, force):
        out = {}
        if (opts.search_mode & 0xFF) == 0:
            opts.search_mode |= LANE  # the certificate sweep is the lane search's (AUTO takes the grid sweep for big batches)
        for mode in ("0", "2" if force else "1"):
            opts.knn_cert = int(mode)
            opts.debug_stats = 1
            s0 = c.cert_stats()
            worst, poses, stats = c.run_batch(inits, opts)
            s1 = c.cert_stats()
            out[mode != "0"] = (poses, stats, tuple(b - a for a, b in zip(s0, s1)))
        (p0, st0, d0), (p1, st1, d1) = out[False], out[True]
        assert d0 == (0, 0, 0)
        needy, tested, passes = d1
        assert passes >= 2 and tested > 0 and needy < tested, d1  # certificates were tested and some held
        for k in range(len(st0)):
            assert (st0[k].status, st0[k].iterations, st0[k].converged) == (st1[k].status, st1[k].iterations, st1[k].converged), k
            assert (st0[k].n_rows, st0[k].n_line, st0[k].n_plane) == (st1[k].n_rows, st1[k].n_line, st1[k].n_plane), k
            assert np.abs(p0[k][3:] - p1[k][3:]).max() <= 2e-6 and np.abs(p0[k][:3] - p1[k][:3]).max() <= 2e-7, k
            assert abs(st0[k].score - st1[k].score) <= 1e-5 * max(1.0, st0[k].score), k
        return 1.0 - needy / tested

    vp = voxel_map_problem
    vctx = vp["ctx"]
    vctx.scan_set_batch(vp["scans"])
    opts = vctx.default_opts()
    opts.scans_in_flight = 3
    held = both(vctx, vp["inits"], opts, force=False)
    assert held > 0.3, held  # the last sweeps of a loop move a scan by millimetres

    pr = small_problem
    world = pr["world"]
    scans, inits = [(pr["corner"], pr["surf"])], [pr["init_pose"]]
    for k in range(4):
        gt = (0.0, 0.01 * k, 0.2 + 0.3 * k, 2.0 - 1.5 * k, -1.0 + k, synth.SENSOR_HEIGHT)
        qc, qs, gt = synth.make_scan(world, 16, 450, gt_pose=gt, seed=500 + k)
        scans.append((qc, qs))
        inits.append(synth.perturb_pose(gt, seed=600 + k))
    inits[2][3] += 400.0
    empty = np.zeros((0, 4), np.float32)
    scans[3] = (empty, empty)
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set_batch(scans)
    for shape in (DEEP, SHALLOW):
        o = ctx.default_opts()
        o.search_mode = LANE | shape
        both(ctx, np.stack(inits), o, force=True)
    # the mapping node's settings (no score gate, looser abort thresholds)
    o = ctx.default_opts()
    o.delta_t_abort = o.delta_r_abort = 0.1
    o.use_score = 0
    both(ctx, np.stack(inits), o, force=True)
    # soundness does not hang on the thresholds that decide WHEN certificates are worth testing: with every scan testing
    # them from its second sweep on, however far it has just moved, the loop still finds the same neighbours
    eager = ctx.default_opts()
    eager.cert_try_m = eager.cert_track_m = 1e9
    both(ctx, np.stack(inits), eager, force=True)
    vctx.scan_set_batch(vp["scans"])
    opts.cert_try_m = opts.cert_track_m = 1e9
    both(vctx, vp["inits"], opts, force=False)


def test_the_bound_a_search_keeps_is_below_the_true_sixth_distance(ctx, small_problem, monkeypatch):
    """The soundness of a certificate rests on one number per point: the lower bound `lb6` of the squared distance of every
    map point outside the five neighbours, which knn5_search<TRACK> collects from the subtrees it does not enter and the
    candidates it turns away.  After loops that ran the certificate sweep, every resident point's bound is held against the
    truth -- the sixth-nearest map point of the position the bound was taken at (scipy's cKDTree on the same map): never
    above it.  And the bounds are worth something: most are within a few per cent of it."""
    from scipy.spatial import cKDTree
    pr = small_problem
    nc, ns = len(pr["corner"]), len(pr["surf"])
    trees = (cKDTree(pr["map_corner"][:, :3].astype(np.float64)), cKDTree(pr["map_surf"][:, :3].astype(np.float64)))
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    for eager, iters in ((None, 2), (None, 10), (1e9, 10)):  # after one tracked sweep; a whole loop; every sweep testing certificates
        o = ctx.default_opts()
        o.knn_cert = 2
        if eager:
            o.cert_try_m = o.cert_track_m = eager
        o.max_iterations = iters
        o.search_mode = LANE | SHALLOW
        status, pose, st = ctx.run(pr["init_pose"], o)
        q, lb = ctx.cert_state(nc + ns)
        assert len(lb) == nc + ns and (lb > 0).mean() > 0.9
        tight = []
        for tree, sl in ((trees[0], slice(0, nc)), (trees[1], slice(nc, nc + ns))):
            d, _ = tree.query(q[sl].astype(np.float64), k=6)
            d6sq = d[:, 5] ** 2
            have = lb[sl] > 0
            assert (lb[sl][have] <= d6sq[have] * (1 + 1e-5) + 1e-9).all(), float((lb[sl][have] / d6sq[have]).max())
            tight.append(lb[sl][have] / d6sq[have])
        assert np.median(np.concatenate(tight)) > 0.8


@pytest.mark.parametrize("search", ["auto", "lane"])
def test_voxel_map_batch_of_48_full_scans_properties(voxel_map_problem, pkg, synth, search):
    """Size-independent properties of the bench's code path at a size the oracle does not finish in seconds: 48 full 64 x 1800
    scans (21 600 blocks, 5.5 M points per sweep) in one batch against the voxel map, certificate sweep on (picked by size).
    (a) two runs give the same bits (no atomics in any sum); (b) a scan's result does not depend on what it is batched with:
    the first 24 alone == the first 24 of the 48, bit for bit, and chunked 16 at a time likewise; (c) against searching
    every point: same iteration and row counts for every scan, poses to rounding; (d) every scan converges to within 5 cm of
    where it was taken."""
    import importlib as il
    synth_gpu = il.import_module("synth_gpu")
    vp = voxel_map_problem
    ctx = vp["ctx"]
    world = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
    lidar = synth_gpu.GpuLidar(world, 0)
    traj = synth_gpu.loop_trajectory(10000)[-4000::10]
    rng = np.random.default_rng(77)
    scans, inits, gts = [], [], []
    for k in range(48):
        g = traj[-120 + int(rng.integers(-12, 12))].copy()
        g[3:5] += rng.uniform(-1.0, 1.0, 2)
        g[2] += rng.uniform(-0.2, 0.2)
        scans.append(lidar.scan(g, 64, 1800, seed=8000 + k))
        gts.append(g.astype(np.float32))
        inits.append(synth.perturb_pose(g, seed=160 + k))
    inits, gts = np.stack(inits), np.stack(gts)
    opts = ctx.default_opts()
    opts.scans_in_flight = 48
    opts.search_mode = LANE if search == "lane" else 0  # auto: the grid sweep (picked by size)
    ctx.scan_set_batch(scans)
    q0, g0 = ctx.cert_stats()[2], ctx.grid_launches()
    _, p1, s1 = ctx.run_batch(inits, opts)
    assert ctx.cert_stats()[2] > q0  # second passes ran (the certificate sweep's / the grid sweep's)
    assert (ctx.grid_launches() > g0) == (search == "auto")
    _, p2, s2 = ctx.run_batch(inits, opts)
    assert np.array_equal(bits(p1), bits(p2))                                                    # (a)
    assert all(s.converged for s in s1) and np.abs(p1[:, 3:] - gts[:, 3:]).max() < 0.05        # (d)
    assert len({s.iterations for s in s1}) > 1
    opts.scans_in_flight = 16                                                                    # (b) chunked
    _, p3, s3 = ctx.run_batch(inits, opts)
    assert np.array_equal(bits(p3), bits(p1)) and [s.iterations for s in s3] == [s.iterations for s in s1]
    opts.knn_cert = 0                                                                            # (c)
    opts.scans_in_flight = 48
    _, p0, s0 = ctx.run_batch(inits, opts)
    opts.knn_cert = 1
    for k in range(48):
        assert (s0[k].iterations, s0[k].n_rows, s0[k].n_line, s0[k].n_plane) == (s1[k].iterations, s1[k].n_rows, s1[k].n_line, s1[k].n_plane), k
    assert np.abs(p0[:, 3:] - p1[:, 3:]).max() <= 2e-6 and np.abs(p0[:, :3] - p1[:, :3]).max() <= 2e-7
    ctx.scan_set_batch(scans[:24])                                                               # (b) another batch
    opts.scans_in_flight = 24
    _, p4, s4 = ctx.run_batch(inits[:24], opts)
    assert np.array_equal(bits(p4), bits(p1[:24]))


def test_fused_solve_equals_the_solve_launch(ctx, synth, small_problem, monkeypatch):
    """The 6 x 6 solve in the tail of the sweep launch (the block that retires the scan's last record reduces and solves) against
    the solve kernel as its own launch: same reduction order, same solve -- same bits; single scan, a small batch with a scan
    that ends early, and the mapping node's settings.  (Measured no faster: lslam_opts.ab_switches & LSLAM_AB_FUSED_SOLVE is an A/B switch, off by default.)"""
    pr = small_problem
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    world = pr["world"]
    scans, inits = [(pr["corner"], pr["surf"])], [pr["init_pose"]]
    for k in range(3):
        gt = (0.0, 0.01 * k, 0.2 + 0.3 * k, 2.0 - 1.5 * k, -1.0 + k, synth.SENSOR_HEIGHT)
        qc, qs, gt = synth.make_scan(world, 16, 450, gt_pose=gt, seed=500 + k)
        scans.append((qc, qs))
        inits.append(synth.perturb_pose(gt, seed=600 + k))
    inits[2][3] += 400.0
    opts = ctx.default_opts()
    mopts = ctx.default_opts()
    mopts.delta_t_abort = mopts.delta_r_abort = 0.1
    mopts.use_score = 0
    res = {}
    # the fused tail rides on the one-launch sweep: the certificate sweep (two launches, the same terms summed in another
    # grouping) is held against it and the oracle by test_certificate_sweep_equals_searching_every_point
    opts.knn_cert = mopts.knn_cert = 0
    for fused in (True, False):
        opts.ab_switches = mopts.ab_switches = 2 if fused else 0  # LSLAM_AB_FUSED_SOLVE
        before = ctx.sweep_launches()
        ctx.scan_set(*scans[0])
        a = ctx.run(inits[0], opts)
        b = ctx.run(inits[0], mopts)
        ctx.scan_set_batch(scans)
        c = ctx.run_batch(np.stack(inits), opts)
        assert set(_ran(before, ctx.sweep_launches())) == {"deep_fused" if fused else "deep"}
        res[fused] = (a, b, c)
    for x, y in ((res[True][0], res[False][0]), (res[True][1], res[False][1])):
        assert np.array_equal(bits(x[1]), bits(y[1])) and x[2].iterations == y[2].iterations and x[2].n_rows == y[2].n_rows
        assert x[2].score == y[2].score and int(x[0]) == int(y[0]) and x[2].sweeps == y[2].sweeps
    (_, pf, sf), (_, pu, su) = res[True][2], res[False][2]
    assert np.array_equal(bits(pf), bits(pu))
    assert [(s.status, s.iterations, s.n_rows, s.sweeps) for s in sf] == [(s.status, s.iterations, s.n_rows, s.sweeps) for s in su]
    assert sf[2].status == 5 and len({s.iterations for s in sf}) > 1


def test_vlp16_mapping_frames_against_the_voxel_map_match_oracle(voxel_map_problem, pkg, oracle, synth):
    """BASELINE configs[1] at its own shape: VLP-16 sweeps of 16 x 1800 points through LaserMapping::process
    (LaserMapping.cpp:39-59 over LaserMatcher.cpp:289-354: transformMerge, VoxelGrid of the frame's features, update +
    surround -> trees, scanMatchScan with 0.1 / 0.1 and the score gate off, addFeatureCloud) against the (reduced) voxel map,
    three consecutive frames -- the map they match against includes what the earlier ones added -- with the device's chain
    held against the same steps made of oracle calls: every frame's map pose to the bar, the same iteration counts."""
    import importlib as il
    from test_gpu_pipeline import OracleChain
    synth_gpu = il.import_module("synth_gpu")
    vp = voxel_map_problem
    ctx = vp["ctx"]
    mc, ms = vp["surround"]
    dims = (21, 21, 11)
    world = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
    lidar = synth_gpu.GpuLidar(world, 0)
    traj = synth_gpu.loop_trajectory(10000)[-4000::10]
    mapper = pkg.LaserMapping(ctx, cube_dims=dims, map_filter_corner=0.2, map_filter_surf=0.4, map_filter=0.6)
    chain = OracleChain(oracle, ctx, dims)
    chain.fm.setup_filter_size(0.2, 0.4, 0.6)
    eye = np.eye(4, dtype=np.float32)
    start = traj[-124]
    for fm in (mapper.feature_map, chain.fm):  # both maps start as the voxel map's surround (already filtered: stays as it is)
        fm.update(start[3:].astype(np.float32))
        fm.add_feature_cloud(mc, ms, eye)
    sr = pkg.scan_registration
    iters = []
    for k in range(3):
        g = traj[-124 + 2 * k].copy()
        _, _, cloud, ranges = lidar.scan(g, 16, 1800, seed=9100 + k, full=True)
        assert len(cloud) > 20000
        f = sr.extract_features(ctx, cloud, ranges)
        of = oracle.extract_features(cloud, ranges)
        for key in ("less_sharp", "less_flat"):
            assert np.array_equal(bits(f[key]), bits(of[key])), (k, key)
        odom = synth_gpu.pose_matrix(synth.perturb_pose(g, seed=70 + k, dt=0.15, dr_deg=0.8))  # what the odometry node would hand over
        M_g = mapper.process(f["less_sharp"], f["less_flat"], odom)
        M_o = chain.mapping(of["less_sharp"], of["less_flat"], odom)
        iters.append(mapper.last_stats.iterations)
        assert mapper.last_stats.n_rows > 1000, (k, mapper.last_stats.n_rows)
        assert np.abs(M_g[:3, 3] - M_o[:3, 3]).max() <= POSE_TOL_M, (k, np.abs(M_g - M_o).max())
        assert np.abs(M_g[:3, :3] - M_o[:3, :3]).max() <= 2e-5, k
        # a real match: the frame lands where it was taken -- within centimetres in the plane; the height of a 16-ring frame
        # thinned to one point per cubic metre (LaserMatcher.cpp:289-301) is held by far fewer rows
        assert np.abs(M_g[:2, 3] - g[3:5]).max() < 0.05 and abs(M_g[2, 3] - g[5]) < 0.3, (k, M_g[:3, 3], g[3:])
    assert min(iters) >= 2
    mapper.feature_map.close()


@pytest.fixture(scope="module")
def full_map_problem(pkg, synth):
    """BASELINE configs[1]'s map at its FULL size -- 10 000 VLP-16 frames through the product's addFeatureCloud, the surround at
    the end of the loop (≈ 157 k corner + 587 k surf points): the map bench.py times against -- and sixteen full 64 x 1800
    scans (configs[2]) taken the way bench.py takes its 960."""
    import importlib as il
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    synth_gpu = il.import_module("synth_gpu")
    world = synth.World(half_extent=300.0, wall_half=295.0, pole_pitch=2.5)
    lidar = synth_gpu.GpuLidar(world, 0)
    traj = synth_gpu.loop_trajectory(10000)
    ctx = pkg.Context(0)
    fm, stats = synth_gpu.build_voxel_map(pkg, ctx, lidar, traj, rings=16)
    assert stats["frames"] == 10000
    fm.update(traj[-1][3:].astype(np.float32))
    mc, ms = fm.get_surround_feature()
    fm.surround_to_map()
    assert len(mc) > 100000 and len(ms) > 400000
    rng = np.random.default_rng(4242)
    dense = synth_gpu.loop_trajectory(100000)
    seg = np.linalg.norm(np.diff(dense[:, 3:5], axis=0), axis=1).mean()
    span = int(25.0 / seg)
    scans, inits = [], []
    for k in range(16):
        g = dense[int(rng.integers(-span, span)) % len(dense)].copy()
        g[3:5] += rng.uniform(-1.0, 1.0, 2)
        g[2] += rng.uniform(-0.2, 0.2)
        scans.append(lidar.scan(g, 64, 1800, seed=900000 + k))
        inits.append(synth.perturb_pose(g, seed=99 + k))
    yield dict(ctx=ctx, fm=fm, surround=(mc, ms), scans=scans, inits=np.stack(inits), lidar=lidar, traj=traj, synth_gpu=synth_gpu)
    fm.close()
    ctx.close()


def test_full_size_configs_map_grid_sweep_against_lane_and_oracle(full_map_problem, oracle):
    """configs[1]'s full-size map x sixteen configs[2] scans in one batch.  (a) the library's choice for a batch this size (the
    grid sweep) against the kd-tree walk of every point: the same iteration and row counts, poses to the rounding of
    differently grouped sums; (b) two of the scans against the oracle on the same clouds (the oracle needs ≈ 2 s per scan on
    this map); (c) the share of points the grid sweep left to the tree search is the few per cent DESIGN quotes, and the
    by-type / by-sweep breakdown of the tap adds up to it."""
    fp = full_map_problem
    ctx, scans, inits = fp["ctx"], fp["scans"], fp["inits"]
    mc, ms = fp["surround"]
    ctx.scan_set_batch(scans)
    o = ctx.default_opts()
    o.scans_in_flight = 16
    o.debug_stats = 1
    g0, s0, b0 = ctx.grid_launches(), ctx.cert_stats(), ctx.grid_stats().astype(np.int64)
    _, p_auto, st_auto = ctx.run_batch(inits, o)
    s1, b1 = ctx.cert_stats(), ctx.grid_stats().astype(np.int64)
    assert ctx.grid_launches() > g0                                                             # picked by size
    listed, swept = s1[0] - s0[0], s1[1] - s0[1]
    assert 0.002 < listed / swept < 0.08, listed / swept                                        # (c)
    by = b1 - b0
    assert by[:, :, 0].sum() == listed and by[:, :, 1].sum() == swept
    assert swept == sum(s.sweeps * (len(c) + len(q)) for s, (c, q) in zip(st_auto, scans))
    assert by[:, 0, 1].sum() == sum(len(c) + len(q) for c, q in scans)                           # every point is in a first sweep
    o.search_mode = LANE
    o.knn_cert = 0
    _, p_lane, st_lane = ctx.run_batch(inits, o)
    for a, b in zip(st_auto, st_lane):                                                          # (a)
        assert a.iterations == b.iterations and a.converged == b.converged
        assert _counts_close((a.n_line, a.n_plane, a.n_rows), (b.n_line, b.n_plane, b.n_rows))
    assert np.abs(p_auto[:, 3:] - p_lane[:, 3:]).max() <= 5e-6 and np.abs(p_auto[:, :3] - p_lane[:, :3]).max() <= 5e-7
    for k in (0, 7):                                                                            # (b)
        ok, opose, ost = oracle.scanmatch_scan(mc, ms, scans[k][0], scans[k][1], inits[k])
        assert st_auto[k].iterations == ost.iterations and st_auto[k].converged == ost.converged
        assert _counts_close((st_auto[k].n_line, st_auto[k].n_plane, st_auto[k].n_rows), (ost.n_line, ost.n_plane, ost.n_rows))
        assert np.abs(p_auto[k][3:] - opose[3:]).max() <= POSE_TOL_M and np.abs(p_auto[k][:3] - opose[:3]).max() <= POSE_TOL_RAD


def test_full_size_map_every_sweep_of_the_grid_loop_matches_oracle_index_for_index(full_map_problem, oracle):
    """The timed kernel's bounded probes at bench size: two full 64 x 1800 scans against the FULL-size configs[1] map, every
    sweep of their Gauss-Newton loops, the grid sweep (first sweep gate-bounded, later ones carried-bounded, both passes)
    against oracle.sweep at the same pose -- flags, coefficients, neighbour indices and distances bit for bit (≈ 115 k points
    x ≈ 4 sweeps x 2 scans)."""
    fp = full_map_problem
    ctx = fp["ctx"]
    mc, ms = fp["surround"]
    tc, ts = oracle.kdtree(mc), oracle.kdtree(ms)
    total = 0
    for k in (1, 9):
        n_sweeps, n_pts = _every_sweep_of_the_loop_against_the_oracle(ctx, oracle, tc, ts, fp["scans"][k][0], fp["scans"][k][1], fp["inits"][k])
        assert n_sweeps >= 3, (k, n_sweeps)
        total += n_pts
    assert total > 6 * 100000, total


def test_full_size_map_vlp16_mapping_frames_match_oracle(full_map_problem, pkg, oracle, synth):
    """BASELINE configs[1] at FULL size through LaserMapping::process: three consecutive 16 x 1800 frames at the end of the
    loop against the 10 000-frame map (the frames' own additions included), the device chain against the same steps made of
    oracle calls -- map poses to the bar, the same iteration counts."""
    from test_gpu_pipeline import OracleChain
    fp = full_map_problem
    synth_gpu, lidar, traj = fp["synth_gpu"], fp["lidar"], fp["traj"]
    mc, ms = fp["surround"]
    dims = (21, 21, 11)
    c2 = pkg.Context(0)
    try:
        mapper = pkg.LaserMapping(c2, cube_dims=dims, map_filter_corner=0.2, map_filter_surf=0.4, map_filter=0.6)
        chain = OracleChain(oracle, c2, dims)
        chain.fm.setup_filter_size(0.2, 0.4, 0.6)
        eye = np.eye(4, dtype=np.float32)
        start = traj[-1]
        for fm in (mapper.feature_map, chain.fm):  # both maps start as the full map's surround (already filtered: stays as it is)
            fm.update(start[3:].astype(np.float32))
            fm.add_feature_cloud(mc, ms, eye)
        sr = pkg.scan_registration
        iters = []
        for k in range(3):
            g = traj[-7 + 3 * k].copy()
            _, _, cloud, ranges = lidar.scan(g, 16, 1800, seed=9300 + k, full=True)
            f = sr.extract_features(c2, cloud, ranges)
            of = oracle.extract_features(cloud, ranges)
            for key in ("less_sharp", "less_flat"):
                assert np.array_equal(bits(f[key]), bits(of[key])), (k, key)
            odom = synth_gpu.pose_matrix(synth.perturb_pose(g, seed=170 + k, dt=0.15, dr_deg=0.8))
            M_g = mapper.process(f["less_sharp"], f["less_flat"], odom)
            M_o = chain.mapping(of["less_sharp"], of["less_flat"], odom)
            iters.append(mapper.last_stats.iterations)
            assert mapper.last_stats.n_rows > 1000, (k, mapper.last_stats.n_rows)
            assert np.abs(M_g[:3, 3] - M_o[:3, 3]).max() <= POSE_TOL_M, (k, np.abs(M_g - M_o).max())
            assert np.abs(M_g[:3, :3] - M_o[:3, :3]).max() <= 2e-5, k
            assert np.abs(M_g[:2, 3] - g[3:5]).max() < 0.05 and abs(M_g[2, 3] - g[5]) < 0.3, (k, M_g[:3, 3], g[3:])
        assert min(iters) >= 2
        mapper.feature_map.close()
    finally:
        c2.close()


def test_full_size_map_wide_probe_equals_tree_search(full_map_problem):
    """The search the mapping node's frames run on (a map without kd-trees: the wide probe over the cell grid) against nanoflann's
    traversal, query by query, on the FULL-size configs[1] map: the points of a 64 x 1800 scan at its initial (displaced) pose and at
    its matched pose, corner and surf.  Wherever the probe decides and the reference looks the five up (d2[4] < 5): nanoflann's
    indices and distances, bit for bit; undecided stays rare."""
    fp = full_map_problem
    ctx = fp["ctx"]
    qc, qs = fp["scans"][3]
    ctx.scan_set(qc, qs)
    o = ctx.default_opts()
    status, pose, st = ctx.run(fp["inits"][3], o)
    n_dec = n_und = 0
    for p in (fp["inits"][3], pose):
        T = ctx.pose_to_isometry(p)
        for which, cloud in ((0, qc), (1, qs[::4])):
            q = (cloud[:, :3] @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
            li, ld = ctx.knn5(which, q, search_mode=LANE)
            wi, wd, und = ctx.knn5_wide(which, q)
            looked_up = ld[:, 4] < 5.0
            dec = (und == 0) & looked_up
            assert np.array_equal(wi[dec], li[dec]) and np.array_equal(bits(wd[dec]), bits(ld[dec])), which
            n_dec += int(dec.sum())
            n_und += int(((und != 0) & looked_up).sum())
    assert n_dec > 60000 and n_und < 1e-3 * n_dec, (n_dec, n_und)
