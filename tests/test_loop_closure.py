"""Loop-closure front end (pose_graph/loop_detector.hpp): the candidate gating on the CPU, a full
detect -> scanMatchLocal -> Loop on the GPU."""
import numpy as np
import pytest


def _kf(pkg, x, z, accum, y=0.0):
    T = np.eye(4)
    T[:3, 3] = (x, y, z)
    return pkg.KeyFrame(T, accum, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))


def test_candidate_gating_follows_the_reference(pkg):
    det = pkg.LoopDetector()
    # a straight 40 m run, then back over the start: keyframe spacing 1 m of travel
    kfs = [_kf(pkg, float(i), 0.0, float(i)) for i in range(41)]
    kfs += [_kf(pkg, 40.0 - 0.5 * i, 1.0, 40.0 + 2.0 * i, y=7.0 * i) for i in range(1, 40)]
    det.update_trajectory(kfs)
    new = _kf(pkg, 2.0, 1.0, 200.0, y=3.0)
    det.single_result_quirk = False
    idx, d2 = det.radius_search(np.array([2.0, 0.0, 1.0], np.float32), 5.0)
    # "radius" 5.0 is compared with squared distances: only points within sqrt(5) m, y ignored
    assert len(idx) > 1 and (d2 < 5.0).all() and np.all(np.diff(d2) >= 0)
    det.single_result_quirk = True  # the reference's wrapper hands back the nearest one only
    idx1, d21 = det.radius_search(np.array([2.0, 0.0, 1.0], np.float32), 5.0)
    assert len(idx1) == 1 and idx1[0] == idx[0]
    assert all(abs(kfs[i].estimate[0, 3] - 2.0) <= np.sqrt(5.0) + 1e-6 for i in idx)
    cand = det.find_nearest_candidates(kfs, new)
    # at most 6, all travelled >= 30 m before `new`, within 5 m of travel of the first candidate
    assert 0 < len(cand) <= 6
    assert all(new.accum_distance - c.accum_distance >= 30.0 for c in cand)
    assert all(abs(c.accum_distance - cand[0].accum_distance) <= 5.0 for c in cand)
    # too little travel since the last loop: no candidates at all
    det.last_loop_accum_distance = 198.5
    assert det.find_nearest_candidates(kfs, new) == []
    det.last_loop_accum_distance = 0.0
    # keyframes travelled less than 30 m ago are skipped
    recent = _kf(pkg, 2.0, 1.0, 20.0)
    assert det.find_nearest_candidates(kfs, recent) == []
    # nothing within sqrt(5) m
    far = _kf(pkg, 500.0, 0.0, 500.0)
    assert det.find_nearest_candidates(kfs, far) == []


@pytest.mark.gpu
def test_detect_nearest_closes_a_loop(pkg, ctx, synth, small_problem):
    """Two candidate keyframes around a place, a new keyframe revisiting it with a drifted estimate:
    the detector gates them in, merges the candidates' clouds, aligns with scanMatchLocal and returns
    the relative pose of the revisit -- close to the ground truth, far closer than the drifted guess."""
    world = small_problem["world"]
    def frame(pose6, seed, accum, drift=None):
        c, s, gt = synth.make_scan(world, 16, 900, gt_pose=pose6, seed=seed)
        R, t = synth.pose_to_Rt(gt)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        est = T.copy()
        if drift is not None:
            est[:3, 3] += drift
        return pkg.KeyFrame(est, accum, c, s), T
    k0, T0 = frame((0, 0, 0.30, 3.0, -2.0, synth.SENSOR_HEIGHT), 11, 0.0)
    k1, T1 = frame((0, 0, 0.32, 3.8, -2.2, synth.SENSOR_HEIGHT), 12, 1.0)
    filler = [_kf(pkg, 50.0 + i, 0.0, 2.0 + i) for i in range(3)]
    new, Tn = frame((0, 0, 0.35, 3.4, -1.8, synth.SENSOR_HEIGHT), 13, 60.0, drift=np.array([0.25, -0.2, 0.0]))
    det = pkg.LoopDetector(ctx=ctx)
    loops = det.detect_nearest([k0, k1] + filler, [new])
    assert len(loops) == 1 and det.get_loop_count() == 1
    lp = loops[0]
    assert lp.key1 in (k0, k1) and lp.key2 is new
    T_ref = {id(k0): T0, id(k1): T1}[id(lp.key1)]
    truth = np.linalg.inv(T_ref) @ Tn
    guess = np.linalg.inv(lp.key1.estimate) @ new.estimate
    err = np.abs(lp.relative_pose[:3, 3] - truth[:3, 3]).max()
    assert err < 0.05 and err < 0.3 * np.abs(guess[:3, 3] - truth[:3, 3]).max()
    assert det.last_loop_accum_distance == 60.0


def test_keyframe_updater_thresholds(pkg):
    """keyframe_updater.hpp:21-44: a new keyframe needs 0.25 m or 0.05 rad; travel accumulates."""
    ku = pkg.KeyframeUpdater()
    T = np.eye(4)
    assert ku.update(T) and ku.get_accum_distance() == 0.0
    T1 = T.copy(); T1[0, 3] = 0.2
    assert not ku.update(T1)
    T2 = T.copy(); T2[0, 3] = 0.3
    assert ku.update(T2) and abs(ku.get_accum_distance() - 0.3) < 1e-12
    c, s = np.cos(0.06), np.sin(0.06)
    T3 = T2.copy(); T3[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    assert ku.update(T3) and abs(ku.get_accum_distance() - 0.3) < 1e-12  # rotation only: no travel added
    assert ku.get_unique_id() == 1 and ku.get_unique_id() == 2


@pytest.mark.gpu
def test_graph_closes_the_loop_end_to_end(pkg, ctx, synth, small_problem):
    """Config 4 end to end at toy size: drive a 40 m loop twice with drifting odometry, let the graph
    pick keyframes, add odometry edges, find the loop on the second lap (radius search -> gating ->
    scanMatchLocal on the device), add the loop edge and optimise: the trajectory error drops."""
    world = small_problem["world"]
    rng = np.random.default_rng(5)
    # The detector flattens the trajectory with y = 0 (loop_detector.hpp:98,120): the reference lives
    # in LOAM's camera-style frame (y up).  The synthetic world is z-up, so poses and clouds are
    # expressed in that frame through the fixed permutation (x, y, z)_loam = (y, z, x)_world.
    P = np.array([[0, 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float64)
    def to_loam_cloud(c):
        o = c.copy()
        o[:, :3] = c[:, [1, 2, 0]]
        return o
    g = pkg.Graph(ctx=ctx)
    g.loop_detector.accum_distance_thresh = 25.0
    # square path of side 10 m around (5, 5), 1 m steps, two laps, sensor yaw fixed
    way = []
    for lap in range(2):
        for side, (dx, dy) in enumerate(((1, 0), (0, 1), (-1, 0), (0, -1))):
            for k in range(10):
                way.append((dx, dy))
    x, y = 0.0, 0.0
    drift = np.zeros(2)
    gts, odoms = [], []
    n_loops = 0
    for step, (dx, dy) in enumerate([(0, 0)] + way):
        x += dx; y += dy
        drift += rng.normal(0, 0.01, 2) + np.array([0.004, -0.003])
        gt = (0.0, 0.0, 0.3, x, y, synth.SENSOR_HEIGHT)
        c, s, gtp = synth.make_scan(world, 16, 450, gt_pose=gt, seed=1000 + step)
        R, t = synth.pose_to_Rt(gtp)
        T = np.eye(4); T[:3, :3], T[:3, 3] = R, t
        O = T.copy(); O[:2, 3] += drift
        T, O = P @ T @ P.T, P @ O @ P.T
        kf = g.add_frame(O, to_loam_cloud(c), to_loam_cloud(s))
        assert kf is not None
        gts.append(T); odoms.append(O)
        loops, its = g.optimize(20)
        n_loops += len(loops)
    assert len(g.keyframes) == len(gts) and n_loops >= 1
    est = np.array([k.estimate[:3, 3] for k in g.keyframes])
    gt_xyz = np.array([T[:3, 3] for T in gts])
    od_xyz = np.array([O[:3, 3] for O in odoms])
    err_est = np.linalg.norm(est - gt_xyz, axis=1)
    err_odo = np.linalg.norm(od_xyz - gt_xyz, axis=1)
    assert err_est[-1] < 0.6 * err_odo[-1] and err_est.mean() < err_odo.mean() and err_est.max() < 0.5


def test_cpp_loop_closure_mirrors_compile(pkg, tmp_path):
    """include/lslam_loop_closure.hpp (LoopDetector / KeyframeUpdater / Graph over the C ABI) builds with g++ -std=c++11
    -Wall -Werror; without a GPU the program reports the missing backend."""
    import subprocess
    import torch
    from test_abi import _build_cpp
    exe = _build_cpp(pkg, tmp_path, "loop_closure_end_to_end")
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked run of the same program")
    (tmp_path / "none.bin").write_bytes(b"")
    out = subprocess.run([str(exe), str(tmp_path / "none.bin"), "25"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "no CPU fallback" in out.stderr


@pytest.mark.gpu
def test_cpp_graph_equals_python_graph(pkg, ctx, synth, small_problem, tmp_path):
    """The two-lap drive of test_graph_closes_the_loop_end_to_end through the C++ Graph / LoopDetector mirrors and the
    Python ones: same loops, same keyframe estimates (both issue the same C-ABI calls)."""
    import subprocess
    from test_abi import _build_cpp
    exe = _build_cpp(pkg, tmp_path, "loop_closure_end_to_end")
    world = small_problem["world"]
    rng = np.random.default_rng(5)
    P = np.array([[0, 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float64)
    g = pkg.Graph(ctx=ctx)
    g.loop_detector.accum_distance_thresh = 25.0
    way = [(dx, dy) for lap in range(2) for (dx, dy) in ((1, 0), (0, 1), (-1, 0), (0, -1)) for k in range(10)]
    x = y = 0.0
    drift = np.zeros(2)
    n_loops = 0
    path = tmp_path / "frames.bin"
    with open(path, "wb") as fo:
        for step, (dx, dy) in enumerate([(0, 0)] + way):
            x += dx; y += dy
            drift += rng.normal(0, 0.01, 2) + np.array([0.004, -0.003])
            c, s, gtp = synth.make_scan(world, 16, 450, gt_pose=(0.0, 0.0, 0.3, x, y, synth.SENSOR_HEIGHT), seed=1000 + step)
            R, t = synth.pose_to_Rt(gtp)
            O = np.eye(4); O[:3, :3], O[:3, 3] = R, t
            O[:2, 3] += drift
            O = P @ O @ P.T
            cl, sl = c.copy(), s.copy()
            cl[:, :3], sl[:, :3] = c[:, [1, 2, 0]], s[:, [1, 2, 0]]
            fo.write(np.ascontiguousarray(O, np.float64).tobytes())
            for a in (cl, sl):
                fo.write(np.uint32(len(a)).tobytes())
                fo.write(np.ascontiguousarray(a, np.float32).tobytes())
            assert g.add_frame(O, cl, sl) is not None
            loops, its = g.optimize(20)
            n_loops += len(loops)
    for d in ("graph2_cpp", "graph2_py"):  # (saveCloudToFiles writes into an existing directory, FeatureMap.h:378-410)
        (tmp_path / d).mkdir()
    out = subprocess.run([str(exe), str(path), "25", str(tmp_path / "graph2_cpp")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    head = [l for l in out.stdout.splitlines() if l.startswith("LOOPS")][0].split()
    assert int(head[1]) == n_loops >= 1 and int(head[5]) == len(g.keyframes)
    kf = np.array([[float(v) for v in l.split()[2:5]] for l in out.stdout.splitlines() if l.startswith("KF ")])
    est = np.array([k.estimate[:3, 3] for k in g.keyframes])
    assert np.abs(kf - est).max() < 1e-6
    # Graph::getFinalFeatureMap through both mirrors: the same ABI calls in the same order
    res = g.get_final_feature_map(ctx, directory=str(tmp_path / "graph2_py"), bootstrap=True)
    res["map"].close()
    fin = [l.split() for l in out.stdout.splitlines() if l.startswith("FINAL")][0]
    assert int(fin[1]) == res["added"] and int(fin[2]) == sum(res["matched"]) and res["added"] >= len(g.keyframes) // 2
    fp = [l.split() for l in out.stdout.splitlines() if l.startswith("FP ")]
    assert [bool(int(w[2])) for w in fp] == res["matched"]
    got = np.array([[float(v) for v in w[3:6]] for w in fp])
    # (the two graphs' estimates agree to 1e-6, not to the bit, and every match is made against a map built from the poses
    # before it: the chains' differences compound over the keyframes -- 3e-4 m after 80)
    assert np.abs(got - np.array([p[:3, 3] for p in res["poses"]])).max() <= 1e-3
    files = [sorted(l.split()[1:5] for l in (tmp_path / d / "index.txt").read_text().splitlines()) for d in ("graph2_cpp", "graph2_py")]
    assert files[0] == files[1] and len(files[0]) >= 2  # the same cubes of the same types


def test_trajectory_radius_search_matches_reference_nanoflann(pkg):
    """LoopDetector.radius_search against the reference's own KdTreeFLANN::radiusSearch
    (oracle/_ref/libref_nanoflann.so, nanoflann_pcl.h:164-186): same indices in the same order, same
    squared distances, on a looping trajectory flattened to y = 0 as loop_detector.hpp:86-96 does."""
    from oracle_lib import RefNanoflann, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built (reference absent)")
    rng = np.random.default_rng(3)
    s = np.linspace(0, 6 * np.pi, 900)
    xz = np.stack([40 * np.cos(s) + rng.normal(0, 0.3, len(s)), 40 * np.sin(s) + rng.normal(0, 0.3, len(s))], 1)
    kfs = [_kf(pkg, float(x), float(z), float(i), y=float(rng.normal(0, 2))) for i, (x, z) in enumerate(xz)]
    det = pkg.LoopDetector()
    det.update_trajectory(kfs)
    ref = RefNanoflann(det._trajectory)  # x, 0, z, index: the cloud the reference builds its tree on
    hits = 0
    for k in range(0, len(kfs), 7):
        q = det._trajectory[k, :3] + rng.normal(0, 0.5, 3).astype(np.float32)
        q[1] = 0.0
        for radius in (5.0, 1.0, 30.0):
            # the search itself: every point within the radius (compared with squared distances), ascending
            det.single_result_quirk = False
            idx, d2 = det.radius_search(q, radius)
            ridx, rd2 = ref.radius(q, radius)
            assert np.array_equal(idx, ridx) and np.array_equal(d2.view(np.uint32), rd2.view(np.uint32))
            hits += len(idx)
            # what KdTreeFLANN::radiusSearch hands to its caller: one result (nFound is a bool)
            det.single_result_quirk = True
            idx1, d21 = det.radius_search(q, radius)
            n_ref = ref.radius_as_wrapped(q, radius)
            if len(ridx):
                assert n_ref == 1 and np.array_equal(idx1, ridx[:1]) and np.array_equal(d21, rd2[:1])
            else:
                assert len(idx1) == 0  # the reference reports 1 and reads an empty vector here (UB)
    assert hits > 1000


@pytest.mark.gpu
def test_final_feature_map_matches_the_oracle_chain(pkg, ctx, oracle, synth, small_problem, tmp_path):
    """Graph::getFinalFeatureMap (pose_graph/graph.cpp:150-199): 40 keyframes rebuilt into a map one after the other -- update,
    surround, VoxelGrid 0.2 / 0.3, scanMatchScan from the keyframe's estimate, addFeatureCloud iff matched -- on the device
    against the same chain made of oracle calls: same matched flags, refined poses to the bar, maps of the same size; and the
    reference's quirk: without the bootstrap the first match has nothing to match against, nothing is ever added, the map
    stays empty."""
    world = small_problem["world"]
    rng = np.random.default_rng(3)
    kfs = []
    for k in range(40):
        gt = (0.0, 0.0, 0.3 + 0.004 * k, 3.0 + 0.3 * k, -2.0 + 0.1 * k, synth.SENSOR_HEIGHT)
        c, s, _ = synth.make_scan(world, 16, 600, gt_pose=gt, seed=700 + k)
        est = ctx.pose_to_isometry(np.array(gt, np.float32)).astype(np.float64)
        est[:3, 3] += rng.uniform(-0.05, 0.05, 3)  # the optimised graph's estimate: near the truth, not on it
        kfs.append(pkg.KeyFrame(est, 0.3 * k, c, s, frame_id=k))
    g = pkg.Graph(ctx=ctx)
    # the reference as written
    res0 = g.get_final_feature_map(ctx, cube_dims=(21, 11, 21), keyframes=kfs[:5])
    assert res0["added"] == 0 and not any(res0["matched"]) and res0["map"].info()["n_surf"] == 0
    res0["map"].close()
    (tmp_path / "graph2").mkdir()
    res = g.get_final_feature_map(ctx, directory=str(tmp_path / "graph2"), cube_dims=(21, 11, 21), bootstrap=True, keyframes=kfs)
    # the same chain, oracle calls only
    ofm = oracle.feature_map(21, 11, 21)
    ofm.setup_filter_size(0.2, 0.2, 0.4)
    o_matched, o_poses = [], []
    for kf in kfs:
        est = kf.estimate.astype(np.float32)
        ofm.update(est[:3, 3])
        mc, ms = ofm.get_surround_feature()
        cc, cs = oracle.voxel_grid(kf.corner_cloud, 0.2), oracle.voxel_grid(kf.surf_cloud, 0.3)
        ok = False
        big = len(mc) >= 50 and len(ms) >= 100
        if big:
            ok, pose, st = oracle.scanmatch_scan(mc, ms, cc, cs, ctx.isometry_to_pose(est), oracle.default_opts())
            if st.status != 1:
                est = ctx.pose_to_isometry(pose)
        if ok or not big:
            ofm.add_feature_cloud(kf.corner_cloud, kf.surf_cloud, est)
        o_matched.append(bool(ok))
        o_poses.append(np.array(est, np.float32))
    assert res["matched"] == o_matched and sum(o_matched) >= 30
    for k, (a, b) in enumerate(zip(res["poses"], o_poses)):
        # (a chain of 40 matches, each against a map made of the poses before it: the two chains' last-digit differences compound)
        assert np.abs(a[:3, 3] - b[:3, 3]).max() <= 1e-3 and np.abs(a[:3, :3] - b[:3, :3]).max() <= 1e-4, k
    n_dev, n_orc = len(res["map"].get_full_map()), len(ofm.get_full_map())
    assert abs(n_dev - n_orc) <= max(3, n_orc // 500)  # (poses differ in their last digits: a centroid may cross a voxel wall)
    assert (tmp_path / "graph2" / "index.txt").exists()
    res["map"].close()
