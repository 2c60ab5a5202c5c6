"""The whole per-sweep chain on the device against the same chain made of oracle calls:
MultiScanRegistration::process -> extractFeatures -> LaserOdometry::process -> LaserMapping::process
(SURVEY 8f n1/n2 around the hot path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# the north-star bar (pose within 1e-4 m of the CPU reference) also for the five-sweep chain: two
# Gauss-Newton loops per sweep, the odometry's per-point de-skew with the device's sin/cos
CHAIN_TOL = 1e-4


def test_transform_to_end_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(2)
    c = rng.uniform(-40, 40, (5000, 4)).astype(np.float32)
    c[:, 3] = rng.integers(0, 16, len(c)) + rng.uniform(0, 0.0999, len(c)).astype(np.float32)
    pose = np.array([0.01, -0.02, 0.03, 0.4, -0.1, 0.05], np.float32)
    got, ref = ctx.transform_to_end(c, pose), oracle.transform_to_end(c, pose)
    assert np.array_equal(got[:, 3], c[:, 3])
    assert np.abs(got[:, :3] - ref[:, :3]).max() <= 2e-5  # per-point sin/cos are the device's
    # the sweep-end point of a zero transform is the point itself
    assert np.array_equal(bits(ctx.transform_to_end(c, np.zeros(6, np.float32))[:, :3]), bits(c[:, :3]))


class OracleChain:
    """LaserOdometry::process + LaserMapping::process written with oracle calls only."""

    def __init__(self, oracle, ctx_for_conversions, cube_dims):
        self.o, self.cv = oracle, ctx_for_conversions  # Twist<->Isometry conversions are host helpers of the ABI
        self.transform = np.zeros(6, np.float32)
        self.Tsum = np.eye(4, dtype=np.float32)
        self.inited = False
        self.fm = oracle.feature_map(*cube_dims)
        self.fm.setup_filter_size(1.0, 1.0, 2.0)
        self.odom_last = np.eye(4, dtype=np.float32)
        self.mapped_last = np.eye(4, dtype=np.float32)

    def odometry(self, f):
        if not self.inited:
            self.tree_c, self.tree_s = f["less_sharp"], f["less_flat"]
            self.inited = True
            return None
        it, pose, st = self.o.odometry_match(self.tree_c, self.tree_s, f["sharp"], f["flat"], self.transform)
        self.transform = pose
        self.Tsum = (self.Tsum @ self.cv.pose_to_isometry(pose)).astype(np.float32)
        ls, lf = self.o.transform_to_end(f["less_sharp"], pose), self.o.transform_to_end(f["less_flat"], pose)
        self.last_c, self.last_s = ls, lf
        if len(ls) > 10 and len(lf) > 100:
            self.tree_c, self.tree_s = ls, lf
        return self.Tsum.copy()

    def mapping(self, corner_last, surf_last, odom_new):
        new = (self.mapped_last @ np.linalg.inv(self.odom_last) @ odom_new).astype(np.float32)
        cds, sds = self.o.voxel_grid(corner_last, 1.0), self.o.voxel_grid(surf_last, 1.0)
        self.fm.update(new[:3, 3])
        mc, ms = self.fm.get_surround_feature()
        if len(mc) or len(ms):
            opts = self.o.default_opts()
            opts.delta_t_abort = opts.delta_r_abort = 0.1
            opts.use_score = 0
            ok, pose, st = self.o.scanmatch_scan(mc, ms, cds, sds, self.cv.isometry_to_pose(new), opts)
            if st.status != 1:
                new = self.cv.pose_to_isometry(pose)
        self.mapped_last, self.odom_last = new.copy(), odom_new.copy()
        self.fm.add_feature_cloud(cds, sds, new)
        return new


def test_chain_through_a_velocity_reversal(pkg, ctx, oracle, synth, small_problem):
    """The same chain over sweeps that go out and come back (0 1 2 3 2 1 0): at the turn the odometry's
    initial guess (the previous sweep's motion, LaserOdometry.cpp:288-326) points the wrong way by 0.8 m,
    so its Gauss-Newton loop starts far from the answer and runs long.  Device and oracle chains must stay
    together through that as well (the paths a smooth drive never takes)."""
    world = small_problem["world"]
    dims = (21, 21, 11)
    odo = pkg.LaserOdometry(ctx)
    mapper = pkg.LaserMapping(ctx, cube_dims=dims)
    chain = OracleChain(oracle, ctx, dims)
    sr = pkg.scan_registration
    iters = []
    for step, k in enumerate((0, 1, 2, 3, 2, 1, 0)):
        gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
        c, s, gtp, cloud, ranges = synth.make_scan(world, 16, 900, gt_pose=gt, seed=300 + k, full=True)
        ring = np.floor(cloud[:, 3]).astype(np.int64)
        raw = cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))]
        reg, rr = sr.multiscan_register(ctx, raw, -15.0, 15.0, 16)
        f = sr.extract_features(ctx, reg, rr)
        of = oracle.extract_features(reg, rr)
        T_g, T_o = odo.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"]), chain.odometry(of)
        if step == 0:
            continue
        iters.append(odo.last_stats.iterations)
        assert np.abs(T_g - T_o).max() <= 2e-3, (step, np.abs(T_g - T_o).max())
        M_g = mapper.process(odo.last_corner, odo.last_surf, T_g)
        M_o = chain.mapping(chain.last_c, chain.last_s, T_o)
        assert np.abs(M_g - M_o).max() <= 2e-3, (step, np.abs(M_g - M_o).max())
    assert max(iters) == 25  # some loops run to the iteration limit (LaserOdometry.cpp: 25) without converging
    mapper.feature_map.close()


def test_registration_to_mapping_chain(pkg, ctx, oracle, synth, small_problem):
    """Five consecutive VLP-16 sweeps of a drive through the synthetic world, raw driver clouds in:
    the device chain and the oracle chain agree on every intermediate product (feature clouds bit for
    bit) and on the odometry and map poses to 1e-4 m (the chain passes through per-point sin/cos and
    two Gauss-Newton loops per sweep; each stage's own parity bar is tighter)."""
    from test_oracle_features import _raw_sweep  # noqa: F401  (same raw-sweep construction)
    world = small_problem["world"]
    dims = (21, 21, 11)
    odo = pkg.LaserOdometry(ctx)
    mapper = pkg.LaserMapping(ctx, cube_dims=dims)
    chain = OracleChain(oracle, ctx, dims)
    sr = pkg.scan_registration
    worst = 0.0
    for k in range(5):
        gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
        c, s, gtp, cloud, ranges = synth.make_scan(world, 16, 900, gt_pose=gt, seed=300 + k, full=True)
        ring = np.floor(cloud[:, 3]).astype(np.int64)
        raw = cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))]
        reg, rr = sr.multiscan_register(ctx, raw, -15.0, 15.0, 16)
        oreg, orr = oracle.multiscan_register(raw, -15.0, 15.0, 16)
        assert np.array_equal(rr, orr) and np.array_equal(bits(reg[:, :3]), bits(oreg[:, :3]))
        # from here on both chains consume the DEVICE registration (its ring + relTime differs from the
        # oracle's in the last bits), so that every later stage is compared on identical input
        f = sr.extract_features(ctx, reg, rr)
        of = oracle.extract_features(reg, rr)
        for key in ("sharp", "less_sharp", "flat", "less_flat"):
            assert np.array_equal(bits(f[key]), bits(of[key])), (k, key)
        T_g, T_o = odo.process(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"]), chain.odometry(of)
        if k == 0:
            assert T_g is None and T_o is None
            continue
        assert np.abs(T_g - T_o).max() <= CHAIN_TOL, (k, np.abs(T_g - T_o).max())
        M_g = mapper.process(odo.last_corner, odo.last_surf, T_g)
        M_o = chain.mapping(chain.last_c, chain.last_s, T_o)
        assert np.abs(M_g - M_o).max() <= CHAIN_TOL, (k, np.abs(M_g - M_o).max())
        worst = max(worst, float(np.abs(T_g - T_o).max()), float(np.abs(M_g - M_o).max()))
    # the sensor moved 1.6 m / 0.6 m between the first and the last sweep: the map pose (relative to the
    # first sweep) has travelled that far
    assert abs(np.linalg.norm(M_g[:3, 3]) - np.hypot(1.6, 0.6)) < 0.1
    print("chain worst |device - oracle| = %.3g (bar %.0e)" % (worst, CHAIN_TOL))
    mapper.feature_map.close()


def test_laser_mapping_default_grid(pkg, ctx, oracle, synth, small_problem):
    """The mapping node's default cube grid (121 x 121 x 11 cubes of 50 m, LaserMatcher.cpp:107-109):
    161 051 cubes on the device, same surround as the oracle after two frames."""
    world = small_problem["world"]
    mapper = pkg.LaserMapping(ctx)
    ofm = oracle.feature_map(121, 121, 11)
    ofm.setup_filter_size(1.0, 1.0, 2.0)
    T = np.eye(4, dtype=np.float32)
    for k in range(2):
        c, s, gt = synth.make_scan(world, 16, 450, gt_pose=(0, 0, 0.3, 3.0 + k, -2.0, synth.SENSOR_HEIGHT), seed=40 + k)
        M = mapper.process(c, s, T)  # identity odometry: the match has to find the metre of motion itself
        cds, sds = oracle.voxel_grid(c, 1.0), oracle.voxel_grid(s, 1.0)
        ofm.update(M[:3, 3])
        ofm.add_feature_cloud(cds, sds, M)
    gc, gs = mapper.feature_map.get_surround_feature()
    oc, os_ = ofm.get_surround_feature()
    assert mapper.feature_map.dims == (121, 121, 11) and len(gs) > 1000
    assert np.array_equal(bits(gc), bits(oc)) and np.array_equal(bits(gs), bits(os_))
    mapper.feature_map.close()


def test_contexts_on_concurrent_threads_equal_serial_runs(pkg, synth, small_problem):
    """The reference runs registration, odometry and mapping as nodelets in their own threads; here: three
    host threads, a context each, hammering feature extraction, VoxelGrid, a kd-tree build + scan match and
    a feature-map insert at the same time.  Every result must equal, bit for bit, what the same call gives
    on its own (the helpers that cache device scratch per device serialise on their locks; the contexts
    share nothing else)."""
    import threading
    pr = small_problem
    _, _, gt, cloud, ranges = synth.make_scan(pr["world"], 16, 900, gt_pose=pr["gt_pose"], seed=11, full=True)

    def work(ctx, kind):
        if kind == 0:
            f = pkg.scan_registration.extract_features(ctx, cloud, ranges)
            return [f[k] for k in ("sharp", "less_sharp", "flat", "less_flat")]
        if kind == 1:
            return [pkg.voxel_grid(ctx, pr["map_surf"], 1.0), pkg.voxel_grid(ctx, pr["map_corner"], 0.7)]
        ctx.map_set(pr["map_corner"], pr["map_surf"])
        s, pose, st = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"])
        fm = pkg.FeatureMap(ctx, 21, 11, 21)
        fm.update(pr["gt_pose"][3:])
        fm.add_feature_cloud(pr["corner"], pr["surf"], np.eye(4, dtype=np.float32))
        c, sf = fm.get_surround_feature()
        fm.close()
        return [pose, c, sf]

    serial_ctx = pkg.Context(0)
    ref = [work(serial_ctx, k) for k in range(3)]
    serial_ctx.close()
    out, err = {}, []

    def run(kind):
        try:
            c = pkg.Context(0)
            for rep in range(6):
                out[(kind, rep)] = work(c, kind)
            c.close()
        except Exception as e:  # pragma: no cover
            err.append(e)

    th = [threading.Thread(target=run, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for (kind, rep), got in out.items():
        assert len(got) == len(ref[kind])
        for a, b in zip(got, ref[kind]):
            assert a.shape == b.shape and np.array_equal(bits(a), bits(b)), (kind, rep)


def test_three_nodes_on_three_threads_equal_the_sequential_chain(pkg, synth, small_problem, tmp_path):
    """tools/cpp/node_threads.cpp: registration, odometry and mapping as three std::threads with a context each on ONE device,
    against the same calls from one thread.  The nodes share nothing but the device -- which is what this test is for: the
    library's per-device scratch caches (VoxelGrid, registration, extraction) were once handed to whichever context asked, and a
    no-wait caller on one thread had its scratch reused by another thread's context while its kernels ran (garbage ring ids, a
    host segfault, a GPU memory fault).  They are per stream now; the two schedules must end at the same map pose."""
    import subprocess
    world = small_problem["world"]
    sweeps = 12
    path = tmp_path / "sweeps.bin"
    with open(path, "wb") as f:
        f.write(np.uint32(16).tobytes() + np.float32(-15.0).tobytes() + np.float32(15.0).tobytes() + np.uint32(sweeps).tobytes())
        for k in range(sweeps):
            gt = (0.0, 0.0, 0.3 + 0.01 * k, 3.0 + 0.4 * k, -2.0 + 0.15 * k, synth.SENSOR_HEIGHT)
            _, _, _, cloud, _ = synth.make_scan(world, 16, 900, gt_pose=gt, seed=300 + k, full=True)
            ring = np.floor(cloud[:, 3]).astype(np.int64)
            a = np.ascontiguousarray(cloud[np.lexsort((ring, -(cloud[:, 3] - ring)))][:, :4], np.float32)
            f.write(np.uint32(len(a)).tobytes())
            f.write(a.tobytes())
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "node_threads"
    libdir = os.path.dirname(pkg.lib_path())
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tools", "cpp", "node_threads.cpp"), "-o", str(exe), "-L", libdir, "-llslam_hip",
                           "-Wl,-rpath," + libdir, "-lpthread"])
    poses = []
    for mode in (["seq"], [], []):
        out = subprocess.run([str(exe), str(path), "2"] + mode, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (mode, out.returncode, out.stderr[-1500:])
        w = out.stdout.split()
        poses.append(float(w[w.index("travelled_m") + 1]))
    assert poses[0] > 3.0 and max(abs(p - poses[0]) for p in poses[1:]) <= 1e-5, poses
