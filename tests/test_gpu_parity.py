"""GPU parity tests proper: the HIP path, called through the C ABI, against the
CPU oracle on the same seeded inputs and against the committed golden vectors.

Bars:  kNN indices and squared distances, fit accept/reject flags, residual
coefficients: bit-exact (the kernels replay the reference's fp32 operation
sequence).  Normal-equation sums: rtol 2e-5 (different summation order only).
Final pose: 1e-4 m / 1e-5 rad (BASELINE.json north_star).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-5


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


SEARCH = {"lane": 1, "packet": 2}  # LSLAM_SEARCH_LANE / _PACKET (include/lslam_c.h)
# the lane search through each traversal-stack shape (LSLAM_STACK_DEEP / _SHALLOW): "lane_shallow" is the instantiation a
# batch launch -- the bench -- takes, sweep_kernel<256, true, false, 12> (12 levels in LDS, the rest in HBM)
SEARCH_SHAPES = {"lane": 1 | 0x100, "lane_shallow": 1 | 0x200, "packet": 2}
SHAPE_VARIANT = {"lane": "deep", "lane_shallow": "shallow", "packet": "packet"}


@pytest.mark.parametrize("search", ["lane", "packet"])
def test_knn5_matches_reference_goldens(ctx, goldens, search):
    """Both search implementations against the outputs of the reference's own nanoflann -- including the
    lattice / duplicate fixtures, where exact distance ties make the answer depend on nanoflann's visit order
    (the packet search detects such queries and redoes them with nanoflann's traversal)."""
    ties_seen = 0
    for name, g in goldens.items():
        pts = g["pts"]
        if len(pts) < 5:
            continue
        # the map API wants a corner and a surf cloud; use the same cloud for both
        ctx.map_set(pts, pts)
        for which in (0, 1):
            idx, d2, ties = ctx.knn5(which, g["queries"], search_mode=SEARCH[search], want_ties=True)
            assert np.array_equal(idx, g["idx"]), (name, which)
            assert np.array_equal(bits(d2), bits(g["d2"])), (name, which)
            ties_seen += ties
    if search == "packet":
        assert ties_seen > 0  # the tie fixtures did go through the redo path


@pytest.mark.parametrize("search", ["lane", "packet"])
def test_knn5_matches_oracle_large(ctx, oracle, synth, search):
    pr = synth.make_problem(rings=16, azimuth_steps=1800, world_half=100.0)
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    info = ctx.map_info()
    assert info.n_surf == len(pr["map_surf"]) and info.depth_surf <= 64
    rng = np.random.default_rng(0)
    for which, cloud in ((0, pr["map_corner"]), (1, pr["map_surf"])):
        tree = oracle.kdtree(cloud)
        q = cloud[rng.integers(0, len(cloud), 4000), :3] + rng.normal(0, 0.5, (4000, 3)).astype(np.float32)
        q = np.concatenate([q, rng.uniform(-150, 150, (500, 3)).astype(np.float32)])  # far outside
        gi, gd = ctx.knn5(which, q, search_mode=SEARCH[search])
        oi, od = tree.knn(q, 5)
        assert np.array_equal(gi, oi)
        assert np.array_equal(bits(gd), bits(od))


def _tree_dump(ctx, which, n_pts):
    import ctypes as C
    lib = ctx.lib
    lib.lslam_debug_tree_dump.restype = C.c_int
    lib.lslam_debug_tree_dump.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_size_t,
                                          C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_int32)]
    cap = 3 * n_pts + 128  # node slots are taken in groups of eight, the subtrees of at most 64 points get five groups each up front
    nodes = np.zeros((cap, 4), np.uint32)
    pts = np.zeros((max(1, n_pts), 4), np.float32)
    root, nn = C.c_uint32(), C.c_int32()
    rc = lib.lslam_debug_tree_dump(ctx.h, which, nodes.ctypes.data_as(C.POINTER(C.c_uint32)), cap,
                                   pts.ctypes.data_as(C.POINTER(C.c_float)), len(pts), C.byref(root), C.byref(nn))
    assert rc == 0
    return nodes[:nn.value], pts[:n_pts], root.value


def _walk(nodes, root):
    """(divfeat, divlow, divhigh) of inner nodes and (left, count) of leaves in nanoflann's
    preorder, independent of where the device placed the nodes."""
    inner, leaves = [], []
    stack = [root]
    f32 = nodes.view(np.float32)
    while stack:
        ref = stack.pop()
        if ref & 0x80000000:
            leaves.append(((ref & 0x7FFFFFFF) >> 4, ref & 15))
            continue
        n = ref >> 2
        inner.append((ref & 3, f32[n, 0], f32[n, 1]))
        stack.append(int(nodes[n, 3]))
        stack.append(int(nodes[n, 2]))
    return inner, leaves


def _assert_tree_is_nanoflann_exact(ctx, oracle, pts):
    pts = np.ascontiguousarray(pts, np.float32)
    ctx.map_set(pts, pts)
    info = ctx.map_info()
    assert info.built_on_device == 1  # there is no other builder
    tree = oracle.kdtree(pts)
    nodes, dpts, root = _tree_dump(ctx, 1, len(pts))
    assert np.array_equal(dpts[:, 3].view(np.int32), tree.vind())      # same permutation
    assert np.array_equal(dpts[:, :3], pts[tree.vind()])
    inner, leaves = _walk(nodes, root)
    on = tree.nodes()
    o_inner = [(int(a), lo, hi) for k, a, lo, hi in zip(on["kind"], on["a"], on["divlow"], on["divhigh"]) if k == 1]
    o_leaves = [(int(a), int(b - a)) for k, a, b in zip(on["kind"], on["a"], on["b"]) if k == 0]
    assert len(inner) == len(o_inner) and len(leaves) == len(o_leaves)
    assert leaves == o_leaves
    assert [i[0] for i in inner] == [i[0] for i in o_inner]
    assert np.array_equal(np.array([i[1:] for i in inner], np.float32).view(np.int32),
                          np.array([i[1:] for i in o_inner], np.float32).view(np.int32))
    assert info.depth_surf == tree.max_depth()


@pytest.mark.parametrize("kind", ["planar", "uniform", "lattice", "lattice_big", "duplicates", "big"])
def test_device_tree_build_is_nanoflann_exact(ctx, oracle, synth, kind):
    """The GPU-built tree has nanoflann's split at every node and nanoflann's point order
    (vind): compared with the oracle's restatement, which is pinned to the reference."""
    rng = np.random.default_rng(11)
    if kind == "planar":
        n = 60000
        pts = np.stack([rng.uniform(-60, 60, n), rng.uniform(-60, 60, n), rng.normal(0, 0.02, n)], 1)
    elif kind == "uniform":
        pts = rng.uniform(-30, 30, (40000, 3))
    elif kind == "lattice":  # equal coordinates everywhere: the '== cutval' runs matter
        g = np.arange(-12, 12, 0.4)
        X, Y = np.meshgrid(g, g)
        pts = np.concatenate([np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], 1),
                              np.stack([np.full(X.size, 5.0), X.ravel(), Y.ravel() + 12], 1)])
    elif kind == "lattice_big":  # > 32768 points per node: the level-synchronous phase with '== cutval' runs
        g = np.arange(-12, 12, 0.1)
        X, Y = np.meshgrid(g, g)
        pts = np.concatenate([np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], 1),
                              np.stack([np.full(X.size, 5.0), X.ravel(), Y.ravel() + 12], 1)])
    elif kind == "duplicates":
        base = rng.uniform(-5, 5, (3000, 3))
        pts = np.repeat(base, 4, axis=0)[rng.permutation(12000)]
    else:
        pr = synth.make_problem(rings=16, azimuth_steps=900, world_half=100.0)
        pts = pr["map_surf"][:, :3]
    _assert_tree_is_nanoflann_exact(ctx, oracle, pts)


@pytest.mark.parametrize("huge_min", [1536, 4096, 32768])
def test_device_tree_build_hand_over_sizes(ctx, oracle, monkeypatch, huge_min):
    """The size at which the level phase hands a node to the workgroup-per-node phase (kd_build_medium_kernel, default
    8 192 points) changes the schedule, not the tree: the chain to the end (1 536), an early and a late hand-over -- the
    last with a dozen pending siblings on a workgroup's stack -- on a planar cloud, a lattice with '== cutval' runs
    everywhere and a cloud with a dense corner (unbalanced splits), node by node against the oracle."""
    monkeypatch.setenv("LSLAM_HUGE_MIN", str(huge_min))
    rng = np.random.default_rng(23)
    n = 150000
    planar = np.stack([rng.uniform(-80, 80, n), rng.uniform(-80, 80, n), rng.normal(0, 0.02, n)], 1)
    g = np.arange(-12, 12, 0.1)
    X, Y = np.meshgrid(g, g)
    lattice = np.concatenate([np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], 1),
                              np.stack([np.full(X.size, 5.0), X.ravel(), Y.ravel() + 12], 1)])
    corner = np.concatenate([rng.uniform(-100, 100, (60000, 3)), rng.normal(0, 0.5, (40000, 3)) + 90.0])
    for pts in (planar, lattice, corner):
        _assert_tree_is_nanoflann_exact(ctx, oracle, pts)


def test_device_tree_build_small_and_degenerate_clouds(ctx, oracle):
    """Sizes around the 10-point leaf, the 64-point register path and its hand-over from the LDS path,
    with exact duplicates, all-equal clouds and points on a line (every '== cutval' branch of
    planeSplit, roots that are themselves register nodes)."""
    rng = np.random.default_rng(5)
    for n in (11, 12, 20, 21, 22, 63, 64, 65, 127, 128, 129, 640, 1536, 1537, 3000):
        for kind in ("random", "dups", "equal", "line"):
            if kind == "random":
                pts = rng.uniform(-3, 3, (n, 3))
            elif kind == "dups":
                pts = rng.uniform(-3, 3, (max(2, n // 5), 3))[rng.integers(0, max(2, n // 5), n)]
            elif kind == "equal":
                pts = np.tile(rng.uniform(-3, 3, (1, 3)), (n, 1))
            else:
                pts = np.zeros((n, 3))
                # many ties along the only spread axis.  "+ 0.0" turns the -0.0 that np.round produces
                # into +0.0: a node holding both zeros gets divlow/divhigh = -0.0 from the hardware's
                # min/max (-0 < +0) where nanoflann's "if (v < min)" keeps whichever came first -- the
                # two differ in the sign bit of a zero only, which no distance or comparison can see.
                pts[:, 1] = np.round(rng.uniform(-3, 3, n), 1) + 0.0
            _assert_tree_is_nanoflann_exact(ctx, oracle, pts)


def test_knn5_pointxyzi_stride(ctx, oracle):
    """32-byte pcl::PointXYZI layout (quirk Q9) gives the same answers as packed xyz."""
    rng = np.random.default_rng(1)
    pts = np.zeros((5000, 8), np.float32)
    pts[:, :3] = rng.normal(0, 8, (5000, 3))
    pts[:, 4] = rng.uniform(0, 64, 5000)
    q = np.zeros((700, 8), np.float32)
    q[:, :3] = rng.normal(0, 9, (700, 3))
    ctx.map_set(pts, pts)
    gi, gd = ctx.knn5(1, q)
    oi, od = oracle.kdtree(pts).knn(q, 5)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))


def test_knn5_deep_tree_uses_overflow_stack(ctx, oracle):
    """Geometrically spaced clusters make nanoflann's mid-split tree ~45 levels deep:
    deeper than the LDS part of the traversal stack (32 entries)."""
    rng = np.random.default_rng(2)
    centers = 1.6 ** np.arange(56)
    pts = np.concatenate([np.stack([c + rng.uniform(0, 0.01 * c, 12), rng.uniform(0, 0.01, 12),
                                    rng.uniform(0, 0.01, 12)], 1) for c in centers]).astype(np.float32)
    tree = oracle.kdtree(pts)
    assert 34 < tree.max_depth() <= 64
    q = np.concatenate([pts[rng.integers(0, len(pts), 300)] * np.float32(1.001),
                        rng.uniform(0, 100, (200, 3)).astype(np.float32)])
    ctx.map_set(pts, pts)
    assert ctx.map_info().depth_surf == tree.max_depth()
    gi, gd = ctx.knn5(1, q)
    oi, od = tree.knn(q, 5)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))


def test_degeneracy_decision_across_the_threshold(ctx, oracle):
    """ScanMatch.cpp:222-233 asks "is the smallest eigenvalue of A^T A below 100?".  The solve answers with a cheap sufficient test
    first (A - 101 I - 1e-5 tr(A) I positive definite: certainly not) and computes the eigenvalues only when that fails; the
    decision must be the oracle's eigensolver's on either side of the threshold and inside the margin, with large and small
    traces."""
    rng = np.random.default_rng(11)
    n_deg = 0
    for trial in range(60):
        Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
        thresh = 100.0  # (the tap's threshold, ScanMatch.cpp:223)
        lam_min = thresh * float(rng.choice([0.2, 0.9, 0.99, 0.9999, 1.0001, 1.005, 1.011, 1.02, 1.2, 5.0]))
        top = float(rng.choice([3e2, 1e4, 1e6, 3e7]))
        lam = np.sort(np.concatenate([[lam_min], rng.uniform(lam_min * 1.5 + 1.0, max(top, lam_min * 2 + 2.0), 5)]))
        A = (Q * lam) @ Q.T
        AtA = ((A + A.T) / 2).astype(np.float32)
        Atb = rng.normal(0, 1.0, 6).astype(np.float32)
        pose = np.zeros(6, np.float32)
        o = oracle.gn_step(AtA, Atb, 0, pose, np.zeros(36), False, eig_thresh=thresh)
        g = ctx.gn_step(AtA, Atb, 0, pose, np.zeros(36), False)
        assert bool(g["degenerate"]) == bool(o["degenerate"]), (trial, lam_min, thresh, top)
        n_deg += int(bool(o["degenerate"]))
        if not o["degenerate"]:
            assert np.array_equal(bits(g["x"]), bits(o["x"]))
    assert 10 < n_deg < 50


@pytest.mark.parametrize("search", ["lane", "lane_shallow", "packet"])
@pytest.mark.parametrize("jtj_mode", [0, 1])
def test_sweep_matches_oracle(ctx, oracle, small_problem, jtj_mode, search):
    pr = small_problem
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    tc, ts = oracle.kdtree(pr["map_corner"]), oracle.kdtree(pr["map_surf"])
    before = ctx.sweep_launches()
    for pose in (pr["init_pose"], pr["gt_pose"]):
        g = ctx.sweep(pose, jtj_mode=jtj_mode, search_mode=SEARCH_SHAPES[search])
        o = oracle.sweep(tc, ts, pr["corner"], pr["surf"], pose)
        assert np.array_equal(g["idx"], o["idx"])
        assert np.array_equal(bits(g["d2"]), bits(o["d2"]))
        assert np.array_equal(g["flags"], o["flags"])
        assert np.array_equal(bits(g["coeff"]), bits(o["coeff"]))
        # sums: 21 AtA + 6 Atb (order-of-summation tolerance), then exact counters
        scale = np.abs(o["sums"][:27]).max()
        assert np.abs(g["sums"][:27] - o["sums"][:27]).max() <= 2e-5 * scale
        assert g["sums"][27] == o["sums"][27] and g["sums"][28] == o["sums"][28]
        assert (o["flags"] & 4).sum() > 1000
    after = ctx.sweep_launches()  # the instantiation asked for is the one that ran
    assert after[SHAPE_VARIANT[search]] - before[SHAPE_VARIANT[search]] == 2
    assert sum(after.values()) - sum(before.values()) == 2


def test_sweep_mfma_equals_valu_path(ctx, small_problem):
    pr = small_problem
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    a = ctx.sweep(pr["init_pose"], jtj_mode=0, taps=False)["sums"]
    b = ctx.sweep(pr["init_pose"], jtj_mode=1, taps=False)["sums"]
    assert np.abs(a[:27] - b[:27]).max() <= 1e-5 * np.abs(a[:27]).max()
    assert np.array_equal(a[27:29], b[27:29])


def test_gn_step_matches_oracle(ctx, oracle, small_problem):
    pr = small_problem
    tc, ts = oracle.kdtree(pr["map_corner"]), oracle.kdtree(pr["map_surf"])
    s = oracle.sweep(tc, ts, pr["corner"], pr["surf"], pr["init_pose"])["sums"]
    AtA = np.zeros((6, 6), np.float32)
    k = 0
    for i in range(6):
        for j in range(i, 6):
            AtA[i, j] = AtA[j, i] = s[k]
            k += 1
    Atb = s[21:27]
    o = oracle.gn_step(AtA, Atb, 0, pr["init_pose"], np.zeros(36), False)
    g = ctx.gn_step(AtA, Atb, 0, pr["init_pose"], np.zeros(36), False)
    # same fp32 operation sequence on identical input: bit-exact update
    assert np.array_equal(bits(g["x"]), bits(o["x"]))
    assert np.array_equal(bits(g["pose"]), bits(o["pose"]))
    assert g["degenerate"] == o["degenerate"] and g["converged"] == o["converged"]
    assert abs(g["delta_r"] - o["delta_r"]) <= 1e-6 * max(1, abs(o["delta_r"]))
    assert abs(g["delta_t"] - o["delta_t"]) <= 1e-6 * max(1, abs(o["delta_t"]))


def test_gn_step_degenerate_projection(ctx, oracle):
    """Corridor-like normal equations: eigenvalues below 100 trigger the projector
    (quirk Q2).  Behavioural check against the oracle + tolerance on the numbers."""
    rng = np.random.default_rng(5)
    J = rng.normal(size=(400, 6)) * np.array([3, 3, 3, 1, 1, 0.05])
    AtA = (J.T @ J).astype(np.float32)
    Atb = (J.T @ rng.normal(0, 0.1, 400)).astype(np.float32)
    pose = np.zeros(6, np.float32)
    o = oracle.gn_step(AtA, Atb, 0, pose, np.zeros(36), False)
    g = ctx.gn_step(AtA, Atb, 0, pose, np.zeros(36), False)
    assert o["degenerate"] and g["degenerate"]
    assert np.abs(g["matP"] - o["matP"]).max() < 1e-4
    assert np.abs(g["x"] - o["x"]).max() < 1e-5 + 1e-4 * np.abs(o["x"]).max()


@pytest.mark.parametrize("search", ["lane", "lane_shallow", "packet"])
@pytest.mark.parametrize("jtj_mode", [0, 1])
def test_full_loop_pose_matches_oracle(ctx, oracle, small_problem, jtj_mode, search):
    pr = small_problem
    opts = ctx.default_opts()
    opts.jtj_mode = jtj_mode
    opts.search_mode = SEARCH_SHAPES[search]
    before = ctx.sweep_launches()
    status, pose, st = ctx.scanmatch_full(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                          pr["init_pose"], opts)
    ok, opose, ost = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                           pr["init_pose"])
    assert (status == 0) == ok and st.status == ost.status
    assert st.iterations == ost.iterations and st.converged == ost.converged
    assert (st.n_line, st.n_plane, st.n_rows) == (ost.n_line, ost.n_plane, ost.n_rows)
    assert np.abs(pose[3:] - opose[3:]).max() <= POSE_TOL_M
    assert np.abs(pose[:3] - opose[:3]).max() <= POSE_TOL_RAD
    assert abs(st.score - ost.score) <= 1e-5 * ost.score
    assert abs(st.percent - ost.percent) <= 1e-6
    assert st.point_residuals == ost.point_residuals
    after = ctx.sweep_launches()
    ran = {k for k in after if after[k] != before[k]}
    assert ran == {SHAPE_VARIANT[search]}, ran


@pytest.mark.parametrize("search", ["lane", "lane_shallow"])
def test_fine_score_matches_oracle(ctx, oracle, synth, small_problem, search):
    """setFineScore(true), ScanMatch.cpp:272-321: after a converged loop one more sweep at the final pose, gated on the NEAREST
    neighbour (d2[0] < 0.02 corner / 0.05 surf), gives score2 / percent2 -- printed by the reference, never part of the
    return value.  Device against oracle; single scan and a batch in which one scan does not converge."""
    pr = small_problem
    opts = ctx.default_opts()
    opts.fine_score = 1
    opts.search_mode = SEARCH_SHAPES[search]
    oopts = oracle.default_opts()
    oopts.fine_score = 1
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    status, pose, st = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"], opts)
    ok, opose, ost = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"], oopts)
    assert st.converged and ost.converged and (status == 0) == ok
    assert ost.score2 > 100 and 0.05 < ost.percent2 < ost.percent  # the tighter gate keeps fewer points
    assert abs(st.score2 - ost.score2) <= 2e-3 * ost.score2   # a handful of points sit within the poses' last-bit difference of the gate
    assert abs(st.percent2 - ost.percent2) <= 2e-3
    assert abs(st.score - ost.score) <= 1e-5 * ost.score and abs(st.percent - ost.percent) <= 1e-6  # the gate's inputs: untouched
    assert np.abs(pose - opose).max() <= POSE_TOL_M
    # off: zeros; score gate off: the reference never gets there (:263)
    st0 = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"])[2]
    assert st0.score2 == 0.0 and st0.percent2 == 0.0 and st0.score == st.score
    opts.use_score = 0
    st1 = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"], opts)[2]
    assert st1.score2 == 0.0 and st1.status == 2
    opts.use_score = 1
    # batch: scan 1 is far from the map (no convergence -> no fine sweep for it), scans 0 and 2 get their own numbers
    qc2, qs2, gt2 = synth.make_scan(pr["world"], 16, 450, gt_pose=(0.0, 0.01, 0.5, 1.0, -1.0, synth.SENSOR_HEIGHT), seed=77)
    scans = [(pr["corner"], pr["surf"]), (pr["corner"][:200], pr["surf"][:3000]), (qc2, qs2)]
    inits = np.stack([pr["init_pose"], pr["init_pose"] + np.array([0, 0, 0, 400, 0, 0], np.float32), synth.perturb_pose(gt2, seed=5)])
    ctx.scan_set_batch(scans)
    worst, poses, stats = ctx.run_batch(inits, opts)
    assert abs(stats[0].score2 - st.score2) <= 1e-9 * st.score2 and stats[0].percent2 == st.percent2  # same bits as alone
    assert stats[1].status == 5 and stats[1].score2 == 0.0
    ok2, opose2, ost2 = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], qc2, qs2, inits[2], oopts)
    assert stats[2].converged and ost2.converged
    assert abs(stats[2].score2 - ost2.score2) <= 2e-3 * ost2.score2 and abs(stats[2].percent2 - ost2.percent2) <= 2e-3


def test_full_loop_mapping_settings(ctx, oracle, small_problem):
    """LaserMatcher.cpp:94-95: thresholds 0.1/0.1 and score gate off -> always 'false', pose used (Q7)."""
    pr = small_problem
    opts = ctx.default_opts()
    opts.delta_t_abort = opts.delta_r_abort = 0.1
    opts.use_score = 0
    oopts = oracle.default_opts()
    oopts.delta_t_abort = oopts.delta_r_abort = 0.1
    oopts.use_score = 0
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    status, pose, st = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"], opts)
    ok, opose, ost = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                           pr["init_pose"], oopts)
    assert status == 2 and not ok and st.converged and ost.converged
    assert st.iterations == ost.iterations
    assert np.abs(pose[3:] - opose[3:]).max() <= POSE_TOL_M
    assert np.abs(pose[:3] - opose[:3]).max() <= POSE_TOL_RAD


def test_guards_and_edge_cases(ctx, pkg, oracle, small_problem):
    pr = small_problem
    S = pkg.Status
    # ScanMatch.cpp:57-61: too few reference points, pose untouched
    status, pose, st = ctx.scanmatch_full(pr["map_corner"][:49], pr["map_surf"], pr["corner"], pr["surf"],
                                          pr["init_pose"])
    assert status == S.TOO_FEW_REF and np.array_equal(pose, pr["init_pose"])
    status, pose, st = ctx.scanmatch_full(pr["map_corner"], pr["map_surf"][:99], pr["corner"], pr["surf"],
                                          pr["init_pose"])
    assert status == S.TOO_FEW_REF
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    # too few matches: scan far away from the map (ScanMatch.cpp:141-145)
    far = pr["init_pose"].copy()
    far[3] += 500
    status, pose, st = ctx.scanmatch_scan(pr["corner"], pr["surf"], far)
    assert status == S.TOO_FEW_MATCHES and st.iterations == 0 and np.array_equal(pose, far)
    assert st.sweeps == 1
    # empty scan: no rows -> same path
    empty = np.zeros((0, 4), np.float32)
    status, pose, st = ctx.scanmatch_scan(empty, empty, pr["init_pose"])
    assert status == S.TOO_FEW_MATCHES and st.n_rows == 0
    # ragged sizes (not multiples of the block size), corner-only and surf-only scans
    for nc, ns in ((1, 777), (130, 0), (0, 1000), (129, 4099)):
        c, s = pr["corner"][:nc], pr["surf"][:ns]
        status, pose, st = ctx.scanmatch_scan(c, s, pr["init_pose"])
        ok, opose, ost = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], c, s, pr["init_pose"])
        assert st.iterations == ost.iterations and st.n_rows == ost.n_rows, (nc, ns)
        assert np.abs(pose - opose).max() <= POSE_TOL_M, (nc, ns)
    # max_iterations = 1: exactly one solve
    opts = ctx.default_opts()
    opts.max_iterations = 1
    status, pose, st = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"], opts)
    assert st.iterations == 1 and st.sweeps == 1
    # API misuse is reported, not crashed
    c2 = pkg.Context(0)
    with pytest.raises(pkg.LslamError) as e:
        c2.run(pr["init_pose"])
    assert e.value.code == S.ERR_NO_MAP
    c2.close()


def test_batch_equals_individual_runs(ctx, synth, small_problem):
    """Independent scans matched together (one launch sequence) give, bit for bit, what
    each gives alone: every scan keeps its own block decomposition and summation order."""
    pr = small_problem
    world = pr["world"]
    scans, inits = [], []
    for k, (rings, steps) in enumerate(((16, 900), (16, 450), (8, 300), (16, 1200))):
        gt = (0.01 * k, -0.01, 0.3 + 0.2 * k, 3.0 - 2 * k, -2.0 + k, synth.SENSOR_HEIGHT)
        qc, qs, gt = synth.make_scan(world, rings, steps, gt_pose=gt, seed=100 + k)
        scans.append((qc, qs))
        inits.append(synth.perturb_pose(gt, seed=200 + k))
    inits[2][3] += 400.0  # this one is far from the map: too few matches, pose untouched
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    single = []
    for (qc, qs), p0 in zip(scans, inits):
        single.append(ctx.scanmatch_scan(qc, qs, p0))
    ctx.scan_set_batch(scans)
    worst, poses, stats = ctx.run_batch(np.stack(inits))
    for k, (status, pose, st) in enumerate(single):
        assert stats[k].status == st.status and stats[k].iterations == st.iterations, k
        assert (stats[k].n_rows, stats[k].n_line, stats[k].n_plane) == (st.n_rows, st.n_line, st.n_plane)
        assert np.array_equal(bits(poses[k]), bits(pose)), k
        assert stats[k].point_residuals == st.point_residuals
    assert stats[2].status == 5 and np.array_equal(poses[2], inits[2])
    assert len({s.iterations for s in stats}) > 1  # scans really stop at different iterations
    assert worst == 5
    # a second batch call on the same resident scans is deterministic
    _, poses2, _ = ctx.run_batch(np.stack(inits))
    assert np.array_equal(bits(poses2), bits(poses))


def test_cube_map_variant_matches_oracle(ctx, oracle, small_problem):
    """Variant C, FeatureMap::scanMatchScan (util/FeatureMap.h:490-691): per-cube kd-trees.
    20 m cubes so that the 120 m test map spans many cubes and points near cube borders
    really see a different neighbourhood than with the whole-map tree."""
    pr = small_problem
    grid = dict(cube_size=20.0, origin=(5, 5, 1), dims=(11, 11, 3))
    ctx.cubemap_set(pr["map_corner"], pr["map_surf"], **grid)
    assert ctx.map_info().built_on_device == 1  # all cube trees in one device build
    opts = ctx.default_opts()
    opts.use_score = 0  # no score gate in this variant
    status, pose, st = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"], opts)
    ok, opose, ost = oracle.scanmatch_cubes(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                            pr["init_pose"], grid["cube_size"], grid["origin"], grid["dims"])
    assert st.converged == ost.converged == 1 and st.iterations == ost.iterations
    assert (st.n_line, st.n_plane, st.n_rows) == (ost.n_line, ost.n_plane, ost.n_rows)
    assert np.abs(pose[3:] - opose[3:]).max() <= POSE_TOL_M and np.abs(pose[:3] - opose[:3]).max() <= POSE_TOL_RAD
    # it is a different computation from the whole-map search
    ok2, wpose, wst = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                            pr["init_pose"])
    assert wst.n_rows != ost.n_rows
    # a grid that does not cover the scene: every point falls outside -> too few matches
    ctx.cubemap_set(pr["map_corner"], pr["map_surf"], cube_size=20.0, origin=(-50, -50, 0), dims=(2, 2, 1))
    status, pose, st = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"], opts)
    assert status == 5 and st.n_rows == 0
    ctx.map_set(pr["map_corner"], pr["map_surf"])  # back to whole-map trees
    status, pose, st = ctx.scanmatch_scan(pr["corner"], pr["surf"], pr["init_pose"])
    assert st.n_rows == wst.n_rows


def _odometry_pair(synth, world, k=0):
    """Two consecutive 16-ring scans: targets = features of the previous scan (scan order),
    queries = a subset of the current scan's corner / surf points (sharp / flat)."""
    gt0 = (0.0, 0.0, 0.30 + 0.05 * k, 3.0, -2.0, synth.SENSOR_HEIGHT)
    gt1 = (0.002, -0.003, 0.33 + 0.05 * k, 3.35, -1.9, synth.SENSOR_HEIGHT)
    lc, ls, _ = synth.make_scan(world, 16, 900, gt_pose=gt0, seed=40 + k)
    c1, s1, _ = synth.make_scan(world, 16, 900, gt_pose=gt1, seed=41 + k)
    return lc, ls[::2], c1[::3], s1[::20]


def test_odometry_variant_matches_oracle(ctx, oracle, synth, small_problem):
    """Variant B, LaserOdometry::scanMatch (odometry/LaserOdometry.cpp:328-647).  The per-point
    de-skew uses device sin/cos (not glibc's), so parity is by tolerance here."""
    lc, ls, sharp, flat = _odometry_pair(synth, small_problem["world"])
    assert 100 < len(sharp) < len(lc) and 300 < len(flat) < len(ls)
    p0 = np.zeros(6, np.float32)
    status, pose, st = ctx.odometry_match(lc, ls, sharp, flat, p0)
    n, opose, ost = oracle.odometry_match(lc, ls, sharp, flat, p0)
    assert st.converged == ost.converged and st.iterations == ost.iterations and st.sweeps == n
    assert (st.n_rows, st.n_line, st.n_plane) == (ost.n_rows, ost.n_line, ost.n_plane)
    assert np.abs(pose[3:] - opose[3:]).max() <= POSE_TOL_M and np.abs(pose[:3] - opose[:3]).max() <= POSE_TOL_RAD
    assert np.abs(pose).max() > 1e-3  # it moved
    # persistent _transform as the next initial guess; few iterations allowed
    status2, pose2, st2 = ctx.odometry_match(lc, ls, sharp, flat, pose, max_iterations=3)
    n2, opose2, ost2 = oracle.odometry_match(lc, ls, sharp, flat, opose, max_iterations=3)
    assert st2.iterations == ost2.iterations and np.abs(pose2 - opose2).max() <= POSE_TOL_M
    # guard of LaserOdometry.cpp:337
    status3, pose3, st3 = ctx.odometry_match(lc[:10], ls, sharp, flat, p0)
    assert status3 == 1 and np.array_equal(pose3, p0)
    # PointXYZI layout (32-byte stride, intensity at byte 16)
    def xyzi(a):
        o = np.zeros((len(a), 8), np.float32)
        o[:, :3] = a[:, :3]
        o[:, 4] = a[:, 3]
        return o
    status4, pose4, st4 = ctx.odometry_match(xyzi(lc), xyzi(ls), xyzi(sharp), xyzi(flat), p0)
    assert np.array_equal(bits(pose4), bits(pose))
    # the map slot was used for the odometry trees: a scan match now needs a new map
    ctx.map_set(small_problem["map_corner"], small_problem["map_surf"])


def test_scanmatch_class_mirrors_reference_api(pkg, oracle, small_problem):
    pr = small_problem
    sm = pkg.ScanMatch(10)
    sm.setConvergeThreshold(0.05, 0.05)
    ok, pose = sm.scanMatchScan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])
    ok_o, opose, ost = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                             pr["init_pose"])
    assert ok == ok_o and np.abs(pose - opose).max() <= POSE_TOL_M
    assert abs(sm.getAverageScore() - ost.score) <= 1e-5 * ost.score
    # Isometry3f overload (ScanMatch.cpp:349-360)
    T0 = sm.ctx.pose_to_isometry(pr["init_pose"])
    ok2, T = sm.scanMatchScan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], T0)
    assert ok2 and T.shape == (4, 4)
    R, t = oracle.pose_to_Rt(opose)
    assert np.abs(T[:3, 3] - t).max() <= 2 * POSE_TOL_M and np.abs(T[:3, :3] - R).max() <= 1e-4
    # score gate (loop-closure usage): impossible threshold -> false, pose still written
    sm.setScoreThreshold(1e9)
    ok3, pose3 = sm.scanMatchScan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])
    assert not ok3 and sm.last_stats.status == pkg.Status.LOW_SCORE
    assert np.abs(pose3 - opose).max() <= POSE_TOL_M


def test_full_size_64ring_properties(ctx, synth):
    """BASELINE config 3 size (64x1800 = 115 200 points, ~1.3 M-point map): too big for
    the oracle in a unit test, so check size-independent properties: recovery of the
    ground-truth pose, idempotence of a converged pose, determinism, VALU == MFMA."""
    pr = synth.make_problem(rings=64, azimuth_steps=1800)
    assert len(pr["corner"]) + len(pr["surf"]) == 64 * 1800
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    status, pose, st = ctx.run(pr["init_pose"])
    assert st.converged and status == 0
    assert np.abs(pose[3:] - pr["gt_pose"][3:]).max() < 0.02
    assert np.abs(pose[:3] - pr["gt_pose"][:3]).max() < 1e-3
    assert st.point_residuals == st.sweeps * 115200
    # determinism: same input, same bits
    status2, pose2, st2 = ctx.run(pr["init_pose"])
    assert np.array_equal(bits(pose), bits(pose2)) and st2.iterations == st.iterations
    # idempotence: restarting from the converged pose converges at once and stays put
    status3, pose3, st3 = ctx.run(pose)
    assert st3.iterations <= 2 and np.abs(pose3 - pose).max() < 2e-3
    # MFMA J^T J agrees with the VALU path
    opts = ctx.default_opts()
    opts.jtj_mode = 1
    status4, pose4, st4 = ctx.run(pr["init_pose"], opts)
    assert st4.iterations == st.iterations and np.abs(pose4 - pose).max() <= POSE_TOL_M
    # the two search implementations find the same neighbours: same rows, same sums order -> same bits
    res = {}
    for name, mode in SEARCH.items():
        o2 = ctx.default_opts()
        o2.search_mode = mode
        res[name] = ctx.run(pr["init_pose"], o2)
        sw = ctx.sweep(pr["init_pose"], jtj_mode=1, search_mode=mode)
        res[name + "_sweep"] = sw
    assert np.array_equal(bits(res["lane"][1]), bits(res["packet"][1]))
    assert res["lane"][2].iterations == res["packet"][2].iterations and res["lane"][2].n_rows == res["packet"][2].n_rows
    for k in ("idx", "d2", "flags", "coeff"):
        assert np.array_equal(res["lane_sweep"][k].view(np.uint8), res["packet_sweep"][k].view(np.uint8)), k


# ---------------------------------------------------------------------------------------
# pose-graph LM (config 4): HIP kernels vs the numpy oracle (parity unpinned: no g2o here)
# ---------------------------------------------------------------------------------------
def _pg_oracle():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import posegraph_oracle as po
    return po


def test_posegraph_linearize_matches_oracle(pkg):
    po = _pg_oracle()
    g = po.make_graph(n_kf=150, n_loop=500, laps=3, radius=18.0)
    pg = pkg.PoseGraph(0)
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    s = pg.linearize()
    H, b, c2 = po.linearize(g["init"], g["ij"], g["meas"], g["info"])  # numeric Jacobians
    Hd = H.toarray()
    n = len(g["init"])
    scale = np.abs(Hd).max()
    assert abs(s["chi2"] - c2) <= 1e-10 * c2
    for v in range(1, n):  # vertex 0 is fixed: identity block, zero rhs
        assert np.abs(s["diag"][v] - Hd[6 * v:6 * v + 6, 6 * v:6 * v + 6]).max() <= 2e-6 * scale
    assert np.array_equal(s["diag"][0], np.eye(6)) and not s["b"][:6].any()
    assert np.abs(s["b"][6:] - b[6:]).max() <= 2e-6 * np.abs(b).max()
    for (i, j), blk in zip(s["off_ij"], s["off"]):
        ref = Hd[6 * i:6 * i + 6, 6 * j:6 * j + 6] if i != 0 else np.zeros((6, 6))
        assert i < j and np.abs(blk - ref).max() <= 2e-6 * scale
    # one damped solve against scipy's sparse LU on the oracle system
    lam = 1e-5 * Hd.diagonal()[6:].max()
    dx, cg = pg.solve(lam)
    ref = po.solve_damped(H, b, lam, 0)
    assert cg > 0 and np.abs(dx - ref).max() <= 1e-5 * np.abs(ref).max()
    pg.close()


def test_posegraph_optimize_matches_oracle(pkg):
    po = _pg_oracle()
    g = po.make_graph(n_kf=300, n_loop=1200, laps=3, radius=40.0)
    pg = pkg.PoseGraph(0)
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    iters = pg.optimize(8)
    st = pg.last_stats
    P, hist = po.optimize(g["init"], g["ij"], g["meas"], g["info"], max_iters=8)
    assert iters == len(hist) == 8
    assert abs(st.chi2_initial - po.chi2(g["init"], g["ij"], g["meas"], g["info"])) <= 1e-9 * st.chi2_initial
    assert abs(st.chi2_final - hist[-1]["chi2"]) <= 1e-4 * hist[-1]["chi2"]
    got = pg.poses()
    assert np.abs(got[:, :3] - P[:, :3]).max() < 1e-4 and np.abs(got[:, 3:] - P[:, 3:]).max() < 1e-5
    assert np.allclose(got[0], g["init"][0])
    pg.close()


def test_posegraph_solver_g2o_style_api(pkg):
    """add_se3_node / add_se3_edge / optimize as pose_graph/solver_g2o.cpp:51-95, with the
    information matrices of pose_graph/graph.cpp:279-288,333-339."""
    po = _pg_oracle()
    g = po.make_graph(n_kf=80, n_loop=200, laps=2, radius=14.0)
    from importlib import import_module
    pgm = import_module("the-cooper-mapper_amd.pose_graph")
    pg = pkg.PoseGraph(0)
    ids = [pg.add_se3_node(pgm.pose7_to_mat(p)) for p in g["init"]]
    assert ids == list(range(80))
    for (i, j), z, w in zip(g["ij"], g["meas"], g["info"]):
        pg.add_se3_edge(i, j, pgm.pose7_to_mat(z), w)
    c0 = po.chi2(g["init"], g["ij"], g["meas"], g["info"])
    assert pg.optimize(10) >= 1
    assert pg.last_stats.chi2_final < 1e-2 * c0
    T0 = pg.estimate(0)
    assert np.allclose(T0, pgm.pose7_to_mat(g["init"][0]))  # first node fixed
    pg.close()


def test_posegraph_sharded_equals_full(pkg):
    """Two edge shards summed through the all-reduce hook (here: a local sum standing in
    for RCCL) give the single-shard system."""
    import torch
    po = _pg_oracle()
    g = po.make_graph(n_kf=100, n_loop=300, laps=2, radius=16.0)
    full = pkg.PoseGraph(0)
    full.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    ref = full.linearize()
    ne = len(g["ij"])
    parts = []
    for a, e in ((0, ne // 2), (ne // 2, ne)):
        p = pkg.PoseGraph(0)
        p.set_graph(g["init"], g["ij"], g["meas"], g["info"])
        p.set_shard(a, e)  # no all-reduce: raw shard contribution
        parts.append(p.linearize())
        p.close()
    n = len(g["init"])
    assert abs(parts[0]["chi2"] + parts[1]["chi2"] - ref["chi2"]) <= 1e-12 * ref["chi2"]
    assert np.abs(parts[0]["b"] + parts[1]["b"] - ref["b"]).max() <= 1e-12 * np.abs(ref["b"]).max()
    d = parts[0]["diag"] + parts[1]["diag"]
    d[0] = np.eye(6)  # each raw shard carries the fixed vertex's identity block
    assert np.abs(d - ref["diag"]).max() <= 1e-12 * np.abs(ref["diag"]).max()
    assert np.abs(parts[0]["off"] + parts[1]["off"] - ref["off"]).max() <= 1e-12 * np.abs(ref["off"]).max()
    # the hook itself: a "world" of one rank whose all-reduce doubles nothing
    calls = []
    hooked = pkg.PoseGraph(0)
    hooked.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    sysbuf = torch.zeros(hooked.system_doubles(), dtype=torch.float64, device="cuda")
    hooked.set_shard(0, ne, allreduce=lambda ptr, count: calls.append((ptr, count)), system_tensor=sysbuf)
    got = hooked.linearize()
    # the system the ranks sum is [diag | off | b | chi2 | fallback flag]; the row-sharded solve's exchange vector follows it
    assert calls and calls[0][0] == sysbuf.data_ptr()
    core = calls[0][1]
    # (6 n of z, eight scalars, two partial sums per rank for up to 64 ranks: the all-gather form of the exchange)
    assert core + 6 * n + 8 + 2 * 64 == hooked.system_doubles()
    assert np.abs(got["b"] - ref["b"]).max() <= 1e-12 * np.abs(ref["b"]).max()
    assert abs(float(sysbuf[core - 2]) - ref["chi2"]) <= 1e-12 * ref["chi2"]
    hooked.close()
    full.close()


def test_sharded_points_gn_loop(pkg, ctx, oracle, small_problem, monkeypatch):
    """one scan's points sharded over ranks, 32 fp64 sums all-reduced per GN
    iteration.  (a) a world of one rank reproduces lslam_scanmatch_run bit for bit; (b) two
    shards on two contexts (two host threads, a local sum standing in for RCCL) give the
    single-GPU pose to 1e-5 m with identical iteration and match counts."""
    import threading
    import torch
    pr = small_problem
    # the sharded loop searches every point in every sweep; the plain loop it is held against bit for bit does the same
    # here (its certificate sweep adds the same terms in another grouping: tests/test_gpu_stack_shapes.py)
    every = ctx.default_opts()
    every.knn_cert = 0
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    status, pose, st = ctx.run(pr["init_pose"], every)
    calls = []
    x = torch.zeros(32, dtype=torch.float64, device="cuda")
    s1, p1, st1 = ctx.run_sharded(pr["init_pose"], lambda ptr, n: calls.append((ptr, n)), x, opts=every)
    assert calls[0] == (x.data_ptr(), 32) and len(calls) == 1 + st.iterations
    assert s1 == status and st1.iterations == st.iterations and np.array_equal(bits(p1), bits(pose))
    assert (st1.n_rows, st1.n_line, st1.n_plane) == (st.n_rows, st.n_line, st.n_plane)
    assert abs(st1.score - st.score) <= 1e-9 * max(1.0, abs(st.score)) and st1.percent == st.percent

    from importlib import import_module
    dist = import_module("the-cooper-mapper_amd.dist")
    world = 2
    xs = [torch.zeros(32, dtype=torch.float64, device="cuda") for _ in range(world)]
    bar = threading.Barrier(world)
    out = [None] * world
    err = []

    def rank_main(r):
        try:
            c = pkg.Context(0)
            c.map_set(pr["map_corner"], pr["map_surf"])
            cb, ce = dist.shard_range(len(pr["corner"]), r, world)
            sb, se = dist.shard_range(len(pr["surf"]), r, world)
            c.scan_set(pr["corner"][cb:ce], pr["surf"][sb:se])

            def allreduce(ptr, n):
                assert ptr == xs[r].data_ptr() and n == 32
                bar.wait()
                if r == 0:
                    tot = xs[0] + xs[1]
                    xs[0].copy_(tot)
                    xs[1].copy_(tot)
                    torch.cuda.synchronize()
                bar.wait()
            out[r] = c.run_sharded(pr["init_pose"], allreduce, xs[r])
            c.close()
        except Exception as e:  # pragma: no cover
            err.append(e)
            bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for r in range(world):
        s2, p2, st2 = out[r]
        assert s2 == status and st2.iterations == st.iterations
        assert (st2.n_rows, st2.n_line, st2.n_plane) == (st.n_rows, st.n_line, st.n_plane)
        assert np.abs(p2[3:] - pose[3:]).max() <= 1e-5 and np.abs(p2[:3] - pose[:3]).max() <= 1e-6
        assert st2.percent == st.percent
    assert np.array_equal(bits(out[0][1]), bits(out[1][1]))  # every rank ends on the same pose


def test_device_morton_order_equals_host_definition():
    """The resident scans are Morton-ordered on the device (one radix sort per batch); the order is
    defined by the host implementation (ascending (key, index) per cloud).  The summation order of
    the sweep depends on it, so equal pose BITS from two child processes, one ordering on the device
    and one on the host (LSLAM_HOST_MORTON=1), mean the two orders are identical (the switches are
    read once per process, hence the children)."""
    import subprocess
    import sys
    code = (
        "import importlib,sys,numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "pkg=importlib.import_module('the-cooper-mapper_amd'); synth=importlib.import_module('the-cooper-mapper_amd.synth')\n"
        "pr=synth.make_problem(rings=16, azimuth_steps=900, world_half=60.0)\n"
        "c=pkg.Context(0); c.map_set(pr['map_corner'], pr['map_surf'])\n"
        "s,p,st=c.scanmatch_scan(pr['corner'], pr['surf'], pr['init_pose'])\n"
        "print('BITS', ' '.join(str(int(v)) for v in p.view(np.int32)), st.iterations, st.n_rows)\n"
    ) % (ROOT,)
    lines = {}
    for name in ("LSLAM_DEVICE_MORTON_IS_THE_DEFAULT", "LSLAM_HOST_MORTON"):
        env = {k: v for k, v in os.environ.items() if not k.endswith("_MORTON")}
        env[name] = "1"
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("BITS")]
        assert line, out.stderr[-2000:]
        lines[name] = line[0]
    assert lines["LSLAM_DEVICE_MORTON_IS_THE_DEFAULT"] == lines["LSLAM_HOST_MORTON"]


def test_posegraph_full_size_properties(pkg, synth):
    """BASELINE configs[3] at full size (5 000 keyframes / 24 999 edges), through properties that do
    not need the oracle: chi2 falls by orders of magnitude, the fixed vertex does not move, the
    optimised trajectory is closer to the ground truth than the drifted one, and two runs agree bit
    for bit (every reduction in the solver has a fixed order)."""
    g = synth.make_pose_graph()
    runs = []
    for _ in range(2):
        pg = pkg.PoseGraph(0)
        pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
        it = pg.optimize(10)
        st = pg.last_stats
        runs.append((it, st.chi2_initial, st.chi2_final, pg.poses()))
        pg.close()
    it, c0, c1, poses = runs[0]
    assert it == 10 and c1 < 1e-3 * c0
    assert np.array_equal(poses[0], np.asarray(g["init"])[0])
    err0 = np.linalg.norm(np.asarray(g["init"])[:, :3] - np.asarray(g["gt"])[:, :3], axis=1).mean()
    err1 = np.linalg.norm(poses[:, :3] - np.asarray(g["gt"])[:, :3], axis=1).mean()
    assert err1 < err0  # ten iterations do not finish the job on 48 m of drift; they must not make it worse
    assert np.array_equal(runs[0][3].view(np.int64), runs[1][3].view(np.int64)) and runs[0][2] == runs[1][2]
    assert np.allclose(np.linalg.norm(poses[:, 3:], axis=1), 1.0, atol=1e-12)  # unit quaternions


def test_persistent_gn_loop_equals_launch_loop(pkg, synth, monkeypatch):
    """lslam_opts.ab_switches & LSLAM_AB_PERSISTENT_GN: the whole Gauss-Newton loop of a resident scan in one persistent launch (every workgroup
    keeps its own copy of the state, grid-wide exchanges of the blocks' sums, the solve replicated) -- bit for bit the
    launch loop's pose, counters and sums, for the scan-to-map settings and the mapping settings."""
    pr = synth.make_problem(rings=64, azimuth_steps=1800)
    out = {}
    for mode in ("0", "1"):
        c = pkg.Context(0)
        try:
            c.map_set(pr["map_corner"], pr["map_surf"])
            c.scan_set(pr["corner"], pr["surf"])
            res = []
            for max_it, dr, dt in ((10, 0.05, 0.05), (30, 0.01, 0.01), (2, 0.05, 0.05)):
                opts = c.default_opts()
                opts.max_iterations, opts.delta_r_abort, opts.delta_t_abort = max_it, dr, dt
                opts.ab_switches = 1 if mode == "1" else 0  # LSLAM_AB_PERSISTENT_GN
                status, pose, st = c.run(pr["init_pose"], opts)
                res.append((status, bits(pose).tolist(), st.iterations, st.n_rows, st.n_line, st.n_plane, st.converged, st.sweeps))
            out[mode] = res
        finally:
            c.close()
    assert out["0"] == out["1"]
    assert out["1"][0][6] == 1 and out["1"][2][2] == 2  # the first converges, the last stops at its iteration limit


def test_posegraph_persistent_solve_equals_launch_loop(pkg, synth, monkeypatch):
    """The damped solve has two forms -- the whole PCG loop in one persistent launch (one workgroup per aggregate,
    grid-wide exchanges through sentinel slots) and the launch-per-step loop kept for graphs that do not fit.  Same
    operator, same stopping rule: the steps agree to the solves' tolerance with the second preconditioner level off and
    on (persistent Gauss-Jordan inverse against the launch-per-pivot one), the LM runs land on the same chi2, and the
    statistics say which form ran."""
    g = synth.make_pose_graph(n_kf=1500, n_loop=4000, laps=2, radius=30.0)
    out = {}
    for coarse in ("0", "1"):
        monkeypatch.setenv("LSLAM_PG_COARSE", coarse)
        for persistent in ("1", "0"):
            monkeypatch.setenv("LSLAM_PG_PERSISTENT", persistent)
            pg = pkg.PoseGraph(0)
            pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
            pg.linearize()
            dx, it = pg.solve(1e-3)
            its = pg.optimize(6)
            st = pg.last_stats
            out[(coarse, persistent)] = (dx, it, st.chi2_final, st.fused_solves, st.lm_trials, its)
            pg.close()
    for coarse in ("0", "1"):
        a, b = out[(coarse, "1")], out[(coarse, "0")]
        assert a[3] == a[4] > 0 and b[3] == 0          # every solve fused / none
        assert a[1] > 0 and abs(a[1] - b[1]) <= 2 + 0.05 * b[1]   # the same PCG, up to rounding, stops at the same place
        assert np.abs(a[0] - b[0]).max() <= 1e-6 * np.abs(b[0]).max()
        assert a[5] == b[5] and abs(a[2] - b[2]) <= 1e-6 * b[2]
    # the second level does its job in both forms: fewer iterations for the same system
    assert out[("1", "1")][1] < out[("0", "1")][1] and out[("1", "0")][1] < out[("0", "0")][1]


def test_posegraph_persistent_timeout_falls_back_to_launch_loop(pkg, synth, monkeypatch):
    """A grid exchange that runs into its spin limit (workgroups not all resident: e.g. another process's persistent
    kernel on the same device) raises the kernel's abort flag; the solve is then redone by the launch-per-step loop and
    the graph stays on it.  Forced here through the debug hook that raises the flag at iteration 3."""
    g = synth.make_pose_graph(n_kf=600, n_loop=1500, laps=2, radius=20.0)
    ref = pkg.PoseGraph(0)
    ref.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    ref.optimize(5)
    monkeypatch.setenv("LSLAM_DEBUG_PG_ABORT", "3")
    pg = pkg.PoseGraph(0)
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    its = pg.optimize(5)
    st = pg.last_stats
    assert its == ref.last_stats.iterations and st.fused_solves == 0 and st.lm_trials > 0
    assert abs(st.chi2_final - ref.last_stats.chi2_final) <= 1e-6 * ref.last_stats.chi2_final
    assert np.abs(pg.poses() - ref.poses()).max() < 1e-6
    pg.close(); ref.close()
    # the persistent coarse inverse likewise (second level forced on): its abort flag sends the inverse, assembled once
    # more, through the launch-per-pivot loop
    monkeypatch.delenv("LSLAM_DEBUG_PG_ABORT")
    monkeypatch.setenv("LSLAM_PG_COARSE", "1")
    ref = pkg.PoseGraph(0)
    ref.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    ref.optimize(5)
    monkeypatch.setenv("LSLAM_DEBUG_GJ_ABORT", "2")
    pg = pkg.PoseGraph(0)
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    assert pg.optimize(5) == ref.last_stats.iterations
    assert abs(pg.last_stats.chi2_final - ref.last_stats.chi2_final) <= 1e-6 * ref.last_stats.chi2_final
    assert np.abs(pg.poses() - ref.poses()).max() < 1e-6
    pg.close(); ref.close()


# ---- the device builder is the only builder: its structure limits fail loudly -----------------
def _fresh_ctx(pkg):
    return pkg.Context(0)


def test_tree_build_retries_with_more_node_slots(pkg, oracle, monkeypatch):
    """Limit 1 (node-slot array too small) is resolved by the retry with a larger array, and the
    tree that results is still nanoflann's."""
    rng = np.random.default_rng(3)
    pts = rng.uniform(-30, 30, (30000, 3)).astype(np.float32)
    monkeypatch.setenv("LSLAM_DEBUG_NODE_CAP_DIV", "8")  # every attempt gets an eighth of its slots (n/3, 2n/3, n): the first cannot fit
    c = _fresh_ctx(pkg)
    try:
        c.map_set(pts, pts)
        info = c.map_info()
        assert info.build_attempts >= 2 and info.built_on_device == 1
        monkeypatch.delenv("LSLAM_DEBUG_NODE_CAP_DIV")
        tree = oracle.kdtree(pts)
        q = pts[:2000, :3] + rng.normal(0, 0.3, (2000, 3)).astype(np.float32)
        gi, gd = c.knn5(1, q)
        oi, od = tree.knn(q, 5)
        assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))
    finally:
        c.close()


def test_tree_build_node_limit_fails_loudly(pkg, monkeypatch):
    rng = np.random.default_rng(4)
    pts = rng.uniform(-30, 30, (30000, 3)).astype(np.float32)
    monkeypatch.setenv("LSLAM_DEBUG_NODE_CAP_DIV", "100000")  # no attempt can fit
    c = _fresh_ctx(pkg)
    try:
        with pytest.raises(pkg.LslamError) as ei:
            c.map_set(pts, pts)
        assert ei.value.code == pkg.Status.ERR_TREE_BUILD and "limit 1" in str(ei.value)
        monkeypatch.delenv("LSLAM_DEBUG_NODE_CAP_DIV")
        c.map_set(pts, pts)  # the context is still usable
        assert c.map_info().n_surf == len(pts)
    finally:
        c.close()


def test_tree_build_queue_watchdog_reports(pkg, monkeypatch):
    """Limit 2: the watchdog of the persistent phase-A kernel, forced by a zero spin budget."""
    rng = np.random.default_rng(5)
    pts = rng.uniform(-30, 30, (40000, 3)).astype(np.float32)  # <= 49152 points: the phase-A path
    monkeypatch.setenv("LSLAM_DEBUG_SPIN_LIMIT", "0")
    c = _fresh_ctx(pkg)
    try:
        with pytest.raises(pkg.LslamError) as ei:
            c.map_set(pts, pts)
        assert ei.value.code == pkg.Status.ERR_TREE_BUILD and "limit 2" in str(ei.value)
        monkeypatch.delenv("LSLAM_DEBUG_SPIN_LIMIT")
        c.map_set(pts, pts)
    finally:
        c.close()


@pytest.mark.parametrize("n_chain", [100, 8000])
def test_tree_deeper_than_traversal_stack_is_refused(pkg, n_chain):
    """Limits 3 / 4: a cloud whose nanoflann tree is deeper than the 64-level device stack (every
    middleSplit_ peels two points off a geometric progression) is refused with TREE_DEPTH -- by the
    wavefront-local builder for the small cloud, by the level driver / node queue for the big one."""
    k = np.arange(200)
    chain = np.stack([1000.0 * 1.5 ** -k, np.zeros(200), np.zeros(200)], 1)
    rng = np.random.default_rng(6)
    # the big variant hangs a few thousand points on every chain node so that the deep nodes stay big
    pts = chain if n_chain == 100 else np.concatenate([chain[i] + rng.uniform(0, 1e-3 * chain[i, 0], (n_chain // 20, 3))
                                                          for i in range(200)])
    pts = np.ascontiguousarray(pts, np.float32)
    c = _fresh_ctx(pkg)
    try:
        with pytest.raises(pkg.LslamError) as ei:
            c.map_set(pts, pts)
        assert ei.value.code == pkg.Status.ERR_TREE_DEPTH
    finally:
        c.close()


def test_batch_chunking_does_not_change_a_bit(ctx, synth, small_problem):
    """lslam_opts.scans_in_flight only decides how many scans share a launch sequence: 7 scans matched 1, 2, 3 or
    all at a time (and with the default, 0) end with the same pose bits, iteration counts and row counts."""
    pr = small_problem
    world = pr["world"]
    scans, inits = [], []
    for k in range(7):
        gt = (0.005 * k, -0.01, 0.2 + 0.15 * k, 2.5 - 0.8 * k, -2.0 + 0.6 * k, synth.SENSOR_HEIGHT)
        qc, qs, gt = synth.make_scan(world, 16, 300 + 150 * (k % 3), gt_pose=gt, seed=300 + k)
        scans.append((qc, qs))
        inits.append(synth.perturb_pose(gt, seed=400 + k))
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set_batch(scans)
    ref = None
    for in_flight in (0, 1, 2, 3, 7, 64):
        opts = ctx.default_opts()
        opts.scans_in_flight = in_flight
        _, poses, stats = ctx.run_batch(np.stack(inits), opts)
        sig = (bits(poses).tobytes(), [(s.status, s.iterations, s.n_rows, s.point_residuals) for s in stats])
        if ref is None:
            ref = sig
        assert sig == ref, in_flight


def test_survey_8b_alias_exports(ctx, pkg, synth, small_problem):
    """SURVEY 8(b)'s names for three entry points are exported aliases of the ones the tests above drive."""
    import ctypes as C
    from importlib import import_module
    capi = import_module("the-cooper-mapper_amd.capi")
    pr = small_problem
    ctx.map_set(pr["map_corner"], pr["map_surf"])
    ctx.scan_set(pr["corner"], pr["surf"])
    ref = ctx.sweep(pr["init_pose"], jtj_mode=1)
    n = ctx.n_scan
    coeff, valid, jtj = np.zeros((n, 4), np.float32), np.zeros(n, np.uint8), np.zeros(27, np.float32)
    p = np.array(pr["init_pose"], np.float32)
    rc = ctx.lib.lslam_residuals(ctx.h, p.ctypes.data_as(capi.c_float_p), coeff.ctypes.data_as(capi.c_float_p),
                                 valid.ctypes.data_as(capi.c_uint8_p), jtj.ctypes.data_as(capi.c_float_p))
    assert rc == 0
    assert np.array_equal(bits(coeff), bits(ref["coeff"])) and np.array_equal(valid, ref["flags"])
    assert np.array_equal(bits(jtj), bits(ref["sums"][:27]))
    # batch alias
    ctx.scan_set_batch([(pr["corner"], pr["surf"])] * 2)
    _, want, _ = ctx.run_batch(np.stack([p, p]))
    got = np.stack([p, p]).copy()
    st = (capi.LslamStats * 2)()
    assert ctx.lib.lslam_scanmatch_batch(ctx.h, 2, got.ctypes.data_as(capi.c_float_p), None, st) >= 0
    assert np.array_equal(bits(got), bits(want))
    # one-call pose graph
    g = synth.make_pose_graph(n_kf=120, n_loop=240, laps=2, radius=15.0)
    pg = pkg.PoseGraph(0)
    pg.set_graph(g["init"], g["ij"], g["meas"], g["info"])
    pg.optimize(5)
    want = pg.poses()
    poses = np.ascontiguousarray(g["init"], np.float64).copy()
    ij = np.ascontiguousarray(g["ij"], np.int32)
    meas = np.ascontiguousarray(g["meas"], np.float64)
    info = np.ascontiguousarray(g["info"], np.float64)
    pst = capi.LslamPgStats()
    dp = lambda a: a.ctypes.data_as(capi.c_double_p)
    rc = ctx.lib.lslam_posegraph_optimize(0, len(poses), dp(poses), len(ij), ij.ctypes.data_as(capi.c_int32_p), dp(meas),
                                          dp(info), 0, 5, C.byref(pst))
    assert rc >= 0 and pst.iterations == pg.last_stats.iterations
    assert np.array_equal(poses, want)


def test_reference_epoch_residency_follows_the_library(pkg, oracle, small_problem):
    """ScanMatch::setReferenceEpoch skips the map upload while the caller's clouds are resident -- which is the LIBRARY's to say
    (lslam_map_epoch): a map set by another path on the same context (setMap here; a FeatureMap, an odometry or ICP call do the
    same) drops the residency, and the next call uploads again instead of matching against the wrong map."""
    pr = small_problem
    sm = pkg.ScanMatch(10)
    sm.setReferenceEpoch(7)
    lib, h = sm.ctx.lib, sm.ctx.h
    ok0, pose0 = sm.scanMatchScan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])
    e0 = lib.lslam_map_epoch(h)
    assert e0 != 0
    ok1, pose1 = sm.scanMatchScan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])
    assert lib.lslam_map_epoch(h) == e0 and np.array_equal(bits(pose0), bits(pose1))  # resident: no map set
    # somebody else's map on the same context: half the world away
    far = np.array([500.0, 0.0, 0.0, 0.0], np.float32)
    sm.ctx.map_set(pr["map_corner"] + far, pr["map_surf"] + far)
    assert lib.lslam_map_epoch(h) != e0
    ok2, pose2 = sm.scanMatchScan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])
    assert ok2 == ok0 and np.array_equal(bits(pose0), bits(pose2))  # uploaded again: the same answer, not a match against the far map
    e2 = lib.lslam_map_epoch(h)
    assert e2 not in (0, e0)
    sm.scanMatchScan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], pr["init_pose"])
    assert lib.lslam_map_epoch(h) == e2  # and resident again
    sm.close() if hasattr(sm, "close") else None
