"""Cross-checks of the oracle's Eigen-style algebra and geometry against
numpy/LAPACK fp64 and analytic properties (this part of the oracle is "parity
unpinned": the reference has no tests and Eigen is not available)."""
import numpy as np


def test_eig_sym3_vs_numpy(oracle):
    rng = np.random.default_rng(1)
    for _ in range(500):
        B = rng.normal(size=(5, 3)) * rng.uniform(0.01, 2, 3)
        A = (B.T @ B / 5).astype(np.float32)
        e, V = oracle.eig_sym3(A)
        ee = np.linalg.eigvalsh(A.astype(np.float64))
        assert np.all(np.diff(e) >= 0)
        assert np.abs(e - ee).max() <= 2e-6 * max(1e-9, np.abs(ee).max())
        assert np.abs(A.astype(np.float64) @ V - V * e).max() <= 3e-6 * np.abs(ee).max()
        assert np.abs(V.T @ V - np.eye(3)).max() < 3e-6


def test_eig_sym3_reads_lower_triangle_only(oracle):
    A = np.array([[2, 99, 99], [0.5, 3, 99], [0.1, 0.2, 1]], np.float32)
    S = np.tril(A) + np.tril(A, -1).T
    e1, _ = oracle.eig_sym3(A)
    e2, _ = oracle.eig_sym3(S)
    assert np.array_equal(e1, e2)


def test_eig_sym6_vs_numpy(oracle):
    rng = np.random.default_rng(2)
    for _ in range(200):
        J = rng.normal(size=(300, 6)) * rng.uniform(0.1, 30, 6)
        A = (J.T @ J).astype(np.float32)
        e, V = oracle.eig_sym6(A)
        ee = np.linalg.eigvalsh(A.astype(np.float64))
        assert np.abs(e - ee).max() <= 5e-6 * np.abs(ee).max()
        assert np.abs(A.astype(np.float64) @ V - V * e).max() <= 5e-6 * np.abs(ee).max()


def test_qr_solves_vs_numpy(oracle):
    rng = np.random.default_rng(3)
    for _ in range(300):
        A = rng.normal(size=(6, 6)).astype(np.float32)
        A = (A @ A.T + np.eye(6)).astype(np.float32)
        b = rng.normal(size=6).astype(np.float32)
        x = oracle.qr_solve(A, b)
        xx = np.linalg.solve(A.astype(np.float64), b)
        assert np.abs(x - xx).max() <= 2e-5 * np.abs(xx).max()
    for _ in range(300):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        P = rng.normal(size=(5, 3)) * 3
        P -= np.outer(P @ n - 2.0, n)  # points on the plane n.p = 2
        P = P.astype(np.float32)
        b = -np.ones(5, np.float32)
        x = oracle.qr_solve(P, b)
        xx = np.linalg.lstsq(P.astype(np.float64), b, rcond=None)[0]
        assert np.abs(x - xx).max() <= 1e-4 * np.abs(xx).max()


def test_qr_rank_deficient_returns_basic_solution(oracle):
    # a zero column is detected as a zero pivot and its unknown is set to 0
    A = np.zeros((6, 6), np.float32)
    A[:5, :5] = np.diag([5, 4, 3, 2, 1])
    b = np.array([5, 8, 9, 8, 5, 0], np.float32)
    x = oracle.qr_solve(A, b)
    assert np.allclose(x, [1, 2, 3, 4, 5, 0], atol=1e-6)


def test_inverse6(oracle):
    rng = np.random.default_rng(4)
    A = rng.normal(size=(6, 6)).astype(np.float32)
    assert np.abs(oracle.inverse6(A) - np.linalg.inv(A.astype(np.float64))).max() < 1e-4


def test_pose_to_Rt_and_back(oracle, synth):
    rng = np.random.default_rng(5)
    for _ in range(50):
        p = rng.uniform(-1, 1, 6).astype(np.float32)
        R, t = oracle.pose_to_Rt(p)
        R64, t64 = synth.pose_to_Rt(p.astype(np.float64))
        assert np.abs(R - R64).max() < 3e-7
        assert np.array_equal(t, p[3:])
        assert np.abs(oracle.Rt_to_pose(R, t) - p).max() < 1e-6


def test_find_line_and_corner_coeff(oracle):
    rng = np.random.default_rng(6)
    d = np.array([0.2, -0.1, 1.0])
    d /= np.linalg.norm(d)
    c0 = np.array([3.0, -2.0, 1.0])
    s = np.array([-0.4, -0.2, 0.0, 0.2, 0.4])
    pts = (c0[None] + s[:, None] * d[None] + rng.normal(0, 0.003, (5, 3))).astype(np.float32)
    ok, A, B = oracle.find_line(pts, np.arange(5, dtype=np.int32))
    assert ok
    v = (B - A) / np.linalg.norm(B - A)
    assert abs(abs(v @ d) - 1) < 1e-3
    assert abs(np.linalg.norm(B - A) - 0.2) < 1e-5  # c -/+ 0.1 v
    X = (c0 + 0.3 * np.cross(d, [1, 0, 0]) / np.linalg.norm(np.cross(d, [1, 0, 0]))).astype(np.float32)
    ok2, coeff = oracle.corner_coeff(A, B, X)
    dist = np.linalg.norm(np.cross(X - A, X - B)) / np.linalg.norm(A - B)
    w = 1 - 0.9 * dist
    assert ok2 and abs(coeff[3] - w * dist) < 1e-5
    # coeff.xyz = w * unit vector pointing from the line to X
    u = coeff[:3] / w
    assert abs(np.linalg.norm(u) - 1) < 1e-4 and abs(u @ v) < 1e-3 and u @ (X - A) > 0
    # swapping A and B (eigenvector sign) changes nothing
    _, coeff_sw = oracle.corner_coeff(B, A, X)
    assert np.allclose(coeff, coeff_sw, atol=1e-6)
    # isotropic blob: no line
    blob = rng.normal(0, 0.2, (5, 3)).astype(np.float32)
    assert not oracle.find_line(blob, np.arange(5, dtype=np.int32))[0]


def test_find_plane_and_surf_coeff(oracle):
    rng = np.random.default_rng(7)
    n = np.array([0.1, 0.2, 1.0])
    n /= np.linalg.norm(n)
    P = rng.uniform(-1, 1, (5, 3))
    P -= np.outer(P @ n - 4.0, n)  # n.p = 4
    pts = P.astype(np.float32)
    ok, pl = oracle.find_plane(pts, np.arange(5, dtype=np.int32))
    assert ok
    # solving [x y z] n' = -1 gives n' = -n/4: unit normal is -n, D = -n'.centroid = +4
    assert abs(np.linalg.norm(pl[:3]) - 1) < 1e-5
    assert np.allclose(pl[:3], -n, atol=1e-4) and abs(pl[3] - 4.0) < 1e-3
    X = (P.mean(0) + 0.1 * n).astype(np.float32)
    ok2, coeff = oracle.surf_coeff(pl, X)
    d = float(pl[:3] @ X + pl[3])
    w = 1 - 0.9 * abs(d) / np.sqrt(np.linalg.norm(X))
    assert ok2 and abs(d + 0.1) < 1e-3 and abs(coeff[3] - w * d) < 1e-5
    # a point 0.3 m off the plane breaks the 0.2 m inlier test (feature_utils.h:194-201)
    u = np.cross(n, [1, 0, 0]); u /= np.linalg.norm(u)
    v = np.cross(n, u)
    quad = np.array([4 * n + a * u + b * v for a, b in ((-1, -1), (-1, 1), (1, -1), (1, 1), (0, 0))])
    assert oracle.find_plane(quad.astype(np.float32), np.arange(5, dtype=np.int32))[0]
    quad[4] += 1.0 * n  # centre point lifted 1 m: residual 0.8 m at the centre
    assert not oracle.find_plane(quad.astype(np.float32), np.arange(5, dtype=np.int32))[0]


def test_jacobian_row_matches_finite_differences_except_quirk(oracle, synth):
    """d(R p + t)/d(rx,ry,rz,t) . coeff; the reference's arz middle term lacks
    parentheses (quirk Q1, ScanMatch.cpp:194): check it separately."""
    rng = np.random.default_rng(8)
    pose = rng.uniform(-0.5, 0.5, 6)
    p = rng.uniform(-10, 10, 3)
    c = rng.normal(size=3)
    coeff = np.array([*c, 0.37], np.float32)
    sc = np.array([np.sin(pose[0]), np.cos(pose[0]), np.sin(pose[1]), np.cos(pose[1]),
                   np.sin(pose[2]), np.cos(pose[2])], np.float32)
    row, b = oracle.jacobian_row(sc, p.astype(np.float32), coeff)

    def f(q):
        R, t = synth.pose_to_Rt(q)
        return c @ (R @ p + t)
    fd = np.zeros(6)
    for k in range(6):
        e = np.zeros(6)
        e[k] = 1e-6
        fd[k] = (f(pose + e) - f(pose - e)) / 2e-6
    assert abs(b + 0.37) < 1e-7
    assert np.allclose(row[[0, 1, 3, 4, 5]], fd[[0, 1, 3, 4, 5]], rtol=2e-4, atol=2e-4)
    srx, crx, sry, cry, srz, crz = sc.astype(np.float64)
    correct = (crz * sry * crx + srz * srx) * p[2]
    as_written = crz * sry * crx + srz * srx * p[2]
    assert abs((row[2] - fd[2]) - (as_written - correct) * c[1]) < 5e-4


def test_gn_loop_converges_to_ground_truth(oracle, small_problem):
    pr = small_problem
    ok, pose, st = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                         pr["init_pose"])
    assert ok and st.converged and st.status == 0 and not st.degenerate
    assert np.abs(pose[3:] - pr["gt_pose"][3:]).max() < 0.02
    assert np.abs(pose[:3] - pr["gt_pose"][:3]).max() < 2e-3
    assert st.n_rows <= st.n_line + st.n_plane <= len(pr["corner"]) + len(pr["surf"])


def test_gn_loop_guards(oracle, small_problem):
    pr = small_problem
    # too few reference points (ScanMatch.cpp:57-61): pose untouched
    ok, pose, st = oracle.scanmatch_scan(pr["map_corner"][:49], pr["map_surf"], pr["corner"], pr["surf"],
                                         pr["init_pose"])
    assert not ok and st.status == 1 and np.array_equal(pose, pr["init_pose"])
    # too few matches (ScanMatch.cpp:141-145): loop breaks, returns false, pose unchanged
    far = pr["init_pose"].copy()
    far[3] += 500
    ok, pose, st = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"], far)
    assert not ok and st.status == 2 and st.iterations == 0 and np.array_equal(pose, far)
    # score gate off (LaserMatcher.cpp:95): always false, pose still updated (quirk Q7)
    opts = oracle.default_opts()
    opts.use_score = 0
    ok, pose, st = oracle.scanmatch_scan(pr["map_corner"], pr["map_surf"], pr["corner"], pr["surf"],
                                         pr["init_pose"], opts)
    assert not ok and st.converged and not np.array_equal(pose, pr["init_pose"])
