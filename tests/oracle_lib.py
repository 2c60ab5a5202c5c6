"""ctypes bindings for the CPU oracle (oracle/liblslam_oracle.so) and, when it was
built, the reference's own nanoflann (oracle/_ref/libref_nanoflann.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class OracleOpts(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("delta_t_abort", C.c_float),
        ("delta_r_abort", C.c_float),
        ("use_score", C.c_int),
        ("fine_score", C.c_int),
        ("score_threshold", C.c_double),
        ("match_percentage_threshold", C.c_double),
    ]


class OracleCubeGrid(C.Structure):
    _fields_ = [("cube_size", C.c_float), ("origin", C.c_int32 * 3), ("dims", C.c_int32 * 3)]


class OracleOdomOpts(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("delta_t_abort", C.c_float), ("delta_r_abort", C.c_float)]


class OracleRegParams(C.Structure):
    _fields_ = [("n_feature_regions", C.c_int), ("curvature_region", C.c_int), ("max_corner_sharp", C.c_int),
                ("max_surface_flat", C.c_int), ("less_flat_filter_size", C.c_float),
                ("surface_curvature_threshold", C.c_float), ("blind_threshold", C.c_float)]


class OracleStereoCam(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("T_cl", C.c_float * 12), ("weight", C.c_float), ("huber_stereo", C.c_float),
                ("huber_mono", C.c_float), ("gate_outliers", C.c_int32), ("min_depth", C.c_float)]


class OracleStereo(C.Structure):
    _fields_ = [("landmarks", C.POINTER(C.c_float)), ("obs", C.POINTER(C.c_float)),
                ("inv_sigma2", C.POINTER(C.c_float)), ("n", C.c_size_t), ("cam", OracleStereoCam)]


class OracleStats(C.Structure):
    _fields_ = [
        ("status", C.c_int),
        ("iterations", C.c_int),
        ("n_line", C.c_int),
        ("n_plane", C.c_int),
        ("n_rows", C.c_int),
        ("degenerate", C.c_int),
        ("converged", C.c_int),
        ("delta_r", C.c_float),
        ("delta_t", C.c_float),
        ("score", C.c_double),
        ("percent", C.c_double),
        ("t_build", C.c_double),
        ("t_sweep", C.c_double),
        ("t_solve", C.c_double),
        ("point_residuals", C.c_longlong),
        ("score2", C.c_double),
        ("percent2", C.c_double),
    ]


def build_oracle(native=False):
    """Compile the oracle with its Makefile if the .so is missing or stale.  native: -O3
    -march=native (CPU baseline); native == "omp": that plus the OpenMP sweep (all-cores bound)."""
    if os.environ.get("LSLAM_ORACLE_SANITIZE") == "1":  # tools/run_sanitized_oracle_tests.sh: the ASan + UBSan build (make -C oracle asan)
        return os.path.join(ORACLE_DIR, "_asan", "liblslam_oracle.so")
    target = {False: "liblslam_oracle.so", True: "liblslam_oracle_native.so", "omp": "liblslam_oracle_omp.so"}[native]
    so = os.path.join(ORACLE_DIR, target)
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("lslam_oracle.c", "lslam_oracle.h", "fmap_oracle.c", "fmap_oracle.h", "features_oracle.c",
                                                 "features_oracle.h")]
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, target], stdout=subprocess.DEVNULL)
    return so


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _ip(a):
    return a.ctypes.data_as(c_int32_p)


def as_cloud(a):
    """(n,3|4|8) float array -> contiguous float32 (n,S) and stride in floats."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a, a.shape[1]


class Oracle:
    def __init__(self, native=False):
        self.lib = C.CDLL(build_oracle(native))
        L = self.lib
        L.oracle_kdtree_build.restype = C.c_void_p
        L.oracle_kdtree_build.argtypes = [c_float_p, C.c_size_t, C.c_size_t]
        L.oracle_kdtree_free.argtypes = [C.c_void_p]
        L.oracle_kdtree_knn.restype = C.c_int
        L.oracle_kdtree_knn.argtypes = [C.c_void_p, c_float_p, C.c_int, c_int32_p, c_float_p]
        L.oracle_kdtree_num_nodes.restype = C.c_size_t
        L.oracle_kdtree_num_nodes.argtypes = [C.c_void_p]
        L.oracle_kdtree_max_depth.restype = C.c_int
        L.oracle_kdtree_max_depth.argtypes = [C.c_void_p]
        L.oracle_kdtree_vind.argtypes = [C.c_void_p, c_int32_p]
        L.oracle_kdtree_node.argtypes = [C.c_void_p, C.c_size_t, c_int32_p, c_int32_p, c_int32_p,
                                         c_float_p, c_float_p, c_int32_p]
        L.oracle_eig_sym3.argtypes = [c_float_p] * 3
        L.oracle_eig_sym6.argtypes = [c_float_p] * 3
        L.oracle_qr_solve_5x3.argtypes = [c_float_p] * 3
        L.oracle_qr_solve_6x6.argtypes = [c_float_p] * 3
        L.oracle_inverse6.argtypes = [c_float_p] * 2
        L.oracle_pose_to_Rt.argtypes = [c_float_p] * 3
        L.oracle_Rt_to_pose.argtypes = [c_float_p] * 3
        L.oracle_transform_point.argtypes = [c_float_p] * 4
        L.oracle_find_line.restype = C.c_int
        L.oracle_find_line.argtypes = [c_float_p, C.c_size_t, c_int32_p, c_float_p, c_float_p]
        L.oracle_corner_coeff.restype = C.c_int
        L.oracle_corner_coeff.argtypes = [c_float_p] * 4
        L.oracle_find_plane.restype = C.c_int
        L.oracle_find_plane.argtypes = [c_float_p, C.c_size_t, c_int32_p, C.c_float, c_float_p]
        L.oracle_surf_coeff.restype = C.c_int
        L.oracle_surf_coeff.argtypes = [c_float_p] * 3
        L.oracle_jacobian_row.argtypes = [c_float_p] * 5
        L.oracle_default_opts.argtypes = [C.POINTER(OracleOpts)]
        L.oracle_sweep.argtypes = [C.c_void_p, c_float_p, C.c_void_p, c_float_p, C.c_size_t,
                                   c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_size_t,
                                   c_float_p, c_int32_p, c_float_p, c_float_p, c_uint8_p, c_float_p]
        L.oracle_scanmatch_scan.restype = C.c_int
        L.oracle_scanmatch_scan.argtypes = [c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_size_t,
                                            c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_size_t,
                                            c_float_p, C.POINTER(OracleOpts), C.POINTER(OracleStats)]
        L.oracle_stereo_sums.restype = None
        L.oracle_stereo_sums.argtypes = [C.POINTER(OracleStereo), c_float_p, c_float_p, c_float_p]
        L.oracle_scanmatch_joint.restype = C.c_int
        L.oracle_scanmatch_joint.argtypes = [c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_size_t,
                                             c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_size_t,
                                             C.POINTER(OracleStereo), c_float_p, C.POINTER(OracleOpts),
                                             C.POINTER(OracleStats), C.POINTER(C.c_int)]
        L.oracle_scanmatch_cubes.restype = C.c_int
        L.oracle_scanmatch_cubes.argtypes = [c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_size_t,
                                             C.POINTER(OracleCubeGrid), c_float_p, C.c_size_t, c_float_p,
                                             C.c_size_t, C.c_size_t, c_float_p, C.POINTER(OracleStats)]
        L.oracle_odometry_match.restype = C.c_int
        L.oracle_odometry_match.argtypes = [c_float_p, C.c_size_t, c_float_p, C.c_size_t, c_float_p, C.c_size_t,
                                            c_float_p, C.c_size_t, C.c_size_t, c_float_p,
                                            C.POINTER(OracleOdomOpts), C.POINTER(OracleStats)]
        L.oracle_transform_to_end.argtypes = [c_float_p, C.c_size_t, C.c_size_t, c_float_p]
        L.oracle_gn_step.restype = C.c_int
        L.oracle_gn_step.argtypes = [c_float_p, c_float_p, C.c_int, c_float_p, c_float_p,
                                     C.POINTER(C.c_int), C.c_float, C.c_float, C.c_float,
                                     c_float_p, c_float_p, c_float_p]

        L.oracle_voxel_grid.restype = C.c_size_t
        L.oracle_voxel_grid.argtypes = [c_float_p, C.c_size_t, C.c_size_t, C.c_float, c_float_p]
        L.oracle_fmap_create.restype = C.c_void_p
        L.oracle_fmap_create.argtypes = [C.c_int] * 3
        L.oracle_fmap_free.argtypes = [C.c_void_p]
        L.oracle_fmap_setup_filter_size.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.oracle_fmap_setup_cube_size.argtypes = [C.c_void_p, C.c_float]
        L.oracle_fmap_setup_valid_distance.argtypes = [C.c_void_p, C.c_float]
        L.oracle_fmap_origin.argtypes = [C.c_void_p, c_int32_p]
        L.oracle_fmap_update.argtypes = [C.c_void_p, c_float_p]
        L.oracle_fmap_valid_cubes.restype = C.c_size_t
        L.oracle_fmap_valid_cubes.argtypes = [C.c_void_p, c_int32_p, C.c_size_t]
        L.oracle_fmap_add_feature_cloud.argtypes = [C.c_void_p, c_float_p, C.c_size_t, c_float_p, C.c_size_t,
                                                    C.c_size_t, c_float_p]
        L.oracle_fmap_get_surround.restype = C.c_size_t
        L.oracle_fmap_get_surround.argtypes = [C.c_void_p, C.c_int, c_float_p, C.c_size_t]
        L.oracle_fmap_cube_count.restype = C.c_size_t
        L.oracle_fmap_cube_count.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_fmap_get_full_map.restype = C.c_size_t
        L.oracle_fmap_get_full_map.argtypes = [C.c_void_p, c_float_p, C.c_size_t]

        L.oracle_reg_default_params.argtypes = [C.POINTER(OracleRegParams)]
        L.oracle_point_classify.restype = C.c_int
        L.oracle_point_classify.argtypes = [c_float_p, C.c_size_t, C.c_size_t, C.c_int]
        L.oracle_extract_features.argtypes = [c_float_p, C.c_size_t, C.c_size_t, C.c_size_t, c_int32_p, C.c_size_t,
                                              C.POINTER(OracleRegParams), c_float_p, c_float_p, c_float_p, c_float_p,
                                              C.POINTER(C.c_size_t), c_float_p, C.POINTER(C.c_int8),
                                              C.POINTER(C.c_int8)]

        L.oracle_multiscan_register.restype = C.c_size_t
        L.oracle_multiscan_register.argtypes = [c_float_p, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int,
                                                C.c_float, c_float_p, c_int32_p]

    def multiscan_register(self, cloud, lower_deg, upper_deg, n_rings, scan_period=0.1):
        a = np.ascontiguousarray(cloud, dtype=np.float32)
        out = np.zeros((len(a), 4), np.float32)
        ranges = np.zeros((n_rings, 2), np.int32)
        n = self.lib.oracle_multiscan_register(_fp(a), len(a), a.shape[1], lower_deg, upper_deg, n_rings, scan_period,
                                               _fp(out), _ip(ranges))
        return out[:n].copy(), ranges

    # ---- feature extraction --------------------------------------------
    def reg_params(self):
        p = OracleRegParams()
        self.lib.oracle_reg_default_params(C.byref(p))
        return p

    def point_classify(self, cloud, idx, curvature_region=5):
        a = np.ascontiguousarray(cloud, dtype=np.float32)
        return self.lib.oracle_point_classify(_fp(a), a.shape[1], int(idx), curvature_region)

    def extract_features(self, cloud, scan_ranges, params=None, curvature_field=3):
        """-> dict(sharp, less_sharp, flat, less_flat (n,4 each), curvature, picked, label)."""
        a = np.ascontiguousarray(cloud, dtype=np.float32)
        r = np.ascontiguousarray(scan_ranges, dtype=np.int32).reshape(-1, 2)
        if params is None:
            params = self.reg_params()
        n = len(a)
        outs = [np.zeros((n, 4), np.float32) for _ in range(4)]
        counts = (C.c_size_t * 4)()
        curv = np.zeros(n, np.float32)
        picked = np.zeros(n, np.int8)
        label = np.zeros(n, np.int8)
        self.lib.oracle_extract_features(_fp(a), n, a.shape[1], curvature_field, _ip(r), len(r), C.byref(params),
                                         _fp(outs[0]), _fp(outs[1]), _fp(outs[2]), _fp(outs[3]), counts, _fp(curv),
                                         picked.ctypes.data_as(C.POINTER(C.c_int8)),
                                         label.ctypes.data_as(C.POINTER(C.c_int8)))
        return dict(sharp=outs[0][:counts[0]].copy(), less_sharp=outs[1][:counts[1]].copy(),
                    flat=outs[2][:counts[2]].copy(), less_flat=outs[3][:counts[3]].copy(), curvature=curv,
                    picked=picked, label=label)

    # ---- map maintenance -----------------------------------------------
    def voxel_grid(self, cloud, leaf):
        a = np.ascontiguousarray(cloud, dtype=np.float32)
        out = np.zeros((len(a), 4), np.float32)
        n = self.lib.oracle_voxel_grid(_fp(a), len(a), a.shape[1], leaf, _fp(out))
        return out[:n].copy()

    def feature_map(self, w=21, h=11, d=21):
        return OracleFeatureMap(self, w, h, d)

    # ---- kd-tree -------------------------------------------------------
    def kdtree(self, pts):
        return OracleTree(self, pts)

    # ---- dense algebra -------------------------------------------------
    def eig_sym3(self, A):
        A = np.ascontiguousarray(A, np.float32).reshape(9)
        e = np.zeros(3, np.float32)
        V = np.zeros(9, np.float32)
        self.lib.oracle_eig_sym3(_fp(A), _fp(e), _fp(V))
        return e, V.reshape(3, 3)

    def eig_sym6(self, A):
        A = np.ascontiguousarray(A, np.float32).reshape(36)
        e = np.zeros(6, np.float32)
        V = np.zeros(36, np.float32)
        self.lib.oracle_eig_sym6(_fp(A), _fp(e), _fp(V))
        return e, V.reshape(6, 6)

    def qr_solve(self, A, b):
        A = np.ascontiguousarray(A, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        x = np.zeros(A.shape[1], np.float32)
        if A.shape == (5, 3):
            self.lib.oracle_qr_solve_5x3(_fp(A), _fp(b), _fp(x))
        elif A.shape == (6, 6):
            self.lib.oracle_qr_solve_6x6(_fp(A), _fp(b), _fp(x))
        else:
            raise ValueError(A.shape)
        return x

    def inverse6(self, A):
        A = np.ascontiguousarray(A, np.float32).reshape(36)
        o = np.zeros(36, np.float32)
        self.lib.oracle_inverse6(_fp(A), _fp(o))
        return o.reshape(6, 6)

    # ---- geometry --------------------------------------------------------
    def pose_to_Rt(self, pose):
        pose = np.ascontiguousarray(pose, np.float32)
        R = np.zeros(9, np.float32)
        t = np.zeros(3, np.float32)
        self.lib.oracle_pose_to_Rt(_fp(pose), _fp(R), _fp(t))
        return R.reshape(3, 3), t

    def Rt_to_pose(self, R, t):
        R = np.ascontiguousarray(R, np.float32).reshape(9)
        t = np.ascontiguousarray(t, np.float32)
        p = np.zeros(6, np.float32)
        self.lib.oracle_Rt_to_pose(_fp(R), _fp(t), _fp(p))
        return p

    def find_line(self, cloud, idx):
        cloud, s = as_cloud(cloud)
        idx = np.ascontiguousarray(idx, np.int32)
        A = np.zeros(3, np.float32)
        B = np.zeros(3, np.float32)
        ok = self.lib.oracle_find_line(_fp(cloud), s, _ip(idx), _fp(A), _fp(B))
        return bool(ok), A, B

    def corner_coeff(self, A, B, X):
        A, B, X = (np.ascontiguousarray(v, np.float32) for v in (A, B, X))
        c = np.zeros(4, np.float32)
        ok = self.lib.oracle_corner_coeff(_fp(A), _fp(B), _fp(X), _fp(c))
        return bool(ok), c

    def find_plane(self, cloud, idx, max_distance=0.2):
        cloud, s = as_cloud(cloud)
        idx = np.ascontiguousarray(idx, np.int32)
        pl = np.zeros(4, np.float32)
        ok = self.lib.oracle_find_plane(_fp(cloud), s, _ip(idx), max_distance, _fp(pl))
        return bool(ok), pl

    def surf_coeff(self, plane, X):
        plane, X = (np.ascontiguousarray(v, np.float32) for v in (plane, X))
        c = np.zeros(4, np.float32)
        ok = self.lib.oracle_surf_coeff(_fp(plane), _fp(X), _fp(c))
        return bool(ok), c

    def jacobian_row(self, sc, p, coeff):
        sc, p, coeff = (np.ascontiguousarray(v, np.float32) for v in (sc, p, coeff))
        row = np.zeros(6, np.float32)
        b = np.zeros(1, np.float32)
        self.lib.oracle_jacobian_row(_fp(sc), _fp(p), _fp(coeff), _fp(row), _fp(b))
        return row, float(b[0])

    # ---- sweep / full loop ------------------------------------------------
    def default_opts(self):
        o = OracleOpts()
        self.lib.oracle_default_opts(C.byref(o))
        return o

    def sweep(self, tree_c, tree_s, qc, qs, pose):
        qc, sq = as_cloud(qc)
        qs, sq2 = as_cloud(qs)
        assert sq == sq2 and tree_c.stride == tree_s.stride
        n = len(qc) + len(qs)
        pose = np.ascontiguousarray(pose, np.float32)
        idx = np.zeros((n, 5), np.int32)
        d2 = np.zeros((n, 5), np.float32)
        coeff = np.zeros((n, 4), np.float32)
        flags = np.zeros(n, np.uint8)
        sums = np.zeros(29, np.float32)
        self.lib.oracle_sweep(tree_c.h, _fp(tree_c.pts), tree_s.h, _fp(tree_s.pts), tree_c.stride,
                              _fp(qc), len(qc), _fp(qs), len(qs), sq, _fp(pose), _ip(idx), _fp(d2),
                              _fp(coeff), flags.ctypes.data_as(c_uint8_p), _fp(sums))
        return dict(idx=idx, d2=d2, coeff=coeff, flags=flags, sums=sums)

    def scanmatch_scan(self, map_c, map_s, qc, qs, pose, opts=None):
        map_c, sm = as_cloud(map_c)
        map_s, sm2 = as_cloud(map_s)
        qc, sq = as_cloud(qc)
        qs, sq2 = as_cloud(qs)
        assert sm == sm2 and sq == sq2
        pose = np.array(pose, np.float32)
        if opts is None:
            opts = self.default_opts()
        st = OracleStats()
        ok = self.lib.oracle_scanmatch_scan(_fp(map_c), len(map_c), _fp(map_s), len(map_s), sm,
                                            _fp(qc), len(qc), _fp(qs), len(qs), sq, _fp(pose),
                                            C.byref(opts), C.byref(st))
        return bool(ok), pose, st

    # ---- stereo term of the joint system (parity unpinned: no reference code) --------------------
    def _stereo(self, landmarks, obs, inv_sigma2, cam):
        lm = np.ascontiguousarray(landmarks, np.float32).reshape(-1, 3)
        ob = np.ascontiguousarray(obs, np.float32).reshape(-1, 3)
        w = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32).reshape(len(lm))
        oc = OracleStereoCam()
        for f, _ in OracleStereoCam._fields_:  # same field names as lslam_stereo_cam
            setattr(oc, f, getattr(cam, f))
        s = OracleStereo(_fp(lm), _fp(ob), _fp(w) if w is not None else None, len(lm), oc)
        s._keep = (lm, ob, w)
        return s

    def stereo_sums(self, landmarks, obs, inv_sigma2, cam, pose, want_rows=False):
        s = self._stereo(landmarks, obs, inv_sigma2, cam)
        pose = np.array(pose, np.float32)
        sums = np.zeros(29, np.float32)
        rows = np.zeros((s.n, 3, 7), np.float32) if want_rows else None
        self.lib.oracle_stereo_sums(C.byref(s), _fp(pose), _fp(sums), _fp(rows) if want_rows else None)
        return (sums, rows) if want_rows else sums

    def scanmatch_joint(self, map_c, map_s, qc, qs, landmarks, obs, inv_sigma2, cam, pose, opts=None):
        map_c, sm = as_cloud(map_c)
        map_s, _ = as_cloud(map_s)
        qc, sq = as_cloud(qc)
        qs, _ = as_cloud(qs)
        s = self._stereo(landmarks, obs, inv_sigma2, cam)
        pose = np.array(pose, np.float32)
        if opts is None:
            opts = self.default_opts()
        st = OracleStats()
        used = C.c_int(0)
        ok = self.lib.oracle_scanmatch_joint(_fp(map_c), len(map_c), _fp(map_s), len(map_s), sm,
                                             _fp(qc), len(qc), _fp(qs), len(qs), sq, C.byref(s), _fp(pose),
                                             C.byref(opts), C.byref(st), C.byref(used))
        return bool(ok), pose, st, used.value

    def scanmatch_cubes(self, map_c, map_s, qc, qs, pose, cube_size, origin, dims):
        map_c, sm = as_cloud(map_c)
        map_s, _ = as_cloud(map_s)
        qc, sq = as_cloud(qc)
        qs, _ = as_cloud(qs)
        pose = np.array(pose, np.float32)
        g = OracleCubeGrid(cube_size, (C.c_int32 * 3)(*origin), (C.c_int32 * 3)(*dims))
        st = OracleStats()
        ok = self.lib.oracle_scanmatch_cubes(_fp(map_c), len(map_c), _fp(map_s), len(map_s), sm, C.byref(g),
                                             _fp(qc), len(qc), _fp(qs), len(qs), sq, _fp(pose), C.byref(st))
        return bool(ok), pose, st

    def odometry_match(self, last_corner, last_surf, sharp, flat, pose, max_iterations=25, dt=0.1, dr=0.1):
        lc, s = as_cloud(last_corner)
        ls, _ = as_cloud(last_surf)
        sh, _ = as_cloud(sharp)
        fl, _ = as_cloud(flat)
        assert s >= 4
        pose = np.array(pose, np.float32)
        op = OracleOdomOpts(max_iterations, dt, dr)
        st = OracleStats()
        n = self.lib.oracle_odometry_match(_fp(lc), len(lc), _fp(ls), len(ls), _fp(sh), len(sh), _fp(fl), len(fl),
                                           s, _fp(pose), C.byref(op), C.byref(st))
        return n, pose, st

    def transform_to_end(self, cloud, pose):
        a = np.array(cloud, dtype=np.float32, order="C")
        p = np.ascontiguousarray(pose, np.float32).reshape(6)
        self.lib.oracle_transform_to_end(_fp(a), len(a), a.shape[1], _fp(p))
        return a

    def gn_step(self, AtA, Atb, it, pose, matP, degenerate, eig_thresh=100.0, dr=0.05, dt=0.05):
        AtA = np.ascontiguousarray(AtA, np.float32).reshape(36)
        Atb = np.ascontiguousarray(Atb, np.float32)
        pose = np.array(pose, np.float32)
        matP = np.array(matP, np.float32).reshape(36)
        deg = C.c_int(int(degenerate))
        x = np.zeros(6, np.float32)
        dR = np.zeros(1, np.float32)
        dT = np.zeros(1, np.float32)
        conv = self.lib.oracle_gn_step(_fp(AtA), _fp(Atb), it, _fp(pose), _fp(matP), C.byref(deg),
                                       eig_thresh, dr, dt, _fp(x), _fp(dR), _fp(dT))
        return dict(converged=bool(conv), pose=pose, matP=matP.reshape(6, 6), degenerate=bool(deg.value),
                    x=x, delta_r=float(dR[0]), delta_t=float(dT[0]))


class OracleFeatureMap:
    """oracle_fmap_* (oracle/fmap_oracle.c): FeatureMap<PointXYZI> restated on the CPU."""

    def __init__(self, oracle, w, h, d):
        self.L = oracle.lib
        self.h = C.c_void_p(self.L.oracle_fmap_create(w, h, d))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_fmap_free(self.h)
            self.h = None

    def setup_filter_size(self, corner, surf, map_leaf):
        self.L.oracle_fmap_setup_filter_size(self.h, corner, surf, map_leaf)

    def setup_world_cube_size(self, s):
        self.L.oracle_fmap_setup_cube_size(self.h, s)

    def setup_lidar_valid_distance(self, d):
        self.L.oracle_fmap_setup_valid_distance(self.h, d)

    def update(self, pos):
        p = np.ascontiguousarray(pos, dtype=np.float32).reshape(3)
        self.L.oracle_fmap_update(self.h, _fp(p))

    def add_feature_cloud(self, corner, surf, tf):
        c = np.ascontiguousarray(corner, dtype=np.float32)
        s = np.ascontiguousarray(surf, dtype=np.float32)
        T = np.ascontiguousarray(tf, dtype=np.float32).reshape(16)
        self.L.oracle_fmap_add_feature_cloud(self.h, _fp(c), len(c), _fp(s), len(s), c.shape[1], _fp(T))

    def get_surround_feature(self):
        out = []
        for which in (0, 1):
            n = self.L.oracle_fmap_get_surround(self.h, which, None, 0)
            a = np.zeros((n, 4), np.float32)
            self.L.oracle_fmap_get_surround(self.h, which, _fp(a), n)
            out.append(a)
        return tuple(out)

    def get_full_map(self):
        n = self.L.oracle_fmap_get_full_map(self.h, None, 0)
        a = np.zeros((n, 4), np.float32)
        self.L.oracle_fmap_get_full_map(self.h, _fp(a), n)
        return a

    def info(self):
        origin = np.zeros(3, np.int32)
        self.L.oracle_fmap_origin(self.h, _ip(origin))
        n = self.L.oracle_fmap_valid_cubes(self.h, None, 0)
        valid = np.zeros(n, np.int32)
        self.L.oracle_fmap_valid_cubes(self.h, _ip(valid), n)
        return dict(origin=origin, valid=valid)


class OracleTree:
    def __init__(self, oracle, pts):
        self.o = oracle
        self.pts, self.stride = as_cloud(pts)
        self.h = oracle.lib.oracle_kdtree_build(_fp(self.pts), len(self.pts), self.stride)

    def __del__(self):
        try:
            self.o.lib.oracle_kdtree_free(self.h)
        except Exception:
            pass

    def knn(self, q, k=5):
        q = np.ascontiguousarray(q, np.float32)
        nq = len(q)
        idx = np.zeros((nq, k), np.int32)
        d2 = np.zeros((nq, k), np.float32)
        for i in range(nq):
            qi = np.ascontiguousarray(q[i, :3])
            self.o.lib.oracle_kdtree_knn(self.h, _fp(qi), k, _ip(idx[i]), _fp(d2[i]))
        return idx, d2

    def num_nodes(self):
        return self.o.lib.oracle_kdtree_num_nodes(self.h)

    def max_depth(self):
        return self.o.lib.oracle_kdtree_max_depth(self.h)

    def vind(self):
        out = np.zeros(len(self.pts), np.int32)
        self.o.lib.oracle_kdtree_vind(self.h, _ip(out))
        return out

    def nodes(self):
        n = self.num_nodes()
        kind = np.zeros(n, np.int32)
        a = np.zeros(n, np.int32)
        b = np.zeros(n, np.int32)
        lo = np.zeros(n, np.float32)
        hi = np.zeros(n, np.float32)
        c2 = np.zeros(n, np.int32)
        k_, a_, b_, c_ = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        l_, h_ = C.c_float(), C.c_float()
        for i in range(n):
            self.o.lib.oracle_kdtree_node(self.h, i, C.byref(k_), C.byref(a_), C.byref(b_),
                                          C.byref(l_), C.byref(h_), C.byref(c_))
            kind[i], a[i], b[i], lo[i], hi[i], c2[i] = k_.value, a_.value, b_.value, l_.value, h_.value, c_.value
        return dict(kind=kind, a=a, b=b, divlow=lo, divhigh=hi, child2=c2)


# ---------------------------------------------------------------------------
# The reference's own nanoflann, when oracle/_ref was built (container only, or
# shipped prebuilt to the GPU box).
# ---------------------------------------------------------------------------
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_nanoflann.so")


def have_ref():
    return os.path.exists(REF_SO)


class RefNanoflann:
    def __init__(self, pts):
        self.lib = C.CDLL(REF_SO)
        self.lib.ref_kdtree_build.restype = C.c_void_p
        self.lib.ref_kdtree_build.argtypes = [c_float_p, C.c_size_t, C.c_size_t]
        self.lib.ref_kdtree_free.argtypes = [C.c_void_p]
        self.lib.ref_kdtree_knn_batch.argtypes = [C.c_void_p, c_float_p, C.c_size_t, C.c_size_t,
                                                  C.c_int, c_int32_p, c_float_p]
        self.pts, self.stride = as_cloud(pts)
        self.h = self.lib.ref_kdtree_build(_fp(self.pts), len(self.pts), self.stride)

    def __del__(self):
        try:
            self.lib.ref_kdtree_free(self.h)
        except Exception:
            pass

    def radius(self, q, radius, cap=4096):
        """KdTreeFLANN::radiusSearch of the reference (radius against squared distances, sorted)."""
        self.lib.ref_kdtree_radius.restype = C.c_int
        self.lib.ref_kdtree_radius.argtypes = [C.c_void_p, c_float_p, C.c_float, c_int32_p, c_float_p, C.c_int]
        q = np.ascontiguousarray(q, np.float32).reshape(-1)
        idx = np.zeros(cap, np.int32)
        d2 = np.zeros(cap, np.float32)
        n = self.lib.ref_kdtree_radius(self.h, _fp(q), float(radius), _ip(idx), _fp(d2), cap)
        assert n <= cap
        return idx[:n], d2[:n]

    def radius_as_wrapped(self, q, radius):
        """nFound of nanoflann_pcl.h:173 (the bool findNeighbors returns, as a count)."""
        self.lib.ref_kdtree_radius_nfound.restype = C.c_int
        self.lib.ref_kdtree_radius_nfound.argtypes = [C.c_void_p, c_float_p, C.c_float]
        q = np.ascontiguousarray(q, np.float32).reshape(-1)
        return self.lib.ref_kdtree_radius_nfound(self.h, _fp(q), float(radius))

    def knn(self, q, k=5):
        q = np.ascontiguousarray(q, np.float32)
        idx = np.zeros((len(q), k), np.int32)
        d2 = np.zeros((len(q), k), np.float32)
        self.lib.ref_kdtree_knn_batch(self.h, _fp(q), len(q), q.shape[1], k, _ip(idx), _fp(d2))
        return idx, d2
