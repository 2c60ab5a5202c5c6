"""Host-side mirror of the reference's scan-match interface over the C ABI.

``ScanMatch`` keeps the method names, argument meaning and return behaviour of
``lidar_slam::ScanMatch`` (/root/reference/L_SLAM/src/scan_to_scan_match/
ScanMatch.h:21-61, ScanMatch.cpp:21-398); ``Context`` is the thin handle over
``include/lslam_c.h`` that tests and bench.py drive directly.
"""
import ctypes as C

import numpy as np

from .capi import (LslamError, LslamMapInfo, LslamOpts, LslamStats, Status, c_float_p, c_int32_p,
                   c_uint8_p, load_library)


def _cloud(a):
    """(n, >=3) array -> C-contiguous float32 and its stride in bytes."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("cloud must be (n, >=3) float32, got %r" % (a.shape,))
    return a, a.shape[1] * 4


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _fp(a):
    return a.ctypes.data_as(c_float_p)


class Comm:
    """lslam_comm: the library's own RCCL communicator (include/lslam_c.h "collectives").  Rank 0 calls
    ``Comm.unique_id()`` and the host program hands the 128 bytes to every rank."""

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = np.zeros(128, np.uint8)
        rc = lib.lslam_comm_unique_id(buf.ctypes.data_as(c_uint8_p))
        if rc != 0:
            raise LslamError(rc, lib.lslam_last_error().decode())
        return buf

    @staticmethod
    def version():
        """ncclGetVersion of the librccl the library loaded (22606 = 2.26.6)."""
        lib = load_library()
        v = C.c_int32(0)
        rc = lib.lslam_comm_version(C.byref(v))
        if rc != 0:
            raise LslamError(rc, lib.lslam_last_error().decode())
        return int(v.value)

    def __init__(self, device, unique_id, rank, world):
        self.lib = load_library()
        uid = np.ascontiguousarray(unique_id, np.uint8).reshape(128)
        h = C.c_void_p()
        rc = self.lib.lslam_comm_create(int(device), uid.ctypes.data_as(c_uint8_p), int(rank), int(world), C.byref(h))
        if rc != 0:
            raise LslamError(rc, self.lib.lslam_last_error().decode())
        self.h, self.rank, self.world = h, int(rank), int(world)

    def allreduce_f64(self, device_ptr, count, stream=None):
        rc = self.lib.lslam_comm_allreduce_f64(self.h, C.c_void_p(int(device_ptr)), int(count), C.c_void_p(stream or 0))
        if rc != 0:
            raise LslamError(rc, self.lib.lslam_last_error().decode())

    def info(self):
        """(rank, world) as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount through
        lslam_comm_info): what `bench.py --gpus N` prints as `rccl_ranks`."""
        r, w = C.c_int32(-1), C.c_int32(-1)
        rc = self.lib.lslam_comm_info(self.h, C.byref(r), C.byref(w))
        if rc != 0:
            raise LslamError(rc, self.lib.lslam_last_error().decode())
        return int(r.value), int(w.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.lslam_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One lslam_ctx: a HIP stream, an HBM-resident map and scan, the device GN state."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.lslam_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise LslamError(rc, self.lib.lslam_last_error().decode())
        self.h = h
        self.device = device
        self.n_scan = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.lslam_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise LslamError(rc, self.lib.lslam_last_error().decode())
        return rc

    def default_opts(self):
        o = LslamOpts()
        self.lib.lslam_default_opts(C.byref(o))
        return o

    def sweep_launches(self):
        """lslam_debug_sweep_launches: sweep launches of this context so far per kernel instantiation
        (capi.SWEEP_VARIANTS: 'deep', 'deep_ovf', 'shallow' = sweep_kernel<256,true,false,12>, ...)."""
        from .capi import SWEEP_VARIANTS
        out = (C.c_uint64 * 8)()
        self.lib.lslam_debug_sweep_launches(self.h, out)
        return dict(zip(SWEEP_VARIANTS, (int(v) for v in out)))

    def defer_trees(self, on=True):
        """lslam_map_defer_trees: maps handed over on the device from now on get their cell grids at once and their kd-trees
        on first need (the mapping node's per-frame map)."""
        self._check(self.lib.lslam_map_defer_trees(self.h, int(bool(on))))

    def lazy_trees(self):
        """lslam_debug_lazy_trees: (maps set with deferred trees, of those built after all, resident map's trees pending)."""
        out = (C.c_uint64 * 3)()
        self.lib.lslam_debug_lazy_trees(self.h, out)
        return int(out[0]), int(out[1]), bool(out[2])

    def grid_launches(self):
        """lslam_debug_grid_launches: grid sweeps (sweep_grid_kernel) this context has launched so far."""
        return int(self.lib.lslam_debug_grid_launches(self.h))

    def grid_wide_launches(self):
        """lslam_debug_grid_wide_launches: ... of which in the single-launch form of a map without trees."""
        return int(self.lib.lslam_debug_grid_wide_launches(self.h))

    def scratch(self, name, rows, cols, dtype=np.float32):
        """A host array of at least (rows, cols) kept with the context under `name` (grown by a quarter when it is too small):
        the output staging of calls whose result size is only known afterwards.  Allocating and freeing megabytes per call
        is an mmap / munmap pair each -- tens of microseconds, and now and then milliseconds, of a frame."""
        bufs = self.__dict__.setdefault("_scratch", {})
        a = bufs.get(name)
        if a is None or a.shape[0] < rows or a.shape[1] != cols or a.dtype != np.dtype(dtype):
            a = np.empty((max(1, rows + rows // 4), cols), dtype)
            bufs[name] = a
        return a

    def grid_cells(self):
        """lslam_debug_grid_cells: (cells of the corner table, cells of the surf table) of the resident map's cell grids."""
        out = (C.c_uint64 * 2)()
        self.lib.lslam_debug_grid_cells(self.h, out)
        return int(out[0]), int(out[1])

    def cert_stats(self):
        """lslam_debug_cert_stats: (points left to the second pass, points of certificate-testing workgroups, second-pass
        launches) of this context so far; the first two are counted only in runs with lslam_opts.debug_stats = 1 (the grid
        sweep counts the points it leaves to the tree search the same way)."""
        out = (C.c_uint64 * 3)()
        self.lib.lslam_debug_cert_stats(self.h, out)
        return int(out[0]), int(out[1]), int(out[2])

    def grid_stats(self):
        """lslam_debug_grid_stats: uint64 [2 types][8 sweeps][listed, swept] of the grid sweeps run with debug_stats = 1."""
        out = (C.c_uint64 * 32)()
        self.lib.lslam_debug_grid_stats(self.h, out)
        return np.array(list(out), np.uint64).reshape(2, 8, 2)

    def cert_state(self, n_points):
        """lslam_debug_cert_state: (positions of the last searches [n, 3], bounds [n]) of the resident scan points."""
        q = np.zeros((n_points, 4), np.float32)
        lb = np.zeros(n_points, np.float32)
        n = self.lib.lslam_debug_cert_state(self.h, q.ctypes.data_as(C.POINTER(C.c_float)), lb.ctypes.data_as(C.POINTER(C.c_float)), n_points)
        if n < 0:
            raise RuntimeError("lslam_debug_cert_state: %d" % n)
        return q[:n, :3], lb[:n]

    # -- map / scan ----------------------------------------------------------
    def map_set(self, corner, surf):
        c, sc = _cloud(corner)
        s, ss = _cloud(surf)
        if sc != ss:
            raise ValueError("corner/surf strides differ")
        self._check(self.lib.lslam_map_set(self.h, _vp(c), len(c), _vp(s), len(s), sc))

    def cubemap_set(self, corner, surf, cube_size=50.0, origin=(60, 60, 5), dims=(121, 121, 11)):
        """Per-cube trees as FeatureMap keeps them (util/FeatureMap.h; defaults: LaserMatcher.cpp:107-113)."""
        c, sc = _cloud(corner)
        s, ss = _cloud(surf)
        if sc != ss:
            raise ValueError("corner/surf strides differ")
        o = np.asarray(origin, np.int32)
        d = np.asarray(dims, np.int32)
        self._check(self.lib.lslam_cubemap_set(self.h, _vp(c), len(c), _vp(s), len(s), sc, float(cube_size),
                                               o.ctypes.data_as(c_int32_p), d.ctypes.data_as(c_int32_p)))

    def map_info(self):
        info = LslamMapInfo()
        self._check(self.lib.lslam_map_info_get(self.h, C.byref(info)))
        return info

    def scan_set(self, corner, surf):
        c, sc = _cloud(corner)
        s, ss = _cloud(surf)
        if sc != ss:
            raise ValueError("corner/surf strides differ")
        self._check(self.lib.lslam_scan_set(self.h, _vp(c), len(c), _vp(s), len(s), sc))
        self.n_scan = len(c) + len(s)

    def scan_set_batch(self, scans):
        """scans: list of (corner, surf) clouds made resident together (lslam_scan_set_batch)."""
        n = len(scans)
        keep = []
        cp = (C.c_void_p * n)()
        sp = (C.c_void_p * n)()
        nc = (C.c_size_t * n)()
        ns = (C.c_size_t * n)()
        stride = None
        for i, (corner, surf) in enumerate(scans):
            c, sc = _cloud(corner)
            s, ss = _cloud(surf)
            if sc != ss or (stride is not None and sc != stride):
                raise ValueError("all clouds of a batch must share one stride")
            stride = sc
            keep.append((c, s))
            cp[i], sp[i], nc[i], ns[i] = c.ctypes.data, s.ctypes.data, len(c), len(s)
        self._check(self.lib.lslam_scan_set_batch(self.h, n, cp, nc, sp, ns, stride))
        self.n_scan = sum(len(c) + len(s) for c, s in keep)
        self.n_batch = n

    def run_batch(self, poses, opts=None):
        """lslam_scanmatch_run_batch -> (worst status, poses (n,6), [stats])."""
        p = np.array(poses, dtype=np.float32).reshape(-1, 6)
        n = len(p)
        st = (LslamStats * n)()
        rc = self.lib.lslam_scanmatch_run_batch(self.h, n, _fp(p), C.byref(opts) if opts is not None else None, st)
        self._check(rc)
        return Status(rc), p, list(st)

    # -- GN loop ---------------------------------------------------------------
    def run(self, pose, opts=None):
        """lslam_scanmatch_run on the resident map+scan -> (status, pose, stats)."""
        p = np.array(pose, dtype=np.float32).reshape(6)
        st = LslamStats()
        rc = self.lib.lslam_scanmatch_run(self.h, _fp(p), C.byref(opts) if opts is not None else None,
                                          C.byref(st))
        self._check(rc)
        return Status(rc), p, st

    def set_comm(self, comm):
        """lslam_ctx_set_comm: attach the library's RCCL communicator (None detaches)."""
        self._check(self.lib.lslam_ctx_set_comm(self.h, comm.h if comm is not None else None))
        self._comm = comm

    def run_sharded(self, pose, allreduce=None, xchg=None, opts=None):
        """lslam_scanmatch_run_sharded: the resident scan is this rank's shard of one scan's
        points.  With a communicator attached (set_comm) nothing else is needed: the sums are
        all-reduced by RCCL on the library's stream.  Otherwise `allreduce(ptr, count)` sums `count`
        doubles at device address `ptr` over the ranks in place (e.g. torch.distributed.all_reduce on
        `xchg`, a 32-double CUDA tensor whose data_ptr() is what the library writes the sums to)
        -> (status, pose, stats)."""
        from .capi import ALLREDUCE_FN
        if allreduce is not None:
            def _tramp(_user, ptr, count):
                allreduce(ptr, count)
            cb = ALLREDUCE_FN(_tramp)
            ptr = C.c_void_p(xchg.data_ptr() if hasattr(xchg, "data_ptr") else int(xchg))
        else:
            cb = ALLREDUCE_FN(0)
            ptr = None
        p = np.array(pose, dtype=np.float32).reshape(6)
        st = LslamStats()
        rc = self.lib.lslam_scanmatch_run_sharded(self.h, _fp(p), C.byref(opts) if opts is not None else None,
                                                  cb, None, ptr, C.byref(st))
        self._check(rc)
        return Status(rc), p, st

    # ---- joint LiDAR + stereo system (BASELINE configs[4]; include/lslam_c.h) ----------------
    def default_stereo_cam(self):
        from .capi import LslamStereoCam
        cam = LslamStereoCam()
        self.lib.lslam_stereo_default_cam(C.byref(cam))
        return cam

    def stereo_set(self, landmarks, obs, inv_sigma2=None, cam=None):
        """lslam_stereo_set: landmarks (n,3) in the map frame, obs (n,3) = (uL, v, uR; uR < 0:
        monocular), inv_sigma2 (n,) or None.  The scan matches of a single resident scan then solve
        the joint system.  n = 0 removes the term."""
        lm = np.ascontiguousarray(landmarks, np.float32).reshape(-1, 3)
        ob = np.ascontiguousarray(obs, np.float32).reshape(-1, 3)
        if len(lm) != len(ob):
            raise ValueError("landmarks and observations differ in length")
        w = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32).reshape(len(lm))
        cam = cam if cam is not None else self.default_stereo_cam()
        self._check(self.lib.lslam_stereo_set(self.h, _fp(lm), _fp(ob), _fp(w) if w is not None else None,
                                              len(lm), C.byref(cam)))

    def stereo_clear(self):
        self._check(self.lib.lslam_stereo_clear(self.h))

    def stereo_sums(self, pose):
        """Parity tap: the 32 reduced sums of the stereo term alone at `pose`."""
        from .capi import c_double_p
        p = np.array(pose, dtype=np.float32).reshape(6)
        out = np.zeros(32)
        self._check(self.lib.lslam_stereo_sums(self.h, _fp(p), out.ctypes.data_as(c_double_p)))
        return out

    def scanmatch_scan(self, corner, surf, pose, opts=None):
        c, sc = _cloud(corner)
        s, _ = _cloud(surf)
        p = np.array(pose, dtype=np.float32).reshape(6)
        st = LslamStats()
        rc = self.lib.lslam_scanmatch_scan(self.h, _vp(c), len(c), _vp(s), len(s), sc, _fp(p),
                                           C.byref(opts) if opts is not None else None, C.byref(st))
        self._check(rc)
        self.n_scan = len(c) + len(s)
        return Status(rc), p, st

    def scanmatch_full(self, ref_corner, ref_surf, corner, surf, pose, opts=None):
        rc_, rs = _cloud(ref_corner)
        rs_, _ = _cloud(ref_surf)
        c, sc = _cloud(corner)
        s, _ = _cloud(surf)
        p = np.array(pose, dtype=np.float32).reshape(6)
        st = LslamStats()
        rc = self.lib.lslam_scanmatch_full(self.h, _vp(rc_), len(rc_), _vp(rs_), len(rs_), rs,
                                           _vp(c), len(c), _vp(s), len(s), sc, _fp(p),
                                           C.byref(opts) if opts is not None else None, C.byref(st))
        self._check(rc)
        return Status(rc), p, st

    def odometry_match(self, last_corner, last_surf, sharp, flat, pose, max_iterations=25, dt=0.1, dr=0.1, trees=False):
        """LaserOdometry::scanMatch (odometry/LaserOdometry.cpp:328-647); clouds {x,y,z,intensity}.  ``trees``: through
        kd-trees of the last clouds (lslam_odometry_match_trees) instead of the hashed cell grids -- same result."""
        lc, sb = _cloud(last_corner)
        ls, _ = _cloud(last_surf)
        sh, _ = _cloud(sharp)
        fl, _ = _cloud(flat)
        p = np.array(pose, dtype=np.float32).reshape(6)
        st = LslamStats()
        fn = self.lib.lslam_odometry_match_trees if trees else self.lib.lslam_odometry_match
        rc = fn(self.h, _vp(lc), len(lc), _vp(ls), len(ls), _vp(sh), len(sh),
                                           _vp(fl), len(fl), sb, _fp(p), int(max_iterations), float(dt),
                                           float(dr), C.byref(st))
        self._check(rc)
        return Status(rc), p, st

    def transform_to_end(self, cloud, pose):
        """LaserOdometry::transformToEnd (LaserOdometry.cpp:156-168) -> new (n, 4) cloud."""
        a = np.array(cloud, dtype=np.float32, order="C")
        if a.ndim != 2 or a.shape[1] not in (4, 8):
            raise ValueError("cloud must be (n, 4) {x,y,z,intensity} or (n, 8) PointXYZI")
        p = np.ascontiguousarray(pose, np.float32).reshape(6)
        self._check(self.lib.lslam_transform_to_end(self.h, _vp(a), len(a), a.shape[1] * 4, _fp(p)))
        return a

    def icp_align(self, target, source, guess, max_iterations=10, transformation_epsilon=0.0,
                  max_correspondence_distance=0.0):
        """lslam_icp_align: point-to-point ICP with PCL's defaults (LoopDetector::corseMatching,
        pose_graph/loop_detector.hpp:232-255) -> (T 4x4 float32, converged, iterations, fitness)."""
        from .capi import c_double_p
        t, st = _cloud(target)
        s, ss = _cloud(source)
        if st != ss:
            raise ValueError("target/source strides differ")
        T = np.array(guess, dtype=np.float32).reshape(16)
        fit = C.c_double(0.0)
        conv, its = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.lslam_icp_align(self.h, _vp(t), len(t), _vp(s), len(s), st, _fp(T), int(max_iterations),
                                             float(transformation_epsilon), float(max_correspondence_distance),
                                             C.byref(fit), C.byref(conv), C.byref(its)))
        return T.reshape(4, 4), bool(conv.value), its.value, fit.value

    # -- parity taps -------------------------------------------------------------
    def knn5(self, which_map, queries, search_mode=1, want_ties=False):
        """lslam_knn5_ex; search_mode 1 = one query per lane (nanoflann's traversal), 2 = packet search."""
        q, sq = _cloud(queries)
        idx = np.zeros((len(q), 5), np.int32)
        d2 = np.zeros((len(q), 5), np.float32)
        ties = C.c_int32(0)
        self._check(self.lib.lslam_knn5_ex(self.h, int(which_map), _vp(q), len(q), sq, int(search_mode),
                                           idx.ctypes.data_as(c_int32_p), _fp(d2), C.byref(ties)))
        return (idx, d2, ties.value) if want_ties else (idx, d2)

    def knn5_wide(self, which_map, queries, nf_margin=False):
        """lslam_debug_knn5_wide: the wide probe of a map without kd-trees -> (idx [n, 5], d2 [n, 5], undecided [n])."""
        q, sq = _cloud(queries)
        idx = np.zeros((len(q), 5), np.int32)
        d2 = np.zeros((len(q), 5), np.float32)
        und = np.zeros(len(q), np.uint8)
        self._check(self.lib.lslam_debug_knn5_wide(self.h, int(which_map), _vp(q), len(q), sq, int(bool(nf_margin)),
                                                   idx.ctypes.data_as(c_int32_p), _fp(d2), und.ctypes.data_as(c_uint8_p)))
        return idx, d2, und

    def sort_pairs(self, keys, values):
        """lslam_debug_sort_pairs: (uint64 keys, uint32 values) ascending by key, then value -> (keys, values)."""
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        v = np.ascontiguousarray(values, dtype=np.uint32)
        if k.shape != v.shape or k.ndim != 1:
            raise ValueError("keys and values: one-dimensional, equal length")
        ko, vo = np.zeros_like(k), np.zeros_like(v)
        self._check(self.lib.lslam_debug_sort_pairs(self.h, k.ctypes.data_as(C.POINTER(C.c_uint64)), v.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                    len(k), ko.ctypes.data_as(C.POINTER(C.c_uint64)), vo.ctypes.data_as(C.POINTER(C.c_uint32))))
        return ko, vo

    def sweep(self, pose, jtj_mode=0, taps=True, search_mode=1):
        p = np.array(pose, dtype=np.float32).reshape(6)
        n = self.n_scan
        sums = np.zeros(30, np.float32)
        if taps:
            idx = np.zeros((n, 5), np.int32)
            d2 = np.zeros((n, 5), np.float32)
            coeff = np.zeros((n, 4), np.float32)
            flags = np.zeros(n, np.uint8)
            self._check(self.lib.lslam_sweep_ex(self.h, _fp(p), jtj_mode, int(search_mode), idx.ctypes.data_as(c_int32_p),
                                                _fp(d2), _fp(coeff), flags.ctypes.data_as(c_uint8_p),
                                                _fp(sums)))
            return dict(idx=idx, d2=d2, coeff=coeff, flags=flags, sums=sums)
        self._check(self.lib.lslam_sweep_ex(self.h, _fp(p), jtj_mode, int(search_mode), None, None, None, None, _fp(sums)))
        return dict(sums=sums)

    def gn_step(self, AtA, Atb, it, pose, matP, degenerate, dr=0.05, dt=0.05):
        AtA = np.ascontiguousarray(AtA, np.float32).reshape(36)
        Atb = np.ascontiguousarray(Atb, np.float32).reshape(6)
        p = np.array(pose, dtype=np.float32).reshape(6)
        mp = np.array(matP, dtype=np.float32).reshape(36)
        deg = C.c_int32(int(degenerate))
        x = np.zeros(6, np.float32)
        dR, dT, conv = C.c_float(), C.c_float(), C.c_int32()
        self._check(self.lib.lslam_gn_step(self.h, _fp(AtA), _fp(Atb), int(it), _fp(p), _fp(mp),
                                           C.byref(deg), dr, dt, _fp(x), C.byref(dR), C.byref(dT),
                                           C.byref(conv)))
        return dict(converged=bool(conv.value), pose=p, matP=mp.reshape(6, 6),
                    degenerate=bool(deg.value), x=x, delta_r=dR.value, delta_t=dT.value)

    # -- Isometry <-> Twist (ScanMatch.cpp:349-360) --------------------------------
    def isometry_to_pose(self, T):
        T = np.ascontiguousarray(T, np.float32).reshape(16)
        p = np.zeros(6, np.float32)
        self.lib.lslam_isometry_to_pose(_fp(T), _fp(p))
        return p

    def pose_to_isometry(self, pose):
        p = np.ascontiguousarray(pose, np.float32).reshape(6)
        T = np.zeros(16, np.float32)
        self.lib.lslam_pose_to_isometry(_fp(p), _fp(T))
        return T.reshape(4, 4)


class ScanMatch:
    """Mirror of lidar_slam::ScanMatch (ScanMatch.h:12-86).

    Clouds are (n, >=3) float arrays (x, y, z first).  ``pose`` is either the
    6-vector Twist {rot_x, rot_y, rot_z, x, y, z} or a 4x4 Isometry3f; the method
    returns ``(success, pose)`` with pose in the form it was given, updated exactly
    when the reference writes it back (always, except "reference cloud points too few").
    """

    def __init__(self, maxIterations=10, device=0, ctx=None):
        self.ctx = ctx if ctx is not None else Context(device)
        if ctx is None:
            # scanMatchScan hands the reference clouds over on every call (the reference rebuilds both kd-trees inside): a
            # context of its own gets cell grids per map and trees only when a call needs them (lslam_map_defer_trees)
            self.ctx.defer_trees(True)
        self.opts = self.ctx.default_opts()
        self.opts.max_iterations = int(maxIterations)
        self._match_count = 0       # ScanMatch.h:84
        self._fail_match_count = 0  # ScanMatch.h:85
        self._total_score = 0.0     # ScanMatch.h:83
        self.last_stats = None
        self._ref_epoch = 0         # setReferenceEpoch: the caller's promise (0: none)
        self._resident = None       # (epoch, id / address / shape of the two reference clouds) the resident map was set under
        self._resident_map = 0      # lslam_map_epoch right after this object set that map (0: nothing resident)

    # setters, ScanMatch.h:21-34
    def setPercentThreshold(self, percent):
        self.opts.match_percentage_threshold = float(percent)

    def setScoreThreshold(self, score):
        self.opts.score_threshold = float(score)

    def setFineScore(self, enable):
        self.opts.fine_score = int(bool(enable))

    def setConvergeThreshold(self, deltaTAbort, deltaRAbort):
        self.opts.delta_t_abort = float(deltaTAbort)
        self.opts.delta_r_abort = float(deltaRAbort)

    def setUseCore(self, useScore):
        self.opts.use_score = int(bool(useScore))

    def setReferenceEpoch(self, epoch):
        """include/lslam_scan_match.hpp setReferenceEpoch: with a non-zero epoch the caller promises that the reference clouds
        it hands to scanMatchScan are unchanged while the epoch is; a call with the same two arrays (same buffers and shapes)
        under the same epoch skips their upload and matches against the resident map.  0: every call uploads (the default)."""
        self._ref_epoch = int(epoch)

    def getAverageScore(self):  # ScanMatch.h:59-61
        return self._total_score / self._match_count if self._match_count > 0 else 0.0

    def _finish(self, status, stats):
        self.last_stats = stats
        if status == Status.OK:  # ScanMatch.cpp:336-340
            self._total_score += stats.score
            self._match_count += 1
            return True
        if status != Status.TOO_FEW_REF:  # ScanMatch.cpp:325,332,344
            self._fail_match_count += 1
        return False

    def scanMatchScan(self, referenceCornerCloud, referenceSurfCloud, CornerCloud, SurfCloud, pose):
        """ScanMatch.cpp:51-360 (both overloads); rebuilds the map trees per call (quirk Q4)."""
        pose = np.asarray(pose, dtype=np.float32)
        iso = pose.shape == (4, 4)
        tw = self.ctx.isometry_to_pose(pose) if iso else pose.reshape(6)
        key = None
        if self._ref_epoch and isinstance(referenceCornerCloud, np.ndarray) and isinstance(referenceSurfCloud, np.ndarray):
            key = (self._ref_epoch,) + tuple((a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str)
                                             for a in (referenceCornerCloud, referenceSurfCloud))
        # resident: the caller's promise (epoch, buffers) AND the library's word that the map this object set is still the
        # context's (setMap, a FeatureMap, an odometry / ICP call on the same context replace it: lslam_map_epoch changes)
        if key is not None and key == self._resident and self._resident_map and \
                self.ctx.lib.lslam_map_epoch(self.ctx.h) == self._resident_map:
            try:
                status, tw, st = self.ctx.scanmatch_scan(CornerCloud, SurfCloud, tw, self.opts)
            except LslamError:
                self._resident, self._resident_map = None, 0
                raise
        else:
            self._resident, self._resident_map = None, 0
            status, tw, st = self.ctx.scanmatch_full(referenceCornerCloud, referenceSurfCloud,
                                                     CornerCloud, SurfCloud, tw, self.opts)
            if status >= 0 and status != Status.TOO_FEW_REF:
                self._resident, self._resident_map = key, int(self.ctx.lib.lslam_map_epoch(self.ctx.h))
        ok = self._finish(status, st)
        return ok, (self.ctx.pose_to_isometry(tw) if iso else tw)

    def scanMatchLocal(self, referenceCornerCloud, referenceSurfCloud, CornerCloud, SurfCloud, pose):
        """ScanMatch.cpp:362-398: pcl::VoxelGrid all four clouds (corner leaf 0.2, surf leaf 0.4,
        ScanMatch.cpp:29-30), then scanMatchScan on the downsampled clouds.  Clouds are (n, 4)
        {x, y, z, intensity} (or (n, 8) pcl::PointXYZI)."""
        from .feature_map import voxel_grid
        ds = [voxel_grid(self.ctx, c, leaf) for c, leaf in ((referenceCornerCloud, 0.2), (referenceSurfCloud, 0.4),
                                                            (CornerCloud, 0.2), (SurfCloud, 0.4))]
        keep, self._ref_epoch = self._ref_epoch, 0  # (fresh downsampled arrays every call: nothing to keep resident)
        try:
            return self.scanMatchScan(ds[0], ds[1], ds[2], ds[3], pose)
        finally:
            self._ref_epoch = keep

    def setMap(self, referenceCornerCloud, referenceSurfCloud):
        """Keep a map resident across calls (the FeatureMap::scanMatchScan usage,
        util/FeatureMap.h:490-691, where trees are built once per map update)."""
        self._resident, self._resident_map = None, 0
        self.ctx.map_set(referenceCornerCloud, referenceSurfCloud)

    def scanMatchResident(self, CornerCloud, SurfCloud, pose):
        pose = np.asarray(pose, dtype=np.float32)
        iso = pose.shape == (4, 4)
        tw = self.ctx.isometry_to_pose(pose) if iso else pose.reshape(6)
        status, tw, st = self.ctx.scanmatch_scan(CornerCloud, SurfCloud, tw, self.opts)
        ok = self._finish(status, st)
        return ok, (self.ctx.pose_to_isometry(tw) if iso else tw)
