"""Host-side mirror of ``lidar_slam::FeatureMap<PointXYZI>`` (/root/reference/L_SLAM/src/util/
FeatureMap.h:42-186) over the C ABI: the cube grid, ``addFeatureCloud`` with the per-cube
VoxelGrid, ``update`` (shift + active area) and ``getSurroundFeature`` -- resident in HBM
(``csrc/lslam_fmap.hip``).  Method names follow the reference in snake_case; clouds are
``(n, 4)`` float32 ``{x, y, z, intensity}``.
"""
import ctypes as C

import numpy as np

from .capi import LslamError, c_float_p, c_int32_p


def _xyzi(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] not in (4, 8):
        raise ValueError("cloud must be (n, 4) {x,y,z,intensity} or (n, 8) pcl::PointXYZI, got %r" % (a.shape,))
    return a


def _fp(a):
    return a.ctypes.data_as(c_float_p)


class FeatureMap:
    def __init__(self, ctx, cube_width=21, cube_height=11, cube_depth=21):
        self.ctx = ctx
        self.lib = ctx.lib
        h = C.c_void_p()
        rc = self.lib.lslam_fmap_create(ctx.h, int(cube_width), int(cube_height), int(cube_depth), C.byref(h))
        if rc != 0:
            raise LslamError(rc, self.lib.lslam_last_error().decode())
        self.h = h
        self.dims = (int(cube_width), int(cube_height), int(cube_depth))

    def close(self):
        if getattr(self, "h", None):
            self.lib.lslam_fmap_destroy(self.h)
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

    # ---- FeatureMap.h:72-91 ------------------------------------------------------------
    def setup_filter_size(self, corner, surf, map_leaf):
        self._check(self.lib.lslam_fmap_setup_filter_size(self.h, corner, surf, map_leaf))

    def setup_world_origin(self, ox, oy, oz):
        self._check(self.lib.lslam_fmap_setup_world_origin(self.h, int(ox), int(oy), int(oz)))

    def setup_world_cube_size(self, size):
        self._check(self.lib.lslam_fmap_setup_world_cube_size(self.h, size))

    def setup_lidar_valid_distance(self, dist):
        self._check(self.lib.lslam_fmap_setup_lidar_valid_distance(self.h, dist))

    # ---- FeatureMap.h:218-265 ----------------------------------------------------------
    def update(self, sensor_xyz):
        p = np.ascontiguousarray(sensor_xyz, dtype=np.float32).reshape(3)
        self._check(self.lib.lslam_fmap_update(self.h, _fp(p)))

    def add_feature_cloud(self, corner, surf, tf, wait=True):
        """``wait=False`` (lslam_fmap_add_feature_cloud_begin): enqueued only -- the rebuild is waited for and committed at the
        head of the next call on this map (or by :meth:`wait`); the clouds are copied out of the arrays before the call returns."""
        c, s = _xyzi(corner), _xyzi(surf)
        if c.shape[1] != s.shape[1]:
            raise ValueError("corner and surf clouds must share a point layout")
        T = np.ascontiguousarray(tf, dtype=np.float32).reshape(16)
        fn = self.lib.lslam_fmap_add_feature_cloud if wait else self.lib.lslam_fmap_add_feature_cloud_begin
        self._check(fn(self.h, c.ctypes.data_as(C.c_void_p), len(c), s.ctypes.data_as(C.c_void_p), len(s), c.shape[1] * 4, _fp(T)))

    def wait(self):
        self._check(self.lib.lslam_fmap_wait(self.h))

    def surround_counts(self):
        nc, ns = C.c_size_t(), C.c_size_t()
        self._check(self.lib.lslam_fmap_surround_counts(self.h, C.byref(nc), C.byref(ns)))
        return nc.value, ns.value

    def get_surround_feature(self):
        """-> (corner (n,4), surf (m,4)) on the host."""
        nc, ns = self.surround_counts()
        c = np.zeros((nc, 4), np.float32)
        s = np.zeros((ns, 4), np.float32)
        self._check(self.lib.lslam_fmap_get_surround(self.h, _fp(c), nc, _fp(s), ns))
        return c, s

    def surround_to_map(self):
        """The surround becomes ``ctx``'s map without leaving HBM (device kd-tree build)."""
        self._check(self.lib.lslam_fmap_surround_to_map(self.h))

    def surround_to_map_counts(self):
        """:meth:`surround_to_map`, handing back the sizes of the two surround clouds it reads anyway -> (n_corner, n_surf)."""
        nc, ns = C.c_size_t(), C.c_size_t()
        self._check(self.lib.lslam_fmap_surround_to_map_counts(self.h, C.byref(nc), C.byref(ns)))
        return nc.value, ns.value

    def to_cubemap(self):
        """The active area becomes ``ctx``'s variant-C map: one kd-tree per cube (FeatureMap.h:490-691)."""
        self._check(self.lib.lslam_fmap_to_cubemap(self.h))

    def cubemap_stats(self):
        """(trees built by the last to_cubemap(), trees it kept from earlier calls)."""
        b, r = C.c_int64(), C.c_int64()
        self._check(self.lib.lslam_fmap_cubemap_stats(self.h, C.byref(b), C.byref(r)))
        return b.value, r.value

    def rebuild_stats(self):
        """(add_feature_cloud rebuilds that merged the new points into the sorted arrays, rebuilds that sorted everything)."""
        m, r = C.c_int64(), C.c_int64()
        self._check(self.lib.lslam_fmap_rebuild_stats(self.h, C.byref(m), C.byref(r)))
        return m.value, r.value

    def cubemap_invalidate(self):
        self._check(self.lib.lslam_fmap_cubemap_invalidate(self.h))

    def scan_match_scan(self, corner, surf, pose, opts=None):
        """FeatureMap::scanMatchScan (util/FeatureMap.h:490-691): the scan against the per-cube trees of the active
        area (kept between calls, rebuilt only where the map changed) with the reference's settings -- at most 10
        iterations, thresholds 0.05 / 0.05, no score gate.  -> (status, pose, stats); the reference returns nothing."""
        self.to_cubemap()
        if opts is None:
            opts = self.ctx.default_opts()
            opts.max_iterations = 10
            opts.delta_t_abort = opts.delta_r_abort = 0.05
            opts.use_score = 0
        return self.ctx.scanmatch_scan(corner, surf, pose, opts)

    def get_full_map(self):
        n = C.c_size_t()
        self._check(self.lib.lslam_fmap_get_full_map(self.h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 4), np.float32)
        self._check(self.lib.lslam_fmap_get_full_map(self.h, _fp(out), n.value, C.byref(n)))
        return out

    # ---- FeatureMap.h:378-462 -------------------------------------------------------------
    def save_cloud_to_files(self, directory):
        """saveCloudToFiles: <count>.pcd (binary, x y z intensity) per non-empty cube and type + index.txt."""
        self._check(self.lib.lslam_fmap_save(self.h, str(directory).encode()))
        return True

    def load_cloud_from_files(self, directory):
        """loadCloudFromFiles: False (like the reference) when the directory has no index.txt."""
        import os
        if not os.path.exists(os.path.join(str(directory), "index.txt")):
            return False
        self._check(self.lib.lslam_fmap_load(self.h, str(directory).encode()))
        return True

    def info(self):
        origin = np.zeros(3, np.int32)
        nv = C.c_int32()
        tc, ts = C.c_size_t(), C.c_size_t()
        self._check(self.lib.lslam_fmap_info(self.h, origin.ctypes.data_as(c_int32_p), C.byref(nv), None, 0,
                                             C.byref(tc), C.byref(ts)))
        valid = np.zeros(nv.value, np.int32)
        self._check(self.lib.lslam_fmap_info(self.h, None, None, valid.ctypes.data_as(c_int32_p), nv.value,
                                             None, None))
        return dict(origin=origin, valid=valid, n_corner=tc.value, n_surf=ts.value)


def voxel_grid(ctx, cloud, leaf):
    """pcl::VoxelGrid<PointXYZI> with a cubic leaf (lslam_voxel_grid) -> (m, 4) centroids."""
    a = _xyzi(cloud)
    out = ctx.scratch("voxel_grid", len(a), 4)
    n = C.c_size_t()
    rc = ctx.lib.lslam_voxel_grid(ctx.h, a.ctypes.data_as(C.c_void_p), len(a), a.shape[1] * 4, float(leaf),
                                  _fp(out), len(a), C.byref(n))
    if rc < 0:
        raise LslamError(rc, ctx.lib.lslam_last_error().decode())
    return out[:n.value].copy()


def voxel_grid2(ctx, cloud_a, cloud_b, leaf):
    """Two clouds through pcl::VoxelGrid with the same leaf in one pass (lslam_voxel_grid2: prepareFeatureFrame's corner and
    surface clouds) -> ((ma, 4), (mb, 4)); bit for bit what two voxel_grid calls return."""
    a, b = _xyzi(cloud_a), _xyzi(cloud_b)
    if a.shape[1] != b.shape[1]:
        raise ValueError("both clouds must have the same layout")
    oa = ctx.scratch("voxel_grid2a", len(a), 4)
    ob = ctx.scratch("voxel_grid2b", len(b), 4)
    na, nb = C.c_size_t(), C.c_size_t()
    rc = ctx.lib.lslam_voxel_grid2(ctx.h, a.ctypes.data_as(C.c_void_p), len(a), b.ctypes.data_as(C.c_void_p), len(b), a.shape[1] * 4,
                                   float(leaf), _fp(oa), len(a), C.byref(na), _fp(ob), len(b), C.byref(nb))
    if rc < 0:
        raise LslamError(rc, ctx.lib.lslam_last_error().decode())
    return oa[:na.value].copy(), ob[:nb.value].copy()
