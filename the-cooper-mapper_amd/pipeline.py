"""Node-level mirrors of the two per-sweep state machines around the hot path, on the device through
the C ABI (clouds are (n, 4) float32 {x, y, z, intensity = ring + relTime}):

* :class:`LaserOdometry` -- ``LaserOdometry::process`` (/root/reference/L_SLAM/src/odometry/
  LaserOdometry.cpp:288-326, 649-653): first sweep initialises the "last" clouds; afterwards
  scanMatch (variant B) against them with the persistent ``_transform`` as initial guess,
  ``_Tsum = _Tsum * transform``, transformToEnd of the less-sharp / less-flat clouds, which become the
  next "last" clouds (their kd-trees are refreshed only when they hold > 10 / > 100 points).
* :class:`LaserMapping` -- ``LaserMapping::process`` (LaserMapping.cpp:39-59) over ``LaserMatcher``
  (LaserMatcher.cpp:289-354): transformMerge (odometry prior), VoxelGrid of the frame's features,
  FeatureMap::update + surround, scanMatchScan (thresholds 0.1 / 0.1, score gate off, return value
  ignored), transformUpdate, addFeatureCloud.

ROS plumbing (topics, time-stamp checks, tf, frame skipping) is not mirrored.
"""
import numpy as np

from .feature_map import FeatureMap, voxel_grid, voxel_grid2


class LaserOdometry:
    def __init__(self, ctx, max_iterations=25, delta_t_abort=0.1, delta_r_abort=0.1):
        self.ctx = ctx
        self.max_iterations, self.dt, self.dr = max_iterations, delta_t_abort, delta_r_abort
        self.transform = np.zeros(6, np.float32)  # _transform: sweep-to-sweep motion, kept as next guess
        self.Tsum = np.eye(4, dtype=np.float32)   # _Tsum
        self.system_inited = False
        self.last_corner = self.last_surf = None   # _lastCornerCloud / _lastSurfaceCloud
        self.tree_corner = self.tree_surf = None   # what the kd-trees were last built from
        self.last_stats = None

    def process(self, sharp, less_sharp, flat, less_flat):
        """One sweep's four feature clouds -> _Tsum (4x4) after the sweep (None for the first one)."""
        less_sharp = np.ascontiguousarray(less_sharp, np.float32)
        less_flat = np.ascontiguousarray(less_flat, np.float32)
        if not self.system_inited:  # :295-303
            self.last_corner, self.last_surf = less_sharp, less_flat
            self.tree_corner, self.tree_surf = less_sharp, less_flat
            self.system_inited = True
            return None
        status, pose, st = self.ctx.odometry_match(self.tree_corner, self.tree_surf, sharp, flat, self.transform,
                                                   self.max_iterations, self.dt, self.dr)
        self.last_stats = st
        self.transform = pose
        self.Tsum = (self.Tsum @ self.ctx.pose_to_isometry(pose)).astype(np.float32)  # transformUpdate, :649-653
        ls = self.ctx.transform_to_end(less_sharp, pose)   # :312-313
        lf = self.ctx.transform_to_end(less_flat, pose)
        self.last_corner, self.last_surf = ls, lf            # :315-316
        if len(ls) > 10 and len(lf) > 100:                   # :321-324
            self.tree_corner, self.tree_surf = ls, lf
        return self.Tsum.copy()


class DeviceLaserOdometry:
    """``LaserOdometry::process`` with the node's state in HBM (``lslam_odom``, include/lslam_c.h): the sweep's feature
    clouds arrive as a :class:`~.scan_registration.FeatureSet`, the last clouds and their search grids never leave the
    device, one wait per sweep.  Same numbers as :class:`LaserOdometry`, bit for bit (tests/test_gpu_odom.py)."""

    def __init__(self, ctx, max_iterations=25, delta_t_abort=0.1, delta_r_abort=0.1, publish=True, publish_buffers=8):
        import ctypes as C
        from .capi import LslamError
        self.ctx = ctx
        h = C.c_void_p()
        rc = ctx.lib.lslam_odom_create(ctx.h, int(max_iterations), float(delta_t_abort), float(delta_r_abort), C.byref(h))
        if rc < 0:
            raise LslamError(rc, ctx.lib.lslam_last_error().decode())
        self.h = h
        # publish: the last clouds leave the device with every sweep (what the mapping node subscribes to) -- into a ring of
        # page-locked buffers of the node; last_corner / last_surf are views of them, valid for publish_buffers more sweeps
        self.publish = publish
        if publish:
            rc = ctx.lib.lslam_odom_set_publish(h, int(publish_buffers))
            if rc < 0:
                raise LslamError(rc, ctx.lib.lslam_last_error().decode())
        self.transform = np.zeros(6, np.float32)
        self.Tsum = np.eye(4, dtype=np.float32)
        self.last_corner = self.last_surf = None
        self.last_stats = self.last_ostats = None
        self._out = [None, None]

    def process(self, fset):
        """One sweep's feature set -> _Tsum (4x4) after the sweep (None for the first one)."""
        import ctypes as C
        from .capi import LslamError, LslamOdomStats, LslamStats, c_float_p
        st, ost = LslamStats(), LslamOdomStats()
        tr, Ts = np.zeros(6, np.float32), np.zeros(16, np.float32)
        fp = lambda x: x.ctypes.data_as(c_float_p)
        rc = self.ctx.lib.lslam_odom_process(self.h, fset.h, fp(tr), fp(Ts), C.byref(st), C.byref(ost), None, 0, None, 0)
        if rc < 0:
            raise LslamError(rc, self.ctx.lib.lslam_last_error().decode())
        oc = os_ = None
        if self.publish:
            pc, ps, nc, ns = c_float_p(), c_float_p(), C.c_size_t(), C.c_size_t()
            self.ctx.lib.lslam_odom_last_view(self.h, C.byref(pc), C.byref(nc), C.byref(ps), C.byref(ns))
            oc = np.ctypeslib.as_array(pc, shape=(nc.value, 4)) if nc.value else np.zeros((0, 4), np.float32)
            os_ = np.ctypeslib.as_array(ps, shape=(ns.value, 4)) if ns.value else np.zeros((0, 4), np.float32)
        self.last_stats, self.last_ostats = st, ost
        self.transform, self.Tsum = tr, Ts.reshape(4, 4)
        self.last_corner, self.last_surf = oc, os_
        first = ost.sweeps == 1
        return None if first else self.Tsum.copy()

    def last_clouds(self):
        """The last clouds as they are in HBM now (two (n, 4) arrays)."""
        from .capi import LslamError, c_float_p
        o = self.last_ostats
        oc, os_ = np.empty((o.n_last_corner, 4), np.float32), np.empty((o.n_last_surf, 4), np.float32)
        rc = self.ctx.lib.lslam_odom_last_clouds(self.h, oc.ctypes.data_as(c_float_p), len(oc), os_.ctypes.data_as(c_float_p), len(os_))
        if rc < 0:
            raise LslamError(rc, self.ctx.lib.lslam_last_error().decode())
        return oc, os_

    def close(self):
        if self.h:
            self.ctx.lib.lslam_odom_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LaserMapping:
    def __init__(self, ctx, cube_dims=(121, 121, 11), filter_corner=1.0, filter_surf=1.0, map_filter_corner=1.0,
                 map_filter_surf=1.0, map_filter=2.0, defer_trees=True, defer_add=False):
        # LaserMatcher.cpp:80-116 defaults
        self.ctx = ctx
        self.defer_add = defer_add  # process() ends with lslam_fmap_add_feature_cloud_begin: pays where the host idles between sweeps
        # the per-frame surround map is searched through its cell grids; its kd-trees are built only if a frame needs them
        # (include/lslam_c.h lslam_map_defer_trees) -- same poses either way
        ctx.defer_trees(defer_trees)
        self.filter_corner, self.filter_surf = filter_corner, filter_surf
        self.feature_map = FeatureMap(ctx, *cube_dims)
        self.feature_map.setup_filter_size(map_filter_corner, map_filter_surf, map_filter)
        self.opts = ctx.default_opts()
        self.opts.delta_t_abort = 0.1   # _scan_match.setConvergeThreshold(0.1, 0.1), :94
        self.opts.delta_r_abort = 0.1
        self.opts.use_score = 0         # setUseCore(false), :95
        self.lidar_odom_last = np.eye(4, dtype=np.float32)    # _lidarOdomLast
        self.lidar_mapped_last = np.eye(4, dtype=np.float32)  # _lidarMappedLast
        self.lidar_mapped_new = np.eye(4, dtype=np.float32)   # _lidarMappedNew
        self.last_stats = None

    def _associate(self, l_old, l_new, w_old):
        import ctypes as C
        from .capi import c_float_p
        out = np.zeros(16, np.float32)
        a, b, c = (np.ascontiguousarray(m, np.float32).reshape(16) for m in (l_old, l_new, w_old))
        fp = lambda x: x.ctypes.data_as(c_float_p)
        self.ctx.lib.lslam_transform_associate(fp(a), fp(b), fp(c), fp(out))
        return out.reshape(4, 4)

    def process(self, corner_last, surf_last, lidar_odom_new):
        """The odometry node's last corner / surf clouds (sweep-end frame) and its _Tsum -> the sweep's
        pose in the map (4x4)."""
        lidar_odom_new = np.ascontiguousarray(lidar_odom_new, np.float32).reshape(4, 4)
        # transformMerge, :333-340
        self.lidar_mapped_new = self._associate(self.lidar_odom_last, lidar_odom_new, self.lidar_mapped_last)
        odom_merged = lidar_odom_new
        # prepareFeatureFrame, :289-301
        if self.filter_corner == self.filter_surf:  # the reference's defaults: both clouds in one pass (same bits)
            corner_ds, surf_ds = voxel_grid2(self.ctx, corner_last, surf_last, self.filter_corner)
        else:
            corner_ds = voxel_grid(self.ctx, corner_last, self.filter_corner)
            surf_ds = voxel_grid(self.ctx, surf_last, self.filter_surf)
        # prepareFeatureSurround, :303-325
        self.feature_map.update(self.lidar_mapped_new[:3, 3])
        nc, ns = self.feature_map.surround_to_map_counts()  # (the surround becomes the context's map; its sizes come back with it)
        # optimizeTransform, :327-331 (return value ignored; pose written back unless "too few ref")
        if nc or ns:
            pose = self.ctx.isometry_to_pose(self.lidar_mapped_new)
            status, pose, st = self.ctx.scanmatch_scan(corner_ds, surf_ds, pose, self.opts)
            self.last_stats = st
            if int(status) != 1:  # LSLAM_TOO_FEW_REF leaves the pose untouched (ScanMatch.cpp:57-61)
                self.lidar_mapped_new = self.ctx.pose_to_isometry(pose)
        # transformUpdate, :342-347
        self.lidar_mapped_last = self.lidar_mapped_new.copy()
        self.lidar_odom_last = odom_merged.copy()
        # featureMapUpdate, :349-354 (defer_add: enqueued, not waited for -- the rebuild runs while the node takes up its next sweep,
        # the next call on the map waits and commits first)
        self.feature_map.add_feature_cloud(corner_ds, surf_ds, self.lidar_mapped_new, wait=not self.defer_add)
        return self.lidar_mapped_new.copy()
