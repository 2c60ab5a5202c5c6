// lslam_pipeline.hpp -- header-only C++ mirrors of the two per-sweep state machines either side of the
// scan-match hot path, over the C ABI (lslam_c.h):
//
//   lidar_slam::LaserOdometry::process   /root/reference/L_SLAM/src/odometry/LaserOdometry.cpp:288-326
//                                        (+ scanMatch :328-647 = lslam_odometry_match, transformToEnd
//                                        :156-168 = lslam_transform_to_end, transformUpdate :649-653)
//   lidar_slam::LaserMapping::process    odometry/LaserMapping.cpp:39-59 over LaserMatcher
//                                        (odometry/LaserMatcher.cpp:289-354: prepareFeatureFrame,
//                                        prepareFeatureSurround, optimizeTransform, transformMerge /
//                                        transformUpdate, featureMapUpdate)
//
// ROS plumbing (topics, time-stamp matching, tf, frame skipping) is the host program's.  Clouds are any
// type with `.points` (std::vector-like) of points with float x, y, z and `intensity` (= ring + relTime);
// sizeof(point) is the stride, the intensity is read where the member lies.  Poses are row-major 4x4
// float arrays (Eigen::Isometry3f::matrix() transposed into row order by the caller's adapter).
// Like the reference's nodes these objects never throw; a backend failure makes process() return false
// and leaves the message in lastError().
#pragma once

#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "lslam_c.h"

namespace lidar_slam {

namespace detail {
// pack a cloud into {x, y, z, intensity} floats (the layout every lslam_* cloud entry point accepts with stride 16)
template <typename Cloud>
inline void pack_xyzi(const Cloud &c, std::vector<float> &out) {
  out.resize(4 * c.points.size());
  for (size_t i = 0; i < c.points.size(); ++i) {
    out[4 * i] = c.points[i].x;
    out[4 * i + 1] = c.points[i].y;
    out[4 * i + 2] = c.points[i].z;
    out[4 * i + 3] = c.points[i].intensity;
  }
}
inline void mat_mul4(const float A[16], const float B[16], float C[16]) {
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
      C[r * 4 + c] = s;
    }
}
inline void identity4(float T[16]) {
  std::memset(T, 0, 16 * sizeof(float));
  T[0] = T[5] = T[10] = T[15] = 1.f;
}
}  // namespace detail

// LaserOdometry (variant B, BASELINE configs[0]): first sweep initialises the "last" clouds; afterwards
// scanMatch against them with the persistent _transform as the initial guess, _Tsum = _Tsum * transform,
// transformToEnd of the less-sharp / less-flat clouds, which become the next "last" clouds
// (LaserOdometry.cpp:288-326).  The node's state lives in HBM (lslam_odom, include/lslam_c.h): the last clouds and
// their search grids never leave the device; process() takes the sweep's four clouds from the host (one upload),
// processFeatureSet() takes them where lslam_extract_features_dev left them.
class LaserOdometry {
public:
  explicit LaserOdometry(lslam_ctx *ctx, int maxIterations = 25, float deltaTAbort = 0.1f, float deltaRAbort = 0.1f)
      : _ctx(ctx), _od(nullptr), _fs(nullptr), _matched(false) {
    std::memset(_transform, 0, sizeof(_transform));
    detail::identity4(_Tsum);
    std::memset(&_last, 0, sizeof(_last));
    std::memset(&_ostats, 0, sizeof(_ostats));
    if (lslam_odom_create(ctx, maxIterations, deltaTAbort, deltaRAbort, &_od) != LSLAM_OK ||
        lslam_fset_create(ctx, &_fs) != LSLAM_OK)
      _err = lslam_last_error();
  }
  ~LaserOdometry() {
    if (_od) lslam_odom_destroy(_od);
    if (_fs) lslam_fset_destroy(_fs);
  }
  LaserOdometry(const LaserOdometry &) = delete;
  LaserOdometry &operator=(const LaserOdometry &) = delete;
  // LaserOdometry.cpp:288-326.  Returns false for the first sweep (nothing to match against) and on a backend
  // error; Tsum() is the accumulated sweep-to-sweep motion, lastCornerCloud()/lastSurfaceCloud() the clouds the
  // mapping node receives (/laser_cloud_corner_last, /laser_cloud_surf_last), packed {x,y,z,intensity}.
  template <typename Cloud>
  bool process(const Cloud &cornerPointsSharp, const Cloud &cornerPointsLessSharp, const Cloud &surfPointsFlat,
               const Cloud &surfPointsLessFlat) {
    if (!_od || !_fs) return false;
    std::vector<float> sharp, less_sharp, flat, less_flat;
    detail::pack_xyzi(cornerPointsSharp, sharp);
    detail::pack_xyzi(cornerPointsLessSharp, less_sharp);
    detail::pack_xyzi(surfPointsFlat, flat);
    detail::pack_xyzi(surfPointsLessFlat, less_flat);
    if (lslam_fset_upload(_ctx, _fs, sharp.data(), sharp.size() / 4, less_sharp.data(), less_sharp.size() / 4, flat.data(),
                          flat.size() / 4, less_flat.data(), less_flat.size() / 4, 16) < 0)
      return fail();
    return processFeatureSet(_fs);
  }
  // the same on a feature set that is in HBM already
  bool processFeatureSet(lslam_fset *fs) {
    if (!_od) return false;
    size_t n[4];
    if (lslam_fset_counts(fs, n) < 0) return fail();
    _lastCorner.resize(4 * n[1]);
    _lastSurf.resize(4 * n[3]);
    const uint64_t before = _ostats.sweeps;
    const int st = lslam_odom_process(_od, fs, _transform, _Tsum, &_last, &_ostats, _lastCorner.data(), n[1], _lastSurf.data(), n[3]);
    if (st < 0) return fail();
    _matched = _ostats.matched != 0;
    return before != 0;  // the first sweep only initialises (:295-303)
  }
  const float *Tsum() const { return _Tsum; }
  const float *transform() const { return _transform; }
  const std::vector<float> &lastCornerCloud() const { return _lastCorner; }
  const std::vector<float> &lastSurfaceCloud() const { return _lastSurf; }
  const lslam_stats &lastStats() const { return _last; }
  const lslam_odom_stats &nodeStats() const { return _ostats; }
  bool matched() const { return _matched; }  // false: the last clouds were too small to match against (:337)
  const std::string &lastError() const { return _err; }

private:
  bool fail() {
    _err = lslam_last_error();
    return false;
  }
  lslam_ctx *_ctx;
  lslam_odom *_od;
  lslam_fset *_fs;
  bool _matched;
  float _transform[6];  // _transform: sweep-to-sweep motion, kept as the next initial guess
  float _Tsum[16];      // _Tsum
  std::vector<float> _lastCorner, _lastSurf;
  lslam_stats _last;
  lslam_odom_stats _ostats;
  std::string _err;
};

// LaserMapping (BASELINE configs[1]): per sweep transformMerge (odometry prior), VoxelGrid of the frame's feature
// clouds, FeatureMap::update + surround -> kd-trees (device), scanMatchScan with thresholds 0.1 / 0.1 and the
// score gate off (its return value is ignored, LaserMatcher.cpp:327-331), transformUpdate, addFeatureCloud.
class LaserMapping {
public:
  // LaserMatcher.cpp:80-116 defaults (filter 1.0 / 1.0, map filters 1.0 / 1.0 / 2.0, 121 x 121 x 11 cubes)
  explicit LaserMapping(lslam_ctx *ctx, int cubeX = 121, int cubeY = 121, int cubeZ = 11, float filterCorner = 1.0f,
                        float filterSurf = 1.0f, float mapFilterCorner = 1.0f, float mapFilterSurf = 1.0f, float mapFilter = 2.0f)
      : _ctx(ctx), _fm(nullptr), _filterCorner(filterCorner), _filterSurf(filterSurf) {
    lslam_default_opts(&_opts);
    _opts.delta_t_abort = 0.1f;  // _scan_match.setConvergeThreshold(0.1, 0.1), LaserMatcher.cpp:94
    _opts.delta_r_abort = 0.1f;
    _opts.use_score = 0;         // setUseCore(false), :95
    detail::identity4(_lidarOdomLast);
    detail::identity4(_lidarMappedLast);
    detail::identity4(_lidarMappedNew);
    std::memset(&_last, 0, sizeof(_last));
    if (lslam_fmap_create(ctx, cubeX, cubeY, cubeZ, &_fm) != LSLAM_OK) {
      _fm = nullptr;
      _err = lslam_last_error();
    } else {
      lslam_fmap_setup_filter_size(_fm, mapFilterCorner, mapFilterSurf, mapFilter);
      // the per-frame surround map is searched through its cell grids; its kd-trees are built only if a frame needs them
      // (lslam_map_defer_trees) -- same poses either way
      lslam_map_defer_trees(ctx, 1);
    }
  }
  ~LaserMapping() { lslam_fmap_destroy(_fm); }
  LaserMapping(const LaserMapping &) = delete;
  LaserMapping &operator=(const LaserMapping &) = delete;

  // cornerLast / surfLast: the odometry node's last clouds, packed {x,y,z,intensity} (LaserOdometry::lastCornerCloud());
  // lidarOdomNew: its _Tsum (row-major 4x4).  Returns false on a backend error; lidarMapped() is the sweep's pose in the map.
  bool process(const std::vector<float> &cornerLast, const std::vector<float> &surfLast, const float lidarOdomNew[16]) {
    if (!_fm) return false;
    // transformMerge, :333-340
    lslam_transform_associate(_lidarOdomLast, lidarOdomNew, _lidarMappedLast, _lidarMappedNew);
    // prepareFeatureFrame, :289-301
    if (_filterCorner == _filterSurf) {  // the reference's defaults: both clouds in one pass (lslam_voxel_grid2: same bits)
      _cornerDS.resize(cornerLast.size() + 4);
      _surfDS.resize(surfLast.size() + 4);
      size_t nc2 = 0, ns2 = 0;
      if (lslam_voxel_grid2(_ctx, cornerLast.data(), cornerLast.size() / 4, surfLast.data(), surfLast.size() / 4, 16, _filterCorner,
                            _cornerDS.data(), cornerLast.size() / 4, &nc2, _surfDS.data(), surfLast.size() / 4, &ns2) < 0)
        return fail();
      _cornerDS.resize(4 * nc2);
      _surfDS.resize(4 * ns2);
    } else if (!downsize(cornerLast, _filterCorner, _cornerDS) || !downsize(surfLast, _filterSurf, _surfDS)) {
      return fail();
    }
    // prepareFeatureSurround, :303-325
    const float pos[3] = {_lidarMappedNew[3], _lidarMappedNew[7], _lidarMappedNew[11]};
    if (lslam_fmap_update(_fm, pos) < 0) return fail();
    size_t nc = 0, ns = 0;
    if (lslam_fmap_surround_to_map_counts(_fm, &nc, &ns) < 0) return fail();  // the surround becomes the context's map
    if (nc || ns) {  // optimizeTransform, :327-331
      float pose[6];
      lslam_isometry_to_pose(_lidarMappedNew, pose);
      const int st = lslam_scanmatch_scan(_ctx, _cornerDS.data(), _cornerDS.size() / 4, _surfDS.data(), _surfDS.size() / 4, 16, pose,
                                          &_opts, &_last);
      if (st < 0) return fail();
      if (st != LSLAM_TOO_FEW_REF) lslam_pose_to_isometry(pose, _lidarMappedNew);  // ScanMatch.cpp:57-61 leaves the pose untouched
    }
    // transformUpdate, :342-347
    std::memcpy(_lidarMappedLast, _lidarMappedNew, sizeof(_lidarMappedNew));
    std::memcpy(_lidarOdomLast, lidarOdomNew, sizeof(_lidarOdomLast));
    // featureMapUpdate, :349-354.  -DLSLAM_MAPPING_DEFER_ADD: enqueued, not waited for (lslam_fmap_add_feature_cloud_begin) -- the
    // map's rebuild then runs while the node takes up its next sweep and the next call on the map waits and commits first.  It pays
    // where the host idles between two sweeps (the Python mirror: 0.91 -> 0.82 ms per frame); with three nodes on three threads
    // it does not (the device is the limit there) and the chain fell into its slow mode more often (tools/node_threads_ab.py)
#ifdef LSLAM_MAPPING_DEFER_ADD
    if (lslam_fmap_add_feature_cloud_begin(_fm, _cornerDS.data(), _cornerDS.size() / 4, _surfDS.data(), _surfDS.size() / 4, 16,
                                           _lidarMappedNew) < 0)
      return fail();
#else
    if (lslam_fmap_add_feature_cloud(_fm, _cornerDS.data(), _cornerDS.size() / 4, _surfDS.data(), _surfDS.size() / 4, 16, _lidarMappedNew) < 0)
      return fail();
#endif
    return true;
  }
  const float *lidarMapped() const { return _lidarMappedNew; }
  const lslam_stats &lastStats() const { return _last; }
  const std::string &lastError() const { return _err; }
  lslam_fmap *featureMap() { return _fm; }

private:
  bool downsize(const std::vector<float> &in, float leaf, std::vector<float> &out) {
    out.resize(in.size() + 4);
    size_t n = 0;
    const int st = lslam_voxel_grid(_ctx, in.data(), in.size() / 4, 16, leaf, out.data(), in.size() / 4, &n);
    out.resize(4 * n);
    return st >= 0;
  }
  bool fail() {
    _err = lslam_last_error();
    return false;
  }
  lslam_ctx *_ctx;
  lslam_fmap *_fm;
  float _filterCorner, _filterSurf;
  lslam_opts _opts;
  float _lidarOdomLast[16], _lidarMappedLast[16], _lidarMappedNew[16];
  std::vector<float> _cornerDS, _surfDS;
  lslam_stats _last;
  std::string _err;
};

}  // namespace lidar_slam
