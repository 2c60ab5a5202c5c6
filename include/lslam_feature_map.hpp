// lslam_feature_map.hpp -- header-only C++ shim with the class surface of the reference's
// lidar_slam::FeatureMap<PointT> (/root/reference/L_SLAM/src/util/FeatureMap.h:42-186) over the C ABI
// (lslam_fmap_* in lslam_c.h), so that the call sites in odometry/LaserMatcher.cpp (:107-116 setup,
// :312-318 update + getSurroundFeature, :349-353 addFeatureCloud, saveMap / getFullMap services) compile
// against it.  Templated so that PCL / Eigen are not needed to build the backend:
//   PointCloud : .points (std::vector-like of PointT with float x, y, z, ..., intensity), .clear(),
//                .points.resize(); sizeof(PointT) is the stride, intensity is read at offsetof
//   Isometry   : .matrix()(r,c)
// The map itself lives in HBM; getSurroundFeature copies it out for callers that want the clouds, and
// surroundToMap() hands it to the scan matcher without leaving the device.
#pragma once

#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#include "lslam_c.h"

namespace lidar_slam {

template <typename PointT, typename PointCloud>
class FeatureMap {
public:
  // FeatureMap.h:55-68; `ctx` is the lslam context of the ScanMatch shim (lslam_scan_match.hpp: context())
  FeatureMap(lslam_ctx *ctx, int cubeWidth_ = 21, int cubeHeight_ = 11, int cubeDepth_ = 21) : _fm(nullptr) {
    if (lslam_fmap_create(ctx, cubeWidth_, cubeHeight_, cubeDepth_, &_fm) != LSLAM_OK)
      throw std::runtime_error(std::string("lslam_fmap_create: ") + lslam_last_error());
  }
  ~FeatureMap() { lslam_fmap_destroy(_fm); }
  FeatureMap(const FeatureMap &) = delete;
  FeatureMap &operator=(const FeatureMap &) = delete;

  // :72-91
  inline void setupFilterSize(float corner, float surf, float map) { lslam_fmap_setup_filter_size(_fm, corner, surf, map); }
  inline void setupWorldOrigin(float w, float h, float d) { lslam_fmap_setup_world_origin(_fm, (int32_t)w, (int32_t)h, (int32_t)d); }
  inline void setupWorldCubeSize(float s) { lslam_fmap_setup_world_cube_size(_fm, s); }
  inline void setupLidarValidDistance(float d) { lslam_fmap_setup_lidar_valid_distance(_fm, d); }
  inline void setupFilesDirectory(const std::string &dir) { _filesDirectory = dir; }

  // :218-230
  template <typename Isometry>
  void addFeatureCloud(const PointCloud &cornerCloud, const PointCloud &surfCloud, const Isometry &tf) {
    float T[16];
    Isometry &M = const_cast<Isometry &>(tf);
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) T[r * 4 + c] = (float)M.matrix()(r, c);
    check(lslam_fmap_add_feature_cloud(_fm, cornerCloud.points.data(), cornerCloud.points.size(), surfCloud.points.data(),
                                       surfCloud.points.size(), sizeof(PointT), T));
  }
  // :232-254
  inline void update(const PointT &sensorPose) {
    const float p[3] = {sensorPose.x, sensorPose.y, sensorPose.z};
    check(lslam_fmap_update(_fm, p));
  }
  // :256-265
  void getSurroundFeature(PointCloud &surroundCorner, PointCloud &surroundSurf) {
    size_t nc = 0, ns = 0;
    check(lslam_fmap_surround_counts(_fm, &nc, &ns));
    std::vector<float> c(4 * nc + 4), s(4 * ns + 4);
    check(lslam_fmap_get_surround(_fm, c.data(), nc, s.data(), ns));
    unpack(c, nc, surroundCorner);
    unpack(s, ns, surroundSurf);
  }
  // the same content as the scan matcher's map, without the host round trip
  void surroundToMap() { check(lslam_fmap_surround_to_map(_fm)); }
  // :267-286
  template <typename PointCloudPtr>
  bool getFullMap(PointCloudPtr &mapCloud) {
    size_t n = 0;
    check(lslam_fmap_get_full_map(_fm, nullptr, 0, &n));
    std::vector<float> m(4 * n + 4);
    check(lslam_fmap_get_full_map(_fm, m.data(), n, &n));
    unpack(m, n, *mapCloud);
    return true;
  }
  // :378-462
  bool saveCloudToFiles() { return lslam_fmap_save(_fm, _filesDirectory.c_str()) == LSLAM_OK; }
  bool loadCloudFromFiles() { return lslam_fmap_load(_fm, _filesDirectory.c_str()) == LSLAM_OK; }

  lslam_fmap *handle() { return _fm; }

private:
  static void check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("feature map: ") + lslam_last_error());
  }
  static void unpack(const std::vector<float> &v, size_t n, PointCloud &out) {
    out.points.resize(n);
    for (size_t i = 0; i < n; ++i) {
      PointT p = PointT();
      p.x = v[4 * i];
      p.y = v[4 * i + 1];
      p.z = v[4 * i + 2];
      p.intensity = v[4 * i + 3];
      out.points[i] = p;
    }
  }
  lslam_fmap *_fm;
  std::string _filesDirectory;
};

}  // namespace lidar_slam
