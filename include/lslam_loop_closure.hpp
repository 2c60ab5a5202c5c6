// lslam_loop_closure.hpp -- header-only C++ mirrors of the loop-closure front end and the pose-graph node's
// bookkeeping over the C ABI (lslam_c.h):
//
//   pose_graph::LoopDetector      /root/reference/L_SLAM/src/pose_graph/loop_detector.hpp:50-280
//                                 (trajectory radius search + candidate gating on the host, coarse alignment =
//                                 lslam_icp_align, fine alignment = ScanMatch::scanMatchLocal on the device)
//   pose_graph::KeyframeUpdater   pose_graph/keyframe_updater.hpp:10-60
//   pose_graph::Graph             pose_graph/graph.cpp:230-385 (keyframe queue, odometry / loop edges with the
//                                 reference's information matrices, optimise when a loop was found, odom -> graph)
//
// ROS topics, threads and tf are the host program's.  Poses are row-major 4x4 doubles (Mat4d); clouds are packed
// {x, y, z, intensity} floats.  Nothing here throws; backend failures end up in lastError() and a false / empty
// result, the way the reference's nodes report a failed match.
//
// Three things of the reference are restated as they are, not as they were probably meant (the Python mirror
// the-cooper-mapper_amd/loop_closure.py documents the evidence): the radius handed to radiusSearch is compared
// with SQUARED distances (nanoflann_pcl.h:166-186), KdTreeFLANN::radiusSearch returns ONE result -- the nearest
// (nanoflann_pcl.h:173) --, and the trajectory is flattened with y = 0 (loop_detector.hpp:98,120).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "lslam_c.h"

namespace pose_graph {

struct Mat4d {
  double m[16];
  Mat4d() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = m[15] = 1.0; }
  double &operator()(int r, int c) { return m[r * 4 + c]; }
  double operator()(int r, int c) const { return m[r * 4 + c]; }
};
inline Mat4d operator*(const Mat4d &A, const Mat4d &B) {
  Mat4d C;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A(r, k) * B(k, c);
      C(r, c) = s;
    }
  return C;
}
inline Mat4d inverse(const Mat4d &T) {  // Eigen::Isometry3d::inverse(): [R^T | -R^T t]
  Mat4d I;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) I(r, c) = T(c, r);
    I(r, 3) = -(T(0, r) * T(0, 3) + T(1, r) * T(1, 3) + T(2, r) * T(2, 3));
  }
  return I;
}
inline void to_float16(const Mat4d &T, float out[16]) { for (int i = 0; i < 16; ++i) out[i] = (float)T.m[i]; }
inline Mat4d from_float16(const float in[16]) { Mat4d T; for (int i = 0; i < 16; ++i) T.m[i] = in[i]; return T; }

// 4x4 <-> g2o's {t, q_xyzw} (Eigen::Quaterniond(R))
inline void mat_to_pose7(const Mat4d &T, double p[7]) {
  p[0] = T(0, 3); p[1] = T(1, 3); p[2] = T(2, 3);
  const double tr = T(0, 0) + T(1, 1) + T(2, 2);
  double q[4];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (T(2, 1) - T(1, 2)) * s; q[1] = (T(0, 2) - T(2, 0)) * s; q[2] = (T(1, 0) - T(0, 1)) * s;
  } else {
    int i = 0;
    if (T(1, 1) > T(0, 0)) i = 1;
    if (T(2, 2) > T(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = std::sqrt(T(i, i) - T(j, j) - T(k, k) + 1.0);
    q[i] = 0.5 * s;
    s = 0.5 / s;
    q[3] = (T(k, j) - T(j, k)) * s;
    q[j] = (T(j, i) + T(i, j)) * s;
    q[k] = (T(k, i) + T(i, k)) * s;
  }
  p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
}
inline Mat4d pose7_to_mat(const double p[7]) {
  const double x = p[3], y = p[4], z = p[5], w = p[6];
  Mat4d T;
  T(0, 0) = 1 - 2 * (y * y + z * z); T(0, 1) = 2 * (x * y - z * w);     T(0, 2) = 2 * (x * z + y * w);
  T(1, 0) = 2 * (x * y + z * w);     T(1, 1) = 1 - 2 * (x * x + z * z); T(1, 2) = 2 * (y * z - x * w);
  T(2, 0) = 2 * (x * z - y * w);     T(2, 1) = 2 * (y * z + x * w);     T(2, 2) = 1 - 2 * (x * x + y * y);
  T(0, 3) = p[0]; T(1, 3) = p[1]; T(2, 3) = p[2];
  return T;
}

// pose_graph/keyframe.h
struct KeyFrame {
  typedef std::shared_ptr<KeyFrame> Ptr;
  Mat4d odom, estimate;          // odometry pose at insertion; node->estimate()
  double accum_distance = 0.0;
  std::vector<float> cornerCloud, surfCloud;  // {x,y,z,intensity} in the keyframe's own frame
  int node = -1;                 // vertex id in the solver
  int frame_id = 0;
};

// loop_detector.hpp:18-48
struct Loop {
  typedef std::shared_ptr<Loop> Ptr;
  KeyFrame::Ptr key1, key2;
  Mat4d relative_pose;
};

// keyframe_updater.hpp:10-60
class KeyframeUpdater {
public:
  bool update(const Mat4d &pose) {
    if (is_first) {
      is_first = false;
      prev_keypose = pose;
      return true;
    }
    const Mat4d delta = inverse(prev_keypose) * pose;
    const double dx = std::sqrt(delta(0, 3) * delta(0, 3) + delta(1, 3) * delta(1, 3) + delta(2, 3) * delta(2, 3));
    const double c = std::min(1.0, std::max(-1.0, (delta(0, 0) + delta(1, 1) + delta(2, 2) - 1.0) / 2.0));
    const double da = std::acos(c);  // Eigen::AngleAxisd(R).angle()
    if (dx < keyframe_delta_trans && da < keyframe_delta_angle) return false;
    accum_distance += dx;
    prev_keypose = pose;
    frame_count++;
    return true;
  }
  double get_accum_distance() const { return accum_distance; }
  int get_unique_id() { return ++frame_id; }
  double keyframe_delta_trans = 0.25, keyframe_delta_angle = 0.05;

private:
  bool is_first = true;
  Mat4d prev_keypose;
  double accum_distance = 0.0;
  int frame_count = 0, frame_id = 0;
};

class LoopDetector {
public:
  // loop_detector.hpp:55-63; `ctx` runs the coarse and the fine alignment
  explicit LoopDetector(lslam_ctx *ctx) : _ctx(ctx) { lslam_default_opts(&_opts); }

  double get_distance_thresh() const { return estimated_distance_thresh; }
  int get_loop_count() const { return loop_count; }
  const std::string &lastError() const { return _err; }

  // :66-87
  bool detect_nearest(const std::vector<KeyFrame::Ptr> &keyframes, const std::deque<KeyFrame::Ptr> &new_keyframes,
                      std::vector<Loop::Ptr> &detected_loops) {
    updateTrajectory(keyframes);
    bool loop_found = false;
    for (const auto &nk : new_keyframes) {
      std::vector<KeyFrame::Ptr> candidates;
      if (!find_nearest_candidates(keyframes, nk, candidates)) continue;
      Loop::Ptr loop = matching_nearest(candidates, nk);
      if (loop) {
        detected_loops.push_back(loop);
        loop_count++;
        loop_found = true;
      }
    }
    return loop_found;
  }

  // :93-106
  void updateTrajectory(const std::vector<KeyFrame::Ptr> &keyframes) {
    _traj.resize(keyframes.size() * 3);
    for (size_t i = 0; i < keyframes.size(); ++i) {
      _traj[3 * i] = (float)keyframes[i]->estimate(0, 3);
      _traj[3 * i + 1] = 0.0f;
      _traj[3 * i + 2] = (float)keyframes[i]->estimate(2, 3);
    }
  }

  // :108-164
  bool find_nearest_candidates(const std::vector<KeyFrame::Ptr> &keyframes, const KeyFrame::Ptr &nk,
                               std::vector<KeyFrame::Ptr> &candidates) {
    if (nk->accum_distance - last_loop_accum_distance < last_loop_interval_thresh) return false;
    const float pos[3] = {(float)nk->estimate(0, 3), 0.0f, (float)nk->estimate(2, 3)};
    // KdTreeFLANN::radiusSearch(pos, 5.0): squared distances against 5.0, ONE result -- the nearest
    int best = -1;
    float best_d2 = 5.0f;
    for (size_t i = 0; i < _traj.size() / 3; ++i) {
      const float dx = _traj[3 * i] - pos[0], dy = _traj[3 * i + 1] - pos[1], dz = _traj[3 * i + 2] - pos[2];
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      if (d2 < best_d2) { best_d2 = d2; best = (int)i; }
    }
    if (best < 0) return false;
    if (best_d2 >= estimated_distance_thresh) return false;
    const KeyFrame::Ptr &kf = keyframes[(size_t)best];
    if (nk->accum_distance - kf->accum_distance < accum_distance_thresh) return false;
    candidates.push_back(kf);
    return true;
  }

  // :166-230
  Loop::Ptr matching_nearest(const std::vector<KeyFrame::Ptr> &candidates, const KeyFrame::Ptr &nk) {
    if (candidates.empty()) return nullptr;
    const Mat4d inv = inverse(candidates[0]->estimate);
    std::vector<float> cornerLocal = candidates[0]->cornerCloud, surfLocal = candidates[0]->surfCloud;
    for (size_t i = 1; i < candidates.size(); ++i) {
      float rel[16];
      to_float16(inv * candidates[i]->estimate, rel);
      append_transformed(candidates[i]->cornerCloud, rel, cornerLocal);
      append_transformed(candidates[i]->surfCloud, rel, surfLocal);
    }
    float guess[16];
    to_float16(inv * nk->estimate, guess);
    // corseMatching, :232-255
    if (surfLocal.empty()) return nullptr;
    int32_t conv = 0, its = 0;
    double fit = 0;
    if (lslam_icp_align(_ctx, surfLocal.data(), surfLocal.size() / 4, nk->surfCloud.data(), nk->surfCloud.size() / 4, 16, guess, 10,
                        0.0, 0.0, &fit, &conv, &its) < 0) {
      _err = lslam_last_error();
      return nullptr;
    }
    if (!conv) return nullptr;
    // scan_match.scanMatchLocal(cornerLocal, surfLocal, corner, surf, guess2): VoxelGrid 0.2 / 0.4, then scanMatchScan
    std::vector<float> ds[4];
    const std::vector<float> *in[4] = {&cornerLocal, &surfLocal, &nk->cornerCloud, &nk->surfCloud};
    const float leaf[4] = {0.2f, 0.4f, 0.2f, 0.4f};
    for (int k = 0; k < 4; ++k) {
      ds[k].resize(in[k]->size() + 4);
      size_t n = 0;
      if (lslam_voxel_grid(_ctx, in[k]->data(), in[k]->size() / 4, 16, leaf[k], ds[k].data(), in[k]->size() / 4, &n) < 0) {
        _err = lslam_last_error();
        return nullptr;
      }
      ds[k].resize(4 * n);
    }
    float pose[6];
    lslam_isometry_to_pose(guess, pose);
    lslam_stats st;
    const int rc = lslam_scanmatch_full(_ctx, ds[0].data(), ds[0].size() / 4, ds[1].data(), ds[1].size() / 4, 16, ds[2].data(),
                                        ds[2].size() / 4, ds[3].data(), ds[3].size() / 4, 16, pose, &_opts, &st);
    if (rc < 0) _err = lslam_last_error();
    if (rc != LSLAM_OK) return nullptr;  // hasConverged == false
    lslam_pose_to_isometry(pose, guess);
    last_loop_accum_distance = nk->accum_distance;
    Loop::Ptr lp = std::make_shared<Loop>();
    lp->key1 = candidates[0];
    lp->key2 = nk;
    lp->relative_pose = from_float16(guess);
    return lp;
  }

  double estimated_distance_thresh = 25.0, accum_distance_thresh = 30.0, last_loop_interval_thresh = 3.0,
         fitness_score_thresh = 0.5;

private:
  static void append_transformed(const std::vector<float> &c, const float T[16], std::vector<float> &out) {
    for (size_t i = 0; i + 3 < c.size(); i += 4) {  // pcl::transformPointCloud, fp32
      const float x = c[i], y = c[i + 1], z = c[i + 2];
      out.push_back(((T[0] * x + T[1] * y) + T[2] * z) + T[3]);
      out.push_back(((T[4] * x + T[5] * y) + T[6] * z) + T[7]);
      out.push_back(((T[8] * x + T[9] * y) + T[10] * z) + T[11]);
      out.push_back(c[i + 3]);
    }
  }
  lslam_ctx *_ctx;
  lslam_opts _opts;
  std::vector<float> _traj;
  int loop_count = 0;
  double last_loop_accum_distance = 0.0;
  std::string _err;
};

// pose_graph::Graph (graph.cpp:230-385) with SolverG2O (solver_g2o.cpp:51-95) folded in: vertices / edges are
// kept on the host and handed to lslam_pg_* when an optimisation is due.
class Graph {
public:
  explicit Graph(lslam_ctx *ctx, int device = 0, int max_keyframes_per_update = 10)
      : loop_detector(ctx), _device(device), _max_per_update(max_keyframes_per_update) {}

  // graph.cpp:230-246: returns the queued keyframe, or null when the pose did not move enough
  KeyFrame::Ptr add_frame(const Mat4d &odom, const std::vector<float> &corner, const std::vector<float> &surf) {
    if (!keyframe_updater.update(odom)) return nullptr;
    KeyFrame::Ptr kf = std::make_shared<KeyFrame>();
    kf->odom = odom;
    kf->accum_distance = keyframe_updater.get_accum_distance();
    kf->cornerCloud = corner;
    kf->surfCloud = surf;
    kf->frame_id = keyframe_updater.get_unique_id();
    keyframe_queue.push_back(kf);
    return kf;
  }

  // one pass of the optimisation loop, graph.cpp:313-383: flush the queue (vertices + odometry edges), detect loops
  // among the new keyframes, add loop edges, optimise if a loop was found, update odom -> graph.
  // Returns the number of loops found (-1 on a backend error).
  int optimize(int max_iterations = 1000) {
    if (!flush_keyframe_queue()) return 0;
    std::vector<Loop::Ptr> found;
    std::deque<KeyFrame::Ptr> nk(new_keyframes.begin(), new_keyframes.end());
    loop_detector.detect_nearest(keyframes, nk, found);
    static const double LOOP_INFO[6] = {2, 2, 2, 2, 2, 2};  // graph.cpp:333-339
    for (const auto &lp : found) add_edge(lp->key1->node, lp->key2->node, lp->relative_pose, LOOP_INFO);
    loops.insert(loops.end(), found.begin(), found.end());
    keyframes.insert(keyframes.end(), new_keyframes.begin(), new_keyframes.end());
    new_keyframes.clear();
    last_iterations = 0;
    if (!found.empty()) {
      lslam_pg *pg = nullptr;
      const int ne = (int)(_ij.size() / 2);
      if (lslam_pg_create(_device, (int)(_poses.size() / 7), _poses.data(), ne, _ij.data(), _meas.data(), _info.data(), 0, &pg) !=
          LSLAM_OK) {
        _err = lslam_pg_last_error();
        return -1;
      }
      lslam_pg_stats st;
      const int rc = lslam_pg_optimize(pg, max_iterations, &st);
      if (rc >= 0) lslam_pg_get_poses(pg, _poses.data());
      lslam_pg_destroy(pg);
      if (rc < 0) {
        _err = lslam_pg_last_error();
        return -1;
      }
      last_iterations = st.iterations;
      for (auto &kf : keyframes) kf->estimate = pose7_to_mat(&_poses[7 * (size_t)kf->node]);
    }
    const KeyFrame::Ptr &last = keyframes.back();
    tf_odom2graph = last->estimate * inverse(last->odom);
    return (int)found.size();
  }

  // Graph::getFinalFeatureMap (graph.cpp:150-199; called by Graph::save, :106-147): the optimised keyframes rebuilt into a map
  // ONE AFTER THE OTHER -- keyframe k is matched against a map that already holds keyframes 0 .. k-1:
  //   feature_map2.update(estimate) -> getSurroundFeature -> VoxelGrid 0.2 / 0.3 of the keyframe's clouds -> scanMatchScan (a
  //   default ScanMatch) from the estimate -> addFeatureCloud with the refined estimate iff matched -> saveCloudToFiles(directory)
  // all on the device (lslam_fmap_*: the map never leaves HBM between keyframes).  matched[k] / poses[k]: per keyframe.
  // Quirk kept with bootstrap = false (the reference as written): the map starts empty, the first match returns false for want
  // of reference points (ScanMatch.cpp:57-61), nothing is added -- and so for every keyframe after it.  bootstrap = true adds a
  // keyframe WITHOUT a match while the surround holds fewer than the 50 / 100 points a match needs.  The caller destroys *map_out
  // (lslam_fmap_destroy); returns the number of keyframes added, -1 on a backend error.
  int getFinalFeatureMap(lslam_ctx *ctx, const std::string &directory, bool bootstrap, std::vector<char> &matched,
                         std::vector<Mat4d> &poses, lslam_fmap **map_out, int cubeWidth = 121, int cubeHeight = 111,
                         int cubeDepth = 121) {
    matched.clear();
    poses.clear();
    lslam_fmap *fm = nullptr;
    if (lslam_fmap_create(ctx, cubeWidth, cubeHeight, cubeDepth, &fm) != LSLAM_OK) return fail_final(nullptr);
    lslam_fmap_setup_filter_size(fm, 0.2f, 0.2f, 0.4f);
    lslam_opts opts;
    lslam_default_opts(&opts);  // lidar_slam::ScanMatch scan_match;
    int added = 0;
    std::vector<float> cc, cs;
    for (const auto &kf : keyframes) {
      float T[16];
      to_float16(kf->estimate, T);  // node->estimate().cast<float>()
      const float pos[3] = {T[3], T[7], T[11]};
      if (lslam_fmap_update(fm, pos) < 0) return fail_final(fm);
      size_t nc = 0, ns = 0;
      if (lslam_fmap_surround_counts(fm, &nc, &ns) < 0) return fail_final(fm);
      const size_t kc = kf->cornerCloud.size() / 4, ks = kf->surfCloud.size() / 4;
      cc.resize(4 * kc + 4);
      cs.resize(4 * ks + 4);
      size_t mc = 0, ms = 0;
      if (lslam_voxel_grid(ctx, kf->cornerCloud.data(), kc, 16, 0.2f, cc.data(), kc, &mc) < 0) return fail_final(fm);
      if (lslam_voxel_grid(ctx, kf->surfCloud.data(), ks, 16, 0.3f, cs.data(), ks, &ms) < 0) return fail_final(fm);
      bool ok = false;
      const bool enough = nc >= 50 && ns >= 100;
      if (enough) {
        if (lslam_fmap_surround_to_map(fm) < 0) return fail_final(fm);
        float pose[6];
        lslam_isometry_to_pose(T, pose);
        lslam_stats st;
        const int rc = lslam_scanmatch_scan(ctx, cc.data(), mc, cs.data(), ms, 16, pose, &opts, &st);
        if (rc < 0) return fail_final(fm);
        if (rc != LSLAM_TOO_FEW_REF) lslam_pose_to_isometry(pose, T);  // written back also when the match failed (:342-346)
        ok = rc == LSLAM_OK;
      }
      if (ok || (bootstrap && !enough)) {
        if (lslam_fmap_add_feature_cloud(fm, kf->cornerCloud.data(), kc, kf->surfCloud.data(), ks, 16, T) < 0) return fail_final(fm);
        ++added;
      }
      matched.push_back(ok ? 1 : 0);
      poses.push_back(from_float16(T));
    }
    if (!directory.empty() && lslam_fmap_save(fm, directory.c_str()) < 0) return fail_final(fm);
    if (map_out) *map_out = fm;
    else lslam_fmap_destroy(fm);
    return added;
  }

  LoopDetector loop_detector;
  KeyframeUpdater keyframe_updater;
  std::vector<KeyFrame::Ptr> keyframes, new_keyframes;
  std::deque<KeyFrame::Ptr> keyframe_queue;
  std::vector<Loop::Ptr> loops;
  Mat4d tf_odom2graph;
  int last_iterations = 0;
  const std::string &lastError() const { return _err; }

private:
  int fail_final(lslam_fmap *fm) {
    _err = lslam_last_error();
    if (fm) lslam_fmap_destroy(fm);
    return -1;
  }
  // graph.cpp:248-297
  bool flush_keyframe_queue() {
    if (keyframe_queue.empty()) return false;
    const Mat4d odom2map = tf_odom2graph;
    const int n = std::min<int>((int)keyframe_queue.size(), _max_per_update);
    static const double ODOM_INFO[6] = {0.8, 0.4, 0.8, 1.0, 2.0, 1.0};  // graph.cpp:279-288
    for (int i = 0; i < n; ++i) {
      KeyFrame::Ptr kf = keyframe_queue[(size_t)i];
      new_keyframes.push_back(kf);
      kf->estimate = odom2map * kf->odom;
      double p7[7];
      mat_to_pose7(kf->estimate, p7);
      kf->node = (int)(_poses.size() / 7);
      _poses.insert(_poses.end(), p7, p7 + 7);
      if (i == 0 && keyframes.empty()) continue;
      const KeyFrame::Ptr &prev = i == 0 ? keyframes.back() : keyframe_queue[(size_t)i - 1];
      add_edge(prev->node, kf->node, inverse(prev->odom) * kf->odom, ODOM_INFO);
    }
    keyframe_queue.erase(keyframe_queue.begin(), keyframe_queue.begin() + n);
    return true;
  }
  void add_edge(int a, int b, const Mat4d &rel, const double diag_info[6]) {
    _ij.push_back(a);
    _ij.push_back(b);
    double p7[7];
    mat_to_pose7(rel, p7);
    _meas.insert(_meas.end(), p7, p7 + 7);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) _info.push_back(r == c ? diag_info[r] : 0.0);
  }
  int _device, _max_per_update;
  std::vector<double> _poses, _meas, _info;
  std::vector<int32_t> _ij;
  std::string _err;
};

}  // namespace pose_graph
