// The mapping node in C++: include/lslam_pipeline.hpp's LaserMapping mirror (the reference's LaserMapping::process:
// transformMerge, VoxelGrid of the two feature clouds, FeatureMap::update, surround -> search structure, scanMatchScan,
// addFeatureCloud) called in a loop with no interpreter around it -- what a nodelet written against the mirror pays per sweep.
// Input: a file written by tools/mapping_node_bench.py (surround clouds of the bench map, one sweep's less-sharp / less-flat
// clouds, the pose); output: median / p99 / worst milliseconds per LaserMapping::process over N frames.
// Diagnostics, not a bench line.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "lslam_pipeline.hpp"
#include "lslam_scan_match.hpp"

static bool read_floats(FILE *f, std::vector<float> &v) {
  uint32_t n = 0;
  if (std::fread(&n, 4, 1, f) != 1) return false;
  v.resize(n);
  return n == 0 || std::fread(v.data(), 4, n, f) == n;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const int frames = argc > 2 ? std::atoi(argv[2]) : 400;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<float> sur_c, sur_s, less_sharp, less_flat, pose;  // pose: 16 floats (row-major 4x4), then 16 of the perturbation
  if (!read_floats(f, sur_c) || !read_floats(f, sur_s) || !read_floats(f, less_sharp) || !read_floats(f, less_flat) || !read_floats(f, pose) ||
      pose.size() != 32) {
    std::fprintf(stderr, "bad input file\n");
    return 2;
  }
  std::fclose(f);
  lidar_slam::ScanMatch sm(10);
  if (!sm.ok()) {
    std::fprintf(stderr, "backend unavailable: %s\n", sm.initError().c_str());
    return 1;
  }
  // bench.py's map: 21 x 21 x 11 cubes, corner / surf / map filters 0.2 / 0.4 / 0.6, frame filters 1.0 / 1.0
  lidar_slam::LaserMapping mapping(sm.context(), 21, 21, 11, 1.0f, 1.0f, 0.2f, 0.4f, 0.6f);
  const float pos[3] = {pose[3], pose[7], pose[11]};
  float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (lslam_fmap_update(mapping.featureMap(), pos) < 0 ||
      lslam_fmap_add_feature_cloud(mapping.featureMap(), sur_c.data(), sur_c.size() / 4, sur_s.data(), sur_s.size() / 4, 16, I) < 0) {
    std::fprintf(stderr, "map set-up failed: %s\n", lslam_last_error());
    return 1;
  }
  auto map_sums = [&](int k) {
    size_t nc = 0, ns = 0;
    lslam_fmap_surround_counts(mapping.featureMap(), &nc, &ns);
    std::vector<float> c4(4 * nc + 4), s4(4 * ns + 4);
    lslam_fmap_get_surround(mapping.featureMap(), c4.data(), nc, s4.data(), ns);
    auto sums = [](const std::vector<float> &v, size_t n, unsigned long long &ordered, unsigned long long &free_) {
      ordered = free_ = 0;
      for (size_t i = 0; i < 4 * n; ++i) {
        uint32_t b;
        std::memcpy(&b, &v[i], 4);
        ordered += (unsigned long long)b * (unsigned long long)(i + 1);
        free_ += (unsigned long long)b * (unsigned long long)((i & 3) + 1);
      }
    };
    unsigned long long oc, fc, os, fs;
    sums(c4, nc, oc, fc);
    sums(s4, ns, os, fs);
    std::printf("MAP %d corner %zu ordered %llx free %llx | surf %zu ordered %llx free %llx\n", k, nc, oc, fc, ns, os, fs);
  };
  if (argc > 3 && !std::strcmp(argv[3], "trace")) map_sums(-1);
  // odometry inputs alternate between T and T * delta: every frame starts a perturbation away from where the last one ended
  const float *A = pose.data(), *B = pose.data() + 16;
  std::vector<double> ms;
  int iters = 0;
  long sum_iters = 0, sum_sweeps = 0, sum_rows = 0;
  for (int k = 0; k < frames + 20; ++k) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!mapping.process(less_sharp, less_flat, (k & 1) ? B : A)) {
      std::fprintf(stderr, "mapping failed: %s\n", mapping.lastError().c_str());
      return 1;
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (k >= 20) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
    iters = mapping.lastStats().iterations;
    sum_iters += mapping.lastStats().iterations;
    sum_sweeps += mapping.lastStats().sweeps;
    sum_rows += mapping.lastStats().n_rows;
    if (argc > 3 && !std::strcmp(argv[3], "trace") && k < 3) {
      {  // the frame's inputs to the map insert: the VoxelGrid of the two clouds (recomputed here) and the whole pose
        std::vector<float> o1(less_sharp.size() + 4), o2(less_flat.size() + 4);
        size_t n1 = 0, n2 = 0;
        lslam_voxel_grid(sm.context(), less_sharp.data(), less_sharp.size() / 4, 16, 1.0f, o1.data(), less_sharp.size() / 4, &n1);
        lslam_voxel_grid(sm.context(), less_flat.data(), less_flat.size() / 4, 16, 1.0f, o2.data(), less_flat.size() / 4, &n2);
        unsigned long long h1 = 0, h2 = 0, hp = 0;
        for (size_t i = 0; i < 4 * n1; ++i) { uint32_t b; std::memcpy(&b, &o1[i], 4); h1 += (unsigned long long)b * (i + 1); }
        for (size_t i = 0; i < 4 * n2; ++i) { uint32_t b; std::memcpy(&b, &o2[i], 4); h2 += (unsigned long long)b * (i + 1); }
        const float *Tk = mapping.lidarMapped();
        for (int i = 0; i < 16; ++i) { uint32_t b; std::memcpy(&b, &Tk[i], 4); hp += (unsigned long long)b * (i + 1); }
        std::printf("INPUTS %d voxel(less sharp) %zu %llx voxel(less flat) %zu %llx pose16 %llx\n", k, n1, h1, n2, h2, hp);
      }
      map_sums(k);
      int64_t mg = 0, rs = 0;
      lslam_fmap_rebuild_stats(mapping.featureMap(), &mg, &rs);
      std::printf("REBUILDS %d merged %lld resorted %lld\n", k, (long long)mg, (long long)rs);
    }
    if (argc > 3 && !std::strcmp(argv[3], "trace") && k < 60) {
      const float *Tk = mapping.lidarMapped();
      std::printf("TRACE %d it %d rows %d line %d plane %d pose %a %a %a\n", k, mapping.lastStats().iterations, mapping.lastStats().n_rows,
                  mapping.lastStats().n_line, mapping.lastStats().n_plane, (double)Tk[3], (double)Tk[7], (double)Tk[11]);
    }
  }
  {
    uint64_t lz[3] = {0, 0, 0};
    lslam_debug_lazy_trees(sm.context(), lz);
    uint64_t cs[3] = {0, 0, 0};
    lslam_debug_cert_stats(sm.context(), cs);
    std::printf("  points left to the second pass %llu of %llu swept (LSLAM_DEBUG_CERT_STATS=1)\n", (unsigned long long)cs[0], (unsigned long long)cs[1]);
    uint64_t sv[8] = {0};
    lslam_debug_sweep_launches(sm.context(), sv);
    std::printf("  after the process() loop: maps set without trees %llu, trees built after all %llu; GN iterations %ld, sweeps %ld, rows %ld; grid launches %llu, tree-sweep launches %llu %llu %llu\n",
                (unsigned long long)lz[0], (unsigned long long)lz[1], sum_iters, sum_sweeps, sum_rows, (unsigned long long)lslam_debug_grid_launches(sm.context()),
                (unsigned long long)sv[0], (unsigned long long)sv[1], (unsigned long long)sv[2]);
  }
  if (argc > 3 && !std::strcmp(argv[3], "steps")) {  // the same frame with a clock between the calls (argv[3] present): LaserMapping::process's body, step by step
    std::vector<float> dc(less_sharp.size() + 4), ds(less_flat.size() + 4);
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto msd = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    lslam_opts o;
    lslam_default_opts(&o);
    o.delta_t_abort = o.delta_r_abort = 0.1f;
    o.use_score = 0;
    lslam_stats st;
    float odomLast[16], mappedLast[16], mappedNew[16];
    std::memcpy(odomLast, A, 64);
    std::memcpy(mappedLast, mapping.lidarMapped(), 64);
    for (int k = 1; k < 221; ++k) {
      const float *odomNew = (k & 1) ? B : A;
      size_t nc = 0, ns = 0;
      auto t0 = now();
      lslam_transform_associate(odomLast, odomNew, mappedLast, mappedNew);
      lslam_voxel_grid(sm.context(), less_sharp.data(), less_sharp.size() / 4, 16, 1.0f, dc.data(), less_sharp.size() / 4, &nc);
      auto t1 = now();
      lslam_voxel_grid(sm.context(), less_flat.data(), less_flat.size() / 4, 16, 1.0f, ds.data(), less_flat.size() / 4, &ns);
      auto t2 = now();
      const float p3[3] = {mappedNew[3], mappedNew[7], mappedNew[11]};
      lslam_fmap_update(mapping.featureMap(), p3);
      size_t a_ = 0, b_ = 0;
      lslam_fmap_surround_counts(mapping.featureMap(), &a_, &b_);
      auto t3 = now();
      lslam_fmap_surround_to_map(mapping.featureMap());
      auto t4 = now();
      float p6[6];
      lslam_isometry_to_pose(mappedNew, p6);
      lslam_scanmatch_scan(sm.context(), dc.data(), nc, ds.data(), ns, 16, p6, &o, &st);
      lslam_pose_to_isometry(p6, mappedNew);
      auto t5 = now();
      std::memcpy(mappedLast, mappedNew, 64);
      std::memcpy(odomLast, odomNew, 64);
      lslam_fmap_add_feature_cloud(mapping.featureMap(), dc.data(), nc, ds.data(), ns, 16, mappedNew);
      auto t6 = now();
      if (k >= 21) { acc[0] += msd(t0, t1); acc[1] += msd(t1, t2); acc[2] += msd(t2, t3); acc[3] += msd(t3, t4); acc[4] += msd(t4, t5); acc[5] += msd(t5, t6); acc[6] += 1; }
    }
    std::printf("  step by step, mean ms: VoxelGrid(less sharp) %.3f  VoxelGrid(less flat) %.3f  update + counts %.3f  surround_to_map %.3f  scanMatchScan %.3f  addFeatureCloud %.3f\n",
                acc[0] / acc[6], acc[1] / acc[6], acc[2] / acc[6], acc[3] / acc[6], acc[4] / acc[6], acc[5] / acc[6]);
  }
  std::sort(ms.begin(), ms.end());
  const float *T = mapping.lidarMapped();
  std::printf("LaserMapping::process in C++: %d frames, median %.3f ms, p99 %.3f ms, worst %.3f ms (GN iterations of the last frame %d; pose %.3f %.3f %.3f)\n",
              (int)ms.size(), ms[ms.size() / 2], ms[(size_t)(0.99 * (ms.size() - 1))], ms.back(), iters, T[3], T[7], T[11]);
  return 0;
}
