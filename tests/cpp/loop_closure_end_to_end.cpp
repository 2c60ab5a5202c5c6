// C++ drop-in check of include/lslam_loop_closure.hpp: pose_graph::Graph + LoopDetector + KeyframeUpdater over the
// C ABI.  Reads a stream of frames (16 doubles odometry pose, then corner and surf clouds as uint32 count + count x
// {x,y,z,intensity} floats) written by the test, drives Graph::add_frame / Graph::optimize per frame like the
// reference's node loop (graph.cpp:230-246, 313-383) and prints the loops found and every keyframe's estimate, which
// the test compares with the Python mirror of the same bookkeeping.
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "lslam_loop_closure.hpp"

static bool read_cloud(FILE *f, std::vector<float> &c) {
  uint32_t n = 0;
  if (std::fread(&n, 4, 1, f) != 1) return false;
  c.resize(4 * (size_t)n);
  return n == 0 || std::fread(c.data(), 16, n, f) == n;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  lslam_ctx *ctx = nullptr;
  if (lslam_ctx_create(0, &ctx) != LSLAM_OK) {
    std::fprintf(stderr, "backend unavailable: %s\n", lslam_last_error());
    return 1;
  }
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  pose_graph::Graph g(ctx);
  g.loop_detector.accum_distance_thresh = std::atof(argv[2]);
  int n_loops = 0, frames = 0;
  pose_graph::Mat4d odom;
  std::vector<float> corner, surf;
  while (std::fread(odom.m, sizeof(double), 16, f) == 16 && read_cloud(f, corner) && read_cloud(f, surf)) {
    if (g.add_frame(odom, corner, surf)) {
      const int found = g.optimize(20);
      if (found < 0) {
        std::fprintf(stderr, "optimize failed: %s\n", g.lastError().c_str());
        return 1;
      }
      n_loops += found;
    }
    ++frames;
  }
  std::fclose(f);
  std::printf("LOOPS %d FRAMES %d KEYFRAMES %zu\n", n_loops, frames, g.keyframes.size());
  for (size_t i = 0; i < g.keyframes.size(); ++i)
    std::printf("KF %zu %.17g %.17g %.17g\n", i, g.keyframes[i]->estimate(0, 3), g.keyframes[i]->estimate(1, 3),
                g.keyframes[i]->estimate(2, 3));
  if (argc > 3) {  // Graph::getFinalFeatureMap (graph.cpp:150-199) into the directory given, with the bootstrap
    std::vector<char> matched;
    std::vector<pose_graph::Mat4d> poses;
    const int added = g.getFinalFeatureMap(ctx, argv[3], true, matched, poses, nullptr);
    if (added < 0) {
      std::fprintf(stderr, "getFinalFeatureMap failed: %s\n", g.lastError().c_str());
      return 1;
    }
    int n_matched = 0;
    for (char m : matched) n_matched += m;
    std::printf("FINAL %d %d\n", added, n_matched);
    for (size_t i = 0; i < poses.size(); ++i)
      std::printf("FP %zu %d %.9g %.9g %.9g\n", i, (int)matched[i], poses[i](0, 3), poses[i](1, 3), poses[i](2, 3));
  }
  lslam_ctx_destroy(ctx);
  return 0;
}
