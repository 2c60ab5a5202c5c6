// C++ drop-in check on a GPU: the three header shims (ScanMatch, FeatureMap, SolverG2O) over the C ABI
// with stand-in cloud / pose types (no PCL, no Eigen).  Prints "OK ..." lines that the test parses.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <random>
#include <vector>

#include "lslam_feature_map.hpp"
#include "lslam_scan_match.hpp"
#include "lslam_solver_g2o.hpp"

struct Pt { float x, y, z, pad, intensity, p1, p2, p3; };  // pcl::PointXYZI layout: 32 bytes, intensity at 16
struct Cloud { std::vector<Pt> points; void clear() { points.clear(); } };
template <typename S> struct IsoT {
  S m[16];
  struct M { S *m; S &operator()(int r, int c) { return m[r * 4 + c]; } };
  M matrix() { return M{m}; }
  IsoT() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? S(1) : S(0); }
};
struct Mat6 { double v[36]; double operator()(int r, int c) const { return v[r * 6 + c]; } };

int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  // a synthetic room: floor + two walls as surf points, wall corner + poles as corner points
  std::shared_ptr<Cloud> mc(new Cloud), ms(new Cloud), qc(new Cloud), qs(new Cloud);
  auto add = [](Cloud &c, float x, float y, float z) { Pt p{}; p.x = x; p.y = y; p.z = z; p.intensity = 0; c.points.push_back(p); };
  for (int i = 0; i < 40000; ++i) {
    add(*ms, 20 * u(rng), 20 * u(rng), 0.01f * u(rng));
    add(*ms, 20.f + 0.01f * u(rng), 20 * u(rng), 5 * (u(rng) + 1));
    add(*ms, 20 * u(rng), -20.f + 0.01f * u(rng), 5 * (u(rng) + 1));
  }
  for (int k = 0; k < 12; ++k)
    for (int i = 0; i < 300; ++i) add(*mc, -15.f + 3 * k + 0.005f * u(rng), 10.f + 0.005f * u(rng), 5 * (u(rng) + 1));
  // the "scan": a subset of the map moved by a known small transform (sensor frame = map frame shifted)
  const float dx = 0.15f, dy = -0.1f, yaw = 0.01f;
  auto to_scan = [&](const Pt &p, Cloud &c) {
    const float x = p.x - dx, y = p.y - dy;
    add(c, std::cos(yaw) * x + std::sin(yaw) * y, -std::sin(yaw) * x + std::cos(yaw) * y, p.z);
  };
  for (size_t i = 0; i < ms->points.size(); i += 7) to_scan(ms->points[i], *qs);
  for (size_t i = 0; i < mc->points.size(); i += 3) to_scan(mc->points[i], *qc);

  lidar_slam::ScanMatch sm(10);  // never throws: a missing backend shows in ok() and in every match returning false
  if (!sm.ok()) {
    std::fprintf(stderr, "backend unavailable: %s\n", sm.initError().c_str());
    return 1;
  }
  sm.setConvergeThreshold(0.1f, 0.1f);
  // --- FeatureMap shim: push the map, get the surround back, hand it to the matcher on the device
  lidar_slam::FeatureMap<Pt, Cloud> fmap(sm.context());
  fmap.setupFilterSize(0.2f, 0.4f, 0.6f);
  Pt origin{};
  fmap.update(origin);
  IsoT<float> I;
  fmap.addFeatureCloud(*mc, *ms, I);
  Cloud sc, ss;
  fmap.getSurroundFeature(sc, ss);
  std::printf("OK surround %zu %zu\n", sc.points.size(), ss.points.size());
  // --- ScanMatch shim (Isometry overload) against the surround clouds
  std::shared_ptr<const Cloud> rc(new Cloud(sc)), rs(new Cloud(ss)), c1(qc), s1(qs);
  IsoT<float> pose;
  const bool ok = sm.scanMatchScan(rc, rs, c1, s1, pose);
  std::printf("OK match %d %.4f %.4f\n", ok ? 1 : 0, pose.m[3], pose.m[7]);
  // setReferenceEpoch: the same reference clouds under the same epoch -- the second call skips their upload (one map set fewer)
  // and returns the same bits
  {
    uint64_t lazy0[3], lazy1[3], lazy2[3];
    sm.setReferenceEpoch(7);
    IsoT<float> pa, pb;
    lslam_debug_lazy_trees(sm.context(), lazy0);
    const bool oka = sm.scanMatchScan(rc, rs, c1, s1, pa);
    lslam_debug_lazy_trees(sm.context(), lazy1);
    const bool okb = sm.scanMatchScan(rc, rs, c1, s1, pb);
    lslam_debug_lazy_trees(sm.context(), lazy2);
    bool same = oka == okb && oka == ok;
    for (int i = 0; i < 16; ++i) same = same && pa.m[i] == pb.m[i] && pa.m[i] == pose.m[i];
    std::printf("OK epoch %d %d %d\n", same ? 1 : 0, (int)(lazy1[0] - lazy0[0]), (int)(lazy2[0] - lazy1[0]));
    sm.setReferenceEpoch(0);
  }
  // --- SolverG2O shim: a square of four poses with a drifted guess and a loop edge
  typedef pose_graph::SolverG2OT<IsoT<double>, Mat6> Solver;
  Solver solver;
  Mat6 info{};
  for (int i = 0; i < 6; ++i) info.v[i * 7] = 1.0;
  std::vector<Solver::VertexSE3 *> v;
  const double gx[5] = {0, 1, 1, 0, 0}, gy[5] = {0, 0, 1, 1, 0};
  for (int i = 0; i < 5; ++i) {
    IsoT<double> T;
    T.m[3] = gx[i] + 0.05 * i;
    T.m[7] = gy[i] - 0.04 * i;
    v.push_back(solver.add_se3_node(T));
  }
  for (int i = 0; i + 1 < 5; ++i) {
    IsoT<double> Z;
    Z.m[3] = gx[i + 1] - gx[i];
    Z.m[7] = gy[i + 1] - gy[i];
    solver.add_se3_edge(v[i], v[i + 1], Z, info);
  }
  IsoT<double> Zl;  // vertex 4 is vertex 0 revisited
  solver.add_se3_edge(v[0], v[4], Zl, info);
  solver.optimize();
  IsoT<double> e4 = v[4]->estimate();
  std::printf("OK graph %.5f %.5f %d\n", e4.m[3], e4.m[7], solver.lastStats().iterations);
  return 0;
}
