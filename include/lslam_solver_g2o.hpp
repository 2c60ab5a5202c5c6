// lslam_solver_g2o.hpp -- header-only C++ shim with the class surface of the reference's
// pose_graph::SolverG2O (/root/reference/L_SLAM/src/pose_graph/solver_g2o.h:46-72,
// solver_g2o.cpp:51-100) over the C ABI (lslam_pg_* in lslam_c.h), so that the call sites
//   pose_graph/graph.cpp:261,289 (add_se3_node / add_se3_edge), :341 (loop edges), :352 (optimize)
// compile against it unchanged.  Templated on the pose / matrix types so that neither Eigen nor g2o
// is needed to build the backend:
//   Isometry : .matrix()(r,c) read/write, default constructible (Eigen::Isometry3d works)
//   Matrix   : (r,c) read access to a 6x6 (Eigen::MatrixXd works)
// Vertices are returned as pointers with ->id() and ->estimate(), like g2o::VertexSE3*.
#pragma once

#include <cmath>
#include <cstdint>
#include <deque>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "lslam_c.h"

namespace pose_graph {

template <typename Isometry, typename Matrix>
class SolverG2OT {
public:
  struct VertexSE3 {
    int id() const { return _id; }
    Isometry estimate() const { return _owner->estimate_of(_id); }
    int _id;
    SolverG2OT *_owner;
  };
  struct EdgeSE3 { int index; };

  explicit SolverG2OT(int device = 0) : is_first(true), _device(device), _pg(nullptr) {}
  ~SolverG2OT() { lslam_pg_destroy(_pg); }
  SolverG2OT(const SolverG2OT &) = delete;
  SolverG2OT &operator=(const SolverG2OT &) = delete;

  // solver_g2o.cpp:51-63 -- the first vertex is fixed
  VertexSE3 *add_se3_node(const Isometry &pose) {
    pull();
    double p[7];
    to_pose7(pose, p);
    _poses.insert(_poses.end(), p, p + 7);
    is_first = false;
    _vertices.push_back(VertexSE3{(int)_vertices.size(), this});
    return &_vertices.back();
  }
  // solver_g2o.cpp:65-77
  EdgeSE3 *add_se3_edge(VertexSE3 *v1, VertexSE3 *v2, const Isometry &relative_pose, const Matrix &information_matrix) {
    pull();
    double z[7];
    to_pose7(relative_pose, z);
    _ij.push_back(v1->id());
    _ij.push_back(v2->id());
    _meas.insert(_meas.end(), z, z + 7);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) _info.push_back(information_matrix(r, c));
    _edges.push_back(EdgeSE3{(int)_edges.size()});
    return &_edges.back();
  }
  // solver_g2o.cpp:79-95: graph->optimize(1000) with "lm_var"
  void optimize() {
    build();
    lslam_pg_stats st;
    std::cout << "\n--- g2o optimization ---\nnodes: " << _vertices.size() << "   edges: " << _edges.size() << std::endl;
    if (lslam_pg_optimize(_pg, 1000, &st) < 0) throw std::runtime_error(std::string("lslam_pg_optimize: ") + lslam_pg_last_error());
    _last = st;
    std::cout << "iterations: " << st.iterations << std::endl;
  }
  // solver_g2o.cpp:97-100
  void save(const std::string &filename) {
    build();
    if (lslam_pg_save_g2o(_pg, filename.c_str()) < 0) throw std::runtime_error(lslam_pg_last_error());
  }
  const lslam_pg_stats &lastStats() const { return _last; }
  // No reference counterpart ("lm_var" factorises: its solves are exact).  rel_tol in (0, 1): the relative residual at which a
  // damped solve's PCG stops from now on; 0 (default) leaves the library's 1e-8.  Looser = an inexact LM: same optimum, another
  // trajectory of iterates (lslam_pg_set_solve_tolerance, include/lslam_c.h).
  void setSolveTolerance(double rel_tol) { _solve_tol = rel_tol; if (_pg && rel_tol > 0.0) (void)lslam_pg_set_solve_tolerance(_pg, rel_tol); }

  bool is_first;

private:
  friend struct VertexSE3;
  static void to_pose7(const Isometry &T, double p[7]) {  // {t, Eigen::Quaterniond(R)} (Shepperd)
    double R[3][3];
    Isometry &M = const_cast<Isometry &>(T);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) R[r][c] = M.matrix()(r, c);
      p[r] = M.matrix()(r, 3);
    }
    const double tr = R[0][0] + R[1][1] + R[2][2];
    double q[4];
    if (tr > 0) {
      double s = std::sqrt(tr + 1.0);
      q[3] = 0.5 * s;
      s = 0.5 / s;
      q[0] = (R[2][1] - R[1][2]) * s; q[1] = (R[0][2] - R[2][0]) * s; q[2] = (R[1][0] - R[0][1]) * s;
    } else {
      int i = 0;
      if (R[1][1] > R[0][0]) i = 1;
      if (R[2][2] > R[i][i]) i = 2;
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      double s = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
      q[i] = 0.5 * s;
      s = 0.5 / s;
      q[3] = (R[k][j] - R[j][k]) * s;
      q[j] = (R[j][i] + R[i][j]) * s;
      q[k] = (R[k][i] + R[i][k]) * s;
    }
    for (int k = 0; k < 4; ++k) p[3 + k] = q[k];
  }
  Isometry estimate_of(int id) {
    pull_keep();
    const double *p = &_poses[(size_t)id * 7];
    const double x = p[3], y = p[4], z = p[5], w = p[6];
    Isometry T;
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                            {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                            {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T.matrix()(r, c) = R[r][c];
      T.matrix()(r, 3) = p[r];
      T.matrix()(3, r) = 0.0;
    }
    T.matrix()(3, 3) = 1.0;
    return T;
  }
  void build() {
    if (_pg) return;
    if (_vertices.empty()) throw std::runtime_error("empty pose graph");
    if (lslam_pg_create(_device, (int32_t)_vertices.size(), _poses.data(), (int32_t)_edges.size(), _ij.data(), _meas.data(),
                        _info.data(), 0, &_pg) != LSLAM_OK)
      throw std::runtime_error(std::string("lslam_pg_create: ") + lslam_pg_last_error());
    if (_solve_tol > 0.0) (void)lslam_pg_set_solve_tolerance(_pg, _solve_tol);
  }
  void pull_keep() {  // current estimates into _poses, device graph kept
    if (_pg) lslam_pg_get_poses(_pg, _poses.data());
  }
  void pull() {  // ... and the device graph dropped: the topology is about to change
    if (!_pg) return;
    pull_keep();
    lslam_pg_destroy(_pg);
    _pg = nullptr;
  }

  int _device;
  lslam_pg *_pg;
  lslam_pg_stats _last{};
  double _solve_tol = 0.0;
  std::deque<VertexSE3> _vertices;  // deque: pointers handed out stay valid
  std::deque<EdgeSE3> _edges;
  std::vector<double> _poses, _meas, _info;
  std::vector<int32_t> _ij;
};

}  // namespace pose_graph
