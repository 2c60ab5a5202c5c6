// ref_nanoflann_wrap.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" shim around the REFERENCE's own vendored nanoflann v1.2.3
// (/root/reference/L_SLAM/src/util/nanoflann.hpp, included from where it lies;
// nothing from the reference is copied into this repo).  It instantiates exactly
// the index type the reference uses in util/nanoflann_pcl.h:100-103:
//   KDTreeSingleIndexAdaptor<SO3_Adaptor<float, Adaptor>, Adaptor, 3, int>
// with the default leaf_max_size (10) and SearchParams() (eps = 0), and answers
// kNN queries the way nanoflann_pcl.h:150-162 nearestKSearch does.
//
// Built by oracle/Makefile into oracle/_ref/libref_nanoflann.so (git-ignored);
// used to pin oracle/lslam_oracle.c's kd-tree restatement and to generate
// tests/golden/knn_*.npz.
#include "nanoflann.hpp"

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace {
struct CloudAdaptor {  // mirrors nanoflann_pcl.h:86-93,189-210 PointCloud_Adaptor
  const float *pts;
  size_t n, stride;
  inline size_t kdtree_get_point_count() const { return n; }
  inline float kdtree_get_pt(const size_t idx, int dim) const {
    const float *p = pts + idx * stride;
    if (dim == 0) return p[0];
    else if (dim == 1) return p[1];
    else if (dim == 2) return p[2];
    else return 0.0;
  }
  template <class BBOX> bool kdtree_get_bbox(BBOX &) const { return false; }
};
typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::SO3_Adaptor<float, CloudAdaptor>,
                                            CloudAdaptor, 3, int>
    RefTree;
struct RefIndex {
  CloudAdaptor adaptor;
  RefTree tree;
  explicit RefIndex(const float *p, size_t n, size_t stride)
      : adaptor{p, n, stride}, tree(3, adaptor) {}
};
}  // namespace

extern "C" {
void *ref_kdtree_build(const float *pts, size_t n, size_t stride_floats) {
  RefIndex *r = new RefIndex(pts, n, stride_floats);
  r->tree.buildIndex();  // nanoflann_pcl.h:141-148 setInputCloud
  return r;
}
void ref_kdtree_free(void *h) { delete static_cast<RefIndex *>(h); }
int ref_kdtree_knn(void *h, const float *q, int k, int32_t *idx_out, float *d2_out) {
  RefIndex *r = static_cast<RefIndex *>(h);
  nanoflann::KNNResultSet<float, int> rs(k);
  rs.init(idx_out, d2_out);
  r->tree.findNeighbors(rs, q, nanoflann::SearchParams());
  return (int)rs.size();
}
// KdTreeFLANN::radiusSearch (nanoflann_pcl.h:164-186): RadiusResultSet(radius) -- `radius` is compared
// with SQUARED distances --, SearchParams() (sorted = true), then std::sort by distance.
int ref_kdtree_radius(void *h, const float *q, float radius, int32_t *idx_out, float *d2_out, int cap) {
  RefIndex *r = static_cast<RefIndex *>(h);
  std::vector<std::pair<int, float>> found;
  found.reserve(128);
  nanoflann::RadiusResultSet<float, int> rs(radius, found);
  const nanoflann::SearchParams params;
  r->tree.findNeighbors(rs, q, params);
  const size_t n = found.size();  // everything the result set collected (see ref_kdtree_radius_nfound)
  if (params.sorted) std::sort(found.begin(), found.end(), nanoflann::IndexDist_Sorter());
  for (size_t i = 0; i < n && (int)i < cap; ++i) {
    idx_out[i] = found[i].first;
    d2_out[i] = found[i].second;
  }
  return (int)n;
}
// The count KdTreeFLANN::radiusSearch itself returns (nanoflann_pcl.h:173: `const size_t nFound =
// _kdtree.findNeighbors(resultSet, ...)`), i.e. the value of findNeighbors converted to size_t.
int ref_kdtree_radius_nfound(void *h, const float *q, float radius) {
  RefIndex *r = static_cast<RefIndex *>(h);
  std::vector<std::pair<int, float>> found;
  nanoflann::RadiusResultSet<float, int> rs(radius, found);
  const size_t nFound = r->tree.findNeighbors(rs, q, nanoflann::SearchParams());
  return (int)nFound;
}
// batch helper so Python does not pay one ctypes call per query
void ref_kdtree_knn_batch(void *h, const float *q, size_t nq, size_t q_stride, int k,
                          int32_t *idx_out, float *d2_out) {
  for (size_t i = 0; i < nq; ++i)
    ref_kdtree_knn(h, q + i * q_stride, k, idx_out + i * k, d2_out + i * k);
}
}
