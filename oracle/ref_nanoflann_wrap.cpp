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

#include <cstddef>
#include <cstdint>

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
// batch helper so Python does not pay one ctypes call per query
void ref_kdtree_knn_batch(void *h, const float *q, size_t nq, size_t q_stride, int k,
                          int32_t *idx_out, float *d2_out) {
  for (size_t i = 0; i < nq; ++i)
    ref_kdtree_knn(h, q + i * q_stride, k, idx_out + i * k, d2_out + i * k);
}
}
