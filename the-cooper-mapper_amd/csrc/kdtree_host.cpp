// kdtree_host.cpp -- host-side kd-tree builder of the product library.
//
// Produces, in the device node format of lslam_device.hpp, the tree that
// nanoflann v1.2.3 builds for the same cloud (util/nanoflann.hpp:931-1078,
// leaf_max_size 10): same split dimension / value / balance rule, same in-place
// permutation of the index array, nodes numbered in allocation (pre-)order.
// Keeping nanoflann's topology AND leaf order makes the device traversal return
// the same neighbour lists as the reference even when squared distances tie.
#include <algorithm>
#include <cstring>

#include "lslam_internal.hpp"

namespace lslam {
namespace {

struct Box { float lo[3], hi[3]; };

struct Builder {
  const float *pts;
  size_t stride;
  std::vector<KdNode> &nodes;
  std::vector<int32_t> &vind;
  int depth = 0;

  float at(int32_t i, int d) const { return pts[(size_t)i * stride + d]; }

  void minmax(const int32_t *ind, int32_t count, int d, float &mn, float &mx) const {
    mn = mx = at(ind[0], d);
    for (int32_t i = 1; i < count; ++i) {
      const float v = at(ind[i], d);
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
  }

  // nanoflann.hpp:1043-1078: two Hoare-style passes -> [< cut | == cut | > cut]
  void plane_split(int32_t *ind, int32_t count, int d, float cut, int32_t &lim1, int32_t &lim2) const {
    int32_t l = 0, r = count - 1;
    for (;;) {
      while (l <= r && at(ind[l], d) < cut) ++l;
      while (r && l <= r && at(ind[r], d) >= cut) --r;
      if (l > r || !r) break;
      std::swap(ind[l], ind[r]);
      ++l; --r;
    }
    lim1 = l;
    r = count - 1;
    for (;;) {
      while (l <= r && at(ind[l], d) <= cut) ++l;
      while (r && l <= r && at(ind[r], d) > cut) --r;
      if (l > r || !r) break;
      std::swap(ind[l], ind[r]);
      ++l; --r;
    }
    lim2 = l;
  }

  // nanoflann.hpp:931-980 divideTree / :982-1031 middleSplit_
  int32_t divide(int32_t left, int32_t right, Box &bb, int level) {
    const int32_t self = (int32_t)nodes.size();
    nodes.push_back(KdNode{0.f, 0.f, 0, 0});
    depth = std::max(depth, level);
    const int32_t count = right - left;
    if (count <= 10) {
      nodes[self].a = left;
      nodes[self].b = ~right;
      for (int d = 0; d < 3; ++d) bb.lo[d] = bb.hi[d] = at(vind[left], d);
      for (int32_t k = left + 1; k < right; ++k)
        for (int d = 0; d < 3; ++d) {
          const float v = at(vind[k], d);
          if (bb.lo[d] > v) bb.lo[d] = v;
          if (bb.hi[d] < v) bb.hi[d] = v;
        }
      return self;
    }
    int32_t *ind = vind.data() + left;
    const float EPS = 0.00001f;
    float max_span = bb.hi[0] - bb.lo[0];
    for (int d = 1; d < 3; ++d) max_span = std::max(max_span, bb.hi[d] - bb.lo[d]);
    float max_spread = -1;
    int cutfeat = 0;
    for (int d = 0; d < 3; ++d) {
      const float span = bb.hi[d] - bb.lo[d];
      if (span > (1 - EPS) * max_span) {
        float mn, mx;
        minmax(ind, count, d, mn, mx);
        const float spread = mx - mn;
        if (spread > max_spread) { cutfeat = d; max_spread = spread; }
      }
    }
    const float split_val = (bb.lo[cutfeat] + bb.hi[cutfeat]) / 2;
    float mn, mx;
    minmax(ind, count, cutfeat, mn, mx);
    const float cutval = split_val < mn ? mn : (split_val > mx ? mx : split_val);
    int32_t lim1, lim2;
    plane_split(ind, count, cutfeat, cutval, lim1, lim2);
    const int32_t idx = lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);

    Box lb = bb;
    lb.hi[cutfeat] = cutval;
    divide(left, left + idx, lb, level + 1);  // child1 == self + 1
    Box rb = bb;
    rb.lo[cutfeat] = cutval;
    const int32_t c2 = divide(left + idx, right, rb, level + 1);

    KdNode &nd = nodes[self];
    nd.lo = lb.hi[cutfeat];  // divlow
    nd.hi = rb.lo[cutfeat];  // divhigh
    nd.a = c2;
    nd.b = cutfeat;
    for (int d = 0; d < 3; ++d) {
      bb.lo[d] = std::min(lb.lo[d], rb.lo[d]);
      bb.hi[d] = std::max(lb.hi[d], rb.hi[d]);
    }
    return self;
  }
};

}  // namespace

void build_kdtree_host(const float *pts, size_t n, size_t stride_floats, HostTree &out) {
  out.nodes.clear();
  out.vind.resize(n);
  for (size_t i = 0; i < n; ++i) out.vind[i] = (int32_t)i;
  out.depth = 0;
  for (int d = 0; d < 3; ++d) out.bb_lo[d] = out.bb_hi[d] = 0.f;
  if (n == 0) return;
  out.nodes.reserve(n / 3 + 16);
  Box bb;
  for (int d = 0; d < 3; ++d) bb.lo[d] = bb.hi[d] = pts[d];
  for (size_t k = 1; k < n; ++k)
    for (int d = 0; d < 3; ++d) {
      const float v = pts[k * stride_floats + d];
      if (v < bb.lo[d]) bb.lo[d] = v;
      if (v > bb.hi[d]) bb.hi[d] = v;
    }
  Builder b{pts, stride_floats, out.nodes, out.vind};
  b.divide(0, (int32_t)n, bb, 1);
  out.depth = b.depth;
  for (int d = 0; d < 3; ++d) { out.bb_lo[d] = bb.lo[d]; out.bb_hi[d] = bb.hi[d]; }
}

}  // namespace lslam
