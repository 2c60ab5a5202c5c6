// kdtree_host.cpp -- host-side kd-tree builder of the product library.
//
// Builds the tree nanoflann v1.2.3 builds for the same cloud
// (util/nanoflann.hpp:931-1078, leaf_max_size 10): same split dimension / value /
// balance rule and the same in-place permutation of the index array, so that the
// device traversal returns the reference's neighbour lists even when squared
// distances tie.  The result is then re-encoded for the GPU (lslam_device.hpp):
// inner nodes only, 32-bit child references, breadth-first groups of 8 nodes per
// 128-byte line.
#include <algorithm>
#include <cstring>
#include <deque>

#include "lslam_internal.hpp"

namespace lslam {
namespace {

struct Box { float lo[3], hi[3]; };

// logical node, numbered in nanoflann's allocation (pre-)order
struct LNode {
  float lo, hi;       // divlow, divhigh
  int32_t c1, c2;     // children (logical ids); -1 for a leaf
  int32_t left, right;  // leaf range in vind
  int32_t feat;
};

struct Builder {
  const float *pts;
  size_t stride;
  std::vector<LNode> &nodes;
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
    nodes.push_back(LNode{0.f, 0.f, -1, -1, left, right, 0});
    depth = std::max(depth, level);
    const int32_t count = right - left;
    if (count <= 10) {
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
    const int32_t c1 = divide(left, left + idx, lb, level + 1);
    Box rb = bb;
    rb.lo[cutfeat] = cutval;
    const int32_t c2 = divide(left + idx, right, rb, level + 1);

    LNode &nd = nodes[self];
    nd.lo = lb.hi[cutfeat];  // divlow
    nd.hi = rb.lo[cutfeat];  // divhigh
    nd.c1 = c1;
    nd.c2 = c2;
    nd.feat = cutfeat;
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
  out.n_leaves = 0;
  out.root_ref = KD_LEAF;  // empty leaf
  for (int d = 0; d < 3; ++d) out.bb_lo[d] = out.bb_hi[d] = 0.f;
  if (n == 0) return;
  std::vector<LNode> ln;
  ln.reserve(n / 3 + 16);
  Box bb;
  for (int d = 0; d < 3; ++d) bb.lo[d] = bb.hi[d] = pts[d];
  for (size_t k = 1; k < n; ++k)
    for (int d = 0; d < 3; ++d) {
      const float v = pts[k * stride_floats + d];
      if (v < bb.lo[d]) bb.lo[d] = v;
      if (v > bb.hi[d]) bb.hi[d] = v;
    }
  Builder b{pts, stride_floats, ln, out.vind};
  b.divide(0, (int32_t)n, bb, 1);
  out.depth = b.depth;
  for (int d = 0; d < 3; ++d) { out.bb_lo[d] = bb.lo[d]; out.bb_hi[d] = bb.hi[d]; }

  // ---- re-encode: inner nodes only, breadth-first groups of 8 per 128-byte line ----
  std::vector<int32_t> slot(ln.size(), -1);
  size_t n_inner = 0;
  for (const LNode &nd : ln) (nd.c1 >= 0 ? n_inner : out.n_leaves) += 1;
  int32_t next = 0;
  if (ln[0].c1 >= 0) {
    std::deque<int32_t> roots{0};
    std::vector<int32_t> group;
    while (!roots.empty()) {
      const int32_t r = roots.front();
      roots.pop_front();
      // up to 8 inner nodes of r's subtree in breadth-first order
      group.clear();
      group.push_back(r);
      for (size_t h = 0; h < group.size(); ++h) {
        const LNode &nd = ln[group[h]];
        for (int32_t c : {nd.c1, nd.c2}) {
          if (ln[c].c1 < 0) continue;  // leaf
          if (group.size() < 8) group.push_back(c);
          else roots.push_back(c);
        }
      }
      // a group never straddles a line: small groups may share one
      const int32_t room = 8 - (next & 7);
      if ((int32_t)group.size() > room) next += room;
      for (int32_t g : group) slot[g] = next++;
      // children of the last members that did not fit were queued above; children of
      // members visited after the group filled up:
    }
  }
  out.nodes.assign((size_t)next, KdNode{0.f, 0.f, KD_LEAF, KD_LEAF});
  auto ref_of = [&](int32_t id) -> uint32_t {
    const LNode &c = ln[id];
    if (c.c1 < 0) return KD_LEAF | ((uint32_t)c.left << 4) | (uint32_t)(c.right - c.left);
    return ((uint32_t)slot[id] << 2) | (uint32_t)c.feat;
  };
  for (size_t i = 0; i < ln.size(); ++i) {
    if (ln[i].c1 < 0) continue;
    KdNode &o = out.nodes[(size_t)slot[i]];
    o.lo = ln[i].lo;
    o.hi = ln[i].hi;
    o.c1 = ref_of(ln[i].c1);
    o.c2 = ref_of(ln[i].c2);
  }
  out.root_ref = ref_of(0);
}

}  // namespace lslam
