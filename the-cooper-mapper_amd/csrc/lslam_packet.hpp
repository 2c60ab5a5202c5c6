// lslam_packet.hpp -- exact 5-NN for 64 neighbouring queries at once ("packet" search, gfx950).
//
// Why: one query per lane walks the kd-tree with divergent 16-byte gathers, and the vector memory
// pipe of a CU retires about one such lane-load per clock whether it hits L1 or not
// (tools/ubench_gather.hip: 65-77 clk per wave-instruction in L1, 146 from L2, ~600 from the
// Infinity Cache) -- the sweep was bound by that rate, not by any bandwidth.  Scan points are
// Morton-ordered, so the 64 queries of a wavefront are neighbours: summed over the lanes they
// visit ~1500 nodes and ~300 leaves, but only ~45 DIFFERENT nodes and ~17 different leaves.
//
// How: the wavefront walks the tree ONCE for all its lanes.  Control flow, the traversal stack and
// every tree access are wave-uniform: a node (PNode: child references + the children's tight
// bounding boxes, 64 B) and a leaf (<= 10 points, 160 B) arrive by SCALAR loads in SGPRs, each lane
// tests them against its own query.  A subtree is entered if ANY lane's lower bound on the distance
// to its box is within that lane's current 5th distance; a leaf's points are offered to every lane.
//
// Why the result is still nanoflann's, bit for bit (nanoflann.hpp:1303-1323, :1433-1497):
//  * nanoflann's answer is the five smallest candidates under the order (distance, visit order);
//    everything it prunes is strictly farther than its final 5th distance.  Distances are evaluated
//    with the reference's operation sequence (dist2_xyz), so any search that offers every point with
//    distance <= the true 5th distance finds the same five -- up to the order of exact TIES.
//  * The box bound is conservative in fp32: per axis e = max(lo - q, q - hi, 0) <= |q - p| for every p
//    in the box after rounding (rounding is monotone), and the squares are summed in dist2_xyz's order,
//    so bound <= dist2_xyz(q, p): a subtree holding a point at distance <= the 5th is never skipped
//    (the test is `bound <= worst`, equality included, like nanoflann's :1487).
//  * Ties are detected, not ordered: a lane raises `tie` when two of its five distances are equal or a
//    candidate it turned away (or dropped) had exactly its final 5th distance.  The caller re-runs such
//    a lane through knn5_search (nanoflann's own traversal order).  Exact fp32 ties between different
//    map points do not happen in real clouds (they do in the lattice / duplicate test fixtures).
#pragma once

#include "lslam_device.hpp"

namespace lslam {

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// One PNode into SGPRs (address wave-uniform).
LSLAM_DEV u32x16 sload_pnode(const PNode *p) {
  u32x16 r;
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  return r;
}

// A leaf's ten points (the point array is padded: reading ten is always in bounds).
LSLAM_DEV void sload_leaf(const float4 *p, u32x16 &a, u32x16 &b, u32x8 &c) {
  asm volatile(
      "s_load_dwordx16 %0, %3, 0x0\n\t"
      "s_load_dwordx16 %1, %3, 0x40\n\t"
      "s_load_dwordx8 %2, %3, 0x80\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(a), "=&s"(b), "=&s"(c)
      : "s"(p)
      : "memory");
}

// Lower bound of dist2_xyz(q, p) over all p in the box (see the header comment).
LSLAM_DEV float box_bound(float qx, float qy, float qz, float lx, float ly, float lz, float hx, float hy, float hz) {
  const float ex = fmaxf(fmaxf(__fsub_rn(lx, qx), __fsub_rn(qx, hx)), 0.0f);
  const float ey = fmaxf(fmaxf(__fsub_rn(ly, qy), __fsub_rn(qy, hy)), 0.0f);
  const float ez = fmaxf(fmaxf(__fsub_rn(lz, qz), __fsub_rn(qz, hz)), 0.0f);
  float r = __fmul_rn(ex, ex);
  r = __fadd_rn(r, __fmul_rn(ey, ey));
  r = __fadd_rn(r, __fmul_rn(ez, ez));
  return r;
}

#ifdef LSLAM_PACKET_STATS
struct PacketStats { unsigned nodes, leaves, inserts, pops; };
#define PS_INC(f) ps.f++;
#else
#define PS_INC(f)
#endif

// Every lane of the wavefront calls this together; T, cap-independent control flow and `bounded` are
// wave-uniform.  `on`: the lane holds a real query.  `cap`: upper bound on the lane's 5th distance known in
// advance (FLT_MAX: none) -- subtrees and points beyond it cannot be among the five nearest.
// d[] ascending, p[] positions in the permuted point array (-1: none).
LSLAM_DEV void knn5_packet(const TreeView &T, float qx, float qy, float qz, bool on, float cap, float (&d)[5],
                           int (&p)[5], bool &tie
#ifdef LSLAM_PACKET_STATS
                           , PacketStats &ps
#endif
) {
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    d[i] = FLT_MAX;
    p[i] = -1;
  }
  tie = false;
  if (T.n_pts == 0) return;
  float min_rej = FLT_MAX;  // smallest distance among the candidates turned away or dropped
  uint32_t stack = 0;       // lane i holds stack entry i: (parent slot << 1) | child index
  int sp = 0;
  const int lane_id = (int)__lane_id();
  const unsigned long long on_mask = __ballot(on);
  const int n_on = __popcll(on_mask);
  uint32_t cur = T.root_ref;
  for (;;) {
    bool need_pop = true;
    if (cur & KD_LEAF) {  // nanoflann.hpp:1438-1457, for every lane at once
      PS_INC(leaves)
      const int l = (int)((cur & ~KD_LEAF) >> 4), cnt = (int)(cur & 15u);
      u32x16 a, b;
      u32x8 c;
      sload_leaf(T.pts + l, a, b, c);
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        if (j < cnt) {  // wave-uniform
          float4 pt;
          if (j < 4) pt = make_float4(__uint_as_float(a[4 * j]), __uint_as_float(a[4 * j + 1]), __uint_as_float(a[4 * j + 2]), 0.f);
          else if (j < 8) pt = make_float4(__uint_as_float(b[4 * (j - 4)]), __uint_as_float(b[4 * (j - 4) + 1]), __uint_as_float(b[4 * (j - 4) + 2]), 0.f);
          else pt = make_float4(__uint_as_float(c[4 * (j - 8)]), __uint_as_float(c[4 * (j - 8) + 1]), __uint_as_float(c[4 * (j - 8) + 2]), 0.f);
          const float dist = dist2_xyz(qx, qy, qz, pt);
          const bool pass = on && dist < fminf(d[4], cap);
          const float old4 = d[4];
          if (__any(pass)) {
            PS_INC(inserts)
            if (pass) knn_insert(d, p, dist, l + j);
          }
          min_rej = fminf(min_rej, pass ? old4 : dist);
        }
      }
    } else {  // inner node: which children can still matter to some lane?
      PS_INC(nodes)
      const u32x16 n = sload_pnode(T.pn + (cur >> 2));
      const float w = fminf(d[4], cap);
      const float b1 = box_bound(qx, qy, qz, __uint_as_float(n[2]), __uint_as_float(n[3]), __uint_as_float(n[4]),
                                 __uint_as_float(n[5]), __uint_as_float(n[6]), __uint_as_float(n[7]));
      const float b2 = box_bound(qx, qy, qz, __uint_as_float(n[8]), __uint_as_float(n[9]), __uint_as_float(n[10]),
                                 __uint_as_float(n[11]), __uint_as_float(n[12]), __uint_as_float(n[13]));
      const bool want1 = __any(on && b1 <= w), want2 = __any(on && b2 <= w);
      if (want1 && want2) {
        // the child most lanes are closer to goes first (their 5th distance shrinks before the other is tested)
        const bool first1 = 2 * __popcll(__ballot(on && b1 <= b2)) >= n_on;
        stack = lane_id == sp ? (uint32_t)(((cur >> 2) << 1) | (first1 ? 1u : 0u)) : stack;  // "writelane"
        ++sp;
        cur = first1 ? n[0] : n[1];
        need_pop = false;
      } else if (want1 || want2) {
        cur = want1 ? n[0] : n[1];
        need_pop = false;
      }
    }
    if (!need_pop) continue;
    // take the most recent deferred child some lane still needs
    bool found = false;
    while (sp > 0) {
      PS_INC(pops)
      --sp;
      const uint32_t e = __builtin_amdgcn_readlane(stack, (uint32_t)sp);
      const u32x16 n = sload_pnode(T.pn + (e >> 1));
      const bool second = (e & 1u) != 0;  // wave-uniform
      const float w = fminf(d[4], cap);
      const float bb = second ? box_bound(qx, qy, qz, __uint_as_float(n[8]), __uint_as_float(n[9]), __uint_as_float(n[10]),
                                          __uint_as_float(n[11]), __uint_as_float(n[12]), __uint_as_float(n[13]))
                              : box_bound(qx, qy, qz, __uint_as_float(n[2]), __uint_as_float(n[3]), __uint_as_float(n[4]),
                                          __uint_as_float(n[5]), __uint_as_float(n[6]), __uint_as_float(n[7]));
      if (__any(on && bb <= w)) {
        cur = second ? n[1] : n[0];
        found = true;
        break;
      }
    }
    if (!found) break;
  }
  // ties: equal distances inside the set, or at its boundary (the order is nanoflann's visit order, which
  // this search does not follow) -- the caller redoes such a lane with knn5_search
  tie = on && ((d[0] == d[1]) || (d[1] == d[2]) || (d[2] == d[3]) || (d[3] == d[4]) || (min_rej == d[4]));
}

}  // namespace lslam
