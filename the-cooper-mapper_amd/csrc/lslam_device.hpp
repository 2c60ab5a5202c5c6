// lslam_device.hpp -- device-side arithmetic of the scan-match hot path (gfx950).
//
// Everything here is written so that, compiled with -ffp-contract=off, each
// per-point result is the same IEEE fp32 operation sequence as the reference's
// single-threaded CPU code (citations relative to /root/reference/L_SLAM/src/).
// Small matrices live in registers: every loop is fully unrolled and every array
// index is a compile-time constant after unrolling (pivoting is done with
// predicated swaps), so nothing except the kd-tree traversal stack goes to scratch.
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

namespace lslam {

#define LSLAM_DEV __device__ __forceinline__
#ifndef LSLAM_BRANCHY_LEAF
#define LSLAM_BRANCHY_LEAF 0  // 1: the per-candidate `if (dist < worst) insert` form (A/B switch)
#endif
#ifndef LSLAM_LEAF_COMPACT
#define LSLAM_LEAF_COMPACT 0  // 1: accept test on all ten slots, sorted insert for the accepted ones only (A/B switch, see knn5_search)
#endif
// (a lane loading only the points its leaf holds -- exec-masked loads -- measured SLOWER, 0.475 against 0.448 ms per
// 2.6 M-point launch: the vector memory pipe is paid per wave-instruction, not per active lane)

// ---------------------------------------------------------------------------
// kd-tree in HBM.  Topology and leaf order are exactly nanoflann v1.2.3's
// (util/nanoflann.hpp:931-1078); only the encoding is ours:
//  * only inner nodes are stored; a child is a 32-bit reference
//      inner: (node index << 2) | divfeat of THAT node        (bit 31 clear)
//      leaf : 0x80000000 | (left << 4) | count                (count <= 10)
//    so reaching a leaf costs no extra dependent load;
//  * node = {divlow, divhigh, ref(child1), ref(child2)}, 16 bytes;
//  * nodes are placed in breadth-first groups of 8 (one 128-byte cache line holds a
//    node and about three levels below it), so most steps of a descent hit a line
//    the lane has just touched;
//  * points are stored permuted by vind as float4 {x,y,z,bitcast(original index)}:
//    a leaf is one contiguous <=160 B run.
// ---------------------------------------------------------------------------
struct alignas(16) KdNode {
  float lo, hi;     // divlow, divhigh (nanoflann.hpp:838)
  uint32_t c1, c2;  // child references
};

// Node of the packet search (lslam_packet.hpp): the two child references of the KdNode in the same
// slot and the TIGHT bounding boxes {min x,y,z, max x,y,z} of the points below each child.  64 bytes,
// fetched with one scalar load by a wavefront that walks the tree for 64 neighbouring queries at once.
struct alignas(64) PNode {
  uint32_t c1, c2;
  float box[2][6];
  uint32_t pad[2];
};

constexpr uint32_t KD_LEAF = 0x80000000u;
constexpr uint32_t KD_MAX_POINTS = 1u << 27;  // leaf reference: 27-bit left
constexpr uint32_t KD_MAX_INNER = 1u << 28;   // stack entry: 28-bit node index

struct TreeView {
  const KdNode *nodes;
  const PNode *pn;  // same slots as `nodes` (null: tree built without boxes)
  const float4 *pts;
  float bb_lo[3], bb_hi[3];  // root_bbox, nanoflann.hpp:1406-1427
  int32_t n_pts, n_nodes;
  uint32_t root_ref;
};

constexpr int KD_STACK_MAX = 64;  // host refuses deeper trees (LSLAM_ERR_TREE_DEPTH)

// nanoflann.hpp:364-372 L2_Simple_Adaptor::evalMetric, x->y->z, no contraction.
#ifndef LSLAM_PK_DIST
#define LSLAM_PK_DIST 0  // 1: x / y differences and squares as packed fp32 operations -- measured no faster (A/B switch)
#endif
typedef float lslam_f2 __attribute__((ext_vector_type(2)));
LSLAM_DEV float dist2_xyz(float qx, float qy, float qz, const float4 &p) {
#if LSLAM_PK_DIST
  // the x and y differences and squares as PACKED fp32 operations (v_pk_add_f32 / v_pk_mul_f32: two IEEE fp32 results
  // per instruction, each rounded exactly like the scalar one; a loaded point's x and y sit in an aligned register
  // pair): six instructions instead of eight, the same values -- and the same speed, 6.98e9 against 7.01e9 point-residuals/s
  // on the bench workload: a packed fp32 operation takes the issue slots of the two it replaces
  const lslam_f2 q2 = {qx, qy}, p2 = {p.x, p.y};
  const lslam_f2 d2 = q2 - p2;
  const lslam_f2 s2 = d2 * d2;
  float r = __fadd_rn(s2.x, s2.y);
  const float dz = __fsub_rn(qz, p.z);
  r = __fadd_rn(r, __fmul_rn(dz, dz));
  return r;
#else
  const float dx = __fsub_rn(qx, p.x);
  float r = __fmul_rn(dx, dx);
  const float dy = __fsub_rn(qy, p.y);
  r = __fadd_rn(r, __fmul_rn(dy, dy));
  const float dz = __fsub_rn(qz, p.z);
  r = __fadd_rn(r, __fmul_rn(dz, dz));
  return r;
#endif
}

// nanoflann.hpp:108-135 KNNResultSet::addPoint for capacity 5.  Empty slots hold
// FLT_MAX (init(), :92-98); the new element goes after every element that is not
// strictly greater (strict '>' in :117), the last one drops out.  Written as
// "insertion index k, then per-slot selects on k" so that it stays straight-line
// v_cmp/v_cndmask code.
LSLAM_DEV void knn_insert(float (&d)[5], int (&p)[5], float dist, int pos) {
  const int k = (int)!(d[0] > dist) + (int)!(d[1] > dist) + (int)!(d[2] > dist) +
                (int)!(d[3] > dist) + (int)!(d[4] > dist);
  float nd[5];
  int np[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float dl = j > 0 ? d[j - 1] : 0.0f;
    const int pl = j > 0 ? p[j - 1] : 0;
    nd[j] = (j < k) ? d[j] : ((j == k) ? dist : dl);
    np[j] = (j < k) ? p[j] : ((j == k) ? pos : pl);
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    d[j] = nd[j];
    p[j] = np[j];
  }
}

// The same insertion as straight-line min / med3 / select code (19 operations, no k): for ascending d and the
// new value x, slot i of the result is clamp(x, d[i-1], d[i]) = med3(d[i-1], d[i], x); the index follows with the
// comparisons c_i = (x < d[i]) -- strict, so x goes after every element that is not strictly greater, as
// KNNResultSet::addPoint does (:117).  x = FLT_MAX (an empty slot's value) leaves the set as it is.
LSLAM_DEV void knn_insert_sorted(float (&d)[5], int (&p)[5], float x, int pos) {
  const bool c0 = x < d[0], c1 = x < d[1], c2 = x < d[2], c3 = x < d[3], c4 = x < d[4];
  const float n0 = fminf(d[0], x);
  const float n1 = __builtin_amdgcn_fmed3f(d[0], d[1], x);
  const float n2 = __builtin_amdgcn_fmed3f(d[1], d[2], x);
  const float n3 = __builtin_amdgcn_fmed3f(d[2], d[3], x);
  const float n4 = __builtin_amdgcn_fmed3f(d[3], d[4], x);
  const int q4 = c3 ? p[3] : (c4 ? pos : p[4]);
  const int q3 = c2 ? p[2] : (c3 ? pos : p[3]);
  const int q2 = c1 ? p[1] : (c2 ? pos : p[2]);
  const int q1 = c0 ? p[0] : (c1 ? pos : p[1]);
  const int q0 = c0 ? pos : p[0];
  d[0] = n0; d[1] = n1; d[2] = n2; d[3] = n3; d[4] = n4;
  p[0] = q0; p[1] = q1; p[2] = q2; p[3] = q3; p[4] = q4;
}

// Per-lane traversal stack.  Entries are two 32-bit words.  The first KD_STACK_LDS
// levels live in LDS (layout [level][thread]: lanes at the same depth hit distinct
// banks); deeper levels -- only reached by trees deeper than KD_STACK_LDS+1 -- go
// to a global overflow buffer [levels beyond the LDS ones][2][n_threads] (as many levels as the trees are deep) that the
// host allocates only for such trees.
constexpr int KD_STACK_LDS = 32;

// The LDS part is addressed through an LDS-address-space pointer: with a generic pointer the compiler merges the
// two arms of get() into ONE flat_load behind an address select (seen in the ISA of the batch kernel: two flat loads per
// popped entry instead of one ds_read2st64_b32).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
template <int BLOCK, bool OVF, int LDS_DEPTH = KD_STACK_LDS>
struct KdStack {
  lds_u32 *lds;   // [2][LDS_DEPTH][BLOCK], already offset by the thread index
  uint32_t *ovf;  // global overflow, already offset by the global thread index (OVF only)
  size_t ovf_stride;  // n_threads
  // OVF=false kernels (every tree at most LDS_DEPTH+1 levels deep) touch LDS only, so the
  // accesses stay ds_read/ds_write; the OVF=true variant branches explicitly (a select
  // between the two address spaces would turn every access into a flat one).  The
  // production sweep keeps only LDS_DEPTH = 12 levels in LDS (the bounded search rarely
  // stacks more) so that four workgroups fit a CU instead of two.
  LSLAM_DEV void put(int e, uint32_t w0, uint32_t w1) {
    if (!OVF || e < LDS_DEPTH) {
      lds[e * BLOCK] = w0;
      lds[(LDS_DEPTH + e) * BLOCK] = w1;
    } else {
      ovf[(size_t)(2 * (e - LDS_DEPTH)) * ovf_stride] = w0;
      ovf[(size_t)(2 * (e - LDS_DEPTH) + 1) * ovf_stride] = w1;
    }
  }
  LSLAM_DEV void get(int e, uint32_t &w0, uint32_t &w1) const {
    if (!OVF || e < LDS_DEPTH) {
      w0 = lds[e * BLOCK];
      w1 = lds[(LDS_DEPTH + e) * BLOCK];
    } else {
      w0 = ovf[(size_t)(2 * (e - LDS_DEPTH)) * ovf_stride];
      w1 = ovf[(size_t)(2 * (e - LDS_DEPTH) + 1) * ovf_stride];
    }
  }
};

#ifdef LSLAM_TRAVERSAL_STATS  // profiling build only (tools/traversal_stats.py)
struct TravStats { unsigned long long t_desc, t_leaf, t_pop, t_take; unsigned n_node, n_leaf, n_pop, n_take, n_popit, n_hit, n_cand, n_cull; };
#define TS_BEGIN unsigned long long _ts = __builtin_readcyclecounter();
#define TS_ADD(f) { unsigned long long _n = __builtin_readcyclecounter(); ts.f += _n - _ts; _ts = _n; }
#define TS_INC(f) ts.f++;
#else
#define TS_BEGIN
#define TS_ADD(f)
#define TS_INC(f)
#endif

// Exact 5-NN: nanoflann.hpp:1303-1323 findNeighbors + :1433-1497 searchLevel
// (eps = 0), one query per lane.  The recursion becomes an explicit stack that
// reproduces nanoflann's mindistsq / dists[] values bit for bit:
//   entry = { parent node | far side<<28 | feat<<29 | active<<31 ,  mindistsq of the far child }
// * A far child is only pushed if its mindistsq is <= the current worst distance:
//   the worst distance never grows, so an entry failing the test now would also
//   fail nanoflann's test (:1487) when the recursion returns -- same visits.
// * When an entry is taken, the parent node is re-read and the split arithmetic
//   (:1459-1475) replayed, which yields the same cut_dist / child without storing
//   them; the entry stays on the stack marked active, its second word now holding
//   the old dists[feat], restored when it is popped (:1494).
// * Entries are popped four at a time (8 LDS reads in flight, one wait).
// * A leaf (<= 10 points, one contiguous run) is fetched with 10 independent
//   16-byte loads before any distance is evaluated (the point array is padded).
// p[] are positions in the permuted point array (pts[p].w carries the original index).
//
// `bound` (FLT_MAX for nanoflann's plain search) is an upper bound of the 5th neighbour's
// squared distance known in advance: subtrees and points beyond it cannot be among the five
// nearest, so they are skipped; what is visited is visited in the same order, hence the same
// result (callers pad the bound by a few ulps' worth to cover the rounding of mindistsq).
#ifndef LSLAM_POPW_SHALLOW
#define LSLAM_POPW_SHALLOW 2
#endif
#ifndef LSLAM_CULL_TAKE
#define LSLAM_CULL_TAKE 0  // 1: far subtrees are tested against their tight box before they are entered (A/B switch)
#endif
// TRACK: also returns in *lb6 a LOWER BOUND of the squared distance of every map point that is NOT among the five returned
// (+inf while fewer than six points have been seen): the minimum over every candidate turned away or pushed out of the result
// set (its real distance), and over the mindistsq of every subtree not entered.  With it the next Gauss-Newton iteration can
// prove, for a query that moved by less than the gap between its fifth neighbour and everything else, that the five are still
// the five nearest -- without searching (sweep_body, "certificate").
// A/B switch (round 3): a tracking search prunes a subtree only beyond KAPPA x the current worst squared distance, so that
// lb6 gets closer to the true sixth distance.  Measured: 17.5 % -> 14.8 % (1.1) / 14.6 % (1.25) of the certificate-testing
// points left to the second pass, and slower overall (the wider searches cost more): 1.0.
#ifndef LSLAM_TRACK_KAPPA
#define LSLAM_TRACK_KAPPA 1.0f
#endif
// POPW_SET: stack entries examined per pop round, 0 = by the stack's shape (below).  The grid sweep's second pass asks for one:
// its searches start from a bound and seldom pop more (1.403 -> 1.417e10, round 6).
template <int BLOCK, bool OVF, int LDS_DEPTH, bool TRACK = false, int POPW_SET = 0>
LSLAM_DEV void knn5_search(const TreeView &T, float qx, float qy, float qz, float (&d)[5],
                           int (&p)[5], KdStack<BLOCK, OVF, LDS_DEPTH> &stk,
#ifdef LSLAM_TRAVERSAL_STATS
                           TravStats &ts,
#endif
                           const float bound = FLT_MAX, float *lb6 = nullptr) {
  // the bound a tracking search keeps is collected in the straight-line leaf only: the two A/B leaf variants never fold the
  // candidates they turn away into it, and a bound that is too large would let a certificate "prove" a wrong neighbour set
  static_assert(!TRACK || (!LSLAM_LEAF_COMPACT && !LSLAM_BRANCHY_LEAF), "knn5_search<TRACK> needs the default leaf (certificates would be unsound)");
  float lb = FLT_MAX;
  // stack entries examined per pop round: four in flight where a wavefront's latency is what counts (single scans,
  // whole stack in LDS), fewer where instruction issue is (the shallow-stack batch variant)
  constexpr int POPW = POPW_SET > 0 ? POPW_SET : ((LDS_DEPTH <= 16 && LDS_DEPTH > 0) ? LSLAM_POPW_SHALLOW : 4);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    d[i] = FLT_MAX;
    p[i] = -1;
  }
  if (T.n_pts == 0) return;  // nanoflann.hpp:1306-1307

  // nanoflann.hpp:1080-1097 computeInitialDistances
  float ds0 = 0.0f, ds1 = 0.0f, ds2 = 0.0f, mind = 0.0f;
  if (qx < T.bb_lo[0]) { ds0 = (qx - T.bb_lo[0]) * (qx - T.bb_lo[0]); mind += ds0; }
  if (qx > T.bb_hi[0]) { ds0 = (qx - T.bb_hi[0]) * (qx - T.bb_hi[0]); mind += ds0; }
  if (qy < T.bb_lo[1]) { ds1 = (qy - T.bb_lo[1]) * (qy - T.bb_lo[1]); mind += ds1; }
  if (qy > T.bb_hi[1]) { ds1 = (qy - T.bb_hi[1]) * (qy - T.bb_hi[1]); mind += ds1; }
  if (qz < T.bb_lo[2]) { ds2 = (qz - T.bb_lo[2]) * (qz - T.bb_lo[2]); mind += ds2; }
  if (qz > T.bb_hi[2]) { ds2 = (qz - T.bb_hi[2]) * (qz - T.bb_hi[2]); mind += ds2; }

  int sp = 0;
  uint32_t ref = T.root_ref;
  TS_BEGIN
  for (;;) {
    while (!(ref & KD_LEAF)) {  // inner node: nanoflann.hpp:1459-1475
      TS_INC(n_node)
      const uint32_t node = ref >> 2;
      const uint32_t feat = ref & 3u;
      const KdNode nd = T.nodes[node];
      const float val = feat == 0 ? qx : (feat == 1 ? qy : qz);
      const float diff1 = val - nd.lo;
      const float diff2 = val - nd.hi;
      const bool left = (diff1 + diff2) < 0.0f;
      const float cd = left ? diff2 * diff2 : diff1 * diff1;  // accum_dist :374-377
      const float dst = feat == 0 ? ds0 : (feat == 1 ? ds1 : ds2);
      const float nm = (mind + cd) - dst;  // :1486
      if (nm <= fminf(d[4], bound) * (TRACK ? LSLAM_TRACK_KAPPA : 1.0f)) {
        stk.put(sp, node | (left ? (1u << 28) : 0u) | (feat << 29), __float_as_uint(nm));
        ++sp;
      } else if (TRACK) {
        lb = fminf(lb, nm);  // the far subtree is never entered: nothing in it is closer than nm
      }
      ref = left ? nd.c1 : nd.c2;
#ifdef LSLAM_NODE_TWICE  // profiling only: the node-step arithmetic once more with no effect
      {
        float off = 0.0f;
        asm volatile("" : "+v"(off));
        const float v2 = (feat == 0 ? qx : (feat == 1 ? qy : qz)) + off;
        const float e1 = v2 - nd.lo, e2 = v2 - nd.hi;
        const bool l2 = (e1 + e2) < 0.0f;
        const float c2 = l2 ? e2 * e2 : e1 * e1;
        const float n2 = (mind + c2) - (feat == 0 ? ds0 : (feat == 1 ? ds1 : ds2));
        const uint32_t r2 = l2 ? nd.c1 : nd.c2;
        if (n2 < off - 1.0f) ref = r2 ^ (node | (feat << 29));  // never
      }
#endif
    }
    TS_ADD(t_desc)
    {  // leaf: nanoflann.hpp:1438-1457
      TS_INC(n_leaf)
      const int l = (int)((ref & ~KD_LEAF) >> 4), cnt = (int)(ref & 15u);
      const float worst = fminf(d[4], bound);  // worst_dist cached once per leaf
      // all of the leaf's loads are issued before any distance is evaluated
      float4 pt[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) pt[j] = T.pts[l + j];
#if LSLAM_LEAF_COMPACT
      // A/B switch (round 3): the cheap accept test on all ten slots first, then the sorted insert only for the ACCEPTED
      // candidates -- a lane accepts 5-8 of the ~44 candidates it is offered in a sweep, the straight-line form below pays
      // the 19-operation insert for all 44.  The accepted slots are taken in slot order (lowest set bit first: the order
      // nanoflann's loop offers them in, so ties resolve the same way); a round of the loop costs the insert plus the
      // extraction of the slot's distance from ten registers by a select tree (a lane cannot index registers), and the
      // wavefront runs as many rounds as its busiest lane has accepted candidates in this leaf.
      {
        float dist[10];
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          dist[j] = dist2_xyz(qx, qy, qz, pt[j]);
          acc |= (j < cnt && dist[j] < worst) ? (1u << j) : 0u;
        }
#ifdef LSLAM_TRAVERSAL_STATS
        ts.n_cand += __popc(acc);
        ts.n_hit += acc ? 1 : 0;
#endif
        while (__builtin_amdgcn_ballot_w64(acc != 0u) != 0ull) {
          const int j = acc ? __builtin_ctz(acc) : 0;
          // select tree over the bits of j
          const bool b0 = (j & 1) != 0, b1 = (j & 2) != 0, b2 = (j & 4) != 0, b3 = (j & 8) != 0;
          const float s01 = b0 ? dist[1] : dist[0], s23 = b0 ? dist[3] : dist[2], s45 = b0 ? dist[5] : dist[4],
                      s67 = b0 ? dist[7] : dist[6], s89 = b0 ? dist[9] : dist[8];
          const float s03 = b1 ? s23 : s01, s47 = b1 ? s67 : s45;
          const float s07 = b2 ? s47 : s03;
          const float sx = b3 ? s89 : s07;
          const float x = acc ? sx : FLT_MAX;  // a lane with nothing left offers FLT_MAX: changes nothing
          knn_insert_sorted(d, p, x, l + j);
          acc &= acc - 1u;
        }
      }
#elif LSLAM_BRANCHY_LEAF
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        if (j < cnt) {
          const float dist = dist2_xyz(qx, qy, qz, pt[j]);
          if (dist < worst) knn_insert(d, p, dist, l + j);
        }
      }
#else
      // Straight-line: every lane offers all ten slots to the sorted insert; a slot beyond the leaf's count, or a
      // candidate nanoflann's test `dist < worst_dist` (:1448) turns away, is offered as FLT_MAX and changes nothing.
      // No divergent branch per candidate: the 64 lanes of a wavefront sit in 64 different leaves, so "any lane
      // inserts" was true for almost every candidate and ran the insert with a handful of lanes active.
#ifdef LSLAM_TRAVERSAL_STATS
      bool hit = false;
#endif
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const float dist = dist2_xyz(qx, qy, qz, pt[j]);
        // TRACK: `bound` prunes subtrees only -- a candidate is turned away by the set's own fifth distance (the insert is a
        // no-op for x >= d[4]), so that whoever is NOT in the set after this step, the candidate or the fifth element it
        // pushes out, is max(x, d[4]) away and nobody leaves unrecorded.  Same five whenever the true fifth distance is inside
        // the bound; otherwise d[4] ends at or beyond the bound and the caller's gate turns the point away as before.
        const float x = TRACK ? (j < cnt ? dist : FLT_MAX) : ((j < cnt && dist < worst) ? dist : FLT_MAX);
#ifdef LSLAM_TRAVERSAL_STATS
        hit = hit || x < d[4];
        ts.n_cand += x < d[4] ? 1 : 0;
#endif
        if (TRACK) lb = fminf(lb, fmaxf(x, d[4]));
        knn_insert_sorted(d, p, x, l + j);
      }
#ifdef LSLAM_TRAVERSAL_STATS
      ts.n_hit += hit ? 1 : 0;
#endif
#ifdef LSLAM_INSERT_TWICE  // profiling only: the ten sorted inserts once more with no effect -> their share of the kernel time
      {
        float big = FLT_MAX;
        asm volatile("" : "+v"(big));
#pragma unroll
        for (int j = 0; j < 10; ++j) knn_insert_sorted(d, p, big, l + j);
      }
#endif
#ifdef LSLAM_LEAF_TWICE  // profiling only: the leaf arithmetic once more with no effect -> its share of the kernel time
      {
        float thr = -1.0f, qz2 = qz;
        asm volatile("" : "+v"(thr), "+v"(qz2));
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const float dist = dist2_xyz(qx, qy, qz2, pt[j]);
          const float x = (j < cnt && dist < thr) ? dist : FLT_MAX;
          knn_insert_sorted(d, p, x, l + j);
        }
      }
#endif
#endif
    }
    TS_ADD(t_leaf)
    bool take = false;
    uint32_t te = 0;
    float tm = 0.0f;
    while (sp > 0 && !take) {
      TS_INC(n_popit)
      uint32_t e[POPW];
      float m[POPW];
#pragma unroll
      for (int j = 0; j < POPW; ++j) {
        const int idx = sp - 1 - j < 0 ? 0 : sp - 1 - j;
        uint32_t w1;
        stk.get(idx, e[j], w1);
        m[j] = __uint_as_float(w1);
      }
#pragma unroll
      for (int j = 0; j < POPW; ++j) {  // predicated: no divergent branches in the pop loop
        const bool valid = !take && sp > 0;
        const uint32_t feat = (e[j] >> 29) & 3u;
        const bool act = (e[j] & 0x80000000u) != 0;  // far subtree finished (:1494)
        const bool pass = !act && (m[j] <= fminf(d[4], bound) * (TRACK ? LSLAM_TRACK_KAPPA : 1.0f));  // mindistsq*epsError <= worstDist (:1487)
        const bool rst = valid && act;
        ds0 = (rst && feat == 0) ? m[j] : ds0;        // dists[idx] = dst
        ds1 = (rst && feat == 1) ? m[j] : ds1;
        ds2 = (rst && feat == 2) ? m[j] : ds2;
        const bool tk = valid && pass;
        te = tk ? e[j] : te;
        tm = tk ? m[j] : tm;
        take = take || tk;
        if (TRACK) lb = (valid && !act && !pass) ? fminf(lb, m[j]) : lb;  // a stacked far subtree that is dropped unentered
        sp -= (valid && !pass) ? 1 : 0;
#ifdef LSLAM_TRAVERSAL_STATS
        ts.n_pop += valid ? 1 : 0;
#endif
      }
    }
    TS_ADD(t_pop)
#if LSLAM_CULL_TAKE
    // A/B switch, measured SLOWER (6.70e9 against 7.02e9 point-residuals/s on the bench workload) and off by default:
    // the far subtree's TIGHT box (recorded by the builder next to the node) against the current worst distance before
    // the subtree is entered.  Every point p of the subtree has fl-dist(q,p) >= fl-dist(q,box) -- per axis
    // |fl(q-p)| >= fl(gap) by monotonic rounding, the squares and the x->y->z sums are the same operations on termwise
    // larger values -- so when the box is not closer than worst_dist no point below would pass nanoflann's
    // `dist < worst_dist` (:1448), now or later: the visit would change nothing.  Only 0.5-1.2 of a query's 3 far
    // subtrees are culled, and the box load + 14 operations in every pop round cost more than those visits.
    while (take && T.pn) {
      const uint32_t parent = te & 0x0FFFFFFFu;
      const float2 *bx = reinterpret_cast<const float2 *>(&T.pn[parent].box[(te >> 28) & 1u][0]);
      const float2 b01 = bx[0], b23 = bx[1], b45 = bx[2];  // {min x, min y} {min z, max x} {max y, max z}
      const float gx = fmaxf(fmaxf(b01.x - qx, qx - b23.y), 0.0f);
      const float gy = fmaxf(fmaxf(b01.y - qy, qy - b45.x), 0.0f);
      const float gz = fmaxf(fmaxf(b23.x - qz, qz - b45.y), 0.0f);
      float bd = gx * gx;
      bd = bd + gy * gy;
      bd = bd + gz * gz;
      if (!(bd >= fminf(d[4], bound))) break;
      TS_INC(n_cull)
      take = false;
      --sp;
      while (sp > 0 && !take) {  // the pop round above, one entry at a time
        uint32_t e, w1;
        stk.get(sp - 1, e, w1);
        const float m = __uint_as_float(w1);
        const uint32_t feat = (e >> 29) & 3u;
        const bool act = (e & 0x80000000u) != 0;
        if (act) {
          ds0 = feat == 0 ? m : ds0;
          ds1 = feat == 1 ? m : ds1;
          ds2 = feat == 2 ? m : ds2;
          --sp;
        } else if (m <= fminf(d[4], bound)) {
          te = e; tm = m; take = true;
        } else {
          --sp;
        }
      }
    }
#endif
    if (!take) break;
    {
      TS_INC(n_take)
      const uint32_t parent = te & 0x0FFFFFFFu;
      const uint32_t feat = (te >> 29) & 3u;
      const KdNode pn = T.nodes[parent];
      const float val = feat == 0 ? qx : (feat == 1 ? qy : qz);
      const float diff1 = val - pn.lo;
      const float diff2 = val - pn.hi;
      const bool left = (diff1 + diff2) < 0.0f;
      const float cd = left ? diff2 * diff2 : diff1 * diff1;
      float old;
      if (feat == 0) { old = ds0; ds0 = cd; }
      else if (feat == 1) { old = ds1; ds1 = cd; }
      else { old = ds2; ds2 = cd; }
      stk.put(sp - 1, te | 0x80000000u, __float_as_uint(old));
      mind = tm;
      ref = left ? pn.c2 : pn.c1;  // the child NOT taken on the way down
    }
    TS_ADD(t_take)
  }
  if (TRACK) *lb6 = lb;
}

// The same search for a POOL of queries, the lanes refilled as they finish (persistent lanes: a lane whose stack has run empty
// hands in its five and takes the next query of the pool while the others walk on) -- the second pass of the grid sweep, whose
// queries are a sparse, hard subset: one query per lane left a wavefront as slow as its longest walk at about half its lanes.
// `src` supplies and takes back: bool next(qx, qy, qz, bound) (false: the pool is empty; called by the lanes that need a query,
// in divergent control flow) and emit(d, p) (this lane's finished query).  The arithmetic of a query's walk is knn5_search's,
// statement for statement (default leaf, no tracking): same visits, same five.
template <int BLOCK, bool OVF, int LDS_DEPTH, typename Src>
LSLAM_DEV void knn5_search_refill(const TreeView &T, KdStack<BLOCK, OVF, LDS_DEPTH> &stk, Src &src) {
  constexpr int POPW = (LDS_DEPTH <= 16 && LDS_DEPTH > 0) ? LSLAM_POPW_SHALLOW : 4;
  float qx = 0.0f, qy = 0.0f, qz = 0.0f, bound = FLT_MAX;
  float d[5];
  int p[5];
  float ds0 = 0.0f, ds1 = 0.0f, ds2 = 0.0f, mind = 0.0f;
  int sp = 0;
  uint32_t ref = T.root_ref;
  auto begin = [&]() {  // knn5_search's head for the query in (qx, qy, qz)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      d[i] = FLT_MAX;
      p[i] = -1;
    }
    ds0 = ds1 = ds2 = mind = 0.0f;
    if (qx < T.bb_lo[0]) { ds0 = (qx - T.bb_lo[0]) * (qx - T.bb_lo[0]); mind += ds0; }
    if (qx > T.bb_hi[0]) { ds0 = (qx - T.bb_hi[0]) * (qx - T.bb_hi[0]); mind += ds0; }
    if (qy < T.bb_lo[1]) { ds1 = (qy - T.bb_lo[1]) * (qy - T.bb_lo[1]); mind += ds1; }
    if (qy > T.bb_hi[1]) { ds1 = (qy - T.bb_hi[1]) * (qy - T.bb_hi[1]); mind += ds1; }
    if (qz < T.bb_lo[2]) { ds2 = (qz - T.bb_lo[2]) * (qz - T.bb_lo[2]); mind += ds2; }
    if (qz > T.bb_hi[2]) { ds2 = (qz - T.bb_hi[2]) * (qz - T.bb_hi[2]); mind += ds2; }
    sp = 0;
    ref = T.root_ref;
  };
  bool live = src.next(qx, qy, qz, bound);
  if (live) begin();
  if (T.n_pts == 0) {  // nanoflann.hpp:1306-1307: every query comes back empty
    while (live) {
      src.emit(d, p);
      live = src.next(qx, qy, qz, bound);
      if (live) begin();
    }
    return;
  }
  while (live) {
    while (!(ref & KD_LEAF)) {  // inner node: nanoflann.hpp:1459-1475
      const uint32_t node = ref >> 2;
      const uint32_t feat = ref & 3u;
      const KdNode nd = T.nodes[node];
      const float val = feat == 0 ? qx : (feat == 1 ? qy : qz);
      const float diff1 = val - nd.lo;
      const float diff2 = val - nd.hi;
      const bool left = (diff1 + diff2) < 0.0f;
      const float cd = left ? diff2 * diff2 : diff1 * diff1;
      const float dst = feat == 0 ? ds0 : (feat == 1 ? ds1 : ds2);
      const float nm = (mind + cd) - dst;
      if (nm <= fminf(d[4], bound)) {
        stk.put(sp, node | (left ? (1u << 28) : 0u) | (feat << 29), __float_as_uint(nm));
        ++sp;
      }
      ref = left ? nd.c1 : nd.c2;
    }
    {  // leaf: nanoflann.hpp:1438-1457
      const int l = (int)((ref & ~KD_LEAF) >> 4), cnt = (int)(ref & 15u);
      const float worst = fminf(d[4], bound);
      float4 pt[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) pt[j] = T.pts[l + j];
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const float dist = dist2_xyz(qx, qy, qz, pt[j]);
        const float x = (j < cnt && dist < worst) ? dist : FLT_MAX;
        knn_insert_sorted(d, p, x, l + j);
      }
    }
    bool take = false;
    uint32_t te = 0;
    float tm = 0.0f;
    while (sp > 0 && !take) {
      uint32_t e[POPW];
      float m[POPW];
#pragma unroll
      for (int j = 0; j < POPW; ++j) {
        const int idx = sp - 1 - j < 0 ? 0 : sp - 1 - j;
        uint32_t w1;
        stk.get(idx, e[j], w1);
        m[j] = __uint_as_float(w1);
      }
#pragma unroll
      for (int j = 0; j < POPW; ++j) {
        const bool valid = !take && sp > 0;
        const uint32_t feat = (e[j] >> 29) & 3u;
        const bool act = (e[j] & 0x80000000u) != 0;
        const bool pass = !act && (m[j] <= fminf(d[4], bound));
        const bool rst = valid && act;
        ds0 = (rst && feat == 0) ? m[j] : ds0;
        ds1 = (rst && feat == 1) ? m[j] : ds1;
        ds2 = (rst && feat == 2) ? m[j] : ds2;
        const bool tk = valid && pass;
        te = tk ? e[j] : te;
        tm = tk ? m[j] : tm;
        take = take || tk;
        sp -= (valid && !pass) ? 1 : 0;
      }
    }
    if (!take) {  // this query is done: hand it in, take the next
      src.emit(d, p);
      live = src.next(qx, qy, qz, bound);
      if (live) begin();
      continue;
    }
    {
      const uint32_t parent = te & 0x0FFFFFFFu;
      const uint32_t feat = (te >> 29) & 3u;
      const KdNode pn = T.nodes[parent];
      const float val = feat == 0 ? qx : (feat == 1 ? qy : qz);
      const float diff1 = val - pn.lo;
      const float diff2 = val - pn.hi;
      const bool left = (diff1 + diff2) < 0.0f;
      const float cd = left ? diff2 * diff2 : diff1 * diff1;
      float old;
      if (feat == 0) { old = ds0; ds0 = cd; }
      else if (feat == 1) { old = ds1; ds1 = cd; }
      else { old = ds2; ds2 = cd; }
      stk.put(sp - 1, te | 0x80000000u, __float_as_uint(old));
      mind = tm;
      ref = left ? pn.c2 : pn.c1;
    }
  }
}

#ifdef LSLAM_TRAVERSAL_STATS  // call sites that keep no statistics
template <int BLOCK, bool OVF, int LDS_DEPTH, bool TRACK = false, int POPW_SET = 0>
LSLAM_DEV void knn5_search(const TreeView &T, float qx, float qy, float qz, float (&d)[5], int (&p)[5],
                           KdStack<BLOCK, OVF, LDS_DEPTH> &stk, const float bound = FLT_MAX, float *lb6 = nullptr) {
  TravStats ts = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  knn5_search<BLOCK, OVF, LDS_DEPTH, TRACK, POPW_SET>(T, qx, qy, qz, d, p, stk, ts, bound, lb6);
}
#endif

// ---------------------------------------------------------------------------
// Small dense algebra -- Eigen 3.3 algorithms (Eigen is a dependency of the
// reference that is not under /root/reference).  Row-major
// register arrays, static indexing only.
// ---------------------------------------------------------------------------

template <typename T>
LSLAM_DEV void cswap(bool c, T &a, T &b) {
  const T ta = a, tb = b;
  a = c ? tb : ta;
  b = c ? ta : tb;
}

// Eigen Jacobi.h JacobiRotation::makeGivens, real case
LSLAM_DEV void make_givens(float p, float q, float &c, float &s) {
  if (q == 0.0f) {
    c = p < 0.0f ? -1.0f : 1.0f;
    s = 0.0f;
  } else if (p == 0.0f) {
    c = 0.0f;
    s = q < 0.0f ? 1.0f : -1.0f;
  } else if (fabsf(p) > fabsf(q)) {
    const float t = q / p;
    float u = sqrtf(1.0f + t * t);
    if (p < 0.0f) u = -u;
    c = 1.0f / u;
    s = -t * c;
  } else {
    const float t = p / q;
    float u = sqrtf(1.0f + t * t);
    if (q < 0.0f) u = -u;
    s = -1.0f / u;
    c = -t * s;
  }
}

// Eigen MathFunctionsImpl.h positive_real_hypot
LSLAM_DEV float eigen_hypot(float x, float y) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float p = ax > ay ? ax : ay;
  if (p == 0.0f) return 0.0f;
  const float q = ax > ay ? ay : ax;
  const float qp = q / p;
  return p * sqrtf(1.0f + qp * qp);
}

// SelfAdjointEigenSolver.h tridiagonal_qr_step on the unreduced block [START,END]
// of an N x N problem, eigenvectors accumulated in Q (row-major).
template <int N, int START, int END, bool VEC = true>
LSLAM_DEV void tridiag_qr_step(float (&diag)[N], float (&sub)[N - 1], float (&Q)[N * N]) {
  const float td = (diag[END - 1] - diag[END]) * 0.5f;
  const float e = sub[END - 1];
  float mu = diag[END];
  if (td == 0.0f) {
    mu -= fabsf(e);
  } else if (e != 0.0f) {
    const float e2 = e * e;
    const float h = eigen_hypot(td, e);
    if (e2 == 0.0f)
      mu -= e / ((td + (td > 0.0f ? h : -h)) / e);
    else
      mu -= e2 / (td + (td > 0.0f ? h : -h));
  }
  float x = diag[START] - mu;
  float z = sub[START];
#pragma unroll
  for (int k = START; k < END; ++k) {
    float c, s;
    make_givens(x, z, c, s);
    const float sdk = s * diag[k] + c * sub[k];
    const float dkp1 = s * sub[k] + c * diag[k + 1];
    diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
    diag[k + 1] = s * sdk + c * dkp1;
    sub[k] = c * sdk - s * dkp1;
    if (k > START) sub[k - 1] = c * sub[k - 1] - s * z;
    x = sub[k];
    if (k < END - 1) {
      z = -s * sub[k + 1];
      sub[k + 1] = c * sub[k + 1];
    }
    if (VEC) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float xi = Q[i * N + k], yi = Q[i * N + k + 1];
        Q[i * N + k] = c * xi - s * yi;
        Q[i * N + k + 1] = s * xi + c * yi;
      }
    }
  }
}

// SelfAdjointEigenSolver<Matrix3f>::compute: scale, 3x3 real tridiagonalisation
// (Tridiagonalization.h, tridiagonalization_inplace_selector<M,3,false>),
// computeFromTridiagonal_impl (maxIterations 30), ascending sort.
// A: row-major, only the lower triangle is read.
LSLAM_DEV void eig_sym3(const float (&A)[9], float (&evals)[3], float (&V)[9]) {
  float m00 = A[0], m10 = A[3], m11 = A[4], m20 = A[6], m21 = A[7], m22 = A[8];
  float scale = fabsf(m00);
  scale = fmaxf(scale, fabsf(m10));
  scale = fmaxf(scale, fabsf(m11));
  scale = fmaxf(scale, fabsf(m20));
  scale = fmaxf(scale, fabsf(m21));
  scale = fmaxf(scale, fabsf(m22));
  if (scale == 0.0f) scale = 1.0f;
  m00 /= scale; m10 /= scale; m11 /= scale; m20 /= scale; m21 /= scale; m22 /= scale;

  float diag[3], sub[2];
  diag[0] = m00;
  const float v1norm2 = m20 * m20;
  if (v1norm2 <= FLT_MIN) {
    diag[1] = m11;
    diag[2] = m22;
    sub[0] = m10;
    sub[1] = m21;
    V[0] = 1; V[1] = 0; V[2] = 0; V[3] = 0; V[4] = 1; V[5] = 0; V[6] = 0; V[7] = 0; V[8] = 1;
  } else {
    const float beta = sqrtf(m10 * m10 + v1norm2);
    const float invBeta = 1.0f / beta;
    const float m01 = m10 * invBeta;
    const float m02 = m20 * invBeta;
    const float q = 2.0f * m01 * m21 + m02 * (m22 - m11);
    diag[1] = m11 + m02 * q;
    diag[2] = m22 - m02 * q;
    sub[0] = beta;
    sub[1] = m21 - m01 * q;
    V[0] = 1; V[1] = 0;   V[2] = 0;
    V[3] = 0; V[4] = m01; V[5] = m02;
    V[6] = 0; V[7] = m02; V[8] = -m01;
  }
  int end = 2, start = 0, iter = 0;
  const float precision = 2.0f * FLT_EPSILON;
  while (end > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i >= start && i < end)
        if (fabsf(sub[i]) <= (fabsf(diag[i]) + fabsf(diag[i + 1])) * precision ||
            fabsf(sub[i]) <= FLT_MIN)
          sub[i] = 0.0f;
    // while (end>0 && subdiag[end-1]==0) end--;
    if (end == 2 && sub[1] == 0.0f) end = 1;
    if (end == 1 && sub[0] == 0.0f) end = 0;
    if (end <= 0) break;
    iter++;
    if (iter > 30 * 3) break;
    start = end - 1;
    if (start == 1 && sub[0] != 0.0f) start = 0;
    if (end == 2) {
      if (start == 0) tridiag_qr_step<3, 0, 2>(diag, sub, V);
      else tridiag_qr_step<3, 1, 2>(diag, sub, V);
    } else {
      tridiag_qr_step<3, 0, 1>(diag, sub, V);
    }
  }
  // selection sort ascending (minCoeff picks the first minimum)
  {
    const bool k2 = diag[2] < (diag[1] < diag[0] ? diag[1] : diag[0]);
    const bool k1 = !k2 && diag[1] < diag[0];
    cswap(k1, diag[0], diag[1]);
    cswap(k1, V[0], V[1]); cswap(k1, V[3], V[4]); cswap(k1, V[6], V[7]);
    cswap(k2, diag[0], diag[2]);
    cswap(k2, V[0], V[2]); cswap(k2, V[3], V[5]); cswap(k2, V[6], V[8]);
    const bool j1 = diag[2] < diag[1];
    cswap(j1, diag[1], diag[2]);
    cswap(j1, V[1], V[2]); cswap(j1, V[4], V[5]); cswap(j1, V[7], V[8]);
  }
  evals[0] = diag[0] * scale;
  evals[1] = diag[1] * scale;
  evals[2] = diag[2] * scale;
}

// SelfAdjointEigenSolver<Matrix<float,6,6>>: EIGENVALUES ONLY, registers only.
// Generic Householder tridiagonalisation (Tridiagonalization.h
// tridiagonalization_inplace) + computeFromTridiagonal_impl.  The eigenvalues do not
// depend on whether eigenvectors are accumulated, so this equals the eigenvalue part of
// the full decomposition bit for bit; the degeneracy test (ScanMatch.cpp:223-233) only
// needs the vectors when an eigenvalue is below the threshold (then the slow full
// version runs).
template <int S, int E>
LSLAM_DEV void eig6_step(float (&diag)[6], float (&sub)[5]) {
  float dummy[36];
  tridiag_qr_step<6, S, E, false>(diag, sub, dummy);
}

LSLAM_DEV void eig_sym6_values(const float (&A)[36], float (&evals)[6]) {
  constexpr int N = 6;
  float m[N * N];
  float scale = 0.0f;
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) {
      m[r * N + c] = (c <= r) ? A[r * N + c] : 0.0f;
      scale = fmaxf(scale, fabsf(m[r * N + c]));
    }
  if (scale == 0.0f) scale = 1.0f;
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) m[r * N + c] /= scale;
#pragma unroll
  for (int i = 0; i < N - 1; ++i) {
    const int rem = N - i - 1;
    float h, beta;
    {  // makeHouseholderInPlace on column i, rows i+1..5
      float tailSqNorm = 0.0f;
#pragma unroll
      for (int r = i + 2; r < N; ++r) tailSqNorm += m[r * N + i] * m[r * N + i];
      const float c0 = m[(i + 1) * N + i];
      if (tailSqNorm <= FLT_MIN) {
        h = 0.0f;
        beta = c0;
#pragma unroll
        for (int r = i + 2; r < N; ++r) m[r * N + i] = 0.0f;
      } else {
        beta = sqrtf(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0f) beta = -beta;
        const float denom = c0 - beta;
#pragma unroll
        for (int r = i + 2; r < N; ++r) m[r * N + i] = m[r * N + i] / denom;
        h = (beta - c0) / beta;
      }
    }
    m[(i + 1) * N + i] = 1.0f;
    float v[N], p[N];
#pragma unroll
    for (int a = 0; a < rem; ++a) v[a] = m[(i + 1 + a) * N + i];
#pragma unroll
    for (int a = 0; a < rem; ++a) {
      float acc = 0.0f;
#pragma unroll
      for (int b = 0; b < rem; ++b) {
        const int r = i + 1 + (a > b ? a : b), c = i + 1 + (a > b ? b : a);
        acc += m[r * N + c] * (h * v[b]);
      }
      p[a] = acc;
    }
    float dot = 0.0f;
#pragma unroll
    for (int a = 0; a < rem; ++a) dot += p[a] * v[a];
    const float alpha = h * -0.5f * dot;
#pragma unroll
    for (int a = 0; a < rem; ++a) p[a] += alpha * v[a];
#pragma unroll
    for (int a = 0; a < rem; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b)
        m[(i + 1 + a) * N + (i + 1 + b)] -= (v[a] * p[b] + p[a] * v[b]);
    m[(i + 1) * N + i] = beta;
  }
  float diag[N], sub[N - 1];
#pragma unroll
  for (int i = 0; i < N; ++i) diag[i] = m[i * N + i];
#pragma unroll
  for (int i = 0; i < N - 1; ++i) sub[i] = m[(i + 1) * N + i];
  int end = N - 1, start = 0, iter = 0;
  const float precision = 2.0f * FLT_EPSILON;
  while (end > 0) {
#pragma unroll
    for (int i = 0; i < N - 1; ++i)
      if (i >= start && i < end)
        if (fabsf(sub[i]) <= (fabsf(diag[i]) + fabsf(diag[i + 1])) * precision ||
            fabsf(sub[i]) <= FLT_MIN)
          sub[i] = 0.0f;
#pragma unroll
    for (int e = N - 1; e >= 1; --e)
      if (end == e && sub[e - 1] == 0.0f) end = e - 1;
    if (end <= 0) break;
    iter++;
    if (iter > 30 * N) break;
    start = end - 1;
#pragma unroll
    for (int st = N - 2; st >= 1; --st)
      if (start == st && sub[st - 1] != 0.0f) start = st - 1;
#define LSLAM_E6(S, E) else if (start == S && end == E) eig6_step<S, E>(diag, sub);
    if (false) {}
    LSLAM_E6(0, 1) LSLAM_E6(0, 2) LSLAM_E6(1, 2) LSLAM_E6(0, 3) LSLAM_E6(1, 3) LSLAM_E6(2, 3)
    LSLAM_E6(0, 4) LSLAM_E6(1, 4) LSLAM_E6(2, 4) LSLAM_E6(3, 4)
    LSLAM_E6(0, 5) LSLAM_E6(1, 5) LSLAM_E6(2, 5) LSLAM_E6(3, 5) LSLAM_E6(4, 5)
#undef LSLAM_E6
  }
  // ascending selection sort (values only)
#pragma unroll
  for (int i = 0; i < N - 1; ++i) {
    float mn = diag[i];
    int k = i;
#pragma unroll
    for (int j = i + 1; j < N; ++j) {
      const bool g = diag[j] < mn;
      mn = g ? diag[j] : mn;
      k = g ? j : k;
    }
#pragma unroll
    for (int j = i + 1; j < N; ++j) cswap(k == j, diag[i], diag[j]);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) evals[i] = diag[i] * scale;
}

// ColPivHouseholderQR<Matrix<float,R,C>>::compute + solve (Eigen 3.3,
// ColPivHouseholderQR.h computeInPlace/_solve_impl; Householder.h
// makeHouseholder/applyHouseholderOnTheLeft).  A row-major (consumed).
template <int R, int C>
LSLAM_DEV void colpiv_qr_solve(float (&qr)[R * C], const float (&b)[R], float (&x)[C]) {
  constexpr int SIZE = R < C ? R : C;
  float hC[SIZE], nU[C], nD[C], c[R];
  int trans[SIZE], perm[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < R; ++i) s += qr[i * C + k] * qr[i * C + k];
    nD[k] = sqrtf(s);
    nU[k] = nD[k];
  }
  float maxn = nU[0];
#pragma unroll
  for (int k = 1; k < C; ++k) maxn = nU[k] > maxn ? nU[k] : maxn;
  const float th = maxn * FLT_EPSILON;
  const float threshold_helper = (th * th) / (float)R;
  const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
  int nonzero_pivots = SIZE;
#pragma unroll
  for (int k = 0; k < SIZE; ++k) {
    int big = k;
    float bigv = nU[k];
#pragma unroll
    for (int j = k + 1; j < C; ++j) {
      const bool g = nU[j] > bigv;
      bigv = g ? nU[j] : bigv;
      big = g ? j : big;
    }
    const float biggest_col_sq_norm = bigv * bigv;
    if (nonzero_pivots == SIZE && biggest_col_sq_norm < threshold_helper * (float)(R - k))
      nonzero_pivots = k;
    trans[k] = big;
#pragma unroll
    for (int j = k + 1; j < C; ++j) {
      const bool sw = (big == j);
#pragma unroll
      for (int i = 0; i < R; ++i) cswap(sw, qr[i * C + k], qr[i * C + j]);
      cswap(sw, nU[k], nU[j]);
      cswap(sw, nD[k], nD[j]);
    }
    // makeHouseholderInPlace on column k, rows k..R-1
    float tau, beta;
    {
      float tailSqNorm = 0.0f;
#pragma unroll
      for (int i = k + 1; i < R; ++i) tailSqNorm += qr[i * C + k] * qr[i * C + k];
      const float c0 = qr[k * C + k];
      if (tailSqNorm <= FLT_MIN) {
        tau = 0.0f;
        beta = c0;
#pragma unroll
        for (int i = k + 1; i < R; ++i) qr[i * C + k] = 0.0f;
      } else {
        beta = sqrtf(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0f) beta = -beta;
        const float denom = c0 - beta;
#pragma unroll
        for (int i = k + 1; i < R; ++i) qr[i * C + k] = qr[i * C + k] / denom;
        tau = (beta - c0) / beta;
      }
    }
    hC[k] = tau;
    qr[k * C + k] = beta;
    // applyHouseholderOnTheLeft to the block rows k..R-1, cols k+1..C-1
    if (R - k == 1) {
#pragma unroll
      for (int j = k + 1; j < C; ++j) qr[k * C + j] *= (1.0f - tau);
    } else if (tau != 0.0f) {
#pragma unroll
      for (int j = k + 1; j < C; ++j) {
        float tmp = 0.0f;
#pragma unroll
        for (int i = k + 1; i < R; ++i) tmp += qr[i * C + k] * qr[i * C + j];
        tmp += qr[k * C + j];
        qr[k * C + j] -= tau * tmp;
#pragma unroll
        for (int i = k + 1; i < R; ++i) qr[i * C + j] -= tau * qr[i * C + k] * tmp;
      }
    }
    // column norm downdate
#pragma unroll
    for (int j = k + 1; j < C; ++j) {
      if (nU[j] != 0.0f) {
        float temp = fabsf(qr[k * C + j]) / nU[j];
        temp = (1.0f + temp) * (1.0f - temp);
        temp = temp < 0.0f ? 0.0f : temp;
        const float r = nU[j] / nD[j];
        const float temp2 = temp * (r * r);
        if (temp2 <= norm_downdate_threshold) {
          float s = 0.0f;
#pragma unroll
          for (int i = k + 1; i < R; ++i) s += qr[i * C + j] * qr[i * C + j];
          nD[j] = sqrtf(s);
          nU[j] = nD[j];
        } else {
          nU[j] *= sqrtf(temp);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < C; ++k) perm[k] = k;
#pragma unroll
  for (int k = 0; k < SIZE; ++k) {
#pragma unroll
    for (int j = k + 1; j < C; ++j) cswap(trans[k] == j, perm[k], perm[j]);
  }
  if (nonzero_pivots == 0) {
#pragma unroll
    for (int i = 0; i < C; ++i) x[i] = 0.0f;
    return;
  }
#pragma unroll
  for (int i = 0; i < R; ++i) c[i] = b[i];
  // c = Q^T c : H_0 first
#pragma unroll
  for (int k = 0; k < SIZE; ++k) {
    if (k < nonzero_pivots) {
      const float tau = hC[k];
      if (R - k == 1) {
        c[k] *= (1.0f - tau);
      } else if (tau != 0.0f) {
        float tmp = 0.0f;
#pragma unroll
        for (int i = k + 1; i < R; ++i) tmp += qr[i * C + k] * c[i];
        tmp += c[k];
        c[k] -= tau * tmp;
#pragma unroll
        for (int i = k + 1; i < R; ++i) c[i] -= tau * qr[i * C + k] * tmp;
      }
    }
  }
  // upper-triangular solve on the leading nonzero_pivots block (column oriented)
#pragma unroll
  for (int i = SIZE - 1; i >= 0; --i) {
    if (i < nonzero_pivots) {
      c[i] /= qr[i * C + i];
#pragma unroll
      for (int r = 0; r < i; ++r) c[r] -= c[i] * qr[r * C + i];
    }
  }
#pragma unroll
  for (int j = 0; j < C; ++j) x[j] = 0.0f;
#pragma unroll
  for (int i = 0; i < SIZE; ++i) {
    if (i < nonzero_pivots) {
#pragma unroll
      for (int j = 0; j < C; ++j)
        if (perm[i] == j) x[j] = c[i];
    }
  }
}

// ---------------------------------------------------------------------------
// Geometry: util/feature_utils.h
// ---------------------------------------------------------------------------

LSLAM_DEV void cross3(const float (&a)[3], const float (&b)[3], float (&o)[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
LSLAM_DEV float norm3(const float (&a)[3]) {
  return sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
}

// feature_utils.h:108-154 findLine on the 5 gathered neighbours
LSLAM_DEV bool find_line(const float4 (&nb)[5], float (&A)[3], float (&B)[3]) {
  float c[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    c[0] += nb[j].x;
    c[1] += nb[j].y;
    c[2] += nb[j].z;
  }
  c[0] /= 5.0f; c[1] /= 5.0f; c[2] /= 5.0f;
  float M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float a0 = nb[j].x - c[0], a1 = nb[j].y - c[1], a2 = nb[j].z - c[2];
    M[0] += a0 * a0;
    M[3] += a0 * a1;
    M[6] += a0 * a2;
    M[4] += a1 * a1;
    M[7] += a1 * a2;
    M[8] += a2 * a2;
  }
  M[0] /= 5.0f; M[3] /= 5.0f; M[6] /= 5.0f; M[4] /= 5.0f; M[7] /= 5.0f; M[8] /= 5.0f;
  float D[3], V[9];
  eig_sym3(M, D, V);
  if (D[2] > 5 * D[1]) {
    const float v[3] = {V[2], V[5], V[8]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      A[i] = c[i] - v[i] * 0.1f;
      B[i] = c[i] + v[i] * 0.1f;
    }
    return true;
  }
  return false;
}

// feature_utils.h:17-26 getLinePointDistance + :63-75 getCornerFeatureCoefficients
LSLAM_DEV bool corner_coeff(const float (&A)[3], const float (&B)[3], const float (&X)[3],
                            float (&coeff)[4]) {
  const float XB[3] = {X[0] - B[0], X[1] - B[1], X[2] - B[2]};
  const float XA[3] = {X[0] - A[0], X[1] - A[1], X[2] - A[2]};
  float n[3];
  cross3(XB, XA, n);
  const float nn = norm3(n);
  const float AB[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]};
  const float lengthAB = norm3(AB);
  const float BA[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
  const float mn[3] = {-n[0], -n[1], -n[2]};
  float cr[3];
  cross3(mn, BA, cr);
  const float den = nn * lengthAB;
  const float distance = nn / lengthAB;
  // `1 - 0.9f * fabs(distance)` with ::fabs(double) -- the overload the reference's template binds (oracle/lslam_oracle.c,
  // corner_coefficients): evaluated in double, rounded once
  const float weight = (float)(1.0 - (double)0.9f * fabs((double)distance));
  coeff[0] = (cr[0] / den) * weight;
  coeff[1] = (cr[1] / den) * weight;
  coeff[2] = (cr[2] / den) * weight;
  coeff[3] = distance * weight;
  return (double)weight > 0.1;  // float vs double literal, as in the reference
}

// feature_utils.h:157-204 findPlane
LSLAM_DEV bool find_plane(const float4 (&nb)[5], float max_distance, float (&plane)[4]) {
  float Am[15];
  const float bm[5] = {-1.0f, -1.0f, -1.0f, -1.0f, -1.0f};
  float x[3];
  float c[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    c[0] += nb[j].x;
    c[1] += nb[j].y;
    c[2] += nb[j].z;
    Am[j * 3 + 0] = nb[j].x;
    Am[j * 3 + 1] = nb[j].y;
    Am[j * 3 + 2] = nb[j].z;
  }
  c[0] /= 5.0f; c[1] /= 5.0f; c[2] /= 5.0f;
  colpiv_qr_solve<5, 3>(Am, bm, x);
  const float norm = sqrtf(((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]) + 0.0f * 0.0f);
  plane[0] = x[0] / norm;
  plane[1] = x[1] / norm;
  plane[2] = x[2] / norm;
  plane[3] = -((plane[0] * c[0] + plane[1] * c[1]) + plane[2] * c[2]);
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float distance = ((plane[0] * nb[j].x + plane[1] * nb[j].y) + plane[2] * nb[j].z) + plane[3];
    ok = ok && !(fabsf(distance) > max_distance);
  }
  return ok;
}

// feature_utils.h:97-106 getSurfaceFeatureCoefficients (double literal 0.9)
LSLAM_DEV bool surf_coeff(const float (&plane)[4], const float (&X)[3], float (&coeff)[4]) {
  const float distance = ((plane[0] * X[0] + plane[1] * X[1]) + plane[2] * X[2]) + plane[3];
  const float xn = norm3(X);
  const float weight = (float)(1 - 0.9 * (double)fabsf(distance) / sqrt((double)xn));
  coeff[0] = plane[0] * weight;
  coeff[1] = plane[1] * weight;
  coeff[2] = plane[2] * weight;
  coeff[3] = distance * weight;
  return (double)weight > 0.1;  // float vs double literal, as in the reference
}

// scan_to_scan_match/ScanMatch.cpp:185-203 with the original operator precedence
// (quirk Q1 in arz).  sc = {srx,crx,sry,cry,srz,crz}
LSLAM_DEV void jacobian_row(const float (&sc)[6], float px, float py, float pz,
                            const float (&coeff)[4], float (&row)[6], float &b) {
  const float srx = sc[0], crx = sc[1], sry = sc[2], cry = sc[3], srz = sc[4], crz = sc[5];
  const float cx = coeff[0], cy = coeff[1], cz = coeff[2];
  const float arx =
      ((crz * sry * crx + srz * srx) * py + (srz * crx - crz * sry * srx) * pz) * cx +
      ((srz * sry * crx - crz * srx) * py - (srz * sry * srx + crz * crx) * pz) * cy +
      (cry * crx * py - cry * srx * pz) * cz;
  const float ary = (-crz * sry * px + crz * cry * srx * py + crz * cry * crx * pz) * cx +
                    (-srz * sry * px + srz * cry * srx * py + srz * cry * crx * pz) * cy +
                    (-cry * px - sry * srx * py - sry * crx * pz) * cz;
  const float arz =
      (-srz * cry * px - (srz * sry * srx + crz * crx) * py + (crz * srx - srz * sry * crx) * pz) * cx +
      (crz * cry * px + (crz * sry * srx - srz * crx) * py + crz * sry * crx + srz * srx * pz) * cy +
      0 * cz;
  row[0] = arx; row[1] = ary; row[2] = arz;
  row[3] = cx; row[4] = cy; row[5] = cz;
  b = -coeff[3];
}

// util/transform_utils.h:288-299 getTransformationTZYX (+ :308-311):
// q = AngleAxis(rz,Z)*AngleAxis(ry,Y)*AngleAxis(rx,X); R = q.toRotationMatrix().
// hs/hc: sin/cos of the half angles (Eigen AngleAxis -> Quaternion), fs/fc: sin/cos of
// the full angles (util/Angle.h:17-18 cached values).  The host evaluates them with
// std::sin/std::cos(float) -- exactly the reference's libm calls -- the device solve
// kernel evaluates the six angles in six lanes.
struct Quat { float w, x, y, z; };
__host__ __device__ inline void sincos_to_Rt_sc(const float pose[6], const float hs[3],
                                                const float hc[3], const float fs[3],
                                                const float fc[3], float R[9], float t[3],
                                                float sc[6]) {
  const Quat qx = {hc[0], hs[0], 0.0f, 0.0f};
  const Quat qy = {hc[1], 0.0f, hs[1], 0.0f};
  const Quat qz = {hc[2], 0.0f, 0.0f, hs[2]};
  auto mul = [](const Quat &a, const Quat &b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
  };
  const Quat q = mul(mul(qz, qy), qx);
  const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
  t[0] = pose[3]; t[1] = pose[4]; t[2] = pose[5];
  sc[0] = fs[0]; sc[1] = fc[0];
  sc[2] = fs[1]; sc[3] = fc[1];
  sc[4] = fs[2]; sc[5] = fc[2];
}

template <typename F>
__host__ __device__ inline void pose_to_Rt_sc(const float pose[6], float R[9], float t[3],
                                              float sc[6], F sincos_f) {
  float hs[3], hc[3], fs[3], fc[3];
  for (int i = 0; i < 3; ++i) {
    sincos_f(0.5f * pose[i], hs[i], hc[i]);
    sincos_f(pose[i], fs[i], fc[i]);
  }
  sincos_to_Rt_sc(pose, hs, hc, fs, fc, R, t, sc);
}

// ---------------------------------------------------------------------------
// Wave-parallel ColPivHouseholderQR<Matrix<float,6,6>>::solve: lane j (0..5) owns
// column j of A^T A, lane 6 owns the right-hand side.  Every element goes through the
// same fp32 operations in the same order as the sequential algorithm above (column
// norms, Householder reflectors and their application are column-local), so the
// result is bit-identical to colpiv_qr_solve<6,6>; only the critical path shrinks.
// col[] is this lane's column (consumed).  Returns x[6] in every lane.
// ---------------------------------------------------------------------------
// (Values cross lanes by v_readlane -- the source lane is uniform everywhere, a constant or the pivot index -- not by
// ds_bpermute shuffles: ~130 dependent LDS round trips were most of the 5 us this solve took.)
LSLAM_DEV float qr_rl(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }
LSLAM_DEV void colpiv_qr_solve6_wave(float (&col)[6], int lane, float (&x)[6]) {
  const bool is_mat = lane < 6;
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) s += col[i] * col[i];
  float nD = sqrtf(s), nU = nD;
  float u[6];
#pragma unroll
  for (int l = 0; l < 6; ++l) u[l] = qr_rl(nU, l);
  float maxn = u[0];
#pragma unroll
  for (int l = 1; l < 6; ++l) maxn = u[l] > maxn ? u[l] : maxn;
  const float th = maxn * FLT_EPSILON;
  const float threshold_helper = (th * th) / 6.0f;
  const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
  int nonzero_pivots = 6;
  int perm[6];
#pragma unroll
  for (int l = 0; l < 6; ++l) perm[l] = l;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int l = 0; l < 6; ++l) u[l] = qr_rl(nU, l);
    int big = k;
    float bigv = u[k];
#pragma unroll
    for (int j = k + 1; j < 6; ++j) {
      const bool g = u[j] > bigv;
      bigv = g ? u[j] : bigv;
      big = g ? j : big;
    }
    if (nonzero_pivots == 6 && bigv * bigv < threshold_helper * (float)(6 - k)) nonzero_pivots = k;
    // column transposition k <-> big (and the permutation bookkeeping)
    const int bigu = __builtin_amdgcn_readfirstlane(big);  // the same in every lane: it comes from the broadcast norms
    {
      const bool at_k = lane == k, at_big = lane == bigu;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float vk = qr_rl(col[i], k), vb = qr_rl(col[i], bigu);
        col[i] = at_k ? vb : (at_big ? vk : col[i]);
      }
      const float uk = qr_rl(nU, k), ub = qr_rl(nU, bigu), dk = qr_rl(nD, k), db = qr_rl(nD, bigu);
      nU = at_k ? ub : (at_big ? uk : nU);
      nD = at_k ? db : (at_big ? dk : nD);
    }
#pragma unroll
    for (int j = k + 1; j < 6; ++j) cswap(big == j, perm[k], perm[j]);
    // Householder vector of column k (every lane computes its own, lane k's is used)
    float tau, beta, ess[6];
    {
      float tailSqNorm = 0.0f;
#pragma unroll
      for (int i = k + 1; i < 6; ++i) tailSqNorm += col[i] * col[i];
      const float c0 = col[k];
      if (tailSqNorm <= FLT_MIN) {
        tau = 0.0f;
        beta = c0;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) ess[i] = 0.0f;
      } else {
        beta = sqrtf(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0f) beta = -beta;
        const float denom = c0 - beta;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) ess[i] = col[i] / denom;
        tau = (beta - c0) / beta;
      }
    }
    tau = qr_rl(tau, k);
    beta = qr_rl(beta, k);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) ess[i] = qr_rl(ess[i], k);
    if (lane == k) {
      col[k] = beta;
#pragma unroll
      for (int i = k + 1; i < 6; ++i) col[i] = ess[i];
    }
    // apply H_k to the remaining columns and (while pivots are non-zero) to the rhs
    const bool apply = (is_mat && lane > k) || (lane == 6 && k < nonzero_pivots);
    if (apply) {
      if (k == 5) {
        col[5] *= (1.0f - tau);
      } else if (tau != 0.0f) {
        float tmp = 0.0f;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) tmp += ess[i] * col[i];
        tmp += col[k];
        col[k] -= tau * tmp;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) col[i] -= tau * ess[i] * tmp;
      }
    }
    // column-norm downdate for the remaining matrix columns
    if (is_mat && lane > k) {
      if (nU != 0.0f) {
        float temp = fabsf(col[k]) / nU;
        temp = (1.0f + temp) * (1.0f - temp);
        temp = temp < 0.0f ? 0.0f : temp;
        const float r = nU / nD;
        const float temp2 = temp * (r * r);
        if (temp2 <= norm_downdate_threshold) {
          float ss = 0.0f;
#pragma unroll
          for (int i = k + 1; i < 6; ++i) ss += col[i] * col[i];
          nD = sqrtf(ss);
          nU = nD;
        } else {
          nU *= sqrtf(temp);
        }
      }
    }
  }
  // gather R (upper triangle) and c = Q^T b into every lane, back-substitute
  float Rm[6][6], c[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    c[i] = qr_rl(col[i], 6);
#pragma unroll
    for (int r = 0; r <= i; ++r) Rm[r][i] = qr_rl(col[r], i);
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    if (i < nonzero_pivots) {
      c[i] /= Rm[i][i];
#pragma unroll
      for (int r = 0; r < i; ++r) c[r] -= c[i] * Rm[r][i];
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) x[j] = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (i < nonzero_pivots) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (perm[i] == j) x[j] = c[i];
    }
  }
}

}  // namespace lslam
