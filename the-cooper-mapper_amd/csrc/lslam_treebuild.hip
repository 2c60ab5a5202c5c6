// lslam_treebuild.hip -- kd-tree construction on the GPU (gfx950).
//
// Builds, for a cloud already in HBM, the tree nanoflann v1.2.3 builds
// (util/nanoflann.hpp:931-1078 divideTree / middleSplit_ / planeSplit, leaf_max_size 10)
// -- same split dimension, split value and balance rule at every node AND the same
// in-place permutation of the points, so leaves hold the same points in the same order
// as the reference's `vind` -- directly in the device encoding of lslam_device.hpp.
//
// How the sequential algorithm is reproduced in parallel
//   * A node is processed by one workgroup: min/max of the three coordinates, the split
//     decision (middleSplit_, :982-1031), the counts lim1/lim2, the partition, the tight
//     bounds divlow/divhigh of the two halves (:966-971).
//   * planeSplit (:1043-1078) is two Hoare passes.  A Hoare pass with pivot predicate P
//     leaves the elements that already are on their side in place and swaps the k-th
//     misplaced element counted from the left with the k-th misplaced element counted
//     from the right.  That pairing is computed with two block-wide prefix scans and
//     applied as independent swaps, which yields exactly nanoflann's arrangement.
//   * Workgroups are persistent: big nodes (> LOCAL_MAX points) travel through a global
//     queue (agent-scope release/acquire around every hand-off, per the CDNA4 guide);
//     a node with <= LOCAL_MAX points is finished, with its whole subtree, by the
//     workgroup that produced it (explicit stack in LDS).
//   * Nodes are placed in groups of 8 slots (one 128-byte line) holding a node, its
//     children and grandchildren (heap order); great-grandchildren open new groups.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "lslam_internal.hpp"

namespace lslam {

namespace {

// Two phases.  Phase A: workgroups of TB_BIG threads take nodes with more than LOCAL_MAX
// points from a global queue (hand-off with agent-scope release/acquire); a child with at
// most LOCAL_MAX points goes to the subtree list instead.  Phase B: one wavefront per
// subtree of that list builds it completely (explicit stack in LDS, no barriers needed).
#ifndef LSLAM_LOCAL_MAX
#define LSLAM_LOCAL_MAX 1536
#endif
constexpr int TB_BIG = 1024;
constexpr int TB_SMALL = 64;
constexpr int TE = 8;              // consecutive elements per thread in the partition scans
constexpr int LOCAL_MAX = LSLAM_LOCAL_MAX;    // subtrees up to this many points are built by one wavefront
constexpr int LOCAL_STACK = KD_STACK_MAX;  // pending siblings along one root-to-leaf path: a deeper tree is refused anyway
// Round 3: the level phase hands a node on as soon as it has at most BuildArgs::huge_min points; what is still above
// LOCAL_MAX then is finished by ONE workgroup per such node (kd_build_medium_kernel).  From its ninth level on the chain of
// seven launches per level of a big tree was kept alive by a handful of stragglers.  Measured on the bench's surround
// (157 k + 587 k points; same box): 1 536 (the chain to the end) 1.216 ms, 8 192 1.125, 16 384 1.265, 32 768 1.625 -- a
// workgroup works its node's descendants off one after the other, which the levels do side by side; for the cube forest
// (22 trees, 319 k points) the chain to the end stays best (0.876 / 0.908 / 0.980 / 1.409 ms) and is what it takes.
#ifndef LSLAM_HUGE_MIN
#define LSLAM_HUGE_MIN 8192
#endif
constexpr int HUGE_MIN_TREE = LSLAM_HUGE_MIN;  // single trees (LSLAM_HUGE_MIN in the environment overrides; LOCAL_MAX: no medium phase)
constexpr int HUGE_MIN_MAX = 32768;            // sizes the medium phase's stack

struct BuildItem {   // one pending inner node
  int32_t l, r;      // point range
  float lo[3], hi[3];  // propagated bounding box (nanoflann.hpp:955-961)
  int32_t slot;      // node slot of this node
  int32_t heap;      // position inside its 8-slot group (0; 1,2; 3..6)
  int32_t parent_word;  // index (in 32-bit words of the node array) of the reference to this
                        // node in its parent, -1 for the root: the node ORs its divfeat there
  int32_t depth;
  int32_t g0, g1;       // phase C entries: node groups [g0, g1) reserved for the subtree by the wavefront that handed it on
};

struct BuildCtl {
  int32_t q_head, q_tail_reserved, q_pending;  // queue indices / nodes not yet finished
  int32_t next_group;   // next free 8-slot group
  int32_t max_depth;
  int32_t overflow;     // node array or queue too small
  int32_t root_feat;
  int32_t n_leaves;
  int32_t n_sub;       // entries of the subtree list
  int32_t sub_next;    // next entry to take in phase B
};

struct BuildArgs {
  float4 *pts;        // permuted in place; .w = original index
  KdNode *nodes;
  int32_t node_cap;   // in nodes (multiple of 8)
  BuildItem *queue;
  int32_t *q_ready;   // per queue entry
  int32_t queue_cap;
  int32_t huge_min;   // nodes above this many points go level by level; what the levels leave above LOCAL_MAX, through the medium phase
  BuildItem *sublist;  // phase B work list (subtree roots with <= LOCAL_MAX points)
  int32_t sub_cap;
  int32_t *tmpA, *tmpB;  // scratch, one int per point
  BuildCtl *ctl;
  int32_t *root_feat;  // split dimension of root t (a root's parent_word is -1 - t)
  int32_t n;
  int32_t reg_nodes;   // 1: nodes of at most 64 points are finished in registers (phase B)
  uint32_t spin_limit; // watchdog of idle phase-A workgroups (polls of ~2 x 127 sleep units)
  int32_t *tiny_acc;    // [TINY_ACC][32]: {leaves, max depth} of phase C, spread over cache lines
  BuildItem *tiny_list; // (or null) phase C work list: nodes of at most 64 points that phase B hands on instead of finishing
  int32_t tiny_cap;
  float *own_box;      // [node_cap][6] (or null): tight bounding box {min xyz, max xyz} of every inner node's points --
                       // the extrema middleSplit_ needs anyway; the packet search's node boxes are made from them
};

__device__ __forceinline__ void store_own_box(const BuildArgs &A, int slot, float n0, float n1, float n2, float x0, float x1,
                                              float x2) {
  if (A.own_box) {
    float *o = A.own_box + (size_t)slot * 6;
    o[0] = n0; o[1] = n1; o[2] = n2; o[3] = x0; o[4] = x1; o[5] = x2;
  }
}

__device__ __forceinline__ float coord(const float4 &p, int d) { return d == 0 ? p.x : (d == 1 ? p.y : p.z); }

template <int TB>
struct Sh {
  float fmin[3][TB / 64], fmax[3][TB / 64];
  int isum[2][TB / 64];
  int scan[TB / 64];
  float bc_f[8];
  int bc_i[8];
  BuildItem item;
  BuildItem stack[LOCAL_STACK];
  int sp;
  int have;
};

// Wavefront-wide reductions on the DPP path (no LDS crossbar): xor-1 and xor-2 inside each quad,
// row_half_mirror and row_mirror complete the 16-lane rows, four readlanes combine the rows.  Every
// lane receives the result.  (__shfl_xor lowers to ds_bpermute, ~10x the latency; a small kd-tree
// node needs about a dozen of these reductions.)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <typename Op>
__device__ __forceinline__ int wave_reduce_bits(int v, Op op) {
  v = op(v, dpp_i<0xB1>(v));   // quad_perm [1,0,3,2]
  v = op(v, dpp_i<0x4E>(v));   // quad_perm [2,3,0,1]
  v = op(v, dpp_i<0x141>(v));  // row_half_mirror
  v = op(v, dpp_i<0x140>(v));  // row_mirror
  const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16),
            r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_min(float v) {
  return __int_as_float(wave_reduce_bits(__float_as_int(v), [](int a, int b) {
    return __float_as_int(fminf(__int_as_float(a), __int_as_float(b)));
  }));
}
__device__ __forceinline__ float wave_max(float v) {
  return __int_as_float(wave_reduce_bits(__float_as_int(v), [](int a, int b) {
    return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b)));
  }));
}
__device__ __forceinline__ int wave_sum_i(int v) {
  return wave_reduce_bits(v, [](int a, int b) { return a + b; });
}

// Inclusive segmented scan (min or max) over lanes [sa, lane] of the lane's segment -- segments are
// disjoint lane ranges, sa <= lane -- on the DPP path: row_shr 1/2/4/8 inside the 16-lane rows (a lane
// shifted in from outside the row keeps its own value), then rows 1 and 3 take the total of the row
// below them (row_bcast15) and rows 2 and 3 that of the lower half (row_bcast31), each step only where
// the source lane belongs to the same segment.  Six ds_bpermute round trips (__shfl_up) become six
// register moves.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <bool MAX>
__device__ __forceinline__ float seg_scan(float v, int lane, int sa) {
  float t;
  t = dpp_f<0x111, 0xf>(v); if (lane - 1 >= sa) v = MAX ? fmaxf(v, t) : fminf(v, t);
  t = dpp_f<0x112, 0xf>(v); if (lane - 2 >= sa) v = MAX ? fmaxf(v, t) : fminf(v, t);
  t = dpp_f<0x114, 0xf>(v); if (lane - 4 >= sa) v = MAX ? fmaxf(v, t) : fminf(v, t);
  t = dpp_f<0x118, 0xf>(v); if (lane - 8 >= sa) v = MAX ? fmaxf(v, t) : fminf(v, t);
  t = dpp_f<0x142, 0xa>(v); if ((lane & 16) && sa < (lane & ~15)) v = MAX ? fmaxf(v, t) : fminf(v, t);
  t = dpp_f<0x143, 0xc>(v); if (lane >= 32 && sa < 32) v = MAX ? fmaxf(v, t) : fminf(v, t);
  return v;
}

// exclusive prefix sum of `flag` over the block's TB threads; returns prefix, *total
template <int TB>
__device__ __forceinline__ int block_excl_scan(int flag, Sh<TB> &sh, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int v = flag;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  if (lane == 63) sh.scan[wave] = v;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < TB / 64; ++w) {
    if (w < wave) base += sh.scan[w];
    tot += sh.scan[w];
  }
  __syncthreads();
  *total = tot;
  return base + v - flag;
}

// One Hoare pass over pts[l+a .. l+b) (nanoflann.hpp:1043-1060 or :1062-1076).
// `lt` selects the predicate: true: "x < cut" belongs left; false: "x <= cut" belongs left.
// L = number of elements that belong left.  Elements [a, a+L) that do not belong left are
// swapped, k-th from the left, with the k-th element from the right of [a+L, b) that does.
template <int TB>
__device__ void hoare_pass(const BuildArgs &A, Sh<TB> &sh, int l, int a, int b, int L, int feat, float cut,
                           bool lt) {
  const int tid = threadIdx.x;
  const int nl = L, nr = (b - a) - L;
  if (nl == 0 || nr == 0) return;
  int run = 0;
  for (int c = 0; c < nl; c += TB * TE) {  // misplaced on the left, increasing index
    const int i0 = c + tid * TE;           // each thread owns TE consecutive elements
    unsigned mask = 0;
#pragma unroll
    for (int u = 0; u < TE; ++u) {
      const int i = i0 + u;
      if (i < nl) {
        const float x = coord(A.pts[l + a + i], feat);
        if (lt ? !(x < cut) : !(x <= cut)) mask |= 1u << u;
      }
    }
    int tot;
    int pre = block_excl_scan(__popc(mask), sh, &tot);
#pragma unroll
    for (int u = 0; u < TE; ++u)
      if (mask & (1u << u)) A.tmpA[l + run + pre++] = a + i0 + u;
    run += tot;
  }
  const int m = run;
  if (m == 0) return;
  run = 0;
  for (int c = 0; c < nr; c += TB * TE) {  // misplaced on the right, decreasing index
    const int i0 = c + tid * TE;
    unsigned mask = 0;
#pragma unroll
    for (int u = 0; u < TE; ++u) {
      const int i = i0 + u;
      if (i < nr) {
        const float x = coord(A.pts[l + b - 1 - i], feat);
        if (lt ? (x < cut) : (x <= cut)) mask |= 1u << u;
      }
    }
    int tot;
    int pre = block_excl_scan(__popc(mask), sh, &tot);
#pragma unroll
    for (int u = 0; u < TE; ++u)
      if (mask & (1u << u)) A.tmpB[l + run + pre++] = b - 1 - (i0 + u);
    run += tot;
  }
  __syncthreads();
  for (int k0 = tid; k0 < m; k0 += TB * 4) {  // independent swaps, four per thread in flight
    int ii[4], jj[4];
    float4 pi[4], pj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * TB < m ? k0 + u * TB : k0;
      ii[u] = l + A.tmpA[l + k];
      jj[u] = l + A.tmpB[l + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pi[u] = A.pts[ii[u]];
      pj[u] = A.pts[jj[u]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + u * TB < m) {
        A.pts[ii[u]] = pj[u];
        A.pts[jj[u]] = pi[u];
      }
  }
  __syncthreads();
}

__device__ __forceinline__ int alloc_group(const BuildArgs &A) {
  const int g = atomicAdd(&A.ctl->next_group, 1);
  if ((g + 1) * 8 > A.node_cap) {
    A.ctl->overflow = 1;
    return 0;
  }
  return g * 8;
}

// Process one inner node (block-cooperative).  Children that are inner nodes are returned
// in out[0..1] (count in *n_out) for the caller to schedule.
template <int TB>
__device__ void process_node(const BuildArgs &A, Sh<TB> &sh, const BuildItem &it, BuildItem out[2], int *n_out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = it.l, n = it.r - it.l;
  // ---- min / max of the three coordinates (computeMinMax, :908-920) ---------------
  // UN loads in flight per thread in the streaming passes: a workgroup streams its node alone, so
  // without them every pass is bound by one load latency per point
  constexpr int UN = TB >= 256 ? 8 : 4;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i0 = tid; i0 < n; i0 += TB * UN) {
    float4 p[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * TB;
      p[u] = A.pts[l + (i < n ? i : n - 1)];  // a clamped duplicate of the last point changes no min/max
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      mn[0] = fminf(mn[0], p[u].x); mx[0] = fmaxf(mx[0], p[u].x);
      mn[1] = fminf(mn[1], p[u].y); mx[1] = fmaxf(mx[1], p[u].y);
      mn[2] = fminf(mn[2], p[u].z); mx[2] = fmaxf(mx[2], p[u].z);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float a = wave_min(mn[d]), b = wave_max(mx[d]);
    if (lane == 0) { sh.fmin[d][wave] = a; sh.fmax[d][wave] = b; }
  }
  __syncthreads();
  if (tid == 0) {  // middleSplit_, :982-1031
    float emin[3], emax[3];
    for (int d = 0; d < 3; ++d) {
      emin[d] = sh.fmin[d][0]; emax[d] = sh.fmax[d][0];
      for (int w = 1; w < TB / 64; ++w) { emin[d] = fminf(emin[d], sh.fmin[d][w]); emax[d] = fmaxf(emax[d], sh.fmax[d][w]); }
    }
    store_own_box(A, it.slot, emin[0], emin[1], emin[2], emax[0], emax[1], emax[2]);
    const float EPS = 0.00001f;
    float max_span = it.hi[0] - it.lo[0];
    for (int d = 1; d < 3; ++d) { const float span = it.hi[d] - it.lo[d]; if (span > max_span) max_span = span; }
    float max_spread = -1;
    int cutfeat = 0;
    for (int d = 0; d < 3; ++d) {
      const float span = it.hi[d] - it.lo[d];
      if (span > (1 - EPS) * max_span) {
        const float spread = emax[d] - emin[d];
        if (spread > max_spread) { cutfeat = d; max_spread = spread; }
      }
    }
    const float split_val = (it.lo[cutfeat] + it.hi[cutfeat]) / 2;
    const float cutval = split_val < emin[cutfeat] ? emin[cutfeat] : (split_val > emax[cutfeat] ? emax[cutfeat] : split_val);
    sh.bc_i[0] = cutfeat;
    sh.bc_f[0] = cutval;
  }
  __syncthreads();
  const int feat = sh.bc_i[0];
  const float cut = sh.bc_f[0];
  // ---- lim1 = #(x < cut), lim2 = #(x <= cut) ---------------------------------------
  int c1 = 0, c2 = 0;
  for (int i0 = tid; i0 < n; i0 += TB * UN) {
    float x[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * TB;
      x[u] = coord(A.pts[l + (i < n ? i : n - 1)], feat);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (i0 + u * TB < n) {
        c1 += x[u] < cut;
        c2 += x[u] <= cut;
      }
  }
  c1 = wave_sum_i(c1);
  c2 = wave_sum_i(c2);
  if (lane == 0) { sh.isum[0][wave] = c1; sh.isum[1][wave] = c2; }
  __syncthreads();
  int lim1 = 0, lim2 = 0;
#pragma unroll
  for (int w = 0; w < TB / 64; ++w) { lim1 += sh.isum[0][w]; lim2 += sh.isum[1][w]; }
  __syncthreads();
  // ---- planeSplit: two Hoare passes -----------------------------------------------
  hoare_pass(A, sh, l, 0, n, lim1, feat, cut, true);
  if (lim2 > lim1) hoare_pass(A, sh, l, lim1, n, lim2 - lim1, feat, cut, false);
  const int half = n / 2;
  const int index = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);  // :1024-1029
  // ---- tight bounds of the two halves along the split dimension (:966-971) ----------
  float lmax = -FLT_MAX, rmin = FLT_MAX;
  for (int i0 = tid; i0 < n; i0 += TB * UN) {
    float x[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * TB;
      x[u] = coord(A.pts[l + (i < n ? i : n - 1)], feat);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * TB;
      if (i < n) {
        if (i < index) lmax = fmaxf(lmax, x[u]); else rmin = fminf(rmin, x[u]);
      }
    }
  }
  lmax = wave_max(lmax);
  rmin = wave_min(rmin);
  if (lane == 0) { sh.fmax[0][wave] = lmax; sh.fmin[0][wave] = rmin; }
  __syncthreads();
  if (tid == 0) {
    float divlow = sh.fmax[0][0], divhigh = sh.fmin[0][0];
    for (int w = 1; w < TB / 64; ++w) { divlow = fmaxf(divlow, sh.fmax[0][w]); divhigh = fminf(divhigh, sh.fmin[0][w]); }
    int cnt = 0;
    uint32_t ref[2];
    for (int c = 0; c < 2; ++c) {
      const int cl = c == 0 ? it.l : it.l + index, cr = c == 0 ? it.l + index : it.r;
      if (cr - cl <= 10) {  // leaf (:936-951)
        ref[c] = KD_LEAF | ((uint32_t)cl << 4) | (uint32_t)(cr - cl);
        atomicAdd(&A.ctl->n_leaves, 1);
        atomicMax(&A.ctl->max_depth, it.depth + 1);
      } else {
        BuildItem ch;
        ch.l = cl;
        ch.r = cr;
        for (int d = 0; d < 3; ++d) { ch.lo[d] = it.lo[d]; ch.hi[d] = it.hi[d]; }
        if (c == 0) ch.hi[feat] = cut; else ch.lo[feat] = cut;  // :955-961
        if (it.heap < 3) {  // child stays in this 8-slot group
          ch.heap = 2 * it.heap + 1 + c;
          ch.slot = it.slot - it.heap + ch.heap;
        } else {
          ch.heap = 0;
          ch.slot = alloc_group(A);
        }
        ch.parent_word = it.slot * 4 + 2 + c;
        ch.depth = it.depth + 1;
        ref[c] = (uint32_t)ch.slot << 2;  // the child ORs its divfeat in when it is processed
        out[cnt++] = ch;
      }
    }
    KdNode nd;
    nd.lo = divlow;
    nd.hi = divhigh;
    nd.c1 = ref[0];
    nd.c2 = ref[1];
    A.nodes[it.slot] = nd;
    if (it.parent_word >= 0)
      atomicOr(reinterpret_cast<unsigned int *>(A.nodes) + it.parent_word, (unsigned int)feat);
    else
      A.root_feat[-1 - it.parent_word] = feat;
    *n_out = cnt;
    sh.bc_i[1] = cnt;
  }
  // every wave drains its stores (point swaps, node record) before the barrier, so that
  // one lane's agent-scope release afterwards covers the whole workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Phase A: nodes with more than LOCAL_MAX points, one workgroup per node, global queue.
__global__ __launch_bounds__(TB_BIG) void kd_build_big_kernel(BuildArgs A) {
  constexpr int TB = TB_BIG;
  __shared__ Sh<TB> sh;
  const int tid = threadIdx.x;
  for (;;) {
    if (tid == 0) {
      sh.have = 0;
      for (unsigned spins = 0;; ++spins) {
        const int head = __hip_atomic_load(&A.ctl->q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int tail = __hip_atomic_load(&A.ctl->q_tail_reserved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (head < tail) {
          int expect = head;
          if (__hip_atomic_compare_exchange_strong(&A.ctl->q_head, &expect, head + 1, __ATOMIC_RELAXED,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            // wait until the producer has published entry `head`
            while (__hip_atomic_load(&A.q_ready[head], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
              __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int *src = reinterpret_cast<const int *>(&A.queue[head]);
            int *dst = reinterpret_cast<int *>(&sh.item);
            for (int k = 0; k < (int)(sizeof(BuildItem) / 4); ++k)
              dst[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.have = 1;
            break;
          }
        } else if (__hip_atomic_load(&A.ctl->q_pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 ||
                   __hip_atomic_load(&A.ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          break;  // every big node is done
        } else {
          // idle workgroups poll one cache line that the working ones need for their own atomics
          // (queue tail, pending count, group allocator): back off hard
          __builtin_amdgcn_s_sleep(127);
          __builtin_amdgcn_s_sleep(127);
          // Watchdog only.  No workgroup ever waits for another one while it holds a node (it finishes
          // the node and publishes the children), so q_pending reaches 0 as long as ONE workgroup is
          // resident; this bound (~1 s) merely turns a broken invariant into an error instead of a hang.
          if (spins > A.spin_limit) { A.ctl->overflow = 2; break; }
        }
      }
    }
    __syncthreads();
    if (!sh.have) return;
    // the points of this range were last written by another workgroup: drop stale L1 lines
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    const BuildItem it = sh.item;
    BuildItem kids[2];
    int nk = 0;
    process_node<TB>(A, sh, it, kids, &nk);
    if (tid == 0) {
      bool released = false;
      for (int c = 0; c < nk; ++c) {
        const int cn = kids[c].r - kids[c].l;
        if (cn <= LOCAL_MAX) {  // phase B (next launch: ordinary kernel-boundary visibility)
          const int e = atomicAdd(&A.ctl->n_sub, 1);
          if (e >= A.sub_cap) { A.ctl->overflow = 3; continue; }
          A.sublist[e] = kids[c];
        } else {
          const int e = atomicAdd(&A.ctl->q_tail_reserved, 1);
          if (e >= A.queue_cap) { A.ctl->overflow = 3; continue; }
          if (!released) {  // make this workgroup's point swaps / node writes visible
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            released = true;
          }
          atomicAdd(&A.ctl->q_pending, 1);
          const int *src = reinterpret_cast<const int *>(&kids[c]);
          int *dst = reinterpret_cast<int *>(&A.queue[e]);
          for (int k = 0; k < (int)(sizeof(BuildItem) / 4); ++k)
            __hip_atomic_store(dst + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&A.q_ready[e], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      atomicSub(&A.ctl->q_pending, 1);  // this node is done
    }
    __syncthreads();
  }
}

// Medium phase (round 3): the nodes the level phase left between LOCAL_MAX and HUGE_MIN points (queue entries [0, tail), all
// there before this launch) -- workgroup w takes entries w, w + grid, ... and finishes each subtree down to the wavefront-local
// size by itself, depth first from a stack in LDS: process_node after process_node behind workgroup barriers, no queue, no
// polling, no agent-scope fences (phase A's global queue with 40 roots and their descendants in flight measured 3.1 against
// 1.35 ms).  Pending siblings are disjoint ranges of more than LOCAL_MAX points each: at most huge_min / LOCAL_MAX + 1 of them.
constexpr int MED_STACK = HUGE_MIN_MAX / LOCAL_MAX + 4;
__global__ __launch_bounds__(TB_BIG) void kd_build_medium_kernel(BuildArgs A) {
  constexpr int TB = TB_BIG;
  __shared__ Sh<TB> sh;
  __shared__ BuildItem stack[MED_STACK];
  __shared__ int sp;
  const int tid = threadIdx.x;
  const int tail = min(A.ctl->q_tail_reserved, A.queue_cap);
  for (int e = blockIdx.x; e < tail; e += gridDim.x) {
    if (tid == 0) {
      stack[0] = A.queue[e];
      sp = 1;
    }
    __syncthreads();
    while (sp > 0) {  // uniform: read behind a barrier, changed by thread 0 between barriers
      const BuildItem it = stack[sp - 1];
      __syncthreads();
      BuildItem kids[2];
      int nk = 0;
      process_node<TB>(A, sh, it, kids, &nk);  // (ends with a barrier; kids / nk are thread 0's)
      if (tid == 0) {
        int top = sp - 1;
        for (int c = 0; c < nk; ++c) {
          if (kids[c].r - kids[c].l <= LOCAL_MAX) {
            const int q = atomicAdd(&A.ctl->n_sub, 1);
            if (q >= A.sub_cap) { A.ctl->overflow = 3; continue; }
            A.sublist[q] = kids[c];
          } else if (top < MED_STACK) {
            stack[top++] = kids[c];
          } else {
            A.ctl->overflow = 3;
          }
        }
        sp = top;
      }
      __syncthreads();
    }
  }
}

// Phase B: one wavefront per listed subtree (<= LOCAL_MAX points).  The subtree's points live in
// LDS for the whole build (structure of arrays: x, y, z, bitcast index = 32 KB) together with the two
// index lists of the Hoare pairing (2 x 4 KB of uint16) and the depth-first stack, so every split
// below the hand-over size costs LDS round trips only; the permuted points are written back once.
// A single wavefront needs no barriers: reductions are wave shuffles, the "k-th misplaced" ranks of
// the Hoare pairing come from ballots.  Node-group slots are taken from the global counter eight
// groups at a time; leaf / depth statistics are accumulated per wavefront and added once.
struct SmallItem {
  int32_t a, b;        // point range relative to the subtree's first point
  float lo[3], hi[3];  // propagated bounding box
  int32_t slot, heap, parent_word, depth;
};

constexpr int TINY_ACC = 64;     // cache lines the phase-C wavefronts' leaf counts / depths are spread over
// A node of at most 64 points and its whole subtree, in registers: one point per lane (px, py, pz, pw; lanes >= n idle).
// The subtree is built a LEVEL at a time: the nodes of a level are disjoint lane ranges ("segments"), and every step of
// divideTree / middleSplit_ / planeSplit is done for all of them at once -- segmented DPP scans for the extents, ballots
// masked to the segment for the counts and for the ranks of the Hoare pairing, lane permutes for the swaps.  Same splits,
// same final order of the points as a node-at-a-time loop.  `base` = index of lane 0's point in the point array (leaf
// references); ta / tb: 64 LDS slots each of the calling wavefront (a workgroup is one wavefront).  Used by phase B
// in place (LSLAM_TINY_PHASE=0) and by kd_build_tiny_kernel, which spreads these subtrees over the whole chip.
__device__ __forceinline__ void build_reg_subtree(const BuildArgs &A, const int lane, const int n, const int base, float &px,
                                                  float &py, float &pz, uint32_t &pw, float lo0, float lo1, float lo2, float hi0,
                                                  float hi1, float hi2, int slot, int heap, int pword, int depth, int &grp_next,
                                                  int &grp_end, int &n_leaves, int &reg_depth, uint16_t *ta, uint16_t *tb) {
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int sa = 0, sb = n;  // my segment [sa, sb), in lanes
  bool act = lane < n;
  if (!act) { sa = lane; sb = lane + 1; }
    bool top = true;  // the segment is the popped node: its parent stored a reference already
    const unsigned long long gt_mask = lane == 63 ? 0ull : (~0ull << (lane + 1));
    while (__any(act)) {
      const int ns = sb - sa;
      // computeMinMax (:908-920): inclusive segmented scans, the segment's last lane has the result
      const float mnx = seg_scan<false>(px, lane, sa), mny = seg_scan<false>(py, lane, sa),
                  mnz = seg_scan<false>(pz, lane, sa);
      const float mxx = seg_scan<true>(px, lane, sa), mxy = seg_scan<true>(py, lane, sa),
                  mxz = seg_scan<true>(pz, lane, sa);
      const int last = sb - 1;
      const float en0 = __shfl(mnx, last, 64), en1 = __shfl(mny, last, 64), en2 = __shfl(mnz, last, 64);
      const float ex0 = __shfl(mxx, last, 64), ex1 = __shfl(mxy, last, 64), ex2 = __shfl(mxz, last, 64);
      // middleSplit_ (:982-1031)
      const float EPS = 0.00001f;
      const float sp0 = hi0 - lo0, sp1 = hi1 - lo1, sp2 = hi2 - lo2;
      float max_span = sp0;
      if (sp1 > max_span) max_span = sp1;
      if (sp2 > max_span) max_span = sp2;
      float max_spread = -1;
      int feat = 0;
      if (sp0 > (1 - EPS) * max_span) { const float spread = ex0 - en0; if (spread > max_spread) { feat = 0; max_spread = spread; } }
      if (sp1 > (1 - EPS) * max_span) { const float spread = ex1 - en1; if (spread > max_spread) { feat = 1; max_spread = spread; } }
      if (sp2 > (1 - EPS) * max_span) { const float spread = ex2 - en2; if (spread > max_spread) { feat = 2; max_spread = spread; } }
      const float flo = feat == 0 ? lo0 : (feat == 1 ? lo1 : lo2);
      const float fhi = feat == 0 ? hi0 : (feat == 1 ? hi1 : hi2);
      const float fmn = feat == 0 ? en0 : (feat == 1 ? en1 : en2);
      const float fmx = feat == 0 ? ex0 : (feat == 1 ? ex1 : ex2);
      const float split_val = (flo + fhi) / 2;
      const float cut = split_val < fmn ? fmn : (split_val > fmx ? fmx : split_val);
      float x = feat == 0 ? px : (feat == 1 ? py : pz);
      const unsigned long long segmask = (sb >= 64 ? ~0ull : ((1ull << sb) - 1)) & ~((1ull << sa) - 1);
      const int lim1 = __popcll(__ballot(act && x < cut) & segmask);
      const int lim2 = __popcll(__ballot(act && x <= cut) & segmask);
      // planeSplit (:1043-1078): two Hoare passes; the k-th misplaced element of the left part
      // (increasing index) changes places with the k-th misplaced of the right part (decreasing)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int pa = pass == 0 ? sa : sa + lim1, Lc = pass == 0 ? lim1 : lim2 - lim1;
        const bool go = act && Lc > 0 && sb - pa - Lc > 0;
        const bool below = pass == 0 ? (x < cut) : (x <= cut);
        const bool misl = go && lane >= pa && lane < pa + Lc && !below;
        const bool misr = go && lane >= pa + Lc && below;
        const unsigned long long bl = __ballot(misl), br = __ballot(misr);
        if ((bl | br) == 0) continue;  // wave-uniform
        const int kl = __popcll(bl & segmask & lt_mask), kr = __popcll(br & segmask & gt_mask);
        if (misl) ta[sa + kl] = (uint16_t)lane;
        if (misr) tb[sa + kr] = (uint16_t)lane;
        __syncthreads();
        int partner = lane;
        if (misl) partner = tb[sa + kl];
        if (misr) partner = ta[sa + kr];
        __syncthreads();
        px = __shfl(px, partner, 64);
        py = __shfl(py, partner, 64);
        pz = __shfl(pz, partner, 64);
        pw = (uint32_t)__shfl((int)pw, partner, 64);
        x = feat == 0 ? px : (feat == 1 ? py : pz);
      }
      const int half = ns / 2;
      const int index = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);  // :1024-1029
      const int mid = sa + index;
      // tight bounds of the halves along the split dimension (:966-971): max of the left
      // child's lanes, min (= -max of the negated values) of the right child's
      const bool left = lane < mid;
      const int csa = left ? sa : mid;
      const float v = seg_scan<true>(left ? x : -x, lane, csa);
      const float divlow = __shfl(v, mid - 1, 64), divhigh = -__shfl(v, last, 64);
      // children: leaves (:936-951) or the segments of the next level
      const int cntL = index, cntR = ns - index;
      const bool leafL = cntL <= 10, leafR = cntR <= 10;
      const bool leader = act && lane == sa;
      const bool needL = leader && heap >= 3 && !leafL, needR = leader && heap >= 3 && !leafR;
      const unsigned long long mL = __ballot(needL), mR = __ballot(needR);
      const int total = __popcll(mL) + __popcll(mR);
      if (total > grp_end - grp_next) {  // wave-uniform
        const int take = total > 8 ? total : 8;
        int g = 0;
        if (lane == 0) g = atomicAdd(&A.ctl->next_group, take);
        g = __shfl(g, 0, 64);
        grp_next = g;
        grp_end = g + take;
      }
      int gL = grp_next + __popcll(mL & lt_mask) + __popcll(mR & lt_mask);
      int gR = gL + (needL ? 1 : 0);
      gL = __shfl(gL, sa, 64);  // every lane of the segment takes its leader's numbers
      gR = __shfl(gR, sa, 64);
      grp_next += total;
      int slotL, heapL, slotR, heapR;
      if (heap < 3) {  // the children stay in this 8-slot group
        heapL = 2 * heap + 1; slotL = slot - heap + heapL;
        heapR = 2 * heap + 2; slotR = slot - heap + heapR;
      } else {
        heapL = 0; slotL = gL * 8;
        heapR = 0; slotR = gR * 8;
        if (!leafL && (gL + 1) * 8 > A.node_cap) { if (leader) A.ctl->overflow = 1; slotL = 0; }
        if (!leafR && (gR + 1) * 8 > A.node_cap) { if (leader) A.ctl->overflow = 1; slotR = 0; }
      }
      if (leader) {
        uint32_t *words = reinterpret_cast<uint32_t *>(A.nodes);
        // my reference in the parent: the popped node's parent holds (slot << 2) already and gets
        // the split dimension ORed in (as below); deeper levels write the whole word -- the parent
        // (this wave, a level ago) left the words of its non-leaf children untouched
        if (top) {
          if (pword >= 0) atomicOr(reinterpret_cast<unsigned int *>(words) + pword, (unsigned int)feat);
          else A.root_feat[-1 - pword] = feat;
        } else {
          words[pword] = ((uint32_t)slot << 2) | (uint32_t)feat;
        }
        float *fw = reinterpret_cast<float *>(words + (size_t)slot * 4);
        fw[0] = divlow;
        fw[1] = divhigh;
        store_own_box(A, slot, en0, en1, en2, ex0, ex1, ex2);
        if (leafL) words[(size_t)slot * 4 + 2] = KD_LEAF | ((uint32_t)(base + sa) << 4) | (uint32_t)cntL;
        if (leafR) words[(size_t)slot * 4 + 3] = KD_LEAF | ((uint32_t)(base + mid) << 4) | (uint32_t)cntR;
        if (leafL || leafR) reg_depth = max(reg_depth, depth + 1);
      }
      n_leaves += __popcll(__ballot(leader && leafL)) + __popcll(__ballot(leader && leafR));
      if (act) {
        pword = slot * 4 + (left ? 2 : 3);
        if (left) {
          sb = mid;
          if (feat == 0) hi0 = cut; else if (feat == 1) hi1 = cut; else hi2 = cut;
          slot = slotL; heap = heapL;
          if (leafL) act = false;
        } else {
          sa = mid;
          if (feat == 0) lo0 = cut; else if (feat == 1) lo1 = cut; else lo2 = cut;
          slot = slotR; heap = heapR;
          if (leafR) act = false;
        }
        depth += 1;
      }
      top = false;
    }
}

__global__ __launch_bounds__(64) void kd_build_small_kernel(BuildArgs A) {
  __shared__ float sx[LOCAL_MAX], sy[LOCAL_MAX], sz[LOCAL_MAX];
  __shared__ uint32_t sw[LOCAL_MAX];
  __shared__ uint16_t ta[LOCAL_MAX], tb[LOCAL_MAX];
  __shared__ SmallItem stack[LOCAL_STACK];
  __shared__ int sh_e;
  const int lane = threadIdx.x;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int grp_next = 0, grp_end = 0;   // wave-uniform: pre-allocated node groups [grp_next, grp_end)
  int n_leaves = 0, max_depth = 0;
  int reg_depth = 0;  // per lane: deepest leaf written by the register path
  // The grid is sized for the largest list; the list is complete before this launch.  Wavefronts
  // beyond its length leave without touching the shared counter: 4 096 atomics on one address are
  // 100 us, the whole cost of this kernel for a small tree.
  if ((int)blockIdx.x >= A.ctl->n_sub) return;
  for (;;) {
    if (lane == 0) sh_e = atomicAdd(&A.ctl->sub_next, 1);
    __syncthreads();
    const int e = sh_e;
    __syncthreads();
    if (e >= A.ctl->n_sub) break;
    const BuildItem root = A.sublist[e];
    const int L0 = root.l, N = root.r - root.l;
    for (int i = lane; i < N; i += 64) {
      const float4 p = A.pts[L0 + i];
      sx[i] = p.x; sy[i] = p.y; sz[i] = p.z; sw[i] = __float_as_uint(p.w);
    }
    if (lane == 0) {
      SmallItem r;
      r.a = 0; r.b = N;
      for (int d = 0; d < 3; ++d) { r.lo[d] = root.lo[d]; r.hi[d] = root.hi[d]; }
      r.slot = root.slot; r.heap = root.heap; r.parent_word = root.parent_word; r.depth = root.depth;
      stack[0] = r;
    }
    int sp = 1;  // wave-uniform
    __syncthreads();
    while (sp > 0) {
      const SmallItem it = stack[sp - 1];
      --sp;
      __syncthreads();
      const int a = it.a, n = it.b - it.a;
      if (n <= 64 && A.reg_nodes) {
        // ---- a node of at most 64 points and its whole subtree, in registers -----------------
        // One point per lane.  The subtree is built a LEVEL at a time: the nodes of a level are
        // disjoint lane ranges ("segments"), and every step of divideTree/middleSplit_/planeSplit
        // is done for all of them at once -- segmented shuffle scans for the extents, ballots
        // masked to the segment for the counts and for the ranks of the Hoare pairing, lane
        // permutes for the swaps.  Same splits, same final order of the points as the
        // node-at-a-time loop below (which costs ~3.5 us per node in dependent LDS round trips;
        // three quarters of a subtree's nodes have fewer than 64 points).
        if (A.tiny_list) {  // handed to kd_build_tiny_kernel: one wavefront of the whole chip per such subtree instead of this one for all of its own
          // its slot in the (zero-filled) list is its first point's index / 11: such subtrees are disjoint ranges of more
          // than 10 points, so no two share a slot -- no counter, no atomics (9 000 of them on one address are 200 us)
          // ... and it takes a few node groups with it out of this wavefront's batch (the batch is refilled 16 groups
          // at a time: one atomic per ~5 subtrees; thousands of phase-C wavefronts allocating on their own were 200 us of
          // atomics on one address); a subtree that needs more -- a degenerate one -- allocates the rest itself
          const int want = 1 + n / 16;  // (wave-uniform) 64 points: 5 groups, 32: 3, 11: 1 -- the rest of what such a subtree needs sits in its parent's group
          if (grp_end - grp_next < want) {
            int g = 0;
            if (lane == 0) g = atomicAdd(&A.ctl->next_group, 16);
            g = __shfl(g, 0, 64);
            grp_next = g;
            grp_end = g + 16;
          }
          const int tg0 = grp_next;
          grp_next += want;
          if (lane == 0) {
            const int e = (L0 + a) / 11;
            if (e < A.tiny_cap && (tg0 + want) * 8 <= A.node_cap) {
              BuildItem t;
              t.g0 = tg0; t.g1 = tg0 + want;
              t.l = L0 + a; t.r = L0 + a + n;
              for (int d = 0; d < 3; ++d) { t.lo[d] = it.lo[d]; t.hi[d] = it.hi[d]; }
              t.slot = it.slot; t.heap = it.heap; t.parent_word = it.parent_word; t.depth = it.depth;
              A.tiny_list[e] = t;
            } else {
              A.ctl->overflow = e < A.tiny_cap ? 1 : 3;  // 1: the node array is too small (retried with more slots)
            }
          }
          __syncthreads();
          continue;
        }
        const bool act0 = lane < n;
        const int li = a + (act0 ? lane : 0);
        float px = sx[li], py = sy[li], pz = sz[li];
        uint32_t pw = sw[li];
        build_reg_subtree(A, lane, n, L0 + a, px, py, pz, pw, it.lo[0], it.lo[1], it.lo[2], it.hi[0], it.hi[1], it.hi[2], it.slot, it.heap,
                          it.parent_word, it.depth, grp_next, grp_end, n_leaves, reg_depth, ta, tb);
        if (lane < n) { sx[a + lane] = px; sy[a + lane] = py; sz[a + lane] = pz; sw[a + lane] = pw; }
        __syncthreads();
        continue;
      }
      // ---- computeMinMax (:908-920) ------------------------------------------------------
      float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (int i = lane; i < n; i += 64) {
        const float x = sx[a + i], y = sy[a + i], z = sz[a + i];
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
      }
      float emin[3], emax[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) { emin[d] = wave_min(mn[d]); emax[d] = wave_max(mx[d]); }
      if (lane == 0) store_own_box(A, it.slot, emin[0], emin[1], emin[2], emax[0], emax[1], emax[2]);
      // ---- middleSplit_ (:982-1031), computed by every lane ------------------------------
      const float EPS = 0.00001f;
      float max_span = it.hi[0] - it.lo[0];
#pragma unroll
      for (int d = 1; d < 3; ++d) { const float span = it.hi[d] - it.lo[d]; if (span > max_span) max_span = span; }
      float max_spread = -1;
      int feat = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float span = it.hi[d] - it.lo[d];
        if (span > (1 - EPS) * max_span) {
          const float spread = emax[d] - emin[d];
          if (spread > max_spread) { feat = d; max_spread = spread; }
        }
      }
      const float flo = feat == 0 ? it.lo[0] : (feat == 1 ? it.lo[1] : it.lo[2]);
      const float fhi = feat == 0 ? it.hi[0] : (feat == 1 ? it.hi[1] : it.hi[2]);
      const float fmn = feat == 0 ? emin[0] : (feat == 1 ? emin[1] : emin[2]);
      const float fmx = feat == 0 ? emax[0] : (feat == 1 ? emax[1] : emax[2]);
      const float split_val = (flo + fhi) / 2;
      const float cut = split_val < fmn ? fmn : (split_val > fmx ? fmx : split_val);
      const float *sc = feat == 0 ? sx : (feat == 1 ? sy : sz);
      // ---- lim1 = #(x < cut), lim2 = #(x <= cut) ----------------------------------------
      int c1 = 0, c2 = 0;
      for (int i = lane; i < n; i += 64) {
        const float x = sc[a + i];
        c1 += x < cut;
        c2 += x <= cut;
      }
      const int lim1 = wave_sum_i(c1), lim2 = wave_sum_i(c2);
      // ---- planeSplit (:1043-1078): two Hoare passes -------------------------------------
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        const int pa = pass == 0 ? 0 : lim1, Lc = pass == 0 ? lim1 : lim2 - lim1;
        const int nl = Lc, nr = (n - pa) - Lc;
        if (nl <= 0 || nr <= 0) continue;
        int m = 0;
        for (int c = 0; c < nl; c += 64) {  // misplaced on the left, increasing index
          const int i = c + lane;
          bool f = false;
          if (i < nl) {
            const float x = sc[a + pa + i];
            f = pass == 0 ? !(x < cut) : !(x <= cut);
          }
          const unsigned long long mask = __ballot(f);
          if (f) ta[m + __popcll(mask & lt_mask)] = (uint16_t)(pa + i);
          m += __popcll(mask);
        }
        if (m == 0) continue;
        int m2 = 0;
        for (int c = 0; c < nr; c += 64) {  // misplaced on the right, decreasing index
          const int i = c + lane;
          bool f = false;
          if (i < nr) {
            const float x = sc[a + n - 1 - i];
            f = pass == 0 ? (x < cut) : (x <= cut);
          }
          const unsigned long long mask = __ballot(f);
          if (f) tb[m2 + __popcll(mask & lt_mask)] = (uint16_t)(n - 1 - i);
          m2 += __popcll(mask);
        }
        __syncthreads();
        for (int k = lane; k < m; k += 64) {
          const int i = a + ta[k], j = a + tb[k];
          const float x = sx[i], y = sy[i], z = sz[i];
          const uint32_t w = sw[i];
          sx[i] = sx[j]; sy[i] = sy[j]; sz[i] = sz[j]; sw[i] = sw[j];
          sx[j] = x; sy[j] = y; sz[j] = z; sw[j] = w;
        }
        __syncthreads();
      }
      const int half = n / 2;
      const int index = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);  // :1024-1029
      // ---- tight bounds of the halves along the split dimension (:966-971) -----------------
      float lmax = -FLT_MAX, rmin = FLT_MAX;
      for (int i = lane; i < n; i += 64) {
        const float x = sc[a + i];
        if (i < index) lmax = fmaxf(lmax, x); else rmin = fminf(rmin, x);
      }
      const float divlow = wave_max(lmax), divhigh = wave_min(rmin);
      // ---- children (every lane computes them; lane 0 writes) --------------------------------
      uint32_t ref[2];
#pragma unroll
      for (int c = 1; c >= 0; --c) {  // right child pushed first: the left one is built next
        const int ca = c == 0 ? it.a : it.a + index, cb = c == 0 ? it.a + index : it.b;
        if (cb - ca <= 10) {  // leaf (:936-951)
          ref[c] = KD_LEAF | ((uint32_t)(L0 + ca) << 4) | (uint32_t)(cb - ca);
          ++n_leaves;
          max_depth = max(max_depth, it.depth + 1);
        } else {
          SmallItem ch;
          ch.a = ca; ch.b = cb;
#pragma unroll
          for (int d = 0; d < 3; ++d) { ch.lo[d] = it.lo[d]; ch.hi[d] = it.hi[d]; }
          if (c == 0) { if (feat == 0) ch.hi[0] = cut; else if (feat == 1) ch.hi[1] = cut; else ch.hi[2] = cut; }
          else        { if (feat == 0) ch.lo[0] = cut; else if (feat == 1) ch.lo[1] = cut; else ch.lo[2] = cut; }
          if (it.heap < 3) {  // child stays in this 8-slot group
            ch.heap = 2 * it.heap + 1 + c;
            ch.slot = it.slot - it.heap + ch.heap;
          } else {
            if (grp_next == grp_end) {  // take eight groups at once
              int g = 0;
              if (lane == 0) g = atomicAdd(&A.ctl->next_group, 8);
              g = __shfl(g, 0, 64);
              grp_next = g;
              grp_end = g + 8;
            }
            ch.heap = 0;
            ch.slot = grp_next * 8;
            if ((grp_next + 1) * 8 > A.node_cap) {
              if (lane == 0) A.ctl->overflow = 1;
              ch.slot = 0;
            }
            ++grp_next;
          }
          ch.parent_word = it.slot * 4 + 2 + c;
          ch.depth = it.depth + 1;
          ref[c] = (uint32_t)ch.slot << 2;  // the child ORs its divfeat in when it is processed
          if (sp < LOCAL_STACK) {
            if (lane == 0) stack[sp] = ch;
            ++sp;
          } else if (lane == 0) {
            A.ctl->overflow = 4;
          }
        }
      }
      if (lane == 0) {
        KdNode nd;
        nd.lo = divlow;
        nd.hi = divhigh;
        nd.c1 = ref[0];
        nd.c2 = ref[1];
        // the children's divfeat bits may already be... no: children run after this store (same wave,
        // program order), and they OR into these words atomically
        A.nodes[it.slot] = nd;
        if (it.parent_word >= 0)
          atomicOr(reinterpret_cast<unsigned int *>(A.nodes) + it.parent_word, (unsigned int)feat);
        else
          A.root_feat[-1 - it.parent_word] = feat;
      }
      __syncthreads();
    }
    for (int i = lane; i < N; i += 64) A.pts[L0 + i] = make_float4(sx[i], sy[i], sz[i], __uint_as_float(sw[i]));
    __syncthreads();
  }
  // n_leaves / max_depth were counted by every lane identically, reg_depth per lane
  int md = max(max_depth, reg_depth);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) md = max(md, __shfl_xor(md, o, 64));
  if (lane == 0) {
    if (n_leaves) atomicAdd(&A.ctl->n_leaves, n_leaves);
    atomicMax(&A.ctl->max_depth, md);
  }
}

// Phase C: the subtrees of at most 64 points -- three quarters of all nodes, and until round 3 two thirds of phase B's time,
// where each wavefront finished the ~30 such subtrees of its own 1 536-point subtree one after the other (~9 us each): here every
// one of them gets a wavefront of its own, the whole list spread over the chip (wavefront w takes entries w, w + G, ...: no
// shared counter).  The points come from and go back to HBM (64 x 16 B per subtree).
constexpr int TINY_SLOTS = 8;
__global__ __launch_bounds__(64) void kd_build_tiny_kernel(BuildArgs A) {
  __shared__ uint16_t ta[64], tb[64];
  const int lane = threadIdx.x;
  int grp_next = 0, grp_end = 0, n_leaves = 0, reg_depth = 0;
  // the list is sparse (slot = first point / 11, an unused slot is all zero; about one slot in six is used): a wavefront
  // looks at TINY_SLOTS slots at a time -- one or two subtrees each, so that the list spreads over ~6 000 wavefronts
  for (int s0 = blockIdx.x * TINY_SLOTS; s0 < A.tiny_cap; s0 += gridDim.x * TINY_SLOTS) {
    const int sidx = s0 + lane;
    const bool have = lane < TINY_SLOTS && sidx < A.tiny_cap && A.tiny_list[sidx].r > A.tiny_list[sidx].l;
    unsigned long long todo = __ballot(have);
    while (todo) {
      const int k = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const BuildItem it = A.tiny_list[s0 + k];
      const int n = it.r - it.l;
      grp_next = it.g0;
      grp_end = it.g1;
      const float4 p = A.pts[it.l + (lane < n ? lane : 0)];
      float px = p.x, py = p.y, pz = p.z;
      uint32_t pw = __float_as_uint(p.w);
      build_reg_subtree(A, lane, n, it.l, px, py, pz, pw, it.lo[0], it.lo[1], it.lo[2], it.hi[0], it.hi[1], it.hi[2], it.slot, it.heap,
                        it.parent_word, it.depth, grp_next, grp_end, n_leaves, reg_depth, ta, tb);
      if (lane < n) A.pts[it.l + lane] = make_float4(px, py, pz, __uint_as_float(pw));
      __syncthreads();
    }
  }
  int md = reg_depth;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) md = max(md, __shfl_xor(md, o, 64));
  if (lane == 0) {
    // thousands of wavefronts end here: their two counters are spread over TINY_ACC cache lines (13 000 atomics on ONE
    // address were most of this kernel's time), summed by the host
    int32_t *acc = A.tiny_acc + (blockIdx.x % TINY_ACC) * 32;
    if (n_leaves) atomicAdd(acc, n_leaves);
    if (md) atomicMax(acc + 1, md);
  }
}

// ---------------------------------------------------------------------------------------------
// Phase 0: nodes with more than HUGE_MIN points, level by level, every node of a level spread
// over the whole chip in chunks of LV_CH points.  One workgroup streaming a 1.26 M-point root alone
// was 2.5 ms of a 10 ms build; here each pass of a level is one launch over all its chunks, with
// per-node results combined by integer atomics (min/max on order-preserving ints, counts), so the
// outcome is independent of scheduling.  The Hoare pairing needs the rank of every misplaced
// element inside its node: per-chunk counts, a per-node scan of the chunk counts, then ranks
// inside the chunk.
constexpr int LV_CH = 4096;
constexpr int LV_TB = 256;
constexpr int LV_PER = LV_CH / LV_TB;  // consecutive elements per thread

struct LvStat {
  int32_t mn[3], mx[3];      // order-preserving ints of the coordinate extrema (written by whoever made the node: the host
                             // for a root -- its bounding box IS its extrema --, the parent's lv_bounds pass for the others)
  int32_t feat;
  float cut;
  int32_t lim1, lim2;
  int32_t m[2];              // misplaced pairs of the two Hoare passes
  int32_t lmax, rmin;        // order-preserving ints
  int32_t cmn[2][3], cmx[2][3];  // extrema of the two children's points (lv_bounds_kernel): next level's mn / mx
};
__device__ __host__ inline void lv_stat_reset(LvStat &st) {  // everything but mn / mx
  st.feat = 0;
  st.cut = 0.f;
  st.lim1 = st.lim2 = 0;
  st.m[0] = st.m[1] = 0;
  st.lmax = INT32_MIN;
  st.rmin = INT32_MAX;
  for (int c = 0; c < 2; ++c)
    for (int d = 0; d < 3; ++d) { st.cmn[c][d] = INT32_MAX; st.cmx[c][d] = INT32_MIN; }
}

// What a chunk's workgroup needs to know about its chunk, in ONE record written by the level's set-up (the passes of a level
// are short kernels whose time is a chain of dependent loads: header -> chunk's node -> node's item -> node's statistics ->
// points was five deep, record -> statistics -> points is three).  n == 0: no such chunk at this level.
struct LvChunk {
  int32_t node, l, n, c0, first, pad[3];
};

struct LvArgs {
  BuildArgs A;
  const BuildItem *items;    // nodes of this level
  LvStat *stat;              // [n_nodes]
  LvStat *stat_next;         // the next level's (filled by lv_final_kernel for the children that stay in phase 0)
  int32_t *final_done;       // lv_final_kernel's ticket counter: its last block sets up the next level
  LvChunk *rec;              // [cap_chunks] the chunks of this level (lv_setup_body)
  const int32_t *chunk_node; // [n_chunks] node of a chunk
  const int32_t *chunk_first;// [n_nodes] first chunk of a node
  int32_t *cntL, *cntR, *baseL, *baseR;  // [n_chunks]
  BuildItem *next_items;     // children that are huge again
  const int32_t *cur_count;  // number of items of this level (written by the previous level / the host)
  int32_t *next_count;
  int32_t *hdr;              // [0] nodes, [1] chunks of this level (lv_setup_kernel)
  int32_t *chunk_node_w, *chunk_first_w;  // writable views for lv_setup_kernel
  int32_t cap_nodes, cap_chunks, next_cap;
};

__device__ __forceinline__ int32_t ord_i(float f) {
  const int32_t i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord_f(int32_t i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// block-wide exclusive scan of one int per thread (LV_TB threads); *total = block sum
__device__ __forceinline__ int lv_scan(int v, int *sh /*[LV_TB/64]*/, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(x, o, 64);
    if (lane >= o) x += t;
  }
  if (lane == 63) sh[wave] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < LV_TB / 64; ++w) {
    if (w < wave) base += sh[w];
    tot += sh[w];
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

#define LV_PROLOGUE                                                     \
  const LvChunk rc_ = L.rec[blockIdx.x];                                \
  if (rc_.n == 0) return;                                               \
  const int node = rc_.node;                                            \
  const int n = rc_.n, l = rc_.l;                                       \
  const int c0 = rc_.c0;                                                \
  const int c1 = min(n, c0 + LV_CH);                                    \
  (void)c1; (void)l;

// One workgroup: the bookkeeping of a level, on the device so that the host never has to look
// between levels -- node count, chunks per node, chunk -> node table and an empty level after it.
// Grids are launched at capacity; blocks beyond hdr[] exit at once.  Run by lv_setup_kernel for the first level and by
// the LAST block of lv_final_kernel for every later one (no launch of its own).  TBS threads (a multiple of 64).
template <int TBS>
__device__ __forceinline__ void lv_setup_body(const LvArgs &L, const BuildItem *items, const int32_t *count, int32_t *empty_count,
                                              int *wsum /* [TBS / 64] */, int *run_sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n_nodes = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n_nodes > L.cap_nodes) {
    if (tid == 0) L.A.ctl->overflow = 3;
    n_nodes = 0;
  }
  if (tid == 0) *run_sh = 0;
  __syncthreads();
  for (int base = 0; base < n_nodes; base += TBS) {
    const int j = base + tid;
    int nch = 0;
    if (j < n_nodes) {
      const int32_t l = __hip_atomic_load(&items[j].l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int32_t r = __hip_atomic_load(&items[j].r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nch = (r - l + LV_CH - 1) / LV_CH;
    }
    int x = nch;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(x, o, 64);
      if (lane >= o) x += t;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int before = *run_sh;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (j < n_nodes) L.chunk_first_w[j] = before + x - nch;
    __syncthreads();
    if (tid == TBS - 1) *run_sh = before + x;
    __syncthreads();
  }
  int n_chunks = *run_sh;
  if (n_chunks > L.cap_chunks) {
    if (tid == 0) L.A.ctl->overflow = 3;
    n_chunks = 0;
    n_nodes = 0;
  }
  __threadfence_block();
  __syncthreads();
  for (int c = tid; c < L.cap_chunks; c += TBS) {  // last node whose first chunk is <= c
    LvChunk rc{};
    if (c < n_chunks) {
      int lo = 0, hi = n_nodes - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (L.chunk_first_w[mid] <= c) lo = mid; else hi = mid - 1;
      }
      L.chunk_node_w[c] = lo;
      rc.node = lo;
      rc.l = __hip_atomic_load(&items[lo].l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      rc.n = __hip_atomic_load(&items[lo].r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - rc.l;
      rc.first = L.chunk_first_w[lo];
      rc.c0 = (c - rc.first) * LV_CH;
    }
    L.rec[c] = rc;  // (n == 0 beyond the level's chunks: those workgroups leave at once)
  }
  if (tid == 0) {
    L.hdr[0] = n_nodes;
    L.hdr[1] = n_chunks;
    *empty_count = 0;
    *L.final_done = 0;
  }
}

__global__ __launch_bounds__(1024) void lv_setup_kernel(LvArgs L) {
  __shared__ int wsum[16];
  __shared__ int run_sh;
  lv_setup_body<1024>(L, L.items, L.cur_count, L.next_count, wsum, &run_sh);
}

// middleSplit_ (:982-1031) from the node's extrema: evaluated identically by every block of the node
__device__ __forceinline__ void lv_split(const BuildItem &it, const LvStat &st, int *feat, float *cut) {
  float emin[3], emax[3];
  for (int d = 0; d < 3; ++d) { emin[d] = ord_f(st.mn[d]); emax[d] = ord_f(st.mx[d]); }
  const float EPS = 0.00001f;
  float max_span = it.hi[0] - it.lo[0];
  for (int d = 1; d < 3; ++d) { const float span = it.hi[d] - it.lo[d]; if (span > max_span) max_span = span; }
  float max_spread = -1;
  int cutfeat = 0;
  for (int d = 0; d < 3; ++d) {
    const float span = it.hi[d] - it.lo[d];
    if (span > (1 - EPS) * max_span) {
      const float spread = emax[d] - emin[d];
      if (spread > max_spread) { cutfeat = d; max_spread = spread; }
    }
  }
  const float split_val = (it.lo[cutfeat] + it.hi[cutfeat]) / 2;
  *feat = cutfeat;
  *cut = split_val < emin[cutfeat] ? emin[cutfeat] : (split_val > emax[cutfeat] ? emax[cutfeat] : split_val);
}

__global__ __launch_bounds__(LV_TB) void lv_count_kernel(LvArgs L) {
  LV_PROLOGUE
  const BuildItem it = L.items[node];
  int feat;
  float cut;
  lv_split(it, L.stat[node], &feat, &cut);
  if (c0 == 0 && threadIdx.x == 0) {  // the node's first chunk publishes the split for the later passes
    {
      const LvStat &sb = L.stat[node];
      store_own_box(L.A, it.slot, ord_f(sb.mn[0]), ord_f(sb.mn[1]), ord_f(sb.mn[2]), ord_f(sb.mx[0]), ord_f(sb.mx[1]), ord_f(sb.mx[2]));
    }
    L.stat[node].feat = feat;
    L.stat[node].cut = cut;
  }
  int a = 0, b = 0;
  {  // a chunk is LV_CH / LV_TB = 16 points per thread: all loads issued before the first use (a rolled loop pays one
     // memory round trip per point: this kernel was 22 us at the top levels)
    float xv[LV_CH / LV_TB];
#pragma unroll
    for (int u = 0; u < LV_CH / LV_TB; ++u) {
      const int i = c0 + threadIdx.x + u * LV_TB;
      xv[u] = coord(L.A.pts[l + (i < c1 ? i : c0)], feat);
    }
#pragma unroll
    for (int u = 0; u < LV_CH / LV_TB; ++u) {
      if (c0 + (int)threadIdx.x + u * LV_TB >= c1) continue;
      a += xv[u] < cut;
      b += xv[u] <= cut;
    }
  }
  a = wave_sum_i(a);
  b = wave_sum_i(b);
  __shared__ int shc[2][LV_TB / 64];
  if ((threadIdx.x & 63) == 0) {
    shc[0][threadIdx.x >> 6] = a;
    shc[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x < 2) {  // one pair of atomics per chunk
    int v = 0;
#pragma unroll
    for (int w = 0; w < LV_TB / 64; ++w) v += shc[threadIdx.x][w];
    if (v) atomicAdd(threadIdx.x == 0 ? &L.stat[node].lim1 : &L.stat[node].lim2, v);
  }
}

// region of Hoare pass p inside the node: [pa, n), its first Lc elements belong left
__device__ __forceinline__ void lv_region(const LvStat &st, int p, int *pa, int *Lc) {
  *pa = p == 0 ? 0 : st.lim1;
  *Lc = p == 0 ? st.lim1 : st.lim2 - st.lim1;
}
// bit u of the masks: element c0 + tid*LV_PER + u is a misplaced one of the left / right part
__device__ __forceinline__ void lv_flags(const LvArgs &L, int l, int n, int c0, int feat, float cut, int p, int pa,
                                         int Lc, unsigned *mL, unsigned *mR) {
  unsigned a = 0, b = 0;
  const int i0 = c0 + threadIdx.x * LV_PER;
  float x[LV_PER];
#pragma unroll
  for (int u = 0; u < LV_PER; ++u) x[u] = (i0 + u < n) ? coord(L.A.pts[l + i0 + u], feat) : 0.0f;
#pragma unroll
  for (int u = 0; u < LV_PER; ++u) {
    const int i = i0 + u;
    if (i >= n || i < pa) continue;
    const bool left_ok = p == 0 ? (x[u] < cut) : (x[u] <= cut);
    if (i < pa + Lc) { if (!left_ok) a |= 1u << u; }
    else if (left_ok) b |= 1u << u;
  }
  *mL = a;
  *mR = b;
}

__global__ __launch_bounds__(LV_TB) void lv_hflags_kernel(LvArgs L, int p) {
  LV_PROLOGUE
  __shared__ int sh[LV_TB / 64];
  const LvStat st = L.stat[node];
  int pa, Lc;
  lv_region(st, p, &pa, &Lc);
  if (Lc == 0 || Lc == n - pa) {  // nothing belongs left, or everything does: no pairs (the usual case of pass 2)
    if (threadIdx.x == 0) {
      L.cntL[blockIdx.x] = 0;
      L.cntR[blockIdx.x] = 0;
    }
    return;
  }
  unsigned mL, mR;
  lv_flags(L, l, n, c0, st.feat, st.cut, p, pa, Lc, &mL, &mR);
  int tL, tR;
  lv_scan(__popc(mL), sh, &tL);
  lv_scan(__popc(mR), sh, &tR);
  if (threadIdx.x == 0) {
    L.cntL[blockIdx.x] = tL;
    L.cntR[blockIdx.x] = tR;
  }
}

__global__ __launch_bounds__(LV_TB) void lv_hwrite_kernel(LvArgs L, int p) {
  LV_PROLOGUE
  __shared__ int sh[LV_TB / 64];
  // ranks of this chunk inside its node: misplaced-left elements of the chunks before it, wanted-left
  // elements of the chunks after it (the k-th from the RIGHT), and the node's total
  const int first = rc_.first, nch = (n + LV_CH - 1) / LV_CH, mine = (int)blockIdx.x - first;
  int bl = 0, br = 0, tot = 0;
  for (int c = threadIdx.x; c < nch; c += LV_TB) {
    const int vl = L.cntL[first + c], vr = L.cntR[first + c];
    tot += vl;
    if (c < mine) bl += vl;
    if (c > mine) br += vr;
  }
  int baseL, baseR, m;
  lv_scan(bl, sh, &baseL);
  lv_scan(br, sh, &baseR);
  lv_scan(tot, sh, &m);
  if (mine == 0 && threadIdx.x == 0) L.stat[node].m[p] = m;
  if (m == 0) return;
  const LvStat st = L.stat[node];
  int pa, Lc;
  lv_region(st, p, &pa, &Lc);
  unsigned mL, mR;
  lv_flags(L, l, n, c0, st.feat, st.cut, p, pa, Lc, &mL, &mR);
  int tL, tR;
  int eL = lv_scan(__popc(mL), sh, &tL);
  const int eR = lv_scan(__popc(mR), sh, &tR);
  const int i0 = c0 + threadIdx.x * LV_PER;
  int posL = baseL + eL;
  int after = tR - eR;  // wanted-left elements of this chunk at or after this thread's first element
#pragma unroll
  for (int u = 0; u < LV_PER; ++u) {
    if (mL & (1u << u)) L.A.tmpA[l + posL++] = i0 + u;
    if (mR & (1u << u)) {
      --after;  // now: those strictly after this one
      L.A.tmpB[l + baseR + after] = i0 + u;
    }
  }
}

__global__ __launch_bounds__(LV_TB) void lv_hswap_kernel(LvArgs L, int p) {
  LV_PROLOGUE
  const int m = L.stat[node].m[p];
  for (int k = c0 + threadIdx.x; k < min(m, c0 + LV_CH); k += LV_TB) {
    const int i = l + L.A.tmpA[l + k], j = l + L.A.tmpB[l + k];
    const float4 t = L.A.pts[i];
    L.A.pts[i] = L.A.pts[j];
    L.A.pts[j] = t;
  }
}

// The SECOND Hoare pass of planeSplit (:1062-1076: the elements equal to the cut value move to the front of the right part) as
// ONE launch: a workgroup per node.  Almost always there is nothing to do -- no element equals the cut, or all of the right
// part does -- and the three launches the first pass needs (flags, ranks, swaps) were three empty launches per level; a node
// that does have such elements is processed here chunk after chunk by its one workgroup (same flags, same ranks, same pairing:
// the k-th misplaced element from the left with the k-th from the right), which is slow only for degenerate clouds.
__global__ __launch_bounds__(LV_TB) void lv_pass2_kernel(LvArgs L) {
  const int node = blockIdx.x;
  if (node >= L.hdr[0]) return;
  const BuildItem it = L.items[node];
  const int n = it.r - it.l, l = it.l;
  const LvStat st = L.stat[node];
  int pa, Lc;
  lv_region(st, 1, &pa, &Lc);
  if (Lc == 0 || Lc == n - pa) return;  // nothing belongs left, or everything does: no pairs
  __shared__ int sh[LV_TB / 64];
  __shared__ int run[3];
  const int first = L.chunk_first[node], nch = (n + LV_CH - 1) / LV_CH;
  const int ch0 = pa / LV_CH;  // chunks before the region hold nothing of it
  for (int c = ch0; c < nch; ++c) {  // per-chunk counts
    unsigned mL, mR;
    lv_flags(L, l, n, c * LV_CH, st.feat, st.cut, 1, pa, Lc, &mL, &mR);
    int tL, tR;
    lv_scan(__popc(mL), sh, &tL);
    lv_scan(__popc(mR), sh, &tR);
    if (threadIdx.x == 0) {
      L.cntL[first + c] = tL;
      L.cntR[first + c] = tR;
    }
  }
  __syncthreads();
  int tot = 0, totR = 0;
  for (int c = ch0 + (int)threadIdx.x; c < nch; c += LV_TB) { tot += L.cntL[first + c]; totR += L.cntR[first + c]; }
  int m, mr;
  lv_scan(tot, sh, &m);
  lv_scan(totR, sh, &mr);
  if (threadIdx.x == 0) L.stat[node].m[1] = m;
  if (m == 0) return;
  if (threadIdx.x == 0) { run[0] = 0; run[1] = mr; }
  __syncthreads();
  for (int c = ch0; c < nch; ++c) {  // ranks: misplaced-left elements of the chunks before, wanted-left elements of the chunks after
    const int cl = L.cntL[first + c], cr = L.cntR[first + c];
    const int baseL = run[0], baseR = run[1] - cr;
    unsigned mL, mR;
    lv_flags(L, l, n, c * LV_CH, st.feat, st.cut, 1, pa, Lc, &mL, &mR);
    int tL, tR;
    const int eL = lv_scan(__popc(mL), sh, &tL);
    const int eR = lv_scan(__popc(mR), sh, &tR);
    const int i0 = c * LV_CH + threadIdx.x * LV_PER;
    int posL = baseL + eL;
    int after = tR - eR;
#pragma unroll
    for (int u = 0; u < LV_PER; ++u) {
      if (mL & (1u << u)) L.A.tmpA[l + posL++] = i0 + u;
      if (mR & (1u << u)) {
        --after;
        L.A.tmpB[l + baseR + after] = i0 + u;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { run[0] = baseL + cl; run[1] = baseR; }
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  for (int k = threadIdx.x; k < m; k += LV_TB) {
    const int i = l + L.A.tmpA[l + k], j = l + L.A.tmpB[l + k];
    const float4 t = L.A.pts[i];
    L.A.pts[i] = L.A.pts[j];
    L.A.pts[j] = t;
  }
}

__device__ __forceinline__ int lv_index(const LvStat &st, int n) {  // :1024-1029
  const int half = n / 2;
  return st.lim1 > half ? st.lim1 : (st.lim2 < half ? st.lim2 : half);
}

// divlow / divhigh (:966-971) and, in the same pass over the node's points, the coordinate extrema of its two children:
// what the next level's middleSplit_ needs (the min / max pass every level used to start with).
__global__ __launch_bounds__(LV_TB) void lv_bounds_kernel(LvArgs L) {
  LV_PROLOGUE
  const LvStat st = L.stat[node];
  const int index = lv_index(st, n);
  float4 pv[LV_CH / LV_TB];
#pragma unroll
  for (int u = 0; u < LV_CH / LV_TB; ++u) {
    const int i = c0 + threadIdx.x + u * LV_TB;
    pv[u] = L.A.pts[l + (i < c1 ? i : c0)];
  }
  // a chunk straddles the children's border at most once: side 0 = left child
  float mn[2][3], mx[2][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int d = 0; d < 3; ++d) { mn[c][d] = FLT_MAX; mx[c][d] = -FLT_MAX; }
#pragma unroll
  for (int u = 0; u < LV_CH / LV_TB; ++u) {
    const int i = c0 + threadIdx.x + u * LV_TB;
    if (i >= c1) continue;
    const float4 p = pv[u];
    const bool right = i >= index;
    mn[0][0] = right ? mn[0][0] : fminf(mn[0][0], p.x); mx[0][0] = right ? mx[0][0] : fmaxf(mx[0][0], p.x);
    mn[0][1] = right ? mn[0][1] : fminf(mn[0][1], p.y); mx[0][1] = right ? mx[0][1] : fmaxf(mx[0][1], p.y);
    mn[0][2] = right ? mn[0][2] : fminf(mn[0][2], p.z); mx[0][2] = right ? mx[0][2] : fmaxf(mx[0][2], p.z);
    mn[1][0] = right ? fminf(mn[1][0], p.x) : mn[1][0]; mx[1][0] = right ? fmaxf(mx[1][0], p.x) : mx[1][0];
    mn[1][1] = right ? fminf(mn[1][1], p.y) : mn[1][1]; mx[1][1] = right ? fmaxf(mx[1][1], p.y) : mx[1][1];
    mn[1][2] = right ? fminf(mn[1][2], p.z) : mn[1][2]; mx[1][2] = right ? fmaxf(mx[1][2], p.z) : mx[1][2];
  }
  __shared__ float shb[12][LV_TB / 64];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float a = wave_min(mn[c][d]), b = wave_max(mx[c][d]);
      if ((threadIdx.x & 63) == 0) {
        shb[c * 3 + d][threadIdx.x >> 6] = a;
        shb[6 + c * 3 + d][threadIdx.x >> 6] = b;
      }
    }
  __syncthreads();
  if (threadIdx.x < 12) {  // one set of atomics per chunk, and only where the chunk improves on what the node already has
    const int k = threadIdx.x;
    const bool is_max = k >= 6;
    float v = shb[k][0];
#pragma unroll
    for (int w = 1; w < LV_TB / 64; ++w) v = is_max ? fmaxf(v, shb[k][w]) : fminf(v, shb[k][w]);
    const int c = (k % 6) / 3, d = k % 3;
    if (is_max ? v > -FLT_MAX : v < FLT_MAX) {  // the chunk holds points of that child
      const int32_t o = ord_i(v);
      int32_t *dst = is_max ? &L.stat[node].cmx[c][d] : &L.stat[node].cmn[c][d];
      const int32_t cur = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (is_max) { if (o > cur) atomicMax(dst, o); }
      else        { if (o < cur) atomicMin(dst, o); }
      // the tight bounds of the split (:966-971) are the children's extrema along the split dimension: the left child's
      // maximum and the right child's minimum
      if (d == st.feat && ((is_max && c == 0) || (!is_max && c == 1))) {
        int32_t *bd = is_max ? &L.stat[node].lmax : &L.stat[node].rmin;
        const int32_t cb = __hip_atomic_load(bd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (is_max) { if (o > cb) atomicMax(bd, o); }
        else        { if (o < cb) atomicMin(bd, o); }
      }
    }
  }
}

// one thread per node: the node record and its two children (what lane 0 of process_node does)
__device__ __forceinline__ void lv_final_node(const LvArgs &L, int node);
__global__ __launch_bounds__(64) void lv_final_kernel(LvArgs L) {
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node < L.hdr[0]) lv_final_node(L, node);
  // The last block to get here sets up the next level (what lv_setup_kernel did in a launch of its own): the children that
  // stay in phase 0 were appended with device-scope atomics and their ranges written through to the coherence point.
  __shared__ int last, wsum[1], run_sh;
  __builtin_amdgcn_s_waitcnt(0);
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(L.final_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = ticket == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  lv_setup_body<64>(L, L.next_items, L.next_count, const_cast<int32_t *>(L.cur_count), wsum, &run_sh);
}
__device__ __forceinline__ void lv_final_node(const LvArgs &L, int node) {
  const BuildArgs &A = L.A;
  const BuildItem it = L.items[node];
  const LvStat st = L.stat[node];
  const int n = it.r - it.l, index = lv_index(st, n), feat = st.feat;
  uint32_t ref[2];
  for (int c = 0; c < 2; ++c) {
    const int cl = c == 0 ? it.l : it.l + index, cr = c == 0 ? it.l + index : it.r;
    if (cr - cl <= 10) {
      ref[c] = KD_LEAF | ((uint32_t)cl << 4) | (uint32_t)(cr - cl);
      atomicAdd(&A.ctl->n_leaves, 1);
      atomicMax(&A.ctl->max_depth, it.depth + 1);
      continue;
    }
    BuildItem ch;
    ch.l = cl;
    ch.r = cr;
    for (int d = 0; d < 3; ++d) { ch.lo[d] = it.lo[d]; ch.hi[d] = it.hi[d]; }
    if (c == 0) ch.hi[feat] = st.cut; else ch.lo[feat] = st.cut;
    if (it.heap < 3) {
      ch.heap = 2 * it.heap + 1 + c;
      ch.slot = it.slot - it.heap + ch.heap;
    } else {
      ch.heap = 0;
      ch.slot = alloc_group(A);
    }
    ch.parent_word = it.slot * 4 + 2 + c;
    ch.depth = it.depth + 1;
    ref[c] = (uint32_t)ch.slot << 2;
    const int cn = cr - cl;
    if (cn > A.huge_min) {
      const int e = atomicAdd(L.next_count, 1);
      if (e < L.next_cap) {
        L.next_items[e] = ch;
        // (the range once more, written through: the block that sets up the next level may sit on another XCD)
        __hip_atomic_store(&L.next_items[e].l, ch.l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&L.next_items[e].r, ch.r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        LvStat cs;
        for (int d = 0; d < 3; ++d) { cs.mn[d] = st.cmn[c][d]; cs.mx[d] = st.cmx[c][d]; }
        lv_stat_reset(cs);
        L.stat_next[e] = cs;
      } else {
        A.ctl->overflow = 3;
      }
    } else if (cn > LOCAL_MAX) {
      const int e = atomicAdd(&A.ctl->q_tail_reserved, 1);
      if (e >= A.queue_cap) { A.ctl->overflow = 3; continue; }
      atomicAdd(&A.ctl->q_pending, 1);
      A.queue[e] = ch;
      A.q_ready[e] = 1;
    } else {
      const int e = atomicAdd(&A.ctl->n_sub, 1);
      if (e >= A.sub_cap) { A.ctl->overflow = 3; continue; }
      A.sublist[e] = ch;
    }
  }
  KdNode nd;
  nd.lo = ord_f(st.lmax);
  nd.hi = ord_f(st.rmin);
  nd.c1 = ref[0];
  nd.c2 = ref[1];
  // the children OR their split dimension into c1/c2 later (next level or phases A/B), never before
  // this store: they are processed by later launches
  KdNode *dst = &A.nodes[it.slot];
  dst->lo = nd.lo;
  dst->hi = nd.hi;
  atomicOr(&dst->c1, nd.c1);  // the slot was zeroed; OR-ing keeps the word consistent with the children's later ORs
  atomicOr(&dst->c2, nd.c2);
  if (it.parent_word >= 0)
    atomicOr(reinterpret_cast<unsigned int *>(A.nodes) + it.parent_word, (unsigned int)feat);
  else
    A.root_feat[-1 - it.parent_word] = feat;
}

// bounding box of the whole cloud (computeBoundingBox, :1406-1427) + identity .w
__global__ __launch_bounds__(256) void kd_bbox_kernel(const float4 *pts, int n, float *part) {
  __shared__ float smin[3][4], smax[3][4];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int d = 0; d < 3; ++d) {
    const float a = wave_min(mn[d]), b = wave_max(mx[d]);
    if (lane == 0) { smin[d][wave] = a; smax[d][wave] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    float a = smin[d][0], b = smax[d][0];
    for (int w = 1; w < 4; ++w) { a = fminf(a, smin[d][w]); b = fmaxf(b, smax[d][w]); }
    part[blockIdx.x * 6 + d] = a;
    part[blockIdx.x * 6 + 3 + d] = b;
  }
}

// The root of a single tree, made on the device from kd_bbox_kernel's partial boxes: control block, root item -- into the
// first level's tables (mode 0), the phase-A queue (1) or the phase-B list (2) -- and the box itself for the host, which
// reads it back with the control block when the build is over.  (Until round 3 the host fetched the box, built the root
// and uploaded four small buffers: a stream wait and five copies at the head of every build.)
struct RootInit {
  const float *part;   // [used][6] partial boxes
  int32_t used;
  int32_t mode;        // 0: first level of phase 0, 1: phase-A queue, 2: phase-B list
  float *bbox_out;     // [6]
  BuildItem *lv_item;  // mode 0: items[0], stat[0], the levels' small counters
  LvStat *lv_stat;
  int32_t *lv_small;
};
__global__ __launch_bounds__(64) void kd_root_kernel(BuildArgs A, RootInit R) {
  const int lane = threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int k = lane; k < R.used; k += 64)
    for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], R.part[k * 6 + d]); mx[d] = fmaxf(mx[d], R.part[k * 6 + 3 + d]); }
  float lo[3], hi[3];
  for (int d = 0; d < 3; ++d) { lo[d] = wave_min(mn[d]); hi[d] = wave_max(mx[d]); }
  if (lane != 0) return;
  BuildCtl c{};
  c.q_head = 0;
  c.q_tail_reserved = R.mode == 1 ? 1 : 0;
  c.q_pending = R.mode == 1 ? 1 : 0;
  c.next_group = 1;  // group 0 holds the root
  c.n_sub = R.mode == 2 ? 1 : 0;
  *A.ctl = c;
  BuildItem root{};
  root.l = 0;
  root.r = A.n;
  for (int d = 0; d < 3; ++d) { root.lo[d] = lo[d]; root.hi[d] = hi[d]; R.bbox_out[d] = lo[d]; R.bbox_out[3 + d] = hi[d]; }
  root.slot = 0;
  root.heap = 0;
  root.parent_word = -1;
  root.depth = 1;
  if (R.mode == 0) {
    *R.lv_item = root;
    LvStat st;
    for (int d = 0; d < 3; ++d) { st.mn[d] = ord_i(lo[d]); st.mx[d] = ord_i(hi[d]); }  // a root's extrema are its box
    lv_stat_reset(st);
    *R.lv_stat = st;
    R.lv_small[0] = 1;
    for (int k = 1; k < 8; ++k) R.lv_small[k] = 0;
  } else if (R.mode == 1) {
    A.queue[0] = root;
    A.q_ready[0] = 1;
  } else {
    A.sublist[0] = root;
  }
}

// FOREST_SLICES workgroups per root: bounding box of a slice of its point range (computeBoundingBox, :1406-1427); the
// slices' boxes are merged where they are used (kd_forest_box)
constexpr int FOREST_SLICES = 8;
__global__ __launch_bounds__(256) void kd_bbox_seg_kernel(const float4 *pts, const int32_t *roots_lr, float *out) {
  __shared__ float smin[3][4], smax[3][4];
  const int l0 = roots_lr[2 * blockIdx.x], r0 = roots_lr[2 * blockIdx.x + 1];
  const int per = (r0 - l0 + FOREST_SLICES - 1) / FOREST_SLICES;
  const int l = l0 + (int)blockIdx.y * per, r = min(r0, l + per);
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = l + threadIdx.x; i < r; i += 256) {
    const float4 p = pts[i];
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int d = 0; d < 3; ++d) {
    const float a = wave_min(mn[d]), b = wave_max(mx[d]);
    if (lane == 0) { smin[d][wave] = a; smax[d][wave] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    float a = smin[d][0], b = smax[d][0];
    for (int w = 1; w < 4; ++w) { a = fminf(a, smin[d][w]); b = fmaxf(b, smax[d][w]); }
    out[(blockIdx.x * FOREST_SLICES + blockIdx.y) * 6 + d] = a;
    out[(blockIdx.x * FOREST_SLICES + blockIdx.y) * 6 + 3 + d] = b;
  }
}
__device__ __forceinline__ void kd_forest_box(const float *part, int t, float (&lo)[3], float (&hi)[3]) {
  for (int d = 0; d < 3; ++d) { lo[d] = FLT_MAX; hi[d] = -FLT_MAX; }
  for (int k = 0; k < FOREST_SLICES; ++k)
    for (int d = 0; d < 3; ++d) {
      lo[d] = fminf(lo[d], part[(t * FOREST_SLICES + k) * 6 + d]);
      hi[d] = fmaxf(hi[d], part[(t * FOREST_SLICES + k) * 6 + 3 + d]);
    }
}
// The roots of a forest are made on the host WITHOUT their boxes (which would cost a round trip at the head of the build):
// root item j of tree t = -1 - parent_word gets its box -- and, as a level item, its extrema: a root's bounding box IS its
// extrema -- here; threads [n_items, n_items + T) write the trees' boxes for the views (read back at the end of the build).
__global__ __launch_bounds__(256) void kd_forest_patch_kernel(const float *part, BuildItem *items, LvStat *stat, int n_items, float *bb, int T) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  float lo[3], hi[3];
  if (j < n_items) {
    const int t = -1 - items[j].parent_word;
    kd_forest_box(part, t, lo, hi);
    for (int d = 0; d < 3; ++d) { items[j].lo[d] = lo[d]; items[j].hi[d] = hi[d]; }
    if (stat) {
      LvStat st;
      for (int d = 0; d < 3; ++d) { st.mn[d] = ord_i(lo[d]); st.mx[d] = ord_i(hi[d]); }
      lv_stat_reset(st);
      for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) { st.cmn[c][d] = 0; st.cmx[c][d] = 0; }
      stat[j] = st;
    }
  } else if (bb && j < n_items + T) {
    const int t = j - n_items;
    kd_forest_box(part, t, lo, hi);
    for (int d = 0; d < 3; ++d) { bb[t * 6 + d] = lo[d]; bb[t * 6 + 3 + d] = hi[d]; }
  }
}

// Packet-search nodes: per used slot the child references and the children's tight boxes -- an inner
// child's box is the extrema its own split computed (own_box), a leaf child's box comes from its <= 10 points.
__global__ __launch_bounds__(256) void kd_pnode_kernel(const KdNode *nodes, int n_slots, const float4 *pts,
                                                       const float *own_box, PNode *pn) {
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= n_slots) return;
  const KdNode nd = nodes[slot];
  PNode o;
  o.c1 = nd.c1;
  o.c2 = nd.c2;
  o.pad[0] = o.pad[1] = 0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const uint32_t ref = c ? nd.c2 : nd.c1;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if ((nd.c1 | nd.c2) == 0) {  // unused slot
    } else if (ref & KD_LEAF) {
      const int l = (int)((ref & ~KD_LEAF) >> 4), cnt = (int)(ref & 15u);
      for (int k = 0; k < cnt; ++k) {
        const float4 p = pts[l + k];
        mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
        mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
        mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
      }
    } else {
      const float *b = own_box + (size_t)(ref >> 2) * 6;
      mn[0] = b[0]; mn[1] = b[1]; mn[2] = b[2]; mx[0] = b[3]; mx[1] = b[4]; mx[2] = b[5];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { o.box[c][d] = mn[d]; o.box[c][3 + d] = mx[d]; }
  }
  pn[slot] = o;
}

}  // namespace

// Build the tree of `n` points at d_pts (float4 {x,y,z,bitcast(original index)}, permuted in
// place).  d_nodes must hold node_cap nodes.  Returns hipSuccess and fills `view`/depth, or
// sets *fallback when a structure limit was hit (the caller retries with more node slots or fails).
namespace {
// Scratch of a build, kept per stream between builds (hipMalloc/hipFree of tens of MB per call cost
// more than a millisecond); released by treebuild_release_scratch.
struct BuildPool {
  void *blob = nullptr, *lv = nullptr, *part = nullptr;
  size_t cap = 0, lv_cap = 0;
};
std::mutex g_pool_mu;
std::map<hipStream_t, BuildPool> g_pool;

hipError_t pool_get(hipStream_t s, bool lv, size_t bytes, void **out) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  BuildPool &p = g_pool[s];
  void *&ptr = lv ? p.lv : p.blob;
  size_t &cap = lv ? p.lv_cap : p.cap;
  if (bytes > cap) {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 4;
    hipError_t e = hipMalloc(&ptr, want);
    if (e != hipSuccess) return e;
    cap = want;
  }
  *out = ptr;
  return hipSuccess;
}
}  // namespace

void treebuild_release_scratch(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_pool.find(s);
  if (it == g_pool.end()) return;
  if (it->second.blob) (void)hipFree(it->second.blob);
  if (it->second.lv) (void)hipFree(it->second.lv);
  if (it->second.part) (void)hipFree(it->second.part);
  g_pool.erase(it);
}

namespace {
// Phase 0 driver: processes `level` (every entry more than HUGE_MIN points) and the levels it
// spawns; smaller children land in A.queue / A.sublist.  n = points the trees span in total.
// n: points of all roots together (capacities); n_largest: of the largest root (how many levels the first batch enqueues)
// hipMemcpyAsync from / into LOCALS of the builders below (item lists, control blocks, read-back buffers): on every normal path
// a builder waits for the stream before it returns -- its read-backs need that anyway.  Its ERROR returns do the same through
// this macro, so that no copy is ever in flight from or into a frame that has been left.  (The tree build is off the per-frame
// path since the trees are deferred: a context-owned pinned staging area would buy nothing here.)
#define TB_RETURN_SETTLED(stream, err)        \
  do {                                        \
    (void)hipStreamSynchronize(stream);       \
    return (err);                             \
  } while (0)

hipError_t run_levels(const BuildArgs &A, std::vector<BuildItem> level, int32_t n, hipStream_t stream, int *fallback, const RootInit *root_init = nullptr,
                      int32_t n_largest = 0, const float *forest_part = nullptr) {
  if (n_largest <= 0) n_largest = n;
  hipError_t e;
  void *lv_blob = nullptr;
  const int n_first = root_init ? 1 : (int)level.size();
  const int cap_nodes = n / A.huge_min * 2 + n_first + 8, cap_chunks = n / LV_CH + cap_nodes + 8;
  const size_t sz_items = (size_t)cap_nodes * sizeof(BuildItem), sz_stat = (size_t)cap_nodes * sizeof(LvStat),
               sz_ci = (size_t)cap_chunks * sizeof(int32_t), sz_ni = (size_t)cap_nodes * sizeof(int32_t);
  if ((e = pool_get(stream, true, 2 * sz_items + 2 * sz_stat + 5 * sz_ci + sz_ni + (size_t)cap_chunks * sizeof(LvChunk) + 128, &lv_blob)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  char *q = static_cast<char *>(lv_blob);
  BuildItem *d_items[2];
  d_items[0] = reinterpret_cast<BuildItem *>(q); q += sz_items;
  d_items[1] = reinterpret_cast<BuildItem *>(q); q += sz_items;
  LvStat *d_stat[2];
  d_stat[0] = reinterpret_cast<LvStat *>(q); q += sz_stat;
  d_stat[1] = reinterpret_cast<LvStat *>(q); q += sz_stat;
  int32_t *d_chunk_node = reinterpret_cast<int32_t *>(q); q += sz_ci;
  int32_t *d_cntL = reinterpret_cast<int32_t *>(q); q += sz_ci;
  int32_t *d_cntR = reinterpret_cast<int32_t *>(q); q += sz_ci;
  int32_t *d_baseL = reinterpret_cast<int32_t *>(q); q += sz_ci;
  int32_t *d_baseR = reinterpret_cast<int32_t *>(q); q += sz_ci;
  int32_t *d_chunk_first = reinterpret_cast<int32_t *>(q); q += sz_ni;
  q = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(q) + 31) & ~(uintptr_t)31);
  LvChunk *d_rec = reinterpret_cast<LvChunk *>(q); q += (size_t)cap_chunks * sizeof(LvChunk);
  int32_t *d_small = reinterpret_cast<int32_t *>(q);  // [0],[1] item counts (ping-pong), [2],[3] level header, [4] lv_final's tickets
  if (n_first > cap_nodes) { *fallback = 3; return hipSuccess; }
  const int32_t init[8] = {n_first, 0, 0, 0, 0, 0, 0, 0};
  if (root_init) {  // a single tree: the root is made on the device from its box (no host round trip)
    RootInit R = *root_init;
    R.mode = 0;
    R.lv_item = d_items[0];
    R.lv_stat = d_stat[0];
    R.lv_small = d_small;
    hipLaunchKernelGGL(kd_root_kernel, dim3(1), dim3(64), 0, stream, A, R);
  }
  // a root's extrema are its bounding box (kd_bbox_kernel / kd_bbox_seg_kernel computed exactly that)
  std::vector<LvStat> stat0(root_init ? 0 : (size_t)n_first);
  if (!root_init) {
  for (int j = 0; j < n_first; ++j) {
    for (int d = 0; d < 3; ++d) {
      const int32_t a = __builtin_bit_cast(int32_t, level[(size_t)j].lo[d]), b = __builtin_bit_cast(int32_t, level[(size_t)j].hi[d]);
      stat0[(size_t)j].mn[d] = a >= 0 ? a : a ^ 0x7FFFFFFF;
      stat0[(size_t)j].mx[d] = b >= 0 ? b : b ^ 0x7FFFFFFF;
    }
    lv_stat_reset(stat0[(size_t)j]);
  }
  // (sources are locals: every return below this point -- TB_RETURN_SETTLED on the error paths, the read-back's wait on the normal one -- leaves the stream settled)
  if ((e = hipMemcpyAsync(d_items[0], level.data(), (size_t)n_first * sizeof(BuildItem), hipMemcpyHostToDevice, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if ((e = hipMemcpyAsync(d_stat[0], stat0.data(), (size_t)n_first * sizeof(LvStat), hipMemcpyHostToDevice, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if ((e = hipMemcpyAsync(d_small, init, sizeof(init), hipMemcpyHostToDevice, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  // a forest's roots arrive without their boxes: filled in from the per-slice boxes, with the extrema (kd_forest_patch_kernel)
  if (forest_part)
    hipLaunchKernelGGL(kd_forest_patch_kernel, dim3((n_first + 255) / 256), dim3(256), 0, stream, forest_part, d_items[0], d_stat[0], n_first,
                       (float *)nullptr, 0);
  }
  // Levels are enqueued in batches without looking at their outcome (grids at capacity, exhausted
  // levels cost a handful of empty launches); the host checks the item count once per batch.
  const dim3 gc(cap_chunks), gn((cap_nodes + 63) / 64), bt(LV_TB);
  int lvl = 0;
  const bool dbg = env_once().debug;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int batch = 0; batch < 16; ++batch) {
    const double t_enq0 = now_us();
    // nanoflann splits at the middle of the box, not at the median: the trees of a voxel map need ~4 levels more than a
    // balanced one would (13 for the 587 k-point surface cloud) -- an exhausted level costs seven empty launches (~25 us), a
    // second batch a host round trip on top of its levels
    // (a forest's levels are as many as its LARGEST tree needs: sized by all its points together, the 22 cube trees of a
    // mapping frame were given 13 levels and used 7 -- six times seven empty launches, 0.2 ms)
    const int per_batch = batch == 0 ? std::max(2, (int)std::ceil(std::log2(std::max(2.0, (double)n_largest / A.huge_min))) + 4) : 3;
    for (int k = 0; k < per_batch; ++k, ++lvl) {
      LvArgs L{};
      L.A = A;
      L.items = d_items[lvl & 1];
      L.next_items = d_items[(lvl + 1) & 1];
      L.stat = d_stat[lvl & 1];
      L.stat_next = d_stat[(lvl + 1) & 1];
      L.final_done = d_small + 4;
      L.rec = d_rec;
      L.chunk_node = d_chunk_node;
      L.chunk_first = d_chunk_first;
      L.chunk_node_w = d_chunk_node;
      L.chunk_first_w = d_chunk_first;
      L.cntL = d_cntL; L.cntR = d_cntR; L.baseL = d_baseL; L.baseR = d_baseR;
      L.cur_count = d_small + (lvl & 1);
      L.next_count = d_small + ((lvl + 1) & 1);
      L.hdr = d_small + 2;
      L.cap_nodes = cap_nodes;
      L.cap_chunks = cap_chunks;
      L.next_cap = cap_nodes;
      // seven launches per level (eleven until round 3): the extrema come from the parent's bounds pass, the second Hoare
      // pass is one launch, the next level is set up by lv_final_kernel's last block
      if (lvl == 0) hipLaunchKernelGGL(lv_setup_kernel, dim3(1), dim3(1024), 0, stream, L);
      hipLaunchKernelGGL(lv_count_kernel, gc, bt, 0, stream, L);
      hipLaunchKernelGGL(lv_hflags_kernel, gc, bt, 0, stream, L, 0);
      hipLaunchKernelGGL(lv_hwrite_kernel, gc, bt, 0, stream, L, 0);
      hipLaunchKernelGGL(lv_hswap_kernel, gc, bt, 0, stream, L, 0);
      hipLaunchKernelGGL(lv_pass2_kernel, dim3(cap_nodes), bt, 0, stream, L);
      hipLaunchKernelGGL(lv_bounds_kernel, gc, bt, 0, stream, L);
      hipLaunchKernelGGL(lv_final_kernel, gn, dim3(64), 0, stream, L);
    }
    int32_t remaining = 0;
    const double t_enq1 = now_us();
    if ((e = hipMemcpyAsync(&remaining, d_small + (lvl & 1), 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
    if (dbg) fprintf(stderr, "[lslam] levels batch %d: %d levels enqueued in %.0f us (host), device done %.0f us later\n", batch, per_batch,
                     t_enq1 - t_enq0, now_us() - t_enq1);
    if (remaining == 0) return hipSuccess;
  }
  *fallback = 3;  // 50+ levels above the wavefront-local size: deeper than the traversal stack allows anyway
  return hipSuccess;
}
}  // namespace

static bool tiny_phase_enabled() {
  const bool off = env_once().tiny_phase_off;  // A/B switch
  return !off;
}
static int32_t reg_nodes_enabled() {
  const bool off = env_once().no_reg_nodes;  // A/B switch
  return off ? 0 : 1;
}

hipError_t build_kdtree_device(float4 *d_pts, int32_t n, KdNode *d_nodes, float *d_own_box, int32_t node_cap,
                               hipStream_t stream, TreeView *view, int *depth, size_t *n_leaves,
                               int *fallback) {
  const bool dbg = env_once().debug;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double T0 = now();
  *fallback = 0;
  view->nodes = d_nodes;
  view->pn = nullptr;  // the packet search's nodes are made from d_own_box when somebody asks for that search (build_packet_nodes)
  view->pts = d_pts;
  view->n_pts = n;
  view->n_nodes = 0;
  view->root_ref = KD_LEAF;
  for (int d = 0; d < 3; ++d) view->bb_lo[d] = view->bb_hi[d] = 0.f;
  *depth = 0;
  *n_leaves = 0;
  if (n == 0) return hipSuccess;
  hipError_t e;
  // bounding box
  constexpr int NB = 256;
  float *d_part = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    BuildPool &bp = g_pool[stream];
    if (!bp.part && (e = hipMalloc(&bp.part, NB * 6 * sizeof(float))) != hipSuccess) TB_RETURN_SETTLED(stream, e);
    d_part = static_cast<float *>(bp.part);
  }
  hipLaunchKernelGGL(kd_bbox_kernel, dim3(NB), dim3(256), 0, stream, d_pts, n, d_part);
  const int used = std::min(NB, (n + 255) / 256);
  if (n <= 10) {  // the root is a leaf: only the box is needed
    float h_part[NB * 6];
    if ((e = hipMemcpyAsync(h_part, d_part, sizeof(float) * 6 * (size_t)used, hipMemcpyDeviceToHost, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
    for (int d = 0; d < 3; ++d) {
      float a = h_part[d], b = h_part[3 + d];
      for (int k = 1; k < used; ++k) { a = std::min(a, h_part[k * 6 + d]); b = std::max(b, h_part[k * 6 + 3 + d]); }
      view->bb_lo[d] = a;
      view->bb_hi[d] = b;
    }
  }
  const double T1 = now();
  if (n <= 10) {  // the root is a leaf
    view->root_ref = KD_LEAF | (uint32_t)n;
    *depth = 1;
    *n_leaves = 1;
    return hipSuccess;
  }
  // queue entries are nodes above LOCAL_MAX points: at most n / LOCAL_MAX of them on one level and the
  // device traversal stack refuses trees deeper than KD_STACK_MAX levels, so this never overflows for a
  // tree the search could use; subtree roots hold more than 10 points each (n / 8 is generous)
  const int32_t queue_cap = KD_STACK_MAX * (n / LOCAL_MAX + 1) + 64;
  const int32_t sub_cap = std::max(64, 4 * (n / LOCAL_MAX + 16) + n / 8);
  BuildArgs A{};
  A.pts = d_pts;
  A.nodes = d_nodes;
  A.node_cap = node_cap & ~7;
  A.queue_cap = queue_cap;
  {
    const char *hm = debug_env("LSLAM_HUGE_MIN");  // A/B switch and test hook (LSLAM_DEBUG_HOOKS=1), read per build
    const int env_huge = hm ? std::atoi(hm) : 0;
    A.huge_min = std::min(HUGE_MIN_MAX, std::max(LOCAL_MAX, env_huge > 0 ? env_huge : HUGE_MIN_TREE));
  }
  A.n = n;
  A.reg_nodes = reg_nodes_enabled();
  A.spin_limit = 1u << 22;
  if (const char *sl = debug_env("LSLAM_DEBUG_SPIN_LIMIT")) A.spin_limit = (uint32_t)strtoul(sl, nullptr, 10);  // tests
  void *blob = nullptr;
  A.sub_cap = sub_cap;
  const size_t sz_queue = (size_t)queue_cap * sizeof(BuildItem), sz_ready = (size_t)queue_cap * sizeof(int32_t),
               sz_tmp = (size_t)n * sizeof(int32_t), sz_ctl = 256, sz_sub = (size_t)sub_cap * sizeof(BuildItem);
  const bool tiny_phase = tiny_phase_enabled() && A.reg_nodes;
  const int32_t tiny_cap = tiny_phase ? n / 11 + 2 : 0;
  const size_t sz_tiny = tiny_phase ? ((size_t)tiny_cap * sizeof(BuildItem) + TINY_ACC * 128 + 127) & ~(size_t)127 : 0;
  if ((e = pool_get(stream, false, sz_queue + sz_sub + sz_ready + 2 * sz_tmp + sz_ctl + sz_tiny + 128, &blob)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  char *p = static_cast<char *>(blob);
  A.queue = reinterpret_cast<BuildItem *>(p); p += sz_queue;
  A.sublist = reinterpret_cast<BuildItem *>(p); p += sz_sub;
  A.q_ready = reinterpret_cast<int32_t *>(p); p += sz_ready;
  A.tmpA = reinterpret_cast<int32_t *>(p); p += sz_tmp;
  A.tmpB = reinterpret_cast<int32_t *>(p); p += sz_tmp;
  A.ctl = reinterpret_cast<BuildCtl *>(p); p += sz_ctl;
  p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + 127) & ~(uintptr_t)127);
  A.tiny_acc = tiny_phase ? reinterpret_cast<int32_t *>(p) : nullptr;
  A.tiny_list = tiny_phase ? reinterpret_cast<BuildItem *>(p + TINY_ACC * 128) : nullptr;
  A.tiny_cap = tiny_cap;
  if (tiny_phase && (e = hipMemsetAsync(p, 0, sz_tiny, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  A.own_box = d_own_box;
  A.root_feat = &A.ctl->root_feat;
  if ((e = hipMemsetAsync(A.q_ready, 0, sz_ready, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if ((e = hipMemsetAsync(d_nodes, 0, (size_t)A.node_cap * sizeof(KdNode), stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  BuildCtl ctl{};
  const bool root_small = n <= LOCAL_MAX;
  const bool no_levels = env_once().no_level_build;  // A/B switch
  // Below ~50 k points the launches of the level phase cost more host time than the persistent phase-A
  // kernel (one workgroup per node, one launch) costs device time: 0.49 against 0.62 ms at 16 k
  // points, equal at 64 k, 4.3 against 1.6 ms at 512 k (tools/tree_size_sweep.py).
  constexpr int LEVELS_MIN_POINTS = 49152;
  const bool root_huge = n > A.huge_min && n > LEVELS_MIN_POINTS && !no_levels;
  // control block, root item and the root's box are made on the device (kd_root_kernel): the box comes back with the control
  // block at the end of the build
  RootInit R{};
  R.part = d_part;
  R.used = used;
  R.bbox_out = reinterpret_cast<float *>(reinterpret_cast<char *>(A.ctl) + 128);
  if (root_huge) {
    // ---- phase 0: level-synchronous processing of the nodes with more than HUGE_MIN points ----
    if ((e = run_levels(A, std::vector<BuildItem>(), n, stream, fallback, &R)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
    if (*fallback) return hipSuccess;
  } else {
    R.mode = root_small ? 2 : 1;  // the whole tree is one phase-B subtree / the root enters the phase-A queue
    hipLaunchKernelGGL(kd_root_kernel, dim3(1), dim3(64), 0, stream, A, R);
  }
  // persistent grid: every workgroup must be resident (they wait on each other's output)
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (dbg && (e = hipStreamSynchronize(stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);  // only to split the timing below
  const double T2 = now();
  // after the levels: what they left between LOCAL_MAX and HUGE_MIN, one workgroup per node; without levels (a root below
  // LEVELS_MIN_POINTS): the root enters phase A's queue
  if (root_huge) {
    if (A.huge_min > LOCAL_MAX) hipLaunchKernelGGL(kd_build_medium_kernel, dim3(cus), dim3(TB_BIG), 0, stream, A);
  } else if (!root_small) {
    hipLaunchKernelGGL(kd_build_big_kernel, dim3(cus), dim3(TB_BIG), 0, stream, A);
  }
  hipLaunchKernelGGL(kd_build_small_kernel, dim3(cus * 16), dim3(TB_SMALL), 0, stream, A);
  if (tiny_phase) hipLaunchKernelGGL(kd_build_tiny_kernel, dim3(std::min(cus * 64, (tiny_cap + TINY_SLOTS - 1) / TINY_SLOTS)), dim3(64), 0, stream, A);
  if ((e = hipGetLastError()) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  char ctl_blob[256];  // [BuildCtl | ... | box at byte 128]
  if ((e = hipMemcpyAsync(ctl_blob, A.ctl, sizeof(ctl_blob), hipMemcpyDeviceToHost, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  int32_t acc[TINY_ACC * 32];
  if (tiny_phase && (e = hipMemcpyAsync(acc, A.tiny_acc, sizeof(acc), hipMemcpyDeviceToHost, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  std::memcpy(&ctl, ctl_blob, sizeof(ctl));
  {
    float bb[6];
    std::memcpy(bb, ctl_blob + 128, sizeof(bb));
    for (int d = 0; d < 3; ++d) { view->bb_lo[d] = bb[d]; view->bb_hi[d] = bb[3 + d]; }
  }
  if (tiny_phase)
    for (int k = 0; k < TINY_ACC; ++k) { ctl.n_leaves += acc[k * 32]; ctl.max_depth = std::max(ctl.max_depth, acc[k * 32 + 1]); }
  const double T3 = now();
  if (dbg)
    fprintf(stderr, "[lslam] tree build n=%d: bbox %.2f ms, setup %.2f ms, build kernel %.2f ms, free %.2f ms (overflow %d, groups %d)\n",
            n, T1 - T0, T2 - T1, T3 - T2, now() - T3, ctl.overflow, ctl.next_group);
  if (ctl.overflow) {
    *fallback = ctl.overflow;
    return hipSuccess;
  }
  view->n_nodes = std::min(ctl.next_group * 8, A.node_cap);  // groups are taken eight at a time: the last few may be unused
  view->root_ref = (0u << 2) | (uint32_t)ctl.root_feat;
  *depth = ctl.max_depth;
  *n_leaves = (size_t)ctl.n_leaves;
  return hipSuccess;
}

// The packet search's nodes (lslam_packet.hpp) of a built tree, from the own boxes its build recorded: made on demand -- the
// packet search is an explicit choice (it measured slower than the lane search) and a map rebuilt every frame
// should not pay 64 B per node and a launch for it.
hipError_t build_packet_nodes(const TreeView &view, const float *d_own_box, PNode *d_pn, hipStream_t stream) {
  if (view.n_nodes <= 0) return hipSuccess;
  hipLaunchKernelGGL(kd_pnode_kernel, dim3((view.n_nodes + 255) / 256), dim3(256), 0, stream, view.nodes, view.n_nodes, view.pts,
                     d_own_box, d_pn);
  return hipGetLastError();
}

// Many trees at once (variant C: one tree per map cube): `roots_lr` holds T point ranges of d_pts
// (host array), all trees share d_nodes; tree t's root sits in node group t.  Phase 0 takes every
// root above HUGE_MIN points, phase B the rest; roots of at most 10 points are leaves.  views[t] is
// filled for every root (nodes = d_nodes, pts = d_pts: references are absolute).
hipError_t build_kdforest_device(float4 *d_pts, int32_t n_total, const int32_t *roots_lr, int T, KdNode *d_nodes,
                                 PNode *d_pn, int32_t node_cap, hipStream_t stream, TreeView *views, int *max_depth,
                                 size_t *n_leaves, int *fallback) {
  *fallback = 0;
  *max_depth = 0;
  *n_leaves = 0;
  if (T <= 0) return hipSuccess;
  hipError_t e;
  const int32_t sub_cap = std::max(64, 4 * (n_total / LOCAL_MAX + 16) + n_total / 8 + T);
  const int32_t queue_cap = 2 * (n_total / LOCAL_MAX + 1) + T + 64;  // medium phase: roots and level-phase nodes between LOCAL_MAX and HUGE_MIN
  BuildArgs A{};
  A.pts = d_pts;
  A.nodes = d_nodes;
  A.node_cap = node_cap & ~7;
  A.queue_cap = queue_cap;
  A.huge_min = LOCAL_MAX;  // the forest's levels run to the end (measured: see LSLAM_HUGE_MIN above); roots in between take the medium phase
  A.sub_cap = sub_cap;
  A.n = n_total;
  A.reg_nodes = reg_nodes_enabled();
  const size_t sz_queue = (size_t)queue_cap * sizeof(BuildItem), sz_ready = (size_t)queue_cap * sizeof(int32_t),
               sz_tmp = (size_t)std::max(n_total, 1) * sizeof(int32_t), sz_ctl = 256, sz_sub = (size_t)sub_cap * sizeof(BuildItem),
               sz_rf = ((size_t)T * 4 + 15) & ~(size_t)15, sz_lr = ((size_t)T * 8 + 15) & ~(size_t)15,
               sz_bb = ((size_t)T * 24 * (1 + FOREST_SLICES) + 15) & ~(size_t)15, sz_own = d_pn ? (size_t)A.node_cap * 6 * sizeof(float) : 0;
  const bool tiny_phase = tiny_phase_enabled() && A.reg_nodes;
  const int32_t tiny_cap = tiny_phase ? n_total / 11 + 2 : 0;
  const size_t sz_tiny = tiny_phase ? ((size_t)tiny_cap * sizeof(BuildItem) + TINY_ACC * 128 + 127) & ~(size_t)127 : 0;
  void *blob = nullptr;
  if ((e = pool_get(stream, false, sz_queue + sz_sub + sz_ready + 2 * sz_tmp + sz_ctl + sz_rf + sz_lr + sz_bb + sz_tiny + sz_own + 128, &blob)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  char *p = static_cast<char *>(blob);
  A.queue = reinterpret_cast<BuildItem *>(p); p += sz_queue;
  A.sublist = reinterpret_cast<BuildItem *>(p); p += sz_sub;
  A.q_ready = reinterpret_cast<int32_t *>(p); p += sz_ready;
  A.tmpA = reinterpret_cast<int32_t *>(p); p += sz_tmp;
  A.tmpB = reinterpret_cast<int32_t *>(p); p += sz_tmp;
  A.ctl = reinterpret_cast<BuildCtl *>(p); p += sz_ctl;
  A.root_feat = reinterpret_cast<int32_t *>(p); p += sz_rf;
  int32_t *d_lr = reinterpret_cast<int32_t *>(p); p += sz_lr;
  float *d_bb = reinterpret_cast<float *>(p);  // [T][6] the roots' boxes, then [T][FOREST_SLICES][6] their slices'
  float *d_part = d_bb + (size_t)T * 6;
  p += sz_bb;
  p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + 127) & ~(uintptr_t)127);
  A.tiny_acc = tiny_phase ? reinterpret_cast<int32_t *>(p) : nullptr;
  A.tiny_list = tiny_phase ? reinterpret_cast<BuildItem *>(p + TINY_ACC * 128) : nullptr;
  A.tiny_cap = tiny_cap;
  if (tiny_phase && (e = hipMemsetAsync(p, 0, sz_tiny, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  p += sz_tiny;
  A.own_box = d_pn ? reinterpret_cast<float *>(p) : nullptr;
  A.spin_limit = 1u << 22;
  if ((e = hipMemcpyAsync(d_lr, roots_lr, (size_t)T * 8, hipMemcpyHostToDevice, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  // per-slice boxes of every root; nobody waits for them: the root items are made below without their boxes and patched
  // on the device, the views get theirs with the read-back at the end of the build
  hipLaunchKernelGGL(kd_bbox_seg_kernel, dim3(T, FOREST_SLICES), dim3(256), 0, stream, d_pts, d_lr, d_part);
  if ((e = hipMemsetAsync(d_nodes, 0, (size_t)A.node_cap * sizeof(KdNode), stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if ((e = hipMemsetAsync(A.root_feat, 0, sz_rf, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  std::vector<BuildItem> level, medium, small;
  std::vector<int32_t> slot_of(T, -1);
  int groups = 0, leaves = 0;
  for (int t = 0; t < T; ++t) {
    const int l = roots_lr[2 * t], r = roots_lr[2 * t + 1], n = r - l;
    TreeView &v = views[t];
    v.nodes = d_nodes;
    v.pn = d_pn;
    v.pts = d_pts;
    v.n_pts = n;
    v.n_nodes = 0;
    for (int d = 0; d < 3; ++d) { v.bb_lo[d] = 0.f; v.bb_hi[d] = 0.f; }  // (from the device at the end)
    if (n <= 10) {  // the root is a leaf (nanoflann.hpp:936-951)
      v.root_ref = KD_LEAF | ((uint32_t)l << 4) | (uint32_t)std::max(n, 0);
      if (n > 0) { ++leaves; *max_depth = std::max(*max_depth, 1); }
      continue;
    }
    BuildItem it{};
    it.l = l;
    it.r = r;
    for (int d = 0; d < 3; ++d) { it.lo[d] = 0.f; it.hi[d] = 0.f; }  // kd_forest_patch_kernel
    it.slot = groups * 8;
    it.heap = 0;
    it.parent_word = -1 - t;
    it.depth = 1;
    slot_of[t] = it.slot;
    ++groups;
    (n > A.huge_min ? level : (n > LOCAL_MAX ? medium : small)).push_back(it);
  }
  if ((groups + 1) * 8 > A.node_cap || (int32_t)small.size() > sub_cap || (int32_t)medium.size() > queue_cap) {
    *fallback = 1;
    return hipSuccess;
  }
  BuildCtl ctl{};
  ctl.next_group = groups;
  ctl.n_sub = (int32_t)small.size();
  ctl.q_tail_reserved = (int32_t)medium.size();
  ctl.n_leaves = leaves;
  if ((e = hipMemcpyAsync(A.ctl, &ctl, sizeof(ctl), hipMemcpyHostToDevice, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  // (sources are locals: every return below this point -- TB_RETURN_SETTLED on the error paths, the read-back's wait on the normal one -- leaves the stream settled)
  if (!small.empty() &&
      (e = hipMemcpyAsync(A.sublist, small.data(), small.size() * sizeof(BuildItem), hipMemcpyHostToDevice, stream)) != hipSuccess)
    TB_RETURN_SETTLED(stream, e);
  if (!medium.empty() &&
      (e = hipMemcpyAsync(A.queue, medium.data(), medium.size() * sizeof(BuildItem), hipMemcpyHostToDevice, stream)) != hipSuccess)
    TB_RETURN_SETTLED(stream, e);
  // the wavefront-local roots' boxes, and every tree's box for its view; the medium roots' boxes
  hipLaunchKernelGGL(kd_forest_patch_kernel, dim3(((int)small.size() + T + 255) / 256), dim3(256), 0, stream, d_part, A.sublist, (LvStat *)nullptr,
                     (int)small.size(), d_bb, T);
  if (!medium.empty())
    hipLaunchKernelGGL(kd_forest_patch_kernel, dim3(((int)medium.size() + 255) / 256), dim3(256), 0, stream, d_part, A.queue, (LvStat *)nullptr,
                       (int)medium.size(), (float *)nullptr, 0);
  if (!level.empty()) {
    int32_t n_largest = 0;
    for (const BuildItem &it : level) n_largest = std::max(n_largest, it.r - it.l);
    if ((e = run_levels(A, level, n_total, stream, fallback, nullptr, n_largest, d_part)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
    if (*fallback) return hipSuccess;
  }
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (A.huge_min > LOCAL_MAX) hipLaunchKernelGGL(kd_build_medium_kernel, dim3(cus), dim3(TB_BIG), 0, stream, A);
  hipLaunchKernelGGL(kd_build_small_kernel, dim3(cus * 16), dim3(TB_SMALL), 0, stream, A);
  if (tiny_phase) hipLaunchKernelGGL(kd_build_tiny_kernel, dim3(std::min(cus * 64, (tiny_cap + TINY_SLOTS - 1) / TINY_SLOTS)), dim3(64), 0, stream, A);
  if ((e = hipGetLastError()) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  std::vector<int32_t> rf(T);
  std::vector<float> bb((size_t)T * 6);
  if ((e = hipMemcpyAsync(&ctl, A.ctl, sizeof(ctl), hipMemcpyDeviceToHost, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if ((e = hipMemcpyAsync(rf.data(), A.root_feat, (size_t)T * 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if ((e = hipMemcpyAsync(bb.data(), d_bb, bb.size() * 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  int32_t acc[TINY_ACC * 32];
  if (tiny_phase && (e = hipMemcpyAsync(acc, A.tiny_acc, sizeof(acc), hipMemcpyDeviceToHost, stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);
  if (tiny_phase)
    for (int k = 0; k < TINY_ACC; ++k) { ctl.n_leaves += acc[k * 32]; ctl.max_depth = std::max(ctl.max_depth, acc[k * 32 + 1]); }
  if (ctl.overflow) {
    *fallback = ctl.overflow;
    return hipSuccess;
  }
  const int32_t n_nodes = std::min(ctl.next_group * 8, A.node_cap);
  if (d_pn) {
    hipLaunchKernelGGL(kd_pnode_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, stream, d_nodes, n_nodes, d_pts, A.own_box, d_pn);
    if ((e = hipGetLastError()) != hipSuccess) TB_RETURN_SETTLED(stream, e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) TB_RETURN_SETTLED(stream, e);  // the scratch holding own_box is reused by the next build
  }
  for (int t = 0; t < T; ++t) {
    views[t].n_nodes = n_nodes;
    if (slot_of[t] >= 0) views[t].root_ref = ((uint32_t)slot_of[t] << 2) | (uint32_t)rf[t];
    if (views[t].n_pts > 0)
      for (int d = 0; d < 3; ++d) { views[t].bb_lo[d] = bb[(size_t)t * 6 + d]; views[t].bb_hi[d] = bb[(size_t)t * 6 + 3 + d]; }
  }
  *max_depth = std::max(*max_depth, ctl.max_depth);
  *n_leaves = (size_t)ctl.n_leaves;
  return hipSuccess;
}

}  // namespace lslam
