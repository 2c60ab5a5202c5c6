// lslam_odom.hip -- the scan-to-scan odometry node (variant B) resident on the device.
//
//   LaserOdometry::process     odometry/LaserOdometry.cpp:288-326
//   LaserOdometry::scanMatch   :328-647 (the Gauss-Newton loop; its residual pass and its solve are odom_sweep_kernel /
//                              solve_kernel of lslam_kernels.hip, unchanged)
//   transformToEnd             :156-168
//
// What this file adds is the shape of the node, not new arithmetic:
//   * a sweep's four feature clouds stay in HBM (lslam_fset) from the extraction kernels to this node;
//   * the last clouds stay in HBM from one sweep to the next, moved to the sweep end by the launch that also starts their
//     search structure;
//   * that structure is two HASHED CELL GRIDS per cloud instead of a kd-tree (1 m cells, then 5.02 m cells): counted, scanned
//     and scattered in three launches -- no bounding box, no host round trip, 1 MB of tables;
//   * nearestKSearch(pointSel, 1) of :359 / :425 is answered by one wavefront per query: the points of the 27 cells around
//     the query, 64 candidates per round, the reference's fp32 distance (nanoflann.hpp:364-372), and the proof of
//     lslam_grid.hpp restated for k = 1 -- the smallest distance is nanoflann's answer when it is below the probe's
//     guaranteed radius and unique.  The coarse level's probe reaches farther than the 5 m gate of :363 / :429, so it decides
//     every query the fine level cannot; the same wavefront then walks the ring window (:366-403 / :430-477).
// An exact distance tie between two different points -- which only nanoflann's visit order decides -- raises a flag; the call
// is then redone through kd-trees (odometry_match_trees, the launch-per-step implementation of earlier rounds).
#include "../../include/lslam_c.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "lslam_internal.hpp"
#include "lslam_odom_dev.hpp"
#include "lslam_solve_dev.hpp"

using namespace lslam;

namespace lslam {
// lslam_api.hip
int odometry_match_trees(lslam_ctx *ctx, const void *last_corner, size_t n_lc, const void *last_surf, size_t n_ls, const void *sharp,
                         size_t n_sharp, const void *flat, size_t n_flat, size_t stride_bytes, float pose[6], int32_t max_iterations,
                         float delta_t_abort, float delta_r_abort, lslam_stats *stats);
}  // namespace lslam

namespace {

#define OD_TRY(expr)                                                                                           \
  do {                                                                                                         \
    hipError_t _e = (expr);                                                                                    \
    if (_e != hipSuccess) {                                                                                    \
      char _b[256];                                                                                            \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);     \
      lslam::set_error(_b);                                                                                    \
      return LSLAM_ERR_HIP;                                                                                    \
    }                                                                                                          \
  } while (0)

// ---- hashed cell grids -----------------------------------------------------------------------------------------------------
constexpr int OH_BITS = 16;
constexpr uint32_t OH_SIZE = 1u << OH_BITS;
constexpr float OH_CELL[2] = {1.0f, 5.02f};  // level 1's probe: >= 5.02 (1 - slack) = 5.009 m > the 5 m gate
// Cell coordinate of v: floor(fl(v * inv_c)), the same fp32 operation for cloud points and queries; relative error 2^-24 of a
// value below OH_U_MAX, so two points whose cell coordinates differ by w are at least (w - OH_U_SLACK) cells apart on that axis
constexpr float OH_U_MAX = 8192.0f;
constexpr float OH_U_SLACK = 2.0e-3f;

struct HashView {
  const uint32_t *start;  // [2][OH_SIZE + 1] level, bucket -> first point
  const float4 *pts[2];   // per level: the cloud's points in bucket order, {x, y, z, bitcast(index in the scan-order cloud)}
  int32_t n;
};

LSLAM_DEV int oh_cell(float v, float inv_c) {
  float u = __fmul_rn(v, inv_c);
  u = fminf(fmaxf(u, -4.0f * OH_U_MAX), 4.0f * OH_U_MAX);  // (NaN -> -4 OH_U_MAX: such a point is never anybody's neighbour)
  return (int)floorf(u);
}
LSLAM_DEV uint32_t oh_bucket(int ix, int iy, int iz) {
  return (((uint32_t)ix * 73856093u) ^ ((uint32_t)iy * 19349663u) ^ ((uint32_t)iz * 83492791u)) & (OH_SIZE - 1u);
}

constexpr int OH_RINGS = 256;          // ring ids a sorted point's tag and the ring table hold
constexpr uint32_t OH_IDX_BITS = 24;   // a sorted point carries (index in the scan-order cloud) | (ring << 24)
constexpr uint32_t OH_IDX_MASK = (1u << OH_IDX_BITS) - 1u;

// One atomic per RUN of equal buckets among neighbouring lanes of a wavefront: a cloud comes in scan order, so the points of
// neighbouring lanes are neighbours in space and share their coarse cell (nearly always) and their fine cell (often).  Returns,
// for every lane, what the run's first lane got back from atomicAdd(table + bucket, run length) plus the lane's place in the run:
// a slot of its own in the bucket (the scatter), or just the count done (the counting pass ignores the result).
LSLAM_DEV uint32_t oh_run_add(uint32_t *table, const uint32_t b, const bool on) {
  const int lane = threadIdx.x & 63;
  const unsigned long long alive = __ballot(on);
  const uint32_t prev = (uint32_t)__shfl_up((int)b, 1, 64);
  const bool prev_on = lane > 0 && ((alive >> (lane - 1)) & 1ull);
  const bool head = on && (!prev_on || prev != b);
  const unsigned long long heads = __ballot(head);
  // the run this lane belongs to starts at the highest head at or below it
  const unsigned long long below = heads & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  const int h = below ? 63 - __builtin_clzll(below) : 0;
  uint32_t base = 0;
  if (head) {
    const unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int upto = after ? lane + 1 + __builtin_ctzll(after) : 64;  // the next head, or the end of the wavefront
    const unsigned long long span = (upto == 64 ? ~0ull : ((1ull << upto) - 1ull)) & ~((1ull << lane) - 1ull);
    // (lanes that are off end a run: a run never spans one)
    const unsigned long long dead = ~alive & span;
    const int end = dead ? __builtin_ctzll(dead) : upto;
    base = atomicAdd(table + b, (uint32_t)(end - lane));
  }
  base = (uint32_t)__shfl((int)base, h, 64);
  return base + (uint32_t)(lane - h);
}

struct PrepArgs {
  const float4 *src[2];  // less-sharp, less-flat lists of the sweep (sensor frame at the point's own time)
  int32_t n[2];
  float4 *org[2];        // out: the clouds in scan order, moved to the sweep end when to_end
  uint32_t *cnt;         // [2 clouds][2 levels][OH_SIZE] zero on entry
  uint32_t *hdr;         // [2] per cloud, zero on entry: |= 1 when the cloud is not in ring order or a ring id is not in [0, 255]
  int32_t *ring_start;   // [2][OH_RINGS + 1] per cloud: first index whose ring id is >= r (meaningful for a cloud in ring order)
  const GNState *gate;   // not null: only when the loop has ended (the launches are enqueued before the host knows)
  int32_t to_end;        // 0: the first sweep's clouds are taken as they are (:295-303)
};

// LaserOdometry::transformToEnd (:156-168) for both clouds + the cell counts of their grids.
__global__ __launch_bounds__(256) void odom_prep_kernel(PrepArgs a) {
  if (a.gate && !a.gate->done) return;
  __shared__ float s_pose[6], s_R[9], s_ti[3];
  if (a.to_end && threadIdx.x == 0) {  // the sweep's whole transform and its inverse, once per workgroup
    float pose[6], R[9], t[3], scd[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) pose[k] = a.gate->pose[k];
    pose_to_Rt_sc(pose, R, t, scd, DevSinCosF());
#pragma unroll
    for (int k = 0; k < 6; ++k) s_pose[k] = pose[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) s_R[k] = R[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) s_ti[r] = ((-R[r] * t[0]) + (-R[3 + r] * t[1])) + (-R[6 + r] * t[2]);  // Eigen Isometry inverse: R^T, (-R^T) t
  }
  __syncthreads();
  int i = blockIdx.x * 256 + threadIdx.x;
  int c = 0;
  if (i >= a.n[0]) {
    i -= a.n[0];
    c = 1;
  }
  const bool on = i < a.n[c];
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on) {
    q = a.src[c][i];
    // ring order (what lets the search take the ring windows of :366-403 / :430-477 as index ranges) and ring ids that fit a byte
    const int ring = (int)q.w;
    const int prev = i > 0 ? (int)a.src[c][i - 1].w : ring;
    if (ring < prev || ring < 0 || ring > 255 || !(q.w == q.w)) atomicOr(a.hdr + c, 1u);
    // the ring table: every r in (previous point's ring, this point's ring] starts here; the last point closes the table
    int32_t *rs = a.ring_start + c * (OH_RINGS + 1);
    const int r0 = i > 0 ? min(max(prev, -1), OH_RINGS - 1) : -1, r1 = min(max(ring, -1), OH_RINGS - 1);
    for (int r = r0 + 1; r <= r1; ++r) rs[r] = i;
    if (i == a.n[c] - 1)
      for (int r = r1 + 1; r <= OH_RINGS; ++r) rs[r] = a.n[c];
    if (a.to_end) {  // transformToStart (:135-142) with the point's own share of the sweep, then the whole transform's inverse
      const float s = 10 * (q.w - (int)q.w);
      float ps[6], Rs[9], ts[3], scd[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) ps[k] = s_pose[k] * s;
      pose_to_Rt_sc(ps, Rs, ts, scd, DevSinCosF());
      const float a0 = ((Rs[0] * q.x + Rs[1] * q.y) + Rs[2] * q.z) + ts[0];
      const float a1 = ((Rs[3] * q.x + Rs[4] * q.y) + Rs[5] * q.z) + ts[1];
      const float a2 = ((Rs[6] * q.x + Rs[7] * q.y) + Rs[8] * q.z) + ts[2];
      float4 o;
      o.x = ((s_R[0] * a0 + s_R[3] * a1) + s_R[6] * a2) + s_ti[0];
      o.y = ((s_R[1] * a0 + s_R[4] * a1) + s_R[7] * a2) + s_ti[1];
      o.z = ((s_R[2] * a0 + s_R[5] * a1) + s_R[8] * a2) + s_ti[2];
      o.w = q.w;
      q = o;
    }
    a.org[c][i] = q;
  }
  // (a wavefront that straddles the two clouds counts into two sets of tables: the cloud is part of the run's key)
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const float inv_c = 1.0f / OH_CELL[l];
    const uint32_t b = oh_bucket(oh_cell(q.x, inv_c), oh_cell(q.y, inv_c), oh_cell(q.z, inv_c)) + (uint32_t)(2 * c + l) * OH_SIZE;
    (void)oh_run_add(a.cnt, b, on);
  }
}

// Exclusive scan of the tables' counts: 16 workgroups per (cloud, level), each over 4096 buckets; its base is the sum of the
// buckets in front of it, which it adds up itself (at most 240 KB out of the L2: cheaper than a chain of workgroups waiting
// for each other, or than contended coarse counters in the counting launch -- measured: 120 us of atomics for a 42 000-point
// cloud).  Leaves start[] and a copy in cursor[] for the scatter; the counts are zeroed by a memset in front of the next build.
__global__ __launch_bounds__(256) void odom_scan_kernel(const uint32_t *cnt, uint32_t *start, uint32_t *cursor, const GNState *gate) {
  if (gate && !gate->done) return;
  __shared__ uint32_t part[4];
  __shared__ uint32_t s_base;
  const int w = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t *c0 = cnt + (size_t)t * OH_SIZE;
  const uint32_t *c = c0 + (size_t)w * 4096;
  uint32_t *st = start + (size_t)t * (OH_SIZE + 1) + (size_t)w * 4096, *cu = cursor + (size_t)t * OH_SIZE + (size_t)w * 4096;
  uint32_t bsum = 0;
  for (int k = 0; k < w; ++k) {  // (coalesced: 16 bytes per thread and chunk quarter)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint4 x = reinterpret_cast<const uint4 *>(c0 + (size_t)k * 4096)[u * 256 + tid];
      bsum += x.x + x.y + x.z + x.w;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bsum += __shfl_xor(bsum, o, 64);
  if (lane == 0) part[wave] = bsum;
  __syncthreads();
  if (tid == 0) s_base = part[0] + part[1] + part[2] + part[3];
  uint4 v[4];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = reinterpret_cast<const uint4 *>(c + tid * 16)[k];
    sum += v[k].x + v[k].y + v[k].z + v[k].w;
  }
  uint32_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  __syncthreads();  // (part is read above)
  if (lane == 63) part[wave] = incl;
  __syncthreads();
  uint32_t before = s_base, all = s_base;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t p = part[k];
    before += k < wave ? p : 0u;
    all += p;
  }
  uint32_t run = before + incl - sum;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint4 o;
    o.x = run; run += v[k].x;
    o.y = run; run += v[k].y;
    o.z = run; run += v[k].z;
    o.w = run; run += v[k].w;
    // (a table of start has OH_SIZE + 1 entries: its rows are not 16-byte aligned from the second table on)
    st[tid * 16 + 4 * k] = o.x; st[tid * 16 + 4 * k + 1] = o.y; st[tid * 16 + 4 * k + 2] = o.z; st[tid * 16 + 4 * k + 3] = o.w;
    reinterpret_cast<uint4 *>(cu + tid * 16)[k] = o;
  }
  if (w == 15 && tid == 0) st[4096] = all;  // start[OH_SIZE]
}

struct ScatterArgs {
  const float4 *org[2];
  int32_t n[2];
  uint32_t *cursor;    // [2][2][OH_SIZE]
  float4 *sorted[4];   // [cloud * 2 + level]
  const GNState *gate;
};
// (the order of the points inside a bucket is whatever the atomics give: the search takes minima over them with complete
// tie rules, so its answer does not depend on it)
__global__ __launch_bounds__(256) void odom_scatter_kernel(ScatterArgs a) {
  if (a.gate && !a.gate->done) return;
  int i = blockIdx.x * 256 + threadIdx.x;
  int c = 0;
  if (i >= a.n[0]) {
    i -= a.n[0];
    c = 1;
  }
  const bool on = i < a.n[c];
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on) q = a.org[c][i];
  const uint32_t tag = (uint32_t)i | (((uint32_t)(int)q.w & 255u) << OH_IDX_BITS);
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const float inv_c = 1.0f / OH_CELL[l];
    const uint32_t b = oh_bucket(oh_cell(q.x, inv_c), oh_cell(q.y, inv_c), oh_cell(q.z, inv_c)) + (uint32_t)(2 * c + l) * OH_SIZE;
    const uint32_t pos = oh_run_add(a.cursor, b, on);
    if (on) a.sorted[2 * c + l][pos] = make_float4(q.x, q.y, q.z, __uint_as_float(tag));
  }
}

// ---- the correspondence refresh of every fifth iteration (:357-408, :423-483), one wavefront per query ---------------------
struct SearchArgs {
  HashView h[2];          // corner, surf
  const float4 *org[2];   // the last clouds in scan order
  int32_t n_org[2];
  const float4 *q[2];     // sharp, flat
  int32_t nq[2];
  int32_t *ind;           // [3][n_sharp + n_flat]
  const GNState *state;
  const uint32_t *hdr;    // [2] per last cloud: bit 0 = not in ring order / ring ids beyond a byte (odom_prep_kernel)
  const int32_t *ring_start;  // [2][OH_RINGS + 1]
  uint32_t *flags;        // [0] |= 1: a query's nearest neighbour needs nanoflann's visit order
  float nf_slack;         // 0, or lslam_grid.hpp's GRID_NF_PRUNE_SLACK_WIDE (LSLAM_AB_WIDE_NF_MARGIN)
  int32_t literal_window; // 1: the ring windows walked point by point whatever the cloud (A/B switch, LSLAM_ODOM_LITERAL_WINDOW=1)
  uint32_t *dbg;          // (or null) profiling tap, [queries][4]: 10 ns ticks, candidates of the nearest-neighbour passes, of the
                          // category passes, bit 0 / 1: a coarse-level pass in the former / the latter
};

// The runs of level `lvl`'s 27 cells around `sel` into LDS ([0..31] exclusive prefix of the run lengths, entries 27.. = total;
// [32..63] run starts); a bucket two of the cells share is taken once.  Returns the number of candidates, or -1 when the level
// cannot be used for this query (coordinates beyond the rounding analysis, NaN).  rg2: every point NOT among the candidates is
// at least this far away (squared).
LSLAM_DEV int oh_runs(const HashView &H, const int lvl, const float (&sel)[3], const int lane, uint32_t *lds, float &rg2,
                      const float bound = FLT_MAX) {
  const float c = OH_CELL[lvl], inv_c = 1.0f / OH_CELL[lvl];
  const float ux = __fmul_rn(sel[0], inv_c), uy = __fmul_rn(sel[1], inv_c), uz = __fmul_rn(sel[2], inv_c);
  rg2 = 0.0f;
  if (!(fabsf(ux) < OH_U_MAX && fabsf(uy) < OH_U_MAX && fabsf(uz) < OH_U_MAX)) return -1;
  const float fx = floorf(ux), fy = floorf(uy), fz = floorf(uz);
  const float ex = ux - fx, ey = uy - fy, ez = uz - fz;
  const float wall = fminf(fminf(fminf(ex, 1.0f - ex), fminf(ey, 1.0f - ey)), fminf(ez, 1.0f - ez));
  const float rg = c * (1.0f - 1.0e-6f) * ((1.0f - OH_U_SLACK) + wall);
  rg2 = (rg * rg) * (1.0f - 1.0e-5f);
  const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
  const uint32_t b = oh_bucket((int)fx + dx, (int)fy + dy, (int)fz + dz);
  bool off = lane >= 27;  // this lane's cell is not looked at
  if (bound < 1.0e30f) {
    // a cell whose box is farther from the query than sqrt(bound) holds nothing the caller could use: dropped.  (gaps to the
    // cell walls in cell units, a rounding slack short)
    const float gx = dx == 0 ? 0.0f : fmaxf((dx < 0 ? ex : 1.0f - ex) - OH_U_SLACK, 0.0f);
    const float gy = dy == 0 ? 0.0f : fmaxf((dy < 0 ? ey : 1.0f - ey) - OH_U_SLACK, 0.0f);
    const float gz = dz == 0 ? 0.0f : fmaxf((dz < 0 ? ez : 1.0f - ez) - OH_U_SLACK, 0.0f);
    const float cc = c * (1.0f - 1.0e-6f);
    off = off || ((gx * gx + gy * gy) + gz * gz) * (cc * cc) * (1.0f - 1.0e-5f) > bound;
  }
  bool dup = off;  // ... or its bucket is the one of an earlier lane that is looked at
  for (int j = 0; j < 26; ++j) {
    const uint32_t bj = __shfl(b, j, 64);
    const int offj = __shfl((int)off, j, 64);
    dup = dup || (j < lane && bj == b && !offj);
  }
  const uint32_t *st = H.start + (size_t)lvl * (OH_SIZE + 1);
  uint32_t s0 = 0, len = 0;
  if (!dup) {
    s0 = st[b];
    len = st[b + 1] - s0;
  }
  uint32_t incl = len;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const uint32_t u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  const uint32_t total = __shfl(incl, 31, 64);
  __syncthreads();  // (one wavefront per workgroup: the table may still be read by the scan before)
  if (lane < 32) {
    lds[lane] = incl - len;
    lds[32 + lane] = s0;
  }
  __syncthreads();
  return (int)total;
}

// position in the level's point array of candidate k (k < total) of the runs in LDS
LSLAM_DEV uint32_t oh_candidate(const uint32_t *lds, const uint32_t k) {
  int r = 0;  // the last run whose prefix is <= k (empty runs share a prefix with their successor: the last of them holds k)
#pragma unroll
  for (int step = 16; step > 0; step >>= 1) {
    const int m = r + step;
    if (m < 32 && lds[m] <= k) r = m;
  }
  return lds[32 + r] + (k - lds[r]);
}

// smallest (best) and second smallest (sec; a second point AT the smallest distance counts) of the per-lane (b1, b2), and the
// winner's payload
LSLAM_DEV void oh_reduce_nn(const int lane, const float b1, const float b2, const int i1, float &best, float &sec, int &bi) {
  float dmin = b1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, off, 64));
  best = FLT_MAX;
  sec = FLT_MAX;
  bi = -1;
  const unsigned long long win = __ballot(b1 == dmin && i1 >= 0);
  if (win == 0ull) return;  // no candidate at all (or only NaN distances)
  const int wl = __ffsll((long long)win) - 1;
  float s2 = lane == wl ? b2 : b1;  // everybody's best but the winner's own, and the winner's second
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s2 = fminf(s2, __shfl_xor(s2, off, 64));
  best = dmin;
  sec = s2;
  bi = __shfl(i1, wl, 64);
}

// nearest candidate of a category: smallest distance, among equal distances the one the reference's walk meets first (rank)
struct CatBest {
  float d = FLT_MAX;
  uint32_t rank = 0xFFFFFFFFu;
  int j = -1;
  LSLAM_DEV void take(float dd, uint32_t rk, int jj) {
    if (dd < d || (dd == d && rk < rank)) {
      d = dd;
      rank = rk;
      j = jj;
    }
  }
};
LSLAM_DEV void oh_reduce_cat(CatBest &b) {
  float dmin = b.d;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, off, 64));
  uint32_t rk = (b.d == dmin && b.j >= 0) ? b.rank : 0xFFFFFFFFu;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) rk = min(rk, (uint32_t)__shfl_xor((int)rk, off, 64));
  const unsigned long long win = __ballot(b.d == dmin && b.j >= 0 && b.rank == rk);
  int j = -1;
  if (win != 0ull) j = __shfl(b.j, __ffsll((long long)win) - 1, 64);
  b.d = win != 0ull ? dmin : FLT_MAX;
  b.rank = rk;
  b.j = j;
}

__global__ __launch_bounds__(64) void odom_search_kernel(SearchArgs a) {
  const GNState *st = a.state;
  if (st->done) return;
  __shared__ uint32_t lds[64];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x;
  const unsigned long long t_begin = a.dbg ? wall_clock64() : 0ull;
  uint32_t dbg_c1 = 0, dbg_c2 = 0, dbg_f = 0;
  const int nall = a.nq[0] + a.nq[1];
  const int c = qi >= a.nq[0] ? 1 : 0;
  const int li = c ? qi - a.nq[0] : qi;
  const float4 q = a.q[c][li];
  // transformToStart (:135-142), wave-uniform
  const float s = 10 * (q.w - (int)q.w);
  float ps[6], R[9], t[3], scd[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) ps[k] = st->pose[k] * s;
  pose_to_Rt_sc(ps, R, t, scd, DevSinCosF());
  float sel[3];
  sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
  sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
  sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
  const HashView &H = a.h[c];
  // ---- nearestKSearch(pointSel, 1) --------------------------------------------------------------------------------------
  float best = FLT_MAX, sec = FLT_MAX, rg2 = 0.0f;
  int bi = -1, lds_level = -1, lds_total = 0;
  float lds_rg2 = 0.0f;
  bool decided = false;
  for (int lvl = 0; lvl < 2 && !decided; ++lvl) {
    // (the coarse level only where something nearer than what the fine level saw, and than the gate, could be)
    const float bound = lvl == 0 ? FLT_MAX : fminf(best, 25.0f);
    const int total = oh_runs(H, lvl, sel, lane, lds, rg2, bound);
    if (total < 0) continue;
    lds_level = lvl;
    lds_total = total;
    lds_rg2 = rg2;
    dbg_c1 += (uint32_t)total;
    dbg_f |= lvl == 1 ? 1u : 0u;
    const float4 *P = H.pts[lvl];
    float b1 = FLT_MAX, b2 = FLT_MAX;
    int i1 = -1;
    for (uint32_t k0 = lane; k0 < (uint32_t)total; k0 += 256) {  // four candidates per lane in flight
      float4 p[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t k = k0 + 64u * u;
        ok[u] = k < (uint32_t)total;
        p[u] = P[ok[u] ? oh_candidate(lds, k) : 0u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = ok[u] ? dist2_xyz(sel[0], sel[1], sel[2], p[u]) : FLT_MAX;
        if (d < b1) {
          b2 = b1;
          b1 = d;
          i1 = (int)(__float_as_uint(p[u].w) & OH_IDX_MASK);
        } else if (d < b2) {
          b2 = d;
        }
      }
    }
    oh_reduce_nn(lane, b1, b2, i1, best, sec, bi);
    // the fine level decides when its nearest is inside its guaranteed radius; the coarse level's radius is beyond the gate:
    // either its nearest is inside (and then nearer than everything not looked at) or nothing is within 5 m
    decided = best < rg2 * (1.0f - a.nf_slack) || lvl == 1;
  }
  if (!decided) {  // coordinates beyond the grids' rounding analysis: the whole cloud, 64 points per round
    float b1 = FLT_MAX, b2 = FLT_MAX;
    int i1 = -1;
    for (int k = lane; k < a.n_org[c]; k += 64) {
      const float4 p = a.org[c][k];
      const float d = dist2_xyz(sel[0], sel[1], sel[2], p);
      if (d < b1) { b2 = b1; b1 = d; i1 = k; }
      else if (d < b2) b2 = d;
    }
    oh_reduce_nn(lane, b1, b2, i1, best, sec, bi);
  }
  int i1 = -1, i2 = -1, i3 = -1;
  if (bi >= 0 && best < 25.0f) {  // :363 / :429
    // two different points at the winner's distance (or, with the margin on, within nanoflann's own pruning rounding of it):
    // which of them nearestKSearch returns is its traversal's business
    if (!(best < sec * (1.0f - a.nf_slack)) || best == sec) {
      if (lane == 0) atomicOr(a.flags, 1u);
    }
    i1 = bi;
    const bool is_flat = c != 0;
    bool walked = false;
    if (decided && !a.literal_window && (a.hdr[c] & 1u) == 0u) {
      // ---- the ring windows (:366-403 / :430-477) as nearest-neighbour questions -------------------------------------------
      // The cloud is in ring order, so what the two walks from i1 visit before they break is an index range: upwards
      // j in (i1, limit) with ring <= scan + 2, downwards j < i1 with ring >= scan - 2.  Per category the walks keep the FIRST
      // strictly smaller distance below 25: the smallest distance, and among equal ones the point met first (upwards before
      // downwards, nearer to i1 first) -- the `rank` below.  The nearest point of a category is found like the nearest point
      // of the cloud: among the points of the 27 cells, proven when it is inside the level's guaranteed radius; the coarse
      // level's radius is beyond the 5 m the categories start from, so it decides what the fine level cannot.
      const int scan = (int)a.org[c][i1].w;
      const int limit = min(a.nq[c], a.n_org[c]);
      bool dec2 = false, dec3 = !is_flat;
      walked = true;
      {  // the fine level: the candidates of the 27 cells (still in LDS from the nearest-neighbour pass as a rule)
        int total = lds_total;
        float r2 = lds_rg2;
        if (lds_level != 0) {
          total = oh_runs(H, 0, sel, lane, lds, r2);
          lds_level = 0;
          lds_total = total;
          lds_rg2 = r2;
        }
        if (total > 0) {
          dbg_c2 += (uint32_t)total;
          const float4 *P = H.pts[0];
          CatBest c2, c3;
          for (uint32_t k0 = lane; k0 < (uint32_t)total; k0 += 256) {
            float4 p[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const uint32_t k = k0 + 64u * u;
              ok[u] = k < (uint32_t)total;
              p[u] = P[ok[u] ? oh_candidate(lds, k) : 0u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const uint32_t tag = __float_as_uint(p[u].w);
              const int j = (int)(tag & OH_IDX_MASK), ring = (int)(tag >> OH_IDX_BITS);
              const float dd = sq_diff3(p[u], sel);  // calcSquaredDiff(cloud point, pointSel)
              const bool up = j > i1 && j < limit && ring <= scan + 2;
              const bool down = j < i1 && ring >= scan - 2;
              if (!ok[u] || !(up || down) || !(dd < 25.0f)) continue;
              const uint32_t rank = up ? (uint32_t)(j - i1) : (1u << OH_IDX_BITS) + (uint32_t)(i1 - j);
              if (!is_flat) {
                if (up ? ring > scan : ring < scan) c2.take(dd, rank, j);
              } else if (ring == scan) {
                c2.take(dd, rank, j);
              } else {
                c3.take(dd, rank, j);
              }
            }
          }
          oh_reduce_cat(c2);
          if (c2.d < r2) {
            dec2 = true;
            i2 = c2.j;
          }
          if (is_flat) {
            oh_reduce_cat(c3);
            if (c3.d < r2) {
              dec3 = true;
              i3 = c3.j;
            }
          }
        }
      }
      if (!(dec2 && dec3)) {
        // What the fine level cannot prove: the category's own points, all of them -- in a cloud in ring order the rings a
        // category takes are index ranges (the ring table), so this is the reference's walk without the points of the other
        // categories: at most two rings below and two above i1's (sharp, flat's third point), or i1's own ring (flat's second).
        dbg_f |= 2u;
        const int32_t *rs = a.ring_start + c * (OH_RINGS + 1);
        auto rsat = [&](int r) { return rs[min(max(r, 0), OH_RINGS)]; };
        const float4 *O = a.org[c];
        // ranges [lo, hi) below i1 and above it, per category
        const int lo_adj = rsat(scan - 2), hi_adj_dn = rsat(scan);                 // rings scan - 2, scan - 1
        const int lo_adj_up = rsat(scan + 1), hi_adj = min(rsat(scan + 3), limit); // rings scan + 1, scan + 2 (j < limit: quirk Q5)
        const int lo_own = rsat(scan), hi_own = min(rsat(scan + 1), limit);        // i1's own ring
        auto scan_range = [&](int lo, int hi, bool upward, CatBest &cb) {
          for (int j0 = lo + lane; j0 < hi; j0 += 256) {
            float4 p[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + 64 * u;
              ok[u] = j < hi;
              p[u] = O[ok[u] ? j : lo];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + 64 * u;
              const float dd = sq_diff3(p[u], sel);
              if (!ok[u] || !(dd < 25.0f)) continue;
              cb.take(dd, upward ? (uint32_t)(j - i1) : (1u << OH_IDX_BITS) + (uint32_t)(i1 - j), j);
            }
          }
        };
        if (!dec2) {
          CatBest c2;
          if (!is_flat) {
            scan_range(max(lo_adj_up, i1 + 1), hi_adj, true, c2);
            scan_range(lo_adj, min(hi_adj_dn, i1), false, c2);
            dbg_c2 += (uint32_t)(max(hi_adj - max(lo_adj_up, i1 + 1), 0) + max(min(hi_adj_dn, i1) - lo_adj, 0));
          } else {
            scan_range(max(lo_own, i1 + 1), hi_own, true, c2);
            scan_range(lo_own, min(rsat(scan + 1), i1), false, c2);
            dbg_c2 += (uint32_t)(max(hi_own - max(lo_own, i1 + 1), 0) + max(min(rsat(scan + 1), i1) - lo_own, 0));
          }
          oh_reduce_cat(c2);
          i2 = c2.j;
        }
        if (!dec3) {
          CatBest c3;
          scan_range(max(lo_adj_up, i1 + 1), hi_adj, true, c3);
          scan_range(lo_adj, min(hi_adj_dn, i1), false, c3);
          dbg_c2 += (uint32_t)(max(hi_adj - max(lo_adj_up, i1 + 1), 0) + max(min(hi_adj_dn, i1) - lo_adj, 0));
          oh_reduce_cat(c3);
          i3 = c3.j;
        }
      }
    }
    if (!walked) odom_window_walk(a.org[c], a.n_org[c], a.nq[c], is_flat, i1, sel, lane, i2, i3);
  }
  if (lane == 0) {
    a.ind[qi] = i1;
    a.ind[nall + qi] = i2;
    a.ind[2 * nall + qi] = i3;
    if (a.dbg) {
      a.dbg[4 * qi] = (uint32_t)(wall_clock64() - t_begin);
      a.dbg[4 * qi + 1] = dbg_c1;
      a.dbg[4 * qi + 2] = dbg_c2;
      a.dbg[4 * qi + 3] = dbg_f;
    }
  }
}

// ---- up to five iterations of the loop in one launch ------------------------------------------------------------------------
// Between two correspondence refreshes the loop of :328-647 is residual pass -> 6x6 solve -> residual pass ...: a few dozen
// microseconds of work per iteration for a handful of workgroups, which as launches costs more in gaps than in work.  Here the
// workgroups of the residual pass stay resident for the (up to) five iterations to the next refresh: each keeps ITS OWN copy of
// the loop's state in LDS, leaves its block's 32 sums in a slot of the iteration's generation, polls the slots of all blocks
// (sentinel = "not written yet"; the host fills every generation with it before the launch, so nothing is ever reset inside),
// adds them up in the solve kernel's order and runs the solve REPLICATED -- every workgroup computes the same next pose from
// the same numbers, nothing travels back.  Same bits as odom_sweep_kernel + solve_kernel (tests/test_gpu_odom.py holds the two
// against each other through the kd-tree implementation, which keeps the launches).  A spin limit turns a grid that is not
// co-resident (another process's persistent kernel on the device) into an abort flag and a fallback to the launches.
constexpr uint32_t OGN_SENT = 0xFFF8DEADu;  // a NaN payload no sum produces
constexpr int OGN_MAX_BLOCKS = 64;
constexpr int OGN_GENERATIONS = 5;
struct GnSegArgs {
  OdomArgs oa;
  float *slots;     // [OGN_GENERATIONS][nb_total][NCOL]
  uint32_t *abort;  // [1]
  GNState *state;   // in / out
  SolveParams sp;
  uint32_t spin_limit;
};

__global__ __launch_bounds__(256) void odom_gn_kernel(GnSegArgs g) {
  constexpr int BLOCK = 256, NWAVE = 4;
  __shared__ GNState lst;
  __shared__ float red[NWAVE][NCOL];
  __shared__ float pv[OGN_MAX_BLOCKS * NCOL];
  __shared__ double redd[SOLVE_GROUPS][NCOL];
  __shared__ double tot[NCOL];
  __shared__ GnShared sh;
  __shared__ int go;
  const OdomArgs &a = g.oa;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lb = blockIdx.x, nb = a.nb_total;
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(g.state);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&lst);
    for (int i = tid; i < (int)(sizeof(GNState) / 4); i += BLOCK) dst[i] = src[i];
  }
  __syncthreads();
  if (lst.done) return;
  if (g.spin_limit == 0u) {  // test hook (LSLAM_ODOM_SPIN_LIMIT=0): the way out an exchange that times out takes
    if (tid == 0) __hip_atomic_store(g.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const bool is_flat = lb >= a.nb_sharp;
  const int li = (is_flat ? lb - a.nb_sharp : lb) * BLOCK + tid;
  const int nq = is_flat ? a.n_flat : a.n_sharp;
  const bool active = li < nq;
  const int qi = is_flat ? a.n_sharp + li : li;
  const int nall = a.n_sharp + a.n_flat;
  const float4 *org = is_flat ? a.os : a.oc;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  int i1 = -1, i2 = -1, i3 = -1;
  if (active) {
    q = is_flat ? a.qf[li] : a.q[li];
    i1 = a.ind[qi];
    i2 = a.ind[nall + qi];
    i3 = a.ind[2 * nall + qi];
  }
  for (int k = 0; k < OGN_GENERATIONS; ++k) {
    // ---- this block's residual pass: odom_sweep_kernel with the correspondences as they are (mode 2) ----------------------
    const int iter = lst.loop_iter;
    float row[6] = {0, 0, 0, 0, 0, 0};
    float rb = 0.0f, kept = 0.0f;
    if (active) {
      const float s = 10 * (q.w - (int)q.w);  // transformToStart (:135-142)
      float ps[6], R[9], t[3], scd[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) ps[i] = lst.pose[i] * s;
      pose_to_Rt_sc(ps, R, t, scd, DevSinCosF());
      float sel[3];
      sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
      sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
      sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
      float coeff[4];
      bool ok = false;
      if (!is_flat) {
        if (i2 >= 0) ok = odom_corner_coeff(org[i1], org[i2], sel, iter, coeff);
      } else {
        if (i2 >= 0 && i3 >= 0) ok = odom_surf_coeff(org[i1], org[i2], org[i3], sel, iter, coeff);
      }
      if (ok) {
        float sc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) sc[i] = lst.sc[i];
        jacobian_row(sc, q.x, q.y, q.z, coeff, row, rb);
        rb = (float)(-0.05 * (double)coeff[3]);  // :575
        kept = 1.0f;
      }
    }
    float v[NCOL];
#pragma unroll
    for (int i = 0; i < NCOL; ++i) v[i] = 0.0f;
    int kk = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) v[kk++] = row[i] * row[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[COL_ATB + i] = row[i] * rb;
    v[COL_ROWS] = kept;
    v[COL_LINE] = is_flat ? 0.0f : kept;
    v[COL_PLANE] = is_flat ? kept : 0.0f;
#pragma unroll
    for (int i = 0; i < 31; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NCOL; ++i) red[wave][i] = v[i];
    }
    __syncthreads();
    // ---- exchange ----------------------------------------------------------------------------------------------------------
    float *slot = g.slots + (size_t)k * nb * NCOL;
    if (tid < NCOL) {
      float sacc = red[0][tid];
#pragma unroll
      for (int w = 1; w < NWAVE; ++w) sacc += red[w][tid];
      __hip_atomic_store(slot + (size_t)lb * NCOL + tid, sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t spins = 0;; ++spins) {
      int bad = 0;
      for (int i = tid; i < nb * NCOL; i += BLOCK) {
        const float x = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pv[i] = x;
        bad |= (__float_as_uint(x) == OGN_SENT) ? 1 : 0;
      }
      if (!__syncthreads_or(bad)) break;
      if ((spins & 63u) == 63u) {  // (uniform: every thread takes the same way out)
        const int ab = (spins > g.spin_limit || __hip_atomic_load(g.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
        if (__syncthreads_or(ab)) {
          if (tid == 0) __hip_atomic_store(g.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    // ---- the solve kernel's reduction (rows grp, grp + 32, ... in order, then the 32 groups in order; fp64) and its solve ------
    {
      const int col = tid & 31, g0 = tid >> 5;
#pragma unroll
      for (int gg = 0; gg < SOLVE_GROUPS / 8; ++gg) {
        const int grp = g0 + 8 * gg;
        double sgrp = 0.0;
        for (int b = grp; b < nb; b += SOLVE_GROUPS) sgrp += (double)pv[b * NCOL + col];
        redd[grp][col] = sgrp;
      }
    }
    __syncthreads();
    if (tid < NCOL) {
      double x = 0.0;
#pragma unroll
      for (int gi = 0; gi < SOLVE_GROUPS; ++gi) x += redd[gi][tid];
      tot[tid] = x;
      lst.sums[tid] = x;
    }
    __syncthreads();
    solve_finish(&lst, tot, sh, go, g.sp);
    __syncthreads();
    if (lst.done || lst.loop_iter % 5 == 0) break;  // the loop has ended, or the correspondences are due (:357, :423)
  }
  if (blockIdx.x == 0) {
    uint32_t *dst = reinterpret_cast<uint32_t *>(g.state);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&lst);
    for (int i = tid; i < (int)(sizeof(GNState) / 4); i += BLOCK) dst[i] = src[i];
  }
}

template <typename T>
struct Buf {
  T *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 256;
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// one generation of last clouds with its grids
struct Side {
  Buf<float4> org[2];
  Buf<float4> sorted[4];  // [cloud * 2 + level]
  uint32_t *start = nullptr;  // [4][OH_SIZE + 1]
  uint32_t *hdr = nullptr;    // [2] per cloud: bit 0 = not in ring order (odom_prep_kernel); behind it the ring table [2][OH_RINGS + 1]
  size_t n[2] = {0, 0};
};

}  // namespace

struct lslam_odom {
  lslam_ctx *ctx = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  int32_t max_it = 25;
  float dt = 0.1f, dr = 0.1f;
  bool inited = false;
  float transform[6] = {0, 0, 0, 0, 0, 0};
  float Tsum[16];
  Side side[2];
  int cur = 0;
  uint32_t *cnt = nullptr, *cursor = nullptr;  // [4][OH_SIZE]
  int32_t literal_window = 0;
  // odom_gn_kernel
  Buf<float> slots;
  bool persistent = true;     // LSLAM_ODOM_PERSISTENT=0 (debug hook): the launch-per-step loop
  bool persistent_ok = true;  // false once an exchange ran into its spin limit
  uint32_t spin_limit = 1u << 18;
  uint64_t persistent_runs = 0, launch_runs = 0;
  Buf<uint32_t> dbg;  // LSLAM_ODOM_SEARCH_TAP=1 (debug hook): per-query profile of the last search launch (lslam_debug_odom_search)
  bool dbg_on = false;
  size_t dbg_n = 0;
  Buf<int32_t> ind;
  Buf<float> partials;
  ProbBlocks *d_probs = nullptr;
  GNState *d_state = nullptr, *h_state = nullptr;  // h_state pinned: [0] the state, then 64 bytes of flags, then the ProbBlocks going up
  int32_t nb_on_device = -1;
  uint32_t *d_flags = nullptr;
  float4 *h_last = nullptr;  // pinned staging of the last clouds going out (= h_own, or a buffer of the publishing ring)
  float4 *h_own = nullptr;
  size_t h_own_cap = 0;
  // lslam_odom_set_publish: a ring of pinned buffers the last clouds are copied to with every sweep, handed out as views
  std::vector<float4 *> pub;
  std::vector<size_t> pub_cap;
  int pub_next = 0;
  const float4 *view_c = nullptr, *view_s = nullptr;
  size_t view_nc = 0, view_ns = 0;
  float4 *h_up = nullptr;    // pinned staging of host clouds coming in (lslam_fset_upload, lslam_odometry_match)
  size_t h_up_cap = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int iter_hint = 6;
  uint64_t sweeps = 0;
  int32_t tree_fallbacks = 0;
  lslam_fset *own_fs = nullptr;  // lslam_odometry_match's clouds
  float nf_slack = 0.0f;
};

namespace {

void od_free(lslam_odom *od) {
  if (!od) return;
  (void)hipSetDevice(od->device);
  if (od->stream && lslam::ctx_alive(od->ctx)) (void)hipStreamSynchronize(od->stream);
  for (Side &s : od->side) {
    for (auto &b : s.org) b.release();
    for (auto &b : s.sorted) b.release();
    if (s.start) (void)hipFree(s.start);
    if (s.hdr) (void)hipFree(s.hdr);
  }

  if (od->cnt) (void)hipFree(od->cnt);
  if (od->cursor) (void)hipFree(od->cursor);
  od->ind.release();
  od->partials.release();
  od->slots.release();
  od->dbg.release();
  if (od->d_probs) (void)hipFree(od->d_probs);
  if (od->d_state) (void)hipFree(od->d_state);
  if (od->d_flags) (void)hipFree(od->d_flags);
  if (od->h_state) (void)hipHostFree(od->h_state);
  if (od->h_own) (void)hipHostFree(od->h_own);
  for (float4 *b : od->pub)
    if (b) (void)hipHostFree(b);
  if (od->h_up) (void)hipHostFree(od->h_up);
  if (od->ev0) (void)hipEventDestroy(od->ev0);
  if (od->ev1) (void)hipEventDestroy(od->ev1);
  if (od->own_fs) lslam_fset_destroy(od->own_fs);
  delete od;
}

int od_create(lslam_ctx *ctx, int32_t max_iterations, float dt, float dr, lslam_odom **out) {
  lslam_odom *od = new lslam_odom();
  od->ctx = ctx;
  od->device = lslam::ctx_device(ctx);
  od->stream = lslam::ctx_stream(ctx);
  od->max_it = max_iterations < 0 ? 0 : max_iterations;
  od->dt = dt;
  od->dr = dr;
  std::memset(od->Tsum, 0, sizeof(od->Tsum));
  od->Tsum[0] = od->Tsum[5] = od->Tsum[10] = od->Tsum[15] = 1.0f;
  auto fail = [&](hipError_t e) {
    lslam::set_error(hipGetErrorString(e));
    od_free(od);
    return LSLAM_ERR_HIP;
  };
  hipError_t e;
  if ((e = hipSetDevice(od->device)) != hipSuccess) return fail(e);
  for (Side &s : od->side) {
    if ((e = hipMalloc((void **)&s.start, 4 * (size_t)(OH_SIZE + 1) * 4)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **)&s.hdr, 64 + 2 * (OH_RINGS + 1) * 4)) != hipSuccess) return fail(e);
  }

  if ((e = hipMalloc((void **)&od->cnt, 4 * (size_t)OH_SIZE * 4)) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void **)&od->cursor, 4 * (size_t)OH_SIZE * 4)) != hipSuccess) return fail(e);
  if ((e = hipMemsetAsync(od->cnt, 0, 4 * (size_t)OH_SIZE * 4, od->stream)) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void **)&od->d_probs, sizeof(ProbBlocks))) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void **)&od->d_state, sizeof(GNState))) != hipSuccess) return fail(e);
  if ((e = hipMalloc((void **)&od->d_flags, 64)) != hipSuccess) return fail(e);
  if ((e = hipHostMalloc((void **)&od->h_state, sizeof(GNState) + 128, hipHostMallocDefault)) != hipSuccess) return fail(e);
  if ((e = hipEventCreate(&od->ev0)) != hipSuccess) return fail(e);
  if ((e = hipEventCreate(&od->ev1)) != hipSuccess) return fail(e);
  if ((e = hipStreamSynchronize(od->stream)) != hipSuccess) return fail(e);
  const char *m = lslam::debug_env("LSLAM_ODOM_NF_MARGIN");
  if (m && std::atoi(m) != 0) od->nf_slack = GRID_NF_PRUNE_SLACK_WIDE;
  m = lslam::debug_env("LSLAM_ODOM_LITERAL_WINDOW");
  if (m && std::atoi(m) != 0) od->literal_window = 1;
  m = lslam::debug_env("LSLAM_ODOM_SEARCH_TAP");
  if (m && std::atoi(m) != 0) od->dbg_on = true;
  m = lslam::debug_env("LSLAM_ODOM_PERSISTENT");
  if (m && std::atoi(m) == 0) od->persistent = false;
  m = lslam::debug_env("LSLAM_ODOM_SPIN_LIMIT");
  if (m) od->spin_limit = (uint32_t)std::atoi(m);
  *out = od;
  return LSLAM_OK;
}

struct HostSinCosF {
  void operator()(float a, float &s, float &c) const {
    s = std::sin(a);  // util/Angle.h:17-18 std::sin/std::cos(float)
    c = std::cos(a);
  }
};

uint32_t *h_flags_of(lslam_odom *od) { return reinterpret_cast<uint32_t *>(od->h_state + 1); }

// K1-K3: src lists -> side `to`'s clouds (moved to the sweep end with the pose of d_state when to_end) and their grids.
// gate: the launches only act once the loop of d_state has ended.
int enqueue_build(lslam_odom *od, int to, const float4 *less_sharp, size_t n_ls, const float4 *less_flat, size_t n_lf, bool to_end,
                  bool gated) {
  Side &S = od->side[to];
  OD_TRY(S.org[0].reserve(n_ls + 1));
  OD_TRY(S.org[1].reserve(n_lf + 1));
  for (int l = 0; l < 2; ++l) {
    OD_TRY(S.sorted[l].reserve(n_ls + 1));
    OD_TRY(S.sorted[2 + l].reserve(n_lf + 1));
  }
  const size_t n = n_ls + n_lf;
  S.n[0] = n_ls;
  S.n[1] = n_lf;
  PrepArgs pa{};
  pa.src[0] = less_sharp; pa.src[1] = less_flat;
  pa.n[0] = (int32_t)n_ls; pa.n[1] = (int32_t)n_lf;
  pa.org[0] = S.org[0].p; pa.org[1] = S.org[1].p;
  pa.cnt = od->cnt;
  pa.hdr = S.hdr;
  pa.ring_start = reinterpret_cast<int32_t *>(S.hdr + 16);
  if (n_ls >= (1u << OH_IDX_BITS) || n_lf >= (1u << OH_IDX_BITS)) {
    lslam::set_error("a last cloud of more than 16 777 215 points");
    return LSLAM_ERR_INVALID;
  }
  OD_TRY(hipMemsetAsync(S.hdr, 0, 8, od->stream));
  OD_TRY(hipMemsetAsync(od->cnt, 0, 4 * (size_t)OH_SIZE * 4, od->stream));
  pa.gate = (gated || to_end) ? od->d_state : nullptr;
  pa.to_end = to_end ? 1 : 0;
  // (an ungated move to the end still reads the pose from d_state: the caller has put it there and marked it done)
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (nb) hipLaunchKernelGGL(odom_prep_kernel, dim3(nb), dim3(256), 0, od->stream, pa);
  hipLaunchKernelGGL(odom_scan_kernel, dim3(16, 4), dim3(256), 0, od->stream, od->cnt, S.start, od->cursor, pa.gate);
  ScatterArgs sa{};
  sa.org[0] = S.org[0].p; sa.org[1] = S.org[1].p;
  sa.n[0] = (int32_t)n_ls; sa.n[1] = (int32_t)n_lf;
  sa.cursor = od->cursor;
  for (int k = 0; k < 4; ++k) sa.sorted[k] = S.sorted[k].p;
  sa.gate = pa.gate;
  if (nb) hipLaunchKernelGGL(odom_scatter_kernel, dim3(nb), dim3(256), 0, od->stream, sa);
  OD_TRY(hipGetLastError());
  return LSLAM_OK;
}

struct TailSpec {
  bool on = false;
  int to = 0;
  const float4 *less_sharp = nullptr, *less_flat = nullptr;
  size_t n_ls = 0, n_lf = 0;
  bool copy_out = false;  // last clouds -> h_last behind the build
};

int enqueue_tail(lslam_odom *od, const TailSpec &t, bool gated) {
  if (!t.on) return LSLAM_OK;
  int rc = enqueue_build(od, t.to, t.less_sharp, t.n_ls, t.less_flat, t.n_lf, true, gated);
  if (rc) return rc;
  if (t.copy_out) {
    const Side &S = od->side[t.to];
    if (t.n_ls) OD_TRY(hipMemcpyAsync(od->h_last, S.org[0].p, t.n_ls * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
    if (t.n_lf) OD_TRY(hipMemcpyAsync(od->h_last + t.n_ls, S.org[1].p, t.n_lf * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
  }
  return LSLAM_OK;
}

// LaserOdometry::scanMatch (:328-647) of the queries sharp / flat (device) against side `from`; pose in/out.  The tail (the
// next last clouds) is enqueued behind every batch of iterations, acting only once the loop has ended: in the common case the
// host waits once per sweep.  *tie: a nearest neighbour needed nanoflann's visit order (the result is then not to be used).
int match_loop(lslam_odom *od, int from, const float4 *sharp, size_t n_sharp, const float4 *flat, size_t n_flat, float pose[6],
               lslam_stats &st, int32_t *searches, bool *tie, const TailSpec &tail) {
  const Side &S = od->side[from];
  const size_t nq = n_sharp + n_flat;
  OD_TRY(od->ind.reserve(3 * nq + 1));
  OdomArgs oa{};
  oa.oc = S.org[0].p;
  oa.os = S.org[1].p;
  oa.n_oc = (int32_t)S.n[0];
  oa.n_os = (int32_t)S.n[1];
  oa.q = sharp;
  oa.qf = flat;
  oa.n_sharp = (int32_t)n_sharp;
  oa.n_flat = (int32_t)n_flat;
  oa.nb_sharp = (int32_t)((n_sharp + 255) / 256);
  oa.nb_total = oa.nb_sharp + (int32_t)((n_flat + 255) / 256);
  oa.ind = od->ind.p;
  oa.sel = nullptr;
  oa.mode = 2;
  oa.state = od->d_state;
  OD_TRY(od->partials.reserve((size_t)(oa.nb_total ? oa.nb_total : 1) * NCOL));
  oa.partials = od->partials.p;
  SearchArgs sa{};
  for (int c = 0; c < 2; ++c) {
    sa.h[c].start = S.start + (size_t)(2 * c) * (OH_SIZE + 1);
    sa.h[c].pts[0] = S.sorted[2 * c].p;
    sa.h[c].pts[1] = S.sorted[2 * c + 1].p;
    sa.h[c].n = (int32_t)S.n[c];
    sa.org[c] = S.org[c].p;
    sa.n_org[c] = (int32_t)S.n[c];
  }
  sa.q[0] = sharp; sa.q[1] = flat;
  sa.nq[0] = (int32_t)n_sharp; sa.nq[1] = (int32_t)n_flat;
  sa.ind = od->ind.p;
  sa.state = od->d_state;
  sa.flags = od->d_flags;
  sa.hdr = S.hdr;
  sa.ring_start = reinterpret_cast<const int32_t *>(S.hdr + 16);
  sa.literal_window = od->literal_window;
  sa.dbg = nullptr;
  if (od->dbg_on) {
    OD_TRY(od->dbg.reserve(4 * nq + 4));
    OD_TRY(hipMemsetAsync(od->dbg.p, 0, (4 * nq + 4) * 4, od->stream));
    sa.dbg = od->dbg.p;
    od->dbg_n = nq;
  }
  sa.nf_slack = od->nf_slack;
  GNState *hs = od->h_state;
  std::memset(hs, 0, sizeof(GNState));
  for (int i = 0; i < 6; ++i) hs->pose[i] = pose[i];
  pose_to_Rt_sc(pose, hs->R, hs->t, hs->sc, HostSinCosF());
  const int max_it = od->max_it;
  if (max_it == 0 || oa.nb_total == 0) hs->done = 1;
  if (od->nb_on_device != oa.nb_total) {
    ProbBlocks *pb = reinterpret_cast<ProbBlocks *>(h_flags_of(od) + 16);  // pinned
    pb->first_block = 0;
    pb->n_blocks = oa.nb_total;
    OD_TRY(hipMemcpyAsync(od->d_probs, pb, sizeof(ProbBlocks), hipMemcpyHostToDevice, od->stream));
    od->nb_on_device = oa.nb_total;
  }
  OD_TRY(hipMemcpyAsync(od->d_state, hs, sizeof(GNState), hipMemcpyHostToDevice, od->stream));
  OD_TRY(hipMemsetAsync(od->d_flags, 0, 64, od->stream));
  if (nq) OD_TRY(hipMemsetAsync(od->ind.p, 0xFF, 3 * nq * sizeof(int32_t), od->stream));
  SolveArgs so{};
  so.states = od->d_state;
  so.partials = od->partials.p;
  so.probs = od->d_probs;
  so.n_prob = 1;
  so.max_iterations = max_it;
  so.delta_r_abort = od->dr;
  so.delta_t_abort = od->dt;
  so.eig_thresh = 10.0f;  // :596
  so.min_rows = 10;       // :501
  so.too_few_continue = 1;
  so.nan_reset = 1;
  OD_TRY(hipEventRecord(od->ev0, od->stream));
  // loop_iter advances by one per solve until the loop is done, so the host knows which iterations refresh the
  // correspondences (every fifth, :357,:423); launches behind the end of the loop exit at once
  int launched = 0, n_search = 0;
  const bool persistent = od->persistent && od->persistent_ok && oa.nb_total >= 1 && oa.nb_total <= OGN_MAX_BLOCKS && max_it > 0;
  if (persistent) {
    // segments of a refresh + up to five iterations in ONE launch (odom_gn_kernel)
    OD_TRY(od->slots.reserve((size_t)OGN_GENERATIONS * oa.nb_total * NCOL));
    GnSegArgs ga{};
    ga.oa = oa;
    ga.slots = od->slots.p;
    ga.abort = od->d_flags + 1;
    ga.state = od->d_state;
    ga.sp.max_iterations = so.max_iterations;
    ga.sp.min_rows = so.min_rows;
    ga.sp.too_few_continue = so.too_few_continue;
    ga.sp.nan_reset = so.nan_reset;
    ga.sp.delta_r_abort = so.delta_r_abort;
    ga.sp.delta_t_abort = so.delta_t_abort;
    ga.sp.eig_thresh = so.eig_thresh;
    ga.spin_limit = od->spin_limit;
    const int n_seg = (max_it + 4) / 5;
    int seg_done = 0;
    int batch = (od->iter_hint + 4) / 5;
    batch = batch < 1 ? 1 : (batch > 2 ? 2 : batch);
    for (;;) {
      if (batch > n_seg - seg_done) batch = n_seg - seg_done;
      for (int b = 0; b < batch; ++b) {
        OD_TRY(hipMemsetD32Async((hipDeviceptr_t)od->slots.p, (int)OGN_SENT, (size_t)OGN_GENERATIONS * oa.nb_total * NCOL, od->stream));
        if (nq) {
          hipLaunchKernelGGL(odom_search_kernel, dim3((unsigned)nq), dim3(64), 0, od->stream, sa);
          ++n_search;
        }
        hipLaunchKernelGGL(odom_gn_kernel, dim3((unsigned)oa.nb_total), dim3(256), 0, od->stream, ga);
      }
      seg_done += batch;
      launched = seg_done * 5 < max_it ? seg_done * 5 : max_it;
      OD_TRY(hipGetLastError());
      OD_TRY(hipEventRecord(od->ev1, od->stream));
      int rc = enqueue_tail(od, tail, true);
      if (rc) return rc;
      OD_TRY(hipMemcpyAsync(hs, od->d_state, sizeof(GNState), hipMemcpyDeviceToHost, od->stream));
      OD_TRY(hipMemcpyAsync(h_flags_of(od), od->d_flags, 64, hipMemcpyDeviceToHost, od->stream));
      OD_TRY(hipStreamSynchronize(od->stream));
      if (h_flags_of(od)[1] != 0u) {  // an exchange gave up: this match again by launches, and launches from now on
        od->persistent_ok = false;
        return match_loop(od, from, sharp, n_sharp, flat, n_flat, pose, st, searches, tie, tail);
      }
      if (hs->done || seg_done >= n_seg) break;
      batch = n_seg - seg_done;
    }
    od->persistent_runs++;
  } else {
  od->launch_runs++;
  int batch = od->iter_hint < 1 ? 1 : (od->iter_hint > 6 ? 6 : od->iter_hint);
  for (;;) {
    if (batch > max_it - launched) batch = max_it - launched;
    for (int b = 0; b < batch; ++b) {
      const int it = launched + b;
      if (it % 5 == 0 && nq) {
        hipLaunchKernelGGL(odom_search_kernel, dim3((unsigned)nq), dim3(64), 0, od->stream, sa);
        ++n_search;
      }
      OD_TRY(launch_odom_sweep(oa, od->stream));
      OD_TRY(launch_solve(so, od->stream));
    }
    launched += batch;
    OD_TRY(hipEventRecord(od->ev1, od->stream));
    int rc = enqueue_tail(od, tail, true);
    if (rc) return rc;
    OD_TRY(hipMemcpyAsync(hs, od->d_state, sizeof(GNState), hipMemcpyDeviceToHost, od->stream));
    OD_TRY(hipMemcpyAsync(h_flags_of(od), od->d_flags, 64, hipMemcpyDeviceToHost, od->stream));
    OD_TRY(hipStreamSynchronize(od->stream));
    if (hs->done || launched >= max_it) break;
    batch = 10;
  }
  }
  if (!hs->done) {  // max_iterations launches without the loop saying so cannot happen (the solve ends it); be safe
    lslam::set_error("the odometry loop did not end");
    return LSLAM_ERR_HIP;
  }
  od->iter_hint = hs->loop_iter + 1;
  const GNState &g = *hs;
  for (int i = 0; i < 6; ++i) pose[i] = g.pose[i];
  st.iterations = g.iter;
  st.sweeps = g.sweeps;
  st.n_rows = g.n_rows;
  st.n_line = g.n_line;
  st.n_plane = g.n_plane;
  st.degenerate = g.degenerate;
  st.converged = g.converged;
  st.delta_r = g.delta_r;
  st.delta_t = g.delta_t;
  st.point_residuals = (int64_t)g.sweeps * (int64_t)nq;
  OD_TRY(hipEventElapsedTime(&st.gpu_ms_total, od->ev0, od->ev1));
  st.status = g.converged ? LSLAM_OK : LSLAM_NOT_CONVERGED;
  if (searches) *searches = (g.loop_iter + 4) / 5 < n_search ? (g.loop_iter + 4) / 5 : n_search;
  *tie = (h_flags_of(od)[0] & 1u) != 0u;
  return LSLAM_OK;
}

// strided host cloud {x, y, z, intensity} -> packed float4 in pinned memory
void pack_xyzi(const void *src, size_t n, size_t stride_bytes, float4 *out) {
  const char *p = static_cast<const char *>(src);
  const size_t ioff = stride_bytes >= 32 ? 16 : 12;  // pcl::PointXYZI keeps intensity at byte 16
  if (stride_bytes == 16) {
    std::memcpy(out, src, n * sizeof(float4));
    return;
  }
  for (size_t i = 0; i < n; ++i) {
    float xyz[3], w;
    std::memcpy(xyz, p + i * stride_bytes, 12);
    std::memcpy(&w, p + i * stride_bytes + ioff, 4);
    out[i] = make_float4(xyz[0], xyz[1], xyz[2], w);
  }
}

int reserve_pinned(float4 *&p, size_t &cap, size_t n) {
  if (n <= cap) return LSLAM_OK;
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
  const size_t want = n + n / 4 + 256;
  OD_TRY(hipHostMalloc((void **)&p, want * sizeof(float4), hipHostMallocDefault));
  cap = want;
  return LSLAM_OK;
}

// host lists -> the feature set's slices, through `stage` (pinned, at least the four sizes together); enqueued on s
int upload_lists(hipStream_t s, lslam_fset *fs, float4 *stage, const void *const src[4], const size_t n[4], size_t stride_bytes) {
  size_t most = 0;
  for (int k = 0; k < 4; ++k) most = n[k] > most ? n[k] : most;
  OD_TRY(lslam::fset_reserve(fs, most));
  size_t off = 0;
  for (int k = 0; k < 4; ++k) {
    fs->counts[k] = n[k];
    if (!n[k]) continue;
    pack_xyzi(src[k], n[k], stride_bytes, stage + off);
    OD_TRY(hipMemcpyAsync(fs->list(k), stage + off, n[k] * sizeof(float4), hipMemcpyHostToDevice, s));
    off += n[k];
  }
  return LSLAM_OK;
}

// the hidden node behind lslam_odometry_match, one per context
std::mutex g_mu;
std::map<lslam_ctx *, lslam_odom *> g_hidden;

void mat4_mul(const float A[16], const float B[16], float C[16]) {
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
      C[r * 4 + c] = s;
    }
}

}  // namespace

namespace lslam {

hipError_t fset_reserve(lslam_fset *fs, size_t points_per_list) {
  if (points_per_list <= fs->cap && fs->buf) return hipSuccess;
  if (fs->buf) (void)hipFree(fs->buf);
  fs->buf = nullptr;
  fs->cap = 0;
  const size_t want = points_per_list + points_per_list / 4 + 256;
  hipError_t e = hipMalloc((void **)&fs->buf, (16 + 4 * want) * sizeof(float4));
  if (e == hipSuccess) fs->cap = want;
  return e;
}

// lslam_ctx_destroy: the context's hidden odometry node goes with it
void odom_ctx_gone(lslam_ctx *ctx) {
  lslam_odom *od = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_hidden.find(ctx);
    if (it != g_hidden.end()) {
      od = it->second;
      g_hidden.erase(it);
    }
  }
  if (od) od_free(od);
}

}  // namespace lslam

extern "C" {

int lslam_fset_create(lslam_ctx *ctx, lslam_fset **out) {
  if (!ctx || !lslam::ctx_alive(ctx) || !out) {
    lslam::set_error("bad feature-set arguments");
    return LSLAM_ERR_INVALID;
  }
  lslam_fset *fs = new lslam_fset();
  fs->device = lslam::ctx_device(ctx);
  *out = fs;
  return LSLAM_OK;
}

void lslam_fset_destroy(lslam_fset *fs) {
  if (!fs) return;
  (void)hipSetDevice(fs->device);
  if (fs->buf) (void)hipFree(fs->buf);
  if (fs->h_stage) (void)hipHostFree(fs->h_stage);
  delete fs;
}

int lslam_fset_counts(const lslam_fset *fs, size_t counts[4]) {
  if (!fs || !counts) {
    lslam::set_error("bad feature-set arguments");
    return LSLAM_ERR_INVALID;
  }
  for (int k = 0; k < 4; ++k) counts[k] = fs->counts[k];
  return LSLAM_OK;
}

int lslam_fset_upload(lslam_ctx *ctx, lslam_fset *fs, const void *sharp, size_t n_sharp, const void *less_sharp, size_t n_less_sharp,
                      const void *flat, size_t n_flat, const void *less_flat, size_t n_less_flat, size_t stride_bytes) {
  if (!ctx || !lslam::ctx_alive(ctx) || !fs || fs->device != lslam::ctx_device(ctx) || stride_bytes < 16 || (stride_bytes & 3) ||
      (n_sharp && !sharp) || (n_less_sharp && !less_sharp) || (n_flat && !flat) || (n_less_flat && !less_flat) ||
      n_sharp > 0x0FFFFFFFu || n_less_sharp > 0x0FFFFFFFu || n_flat > 0x0FFFFFFFu || n_less_flat > 0x0FFFFFFFu) {
    lslam::set_error("bad feature-set upload arguments (clouds need x,y,z,intensity: stride >= 16)");
    return LSLAM_ERR_INVALID;
  }
  OD_TRY(hipSetDevice(fs->device));
  const void *src[4] = {sharp, less_sharp, flat, less_flat};
  const size_t n[4] = {n_sharp, n_less_sharp, n_flat, n_less_flat};
  const size_t total = n_sharp + n_less_sharp + n_flat + n_less_flat;
  int rc = reserve_pinned(fs->h_stage, fs->h_stage_cap, total + 1);
  if (rc) return rc;
  hipStream_t s = lslam::ctx_stream(ctx);
  rc = upload_lists(s, fs, fs->h_stage, src, n, stride_bytes);
  const hipError_t e = hipStreamSynchronize(s);
  if (rc) return rc;
  OD_TRY(e);
  return LSLAM_OK;
}

int lslam_fset_download(lslam_ctx *ctx, const lslam_fset *fs, int32_t which, float *out_xyzi, size_t cap, size_t *n_out) {
  if (!ctx || !lslam::ctx_alive(ctx) || !fs || which < 0 || which > 3 || !n_out) {
    lslam::set_error("bad feature-set download arguments");
    return LSLAM_ERR_INVALID;
  }
  *n_out = fs->counts[which];
  if (!out_xyzi || fs->counts[which] == 0) return LSLAM_OK;
  if (cap < fs->counts[which]) {
    lslam::set_error("feature-set download buffer too small");
    return LSLAM_ERR_INVALID;
  }
  OD_TRY(hipSetDevice(fs->device));
  hipStream_t s = lslam::ctx_stream(ctx);
  OD_TRY(hipMemcpyAsync(out_xyzi, fs->list(which), fs->counts[which] * sizeof(float4), hipMemcpyDeviceToHost, s));
  OD_TRY(hipStreamSynchronize(s));
  return LSLAM_OK;
}

int lslam_odom_create(lslam_ctx *ctx, int32_t max_iterations, float delta_t_abort, float delta_r_abort, lslam_odom **out) {
  if (!ctx || !lslam::ctx_alive(ctx) || !out) {
    lslam::set_error("bad odometry-node arguments");
    return LSLAM_ERR_INVALID;
  }
  return od_create(ctx, max_iterations, delta_t_abort, delta_r_abort, out);
}

void lslam_odom_destroy(lslam_odom *od) { od_free(od); }

int lslam_odom_set_publish(lslam_odom *od, int32_t n_buffers) {
  if (!od || n_buffers < 0 || n_buffers > 64) {
    lslam::set_error("bad publishing arguments (0 .. 64 buffers)");
    return LSLAM_ERR_INVALID;
  }
  (void)hipSetDevice(od->device);
  if (lslam::ctx_alive(od->ctx)) (void)hipStreamSynchronize(od->stream);
  for (float4 *b : od->pub)
    if (b) (void)hipHostFree(b);
  od->pub.assign((size_t)n_buffers, nullptr);
  od->pub_cap.assign((size_t)n_buffers, 0);
  od->pub_next = 0;
  od->view_c = od->view_s = nullptr;
  od->view_nc = od->view_ns = 0;
  return LSLAM_OK;
}

int lslam_odom_last_view(lslam_odom *od, const float **last_corner, size_t *n_corner, const float **last_surf, size_t *n_surf) {
  if (!od || !last_corner || !n_corner || !last_surf || !n_surf) {
    lslam::set_error("bad view arguments");
    return LSLAM_ERR_INVALID;
  }
  *last_corner = reinterpret_cast<const float *>(od->view_c);
  *last_surf = reinterpret_cast<const float *>(od->view_s);
  *n_corner = od->view_nc;
  *n_surf = od->view_ns;
  return LSLAM_OK;
}

int lslam_debug_odom_search(lslam_odom *od, uint32_t *out, size_t cap_queries) {
  if (!od || !od->dbg_on || !out) return 0;
  const size_t n = od->dbg_n < cap_queries ? od->dbg_n : cap_queries;
  if (n == 0) return 0;
  if (hipMemcpy(out, od->dbg.p, n * 16, hipMemcpyDeviceToHost) != hipSuccess) return LSLAM_ERR_HIP;
  return (int)n;
}

int lslam_odom_reset(lslam_odom *od) {
  if (!od) {
    lslam::set_error("no odometry node");
    return LSLAM_ERR_INVALID;
  }
  od->inited = false;
  for (float &v : od->transform) v = 0.0f;
  std::memset(od->Tsum, 0, sizeof(od->Tsum));
  od->Tsum[0] = od->Tsum[5] = od->Tsum[10] = od->Tsum[15] = 1.0f;
  od->side[0].n[0] = od->side[0].n[1] = od->side[1].n[0] = od->side[1].n[1] = 0;
  od->iter_hint = 6;
  return LSLAM_OK;
}

int lslam_odom_last_clouds(lslam_odom *od, float *last_corner, size_t cap_corner, float *last_surf, size_t cap_surf) {
  if (!od || !lslam::ctx_alive(od->ctx)) {
    lslam::set_error("no odometry node");
    return LSLAM_ERR_INVALID;
  }
  const Side &S = od->side[od->cur];
  if ((last_corner && cap_corner < S.n[0]) || (last_surf && cap_surf < S.n[1])) {
    lslam::set_error("last-cloud buffer too small");
    return LSLAM_ERR_INVALID;
  }
  OD_TRY(hipSetDevice(od->device));
  if (last_corner && S.n[0]) OD_TRY(hipMemcpyAsync(last_corner, S.org[0].p, S.n[0] * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
  if (last_surf && S.n[1]) OD_TRY(hipMemcpyAsync(last_surf, S.org[1].p, S.n[1] * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
  OD_TRY(hipStreamSynchronize(od->stream));
  return LSLAM_OK;
}

int lslam_odom_process(lslam_odom *od, lslam_fset *fs, float transform[6], float Tsum[16], lslam_stats *stats, lslam_odom_stats *ostats,
                       float *last_corner, size_t cap_corner, float *last_surf, size_t cap_surf) {
  if (!od || !lslam::ctx_alive(od->ctx) || !fs || fs->device != od->device) {
    lslam::set_error("bad odometry-node arguments (node, feature set on the node's device)");
    return LSLAM_ERR_INVALID;
  }
  const size_t n_sharp = fs->counts[0], n_ls = fs->counts[1], n_flat = fs->counts[2], n_lf = fs->counts[3];
  if ((last_corner && cap_corner < n_ls) || (last_surf && cap_surf < n_lf)) {
    lslam::set_error("last-cloud buffer too small");
    return LSLAM_ERR_INVALID;
  }
  OD_TRY(hipSetDevice(od->device));
  lslam_stats local;
  lslam_stats &st = stats ? *stats : local;
  std::memset(&st, 0, sizeof(st));
  const bool publish = !od->pub.empty();
  const bool want_out = last_corner || last_surf || publish;
  if (publish) {
    if (od->pub_cap[od->pub_next] < n_ls + n_lf + 1) {
      // the whole ring at once (a page-locked allocation takes milliseconds: not one per sweep while the ring fills); the
      // buffers that are views right now keep their memory until their turn comes
      for (size_t k = 0; k < od->pub.size(); ++k) {
        if (od->pub[k] && (int)k != od->pub_next) continue;
        int rc = reserve_pinned(od->pub[k], od->pub_cap[k], 2 * (n_ls + n_lf) + 1);
        if (rc) return rc;
      }
    }
    od->h_last = od->pub[od->pub_next];
  } else if (want_out) {
    int rc = reserve_pinned(od->h_own, od->h_own_cap, n_ls + n_lf + 1);
    if (rc) return rc;
    od->h_last = od->h_own;
  }
  int status = LSLAM_TOO_FEW_REF;
  int32_t matched = 0, searches = 0;
  const int from = od->cur, to = od->cur ^ 1;
  if (!od->inited) {  // :295-303: the clouds as they are; nothing is published
    int rc = enqueue_build(od, to, fs->list(1), n_ls, fs->list(3), n_lf, false, false);
    if (rc) return rc;
    if (want_out) {
      const Side &S = od->side[to];
      if (n_ls) OD_TRY(hipMemcpyAsync(od->h_last, S.org[0].p, n_ls * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
      if (n_lf) OD_TRY(hipMemcpyAsync(od->h_last + n_ls, S.org[1].p, n_lf * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
    }
    OD_TRY(hipStreamSynchronize(od->stream));
    od->inited = true;
  } else {
    TailSpec tail;
    tail.on = true;
    tail.to = to;
    tail.less_sharp = fs->list(1);
    tail.less_flat = fs->list(3);
    tail.n_ls = n_ls;
    tail.n_lf = n_lf;
    tail.copy_out = want_out;
    const Side &S = od->side[from];
    if (S.n[0] > 10 && S.n[1] > 100) {  // :337
      bool tie = false;
      float pose[6];
      for (int i = 0; i < 6; ++i) pose[i] = od->transform[i];
      int rc = match_loop(od, from, fs->list(0), n_sharp, fs->list(2), n_flat, pose, st, &searches, &tie, tail);
      if (rc) return rc;
      if (tie) {
        // an exact distance tie: the same match through kd-trees (nanoflann's own order), from the clouds' host copies
        std::vector<float4> lc(S.n[0]), ls(S.n[1]), sh(n_sharp), fl(n_flat);
        if (S.n[0]) OD_TRY(hipMemcpyAsync(lc.data(), S.org[0].p, S.n[0] * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
        if (S.n[1]) OD_TRY(hipMemcpyAsync(ls.data(), S.org[1].p, S.n[1] * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
        if (n_sharp) OD_TRY(hipMemcpyAsync(sh.data(), fs->list(0), n_sharp * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
        if (n_flat) OD_TRY(hipMemcpyAsync(fl.data(), fs->list(2), n_flat * sizeof(float4), hipMemcpyDeviceToHost, od->stream));
        OD_TRY(hipStreamSynchronize(od->stream));
        for (int i = 0; i < 6; ++i) pose[i] = od->transform[i];
        rc = lslam::odometry_match_trees(od->ctx, lc.data(), lc.size(), ls.data(), ls.size(), sh.data(), sh.size(), fl.data(), fl.size(), 16,
                                         pose, od->max_it, od->dt, od->dr, &st);
        if (rc < 0) return rc;
        od->tree_fallbacks++;
        // the next last clouds with THAT pose
        GNState *hs = od->h_state;
        std::memset(hs, 0, sizeof(GNState));
        for (int i = 0; i < 6; ++i) hs->pose[i] = pose[i];
        hs->done = 1;
        OD_TRY(hipMemcpyAsync(od->d_state, hs, sizeof(GNState), hipMemcpyHostToDevice, od->stream));
        rc = enqueue_tail(od, tail, false);
        if (rc) return rc;
        OD_TRY(hipStreamSynchronize(od->stream));
      }
      for (int i = 0; i < 6; ++i) od->transform[i] = pose[i];
      status = st.status;
      matched = 1;
    } else {  // nothing to match against: _transform stays, the clouds still move on (:305-316)
      GNState *hs = od->h_state;
      std::memset(hs, 0, sizeof(GNState));
      for (int i = 0; i < 6; ++i) hs->pose[i] = od->transform[i];
      hs->done = 1;
      OD_TRY(hipMemcpyAsync(od->d_state, hs, sizeof(GNState), hipMemcpyHostToDevice, od->stream));
      int rc = enqueue_tail(od, tail, false);
      if (rc) return rc;
      OD_TRY(hipStreamSynchronize(od->stream));
      st.status = LSLAM_TOO_FEW_REF;
    }
    float T[16], S2[16];
    lslam_pose_to_isometry(od->transform, T);  // transformUpdate, :649-653
    mat4_mul(od->Tsum, T, S2);
    std::memcpy(od->Tsum, S2, sizeof(S2));
  }
  od->cur = to;
  od->sweeps++;
  if (want_out) {
    if (last_corner && n_ls) std::memcpy(last_corner, od->h_last, n_ls * sizeof(float4));
    if (last_surf && n_lf) std::memcpy(last_surf, od->h_last + n_ls, n_lf * sizeof(float4));
  }
  if (publish) {
    od->view_c = od->h_last;
    od->view_s = od->h_last + n_ls;
    od->view_nc = n_ls;
    od->view_ns = n_lf;
    od->pub_next = (od->pub_next + 1) % (int)od->pub.size();
  }
  if (transform)
    for (int i = 0; i < 6; ++i) transform[i] = od->transform[i];
  if (Tsum) std::memcpy(Tsum, od->Tsum, sizeof(od->Tsum));
  if (ostats) {
    std::memset(ostats, 0, sizeof(*ostats));
    ostats->matched = matched;
    ostats->tree_fallbacks = od->tree_fallbacks;
    ostats->searches = searches;
    ostats->sweeps = od->sweeps;
    ostats->n_last_corner = od->side[od->cur].n[0];
    ostats->n_last_surf = od->side[od->cur].n[1];
  }
  return status;
}

int lslam_odometry_match_trees(lslam_ctx *ctx, const void *last_corner, size_t n_lc, const void *last_surf, size_t n_ls,
                               const void *sharp, size_t n_sharp, const void *flat, size_t n_flat, size_t stride_bytes, float pose[6],
                               int32_t max_iterations, float delta_t_abort, float delta_r_abort, lslam_stats *stats) {
  return lslam::odometry_match_trees(ctx, last_corner, n_lc, last_surf, n_ls, sharp, n_sharp, flat, n_flat, stride_bytes, pose,
                                     max_iterations, delta_t_abort, delta_r_abort, stats);
}

// Variant B from host clouds: LaserOdometry::scanMatch (odometry/LaserOdometry.cpp:328-647); include/lslam_c.h.
int lslam_odometry_match(lslam_ctx *ctx, const void *last_corner, size_t n_lc, const void *last_surf, size_t n_ls, const void *sharp,
                         size_t n_sharp, const void *flat, size_t n_flat, size_t stride_bytes, float pose[6], int32_t max_iterations,
                         float delta_t_abort, float delta_r_abort, lslam_stats *stats) {
  if (!ctx || !lslam::ctx_alive(ctx)) {
    lslam::set_error("no context");
    return LSLAM_ERR_INVALID;
  }
  if (lslam::env_once().odom_trees)  // A/B switch: the launch-per-step implementation over kd-trees
    return lslam::odometry_match_trees(ctx, last_corner, n_lc, last_surf, n_ls, sharp, n_sharp, flat, n_flat, stride_bytes, pose,
                                       max_iterations, delta_t_abort, delta_r_abort, stats);
  if (stride_bytes < 16 || (stride_bytes & 3) || !pose || (n_lc && !last_corner) || (n_ls && !last_surf) || (n_sharp && !sharp) ||
      (n_flat && !flat) || n_lc >= KD_MAX_POINTS || n_ls >= KD_MAX_POINTS || n_sharp > 0x0FFFFFFFu || n_flat > 0x0FFFFFFFu) {
    lslam::set_error("bad odometry arguments (clouds need x,y,z,intensity: stride >= 16)");
    return LSLAM_ERR_INVALID;
  }
  lslam_stats local;
  lslam_stats &st = stats ? *stats : local;
  std::memset(&st, 0, sizeof(st));
  if (!(n_lc > 10 && n_ls > 100)) {  // :337
    st.status = LSLAM_TOO_FEW_REF;
    return LSLAM_TOO_FEW_REF;
  }
  lslam_odom *od = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_hidden.find(ctx);
    if (it != g_hidden.end()) od = it->second;
  }
  if (!od) {
    int rc = od_create(ctx, max_iterations, delta_t_abort, delta_r_abort, &od);
    if (rc) return rc;
    rc = lslam_fset_create(ctx, &od->own_fs);
    if (rc) {
      od_free(od);
      return rc;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_hidden[ctx] = od;
  }
  OD_TRY(hipSetDevice(od->device));
  od->max_it = max_iterations < 0 ? 0 : max_iterations;
  od->dt = delta_t_abort;
  od->dr = delta_r_abort;
  // the four clouds into the node's own feature set (the last clouds in the less-sharp / less-flat slices), grids, loop
  const void *src[4] = {sharp, last_corner, flat, last_surf};
  const size_t n[4] = {n_sharp, n_lc, n_flat, n_ls};
  int rc = reserve_pinned(od->h_up, od->h_up_cap, n_sharp + n_lc + n_flat + n_ls + 1);
  if (rc) return rc;
  rc = upload_lists(od->stream, od->own_fs, od->h_up, src, n, stride_bytes);
  if (rc) return rc;
  lslam_fset *fs = od->own_fs;
  rc = enqueue_build(od, 0, fs->list(1), n_lc, fs->list(3), n_ls, false, false);
  if (rc) return rc;
  bool tie = false;
  float p[6];
  for (int i = 0; i < 6; ++i) p[i] = pose[i];
  TailSpec none;
  rc = match_loop(od, 0, fs->list(0), n_sharp, fs->list(2), n_flat, p, st, nullptr, &tie, none);
  if (rc) return rc;
  if (tie) {
    od->tree_fallbacks++;
    return lslam::odometry_match_trees(ctx, last_corner, n_lc, last_surf, n_ls, sharp, n_sharp, flat, n_flat, stride_bytes, pose,
                                       max_iterations, delta_t_abort, delta_r_abort, stats);
  }
  for (int i = 0; i < 6; ++i) pose[i] = p[i];
  return st.status;
}

}  // extern "C"
