// lslam_fmap.hip -- map maintenance in HBM: the cube grid of
// util/FeatureMap.h, addFeatureCloud with the per-cube pcl::VoxelGrid downsample, the active
// area, the surround concatenation handed straight to the kd-tree builder, and a stand-alone
// VoxelGrid (LaserMatcher.cpp:289-301, ScanMatch.cpp:362-398).
//
// Layout: per feature type ONE array of float4 {x, y, z, intensity} holding every cube's cloud
// back to back (cube-major, inside a cube in the reference's order), a parallel int32 array with
// the cube of each point, and a [cube] -> (begin, end) table.  addFeatureCloud appends the
// transformed scan and rebuilds the array with ONE stable radix sort on the key
//     (cube | voxel z | voxel y | voxel x)     voxel bits only for cubes that get filtered,
// which at once (i) appends each new point to its cube in push order, (ii) groups the points of a
// filtered cube by voxel in pcl::VoxelGrid's output order (ascending x + y*dx + z*dx*dy, which is
// the lexicographic (z, y, x) order whatever min_b/div_b are) with the members of a voxel in
// input order, and (iii) drops points outside the grid.  One thread per voxel then writes the
// centroid (sum in member order / count, all four fields).
//
// Which cubes are filtered: every cube of the active area, on every addFeatureCloud, exactly as
// FeatureMap.h:288-306 does (re-filtering an already filtered cube is almost always the identity,
// but a centroid that rounds onto a voxel boundary can move -- so nothing is skipped).  The sort
// covers all points anyway; the voxel bits of the key cost nothing extra.
//
// rocPRIM supplies the radix sort and the scan (plain library primitives); the rest is HIP.
#include "../../include/lslam_c.h"

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <chrono>
#include <vector>

#include "lslam_internal.hpp"

namespace {

#define FM_TRY(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      char _b[400];                                                                      \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      lslam::set_error(_b);                                                              \
      return LSLAM_ERR_HIP;                                                              \
    }                                                                                    \
  } while (0)

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
  // grow, keeping the first `keep` elements
  hipError_t grow(size_t n, size_t keep, hipStream_t s) {
    if (n <= cap) return hipSuccess;
    const size_t want = n + n / 4 + 256;
    T *q = nullptr;
    hipError_t e = hipMalloc((void **)&q, want * sizeof(T));
    if (e != hipSuccess) return e;
    if (keep && p) {
      e = hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (p) (void)hipFree(p);
    p = q;
    cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// a pinned host block that lives as long as its owner: asynchronous copies to and from it need no wait to keep their host side
// alive (a std::vector local does), and they run at the link's rate instead of through the runtime's staging
template <typename T>
struct Pin {
  T *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 256;
    hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

constexpr uint64_t KEY_DROP = ~0ull;
constexpr size_t MERGE_LIMIT = 1024 * 1024;  // rocprim::radix_sort_config<>::merge_sort_limit

// How a point becomes a sort key.
struct KeyParams {
  int32_t W, H, D;         // cube grid (1,1,1 for the stand-alone filter)
  int32_t origin[3];
  float cube_size;
  float inv_leaf;          // 1 / leaf (float, as pcl::VoxelGrid computes it)
  int32_t axis_bits;       // bits per voxel axis
  int32_t single;          // 1: stand-alone filter, every point is in cube 0, base = base0
  int32_t base0[3];
};

__device__ __forceinline__ int32_t cube_of_point(const KeyParams &k, float x, float y, float z, int32_t g[3]) {
  // FeatureMap.h:475-487: round(p / size) + origin, float arithmetic, truncated to int
  g[0] = (int32_t)(roundf(x / k.cube_size) + (float)k.origin[0]);
  g[1] = (int32_t)(roundf(y / k.cube_size) + (float)k.origin[1]);
  g[2] = (int32_t)(roundf(z / k.cube_size) + (float)k.origin[2]);
  if (g[0] < 0 || g[0] >= k.W || g[1] < 0 || g[1] >= k.H || g[2] < 0 || g[2] >= k.D) return -1;
  return g[0] + g[1] * k.W + g[2] * k.W * k.H;
}

// new points: p' = R p + t (pcl::transformPointCloud), cube of p'
struct Rigid12 { float m[12]; };  // rows of [R | t], a kernel argument (was a 64-byte upload per call)
__global__ void fm_transform_kernel(const float4 *in, int n, const Rigid12 Tm, KeyParams k, float4 *out,
                                    int32_t *cube_out, uint8_t *touched) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *T = Tm.m;
  const float4 p = in[i];
  float4 q;
  q.x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[0], p.x), __fmul_rn(T[1], p.y)), __fmul_rn(T[2], p.z)), T[3]);
  q.y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[4], p.x), __fmul_rn(T[5], p.y)), __fmul_rn(T[6], p.z)), T[7]);
  q.z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[8], p.x), __fmul_rn(T[9], p.y)), __fmul_rn(T[10], p.z)), T[11]);
  q.w = p.w;
  int32_t g[3];
  const int32_t c = cube_of_point(k, q.x, q.y, q.z, g);
  out[i] = q;
  cube_out[i] = c;
  if (touched && c >= 0) touched[c] = 1;  // this cube's cloud changes: its kd-tree (per-cube forest) is stale
}

// flags[c] != 0: cube c is filtered by this rebuild (active area; every cube for getFullMap);
// flags == nullptr: no cube is
__device__ __forceinline__ bool cube_filtered(const uint8_t *flags, int32_t c) {
  return flags != nullptr && flags[c] != 0;
}

__device__ __forceinline__ int32_t ordered_int(float f) {  // monotone float -> int map for atomicMin/Max
  const int32_t i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ordered_float(int32_t i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// getMinMax3D per filtered cube (pcl::VoxelGrid::applyFilter works on each cube's own cloud).
// Points arrive cube after cube: a wavefront walks MM_RUN contiguous points, keeps per-lane extremes
// for the cube its first point belongs to (points of another cube -- a run boundary -- go straight
// to the atomics), reduces once and issues six atomics.  One wavefront per 64 points doing that was
// 100 k atomics on 64 cache lines: 146 us for a million points, now ~15.
constexpr int MM_RUN = 256;  // points per wavefront
__global__ __launch_bounds__(256) void fm_minmax_kernel(const float4 *pts, const int32_t *cube, int n, const uint8_t *flags,
                                                        int32_t *cmin, int32_t *cmax) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int first = wave * MM_RUN;
  if (first >= n) return;
  // the run's data first, all chunks at once (cube -> flag -> point is a chain of three dependent loads: chunk after chunk it
  // was most of the kernel's time), then the bookkeeping on registers
  constexpr int K = MM_RUN / 64;
  int32_t cc[K];
  bool onn[K];
  float4 pp[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = first + k * 64 + lane;
    cc[k] = i < n ? cube[i] : -2;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) onn[k] = cc[k] >= 0 && flags[cc[k]];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = first + k * 64 + lane;
    pp[k] = onn[k] ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int32_t cur = -1;
  int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  auto flush = [&]() {
    if (cur < 0) return;  // wave-uniform
#pragma unroll
    for (int s = 32; s > 0; s >>= 1)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        lo[d] = min(lo[d], __shfl_xor(lo[d], s, 64));
        hi[d] = max(hi[d], __shfl_xor(hi[d], s, 64));
      }
    if (lane == 0)  // six atomics in flight together (a compare first would be six dependent loads in front of them)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        atomicMin(&cmin[3 * cur + d], lo[d]);
        atomicMax(&cmax[3 * cur + d], hi[d]);
      }
#pragma unroll
    for (int d = 0; d < 3; ++d) { lo[d] = INT32_MAX; hi[d] = INT32_MIN; }
  };
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int base = first + k * 64;
    if (base >= n) break;  // wave-uniform
    const int i = base + lane;
    const bool in = i < n;
    const int32_t c = cc[k];
    const bool on = onn[k];
    const float4 p = pp[k];
    const int32_t o[3] = {ordered_int(p.x), ordered_int(p.y), ordered_int(p.z)};
    // the cube of the chunk's first lane; uniform chunk: every in-range lane has it
    const int32_t c_first = __builtin_amdgcn_readfirstlane(c);
    const bool uniform = __ballot(in && c != c_first) == 0ull;
    if (uniform) {
      const bool first_on = __builtin_amdgcn_readfirstlane((int)on) != 0;  // (the first lane is in range and has c_first)
      if (c_first != cur) {
        flush();
        cur = first_on ? c_first : -1;
      }
      if (on) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { lo[d] = min(lo[d], o[d]); hi[d] = max(hi[d], o[d]); }
      }
    } else if (on) {
      if (c == cur) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { lo[d] = min(lo[d], o[d]); hi[d] = max(hi[d], o[d]); }
      } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          atomicMin(&cmin[3 * c + d], o[d]);
          atomicMax(&cmax[3 * c + d], o[d]);
        }
      }
    }
  }
  flush();
}

// the per-rebuild initial values in one launch: cube extremes, error words
__global__ void fm_init_kernel(int32_t *cmin, int32_t *cmax, int n3, int32_t *err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 4) err[i] = 0;
  if (i < n3) {
    cmin[i] = 0x7f7f7f7f;           // +large (what the byte fill 0x7f wrote)
    cmax[i] = (int32_t)0x80808080;  // -large
  }
}

// The same per-cube quantities WITHOUT looking at the points (the rebuilds of addFeatureCloud: 744 k points of the bench map
// were read for their per-cube extremes, 0.13 ms per frame, to place ~3 000 new ones).  What pcl::VoxelGrid does with a
// cube's bounding box is (a) min_b = floor(min x inv_leaf), subtracted from every voxel index, and (b) the "leaf too small"
// guard.  (a) does not change the result: voxels are cut at absolute multiples of the leaf, and both the grouping (equal
// indices) and the output order (ascending index = lexicographic in (k, j, i)) are invariant under subtracting ANY constant
// per axis that is <= every index of the cube.  A map cube's points lie within cube_size / 2 of its centre on every axis
// (cube_of_point rounds p / cube_size; a voxel centroid lies inside its voxel's points' box), so a base a cell below the
// cube's nominal lower face is such a constant, and the extent is bounded by cube_size / leaf + 6 cells.  (b) cannot fire when
// (cube_size / leaf + 2)^3 <= INT_MAX, which the caller checks; every cube the rebuild filters is then "effective".  The
// key-range check of fm_key_kernel still guards the bound (a violation re-runs the measured path).
__global__ void fm_base_kernel(const uint8_t *flags, int ncube, KeyParams k, uint8_t *eff, int32_t *base, int32_t *err) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < 4) err[c] = 0;
  if (c >= ncube) return;
  const int g[3] = {c % k.W, (c / k.W) % k.H, c / (k.W * k.H)};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double lo = ((double)(g[d] - k.origin[d]) - 0.5) * (double)k.cube_size - 0.01;
    base[3 * c + d] = (int32_t)floor(lo * (double)k.inv_leaf) - 1;
  }
  eff[c] = flags[c] ? 1 : 0;
}

// per cube: min_b, the "leaf too small" guard of applyFilter, the widest voxel extent
__global__ void fm_extent_kernel(const uint8_t *flags, const int32_t *cmin, const int32_t *cmax, int ncube,
                                 float inv_leaf, uint8_t *eff, int32_t *base, int32_t *max_div) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncube) return;
  uint8_t e = 0;
  if (flags[c] && cmin[3 * c] <= cmax[3 * c]) {  // filtered and not empty
    long long vol = 1;
    int32_t div = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float mn = ordered_float(cmin[3 * c + d]), mx = ordered_float(cmax[3 * c + d]);
      vol *= (long long)(__fmul_rn(__fsub_rn(mx, mn), inv_leaf)) + 1;
      const int32_t b0 = (int32_t)floorf(__fmul_rn(mn, inv_leaf)), b1 = (int32_t)floorf(__fmul_rn(mx, inv_leaf));
      base[3 * c + d] = b0;
      div = max(div, b1 - b0 + 1);
    }
    if (vol <= 2147483647ll) {  // else: PCL warns and hands the cloud back unfiltered
      e = 1;
      atomicMax(max_div, div);
    }
  }
  eff[c] = e;
}

__global__ void fm_key_kernel(const float4 *pts, const int32_t *cube, int n, KeyParams k, const uint8_t *flags,
                              const int32_t *cube_base, uint64_t *keys, uint32_t *idx, int32_t *err,
                              uint64_t *merge_keys = nullptr, uint32_t *merge_idx = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idx[i] = (uint32_t)i;
  if (merge_keys) {  // what a slot of the merged sequence holds until its element arrives (see run_pipeline)
    merge_keys[i] = KEY_DROP;
    merge_idx[i] = 0u;
  }
  const int32_t c = k.single ? 0 : cube[i];
  if (c < 0) {
    keys[i] = KEY_DROP;
    return;
  }
  uint64_t key = (uint64_t)c;
  const int ab = k.axis_bits;
  uint64_t vox = 0;
  if (k.single || cube_filtered(flags, c)) {
    const float4 p = pts[i];
    int32_t base[3];
    if (k.single) {
      base[0] = k.base0[0]; base[1] = k.base0[1]; base[2] = k.base0[2];
    } else {
      base[0] = cube_base[3 * c]; base[1] = cube_base[3 * c + 1]; base[2] = cube_base[3 * c + 2];
    }
    const float v[3] = {p.x, p.y, p.z};
    int32_t r[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      r[d] = (int32_t)floorf(__fmul_rn(v[d], k.inv_leaf)) - base[d];
      if (r[d] < 0 || r[d] >= (1 << ab)) {
        atomicExch(err, 1);
        r[d] = r[d] < 0 ? 0 : (1 << ab) - 1;
      }
    }
    vox = ((uint64_t)r[2] << (2 * ab)) | ((uint64_t)r[1] << ab) | (uint64_t)r[0];
  }
  keys[i] = (key << (3 * ab)) | vox;
}

// [n_sorted keys already in order | n_new keys, sorted separately into kn / in] -> one sorted sequence, old before new among
// equal keys (what a stable sort of the concatenation gives).  Every element finds its place by one binary search in the
// other list.  A prefix that is not in order after all raises *unsorted (the caller sorts everything instead).
__global__ void fm_merge_kernel(const uint64_t *k0, int n_sorted, const uint64_t *kn, const uint32_t *in, int n_new, uint64_t *k1,
                                uint32_t *i1, int32_t *unsorted) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_sorted + n_new) return;
  if (t < n_sorted) {
    const uint64_t key = k0[t];
    if (t > 0 && k0[t - 1] > key) atomicExch(unsorted, 1);
    int lo = 0, hi = n_new;  // new keys < key
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (kn[mid] < key) lo = mid + 1; else hi = mid;
    }
    k1[t + lo] = key;
    i1[t + lo] = (uint32_t)t;
  } else {
    const int j = t - n_sorted;
    const uint64_t key = kn[j];
    int lo = 0, hi = n_sorted;  // old keys <= key
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (k0[mid] <= key) lo = mid + 1; else hi = mid;
    }
    k1[j + lo] = key;
    i1[j + lo] = in[j];
  }
}

// 1 where sorted entry i starts a voxel (or is a point of a cube that is not filtered: its own output), 0 for its further
// members, for dropped points and for i = n (the scan runs over n + 1 values: its last output is the number of points out).
// A functor, not an array: the scan reads it through a transform iterator and the centroid kernel evaluates it where it needs
// it (round 5: a kernel of its own wrote these n + 1 words for the two to read back).
struct HeadOf {
  const uint64_t *keys;
  int n, axis_bits, single;
  const uint8_t *flags;
  __host__ __device__ uint32_t operator()(uint32_t iu) const {
    const int i = (int)iu;
    if (i >= n) return 0u;
    const uint64_t k = keys[i];
    if (k == KEY_DROP) return 0u;
    const int32_t c = (int32_t)(k >> (3 * axis_bits));
    const bool filt = single || (flags != nullptr && flags[c] != 0);
    return (i == 0 || !filt || keys[i - 1] != k) ? 1u : 0u;
  }
};

// Centroid of every voxel over its members in sorted (= input) order.  The sums are sequential by definition (PCL's order), the
// loads are not: every lane fetches ITS point (64 independent gathers per wavefront), then each head lane adds its members'
// values, in order, out of its neighbours' registers; only a voxel that runs past the end of its wavefront goes on from memory.
__global__ __launch_bounds__(256) void fm_centroid_kernel(const float4 *pts, const uint64_t *keys, const uint32_t *idx, const HeadOf head,
                                                          const uint32_t *pos, int n, int axis_bits, float4 *out, int32_t *cube_out,
                                                          int32_t *total_out) {
  const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const bool in = i < n;
  if (i == 0) *total_out = (int32_t)pos[n];  // the number of heads = points out, next to the error words (one copy fetches all)
  const uint64_t k = in ? keys[i] : KEY_DROP;
  const bool is_head = in && head((uint32_t)i) != 0u;
  const bool member = in && !is_head && k != KEY_DROP;  // (a dropped point is nobody's member: fm_head_kernel gives it no head either)
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (is_head || member) p = pts[min(idx[i], (uint32_t)(n - 1))];  // (clamped: after a merge whose prefix was not in order the slots are
                                                                       // not a permutation -- the result is discarded, the loads must stay inside)
  const unsigned long long mm = __ballot(member);
  // members that follow this lane inside the wavefront
  const unsigned long long after = lane == 63 ? 0ull : ~(mm >> (lane + 1));
  const int len = is_head ? (lane == 63 ? 0 : (after == 0ull ? 63 - lane : min(__builtin_ctzll(after), 63 - lane))) : 0;
  int longest = len;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) longest = max(longest, __shfl_xor(longest, d, 64));
  float sx = p.x, sy = p.y, sz = p.z, sw = p.w;
  for (int m = 1; m <= longest; ++m) {
    const int srcl = min(lane + m, 63);
    const float qx = __shfl(p.x, srcl, 64), qy = __shfl(p.y, srcl, 64), qz = __shfl(p.z, srcl, 64), qw = __shfl(p.w, srcl, 64);
    if (m <= len) {
      sx = __fadd_rn(sx, qx);
      sy = __fadd_rn(sy, qy);
      sz = __fadd_rn(sz, qz);
      sw = __fadd_rn(sw, qw);
    }
  }
  int j = i + 1 + len;
  // a run that reached the end of the wavefront may go on (at most one per wavefront: its last): the whole wavefront fetches
  // the next 64 entries for it, the owner adds them in order
  const unsigned long long cm = __ballot(is_head && lane + len == 63);
  if (cm) {  // wave-uniform
    const int owner = __builtin_ctzll(cm);
    const uint32_t k_lo = (uint32_t)__shfl((int)(uint32_t)k, owner, 64), k_hi = (uint32_t)__shfl((int)(uint32_t)(k >> 32), owner, 64);
    const uint64_t kk = ((uint64_t)k_hi << 32) | k_lo;
    int e0 = (i - lane) + 64;
    for (;;) {
      const int e = e0 + lane;
      const bool valid = e < n && !head((uint32_t)e) && keys[e] == kk;
      const unsigned long long vm = __ballot(valid);
      const int L = ~vm == 0ull ? 64 : __builtin_ctzll(~vm);  // the run's members at the head of this chunk
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < L) q = pts[min(idx[e], (uint32_t)(n - 1))];
      for (int m = 0; m < L; ++m) {
        const float qx = __shfl(q.x, m, 64), qy = __shfl(q.y, m, 64), qz = __shfl(q.z, m, 64), qw = __shfl(q.w, m, 64);
        if (lane == owner) {
          sx = __fadd_rn(sx, qx);
          sy = __fadd_rn(sy, qy);
          sz = __fadd_rn(sz, qz);
          sw = __fadd_rn(sw, qw);
        }
      }
      if (lane == owner) j += L;
      if (L < 64) break;
      e0 += 64;
    }
  }
  if (!is_head) return;
  const float cnt = (float)(j - i);
  // PCL starts from a zero vector: 0 + x = x, so the sums above equal its sums
  out[pos[i]] = make_float4(__fdiv_rn(sx, cnt), __fdiv_rn(sy, cnt), __fdiv_rn(sz, cnt), __fdiv_rn(sw, cnt));
  cube_out[pos[i]] = (int32_t)(k >> (3 * axis_bits));
}

__global__ void fm_segment_kernel(const int32_t *cube, int n, int32_t *begin, int32_t *end) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t c = cube[i];
  if (i == 0 || cube[i - 1] != c) begin[c] = i;
  if (i == n - 1 || cube[i + 1] != c) end[c] = i + 1;
}

// points of flagged cubes are dropped (a cube loaded from a file replaces what was there)
__global__ void fm_two_segments_kernel(int32_t *seg, int na, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) seg[i] = i < na ? 0 : 1;
}
__global__ void fm_dropflag_kernel(int32_t *cube, int n, const uint8_t *flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && cube[i] >= 0 && flags[cube[i]]) cube[i] = -1;
}

__global__ void fm_relabel_kernel(int32_t *cube, int n, const int32_t *new_of_old) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && cube[i] >= 0) cube[i] = new_of_old[cube[i]];
}

// surround: concatenation of the active cubes' segments; seg = {src begin, dst begin} per cube
__global__ void fm_gather_kernel(const float4 *pts, const int32_t *src_begin, const int32_t *dst_begin, int n_seg,
                                 int n_out, int index_in_w, float4 *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  int lo = 0, hi = n_seg - 1;  // last segment with dst_begin <= i
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (dst_begin[mid] <= i) lo = mid; else hi = mid - 1;
  }
  float4 p = pts[src_begin[lo] + (i - dst_begin[lo])];
  if (index_in_w == 1) p.w = __builtin_bit_cast(float, (uint32_t)i);
  else if (index_in_w == 2) p.w = __builtin_bit_cast(float, (uint32_t)(i - dst_begin[lo]));  // index inside the cube
  out[i] = p;
}

// ---- the surround of lslam_fmap_surround_to_map, without the host in between (round 5) ------------------------------------
// The host used to fetch both feature types' segment tables (a wait), walk the active cubes, upload two tables per type and,
// in the map set that followed, run the bounding boxes as launches of their own behind an upload of their initial values.  Now
// one workgroup turns the device's segment table into the gather's tables and the total, and the gather takes the bounding
// box along: a fill, the segments, a plan and a gather per type, one 64-byte copy and one wait for both.
__global__ __launch_bounds__(256) void fm_surround_plan_kernel(const int32_t *valid, int n_valid, const int32_t *seg, int pad, int32_t *g_src,
                                                               int32_t *g_dst, uint32_t *res) {
  __shared__ int part[256];
  const int tid = threadIdx.x;
  const int per = (n_valid + 255) / 256;
  const int v0 = min(n_valid, tid * per), v1 = min(n_valid, v0 + per);
  int sum = 0;
  for (int v = v0; v < v1; ++v) sum += seg[pad + valid[v]] - seg[valid[v]];
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int u = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += u;
    __syncthreads();
  }
  int run = part[tid] - sum;
  for (int v = v0; v < v1; ++v) {
    const int32_t c = valid[v];
    g_src[v] = seg[c];
    g_dst[v] = run;  // (an empty cube shares its place with the next one: the gather's search takes the last of equals)
    run += seg[pad + c] - seg[c];
  }
  if (tid == 255) res[0] = (uint32_t)part[255];
  if (tid < 7) res[1 + tid] = 0u;  // the box: maxima of ~ordered(x) (the minimum) and of ordered(x)
}

__device__ __forceinline__ uint32_t fm_ordered_u32(float f) {  // monotone map float -> uint32 (lslam_grid.hip's)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// res[0] = points to gather; res[1..3] / res[4..6]: the box (see above)
__global__ __launch_bounds__(256) void fm_gather_box_kernel(const float4 *pts, const int32_t *src_begin, const int32_t *dst_begin, int n_seg,
                                                            uint32_t *res, int index_in_w, float4 *out) {
  const int total = (int)res[0];
  uint32_t mm[6] = {0u, 0u, 0u, 0u, 0u, 0u};
  bool any = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int lo = 0, hi = n_seg - 1;  // last segment with dst_begin <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (dst_begin[mid] <= i) lo = mid; else hi = mid - 1;
    }
    float4 p = pts[src_begin[lo] + (i - dst_begin[lo])];
    if (index_in_w == 1) p.w = __builtin_bit_cast(float, (uint32_t)i);
    out[i] = p;
    const uint32_t o[3] = {fm_ordered_u32(p.x), fm_ordered_u32(p.y), fm_ordered_u32(p.z)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mm[a] = max(mm[a], ~o[a]);
      mm[3 + a] = max(mm[3 + a], o[a]);
    }
    any = true;
  }
  if (!__syncthreads_or(any)) return;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1)
#pragma unroll
    for (int a = 0; a < 6; ++a) mm[a] = max(mm[a], (uint32_t)__shfl_xor((int)mm[a], d, 64));
  __shared__ uint32_t part[4][6];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 6; ++a) part[wave][a] = mm[a];
  }
  __syncthreads();
  if (threadIdx.x < 6) {  // six atomics per workgroup (same-address atomics from a whole grid serialise at the memory side)
    const int a = threadIdx.x;
    atomicMax(&res[1 + a], max(max(part[0][a], part[1][a]), max(part[2][a], part[3][a])));
  }
}

int bits_for(double cells) {
  int b = 1;
  while ((double)(1 << b) < cells) ++b;
  return b;
}

struct Scratch {
  Buf<uint64_t> k0, k1, kn;
  Buf<uint32_t> i0, i1, in_, pos;
  Buf<char> tmp;
  Buf<int32_t> err;               // [0] points out, [1] key-range error, [2] the "sorted" prefix was not, [3] widest voxel extent of a filtered cube
  Buf<int32_t> cmin, cmax, base;  // [ncube][3]
  Buf<uint8_t> eff;               // [ncube] cubes this rebuild really filters
  void release() {
    k0.release(); k1.release(); kn.release(); i0.release(); i1.release(); in_.release(); pos.release(); tmp.release();
    err.release(); cmin.release(); cmax.release(); base.release(); eff.release();
  }
};

}  // namespace

struct lslam_fmap {
  lslam_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  int W = 0, H = 0, D = 0, ncube = 0;
  int origin[3] = {0, 0, 0};
  int cur[3] = {0, 0, 0};
  float cube_size = 50.0f, valid_dist = 150.0f;
  float leaf[3] = {0.2f, 0.2f, 0.6f};  // corner, surf, map (FeatureMap.h:64-66)
  std::vector<int32_t> valid;          // _cubeValidInd
  // per type
  Buf<float4> pts[2], pts_alt[2];
  Buf<int32_t> cube[2], cube_alt[2];
  size_t n[2] = {0, 0};
  Buf<uint8_t> active;                 // per cube: 1 = in the active area (_cubeValidInd)
  Buf<int32_t> seg_begin[2];            // [begin (seg_pad words) | end (seg_pad words)] of every cube's points in pts[t]
  std::vector<int32_t> h_begin[2];      // ... on the host: seg_b / seg_e
  bool seg_current[2] = {false, false};
  bool seg_dev_current[2] = {false, false};  // ... the device's table alone (what surround_to_map needs)
  Buf<int32_t> d_valid;                 // the active cubes (fm->valid), uploaded when they change
  std::vector<int32_t> valid_uploaded;  // ... what the device holds (flags in `active`, list in d_valid)
  bool valid_on_device = false;
  Buf<uint32_t> sur_res;                // [2 types][8]: points gathered, bounding box (fm_gather_box_kernel)
  Buf<float4> in_raw, in_tf;
  Buf<int32_t> in_cube;
  // addFeatureCloud runs both feature types behind ONE wait: per-type staging of the new points (pinned: the upload needs no
  // wait to keep its source alive), per-type transformed points, and the rebuilds' results {points out, error} in pinned slots
  Pin<float4> in_pin[2];
  Buf<float4> in_raw_t[2];
  Buf<uint8_t> d_touched_t[2];
  Pin<uint32_t> done;       // [2 types][2]
  Pin<uint8_t> h_touched;   // [2 types][ncube]
  Buf<int32_t> d_remap;
  Buf<int32_t> g_src, g_dst;
  // surround gather of the two feature types behind ONE wait: per-type tables, their host copies kept alive here until the
  // next gather (a pageable hipMemcpyAsync source must outlive the copy)
  Buf<int32_t> g_src_t[2], g_dst_t[2];
  std::vector<int32_t> h_gsrc[2], h_gdst[2];
  Buf<float4> sur[2];
  Scratch sc;
  // addFeatureCloud's two feature types are independent chains of a dozen small launches each: the surf chain runs on the
  // context's second stream beside the corner chain (fork / join by events on the context's stream; LSLAM_FMAP_ONE_STREAM=1: one after
  // the other as before), with a scratch of its own
  Scratch sc2;
  // lslam_fmap_add_feature_cloud_begin: everything enqueued, nothing waited for -- the wait and the commit (finish_add) happen at
  // the head of the next call on this map (check_fm), so a mapping node's next frame is being prepared while the map is rebuilt
  bool add_pending = false;
  size_t add_cnt[2] = {0, 0};
  bool add_track = false;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // ---- per-cube kd-trees kept between frames (lslam_fmap_to_cubemap; FeatureMap.h:438,453 _kdtreeCorner/_kdtreeSurf) ----
  // A cube's tree is rebuilt only when its cloud changed: addFeatureCloud marks the cubes that received points
  // (the VoxelGrid re-filter of an untouched cube reproduces its cloud bit for bit), shifts and loads mark all.
  // Trees built together share one node / point array ("generation"); a generation is freed when none of its trees
  // is current any more.
  struct Generation {
    Buf<lslam::KdNode> nodes;
    Buf<lslam::PNode> pn;
    Buf<float4> pts;
    int live = 0, depth = 0;
  };
  struct CubeTree {
    int gen = -1;           // index into gens (-1: no tree)
    lslam::TreeView view{};
  };
  std::vector<Generation *> gens;       // one list for both feature types: a generation holds corner AND surf trees
  std::vector<Generation *> spare;      // dead generations kept with their allocations: no hipMalloc / hipFree per frame
  std::vector<CubeTree> cube_tree[2];   // [ncube]
  std::vector<uint8_t> dirty[2];        // [ncube]
  Buf<uint8_t> d_touched;               // [ncube] set by the insert kernel
  Buf<int32_t> d_cells[2];
  Buf<lslam::TreeView> d_views[2];
  bool cube_trees_used = false;               // lslam_fmap_to_cubemap has been called: addFeatureCloud keeps the cubes' marks
  int64_t trees_built = 0, trees_reused = 0;  // statistics of the last lslam_fmap_to_cubemap
  int64_t merged_rebuilds = 0, resorted_rebuilds = 0;  // addFeatureCloud rebuilds that merged the new points in / that had to sort everything after all
  int forest_attempt0 = 0;                     // node-slot guess the last forest build succeeded with (lslam_fmap_to_cubemap)
};

namespace {

size_t seg_pad(const lslam_fmap *fm) { return ((size_t)fm->ncube + 3) & ~(size_t)3; }
int32_t seg_b(const lslam_fmap *fm, int t, int c) { return fm->h_begin[t][(size_t)c]; }
int32_t seg_e(const lslam_fmap *fm, int t, int c) { return fm->h_begin[t][seg_pad(fm) + (size_t)c]; }

KeyParams key_params(const lslam_fmap *fm, float leaf) {
  KeyParams k{};
  k.W = fm->W; k.H = fm->H; k.D = fm->D;
  for (int d = 0; d < 3; ++d) k.origin[d] = fm->origin[d];
  k.cube_size = fm->cube_size;
  k.inv_leaf = 1.0f / leaf;
  k.axis_bits = 1;  // set by run_pipeline from the cubes' real extents
  k.single = 0;
  return k;
}

int cube_bits(int ncube) {
  int b = 1;
  while ((1ll << b) < (long long)ncube) ++b;
  return b;
}

// The rebuild: [old points | extra points] -> sorted, filtered, compacted arrays.
// in_pts/in_cube hold n_total points (already concatenated).  Writes out_pts/out_cube, returns
// the number of output points.  flags selects the cubes to filter (nullptr: none); kp.single: the
// stand-alone filter (one pseudo cube, base and bits prepared by the caller).
// assume_axis_bits > 0: the caller knows a bound of the voxel extent of every filtered cube (a map cube is cube_size wide), so
// the widest extent is not read back (one host wait less); the key-range check of fm_key_kernel still guards it.  Only taken
// while the sort is rocPRIM's merge sort, whose cost does not grow with the key width.
// done != nullptr: no wait at the end either -- {points out, key-range error, prefix-not-sorted} land in done[0..2] (pinned)
// behind everything else on the stream; *n_out is not written.
// n_sorted > 0: the first n_sorted points are the output of an earlier rebuild, i.e. in key order already (a centroid lies in
// its voxel) -- only the rest is sorted and the two are merged: one search per element instead of a sort of everything.
// Checked on the way (a centroid CAN round onto a voxel wall, and a cube that has just become active gets voxel keys it did
// not have): if the prefix is not in order the whole input is sorted after all.
int run_pipeline(hipStream_t s, Scratch &sc, const float4 *in_pts, const int32_t *in_cube, size_t n_total,
                 KeyParams kp, int ncube, const uint8_t *flags, float4 *out_pts, int32_t *out_cube, size_t *n_out,
                 int assume_axis_bits = 0, uint32_t *done = nullptr, size_t n_sorted = 0, bool cubes_are_boxes = false) {
  *n_out = 0;
  if (n_total == 0) {
    if (done) done[0] = done[1] = done[2] = 0;
    return LSLAM_OK;
  }
  const int n = (int)n_total;
  const int n_cube_bits = kp.single ? 1 : cube_bits(ncube + 1);
  const dim3 blk(256), grd((n + 255) / 256);
  FM_TRY(sc.err.reserve(4));
  const uint8_t *eff = nullptr;
  if (!kp.single && flags) {
    FM_TRY(sc.cmin.reserve(3 * (size_t)ncube));
    FM_TRY(sc.cmax.reserve(3 * (size_t)ncube));
    FM_TRY(sc.base.reserve(3 * (size_t)ncube));
    FM_TRY(sc.eff.reserve(ncube));
    // a map cube's extent is known without reading the points (fm_base_kernel) whenever the caller passes its bound and
    // pcl's "leaf too small" guard cannot fire for an extent of cube_size
    const double cells = (double)kp.cube_size * (double)kp.inv_leaf;
    const int analytic_bits = bits_for(cells + 6.0);
    // (cubes_are_boxes: the "cubes" are the feature map's cubes of cube_size -- not the segments of voxel_filter_segments)
    const bool analytic = cubes_are_boxes && assume_axis_bits > 0 && n_total <= MERGE_LIMIT && (cells + 2.0) * (cells + 2.0) * (cells + 2.0) <= 2147483647.0 &&
                          3 * analytic_bits + n_cube_bits <= 63 && !lslam::env_once().fmap_measured_extents;
    if (analytic) {
      hipLaunchKernelGGL(fm_base_kernel, dim3((std::max(ncube, 4) + 255) / 256), blk, 0, s, flags, ncube, kp, sc.eff.p, sc.base.p, sc.err.p);
      kp.axis_bits = analytic_bits;
      eff = sc.eff.p;
    } else {
    hipLaunchKernelGGL(fm_init_kernel, dim3((3 * ncube + 255) / 256), blk, 0, s, sc.cmin.p, sc.cmax.p, 3 * ncube, sc.err.p);
    {
      const int waves = (int)((n + MM_RUN - 1) / MM_RUN);
      hipLaunchKernelGGL(fm_minmax_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, in_pts, in_cube, (int)n, flags,
                         sc.cmin.p, sc.cmax.p);
    }
    hipLaunchKernelGGL(fm_extent_kernel, dim3((ncube + 255) / 256), blk, 0, s, flags, sc.cmin.p, sc.cmax.p, ncube,
                       kp.inv_leaf, sc.eff.p, sc.base.p, sc.err.p + 3);
    if (assume_axis_bits > 0 && n_total <= MERGE_LIMIT && 3 * assume_axis_bits + n_cube_bits <= 63) {
      kp.axis_bits = assume_axis_bits;
    } else {
      int32_t max_div = 0;
      FM_TRY(hipMemcpyAsync(&max_div, sc.err.p + 3, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      FM_TRY(hipStreamSynchronize(s));
      kp.axis_bits = bits_for((double)max_div + 1.0);
    }
    eff = sc.eff.p;
    }
  } else {
    FM_TRY(hipMemsetAsync(sc.err.p, 0, 4 * sizeof(int32_t), s));
    if (!kp.single) kp.axis_bits = 1;
  }
  if (3 * kp.axis_bits + n_cube_bits > 63) {
    lslam::set_error("voxel grid too fine for the 63-bit sort key (leaf too small for this extent)");
    return LSLAM_ERR_INVALID;
  }
  FM_TRY(sc.k0.reserve(n_total));
  FM_TRY(sc.k1.reserve(n_total));
  FM_TRY(sc.i0.reserve(n_total));
  FM_TRY(sc.i1.reserve(n_total));
  FM_TRY(sc.pos.reserve(n_total + 1));
  const bool merge = n_sorted > 0 && n_sorted < n_total;
  hipLaunchKernelGGL(fm_key_kernel, grd, blk, 0, s, in_pts, in_cube, n, kp, eff, sc.base.p, sc.k0.p, sc.i0.p, sc.err.p + 1,
                     merge ? sc.k1.p : (uint64_t *)nullptr, merge ? sc.i1.p : (uint32_t *)nullptr);
  const unsigned end_bit = (unsigned)(3 * kp.axis_bits + n_cube_bits);
  // KEY_DROP has every bit set: it is the largest key, so dropped points sort last.  rocPRIM's radix_sort_pairs (below
  // MERGE_LIMIT keys a merge sort whose cost does not depend on end_bit; measured against its onesweep form on the 64-bit keys
  // here: 60 us against 125 for 157 k keys, 136 against 141 for 587 k); LSLAM_SMALL_SORT=1: lslam_sort.hip for a frame's sorts
  auto sort_pairs = [&](const uint64_t *ki, uint64_t *ko, const uint32_t *vi, uint32_t *vo, size_t cnt, size_t lib_tmp) -> int {
    if (cnt <= lslam::SMALL_SORT_MAX && lslam::env_once().small_sort) {
      FM_TRY(lslam::small_sort_pairs(s, ki, ko, vi, vo, cnt, (void *)sc.tmp.p));
    } else {
      FM_TRY(rocprim::radix_sort_pairs((void *)sc.tmp.p, lib_tmp, ki, ko, vi, vo, cnt, 0u, end_bit, s));
    }
    return LSLAM_OK;
  };
  size_t tmp_bytes = 0;
  FM_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, sc.k0.p, sc.k1.p, sc.i0.p, sc.i1.p, n_total, 0u, end_bit, s));
  size_t tmp2 = 0;
  const HeadOf head_of{sc.k1.p, n, kp.axis_bits, kp.single, eff};
  auto heads = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0u), head_of);
  FM_TRY(rocprim::exclusive_scan(nullptr, tmp2, heads, sc.pos.p, 0u, n_total + 1, rocprim::plus<uint32_t>(), s));
  if (merge) {
    const size_t n_new = n_total - n_sorted;
    size_t tmp3 = 0;
    FM_TRY(sc.kn.reserve(n_new));
    FM_TRY(sc.in_.reserve(n_new));
    FM_TRY(rocprim::radix_sort_pairs(nullptr, tmp3, sc.k0.p + n_sorted, sc.kn.p, sc.i0.p + n_sorted, sc.in_.p, n_new, 0u, end_bit, s));
    FM_TRY(sc.tmp.reserve(std::max(std::max(std::max(tmp_bytes, tmp2), tmp3), lslam::small_sort_tmp_bytes(n_new))));
    {
      const int rc_sort = sort_pairs(sc.k0.p + n_sorted, sc.kn.p, sc.i0.p + n_sorted, sc.in_.p, n_new, tmp3);
      if (rc_sort) return rc_sort;
    }
    // (if the prefix turns out not to be in order the places below are not a permutation: slots nobody writes must hold
    // something the kernels after this one can digest until the host sees the flag -- a dropped key, index 0)
    // -- written by fm_key_kernel (merge_defaults)
    hipLaunchKernelGGL(fm_merge_kernel, grd, blk, 0, s, (const uint64_t *)sc.k0.p, (int)n_sorted, (const uint64_t *)sc.kn.p,
                       (const uint32_t *)sc.in_.p, (int)n_new, sc.k1.p, sc.i1.p, sc.err.p + 2);
  } else {
    FM_TRY(sc.tmp.reserve(std::max(std::max(tmp_bytes, tmp2), lslam::small_sort_tmp_bytes(n_total))));
    const int rc_sort = sort_pairs(sc.k0.p, sc.k1.p, sc.i0.p, sc.i1.p, n_total, tmp_bytes);
    if (rc_sort) return rc_sort;
  }
  FM_TRY(rocprim::exclusive_scan((void *)sc.tmp.p, tmp2, heads, sc.pos.p, 0u, n_total + 1, rocprim::plus<uint32_t>(), s));
  hipLaunchKernelGGL(fm_centroid_kernel, grd, blk, 0, s, in_pts, sc.k1.p, sc.i1.p, head_of, sc.pos.p, n,
                     kp.axis_bits, out_pts, out_cube, sc.err.p);
  if (done) {  // {points out, key-range error, prefix not sorted}: three adjacent words, one copy
    FM_TRY(hipMemcpyAsync(done, sc.err.p, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return LSLAM_OK;
  }
  int32_t res[3] = {0, 0, 0};
  FM_TRY(hipMemcpyAsync(res, sc.err.p, sizeof(res), hipMemcpyDeviceToHost, s));
  FM_TRY(hipStreamSynchronize(s));
  const uint32_t total = (uint32_t)res[0];
  const int32_t err = res[1], unsorted = res[2];
  if (unsorted)  // the prefix was not in key order: everything is sorted
    return run_pipeline(s, sc, in_pts, in_cube, n_total, kp, ncube, flags, out_pts, out_cube, n_out, assume_axis_bits, nullptr, 0, cubes_are_boxes);
  if (err) {
    if (assume_axis_bits > 0)  // the bound did not hold (it always should): the measured extent decides
      return run_pipeline(s, sc, in_pts, in_cube, n_total, kp, ncube, flags, out_pts, out_cube, n_out, 0, nullptr, 0);
    lslam::set_error("voxel index outside its key range (non-finite point?)");
    return LSLAM_ERR_INVALID;
  }
  *n_out = total;
  return LSLAM_OK;
}

// KEY_DROP & the cube field: a dropped point's key must compare above every real key inside the
// sorted bit range; real cube ids are < ncube <= 2^bits - 1 only if ncube is not a power of two...
// so one more bit is reserved (cube_bits(ncube + 1)).

// wait = false: the caller waits for the stream itself before it reads h_begin / h_end (both feature types behind one wait)
int refresh_segments_device(lslam_fmap *fm, int t) {
  if (fm->seg_dev_current[t] || fm->seg_current[t]) return LSLAM_OK;
  hipStream_t s = fm->stream;
  // [begin | end] in ONE buffer of 2 x seg_pad words (seg_pad = ncube rounded up to 4 words: the fill is one aligned launch
  // instead of an aligned part and a tail per array)
  const size_t pad = seg_pad(fm);
  FM_TRY(fm->seg_begin[t].reserve(2 * pad));
  FM_TRY(hipMemsetAsync(fm->seg_begin[t].p, 0, sizeof(int32_t) * 2 * pad, s));
  if (fm->n[t])
    hipLaunchKernelGGL(fm_segment_kernel, dim3(((int)fm->n[t] + 255) / 256), dim3(256), 0, s, fm->cube[t].p,
                       (int)fm->n[t], fm->seg_begin[t].p, fm->seg_begin[t].p + pad);
  fm->seg_dev_current[t] = true;
  return LSLAM_OK;
}
int refresh_segments(lslam_fmap *fm, int t, bool wait = true) {
  if (fm->seg_current[t]) return LSLAM_OK;
  int rc = refresh_segments_device(fm, t);
  if (rc) return rc;
  hipStream_t s = fm->stream;
  const size_t pad = seg_pad(fm);
  fm->h_begin[t].resize(2 * pad);
  FM_TRY(hipMemcpyAsync(fm->h_begin[t].data(), fm->seg_begin[t].p, sizeof(int32_t) * 2 * pad, hipMemcpyDeviceToHost, s));
  if (wait) FM_TRY(hipStreamSynchronize(s));
  fm->seg_current[t] = true;
  return LSLAM_OK;
}

// bits per voxel axis that hold any map cube's extent: points of a cube lie within cube_size of each other on every axis
// (cube_of_point rounds p / cube_size), two cells of slack for the rounding of the two floors
int map_axis_bits(const lslam_fmap *fm, float leaf) { return bits_for((double)fm->cube_size / (double)leaf + 3.0); }

// rebuild type t from its current points plus n_new transformed points (in_tf / in_cube; default: fm->in_tf / in_cube).
// done != nullptr: everything is enqueued and nothing waited for -- rebuild_finish after the caller's wait.
// old_sorted: the current points are an earlier rebuild's output with the same leaf and active set, i.e. in key order (checked
// on the device; see run_pipeline) -- addFeatureCloud's case.
int rebuild_begin(lslam_fmap *fm, int t, size_t n_new, bool allow_filter, const uint8_t *flags_override, uint32_t *done,
                  const float4 *in_tf = nullptr, const int32_t *in_cube = nullptr, size_t *n_out_sync = nullptr,
                  bool old_sorted = false, hipStream_t on_stream = nullptr, Scratch *scratch = nullptr) {
  hipStream_t s = on_stream ? on_stream : fm->stream;
  Scratch &sc = scratch ? *scratch : fm->sc;
  const size_t n_old = fm->n[t], n_total = n_old + n_new;
  if (done) done[0] = done[1] = done[2] = 0;
  if (n_total == 0) return LSLAM_OK;
  FM_TRY(fm->pts[t].grow(n_total, n_old, s));
  FM_TRY(fm->cube[t].grow(n_total, n_old, s));
  if (n_new) {  // (addFeatureCloud's transform writes the new points where they go: nothing to copy then)
    const float4 *src_p = in_tf ? in_tf : fm->in_tf.p;
    const int32_t *src_c = in_cube ? in_cube : fm->in_cube.p;
    if (src_p != fm->pts[t].p + n_old) FM_TRY(hipMemcpyAsync(fm->pts[t].p + n_old, src_p, n_new * sizeof(float4), hipMemcpyDeviceToDevice, s));
    if (src_c != fm->cube[t].p + n_old) FM_TRY(hipMemcpyAsync(fm->cube[t].p + n_old, src_c, n_new * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  }
  FM_TRY(fm->pts_alt[t].reserve(n_total));
  FM_TRY(fm->cube_alt[t].reserve(n_total));
  KeyParams kp = key_params(fm, fm->leaf[t]);
  size_t n_out = 0;
  int rc = run_pipeline(s, sc, fm->pts[t].p, fm->cube[t].p, n_total, kp, fm->ncube,
                        flags_override ? flags_override : (allow_filter ? fm->active.p : nullptr), fm->pts_alt[t].p,
                        fm->cube_alt[t].p, &n_out, map_axis_bits(fm, fm->leaf[t]), done, old_sorted && n_new ? n_old : 0, true);
  if (n_out_sync) *n_out_sync = n_out;
  return rc;
}
void rebuild_commit(lslam_fmap *fm, int t, size_t n_new, size_t n_out, bool all_dirty) {
  if (fm->n[t] + n_new == 0) return;
  std::swap(fm->pts[t], fm->pts_alt[t]);
  std::swap(fm->cube[t], fm->cube_alt[t]);
  fm->n[t] = n_out;
  fm->seg_current[t] = false;
  fm->seg_dev_current[t] = false;
  if (all_dirty) fm->dirty[t].assign((size_t)fm->ncube, 1);  // shift / load: every cube's cloud may have moved
}
int rebuild(lslam_fmap *fm, int t, size_t n_new, bool allow_filter, const uint8_t *flags_override = nullptr) {
  size_t n_out = 0;
  int rc = rebuild_begin(fm, t, n_new, allow_filter, flags_override, nullptr, nullptr, nullptr, &n_out);
  if (rc) return rc;
  rebuild_commit(fm, t, n_new, n_out, n_new == 0 || flags_override);
  return LSLAM_OK;
}

bool idx_valid(const lslam_fmap *fm, int i, int j, int k) {
  return 0 <= i && i < fm->W && 0 <= j && j < fm->H && 0 <= k && k < fm->D;
}
int to_index(const lslam_fmap *fm, int i, int j, int k) { return i + j * fm->W + k * fm->W * fm->H; }

// FeatureMap.h:307-352
void compute_active_area(lslam_fmap *fm, const float pos[3]) {
  fm->valid.clear();
  const int win = (int)std::ceil(fm->valid_dist / fm->cube_size);
  for (int i = fm->cur[0] - win; i <= fm->cur[0] + win; ++i)
    for (int j = fm->cur[1] - win; j <= fm->cur[1] + win; ++j)
      for (int k = fm->cur[2] - win; k <= fm->cur[2] + win; ++k) {
        if (!idx_valid(fm, i, j, k)) continue;
        const float cx = fm->cube_size * (float)(i - fm->origin[0]);
        const float cy = fm->cube_size * (float)(j - fm->origin[1]);
        const float cz = fm->cube_size * (float)(k - fm->origin[2]);
        bool in_fov = false;
        for (int ii = -1; ii <= 1 && !in_fov; ii += 2)
          for (int jj = -1; jj <= 1 && !in_fov; jj += 2)
            for (int kk = -1; kk <= 1 && !in_fov; kk += 2) {
              const float x = (float)((double)cx + (double)fm->cube_size / 2.0 * ii);
              const float y = (float)((double)cy + (double)fm->cube_size / 2.0 * jj);
              const float z = (float)((double)cz + (double)fm->cube_size / 2.0 * kk);
              const float dx = pos[0] - x, dy = pos[1] - y, dz = pos[2] - z;
              const float sq = dx * dx + dy * dy + dz * dz;
              if (std::sqrt((double)sq) < (double)fm->valid_dist) in_fov = true;
            }
        if (in_fov) fm->valid.push_back(to_index(fm, i, j, k));
      }
}

int upload_active(lslam_fmap *fm) {
  // the active area is the same from sweep to sweep until the sensor crosses into another cube's reach: nothing to upload then
  if (fm->valid_on_device && fm->valid == fm->valid_uploaded) return LSLAM_OK;
  std::vector<uint8_t> h(fm->ncube, 0);
  for (int32_t c : fm->valid) h[c] = 1;
  FM_TRY(hipMemcpyAsync(fm->active.p, h.data(), fm->ncube, hipMemcpyHostToDevice, fm->stream));
  FM_TRY(fm->d_valid.reserve(fm->valid.size() + 1));
  if (!fm->valid.empty())
    FM_TRY(hipMemcpyAsync(fm->d_valid.p, fm->valid.data(), fm->valid.size() * sizeof(int32_t), hipMemcpyHostToDevice, fm->stream));
  FM_TRY(hipStreamSynchronize(fm->stream));  // h is a local
  fm->valid_uploaded = fm->valid;
  fm->valid_on_device = true;
  return LSLAM_OK;
}

// host cloud -> pinned staging -> dst (device), no wait: `pin` lives as long as its owner and is not written again before
// the owner's next wait.  mn / mx (optional): the cloud's bounding box, taken in the same pass.
int pack_input(hipStream_t s, Pin<float4> &pin, Buf<float4> &dst, const void *src, size_t n, size_t stride_bytes,
               float *mn = nullptr, float *mx = nullptr) {
  // {x,y,z} at offset 0; intensity at offset 12 for 16-byte points, 16 for pcl::PointXYZI (32 bytes)
  FM_TRY(pin.reserve(n));
  FM_TRY(dst.reserve(n));
  float4 *h = pin.p;
  const char *p = static_cast<const char *>(src);
  const size_t ioff = stride_bytes == 16 ? 12 : 16;
  if (stride_bytes == 16) {
    std::memcpy(h, p, n * sizeof(float4));
  } else {
    for (size_t i = 0; i < n; ++i) {
      float v[3], w = 0.0f;
      std::memcpy(v, p + i * stride_bytes, 12);
      if (stride_bytes >= ioff + 4) std::memcpy(&w, p + i * stride_bytes + ioff, 4);
      h[i] = make_float4(v[0], v[1], v[2], w);
    }
  }
  if (mn && mx) {
    // four independent accumulators of four lanes each (the fourth lane, the intensity, rides along unused): the same
    // compare-and-keep per component as a scalar loop (a NaN never replaces an extreme), at the vector units' rate
    float lo[4][4], hi[4][4];
    for (int a = 0; a < 4; ++a)
      for (int d = 0; d < 4; ++d) { lo[a][d] = INFINITY; hi[a][d] = -INFINITY; }
    const float *f = reinterpret_cast<const float *>(h);
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
      for (int a = 0; a < 4; ++a)
        for (int d = 0; d < 4; ++d) {
          const float v = f[4 * (i + a) + d];
          lo[a][d] = v < lo[a][d] ? v : lo[a][d];
          hi[a][d] = v > hi[a][d] ? v : hi[a][d];
        }
    for (; i < n; ++i)
      for (int d = 0; d < 4; ++d) {
        const float v = f[4 * i + d];
        lo[0][d] = v < lo[0][d] ? v : lo[0][d];
        hi[0][d] = v > hi[0][d] ? v : hi[0][d];
      }
    for (int d = 0; d < 3; ++d)
      for (int a = 0; a < 4; ++a) {
        mn[d] = lo[a][d] < mn[d] ? lo[a][d] : mn[d];
        mx[d] = hi[a][d] > mx[d] ? hi[a][d] : mx[d];
      }
  }
  FM_TRY(hipMemcpyAsync(dst.p, h, n * sizeof(float4), hipMemcpyHostToDevice, s));
  return LSLAM_OK;
}

int finish_add(lslam_fmap *fm);
int check_fm(lslam_fmap *fm) {
  if (!fm || !lslam::ctx_alive(fm->ctx)) {
    lslam::set_error("null feature map, or its ctx was destroyed");
    return LSLAM_ERR_INVALID;
  }
  FM_TRY(hipSetDevice(lslam::ctx_device(fm->ctx)));
  if (fm->add_pending) return finish_add(fm);  // an addFeatureCloud begun and not yet committed: first of all that
  return LSLAM_OK;
}

// surround arrays of type t into fm->sur[t]; returns count
int gather_surround(lslam_fmap *fm, int t, int index_in_w, size_t *n_out, int min_points = 1,
                    std::vector<int32_t> *roots_lr = nullptr, std::vector<int32_t> *cell_tree = nullptr, bool no_wait = false) {
  int rc = refresh_segments(fm, t);
  if (rc) return rc;
  std::vector<int32_t> src_local, dst_local;
  std::vector<int32_t> &src = no_wait ? fm->h_gsrc[t] : src_local, &dst = no_wait ? fm->h_gdst[t] : dst_local;
  src.clear();
  dst.clear();
  size_t total = 0;
  if (cell_tree) cell_tree->assign((size_t)fm->ncube, -1);
  for (int32_t c : fm->valid) {
    const int32_t b = seg_b(fm, t, c), e = seg_e(fm, t, c);
    if (e - b >= min_points && e > b) {
      if (roots_lr) {
        if (cell_tree) (*cell_tree)[(size_t)c] = (int32_t)(roots_lr->size() / 2);
        roots_lr->push_back((int32_t)total);
        roots_lr->push_back((int32_t)(total + (size_t)(e - b)));
      }
      src.push_back(b);
      dst.push_back((int32_t)total);
      total += (size_t)(e - b);
    }
  }
  *n_out = total;
  if (!total) return LSLAM_OK;
  hipStream_t s = fm->stream;
  Buf<int32_t> &gs = no_wait ? fm->g_src_t[t] : fm->g_src, &gd = no_wait ? fm->g_dst_t[t] : fm->g_dst;
  FM_TRY(gs.reserve(src.size()));
  FM_TRY(gd.reserve(dst.size()));
  FM_TRY(fm->sur[t].reserve(total));
  FM_TRY(hipMemcpyAsync(gs.p, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  FM_TRY(hipMemcpyAsync(gd.p, dst.data(), dst.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(fm_gather_kernel, dim3(((int)total + 255) / 256), dim3(256), 0, s, fm->pts[t].p, gs.p,
                     gd.p, (int)src.size(), (int)total, index_in_w, fm->sur[t].p);
  if (!no_wait) FM_TRY(hipStreamSynchronize(s));  // src/dst are locals
  return LSLAM_OK;
}

}  // namespace

namespace lslam {
// VoxelGrid per segment for other translation units (feature extraction: one segment per scan ring):
// in_seg[i] is the segment of point i (ascending or not), every segment is filtered with `leaf`;
// output ordered by segment, inside a segment in VoxelGrid order.  Scratch is cached per process.
// done != nullptr (pinned, two words): nothing is waited for -- the widest voxel extent is not read back (the key gets every
// bit the 63 allow: while the sort is a merge sort its width costs nothing) and {points out, key-range error} arrive in
// done[0..1] behind everything else on the stream; on an error the caller calls again without `done`.
int voxel_filter_segments(hipStream_t s, const float4 *in_pts, const int32_t *in_seg, size_t n, int nseg, float leaf,
                          float4 *out_pts, int32_t *out_seg, size_t *n_out, bool filter, uint32_t *done) {
  struct Cache {
    Scratch sc;
    Buf<uint8_t> all;
    size_t all_set = 0;  // entries of `all` that hold their 1 already
  };
  // Scratch per STREAM (= per context), not per device: in the no-wait form (`done`) this function returns with its kernels
  // still running, and a caller on another context of the same device -- the registration node beside the mapping node, each
  // on its own thread -- would otherwise be handed the same scratch while they do.  (It was: garbage segment ids, a host
  // segfault in the registration, "output buffer too small" in the map, found by tools/cpp/node_threads.cpp.)  Work on ONE
  // stream is ordered by the stream.  The lock covers the table of caches only; a map's nodes do not move.
  static std::map<hipStream_t, Cache> caches;
  static std::mutex mu;
  Cache *cache_p;
  {
    std::lock_guard<std::mutex> lk(mu);
    cache_p = &caches[s];
  }
  Scratch &sc = cache_p->sc;
  Buf<uint8_t> &all = cache_p->all;
  size_t &all_set = cache_p->all_set;
  {  // "every segment is filtered": ones, written when the array grows -- not once per call
    const uint8_t *before = all.p;
    FM_TRY(all.reserve((size_t)nseg));
    if (all.p != before) all_set = 0;
    if (all_set < (size_t)nseg) {
      FM_TRY(hipMemsetAsync(all.p, 1, all.cap, s));
      FM_TRY(hipStreamSynchronize(s));  // (once per growth; the array is shared by the contexts of a device, whatever their streams)
      all_set = all.cap;
    }
  }
  KeyParams kp{};
  kp.W = nseg; kp.H = 1; kp.D = 1;
  kp.cube_size = 1.0f;
  kp.inv_leaf = 1.0f / leaf;
  kp.axis_bits = 1;
  kp.single = 0;
  const int wide = std::min(20, (63 - cube_bits(nseg + 1)) / 3);
  // filter == false: only the stable grouping by segment (points with segment -1 are dropped)
  const bool no_wait = done && n <= MERGE_LIMIT;
  const int rc = run_pipeline(s, sc, in_pts, in_seg, n, kp, nseg, filter ? all.p : nullptr, out_pts, out_seg, n_out,
                              no_wait ? wide : 0, no_wait ? done : nullptr);
  if (rc == LSLAM_OK && done && !no_wait) {  // waited after all: the result where the caller looks for it
    done[0] = (uint32_t)*n_out;
    done[1] = 0;
  }
  return rc;
}
}  // namespace lslam

extern "C" {

int lslam_fmap_create(lslam_ctx *ctx, int32_t w, int32_t h, int32_t d, lslam_fmap **out) {
  if (!ctx || !out || w <= 0 || h <= 0 || d <= 0 || (long long)w * h * d > (1ll << 24)) {
    lslam::set_error("bad feature-map arguments");
    return LSLAM_ERR_INVALID;
  }
  *out = nullptr;
  FM_TRY(hipSetDevice(lslam::ctx_device(ctx)));
  lslam_fmap *fm = new lslam_fmap();
  fm->ctx = ctx;
  fm->stream = (hipStream_t)lslam_stream(ctx);
  fm->W = w; fm->H = h; fm->D = d;
  fm->ncube = w * h * d;
  // FeatureMap.h:60-62: origin = round(--size / 2.0)
  fm->origin[0] = (int)std::round((w - 1) / 2.0);
  fm->origin[1] = (int)std::round((h - 1) / 2.0);
  fm->origin[2] = (int)std::round((d - 1) / 2.0);
  FM_TRY(fm->active.reserve(fm->ncube));
  FM_TRY(hipMemsetAsync(fm->active.p, 0, fm->ncube, fm->stream));
  FM_TRY(hipStreamSynchronize(fm->stream));
  *out = fm;
  return LSLAM_OK;
}

void lslam_fmap_destroy(lslam_fmap *fm) {
  if (!fm) return;
  if (lslam::ctx_alive(fm->ctx)) {  // a feature map may outlive its ctx; its stream is then gone
    (void)hipSetDevice(lslam::ctx_device(fm->ctx));
    (void)hipStreamSynchronize(fm->stream);
  }
  for (int t = 0; t < 2; ++t) {
    fm->pts[t].release(); fm->pts_alt[t].release(); fm->cube[t].release(); fm->cube_alt[t].release();
    fm->seg_begin[t].release(); fm->sur[t].release();
  }
  for (lslam_fmap::Generation *g : fm->gens)
    if (g) { g->nodes.release(); g->pn.release(); g->pts.release(); delete g; }
  fm->gens.clear();
  for (lslam_fmap::Generation *g : fm->spare) { g->nodes.release(); g->pn.release(); g->pts.release(); delete g; }
  fm->spare.clear();
  fm->d_touched.release();
  if (lslam::ctx_alive(fm->ctx)) lslam::cubemap_drop_views(fm->ctx);  // the context may still point at this map's trees
  fm->active.release(); fm->in_raw.release(); fm->in_tf.release(); fm->in_cube.release();
  fm->d_remap.release(); fm->g_src.release(); fm->g_dst.release();
  for (int t = 0; t < 2; ++t) {
    fm->g_src_t[t].release(); fm->g_dst_t[t].release();
    fm->d_valid.release(); fm->sur_res.release();
    fm->in_pin[t].release(); fm->in_raw_t[t].release();
    fm->d_touched_t[t].release();
  }
  fm->done.release(); fm->h_touched.release();
  fm->sc.release();
  fm->sc2.release();
  if (fm->stream2 && lslam::ctx_alive(fm->ctx)) (void)hipStreamSynchronize(fm->stream2);  // (the context's: not ours to destroy)
  if (fm->ev_fork) (void)hipEventDestroy(fm->ev_fork);
  if (fm->ev_join) (void)hipEventDestroy(fm->ev_join);
  delete fm;
}

int lslam_fmap_setup_filter_size(lslam_fmap *fm, float corner, float surf, float map) {
  if (!fm || !(corner > 0.f) || !(surf > 0.f) || !(map > 0.f)) return LSLAM_ERR_INVALID;
  fm->leaf[0] = corner; fm->leaf[1] = surf; fm->leaf[2] = map;
  return LSLAM_OK;
}
int lslam_fmap_setup_world_cube_size(lslam_fmap *fm, float size) {
  if (!fm || !(size > 0.f)) return LSLAM_ERR_INVALID;
  fm->cube_size = size;
  return LSLAM_OK;
}
int lslam_fmap_setup_lidar_valid_distance(lslam_fmap *fm, float dist) {
  if (!fm) return LSLAM_ERR_INVALID;
  fm->valid_dist = dist;
  return LSLAM_OK;
}
int lslam_fmap_setup_world_origin(lslam_fmap *fm, int32_t ox, int32_t oy, int32_t oz) {
  if (!fm) return LSLAM_ERR_INVALID;
  fm->origin[0] = ox; fm->origin[1] = oy; fm->origin[2] = oz;
  return LSLAM_OK;
}

int lslam_fmap_update(lslam_fmap *fm, const float pos[3]) {
  int rc = check_fm(fm);
  if (rc) return rc;
  if (!pos) return LSLAM_ERR_INVALID;
  hipStream_t s = fm->stream;
  int g[3];
  const int lim[3] = {fm->W, fm->H, fm->D};
  for (int d = 0; d < 3; ++d) g[d] = (int)(std::round(pos[d] / fm->cube_size) + (float)fm->origin[d]);
  const int PAD = 3;  // FeatureMap.h:236
  int ng[3], dl[3];
  for (int d = 0; d < 3; ++d) {
    ng[d] = std::min(std::max(g[d], PAD), lim[d] - PAD - 1);
    dl[d] = ng[d] - g[d];
  }
  if (dl[0] || dl[1] || dl[2]) {
    // FeatureMap.h:353-377 verbatim on cube handles: the loop swaps cube POINTERS while it walks
    // the grid, so what ends up where is whatever this sequential swap chain produces
    std::vector<int32_t> content(fm->ncube);  // content[slot] = original slot whose cloud sits here, -1 = cleared
    for (int c = 0; c < fm->ncube; ++c) content[c] = c;
    for (int i = 0; i < fm->W; ++i)
      for (int j = 0; j < fm->H; ++j)
        for (int k = 0; k < fm->D; ++k) {
          const int oi = i - dl[0], oj = j - dl[1], ok = k - dl[2];
          const int a = to_index(fm, i, j, k);
          if (idx_valid(fm, oi, oj, ok)) std::swap(content[a], content[to_index(fm, oi, oj, ok)]);
          else content[a] = -1;
        }
    std::vector<int32_t> new_of_old(fm->ncube, -1);
    for (int c = 0; c < fm->ncube; ++c)
      if (content[c] >= 0) new_of_old[content[c]] = c;
    FM_TRY(fm->d_remap.reserve(fm->ncube));
    FM_TRY(hipMemcpyAsync(fm->d_remap.p, new_of_old.data(), sizeof(int32_t) * fm->ncube, hipMemcpyHostToDevice, s));
    for (int t = 0; t < 2; ++t) {
      if (fm->n[t])
        hipLaunchKernelGGL(fm_relabel_kernel, dim3(((int)fm->n[t] + 255) / 256), dim3(256), 0, s, fm->cube[t].p,
                           (int)fm->n[t], fm->d_remap.p);
      rc = rebuild(fm, t, 0, false);  // re-sort by the new cube ids, drop cleared cubes, no filtering
      if (rc) return rc;
    }
  }
  for (int d = 0; d < 3; ++d) {
    fm->origin[d] += dl[d];
    fm->cur[d] = ng[d];
  }
  compute_active_area(fm, pos);
  return upload_active(fm);
}

static int add_feature_cloud_impl(lslam_fmap *fm, const void *corner, size_t n_corner, const void *surf, size_t n_surf,
                                  size_t stride_bytes, const float T[16]);
static int add_settle(lslam_fmap *fm, int rc) {
  // a failure half way leaves copies out of the pinned staging in flight: nothing may reuse it before they are done
  if (rc != LSLAM_OK && fm && lslam::ctx_alive(fm->ctx)) {
    fm->add_pending = false;
    if (fm->stream2) (void)hipStreamSynchronize(fm->stream2);
    (void)hipStreamSynchronize(fm->stream);
  }
  return rc;
}
int lslam_fmap_add_feature_cloud_begin(lslam_fmap *fm, const void *corner, size_t n_corner, const void *surf,
                                       size_t n_surf, size_t stride_bytes, const float T[16]) {
  return add_settle(fm, add_feature_cloud_impl(fm, corner, n_corner, surf, n_surf, stride_bytes, T));
}
int lslam_fmap_add_feature_cloud(lslam_fmap *fm, const void *corner, size_t n_corner, const void *surf,
                                 size_t n_surf, size_t stride_bytes, const float T[16]) {
  int rc = add_feature_cloud_impl(fm, corner, n_corner, surf, n_surf, stride_bytes, T);
  if (rc == LSLAM_OK) rc = finish_add(fm);
  return add_settle(fm, rc);
}
int lslam_fmap_wait(lslam_fmap *fm) { return add_settle(fm, check_fm(fm)); }
static int add_feature_cloud_impl(lslam_fmap *fm, const void *corner, size_t n_corner, const void *surf, size_t n_surf,
                                  size_t stride_bytes, const float T[16]) {
  int rc = check_fm(fm);
  if (rc) return rc;
  if (!T || stride_bytes < 12 || (stride_bytes & 3) || (n_corner && !corner) || (n_surf && !surf)) {
    lslam::set_error("bad cloud arguments");
    return LSLAM_ERR_INVALID;
  }
  hipStream_t s = fm->stream;
  // everything for both feature types is enqueued, then ONE wait: uploads come from pinned staging, the voxel extent of a map
  // cube is bounded (map_axis_bits), the rebuilds' results land in pinned slots
  FM_TRY(fm->done.reserve(8 + 16));
  Rigid12 Tm;
  std::memcpy(Tm.m, T, 12 * sizeof(float));
  // which cubes received points matters to the per-cube kd-trees only (lslam_fmap_to_cubemap builds a cube's tree again when
  // its cloud changed): until the first such call every cube counts as changed anyway, and the marks -- a fill, a copy back
  // and a pass over the cubes per feature type -- are not kept
  const bool track = fm->cube_trees_used;
  const void *src[2] = {corner, surf};
  const size_t cnt[2] = {n_corner, n_surf};
  FM_TRY(fm->h_touched.reserve(2 * (size_t)fm->ncube));
  // two chains, two streams -- when both types have something to rebuild
  const bool two = !lslam::env_once().fmap_one_stream && (fm->n[0] + cnt[0]) && (fm->n[1] + cnt[1]);
  if (two) {
    if (!fm->stream2) {  // the context's second stream (the cell grids fork onto it too, never at the same time: one call per context)
      fm->stream2 = lslam::ctx_stream2(fm->ctx);
      if (!fm->stream2) return LSLAM_ERR_HIP;
      FM_TRY(hipEventCreateWithFlags(&fm->ev_fork, hipEventDisableTiming));
      FM_TRY(hipEventCreateWithFlags(&fm->ev_join, hipEventDisableTiming));
    }
    FM_TRY(hipEventRecord(fm->ev_fork, s));  // behind whatever the context's stream still does with the surf arrays
    FM_TRY(hipStreamWaitEvent(fm->stream2, fm->ev_fork, 0));
  }
  for (int t = 0; t < 2; ++t) {
    const size_t n = cnt[t];
    hipStream_t st = (two && t == 1) ? fm->stream2 : s;
    if (n) {
      rc = pack_input(st, fm->in_pin[t], fm->in_raw_t[t], src[t], n, stride_bytes);
      if (rc) return rc;
      // the transformed points go straight behind the type's current points, where the rebuild wants them (was: a staging
      // array of their own and two device-to-device copies per type)
      FM_TRY(fm->pts[t].grow(fm->n[t] + n, fm->n[t], st));
      FM_TRY(fm->cube[t].grow(fm->n[t] + n, fm->n[t], st));
      KeyParams kp = key_params(fm, fm->leaf[t]);
      if (track) {
        FM_TRY(fm->d_touched_t[t].reserve(((size_t)fm->ncube + 15) & ~(size_t)15));
        FM_TRY(hipMemsetAsync(fm->d_touched_t[t].p, 0, ((size_t)fm->ncube + 15) & ~(size_t)15, st));  // (a whole number of 16-byte words: one fill launch, no tail)
      }
      hipLaunchKernelGGL(fm_transform_kernel, dim3(((int)n + 255) / 256), dim3(256), 0, st, fm->in_raw_t[t].p, (int)n,
                         Tm, kp, fm->pts[t].p + fm->n[t], fm->cube[t].p + fm->n[t], track ? fm->d_touched_t[t].p : (uint8_t *)nullptr);
      if (track) FM_TRY(hipMemcpyAsync(fm->h_touched.p + (size_t)t * fm->ncube, fm->d_touched_t[t].p, fm->ncube, hipMemcpyDeviceToHost, st));
    }
    rc = rebuild_begin(fm, t, n, true, nullptr, fm->done.p + 4 * t, fm->pts[t].p + fm->n[t], fm->cube[t].p + fm->n[t], nullptr, true, st,
                       (two && t == 1) ? &fm->sc2 : &fm->sc);
    if (rc) return rc;
  }
  if (two) {
    FM_TRY(hipEventRecord(fm->ev_join, fm->stream2));
    FM_TRY(hipStreamWaitEvent(s, fm->ev_join, 0));
  }
  fm->add_cnt[0] = cnt[0];
  fm->add_cnt[1] = cnt[1];
  fm->add_track = track;
  fm->add_pending = true;
  return LSLAM_OK;
}

namespace {
// the wait and the commit of an addFeatureCloud begun by add_feature_cloud_impl
int finish_add(lslam_fmap *fm) {
  fm->add_pending = false;
  hipStream_t s = fm->stream;
  const size_t *cnt = fm->add_cnt;
  const bool track = fm->add_track;
  int rc = LSLAM_OK;
  FM_TRY(hipStreamSynchronize(s));
  for (int t = 0; t < 2; ++t) {
    const size_t n = cnt[t];
    if (fm->done.p[4 * t + 1] || fm->done.p[4 * t + 2]) {
      // a voxel index outside the key range the cube size promises (or a non-finite point), or current points that were not
      // in key order after all (a cube that has just become active): once more, sorting everything and waiting for the
      // measured extents -- the inputs are untouched, the appended points are written again where they are
      size_t n_out = 0;
      rc = rebuild_begin(fm, t, n, true, nullptr, nullptr, fm->pts[t].p + fm->n[t], fm->cube[t].p + fm->n[t], &n_out);
      if (rc) return rc;
      rebuild_commit(fm, t, n, n_out, n == 0);
      fm->resorted_rebuilds++;
    } else {
      rebuild_commit(fm, t, n, fm->done.p[4 * t], n == 0);
      if (n && fm->n[t]) fm->merged_rebuilds++;
    }
    if (n && track) {
      fm->dirty[t].resize((size_t)fm->ncube, 1);
      const uint8_t *ht = fm->h_touched.p + (size_t)t * fm->ncube;
      for (int c = 0; c < fm->ncube; ++c) fm->dirty[t][(size_t)c] |= ht[(size_t)c];
    }
  }
  return LSLAM_OK;
}
}  // namespace

int lslam_fmap_surround_counts(lslam_fmap *fm, size_t *n_corner, size_t *n_surf) {
  int rc = check_fm(fm);
  if (rc) return rc;
  size_t *out[2] = {n_corner, n_surf};
  for (int t = 0; t < 2; ++t) {
    rc = refresh_segments(fm, t);
    if (rc) return rc;
    size_t total = 0;
    for (int32_t c : fm->valid) total += (size_t)(seg_e(fm, t, c) - seg_b(fm, t, c));
    if (out[t]) *out[t] = total;
  }
  return LSLAM_OK;
}

int lslam_fmap_get_surround(lslam_fmap *fm, float *corner_xyzi, size_t cap_corner, float *surf_xyzi,
                            size_t cap_surf) {
  int rc = check_fm(fm);
  if (rc) return rc;
  float *out[2] = {corner_xyzi, surf_xyzi};
  const size_t cap[2] = {cap_corner, cap_surf};
  for (int t = 0; t < 2; ++t) {
    if (!out[t]) continue;
    size_t n = 0;
    rc = gather_surround(fm, t, 0, &n);
    if (rc) return rc;
    if (n > cap[t]) {
      lslam::set_error("surround buffer too small");
      return LSLAM_ERR_INVALID;
    }
    if (n) FM_TRY(hipMemcpyAsync(out[t], fm->sur[t].p, n * sizeof(float4), hipMemcpyDeviceToHost, fm->stream));
  }
  FM_TRY(hipStreamSynchronize(fm->stream));
  return LSLAM_OK;
}

int lslam_fmap_surround_to_map(lslam_fmap *fm) { return lslam_fmap_surround_to_map_counts(fm, nullptr, nullptr); }

int lslam_fmap_surround_to_map_counts(lslam_fmap *fm, size_t *n_corner, size_t *n_surf) {
  if (n_corner) *n_corner = 0;
  if (n_surf) *n_surf = 0;
  int rc = check_fm(fm);
  if (rc) return rc;
  hipStream_t s = fm->stream;
  const int n_valid = (int)fm->valid.size();
  size_t n[2] = {0, 0};
  float lo[2][3], hi[2][3];
  FM_TRY(fm->sur_res.reserve(16));
  FM_TRY(fm->done.reserve(8 + 16));
  uint32_t *h_res = fm->done.p + 8;  // pinned: [2 types][8]
  for (int k = 0; k < 16; ++k) h_res[k] = 0u;
  bool any = false;
  for (int t = 0; t < 2; ++t) {
    if (!fm->n[t] || !n_valid) {  // nothing of this type: its words of the result read "no points"
      FM_TRY(hipMemsetAsync(fm->sur_res.p + 8 * t, 0, 8 * sizeof(uint32_t), s));
      continue;
    }
    rc = refresh_segments_device(fm, t);
    if (rc) return rc;
    FM_TRY(fm->g_src_t[t].reserve((size_t)n_valid));
    FM_TRY(fm->g_dst_t[t].reserve((size_t)n_valid));
    FM_TRY(fm->sur[t].reserve(fm->n[t]));  // (the surround is at most the map)
    hipLaunchKernelGGL(fm_surround_plan_kernel, dim3(1), dim3(256), 0, s, (const int32_t *)fm->d_valid.p, n_valid,
                       (const int32_t *)fm->seg_begin[t].p, (int)seg_pad(fm), fm->g_src_t[t].p, fm->g_dst_t[t].p, fm->sur_res.p + 8 * t);
    const unsigned blocks = (unsigned)std::min<size_t>(512, (fm->n[t] + 255) / 256);
    hipLaunchKernelGGL(fm_gather_box_kernel, dim3(blocks), dim3(256), 0, s, (const float4 *)fm->pts[t].p, (const int32_t *)fm->g_src_t[t].p,
                       (const int32_t *)fm->g_dst_t[t].p, n_valid, fm->sur_res.p + 8 * t, 1, fm->sur[t].p);
    FM_TRY(hipGetLastError());
    any = true;
  }
  if (any) {  // one copy and the one wait: both types' totals and boxes (a type that was skipped left its words untouched: zeroed below)
    FM_TRY(hipMemcpyAsync(h_res, fm->sur_res.p, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    FM_TRY(hipStreamSynchronize(s));
  }
  for (int t = 0; t < 2; ++t) {
    n[t] = h_res[8 * t];
    for (int a = 0; a < 3; ++a) {
      auto back = [](uint32_t o) {  // fm_ordered_u32's inverse
        const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
      };
      lo[t][a] = back(~h_res[8 * t + 1 + a]);
      hi[t][a] = back(h_res[8 * t + 4 + a]);
    }
  }
  if (n_corner) *n_corner = n[0];
  if (n_surf) *n_surf = n[1];
  if (n[0] == 0 && n[1] == 0) return lslam_map_set(fm->ctx, nullptr, 0, nullptr, 0, 16);
  FM_TRY(fm->sur[0].reserve(1));
  FM_TRY(fm->sur[1].reserve(1));
  return lslam::map_set_device(fm->ctx, fm->sur[0].p, n[0], fm->sur[1].p, n[1], lo, hi);
}

static void drop_cube_trees(lslam_fmap *fm, int t) {
  for (lslam_fmap::CubeTree &ct : fm->cube_tree[t]) {
    if (ct.gen >= 0) fm->gens[(size_t)ct.gen]->live--;
    ct.gen = -1;
  }
}

// Forget the cached per-cube trees: the next lslam_fmap_to_cubemap builds every tree of the active area anew.
int lslam_fmap_cubemap_invalidate(lslam_fmap *fm) {
  if (!fm) return LSLAM_ERR_INVALID;
  for (int t = 0; t < 2; ++t) drop_cube_trees(fm, t);
  return LSLAM_OK;
}

// The active area as a variant-C map (FeatureMap::scanMatchScan, FeatureMap.h:490-691: one kd-tree per cube,
// cubes with fewer than 5 points skipped :524,546).  The reference builds a cube's tree when its cloud is loaded
// and keeps it (:438,453); here a cube's tree is kept as long as its cloud is unchanged: only the cubes that
// received points since their tree was built (or that never had one) are gathered and built -- all of them in one
// go by the device forest builder -- the others keep theirs.
int lslam_fmap_to_cubemap(lslam_fmap *fm) {
  int rc = check_fm(fm);
  if (rc) return rc;
  hipStream_t s = fm->stream;
  std::vector<lslam::TreeView> views[2];
  std::vector<int32_t> cells[2];
  size_t n_pts[2] = {0, 0};
  int depth[2] = {0, 0};
  fm->trees_built = fm->trees_reused = 0;
  fm->cube_trees_used = true;  // (from now on addFeatureCloud keeps the cubes' marks; until now every cube counts as changed)
  const bool timing = lslam::env_once().fmap_timing;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t0 = now();
  // Corner and surf cubes that need a tree are built TOGETHER: one generation, one forest build (a forest build costs
  // ~0.9 ms of dependent launches whatever its size, so two of them per frame were most of this call).  Generations
  // live in one list; CubeTree::gen indexes it for both feature types.
  std::vector<lslam_fmap::Generation *> &gens = fm->gens;
  for (int t = 0; t < 2; ++t) {
    rc = refresh_segments(fm, t, false);
    if (rc) return rc;
    fm->cube_tree[t].resize((size_t)fm->ncube);
    fm->dirty[t].resize((size_t)fm->ncube, 1);
  }
  FM_TRY(hipStreamSynchronize(s));  // both types' segment tables are on the host
  // compaction: too many generations alive means many half-empty arrays -- drop every tree, the active area is
  // rebuilt into one generation below
  if (gens.size() > 12) {
    drop_cube_trees(fm, 0);
    drop_cube_trees(fm, 1);
  }
  std::vector<int32_t> build[2], src[2], dst[2], roots_lr;
  size_t total_t[2] = {0, 0};
  for (int t = 0; t < 2; ++t) {
    const size_t base = t == 0 ? 0 : total_t[0];
    for (int32_t c : fm->valid) {
      const int32_t b = seg_b(fm, t, c), e = seg_e(fm, t, c);
      lslam_fmap::CubeTree &ct = fm->cube_tree[t][(size_t)c];
      const bool wants_tree = e - b >= 5;
      if (!wants_tree || fm->dirty[t][(size_t)c]) {
        if (ct.gen >= 0) {  // its tree is stale
          gens[(size_t)ct.gen]->live--;
          ct.gen = -1;
        }
      }
      if (wants_tree && ct.gen < 0) {
        build[t].push_back(c);
        src[t].push_back(b);
        dst[t].push_back((int32_t)total_t[t]);  // relative to this type's part of the generation
        roots_lr.push_back((int32_t)(base + total_t[t]));
        roots_lr.push_back((int32_t)(base + total_t[t] + (size_t)(e - b)));
        total_t[t] += (size_t)(e - b);
      }
      fm->dirty[t][(size_t)c] = 0;
    }
  }
  const size_t total = total_t[0] + total_t[1];
  const int T = (int)(build[0].size() + build[1].size());
  const auto t1 = now();
  if (T > 0) {
    lslam_fmap::Generation *g = nullptr;
    {  // a spare generation (the one with the largest point array: reserve() only ever grows it) or a new one
      size_t best = 0;
      for (size_t k = 0; k < fm->spare.size(); ++k)
        if (fm->spare[k]->pts.cap >= fm->spare[best]->pts.cap) best = k;
      if (!fm->spare.empty()) {
        g = fm->spare[best];
        fm->spare.erase(fm->spare.begin() + (long)best);
      } else {
        g = new lslam_fmap::Generation();
      }
    }
    int gi = -1;
    for (size_t k = 0; k < gens.size(); ++k)
      if (!gens[k]) { gi = (int)k; break; }
    if (gi < 0) { gi = (int)gens.size(); gens.push_back(nullptr); }
    gens[(size_t)gi] = g;
    FM_TRY(g->pts.reserve(total + 16));
    const size_t n_seg = src[0].size() + src[1].size();
    FM_TRY(fm->g_src.reserve(n_seg));
    FM_TRY(fm->g_dst.reserve(n_seg));
    auto gather = [&]() -> hipError_t {
      size_t seg0 = 0, base = 0;
      for (int t = 0; t < 2; ++t) {
        if (!src[t].empty()) {
          hipLaunchKernelGGL(fm_gather_kernel, dim3(((int)total_t[t] + 255) / 256), dim3(256), 0, s, fm->pts[t].p,
                             fm->g_src.p + seg0, fm->g_dst.p + seg0, (int)src[t].size(), (int)total_t[t], 2, g->pts.p + base);
        }
        seg0 += src[t].size();
        base += total_t[t];
      }
      return hipGetLastError();
    };
    {
      size_t seg0 = 0;
      for (int t = 0; t < 2; ++t) {
        if (src[t].empty()) continue;
        FM_TRY(hipMemcpyAsync(fm->g_src.p + seg0, src[t].data(), src[t].size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        FM_TRY(hipMemcpyAsync(fm->g_dst.p + seg0, dst[t].data(), dst[t].size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        seg0 += src[t].size();
      }
    }
    FM_TRY(gather());  // (src / dst: pageable sources are consumed when hipMemcpyAsync returns, no wait needed)
    const auto t2 = now();
    std::vector<lslam::TreeView> built((size_t)T);
    int fallback = 0, max_depth = 0;
    size_t n_leaves = 0;
    // node slots: the guess that worked last time first (a guess that is too small costs a whole failed build and a second
    // gather: with 2/3 slot per point every frame of the bench map built its forest twice)
    const int first_attempt = fm->forest_attempt0;
    for (int attempt = first_attempt; attempt < 3; ++attempt) {
      const size_t mult[3] = {2, 8, 24};
      const size_t cap = ((mult[attempt] * total / 3 + 64 + 8 * (size_t)T) + 7) & ~(size_t)7;
      FM_TRY(g->nodes.reserve(cap));
      if (attempt > first_attempt) FM_TRY(gather());  // the failed attempt permuted the points: gather them again
      FM_TRY(lslam::build_kdforest_device(g->pts.p, (int32_t)total, roots_lr.data(), T, g->nodes.p, nullptr, (int32_t)cap, s,
                                          built.data(), &max_depth, &n_leaves, &fallback));
      if (fallback != 1) {
        fm->forest_attempt0 = attempt;
        break;
      }
    }
    if (fallback) {
      lslam::set_error("device cube-tree build hit a structure limit");
      return fallback == 1 || fallback == 2 ? LSLAM_ERR_TREE_BUILD : LSLAM_ERR_TREE_DEPTH;
    }
    if (timing)
      std::fprintf(stderr, "[to_cubemap] segments %.3f ms, gather of %zu + %zu points in %zu + %zu cubes %.3f ms, forest %.3f ms\n",
                   ms(t0, t1), total_t[0], total_t[1], build[0].size(), build[1].size(), ms(t1, t2), ms(t2, now()));
    g->depth = max_depth;
    g->live = T;
    int k = 0;
    for (int t = 0; t < 2; ++t)
      for (int32_t c : build[t]) {
        lslam_fmap::CubeTree &ct = fm->cube_tree[t][(size_t)c];
        ct.gen = gi;
        ct.view = built[(size_t)k++];
      }
    fm->trees_built += T;
  }
  // generations none of whose trees is current any more: kept as spares with their allocations
  for (size_t k = 0; k < gens.size(); ++k) {
    lslam_fmap::Generation *g = gens[k];
    if (g && g->live <= 0) {
      g->live = g->depth = 0;
      if (fm->spare.size() < 8) {
        fm->spare.push_back(g);
      } else {
        g->nodes.release(); g->pn.release(); g->pts.release();
        delete g;
      }
      gens[k] = nullptr;
    }
  }
  while (!gens.empty() && !gens.back()) gens.pop_back();
  // the tables the sweep reads: cell -> tree view, active cubes only
  for (int t = 0; t < 2; ++t) {
    cells[t].assign((size_t)fm->ncube, -1);
    for (int32_t c : fm->valid) {
      const lslam_fmap::CubeTree &ct = fm->cube_tree[t][(size_t)c];
      if (ct.gen < 0) continue;
      cells[t][(size_t)c] = (int32_t)views[t].size();
      views[t].push_back(ct.view);
      n_pts[t] += (size_t)ct.view.n_pts;
      depth[t] = std::max(depth[t], gens[(size_t)ct.gen]->depth);
    }
    fm->trees_reused += (int64_t)views[t].size();
  }
  fm->trees_reused -= fm->trees_built;
  const int32_t dims[3] = {fm->W, fm->H, fm->D};
  return lslam::cubemap_set_views(fm->ctx, views[0], cells[0], n_pts[0], depth[0], views[1], cells[1], n_pts[1], depth[1],
                                  fm->cube_size, fm->origin, dims);
}

// statistics of the last lslam_fmap_to_cubemap: trees built in that call / kept from earlier calls
int lslam_fmap_rebuild_stats(lslam_fmap *fm, int64_t *merged, int64_t *resorted) {
  if (!fm) return LSLAM_ERR_INVALID;
  if (merged) *merged = fm->merged_rebuilds;
  if (resorted) *resorted = fm->resorted_rebuilds;
  return LSLAM_OK;
}

int lslam_fmap_cubemap_stats(lslam_fmap *fm, int64_t *trees_built, int64_t *trees_reused) {
  if (!fm) return LSLAM_ERR_INVALID;
  if (trees_built) *trees_built = fm->trees_built;
  if (trees_reused) *trees_reused = fm->trees_reused;
  return LSLAM_OK;
}

int lslam_fmap_info(lslam_fmap *fm, int32_t origin[3], int32_t *n_valid, int32_t *valid_out, size_t cap,
                    size_t *n_corner_total, size_t *n_surf_total) {
  if (!fm) return LSLAM_ERR_INVALID;
  if (origin) for (int d = 0; d < 3; ++d) origin[d] = fm->origin[d];
  if (n_valid) *n_valid = (int32_t)fm->valid.size();
  if (valid_out) for (size_t i = 0; i < fm->valid.size() && i < cap; ++i) valid_out[i] = fm->valid[i];
  if (n_corner_total) *n_corner_total = fm->n[0];
  if (n_surf_total) *n_surf_total = fm->n[1];
  return LSLAM_OK;
}

// FeatureMap.h:267-286: per cube, VoxelGrid(map leaf) of the corner cloud then of the surf cloud
int lslam_fmap_get_full_map(lslam_fmap *fm, float *out_xyzi, size_t cap, size_t *n_out) {
  int rc = check_fm(fm);
  if (rc) return rc;
  hipStream_t s = fm->stream;
  std::vector<float4> h[2];
  std::vector<int32_t> hc[2];
  Buf<uint8_t> force;
  FM_TRY(force.reserve(fm->ncube));
  FM_TRY(hipMemsetAsync(force.p, 1, fm->ncube, s));  // filter every cube
  for (int t = 0; t < 2; ++t) {
    const size_t n = fm->n[t];
    if (!n) continue;
    FM_TRY(fm->pts_alt[t].reserve(n));
    FM_TRY(fm->cube_alt[t].reserve(n));
    KeyParams kp = key_params(fm, fm->leaf[2]);
    size_t m = 0;
    rc = run_pipeline(s, fm->sc, fm->pts[t].p, fm->cube[t].p, n, kp, fm->ncube, force.p, fm->pts_alt[t].p,
                      fm->cube_alt[t].p, &m);
    if (rc) { force.release(); return rc; }
    h[t].resize(m);
    hc[t].resize(m);
    if (m) {
      FM_TRY(hipMemcpyAsync(h[t].data(), fm->pts_alt[t].p, m * sizeof(float4), hipMemcpyDeviceToHost, s));
      FM_TRY(hipMemcpyAsync(hc[t].data(), fm->cube_alt[t].p, m * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    FM_TRY(hipStreamSynchronize(s));
  }
  force.release();
  const size_t total = h[0].size() + h[1].size();
  if (n_out) *n_out = total;
  if (!out_xyzi) return LSLAM_OK;
  if (total > cap) {
    lslam::set_error("full-map buffer too small");
    return LSLAM_ERR_INVALID;
  }
  size_t a = 0, b = 0, o = 0;  // merge by cube: corner of cube c, then surf of cube c
  while (a < h[0].size() || b < h[1].size()) {
    const int32_t ca = a < h[0].size() ? hc[0][a] : INT32_MAX, cb = b < h[1].size() ? hc[1][b] : INT32_MAX;
    const int32_t c = ca < cb ? ca : cb;
    while (a < h[0].size() && hc[0][a] == c) std::memcpy(out_xyzi + 4 * o++, &h[0][a++], 16);
    while (b < h[1].size() && hc[1][b] == c) std::memcpy(out_xyzi + 4 * o++, &h[1][b++], 16);
  }
  return LSLAM_OK;
}

// ---- on-disk format: FeatureMap::saveCloudToFiles / loadCloudFromFiles,
// util/FeatureMap.h:378-462 -- one binary PCD per non-empty (cube, type) named <count>.pcd plus
// index.txt with lines "count type i j k size".
namespace {
bool write_pcd_binary(const std::string &path, const float4 *p, size_t n) {
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  // pcl::io::savePCDFileBinary of a PointXYZI cloud: the four named fields, 16 bytes per point
  std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\n"
                  "TYPE F F F F\nCOUNT 1 1 1 1\nWIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n", n, n);
  const bool ok = std::fwrite(p, sizeof(float4), n, f) == n;
  return std::fclose(f) == 0 && ok;
}

// reads x y z intensity from an ascii or binary PCD (any field order / extra fields of 4-byte types)
bool read_pcd(const std::string &path, std::vector<float4> &out, std::string &err) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) { err = path + " not exist!"; return false; }
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  size_t points = 0;
  std::string data;
  char line[1024];
  while (std::fgets(line, sizeof(line), f)) {
    std::istringstream ls(line);
    std::string tag;
    ls >> tag;
    if (tag == "FIELDS") { std::string v; while (ls >> v) fields.push_back(v); }
    else if (tag == "SIZE") { int v; while (ls >> v) sizes.push_back(v); }
    else if (tag == "COUNT") { int v; while (ls >> v) counts.push_back(v); }
    else if (tag == "POINTS") ls >> points;
    else if (tag == "DATA") { ls >> data; break; }
  }
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (fields.empty() || sizes.size() != fields.size() || counts.size() != fields.size()) {
    std::fclose(f); err = "malformed PCD header in " + path; return false;
  }
  int off[4] = {-1, -1, -1, -1}, step = 0, col[4] = {-1, -1, -1, -1}, ncol = 0;
  const char *want[4] = {"x", "y", "z", "intensity"};
  for (size_t k = 0; k < fields.size(); ++k) {
    for (int w = 0; w < 4; ++w)
      if (fields[k] == want[w] && sizes[k] == 4) { off[w] = step; col[w] = ncol; }
    step += sizes[k] * counts[k];
    ncol += counts[k];
  }
  if (off[0] < 0 || off[1] < 0 || off[2] < 0) { std::fclose(f); err = "PCD without float x y z: " + path; return false; }
  out.assign(points, make_float4(0.f, 0.f, 0.f, 0.f));
  bool ok = true;
  if (data == "binary") {
    std::vector<char> rec((size_t)step);
    for (size_t i = 0; i < points && ok; ++i) {
      ok = std::fread(rec.data(), 1, (size_t)step, f) == (size_t)step;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < 4; ++w) if (off[w] >= 0) std::memcpy(&v[w], rec.data() + off[w], 4);
      out[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else if (data == "ascii") {
    std::vector<double> row((size_t)ncol);
    for (size_t i = 0; i < points && ok; ++i) {
      for (int c = 0; c < ncol && ok; ++c) ok = std::fscanf(f, "%lf", &row[(size_t)c]) == 1;
      out[i] = make_float4((float)row[(size_t)col[0]], (float)row[(size_t)col[1]], (float)row[(size_t)col[2]],
                           col[3] >= 0 ? (float)row[(size_t)col[3]] : 0.f);
    }
  } else {
    ok = false;
    err = "unsupported PCD DATA '" + data + "' in " + path;
  }
  std::fclose(f);
  if (!ok && err.empty()) err = "truncated PCD " + path;
  return ok;
}
}  // namespace

int lslam_fmap_save(lslam_fmap *fm, const char *directory) {
  int rc = check_fm(fm);
  if (rc) return rc;
  if (!directory) return LSLAM_ERR_INVALID;
  hipStream_t s = fm->stream;
  std::vector<float4> h[2];
  for (int t = 0; t < 2; ++t) {
    rc = refresh_segments(fm, t);
    if (rc) return rc;
    h[t].resize(fm->n[t]);
    if (fm->n[t]) FM_TRY(hipMemcpyAsync(h[t].data(), fm->pts[t].p, fm->n[t] * sizeof(float4), hipMemcpyDeviceToHost, s));
  }
  FM_TRY(hipStreamSynchronize(s));
  const std::string dir(directory);
  std::ofstream fout(dir + "/index.txt");
  if (!fout) {
    lslam::set_error("save files error!");
    return LSLAM_ERR_INVALID;
  }
  int count = 0;
  for (int i = 0; i < fm->W; ++i)
    for (int j = 0; j < fm->H; ++j)
      for (int k = 0; k < fm->D; ++k) {
        const int c = to_index(fm, i, j, k);
        for (int t = 0; t < 2; ++t) {
          const int32_t b = seg_b(fm, t, c), e = seg_e(fm, t, c);
          if (e <= b) continue;
          if (!write_pcd_binary(dir + "/" + std::to_string(count) + ".pcd", h[t].data() + b, (size_t)(e - b))) {
            lslam::set_error("cannot write a cube file");
            return LSLAM_ERR_INVALID;
          }
          fout << count << " " << t << " " << i << " " << j << " " << k << " " << (e - b) << std::endl;
          ++count;
        }
      }
  return LSLAM_OK;
}

int lslam_fmap_load(lslam_fmap *fm, const char *directory) {
  int rc = check_fm(fm);
  if (rc) return rc;
  if (!directory) return LSLAM_ERR_INVALID;
  const std::string dir(directory);
  std::ifstream fin(dir + "/index.txt");
  if (!fin) {
    lslam::set_error("no index.txt in the directory");
    return LSLAM_ERR_INVALID;  // the reference returns false and starts with an empty map
  }
  std::vector<float4> pts[2];
  std::vector<int32_t> cube[2];
  std::vector<uint8_t> loaded[2];
  loaded[0].assign(fm->ncube, 0);
  loaded[1].assign(fm->ncube, 0);
  int count, type, i, j, k, size;
  std::string err;
  while (fin >> count >> type >> i >> j >> k >> size) {  // (the reference's eof() loop re-reads the last
    if ((type != 0 && type != 1) || !idx_valid(fm, i, j, k)) continue;  // line: same result)
    std::vector<float4> cl;
    if (!read_pcd(dir + "/" + std::to_string(count) + ".pcd", cl, err)) continue;  // reference: prints, goes on
    const int c = to_index(fm, i, j, k);
    if (loaded[type][c]) {  // a later entry for the same cube replaces the earlier one
      size_t w = 0;
      for (size_t q = 0; q < pts[type].size(); ++q)
        if (cube[type][q] != c) { pts[type][w] = pts[type][q]; cube[type][w] = cube[type][q]; ++w; }
      pts[type].resize(w);
      cube[type].resize(w);
    }
    loaded[type][c] = 1;
    pts[type].insert(pts[type].end(), cl.begin(), cl.end());
    cube[type].insert(cube[type].end(), cl.size(), c);
  }
  hipStream_t s = fm->stream;
  Buf<uint8_t> d_flags;
  FM_TRY(d_flags.reserve(fm->ncube));
  for (int t = 0; t < 2 && rc == LSLAM_OK; ++t) {
    const size_t n = pts[t].size();
    if (hipMemcpyAsync(d_flags.p, loaded[t].data(), fm->ncube, hipMemcpyHostToDevice, s) != hipSuccess) { rc = LSLAM_ERR_HIP; break; }
    if (fm->n[t])
      hipLaunchKernelGGL(fm_dropflag_kernel, dim3(((int)fm->n[t] + 255) / 256), dim3(256), 0, s, fm->cube[t].p, (int)fm->n[t],
                         d_flags.p);
    if (n) {
      if (fm->in_tf.reserve(n) != hipSuccess || fm->in_cube.reserve(n) != hipSuccess ||
          hipMemcpyAsync(fm->in_tf.p, pts[t].data(), n * sizeof(float4), hipMemcpyHostToDevice, s) != hipSuccess ||
          hipMemcpyAsync(fm->in_cube.p, cube[t].data(), n * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess) {
        rc = LSLAM_ERR_HIP;
        break;
      }
    }
    // each loaded cube goes through its type's VoxelGrid (FeatureMap.h:432-436,448-452)
    rc = rebuild(fm, t, n, true, d_flags.p);
    if (rc == LSLAM_OK && hipStreamSynchronize(s) != hipSuccess) rc = LSLAM_ERR_HIP;
  }
  d_flags.release();
  if (rc == LSLAM_ERR_HIP) lslam::set_error("HIP error while loading cube files");
  return rc;
}

// pcl::VoxelGrid<PointXYZI> with a cubic leaf on one host cloud (LaserMatcher.cpp:289-301)
static int voxel_grid_impl(lslam_ctx *ctx, const void *cloud, size_t n, size_t stride_bytes, float leaf, float *out_xyzi, size_t cap,
                           size_t *n_out);
int lslam_voxel_grid(lslam_ctx *ctx, const void *cloud, size_t n, size_t stride_bytes, float leaf,
                     float *out_xyzi, size_t cap, size_t *n_out) {
  const int rc = voxel_grid_impl(ctx, cloud, n, stride_bytes, leaf, out_xyzi, cap, n_out);
  if (rc != LSLAM_OK && ctx && lslam::ctx_alive(ctx)) (void)hipStreamSynchronize((hipStream_t)lslam_stream(ctx));  // (the staging is shared by the next call)
  return rc;
}
static int voxel_grid_impl(lslam_ctx *ctx, const void *cloud, size_t n, size_t stride_bytes, float leaf, float *out_xyzi, size_t cap,
                           size_t *n_out) {
  if (!ctx || !n_out || !(leaf > 0.f) || stride_bytes < 12 || (stride_bytes & 3) || (n && !cloud)) {
    lslam::set_error("bad voxel-grid arguments");
    return LSLAM_ERR_INVALID;
  }
  *n_out = 0;
  if (n == 0) return LSLAM_OK;
  FM_TRY(hipSetDevice(lslam::ctx_device(ctx)));
  hipStream_t s = (hipStream_t)lslam_stream(ctx);
  // staging and scratch are kept between calls (allocation costs more than the filter itself)
  struct Cache {
    Pin<float4> in_pin, out_pin;
    Pin<uint32_t> done;
    Buf<float4> in_raw, out;
    Buf<int32_t> oc;
    Scratch sc;
  };
  static std::map<hipStream_t, Cache> caches;  // per stream = per context (see voxel_filter_segments)
  static std::mutex mu;
  Cache *cache_p;
  {
    std::lock_guard<std::mutex> lk(mu);
    cache_p = &caches[s];
  }
  Cache &cache = *cache_p;
  Buf<float4> &out = cache.out;
  Buf<int32_t> &oc = cache.oc;
  Scratch &sc = cache.sc;
  // upload from pinned staging (no wait), min/max on the host in the same pass (VoxelGrid::applyFilter: getMinMax3D)
  const float inv = 1.0f / leaf;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int rc = pack_input(s, cache.in_pin, cache.in_raw, cloud, n, stride_bytes, mn, mx);
  if (rc) return rc;
  KeyParams kp{};
  kp.W = kp.H = kp.D = 1;
  kp.cube_size = 1.0f;
  kp.inv_leaf = inv;
  kp.single = 1;
  double cells = 1.0;
  long long vol = 1;
  for (int d = 0; d < 3; ++d) {
    kp.base0[d] = (int32_t)std::floor(mn[d] * inv);
    const double c = (double)((int32_t)std::floor(mx[d] * inv) - kp.base0[d] + 1);
    cells = c > cells ? c : cells;
    vol *= (long long)((mx[d] - mn[d]) * inv) + 1;
  }
  kp.axis_bits = bits_for(cells + 1.0);
  if (!(vol <= (long long)INT32_MAX)) {
    // applyFilter: "Leaf size is too small for the input dataset" -> the input is returned unfiltered
    if (n > cap) { lslam::set_error("voxel-grid output buffer too small"); return LSLAM_ERR_INVALID; }
    FM_TRY(hipStreamSynchronize(s));  // (the upload: nothing else was enqueued)
    if (out_xyzi) std::memcpy(out_xyzi, cache.in_pin.p, n * sizeof(float4));
    *n_out = n;
    return LSLAM_OK;
  }
  FM_TRY(out.reserve(n));
  FM_TRY(oc.reserve(n));
  FM_TRY(cache.done.reserve(4));
  // small clouds (a sweep's features): the whole output area comes back behind the count in ONE wait and the m points are
  // copied out of pinned memory; large ones wait for the count first and fetch exactly m points
  const bool one_wait = n * sizeof(float4) <= (size_t)1 << 20;
  size_t m = 0;
  if (one_wait) {
    FM_TRY(cache.out_pin.reserve(n));
    // (the centroids are written to pinned memory by their kernel: no copy of all n slots behind it)
    rc = run_pipeline(s, sc, cache.in_raw.p, nullptr, n, kp, 1, nullptr, cache.out_pin.p, oc.p, &m, 0, cache.done.p);
    if (rc) return rc;
    FM_TRY(hipStreamSynchronize(s));
    if (cache.done.p[1]) {
      lslam::set_error("voxel index outside its key range (non-finite point?)");
      return LSLAM_ERR_INVALID;
    }
    m = cache.done.p[0];
  } else {
    rc = run_pipeline(s, sc, cache.in_raw.p, nullptr, n, kp, 1, nullptr, out.p, oc.p, &m);
    if (rc) return rc;
  }
  if (m > cap) {
    lslam::set_error("voxel-grid output buffer too small");
    return LSLAM_ERR_INVALID;
  }
  if (out_xyzi && m) {
    if (one_wait) {
      std::memcpy(out_xyzi, cache.out_pin.p, m * sizeof(float4));
    } else if (hipMemcpyAsync(out_xyzi, out.p, m * sizeof(float4), hipMemcpyDeviceToHost, s) != hipSuccess ||
               hipStreamSynchronize(s) != hipSuccess) {
      lslam::set_error("voxel-grid download failed");
      return LSLAM_ERR_HIP;
    }
  }
  *n_out = m;
  return LSLAM_OK;
}

// Two clouds through pcl::VoxelGrid with the same leaf in ONE pass (LaserMatcher.cpp:289-301 filters the frame's corner and
// surface clouds one after the other): the clouds are two segments of one pipeline run -- each with its own min_b and its own
// "leaf too small" guard, as two VoxelGrid objects would have -- one upload, one wait, one download.  Same bits as two
// lslam_voxel_grid calls (tests/test_gpu_fmap.py).
static int voxel_grid2_impl(lslam_ctx *ctx, const void *a, size_t na, const void *b, size_t nb, size_t stride_bytes, float leaf,
                            float *out_a, size_t cap_a, size_t *n_a, float *out_b, size_t cap_b, size_t *n_b) {
  if (!ctx || !n_a || !n_b || !(leaf > 0.f) || stride_bytes < 12 || (stride_bytes & 3) || (na && !a) || (nb && !b)) {
    lslam::set_error("bad voxel-grid arguments");
    return LSLAM_ERR_INVALID;
  }
  *n_a = *n_b = 0;
  const size_t n = na + nb;
  if (n == 0) return LSLAM_OK;
  FM_TRY(hipSetDevice(lslam::ctx_device(ctx)));
  hipStream_t s = (hipStream_t)lslam_stream(ctx);
  struct Cache {
    Pin<float4> in_pin, out_pin;
    Pin<int32_t> seg_pin;
    Pin<uint32_t> done;
    Buf<float4> in_raw;
    Buf<int32_t> seg;
  };
  static std::map<hipStream_t, Cache> caches;  // per stream = per context (see voxel_filter_segments)
  static std::mutex mu;
  Cache *cache_p;
  {
    std::lock_guard<std::mutex> lk(mu);
    cache_p = &caches[(hipStream_t)lslam_stream(ctx)];
  }
  Cache &c = *cache_p;
  FM_TRY(c.in_pin.reserve(n));
  FM_TRY(c.out_pin.reserve(n));
  FM_TRY(c.seg_pin.reserve(n));
  FM_TRY(c.done.reserve(4));
  FM_TRY(c.in_raw.reserve(n));
  FM_TRY(c.seg.reserve(n));
  const void *src[2] = {a, b};
  const size_t cnt[2] = {na, nb};
  size_t at = 0;
  for (int k = 0; k < 2; ++k) {
    const char *p = static_cast<const char *>(src[k]);
    float4 *h = c.in_pin.p + at;
    if (stride_bytes == 16) {
      if (cnt[k]) std::memcpy(h, p, cnt[k] * sizeof(float4));
    } else {
      const size_t ioff = 16;  // pcl::PointXYZI keeps the intensity at byte 16
      for (size_t i = 0; i < cnt[k]; ++i) {
        float v[3], w = 0.0f;
        std::memcpy(v, p + i * stride_bytes, 12);
        if (stride_bytes >= ioff + 4) std::memcpy(&w, p + i * stride_bytes + ioff, 4);
        h[i] = make_float4(v[0], v[1], v[2], w);
      }
    }
    at += cnt[k];
  }
  FM_TRY(hipMemcpyAsync(c.in_raw.p, c.in_pin.p, n * sizeof(float4), hipMemcpyHostToDevice, s));
  // segment 0 for the first cloud's points, 1 for the second's: one launch (two fills were up to four: aligned part + tail each)
  hipLaunchKernelGGL(fm_two_segments_kernel, dim3(((unsigned)n + 255) / 256), dim3(256), 0, s, c.seg.p, (int)na, (int)n);
  size_t m = 0;
  // the centroids and their segments are WRITTEN to pinned memory by the kernel that makes them (the m of them: the count is
  // not known on the host yet -- the copies this replaces moved all n slots, 0.8 MB for a sweep's 42 k points)
  int rc = lslam::voxel_filter_segments(s, c.in_raw.p, c.seg.p, n, 2, leaf, c.out_pin.p, c.seg_pin.p, &m, true, c.done.p);
  if (rc) return rc;
  FM_TRY(hipStreamSynchronize(s));
  if (c.done.p[1]) {  // the wide key did not hold an extent: once more with the measured one (waits inside)
    rc = lslam::voxel_filter_segments(s, c.in_raw.p, c.seg.p, n, 2, leaf, c.out_pin.p, c.seg_pin.p, &m, true, nullptr);
    if (rc) return rc;
    FM_TRY(hipStreamSynchronize(s));
  } else {
    m = c.done.p[0];
  }
  size_t split = 0;  // the output is grouped by segment: the first cloud's points come first
  while (split < m && c.seg_pin.p[split] == 0) ++split;
  if (split > cap_a || m - split > cap_b) {
    lslam::set_error("voxel-grid output buffer too small");
    return LSLAM_ERR_INVALID;
  }
  if (out_a && split) std::memcpy(out_a, c.out_pin.p, split * sizeof(float4));
  if (out_b && m - split) std::memcpy(out_b, c.out_pin.p + split, (m - split) * sizeof(float4));
  *n_a = split;
  *n_b = m - split;
  return LSLAM_OK;
}
int lslam_voxel_grid2(lslam_ctx *ctx, const void *a, size_t na, const void *b, size_t nb, size_t stride_bytes, float leaf,
                      float *out_a, size_t cap_a, size_t *n_a, float *out_b, size_t cap_b, size_t *n_b) {
  const int rc = voxel_grid2_impl(ctx, a, na, b, nb, stride_bytes, leaf, out_a, cap_a, n_a, out_b, cap_b, n_b);
  if (rc != LSLAM_OK && ctx && lslam::ctx_alive(ctx)) (void)hipStreamSynchronize((hipStream_t)lslam_stream(ctx));
  return rc;
}

}  // extern "C"
