// lslam_features.hip -- feature-extraction front end on the device:
// ScanRegistration::extractFeatures (odometry/ScanRegistration.cpp:190-425) on the ring-sorted
// full-resolution cloud, one workgroup per scan ring.
//
// What is parallel and what is not.  Everything that does not depend on the order of picking runs
// across the workgroup: the neighbour tests behind setScanBuffersFor (:471-531), the curvature of
// every region point (:427-455), the stable ascending order by curvature (a rank sort: rank =
// #smaller + #equal-with-lower-index, which is what the reference's `<=` merge sort produces,
// :151-186), and pointClassify (:557-687, two 6-point line fits with the 3x3 symmetric
// eigen-solver) for every point whose curvature reaches the threshold -- exactly the set the
// reference's third loop visits.  The picking itself (marks applied in scan order, at most
// maxSurfaceFlat / maxCornerSharp picks per region with markAsPicked exclusion, regions of a ring
// sharing one mark array) is sequential by definition and is done by lane 0 on LDS-resident state:
// a few thousand integer steps per ring, all rings in parallel.  The per-ring VoxelGrid of the
// less-flat points (:398-407) is the segment filter of lslam_fmap.hip.
#include "../../include/lslam_c.h"

#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "lslam_device.hpp"
#include "lslam_internal.hpp"

namespace {

#define FX_TRY(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      char _b[400];                                                                      \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      lslam::set_error(_b);                                                              \
      return LSLAM_ERR_HIP;                                                              \
    }                                                                                    \
  } while (0)

constexpr int FX_BLOCK = 1024;  // a ring of 1 800 points: every point of every region has a thread in the parallel phases
constexpr int MAXR = 2560;  // points per ring held in LDS
constexpr int SKEY = 8192;  // sort words of a ring's regions (each region padded to a power of two); later 2 x MAXR x 3 floats of pointClassify

// PointLabel, ScanRegistration.h:22-42
enum : int {
  L_UNKNOW = 6, L_SURF_PICKED_NEAR = 3, L_CORNER_SHARP = 1, L_SURFACE_LESS_FLAT = 0, L_SURFACE_FLAT = -1,
  L_ONESIDE_FLAT = 5, L_EDGE_BROKEN = -2, L_NEAR_BLOCK = -3, L_BLIND_BLOCK = -4, L_MESSY = 9
};

struct FxArgs {
  const float4 *pts;        // n_points {x, y, z, intensity-to-copy}
  const int32_t *ranges;    // n_scans x {first, last}
  int32_t n_scans;
  int32_t nf, cr, max_sharp, max_flat;
  float surf_thr, blind_thr;
  double c175, c5, c135, c45;  // cos of the pointClassify angle gates, evaluated on the host
  // staging: every ring writes, from its own first index on, the cloud indices of its picks (index
  // stores only: the sequential picking never waits for a global load)
  int32_t *st_sharp, *st_less_sharp, *st_flat, *st_less_flat;
  int32_t *counts;          // [n_scans][4]
  float *curv_out;          // optional taps, [n_points]
  int8_t *picked_out, *label_out;
  // pointClassify by helper workgroups (fx_ring_kernel): `helpers` per ring, their verdicts in cls_g[n_points], ready[ring]
  // counts the helpers that have published
  int32_t helpers;
  int32_t *cls_g, *ready;
};

__device__ __forceinline__ float fx_pdist(float x, float y, float z) {
  return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
}

// one half of pointClassify; sign < 0: points idx-0 .. idx-cr, sign > 0: idx+cr .. idx+0
__device__ bool fx_one_sided_line(const float *sx, const float *sy, const float *sz, int idx, int cr, int sign,
                                  float (&v)[3]) {
  float c[3] = {0.f, 0.f, 0.f};
  for (int q = 0; q <= cr; ++q) {
    const int k = sign < 0 ? idx - q : idx + (cr - q);
    c[0] = __fadd_rn(c[0], sx[k]);
    c[1] = __fadd_rn(c[1], sy[k]);
    c[2] = __fadd_rn(c[2], sz[k]);
  }
  const float inv = (float)(cr + 1);
  c[0] = __fdiv_rn(c[0], inv); c[1] = __fdiv_rn(c[1], inv); c[2] = __fdiv_rn(c[2], inv);
  float A[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q <= cr; ++q) {
    const int k = sign < 0 ? idx - q : idx + (cr - q);
    const float a0 = __fsub_rn(sx[k], c[0]), a1 = __fsub_rn(sy[k], c[1]), a2 = __fsub_rn(sz[k], c[2]);
    A[0] = __fadd_rn(A[0], __fmul_rn(a0, a0));
    A[3] = __fadd_rn(A[3], __fmul_rn(a0, a1));
    A[6] = __fadd_rn(A[6], __fmul_rn(a0, a2));
    A[4] = __fadd_rn(A[4], __fmul_rn(a1, a1));
    A[7] = __fadd_rn(A[7], __fmul_rn(a1, a2));
    A[8] = __fadd_rn(A[8], __fmul_rn(a2, a2));
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) A[k] = __fdiv_rn(A[k], inv);
  float D[3], V[9];
  lslam::eig_sym3(A, D, V);
  if (!(D[2] > __fmul_rn(100.0f, D[1]) && D[2] > __fmul_rn(10000.0f, D[0]))) return false;
  v[0] = V[2]; v[1] = V[5]; v[2] = V[8];
  const float vn = fx_pdist(v[0], v[1], v[2]);
  for (int q = 0; q <= cr; ++q) {
    const int k = sign < 0 ? idx - q : idx + (cr - q);
    const float a0 = __fsub_rn(sx[k], c[0]), a1 = __fsub_rn(sy[k], c[1]), a2 = __fsub_rn(sz[k], c[2]);
    const float cx = __fsub_rn(__fmul_rn(a1, v[2]), __fmul_rn(a2, v[1]));
    const float cy = __fsub_rn(__fmul_rn(a2, v[0]), __fmul_rn(a0, v[2]));
    const float cz = __fsub_rn(__fmul_rn(a0, v[1]), __fmul_rn(a1, v[0]));
    const float distance = __fdiv_rn(fx_pdist(cx, cy, cz), vn);
    if ((double)fabsf(distance) > 0.08) return false;
  }
  return true;
}

// pointClassify from its two one-sided fits (:557-687)
__device__ int fx_point_classify(const bool line1, const float (&v1)[3], const bool line2, const float (&v2)[3], const FxArgs &a) {
  if (line1 && line2) {
    const float ab = __fadd_rn(__fadd_rn(__fmul_rn(v1[0], v2[0]), __fmul_rn(v1[1], v2[1])), __fmul_rn(v1[2], v2[2]));
    const float dis = __fmul_rn(fx_pdist(v1[0], v1[1], v1[2]), fx_pdist(v2[0], v2[1], v2[2]));
    const double diff = (double)__fdiv_rn(ab, dis);
    if (diff < a.c175 || diff > a.c5) return L_SURFACE_FLAT;
    if (diff > a.c135 && diff < a.c45) return L_CORNER_SHARP;
  }
  return (line1 || line2) ? L_ONESIDE_FLAT : L_MESSY;
}

// setRegionBuffersFor, :437-454: the curvature of ring point i
__device__ __forceinline__ float fx_curvature(const float *sx, const float *sy, const float *sz, int i, int cr) {
  const float w = (float)(-2 * cr);
  float dx = __fmul_rn(w, sx[i]), dy = __fmul_rn(w, sy[i]), dz = __fmul_rn(w, sz[i]);
  for (int q = 1; q <= cr; ++q) {
    dx = __fadd_rn(dx, __fadd_rn(sx[i + q], sx[i - q]));
    dy = __fadd_rn(dy, __fadd_rn(sy[i + q], sy[i - q]));
    dz = __fadd_rn(dz, __fadd_rn(sz[i + q], sz[i - q]));
  }
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// Bitonic sort of `count` words in LDS, in independent stretches of `seg` words (a power of two >= 2; every stretch ascending),
// by the whole workgroup.  The sort is bound by LDS traffic (a stage reads and writes every word: sixteen wavefronts' 64-bit
// accesses, 66 stages for 2 048 words were 20 us), so up to three consecutive stages -- strides j, j/2, j/4 -- are done on
// eight words in registers between one read and one write: a third of the traffic and of the barriers.  Word p lives at
// fx_at(p) = p + p / 8: a thread's eight neighbouring words then start 72 bytes after its neighbour's, not 64, and a
// wavefront's accesses spread over all banks.
__device__ __forceinline__ int fx_at(int p) { return p + (p >> 3); }
template <int M>
__device__ __forceinline__ void fx_bitonic_group(unsigned long long *w, int count, int seg, int k, int j) {
  constexpr int PER = 1 << M;
  const int jl = j >> (M - 1);  // smallest stride of the group
  for (int t = threadIdx.x; t < (count >> M); t += FX_BLOCK) {
    const int base = ((t & ~(jl - 1)) << M) | (t & (jl - 1));  // M zero bits inserted above the low log2(jl) bits
    unsigned long long x[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) x[e] = w[fx_at(base + e * jl)];
    const bool up = ((base & (seg - 1)) & k) == 0;  // (j < k: the same for the eight)
#pragma unroll
    for (int sbit = M - 1; sbit >= 0; --sbit) {
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        if (!(e & (1 << sbit))) {
          const unsigned long long a = x[e], b = x[e | (1 << sbit)];
          const bool sw = (a > b) == up;
          x[e] = sw ? b : a;
          x[e | (1 << sbit)] = sw ? a : b;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) w[fx_at(base + e * jl)] = x[e];
  }
  __syncthreads();
}
__device__ void fx_bitonic(unsigned long long *w, int count, int seg) {
  for (int k = 2; k <= seg; k <<= 1) {
    int j = k >> 1;
    while (j > 0) {
      if (j >= 4) { fx_bitonic_group<3>(w, count, seg, k, j); j >>= 3; }
      else if (j == 2) { fx_bitonic_group<2>(w, count, seg, k, j); j = 0; }
      else { fx_bitonic_group<1>(w, count, seg, k, j); j = 0; }
    }
  }
}

#ifdef LSLAM_FX_CLOCKS  // profiling build: where a ring's workgroup spends its time (100 MHz ticks, summed over rings and calls)
__device__ unsigned long long fx_clk[8];
#define FX_T(i) if (threadIdx.x == 0) { const unsigned long long _n = wall_clock64(); atomicAdd(&fx_clk[i], _n - fx_last); fx_last = _n; }
#else
#define FX_T(i)
#endif
__global__ __launch_bounds__(FX_BLOCK) void fx_ring_kernel(FxArgs a) {
#ifdef LSLAM_FX_CLOCKS
  unsigned long long fx_last = wall_clock64();
#endif
  __shared__ float sx[MAXR], sy[MAXR], sz[MAXR];
  __shared__ float curv_ring[MAXR];
  __shared__ uint16_t sorted_ring[MAXR];
  __shared__ int8_t picked[MAXR], cls_ring[MAXR], rlabel_ring[MAXR];
  __shared__ uint8_t pfl[MAXR];
  __shared__ uint16_t need[MAXR];
  __shared__ unsigned long long skey[SKEY + SKEY / 8];  // (fx_at)
  __shared__ int reg_sp[512], reg_ep[512];  // (n_feature_regions <= 512)
  __shared__ uint8_t side_ok[2 * MAXR];
  __shared__ int n_need;
  __shared__ int s_ready;
  // The first helpers * n_scans workgroups are HELPERS: pointClassify -- two 6-point line fits through an iterative 3 x 3
  // eigen-solver for ~1 000 points of a ring, a third of this kernel on its one CU while three quarters of the chip idle -- is
  // done for a ring by `helpers` workgroups on CUs of their own, each a share of the ring, published through cls_g / ready
  // while the ring's own workgroup marks and sorts.  Helpers have the lower workgroup indices (dispatched first) and wait
  // for nothing; a ring's workgroup that does not see them in time classifies for itself.  Same values either way.
  const int H = a.helpers, tid = threadIdx.x;
  const bool helper = (int)blockIdx.x < H * a.n_scans;
  const int ring = helper ? (int)blockIdx.x / H : (int)blockIdx.x - H * a.n_scans;
  if (tid == 0) n_need = 0;
  const int start = a.ranges[2 * ring], end = a.ranges[2 * ring + 1];
  const int cr = a.cr, nf = a.nf;
  int n_sharp = 0, n_less_sharp = 0, n_flat = 0, n_less = 0;  // uniform across wavefront 0
  if (!(end <= start + 2 * cr)) {  // :205-207
    const int n = end - start + 1;
    for (int i = tid; i < n; i += FX_BLOCK) {
      const float4 p = a.pts[start + i];
      sx[i] = p.x; sy[i] = p.y; sz[i] = p.z;
    }
    __syncthreads();
    // the points the third loop visits (curvature at or above the threshold), collected first -- any order: a point's class
    // depends on nothing else -- so that the eigen-solver runs on full wavefronts; one LDS atomic per wavefront
    auto collect = [&](bool wanted, int i) {
      const unsigned long long wm = __ballot(wanted);
      int wbase = 0;
      if ((tid & 63) == 0 && wm) wbase = atomicAdd(&n_need, __popcll(wm));
      wbase = __builtin_amdgcn_readfirstlane(wbase);
      if (wanted) need[wbase + __popcll(wm & ((1ull << (tid & 63)) - 1ull))] = (uint16_t)i;
    };
    // the two one-sided fits of a point are work items of their own: direction and verdict wait in the sort words' LDS
    float *side_v = reinterpret_cast<float *>(skey);  // [2 n_need][3]
    auto fit_sides = [&]() {
      for (int t = tid; t < 2 * n_need; t += FX_BLOCK) {
        float v[3] = {0.f, 0.f, 0.f};
        const bool ok = fx_one_sided_line(sx, sy, sz, need[t >> 1], cr, (t & 1) ? +1 : -1, v);
        side_v[3 * t] = v[0]; side_v[3 * t + 1] = v[1]; side_v[3 * t + 2] = v[2];
        side_ok[t] = ok ? 1 : 0;
      }
      __syncthreads();
    };
    auto class_of = [&](int t) {
      const float v1[3] = {side_v[6 * t], side_v[6 * t + 1], side_v[6 * t + 2]};
      const float v2[3] = {side_v[6 * t + 3], side_v[6 * t + 4], side_v[6 * t + 5]};
      return fx_point_classify(side_ok[2 * t] != 0, v1, side_ok[2 * t + 1] != 0, v2, a);
    };
    if (helper) {  // block-uniform
      const int part = (int)blockIdx.x % H, total = n - 2 * cr, chunk = (total + H - 1) / H;
      const int i0 = cr + part * chunk, i1 = min(n - cr, i0 + chunk);
      for (int ib = i0; ib < i1; ib += FX_BLOCK) {
        const int i = ib + tid;
        collect(i < i1 && !(fx_curvature(sx, sy, sz, min(i, i1 - 1), cr) < a.surf_thr), i);
      }
      __syncthreads();
      fit_sides();
      for (int t = tid; t < n_need; t += FX_BLOCK)
        __hip_atomic_store(a.cls_g + start + need[t], class_of(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // every thread's own stores out at agent scope BEFORE the barrier: thread 0's release below orders thread 0's stores, and
      // the barrier's workgroup-scope fence does not wait for the other wavefronts' stores to reach the L2 -- a ring workgroup
      // on another CU could see the count and still load a stale class
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.ready + ring, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    // ---- neighbour tests of setScanBuffersFor, one per point pair (i, i+1) ------------------------
    // bit0: cos(angle(p_i, p_i+1)) < blindThreshold   bit1: |p_i+1 - p_i|^2 > 1.0
    // bit2: depth_i > depth_i+1                          bit3: |p_i-1 - p_i|^2 / |p_i+1 - p_i|^2 < 0.2
    for (int i = tid; i < n - 1; i += FX_BLOCK) {
      const float x = sx[i], y = sy[i], z = sz[i], xn = sx[i + 1], yn = sy[i + 1], zn = sz[i + 1];
      const float ab = __fadd_rn(__fadd_rn(__fmul_rn(x, xn), __fmul_rn(y, yn)), __fmul_rn(z, zn));
      const float d1 = fx_pdist(x, y, z), d2 = fx_pdist(xn, yn, zn);
      const float cosang = __fdiv_rn(ab, __fmul_rn(d1, d2));
      const float dx = __fsub_rn(xn, x), dy = __fsub_rn(yn, y), dz = __fsub_rn(zn, z);
      const float diff_next = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      uint8_t f = 0;
      if (cosang < a.blind_thr) f |= 1;
      if ((double)diff_next > 1.0) f |= 2;
      if (d1 > d2) f |= 4;
      if (i > 0) {
        const float px = __fsub_rn(sx[i - 1], x), py = __fsub_rn(sy[i - 1], y), pz = __fsub_rn(sz[i - 1], z);
        const float diff_prev = __fadd_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)), __fmul_rn(pz, pz));
        if ((double)__fdiv_rn(diff_prev, diff_next) < 0.2) f |= 8;
      }
      pfl[i] = f;
    }
    __syncthreads();
    // The marks of setScanBuffersFor (:477-530) are applied in scan order, later writes over earlier ones, and two of them read
    // what is there -- but a step at pair i only touches points within cr of i, so every point replays, in order, the <= 2 cr
    // steps that can reach it and nothing else (round 5; lane 0 applying them one after the other was a sixth of the kernel).
    for (int p = tid; p < n; p += FX_BLOCK) {
      int v = 0;
      for (int i = 0; i < cr; ++i) {  // :477-495, the ring's two ends (only one value is written: any order)
        if ((pfl[i] & 1) && p >= i && p <= i + cr) v = L_BLIND_BLOCK;
        if ((pfl[n - 1 - i - 1] & 1) && p >= n - 1 - i - cr && p <= n - 1 - i) v = L_BLIND_BLOCK;
      }
      const int i_lo = max(cr, p - cr), i_hi = min(n - 2 - cr, p + cr - 1);
      for (int i = i_lo; i <= i_hi; ++i) {  // :497-530
        const int f = pfl[i];
        if ((f & 3) == 0) continue;
        if (f & 1) {
          if (p >= i - cr + 1 && p <= i + cr) v = L_BLIND_BLOCK;
        } else if (f & 4) {
          if (p == i + 1 && v > L_NEAR_BLOCK && (f & 8)) v = L_EDGE_BROKEN;
          if (p >= i - cr + 1 && p <= i) v = L_NEAR_BLOCK;
        } else {
          if (p == i && v > L_NEAR_BLOCK && (f & 8)) v = L_EDGE_BROKEN;
          if (p >= i + 1 && p <= i + cr) v = L_NEAR_BLOCK;
        }
      }
      picked[p] = (int8_t)v;
    }
    __syncthreads();
    FX_T(0)
    if (a.picked_out)
      for (int i = tid; i < n; i += FX_BLOCK) a.picked_out[start + i] = picked[i];
    // ---- everything about the regions that does not depend on the picking, for ALL regions at once -------------
    // (curvature, the stable rank inside the region, pointClassify: a region is ~300 points, the ring ~1800 -- region
    // after region a 256-thread workgroup ran these in twelve rounds, most of the time of the kernel)
    const size_t S = (size_t)start + cr, E = (size_t)end - cr;
    // the regions' bounds (:208-209), once per ring: two 64-bit divisions each, which every thread of every phase repeated
    for (int j = tid; j < nf; j += FX_BLOCK) {
      reg_sp[j] = (int)((S * (size_t)(nf - j) + E * (size_t)j) / (size_t)nf);
      reg_ep[j] = (int)((S * (size_t)(nf - 1 - j) + E * (size_t)(j + 1)) / (size_t)nf) - 1;
    }
    __syncthreads();
    auto region_of = [&](int j, int &sp, int &ep) {
      sp = reg_sp[j];
      ep = reg_ep[j];
    };
    {
      for (int i = cr + tid; i <= n - 1 - cr; i += FX_BLOCK) {
        curv_ring[i] = fx_curvature(sx, sy, sz, i, cr);
        rlabel_ring[i] = L_UNKNOW;
      }
      __syncthreads();
      FX_T(1)
      // The stable ascending order by curvature inside every region (what the reference's `<=` merge sort yields, :151-186):
      // curvature is a sum of squares, so its bit pattern orders like its value, and (bits << 16 | index in region) sorts by
      // curvature, then index.  All regions at once, each in its own power-of-two stretch of `skey`, bitonic (round 5: the rank
      // sort this replaces -- every point counting the smaller ones of its region -- was a quarter of the kernel).
      int rs_max = 0;
      for (int j = 0; j < nf; ++j) {
        int sp, ep;
        region_of(j, sp, ep);
        rs_max = max(rs_max, ep - sp + 1);
      }
      int rpad = 2, rlg = 1;
      while (rpad < rs_max) { rpad <<= 1; ++rlg; }  // block-uniform; nf * rpad <= SKEY (regions are equal to +-1; the host bounds nf)
      for (int t = tid; t < nf * rpad; t += FX_BLOCK) {
        const int j = t >> rlg, r = t & (rpad - 1);
        int sp, ep;
        region_of(j, sp, ep);
        const int rs = ep > sp ? ep - sp + 1 : 0;  // :210 skips a region of one point
        unsigned long long w = ~0ull;
        bool wanted = false;
        int i = 0;
        if (r < rs) {
          i = sp - start + r;
          const float c = curv_ring[i];
          w = ((unsigned long long)__float_as_uint(c) << 16) | (unsigned long long)r;
          cls_ring[i] = (int8_t)L_UNKNOW;
          wanted = !(c < a.surf_thr);
          if (a.curv_out) a.curv_out[start + i] = c;
        }
        skey[fx_at(t)] = w;
        collect(wanted, i);
      }
      __syncthreads();
      fx_bitonic(skey, nf * rpad, rpad);
      for (int t = tid; t < nf * rpad; t += FX_BLOCK) {
        const int j = t >> rlg, r = t & (rpad - 1);
        int sp, ep;
        region_of(j, sp, ep);
        const int rs = ep > sp ? ep - sp + 1 : 0;
        if (r < rs) sorted_ring[sp - start + r] = (uint16_t)(skey[fx_at(t)] & 0xFFFFull);
      }
      __syncthreads();
      FX_T(5)
      // pointClassify of the collected points: the helpers' verdicts if they have all published, else here
      bool have = false;
      if (H > 0) {
        if (tid == 0) {
          int ok = 0;
          for (int spin = 0; spin < (1 << 14) && !ok; ++spin) {
            ok = __hip_atomic_load(a.ready + ring, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= H;
            if (!ok) __builtin_amdgcn_s_sleep(16);
          }
          s_ready = ok;
        }
        __syncthreads();
        have = s_ready != 0;
      }
#ifdef LSLAM_FX_CLOCKS
      if (tid == 0) { atomicAdd(&fx_clk[6], (unsigned long long)n_need); atomicAdd(&fx_clk[7], 1ull); }
#endif
      if (have) {
        for (int t = tid; t < n_need; t += FX_BLOCK)
          cls_ring[need[t]] = (int8_t)__hip_atomic_load(a.cls_g + start + need[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        fit_sides();
        for (int t = tid; t < n_need; t += FX_BLOCK) cls_ring[need[t]] = (int8_t)class_of(t);
      }
      __syncthreads();
      FX_T(2)
    }
    // ---- regions: the picking, in order (regions of a ring share one mark array) -----------------------------
    for (int j = 0; j < nf; ++j) {
      int sp, ep;
      region_of(j, sp, ep);
      if (ep <= sp) continue;  // block-uniform
      const int rs = ep - sp + 1, r0 = sp - start;  // r0: ring-relative index of the region's first point
      const float *curv = curv_ring + r0;            // region-relative views
      const uint16_t *sorted = sorted_ring + r0;
      const int8_t *cls = cls_ring + r0;
      int8_t *rlabel = rlabel_ring + r0;
      if (tid < 64) {  // wavefront 0; the counters are wave-uniform
        const unsigned long long lt_mask = tid == 0 ? 0ull : (~0ull >> (64 - tid));
        // flat surface features, :268-284 -- sequential: every pick excludes its neighbourhood
        // A pick excludes its neighbourhood from the later ones, so the picks are sequential -- but the candidates are not: 64
        // of them (ascending curvature) are tested at once, the first eligible lane is the pick, the lanes within cr of it drop
        // out in registers, its marks are written by 2 cr + 1 lanes (round 5: lane 0 walking the order with two dependent LDS
        // reads per candidate, most of them neighbours of a pick, was a seventh of the kernel).
        int surf_picked = 0;
        for (int base = 0; base < rs && surf_picked < a.max_flat; base += 64) {
          const int k = base + tid;
          const int r = k < rs ? (int)sorted[k] : 0, si = r0 + r;
          bool elig = k < rs && curv[r] < a.surf_thr && picked[si] != L_SURF_PICKED_NEAR;
          const bool more = __ballot(k < rs && curv[r] < a.surf_thr) != 0ull;  // ascending: a chunk without one ends the search
          unsigned long long m;
          while (surf_picked < a.max_flat && (m = __ballot(elig)) != 0ull) {
            const int L = __builtin_ctzll(m);
            const int si_L = __builtin_amdgcn_readlane(si, L);
            if (tid == L) {
              rlabel[r] = L_SURFACE_FLAT;
              a.st_flat[start + n_flat + surf_picked] = sp + r;
            }
            if (tid <= 2 * cr) picked[si_L - cr + tid] = L_SURF_PICKED_NEAR;  // markAsPicked, :533-555
            elig = elig && abs(si - si_L) > cr;
            ++surf_picked;
          }
          if (!more) break;
        }
        n_flat += surf_picked;  // (wave-uniform)
        FX_T(3)
        // less flat + broken edges, :286-302 -- an ordered compaction, 64 region points at a time
        int n_low = 0;
        for (int base = 0; base < rs; base += 64) {
          const int k = base + tid;
          const bool valid = k < rs;
          const bool low = valid && curv[valid ? k : 0] < a.surf_thr;
          const bool eb = valid && picked[r0 + (valid ? k : 0)] == L_EDGE_BROKEN;
          const unsigned long long m1 = __ballot(low), m2 = __ballot(eb);
          if (low) {
            a.st_less_flat[start + n_less + __popcll(m1 & lt_mask)] = sp + k;
            if (rlabel[k] != L_SURFACE_FLAT) rlabel[k] = L_SURFACE_LESS_FLAT;
          }
          if (eb) {
            const int pos = __popcll(m2 & lt_mask);
            a.st_sharp[start + n_sharp + pos] = sp + k;
            a.st_less_sharp[start + n_less_sharp + pos] = sp + k;
            rlabel[k] = L_CORNER_SHARP;
          }
          n_less += __popcll(m1);
          n_low += __popcll(m1);
          n_sharp += __popcll(m2);
          n_less_sharp += __popcll(m2);
        }
        // classified features in descending curvature, :304-354: the loop only reads the marks and
        // counts its own picks, so it is an ordered compaction too (the saturating counters of the
        // reference equal min(max, number of earlier candidates of that kind))
        int run_surf = 0, run_corner = 0;
        const int T = rs - n_low;  // points with curvature >= threshold: sorted[n_low .. rs)
        for (int base = 0; base < T; base += 64) {
          const int t = base + tid;
          const bool valid = t < T;
          const int r = sorted[valid ? rs - 1 - t : 0], si = r0 + r;
          const int lab = valid ? (int)cls[r] : L_MESSY;
          const bool is_f = lab == L_SURFACE_FLAT, is_o = lab == L_ONESIDE_FLAT;
          const bool is_c = lab == L_CORNER_SHARP && picked[si] > L_EDGE_BROKEN;
          const unsigned long long ms = __ballot(is_f || is_o), mc = __ballot(is_c), mo = __ballot(is_o);
          const int surf_before = run_surf + __popcll(ms & lt_mask);
          const int corner_before = run_corner + __popcll(mc & lt_mask);
          // flat picks among the one-side-flat candidates: those with fewer than max earlier surf-like ones
          const bool flat_pick = is_o && surf_before < a.max_flat;
          const bool sharp_pick = is_c && corner_before < a.max_sharp;
          const unsigned long long mfp = __ballot(flat_pick), msp = __ballot(sharp_pick);
          if (is_f || is_o) a.st_less_flat[start + n_less + __popcll(ms & lt_mask)] = sp + r;
          if (flat_pick) a.st_flat[start + n_flat + __popcll(mfp & lt_mask)] = sp + r;
          if (is_c) a.st_less_sharp[start + n_less_sharp + __popcll(mc & lt_mask)] = sp + r;
          if (sharp_pick) a.st_sharp[start + n_sharp + __popcll(msp & lt_mask)] = sp + r;
          if (is_f) rlabel[r] = L_SURFACE_FLAT;
          else if (is_o) rlabel[r] = L_ONESIDE_FLAT;
          else if (is_c) rlabel[r] = L_CORNER_SHARP;
          (void)mo;
          n_less += __popcll(ms);
          n_flat += __popcll(mfp);
          n_less_sharp += __popcll(mc);
          n_sharp += __popcll(msp);
          run_surf += __popcll(ms);
          run_corner += __popcll(mc);
        }
        FX_T(4)
      }
      __syncthreads();
      if (a.label_out)
        for (int r = tid; r < rs; r += FX_BLOCK) a.label_out[sp + r] = rlabel[r];
      __syncthreads();
    }
  }
  if (tid == 0 && !helper) {
    a.counts[4 * ring + 0] = n_sharp;
    a.counts[4 * ring + 1] = n_less_sharp;
    a.counts[4 * ring + 2] = n_flat;
    a.counts[4 * ring + 3] = n_less;
  }
}

// ring r's list (written from its first index) -> out[off[r] ...]; seg_out optional
__global__ void fx_compact_kernel(const float4 *pts, const int32_t *stage, const int32_t *ranges, const int32_t *off,
                                  int n_scans, int total, float4 *out, int32_t *seg_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = n_scans - 1;  // last ring with off <= i
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  out[i] = pts[stage[ranges[2 * lo] + (i - off[lo])]];
  if (seg_out) seg_out[i] = lo;
}


// ---- the lists of a sweep, without the host in between (round 5) ---------------------------------------------------------
// Until round 5 the host fetched the rings' counts, made the offsets, launched four compactions and ran the per-ring
// VoxelGrid of the less-flat points (:398-407) through the general segment filter of lslam_fmap.hip -- a 64-bit merge sort of
// all rings' points at once, sixteen launches and three waits for what is 64 independent problems of <= 2 560 points.  Now:
//   fx_lists_block         one workgroup per small list (sharp, less sharp, flat): the rings' offsets by a scan in LDS, the
//                          points gathered in ring order STRAIGHT INTO PINNED HOST MEMORY, the total next to them (the last
//                          three workgroups of the next kernel's launch)
//   fx_ring_voxel_kernel   one workgroup per ring: pcl::VoxelGrid::applyFilter on the ring's less-flat list entirely in LDS --
//                          bounding box, the "leaf too small" guard, voxel index ijk0 + ijk1 * div0 + ijk2 * div0 * div1,
//                          a bitonic sort of (index, position in the list), heads, their scan, centroids summed in list order
//   fx_lessflat_out_kernel one workgroup per ring: the ring's centroids behind those of the rings before it, into pinned memory
// and ONE wait.  Same arithmetic, same order as the segment filter (tests/test_gpu_features.py holds both against the oracle).
constexpr int VX_PAD = 4096;  // MAXR rounded up to a power of two
static_assert(MAXR <= VX_PAD && VX_PAD <= 4096, "the list position takes the low 12 bits of the sort word");

struct FxOutArgs {
  const float4 *pts;
  const int32_t *ranges, *counts;  // counts[4 * ring + list]
  const int32_t *stage[4];         // per list: the rings' picks (cloud indices), each ring's from its first index on
  int32_t n_scans, cap;            // cap: points a list's slice of `host` holds
  float inv_leaf;                  // 1 / lessFlatFilterSize, as pcl::VoxelGrid forms it
  float4 *vox_stage;               // [n_points] ring r's centroids from ranges[2 r] on
  int32_t *ring_out;               // [n_scans] centroids per ring
  float4 *host;                    // pinned: 16 header slots, then four slices of `cap` points
  uint32_t *hdr;                   // = (uint32_t *)host: [0..2] totals of the small lists, [3] less-flat points out, [4] error
};

// exclusive scan of v over the workgroup's 1024 threads (LDS part[>= 16]); returns this thread's prefix, *total = the sum
__device__ int fx_block_scan(int v, int *part, int *total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  __syncthreads();  // (part may still be read from an earlier call)
  if (lane == 63) part[wave] = incl;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < FX_BLOCK / 64; ++w) {
    const int c = part[w];
    before += w < wave ? c : 0;
    all += c;
  }
  *total = all;
  return before + incl - v;
}

__device__ void fx_lists_block(const FxOutArgs &a, const int k) {
  __shared__ int part[FX_BLOCK];
  __shared__ int32_t off[4096 + 1];
  const int tid = threadIdx.x;
  const int per = (a.n_scans + FX_BLOCK - 1) / FX_BLOCK;
  const int r0 = min(a.n_scans, tid * per), r1 = min(a.n_scans, r0 + per);
  int sum = 0;
  for (int r = r0; r < r1; ++r) sum += a.counts[4 * r + k];
  int total;
  int run = fx_block_scan(sum, part, &total);
  for (int r = r0; r < r1; ++r) {
    off[r] = run;
    run += a.counts[4 * r + k];
  }
  if (tid == 0) {
    off[a.n_scans] = total;
    a.hdr[k] = (uint32_t)total;
    if (total > a.cap) a.hdr[4] = 2u;  // (cannot happen: a list holds a point at most once)
  }
  __syncthreads();
  float4 *out = a.host + 16 + (size_t)k * a.cap;
  const int32_t *stage = a.stage[k];
  for (int i = tid; i < min(total, a.cap); i += FX_BLOCK) {
    int lo = 0, hi = a.n_scans - 1;  // last ring with off <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    out[i] = a.pts[stage[a.ranges[2 * lo] + (i - off[lo])]];
  }
}

#ifdef LSLAM_FX_CLOCKS
__device__ unsigned long long vx_clk[8];
#define VX_T(i) if (threadIdx.x == 0) { const unsigned long long _n = wall_clock64(); atomicAdd(&vx_clk[i], _n - vx_last); vx_last = _n; }
#else
#define VX_T(i)
#endif
// one launch: workgroups [0, n_scans) filter a ring each, the three after them gather the small lists (fx_lists_block)
__global__ __launch_bounds__(FX_BLOCK) void fx_ring_voxel_kernel(FxOutArgs a) {
#ifdef LSLAM_FX_CLOCKS
  unsigned long long vx_last = wall_clock64();
#endif
  if ((int)blockIdx.x >= a.n_scans) {  // block-uniform
    fx_lists_block(a, (int)blockIdx.x - a.n_scans);
    return;
  }
  __shared__ float4 sp[MAXR];
  __shared__ unsigned long long key[VX_PAD + VX_PAD / 8];  // (fx_at)
  __shared__ float wred[6][FX_BLOCK / 64];
  __shared__ int part[FX_BLOCK];
  __shared__ int s_err;
  const int ring = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int start = a.ranges[2 * ring];
  const int n = min(a.counts[4 * ring + 3], MAXR);
  if (n <= 0) {  // block-uniform
    if (tid == 0) a.ring_out[ring] = 0;
    return;
  }
  if (tid == 0) s_err = 0;
  // ---- the ring's list and its bounding box (getMinMax3D) ----------------------------------------------------------------
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = tid; i < n; i += FX_BLOCK) {
    const float4 p = a.pts[a.stage[3][start + i]];
    sp[i] = p;
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64));
    }
    if (lane == 0) { wred[d][wave] = mn[d]; wred[3 + d][wave] = mx[d]; }
  }
  __syncthreads();
  long long vol = 1;
  int32_t b0[3], div[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float lo = wred[d][0], hi = wred[3 + d][0];
#pragma unroll
    for (int w = 1; w < FX_BLOCK / 64; ++w) { lo = fminf(lo, wred[d][w]); hi = fmaxf(hi, wred[3 + d][w]); }
    // applyFilter's guard and its min_b / div_b (voxel_grid.hpp), as fm_extent_kernel forms them
    vol *= (long long)(__fmul_rn(__fsub_rn(hi, lo), a.inv_leaf)) + 1;
    b0[d] = (int32_t)floorf(__fmul_rn(lo, a.inv_leaf));
    div[d] = (int32_t)floorf(__fmul_rn(hi, a.inv_leaf)) - b0[d] + 1;
  }
  const bool filtered = vol <= 2147483647ll;  // else: PCL warns and hands the cloud back unfiltered
  VX_T(0)
  int npad = 64;
  while (npad < n) npad <<= 1;
  // ---- sort words: voxel index, then the position in the list (a stable sort by index) -----------------------------------
  for (int i = tid; i < npad; i += FX_BLOCK) {
    unsigned long long w = ~0ull;
    if (i < n) {
      unsigned long long idx = (unsigned long long)i;  // unfiltered: every point its own voxel, in list order
      if (filtered) {
        const float4 p = sp[i];
        const float v[3] = {p.x, p.y, p.z};
        int32_t r[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          r[d] = (int32_t)floorf(__fmul_rn(v[d], a.inv_leaf)) - b0[d];
          if (r[d] < 0 || r[d] >= div[d]) {  // a non-finite coordinate
            s_err = 1;
            r[d] = 0;
          }
        }
        idx = ((unsigned long long)r[2] * (unsigned long long)div[1] + (unsigned long long)r[1]) * (unsigned long long)div[0] + (unsigned long long)r[0];
      }
      w = (idx << 12) | (unsigned long long)i;
    }
    key[fx_at(i)] = w;
  }
  __syncthreads();
  VX_T(1)
  fx_bitonic(key, npad, npad);
  VX_T(2)
  // ---- heads, their places, centroids over the members in list order ------------------------------------------------------
  constexpr int PER = VX_PAD / FX_BLOCK;
  const int per = max(1, npad / FX_BLOCK);  // consecutive entries per thread
  int heads = 0;
  for (int u = 0; u < PER; ++u) {
    const int i = tid * per + u;
    if (u < per && i < n) heads += (i == 0 || (key[fx_at(i)] >> 12) != (key[fx_at(i - 1)] >> 12)) ? 1 : 0;
  }
  int total;
  int pos = fx_block_scan(heads, part, &total);
  VX_T(3)
  for (int u = 0; u < PER; ++u) {
    const int i = tid * per + u;
    if (!(u < per && i < n)) continue;
    const unsigned long long vi = key[fx_at(i)] >> 12;
    if (!(i == 0 || vi != (key[fx_at(i - 1)] >> 12))) continue;
    float4 s = sp[(int)(key[fx_at(i)] & 4095ull)];  // PCL starts from a zero vector: 0 + x = x
    int m = i + 1;
    for (; m < n && (key[fx_at(m)] >> 12) == vi; ++m) {
      const float4 q = sp[(int)(key[fx_at(m)] & 4095ull)];
      s.x = __fadd_rn(s.x, q.x);
      s.y = __fadd_rn(s.y, q.y);
      s.z = __fadd_rn(s.z, q.z);
      s.w = __fadd_rn(s.w, q.w);
    }
    const float cnt = (float)(m - i);
    a.vox_stage[start + pos] = make_float4(__fdiv_rn(s.x, cnt), __fdiv_rn(s.y, cnt), __fdiv_rn(s.z, cnt), __fdiv_rn(s.w, cnt));
    ++pos;
  }
  VX_T(4)
  if (tid == 0) {
    a.ring_out[ring] = total;
    if (s_err) a.hdr[4] = 1u;
  }
}

__global__ __launch_bounds__(FX_BLOCK) void fx_lessflat_out_kernel(FxOutArgs a) {
  __shared__ int part[FX_BLOCK];
  const int ring = blockIdx.x, tid = threadIdx.x;
  int before = 0;
  for (int r = tid; r < ring; r += FX_BLOCK) before += a.ring_out[r];
  int off;
  (void)fx_block_scan(before, part, &off);
  const int cnt = a.ring_out[ring], start = a.ranges[2 * ring];
  float4 *out = a.host + 16 + (size_t)3 * a.cap;
  for (int i = tid; i < cnt && off + i < a.cap; i += FX_BLOCK) out[off + i] = a.vox_stage[start + i];
  if (ring == a.n_scans - 1 && tid == 0) a.hdr[3] = (uint32_t)(off + cnt);
}

// ---- MultiScanRegistration::process (MultiScanRegistration.cpp:94-190, no IMU) -------------------
struct MsArgs {
  const float4 *in;   // raw driver cloud {x, y, z, *}
  int n, n_rings;
  float lower, factor, scan_period, start_ori, end_ori;
  float4 *out;        // {x', y', z', ring + relTime} in the registration's swapped axes, arrival order
  int32_t *ring;      // -1 = dropped
  float *ori_raw;
  int32_t *first_half;  // index of the first kept point at which halfPassed flips
};

__device__ __forceinline__ float ms_mode_a(float ori, float start_ori) {  // :150-156
  if ((double)ori < (double)start_ori - M_PI / 2) ori = (float)((double)ori + 2 * M_PI);
  else if ((double)ori > (double)start_ori + M_PI * 3 / 2) ori = (float)((double)ori - 2 * M_PI);
  return ori;
}

__global__ void ms_prep_kernel(MsArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float4 p = a.in[i];
  const float x = p.y, y = p.z, z = p.x;  // :127-129
  int ring = -1;
  float ori = 0.0f;
  if (isfinite(x) && isfinite(y) && isfinite(z) &&
      !((double)__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)) < 0.0001)) {
    const float angle = atanf(__fdiv_rn(y, sqrtf(__fadd_rn(__fmul_rn(x, x), __fmul_rn(z, z)))));
    const int id = (int)((((double)__fmul_rn(angle, 180.0f) / M_PI) - (double)a.lower) * (double)a.factor + 0.5);
    if (id < a.n_rings && id >= 0) {
      ring = id;
      ori = -atan2f(x, z);
      const float oa = ms_mode_a(ori, a.start_ori);
      if ((double)__fsub_rn(oa, a.start_ori) > M_PI) atomicMin(a.first_half, i);
    }
  }
  a.ring[i] = ring;
  a.ori_raw[i] = ori;
}

__global__ void ms_final_kernel(MsArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int ring = a.ring[i];
  if (ring < 0) return;
  const float4 p = a.in[i];
  float ori = a.ori_raw[i];
  if (i <= *a.first_half) {  // the point that flips halfPassed is itself still handled as "not passed"
    ori = ms_mode_a(ori, a.start_ori);
  } else {                   // :157-165
    ori = (float)((double)ori + 2 * M_PI);
    if ((double)ori < (double)a.end_ori - M_PI * 3 / 2) ori = (float)((double)ori + 2 * M_PI);
    else if ((double)ori > (double)a.end_ori + M_PI / 2) ori = (float)((double)ori - 2 * M_PI);
  }
  const float rel = __fdiv_rn(__fmul_rn(a.scan_period, __fsub_rn(ori, a.start_ori)), __fsub_rn(a.end_ori, a.start_ori));
  a.out[i] = make_float4(p.y, p.z, p.x, __fadd_rn((float)ring, rel));
}

}  // namespace

extern "C" {
#ifdef LSLAM_FX_CLOCKS
int lslam_debug_fx_clocks(double out[16]) {
  unsigned long long h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(fx_clk), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 8; ++i) out[i] = (double)h[i];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(vx_clk), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 8; ++i) out[8 + i] = (double)h[i];
  return 0;
}
#endif


void lslam_reg_default_params(lslam_reg_params *p) {
  if (!p) return;
  // RegistrationParams ctor defaults, ScanRegistration.h:49-57 / ScanRegistration.cpp:14-30
  p->n_feature_regions = 6;
  p->curvature_region = 5;
  p->max_corner_sharp = 2;
  p->max_surface_flat = 4;
  p->less_flat_filter_size = 0.2f;
  p->surface_curvature_threshold = 0.02f;
  const float deg = 0.5f;                              // blindDegreeThreshold
  const float rad = (float)(deg * M_PI / 180.0);       // deg2rad(float), util/math_utils.h:37
  p->blind_threshold = (float)std::cos((double)rad);
  p->reserved = 0;
}

static int extract_features_impl(lslam_ctx *ctx, const void *cloud, size_t n_points, size_t stride_bytes,
                                 size_t intensity_offset_bytes, const int32_t *scan_ranges, size_t n_scans,
                                 const lslam_reg_params *params, float *sharp, float *less_sharp, float *flat,
                                 float *less_flat, size_t counts[4], float *curvature_out, int8_t *picked_out,
                                 int8_t *label_out, lslam_fset *dev_out);
int lslam_extract_features_dev(lslam_ctx *ctx, const void *cloud, size_t n_points, size_t stride_bytes,
                               size_t intensity_offset_bytes, const int32_t *scan_ranges, size_t n_scans,
                               const lslam_reg_params *params, lslam_fset *out, size_t counts[4]) {
  if (!out) {
    lslam::set_error("no feature set to extract into");
    return LSLAM_ERR_INVALID;
  }
  size_t local[4];
  const int rc = extract_features_impl(ctx, cloud, n_points, stride_bytes, intensity_offset_bytes, scan_ranges, n_scans, params, nullptr,
                                       nullptr, nullptr, nullptr, counts ? counts : local, nullptr, nullptr, nullptr, out);
  if (rc != LSLAM_OK && ctx && lslam::ctx_alive(ctx)) (void)hipStreamSynchronize((hipStream_t)lslam_stream(ctx));
  return rc;
}
int lslam_extract_features(lslam_ctx *ctx, const void *cloud, size_t n_points, size_t stride_bytes,
                           size_t intensity_offset_bytes, const int32_t *scan_ranges, size_t n_scans,
                           const lslam_reg_params *params, float *sharp, float *less_sharp, float *flat,
                           float *less_flat, size_t counts[4], float *curvature_out, int8_t *picked_out,
                           int8_t *label_out) {
  const int rc = extract_features_impl(ctx, cloud, n_points, stride_bytes, intensity_offset_bytes, scan_ranges, n_scans, params, sharp,
                                       less_sharp, flat, less_flat, counts, curvature_out, picked_out, label_out, nullptr);
  // a failure half way leaves copies out of / into the pinned staging in flight: the next call shares it
  if (rc != LSLAM_OK && ctx && lslam::ctx_alive(ctx)) (void)hipStreamSynchronize((hipStream_t)lslam_stream(ctx));
  return rc;
}
static int extract_features_impl(lslam_ctx *ctx, const void *cloud, size_t n_points, size_t stride_bytes,
                                 size_t intensity_offset_bytes, const int32_t *scan_ranges, size_t n_scans,
                                 const lslam_reg_params *params, float *sharp, float *less_sharp, float *flat,
                                 float *less_flat, size_t counts[4], float *curvature_out, int8_t *picked_out,
                                 int8_t *label_out, lslam_fset *dev_out) {
  if (!ctx || !lslam::ctx_alive(ctx) || !scan_ranges || !counts || (n_points && !cloud) || stride_bytes < 12 ||
      (stride_bytes & 3) || intensity_offset_bytes + 4 > stride_bytes || n_scans == 0 || n_scans > 4096) {
    lslam::set_error("bad feature-extraction arguments");
    return LSLAM_ERR_INVALID;
  }
  lslam_reg_params prm;
  if (params) prm = *params; else lslam_reg_default_params(&prm);
  if (prm.curvature_region < 1 || prm.curvature_region > 16 || prm.n_feature_regions < 1 || prm.n_feature_regions > 512) {  // (512: fx_ring_kernel's sort words)
    lslam::set_error("bad registration parameters");
    return LSLAM_ERR_INVALID;
  }
  for (int k = 0; k < 4; ++k) counts[k] = 0;
  if (dev_out) {
    if (dev_out->device != lslam::ctx_device(ctx)) {
      lslam::set_error("the feature set lives on another device");
      return LSLAM_ERR_INVALID;
    }
    for (int k = 0; k < 4; ++k) dev_out->counts[k] = 0;
  }
  for (size_t s = 0; s < n_scans; ++s) {
    const int32_t a = scan_ranges[2 * s], b = scan_ranges[2 * s + 1];
    if (b < a) continue;  // empty ring (MultiScanRegistration.cpp:184-189 gives {size, size - 1})
    if (a < 0 || (size_t)b >= n_points) {
      lslam::set_error("scan range outside the cloud");
      return LSLAM_ERR_INVALID;
    }
    if (b - a + 1 > MAXR) {
      lslam::set_error("a scan ring has more points than the kernel holds in LDS (2560)");
      return LSLAM_ERR_INVALID;
    }
  }
  if (n_points == 0) return LSLAM_OK;
  FX_TRY(hipSetDevice(lslam::ctx_device(ctx)));
  hipStream_t s = (hipStream_t)lslam_stream(ctx);
  // device scratch and the pinned staging area of the input are kept per device between calls (a
  // sweep arrives every 100 ms)
  struct Cache { char *p = nullptr; size_t cap = 0; float4 *pin = nullptr; size_t pin_cap = 0; float4 *pout = nullptr; size_t pout_cap = 0; };
  static std::map<hipStream_t, Cache> caches;  // per stream = per context: contexts of one device run on their own threads
  static std::mutex mu;
  Cache *cache_p;
  {
    std::lock_guard<std::mutex> lk(mu);
    cache_p = &caches[s];
  }
  Cache &cache = *cache_p;
  const size_t np4 = n_points * sizeof(float4);
  if (n_points > cache.pin_cap) {
    if (cache.pin) (void)hipHostFree(cache.pin);
    cache.pin = nullptr;
    cache.pin_cap = 0;
    FX_TRY(hipHostMalloc((void **)&cache.pin, (n_points + n_points / 4) * sizeof(float4), hipHostMallocDefault));
    cache.pin_cap = n_points + n_points / 4;
  }
  // the outputs are WRITTEN to pinned memory by the kernels that make them: sixteen header slots (totals, error), then a
  // slice of n_points per list
  if (4 * n_points + 16 > cache.pout_cap) {
    if (cache.pout) (void)hipHostFree(cache.pout);
    cache.pout = nullptr;
    cache.pout_cap = 0;
    const size_t want = 4 * (n_points + n_points / 4) + 16;
    FX_TRY(hipHostMalloc((void **)&cache.pout, want * sizeof(float4), hipHostMallocDefault));
    cache.pout_cap = want;
  }
  // pack {x, y, z, intensity-to-copy} (toXYZI, util/pcl_util.h:30-37: the `curvature` field)
  float4 *h = cache.pin;
  const char *src = static_cast<const char *>(cloud);
  bool uploaded = false;
  if (stride_bytes == 16 && intensity_offset_bytes == 12) {
    // already {x, y, z, w}: copied to pinned memory and sent off a quarter at a time -- the link works on one quarter while
    // the host copies the next (the device arrays are carved further down; the upload target is the blob's head)
    uploaded = true;
  } else {
    for (size_t i = 0; i < n_points; ++i) {
      float v[3], w;
      std::memcpy(v, src + i * stride_bytes, 12);
      std::memcpy(&w, src + i * stride_bytes + intensity_offset_bytes, 4);
      h[i] = make_float4(v[0], v[1], v[2], w);
    }
  }
  const size_t bytes = 5 * np4 + 2 * n_scans * 4 + 4 * n_scans * 4 + n_points * 4 + 2 * n_points + (n_scans + 1) * 4 + np4 + n_points * 4 +
                       n_scans * 4 + 256 + 18 * 16;
  if (bytes > cache.cap) {
    if (cache.p) (void)hipFree(cache.p);
    cache.p = nullptr;
    cache.cap = 0;
    FX_TRY(hipMalloc((void **)&cache.p, bytes + bytes / 4));
    cache.cap = bytes + bytes / 4;
  }
  char *blob = cache.p;
  char *q = blob;
  auto take = [&](size_t b) { char *r = q; q += (b + 15) & ~(size_t)15; return r; };
  float4 *d_pts = (float4 *)take(np4);
  int32_t *st0 = (int32_t *)take(np4), *st1 = (int32_t *)take(np4), *st2 = (int32_t *)take(np4), *st3 = (int32_t *)take(np4);
  int32_t *d_ranges = (int32_t *)take(2 * n_scans * 4), *d_counts = (int32_t *)take(4 * n_scans * 4);
  float *d_curv = (float *)take(n_points * 4);
  int8_t *d_picked = (int8_t *)take(n_points), *d_label = (int8_t *)take(n_points);
  int32_t *d_ring_out = (int32_t *)take((n_scans + 1) * 4);  // centroids per ring (fx_ring_voxel_kernel)
  float4 *d_vox = (float4 *)take(np4);                       // ... and the centroids, ring r's from its first index on
  int32_t *d_cls = (int32_t *)take(n_points * 4), *d_ready = (int32_t *)take(n_scans * 4);  // the classify helpers' (fx_ring_kernel)
  int rc = LSLAM_OK;
  auto fail = [&](int code) { return code; };
#define FX_TRY2(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { lslam::set_error(hipGetErrorString(_e)); return fail(LSLAM_ERR_HIP); } } while (0)
  if (uploaded) {
    const size_t quarter = (n_points + 3) / 4;
    for (size_t b = 0; b < n_points; b += quarter) {
      const size_t cnt = std::min(quarter, n_points - b);
      std::memcpy(h + b, src + b * sizeof(float4), cnt * sizeof(float4));
      FX_TRY2(hipMemcpyAsync(d_pts + b, h + b, cnt * sizeof(float4), hipMemcpyHostToDevice, s));
    }
  } else {
    FX_TRY2(hipMemcpyAsync(d_pts, h, np4, hipMemcpyHostToDevice, s));
  }
  FX_TRY2(hipMemcpyAsync(d_ranges, scan_ranges, 2 * n_scans * 4, hipMemcpyHostToDevice, s));
  if (curvature_out) FX_TRY2(hipMemsetAsync(d_curv, 0, n_points * 4, s));
  if (picked_out) FX_TRY2(hipMemsetAsync(d_picked, 0, n_points, s));
  if (label_out) FX_TRY2(hipMemsetAsync(d_label, L_UNKNOW, n_points, s));
  FxArgs a{};
  a.pts = d_pts;
  a.ranges = d_ranges;
  a.n_scans = (int32_t)n_scans;
  a.nf = prm.n_feature_regions;
  a.cr = prm.curvature_region;
  a.max_sharp = prm.max_corner_sharp;
  a.max_flat = prm.max_surface_flat;
  a.surf_thr = prm.surface_curvature_threshold;
  a.blind_thr = prm.blind_threshold;
  a.c175 = std::cos(175.0 * M_PI / 180.0);  // deg2rad(double), util/math_utils.h:29
  a.c5 = std::cos(5.0 * M_PI / 180.0);
  a.c135 = std::cos(135.0 * M_PI / 180.0);
  a.c45 = std::cos(45.0 * M_PI / 180.0);
  a.st_sharp = st0; a.st_less_sharp = st1; a.st_flat = st2; a.st_less_flat = st3;
  a.counts = d_counts;
  a.curv_out = curvature_out ? d_curv : nullptr;
  a.picked_out = picked_out ? d_picked : nullptr;
  a.label_out = label_out ? d_label : nullptr;
  a.helpers = lslam::env_once().fx_helpers;
  a.cls_g = d_cls;
  a.ready = d_ready;
  if (a.helpers > 0) FX_TRY2(hipMemsetAsync(d_ready, 0, n_scans * 4, s));
  hipLaunchKernelGGL(fx_ring_kernel, dim3((unsigned)n_scans * (unsigned)(1 + a.helpers)), dim3(FX_BLOCK), 0, s, a);
  // the four lists, on the device to the end (see fx_lists_block): nothing waits until everything is in pinned memory
  uint32_t *hdr = reinterpret_cast<uint32_t *>(cache.pout);
  for (int k = 0; k < 8; ++k) hdr[k] = 0u;
  if (dev_out) {  // ... or in the feature set's slices in HBM: only the eight header words come back
    FX_TRY2(lslam::fset_reserve(dev_out, n_points));
    FX_TRY2(hipMemsetAsync(dev_out->buf, 0, 8 * sizeof(uint32_t), s));
  }
  FxOutArgs oa{};
  oa.pts = d_pts;
  oa.ranges = d_ranges;
  oa.counts = d_counts;
  oa.stage[0] = st0; oa.stage[1] = st1; oa.stage[2] = st2; oa.stage[3] = st3;
  oa.n_scans = (int32_t)n_scans;
  oa.cap = (int32_t)n_points;
  oa.inv_leaf = 1.0f / prm.less_flat_filter_size;
  oa.vox_stage = d_vox;
  oa.ring_out = d_ring_out;
  oa.host = dev_out ? dev_out->buf : cache.pout;
  oa.hdr = dev_out ? reinterpret_cast<uint32_t *>(dev_out->buf) : hdr;
  if (dev_out) oa.cap = (int32_t)dev_out->cap;
  hipLaunchKernelGGL(fx_ring_voxel_kernel, dim3((unsigned)n_scans + 3u), dim3(FX_BLOCK), 0, s, oa);
  hipLaunchKernelGGL(fx_lessflat_out_kernel, dim3((unsigned)n_scans), dim3(FX_BLOCK), 0, s, oa);
  FX_TRY2(hipGetLastError());
  if (curvature_out) FX_TRY2(hipMemcpyAsync(curvature_out, d_curv, n_points * 4, hipMemcpyDeviceToHost, s));
  if (picked_out) FX_TRY2(hipMemcpyAsync(picked_out, d_picked, n_points, hipMemcpyDeviceToHost, s));
  if (label_out) FX_TRY2(hipMemcpyAsync(label_out, d_label, n_points, hipMemcpyDeviceToHost, s));
  if (dev_out) FX_TRY2(hipMemcpyAsync(hdr, dev_out->buf, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  FX_TRY2(hipStreamSynchronize(s));
  if (hdr[4]) {
    lslam::set_error(hdr[4] == 1u ? "voxel index outside its range (non-finite point?)" : "a feature list overflowed its staging slice");
    return fail(LSLAM_ERR_INVALID);
  }
  float *outs[4] = {sharp, less_sharp, flat, less_flat};
  for (int k = 0; k < 4; ++k) {
    counts[k] = (size_t)hdr[k];
    if (dev_out) dev_out->counts[k] = counts[k];
    if (!dev_out && outs[k] && hdr[k]) std::memcpy(outs[k], cache.pout + 16 + (size_t)k * n_points, (size_t)hdr[k] * sizeof(float4));
  }
  return rc;
}


int lslam_multiscan_register(lslam_ctx *ctx, const void *cloud, size_t n_points, size_t stride_bytes, float lower_deg,
                             float upper_deg, int32_t n_rings, float scan_period, float *out_xyzc, size_t cap,
                             size_t *n_out, int32_t *ranges_out) {
  if (!ctx || !lslam::ctx_alive(ctx) || !n_out || !ranges_out || (n_points && !cloud) || stride_bytes < 12 ||
      (stride_bytes & 3) || n_rings <= 0 || n_rings > 4096 || !(upper_deg > lower_deg) || n_points > 0x3FFFFFFFu) {
    lslam::set_error("bad multi-scan registration arguments");
    return LSLAM_ERR_INVALID;
  }
  *n_out = 0;
  for (int r = 0; r < n_rings; ++r) { ranges_out[2 * r] = 0; ranges_out[2 * r + 1] = 0; }
  if (n_points == 0) return LSLAM_OK;
  FX_TRY(hipSetDevice(lslam::ctx_device(ctx)));
  hipStream_t s = (hipStream_t)lslam_stream(ctx);
  struct Cache { char *p = nullptr; size_t cap = 0; float4 *pin = nullptr; size_t pin_cap = 0; };
  static std::map<hipStream_t, Cache> caches;  // device scratch + pinned input staging kept per stream (= context) between sweeps
  static std::mutex mu;
  Cache *cache_p;
  {
    std::lock_guard<std::mutex> lk(mu);
    cache_p = &caches[s];
  }
  Cache &cache = *cache_p;
  const size_t np4 = n_points * sizeof(float4);
  if (n_points > cache.pin_cap) {
    if (cache.pin) (void)hipHostFree(cache.pin);
    cache.pin = nullptr;
    cache.pin_cap = 0;
    FX_TRY(hipHostMalloc((void **)&cache.pin, (n_points + n_points / 4) * sizeof(float4), hipHostMallocDefault));
    cache.pin_cap = n_points + n_points / 4;
  }
  float4 *h = cache.pin;
  const char *src = static_cast<const char *>(cloud);
  for (size_t i = 0; i < n_points; ++i) {
    float v[3];
    std::memcpy(v, src + i * stride_bytes, 12);
    h[i] = make_float4(v[0], v[1], v[2], 0.0f);
  }
  // sweep start / end orientation (:101-109), on the host with the C library the reference uses
  float start_ori = -std::atan2(h[0].y, h[0].x);
  float end_ori = -std::atan2(h[n_points - 1].y, h[n_points - 1].x) + 2 * float(M_PI);
  if (end_ori - start_ori > 3 * M_PI) end_ori -= 2 * M_PI;
  else if (end_ori - start_ori < M_PI) end_ori += 2 * M_PI;
  const size_t bytes = 3 * np4 + 4 * n_points * 4 + 64;
  if (bytes > cache.cap) {
    if (cache.p) (void)hipFree(cache.p);
    cache.p = nullptr;
    cache.cap = 0;
    FX_TRY(hipMalloc((void **)&cache.p, bytes + bytes / 4));
    cache.cap = bytes + bytes / 4;
  }
  char *blob = cache.p;
  float4 *d_in = (float4 *)blob, *d_tmp = d_in + n_points, *d_out = d_tmp + n_points;
  int32_t *d_ring = (int32_t *)(d_out + n_points), *d_seg = d_ring + n_points;
  float *d_ori = (float *)(d_seg + n_points);
  int32_t *d_first = (int32_t *)(d_ori + n_points);
  auto fail = [&](int code) { return code; };
  const int32_t big = INT32_MAX;
  FX_TRY2(hipMemcpyAsync(d_in, h, np4, hipMemcpyHostToDevice, s));
  FX_TRY2(hipMemcpyAsync(d_first, &big, 4, hipMemcpyHostToDevice, s));
  MsArgs a{};
  a.in = d_in;
  a.n = (int)n_points;
  a.n_rings = n_rings;
  a.lower = lower_deg;
  a.factor = (n_rings - 1) / (upper_deg - lower_deg);  // MultiScanRegistration.h:63
  a.scan_period = scan_period;
  a.start_ori = start_ori;
  a.end_ori = end_ori;
  a.out = d_tmp;
  a.ring = d_ring;
  a.ori_raw = d_ori;
  a.first_half = d_first;
  const dim3 grd((unsigned)((n_points + 255) / 256)), blk(256);
  hipLaunchKernelGGL(ms_prep_kernel, grd, blk, 0, s, a);
  hipLaunchKernelGGL(ms_final_kernel, grd, blk, 0, s, a);
  size_t m = 0;  // per-ring clouds in arrival order (:178-190): a stable grouping by ring
  int rc = lslam::voxel_filter_segments(s, d_tmp, d_ring, n_points, n_rings, 1.0f, d_out, d_seg, &m, false);
  if (rc) return fail(rc);
  if (m > cap && out_xyzc) {
    lslam::set_error("registration output buffer too small");
    return fail(LSLAM_ERR_INVALID);
  }
  std::vector<int32_t> seg(m);
  if (m) {
    if (out_xyzc) FX_TRY2(hipMemcpyAsync(out_xyzc, d_out, m * sizeof(float4), hipMemcpyDeviceToHost, s));
    FX_TRY2(hipMemcpyAsync(seg.data(), d_seg, m * 4, hipMemcpyDeviceToHost, s));
    FX_TRY2(hipStreamSynchronize(s));
  }
  std::vector<size_t> count((size_t)n_rings, 0);
  for (size_t i = 0; i < m; ++i) count[(size_t)seg[i]]++;
  size_t total = 0;
  for (int r = 0; r < n_rings; ++r) {  // IndexRange(first, last), :184-189
    ranges_out[2 * r] = (int32_t)total;
    total += count[(size_t)r];
    ranges_out[2 * r + 1] = total > 0 ? (int32_t)total - 1 : 0;
  }
  *n_out = m;
  return LSLAM_OK;
}

}  // extern "C"
