// lslam_stereo.hip -- the stereo reprojection rows of the joint LiDAR + stereo system
// (BASELINE configs[4]).
//
// The reference has no code for a visual term (README.md:51-71 announces it): the arithmetic is
// the published one of ORB-SLAM2's pose-only stereo edge, stated in include/lslam_c.h and restated
// in oracle/lslam_oracle.c (stereo_rows).  PARITY UNPINNED.
//
// One lane per observation.  The lane forms its (up to) three scaled rows [J | b] with respect to
// the six Twist parameters; the 64 x 3 rows of a wavefront are contracted to the 7 x 7 Gram matrix
// by v_mfma_f32_16x16x4_f32 (48 instructions, exact fp32 products like the LiDAR rows of
// sweep_kernel), the counters go through DPP/shuffle sums, and the block writes one partial-sum
// record in the layout of the LiDAR sweep (lslam_internal.hpp COL_*), so that solve_kernel adds
// both kinds of block into the same 6x6 system in a fixed order.
//
// Bytes: 32 per observation ({X, inv_sigma2}, {uL, v, uR, -}) read once, coalesced; with 1e3-1e5
// observations per frame the kernel is launch-latency bound (a few microseconds).
#include <hip/hip_runtime.h>

#include "lslam_internal.hpp"

namespace lslam {
namespace {

constexpr int ST_BLOCK = 256;
constexpr int ST_WAVES = ST_BLOCK / 64;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float st_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(ST_BLOCK) void stereo_kernel(StereoArgs a) {
  const GNState *st = a.state;
  if (st->done) return;
  __shared__ float jr[ST_WAVES][3][8][64];  // [wave][component][column][lane]
  __shared__ float red[ST_WAVES][NCOL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * ST_BLOCK + tid;
  float rows[3][7];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 7; ++c) rows[r][c] = 0.0f;
  float n_rows = 0.0f, used = 0.0f;
  if (i < a.n) {
    const float4 L = a.landmarks[i];  // X, Y, Z, inv_sigma2
    const float4 ob = a.obs[i];       // uL, v, uR, -
    const StereoCam &c = a.cam;
    float R[9], sc[6];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = st->R[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) sc[k] = st->sc[k];
    const float srx = sc[0], crx = sc[1], sry = sc[2], cry = sc[3], srz = sc[4], crz = sc[5];
    // derivatives of R = Rz Ry Rx by rx, ry, rz
    const float dRx[9] = {0.0f, crz * sry * crx + srz * srx, srz * crx - crz * sry * srx,
                          0.0f, srz * sry * crx - crz * srx, -(srz * sry * srx) - crz * crx,
                          0.0f, cry * crx, -(cry * srx)};
    const float dRy[9] = {-(crz * sry), crz * cry * srx, crz * cry * crx,
                          -(srz * sry), srz * cry * srx, srz * cry * crx,
                          -cry, -(sry * srx), -(sry * crx)};
    const float dRz[9] = {-(srz * cry), -(srz * sry * srx) - crz * crx, crz * srx - srz * sry * crx,
                          crz * cry, crz * sry * srx - srz * crx, crz * sry * crx + srz * srx,
                          0.0f, 0.0f, 0.0f};
    const float d0 = L.x - st->t[0], d1 = L.y - st->t[1], d2 = L.z - st->t[2];
    auto mTv = [&](const float *M, float out[3]) {  // M^T d
      out[0] = M[0] * d0 + M[3] * d1 + M[6] * d2;
      out[1] = M[1] * d0 + M[4] * d1 + M[7] * d2;
      out[2] = M[2] * d0 + M[5] * d1 + M[8] * d2;
    };
    float p[3], G[6][3];
    mTv(R, p);
    mTv(dRx, G[0]);
    mTv(dRy, G[1]);
    mTv(dRz, G[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      G[3 + k][0] = -R[3 * k + 0];
      G[3 + k][1] = -R[3 * k + 1];
      G[3 + k][2] = -R[3 * k + 2];
    }
    const float *T = c.T_cl;
    const float x = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
    const float y = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
    const float z = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
    if (z > c.min_depth) {
      const bool mono = ob.z < 0.0f;
      const float iz = __fdiv_rn(1.0f, z);
      const float uL = c.fx * x * iz + c.cx, v = c.fy * y * iz + c.cy, uR = uL - c.bf * iz;
      const float e0 = uL - ob.x, e1 = v - ob.y, e2 = mono ? 0.0f : uR - ob.z;
      const float chi2 = (e0 * e0 + e1 * e1 + e2 * e2) * L.w;
      const float delta = mono ? c.huber_mono : c.huber_stereo;
      if (!(c.gate_outliers && chi2 > delta * delta)) {
        const float rchi = __fsqrt_rn(chi2);
        const float wh = rchi <= delta ? 1.0f : __fdiv_rn(delta, rchi);
        const float s = __fsqrt_rn(c.weight * L.w * wh);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const float gx = T[0] * G[k][0] + T[1] * G[k][1] + T[2] * G[k][2];
          const float gy = T[4] * G[k][0] + T[5] * G[k][1] + T[6] * G[k][2];
          const float gz = T[8] * G[k][0] + T[9] * G[k][1] + T[10] * G[k][2];
          const float ju = c.fx * iz * (gx - x * iz * gz);
          const float jv = c.fy * iz * (gy - y * iz * gz);
          rows[0][k] = s * ju;
          rows[1][k] = s * jv;
          rows[2][k] = mono ? 0.0f : s * (ju + c.bf * iz * iz * gz);
        }
        rows[0][6] = -(s * e0);
        rows[1][6] = -(s * e1);
        rows[2][6] = mono ? 0.0f : -(s * e2);
        n_rows = mono ? 2.0f : 3.0f;
        used = 1.0f;
      }
    }
  }
  // ---- 7x7 Gram matrix of the wave's 192 rows on the matrix core ------------------------------
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 7; ++c) jr[wave][r][c][lane] = rows[r][c];
    jr[wave][r][7][lane] = 0.0f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  const int i16 = lane & 15, k4 = lane >> 4;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float op = (i16 < 8) ? jr[wave][r][i16][4 * s + k4] : 0.0f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(op, op, acc, 0, 0, 0);
    }
  }
  if (lane < NCOL) red[wave][lane] = 0.0f;
  __builtin_amdgcn_wave_barrier();
  // C/D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int rr = k4 * 4 + r4, cc = i16;
    if (rr < 6 && cc < 7 && cc >= rr) {
      const int col = cc == 6 ? COL_ATB + rr : COL_ATA + (rr * 6 - (rr * (rr - 1)) / 2) + (cc - rr);
      red[wave][col] = acc[r4];
    }
  }
  const float s_rows = st_wave_sum(n_rows), s_used = st_wave_sum(used);
  if (lane == 0) {
    red[wave][COL_ROWS] = s_rows;
    red[wave][COL_STEREO] = s_used;
  }
  __syncthreads();
  if (tid < NCOL) {
    float s = red[0][tid];
#pragma unroll
    for (int w = 1; w < ST_WAVES; ++w) s += red[w][tid];
    a.partials[(size_t)blockIdx.x * NCOL + tid] = s;
  }
}

}  // namespace

int stereo_blocks(int n) { return (n + ST_BLOCK - 1) / ST_BLOCK; }

hipError_t launch_stereo(const StereoArgs &a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(stereo_kernel, dim3(stereo_blocks(a.n)), dim3(ST_BLOCK), 0, s, a);
  return hipGetLastError();
}

}  // namespace lslam
