// lslam_icp.hip -- point-to-point ICP on the device: the coarse alignment of a loop-closure candidate,
// LoopDetector::corseMatching (pose_graph/loop_detector.hpp:232-255), which the reference delegates to
// pcl::IterativeClosestPoint<PointXYZI, PointXYZI> with every setting at its default.
//
// PARITY UNPINNED: PCL is neither under /root/reference nor installed.  Restated here (and, independently,
// in oracle/icp_oracle.py on scipy's cKDTree + numpy's SVD) is PCL's published algorithm with its defaults:
//   * correspondences: the nearest target point of every transformed source point (k = 1, kd-tree), kept
//     when its distance is within max_correspondence_distance (default sqrt(DBL_MAX): all of them);
//   * TransformationEstimationSVD (Umeyama without scale): centroids, 3x3 cross-covariance, SVD,
//     R = V diag(1, 1, det(V U^T)) U^T, t = c_target - R c_source; the increment is composed on the left;
//   * DefaultConvergenceCriteria: at most 10 iterations (reaching them counts as converged), or the relative
//     change of the mean squared correspondence distance below 1e-5, or its absolute value below 1e-12, or
//     an increment with translation^2 <= transformation_epsilon and cos(angle) >= 1 - transformation_epsilon
//     (both impossible at the default epsilon 0); fewer than 3 correspondences: not converged;
//   * getFitnessScore(): mean squared nearest-neighbour distance of the aligned source.
// The nearest neighbour is the first of the exact 5-NN search of lslam_device.hpp on the target's kd-tree
// (device build); the 16 sums of an iteration are reduced in fp64; the 3x3 SVD runs on the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/lslam_c.h"
#include "lslam_internal.hpp"

namespace lslam {

namespace {

constexpr int ICP_BLOCK = 128;
constexpr int ICP_SUMS = 18;  // n, sum d2, sum s (3), sum t (3), sum s t^T (9), pad

struct IcpArgs {
  TreeView T;
  const float4 *src;
  int32_t n_src;
  float R[9], t[3];   // current transform source -> target
  float max_d2;       // correspondence gate (squared), FLT_MAX: none
  uint32_t *stack_ovf;
  double *partials;   // [blocks][ICP_SUMS]
};

template <bool OVF>
__global__ __launch_bounds__(ICP_BLOCK) void icp_corr_kernel(IcpArgs a) {
  __shared__ uint32_t stack_lds[2 * KD_STACK_LDS * ICP_BLOCK];
  __shared__ double red[ICP_BLOCK / 64][ICP_SUMS];
  const int i = blockIdx.x * ICP_BLOCK + threadIdx.x;
  double v[ICP_SUMS];
#pragma unroll
  for (int k = 0; k < ICP_SUMS; ++k) v[k] = 0.0;
  if (i < a.n_src) {
    const float4 s = a.src[i];
    // pcl::transformPointCloud: p' = R p + t in fp32
    const float x = ((a.R[0] * s.x + a.R[1] * s.y) + a.R[2] * s.z) + a.t[0];
    const float y = ((a.R[3] * s.x + a.R[4] * s.y) + a.R[5] * s.z) + a.t[1];
    const float z = ((a.R[6] * s.x + a.R[7] * s.y) + a.R[8] * s.z) + a.t[2];
    float d[5];
    int p[5];
    KdStack<ICP_BLOCK, OVF, KD_STACK_LDS> stk;
    stk.lds = (lds_u32 *)(stack_lds + threadIdx.x);
    stk.ovf = OVF ? a.stack_ovf + ((size_t)blockIdx.x * ICP_BLOCK + threadIdx.x) : nullptr;
    stk.ovf_stride = (size_t)gridDim.x * ICP_BLOCK;
#ifdef LSLAM_TRAVERSAL_STATS
    TravStats ts_unused = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    knn5_search<ICP_BLOCK, OVF, KD_STACK_LDS>(a.T, x, y, z, d, p, stk, ts_unused);
#else
    knn5_search<ICP_BLOCK, OVF, KD_STACK_LDS>(a.T, x, y, z, d, p, stk);
#endif
    if (p[0] >= 0 && d[0] <= a.max_d2) {
      const float4 q = a.T.pts[p[0]];
      v[0] = 1.0;
      v[1] = (double)d[0];
      v[2] = x; v[3] = y; v[4] = z;
      v[5] = q.x; v[6] = q.y; v[7] = q.z;
      v[8] = (double)x * q.x;  v[9] = (double)x * q.y;  v[10] = (double)x * q.z;
      v[11] = (double)y * q.x; v[12] = (double)y * q.y; v[13] = (double)y * q.z;
      v[14] = (double)z * q.x; v[15] = (double)z * q.y; v[16] = (double)z * q.z;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < ICP_SUMS; ++k) {
    double s = v[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < ICP_SUMS) {
    double s = 0.0;
    for (int w = 0; w < ICP_BLOCK / 64; ++w) s += red[w][threadIdx.x];
    a.partials[(size_t)blockIdx.x * ICP_SUMS + threadIdx.x] = s;
  }
}

// one-sided Jacobi SVD of a 3x3 matrix (fp64): A = U diag(s) V^T
void svd3(const double A[9], double U[9], double S[3], double V[9]) {
  double B[9];
  std::memcpy(B, A, sizeof(B));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += B[r * 3 + p] * B[r * 3 + p];
          beta += B[r * 3 + q] * B[r * 3 + q];
          gamma += B[r * 3 + p] * B[r * 3 + q];
        }
        off = std::max(off, std::fabs(gamma) / std::sqrt(std::max(alpha * beta, 1e-300)));
        if (std::fabs(gamma) < 1e-300) continue;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          const double bp = B[r * 3 + p], bq = B[r * 3 + q];
          B[r * 3 + p] = c * bp - s * bq;
          B[r * 3 + q] = s * bp + c * bq;
          const double vp = V[r * 3 + p], vq = V[r * 3 + q];
          V[r * 3 + p] = c * vp - s * vq;
          V[r * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int j = 0; j < 3; ++j) {
    double n = 0;
    for (int r = 0; r < 3; ++r) n += B[r * 3 + j] * B[r * 3 + j];
    S[j] = std::sqrt(n);
  }
  // order the singular values descending (the determinant fix goes to the smallest one)
  int idx[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (S[idx[b]] > S[idx[a]]) std::swap(idx[a], idx[b]);
  double Bs[9], Vs[9], Ss[3];
  for (int j = 0; j < 3; ++j) {
    Ss[j] = S[idx[j]];
    for (int r = 0; r < 3; ++r) { Bs[r * 3 + j] = B[r * 3 + idx[j]]; Vs[r * 3 + j] = V[r * 3 + idx[j]]; }
  }
  std::memcpy(S, Ss, sizeof(Ss));
  std::memcpy(V, Vs, sizeof(Vs));
  for (int j = 0; j < 3; ++j)
    for (int r = 0; r < 3; ++r) U[r * 3 + j] = S[j] > 1e-300 ? Bs[r * 3 + j] / S[j] : 0.0;
  // a rank-deficient column of U: complete the basis by a cross product
  if (S[2] <= 1e-300 * (S[0] + 1e-300) || S[2] == 0.0) {
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

}  // namespace
}  // namespace lslam

using namespace lslam;

extern "C" int lslam_icp_align(lslam_ctx *ctx, const void *target, size_t n_target, const void *source, size_t n_source,
                               size_t stride_bytes, float T_io[16], int32_t max_iterations, double transformation_epsilon,
                               double max_correspondence_distance, double *fitness_out, int32_t *converged_out,
                               int32_t *iterations_out) {
  if (converged_out) *converged_out = 0;
  if (iterations_out) *iterations_out = 0;
  if (fitness_out) *fitness_out = 0.0;
  if (!ctx || !T_io || stride_bytes < 12 || (stride_bytes & 3) || (n_target && !target) || (n_source && !source)) {
    set_error("bad ICP arguments");
    return LSLAM_ERR_INVALID;
  }
  if (n_target == 0) return LSLAM_OK;  // loop_detector.hpp:233-235: empty reference -> not matched
  // the target's kd-tree takes the surf slot of the context's map (the map belongs to this call afterwards)
  int rc = lslam_map_set(ctx, nullptr, 0, target, n_target, stride_bytes);
  if (rc) return rc;
  ctx_invalidate_map(ctx);
  hipStream_t s = ctx_stream(ctx);
  const TreeView tv = ctx_tree_view(ctx, 1);
  float4 *d_src = nullptr;
  double *d_part = nullptr;
  const int blocks = (int)((n_source + ICP_BLOCK - 1) / ICP_BLOCK);
  rc = ctx_scratch(ctx, n_source, (size_t)std::max(blocks, 1) * ICP_SUMS, &d_src, &d_part);
  if (rc) return rc;
  {
    std::vector<float4> h(n_source);
    const char *p = static_cast<const char *>(source);
    for (size_t i = 0; i < n_source; ++i) {
      float v[3];
      std::memcpy(v, p + i * stride_bytes, 12);
      h[i] = make_float4(v[0], v[1], v[2], 0.f);
    }
    if (n_source && hipMemcpyAsync(d_src, h.data(), n_source * sizeof(float4), hipMemcpyHostToDevice, s) != hipSuccess) return LSLAM_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return LSLAM_ERR_HIP;
  }
  uint32_t *ovf = nullptr;
  rc = ctx_stack_ovf_if_deep(ctx, (size_t)std::max(blocks, 1) * ICP_BLOCK, &ovf);
  if (rc) return rc;
  double Tm[16];
  for (int i = 0; i < 16; ++i) Tm[i] = (double)T_io[i];
  const float max_d2 = max_correspondence_distance > 0.0 && max_correspondence_distance < 1e18
                           ? (float)(max_correspondence_distance * max_correspondence_distance) : FLT_MAX;
  std::vector<double> part((size_t)std::max(blocks, 1) * ICP_SUMS);
  auto correspondences = [&](double sums[ICP_SUMS]) -> int {
    IcpArgs a{};
    a.T = tv;
    a.src = d_src;
    a.n_src = (int32_t)n_source;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) a.R[r * 3 + c] = (float)Tm[r * 4 + c];
      a.t[r] = (float)Tm[r * 4 + 3];
    }
    a.max_d2 = max_d2;
    a.stack_ovf = ovf;
    a.partials = d_part;
    for (int k = 0; k < ICP_SUMS; ++k) sums[k] = 0.0;
    if (blocks == 0) return LSLAM_OK;
    if (ovf) hipLaunchKernelGGL(icp_corr_kernel<true>, dim3(blocks), dim3(ICP_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(icp_corr_kernel<false>, dim3(blocks), dim3(ICP_BLOCK), 0, s, a);
    if (hipGetLastError() != hipSuccess) return LSLAM_ERR_HIP;
    if (hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return LSLAM_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return LSLAM_ERR_HIP;
    for (int b = 0; b < blocks; ++b)
      for (int k = 0; k < ICP_SUMS; ++k) sums[k] += part[(size_t)b * ICP_SUMS + k];
    return LSLAM_OK;
  };
  const int max_it = max_iterations > 0 ? max_iterations : 10;
  bool converged = false;
  double prev_mse = 1.7976931348623157e308;
  int it = 0;
  for (;;) {
    double S[ICP_SUMS];
    rc = correspondences(S);
    if (rc) return rc;
    const double n = S[0];
    if (n < 3.0) { converged = false; break; }  // min_number_correspondences_ = 3
    const double cs[3] = {S[2] / n, S[3] / n, S[4] / n}, ct[3] = {S[5] / n, S[6] / n, S[7] / n};
    double H[9];  // sum (s - cs)(t - ct)^T
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) H[r * 3 + c] = S[8 + r * 3 + c] - n * cs[r] * ct[c];
    double U[9], W[3], V[9];
    svd3(H, U, W, V);
    // R = V diag(1,1,d) U^T
    double VUt[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) VUt[r * 3 + c] = V[r * 3] * U[c * 3] + V[r * 3 + 1] * U[c * 3 + 1] + V[r * 3 + 2] * U[c * 3 + 2];
    const double dsign = det3(VUt) < 0 ? -1.0 : 1.0;
    double R[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = V[r * 3] * U[c * 3] + V[r * 3 + 1] * U[c * 3 + 1] + dsign * V[r * 3 + 2] * U[c * 3 + 2];
    double dt[3];
    for (int r = 0; r < 3; ++r) dt[r] = ct[r] - (R[r * 3] * cs[0] + R[r * 3 + 1] * cs[1] + R[r * 3 + 2] * cs[2]);
    // T <- [R | dt] * T
    double Tn[16];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 4; ++c) Tn[r * 4 + c] = R[r * 3] * Tm[c] + R[r * 3 + 1] * Tm[4 + c] + R[r * 3 + 2] * Tm[8 + c];
      Tn[r * 4 + 3] += dt[r];
    }
    Tn[12] = Tn[13] = 0.0; Tn[14] = 0.0; Tn[15] = 1.0;
    std::memcpy(Tm, Tn, sizeof(Tn));
    ++it;
    // DefaultConvergenceCriteria
    const double mse = S[1] / n;
    const double cos_angle = 0.5 * (R[0] + R[4] + R[8] - 1.0);
    const double trans2 = dt[0] * dt[0] + dt[1] * dt[1] + dt[2] * dt[2];
    if (it >= max_it) { converged = true; break; }
    if (cos_angle >= 1.0 - transformation_epsilon && trans2 <= transformation_epsilon) { converged = true; break; }
    if (std::fabs(mse - prev_mse) < 1e-12) { converged = true; break; }
    if (std::fabs(mse - prev_mse) / prev_mse < 1e-5) { converged = true; break; }
    prev_mse = mse;
  }
  for (int i = 0; i < 16; ++i) T_io[i] = (float)Tm[i];
  if (fitness_out || true) {  // getFitnessScore(): mean squared NN distance of the aligned source
    double S[ICP_SUMS];
    rc = correspondences(S);
    if (rc) return rc;
    if (fitness_out) *fitness_out = S[0] > 0 ? S[1] / S[0] : 1.7976931348623157e308;
  }
  if (converged_out) *converged_out = converged ? 1 : 0;
  if (iterations_out) *iterations_out = it;
  return LSLAM_OK;
}
