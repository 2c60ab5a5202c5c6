// lslam_odom_dev.hpp -- device functions of variant B (LaserOdometry::scanMatch, odometry/LaserOdometry.cpp:328-647) shared by
// the launch-per-iteration kernels (lslam_kernels.hip) and the device-resident odometry node (lslam_odom.hip).
#pragma once

#include "lslam_device.hpp"

namespace lslam {

LSLAM_DEV float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

LSLAM_DEV float sq_diff3(const float4 &a, const float (&b)[3]) {  // math_utils.h:47-54
  const float dx = a.x - b[0], dy = a.y - b[1], dz = a.z - b[2];
  return dx * dx + dy * dy + dz * dz;
}

// feature_utils.h:42-61
LSLAM_DEV bool odom_corner_coeff(const float4 &A4, const float4 &B4, const float (&X)[3], int iter,
                                 float (&coeff)[4]) {
  const float A[3] = {A4.x, A4.y, A4.z}, B[3] = {B4.x, B4.y, B4.z};
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
  float weight = 1.0f;
  if (iter >= 5) weight = (float)(1 - 1.8 * (double)fabsf(distance));
  coeff[0] = (cr[0] / den) * weight;
  coeff[1] = (cr[1] / den) * weight;
  coeff[2] = (cr[2] / den) * weight;
  coeff[3] = distance * weight;
  return (double)weight > 0.1 && distance != 0.0f;
}

// feature_utils.h:28-40 + :77-95
LSLAM_DEV bool odom_surf_coeff(const float4 &A4, const float4 &B4, const float4 &C4, const float (&X)[3],
                               int iter, float (&coeff)[4]) {
  const float BA[3] = {B4.x - A4.x, B4.y - A4.y, B4.z - A4.z};
  const float CA[3] = {C4.x - A4.x, C4.y - A4.y, C4.z - A4.z};
  float nrm[3];
  cross3(BA, CA, nrm);
  const float z = (nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2];
  if (z > 0.0f) {
    const float l = sqrtf(z);
    nrm[0] /= l; nrm[1] /= l; nrm[2] /= l;
  }
  const float XA[3] = {X[0] - A4.x, X[1] - A4.y, X[2] - A4.z};
  float distance = (XA[0] * nrm[0] + XA[1] * nrm[1]) + XA[2] * nrm[2];
  const float AX[3] = {A4.x - X[0], A4.y - X[1], A4.z - X[2]};
  const float cosv = distance / norm3(nrm) / norm3(AX);
  if (cosv < 0) { nrm[0] *= -1.0f; nrm[1] *= -1.0f; nrm[2] *= -1.0f; }
  distance = fabsf(distance);
  float weight = 1.0f;
  if (iter >= 5) weight = (float)(1 - 1.8 * (double)fabsf(distance) / sqrt((double)norm3(X)));
  coeff[0] = weight * nrm[0];
  coeff[1] = weight * nrm[1];
  coeff[2] = weight * nrm[2];
  coeff[3] = weight * distance;
  return (double)weight > 0.1 && distance != 0.0f;
}

struct DevSinCosF {
  __device__ void operator()(float a, float &s, float &c) const {
    s = (float)sin((double)a);
    c = (float)cos((double)a);
  }
};


// The ring-window searches of LaserOdometry::scanMatch (:366-403 sharp, :430-477 flat) for one query, by one WAVEFRONT (all 64
// lanes call it with the same arguments but `lane`): the reference walks forwards (to the QUERY count nq, quirk Q5) and
// backwards from the nearest neighbour i1 until the ring id leaves scan +- 2.5, keeping the first strictly smaller squared
// distance per category.  The wavefront takes the window 64 candidates at a time; "first minimum in scan order" = minimum
// distance, lowest position on ties, compared strictly with the running minimum -- the same choice.
LSLAM_DEV void odom_window_walk(const float4 *org, const int n_org, const int nq, const bool is_flat, const int i1,
                                const float (&sel)[3], const int lane, int &i2, int &i3) {
  i2 = -1;
  i3 = -1;
  const int scan = (int)org[i1].w;
  float m2 = 25.0f, m3 = 25.0f;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int dir = 0; dir < 2; ++dir) {
    const int limit = dir == 0 ? min(nq, n_org) : 0;  // forward: j < limit; backward: j >= 0
    for (int base = 0;; base += 64) {
      const int j = dir == 0 ? i1 + 1 + base + lane : i1 - 1 - base - lane;
      const bool valid = dir == 0 ? j < limit : j >= 0;
      if (!__any(valid)) break;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) o = org[j];
      const int ring = (int)o.w;
      const bool brk = valid && (dir == 0 ? (double)ring > scan + 2.5 : (double)ring < scan - 2.5);
      const unsigned long long mb = __ballot(brk);
      const bool use = valid && (mb == 0 || (below & mb) == 0) && !brk;  // candidates before the first break
      const float dd = use ? sq_diff3(o, sel) : 3.0e38f;
      bool c2, c3;
      if (!is_flat) {
        c2 = use && (dir == 0 ? ring > scan : ring < scan);
        c3 = false;
      } else {
        c2 = use && (dir == 0 ? ring <= scan : ring >= scan);
        c3 = use && !c2;
      }
      // category 2
      float dmin = c2 ? dd : 3.0e38f;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, off, 64));
      if (dmin < m2) {
        const unsigned long long hit = __ballot(c2 && dd == dmin);
        const int l0 = __ffsll((long long)hit) - 1;
        m2 = dmin;
        i2 = dir == 0 ? i1 + 1 + base + l0 : i1 - 1 - base - l0;
      }
      if (is_flat) {
        float dmin3 = c3 ? dd : 3.0e38f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmin3 = fminf(dmin3, __shfl_xor(dmin3, off, 64));
        if (dmin3 < m3) {
          const unsigned long long hit = __ballot(c3 && dd == dmin3);
          const int l0 = __ffsll((long long)hit) - 1;
          m3 = dmin3;
          i3 = dir == 0 ? i1 + 1 + base + l0 : i1 - 1 - base - l0;
        }
      }
      if (mb != 0) break;
    }
  }
}

}  // namespace lslam
