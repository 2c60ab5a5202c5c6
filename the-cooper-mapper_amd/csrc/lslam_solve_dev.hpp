// lslam_solve_dev.hpp -- the solve half of a Gauss-Newton iteration as device functions (ScanMatch.cpp:141-145, 206-260;
// LaserOdometry.cpp:501-503, 580-646): the deterministic cross-block reduction, the 6x6 solve, the degeneracy projection, the
// pose update and the convergence test.  Shared by the solve kernel and the persistent loops of lslam_kernels.hip and by the
// odometry node's persistent loop (lslam_odom.hip): one copy of the arithmetic.
#pragma once

#include "lslam_internal.hpp"

namespace lslam {

// ---------------------------------------------------------------------------
// 6x6 symmetric eigen-decomposition and inverse for the degeneracy test (first
// iteration only; single thread, plain arrays).  Eigen 3.3
// SelfAdjointEigenSolver<Matrix<float,6,6>> generic path and
// Matrix::inverse() (PartialPivLU).
// ---------------------------------------------------------------------------
__device__ static void make_householder_dyn(float *c0, float *tail, int tail_len, int stride,
                                            float *tau, float *beta) {
  float tailSqNorm = 0.0f;
  for (int i = 0; i < tail_len; ++i) tailSqNorm += tail[i * stride] * tail[i * stride];
  if (tailSqNorm <= FLT_MIN) {
    *tau = 0.0f;
    *beta = *c0;
    for (int i = 0; i < tail_len; ++i) tail[i * stride] = 0.0f;
  } else {
    float b = sqrtf((*c0) * (*c0) + tailSqNorm);
    if (*c0 >= 0.0f) b = -b;
    const float denom = *c0 - b;
    for (int i = 0; i < tail_len; ++i) tail[i * stride] = tail[i * stride] / denom;
    *tau = (b - *c0) / b;
    *beta = b;
  }
}

__device__ static void eig_sym6_dyn(const float *A, float *evals, float *V) {
  constexpr int N = 6;
  float m[N * N];
  float scale = 0.0f;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      m[r * N + c] = (c <= r) ? A[r * N + c] : 0.0f;
      scale = fmaxf(scale, fabsf(m[r * N + c]));
    }
  if (scale == 0.0f) scale = 1.0f;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c <= r; ++c) m[r * N + c] /= scale;
  float hCoeffs[N - 1];
  for (int i = 0; i < N - 1; ++i) {
    const int rem = N - i - 1;
    float h, beta;
    make_householder_dyn(&m[(i + 1) * N + i], rem > 1 ? &m[(i + 2) * N + i] : &m[(i + 1) * N + i], rem - 1, N, &h, &beta);  // (empty tail for i = N - 2)
    m[(i + 1) * N + i] = 1.0f;
    float v[N], p[N];
    for (int aa = 0; aa < rem; ++aa) v[aa] = m[(i + 1 + aa) * N + i];
    for (int aa = 0; aa < rem; ++aa) {
      float acc = 0.0f;
      for (int bb = 0; bb < rem; ++bb) {
        const int r = i + 1 + (aa > bb ? aa : bb), c = i + 1 + (aa > bb ? bb : aa);
        acc += m[r * N + c] * (h * v[bb]);
      }
      p[aa] = acc;
    }
    float dot = 0.0f;
    for (int aa = 0; aa < rem; ++aa) dot += p[aa] * v[aa];
    const float alpha = h * -0.5f * dot;
    for (int aa = 0; aa < rem; ++aa) p[aa] += alpha * v[aa];
    for (int aa = 0; aa < rem; ++aa)
      for (int bb = 0; bb <= aa; ++bb)
        m[(i + 1 + aa) * N + (i + 1 + bb)] -= (v[aa] * p[bb] + p[aa] * v[bb]);
    m[(i + 1) * N + i] = beta;
    hCoeffs[i] = h;
  }
  float diag[N], sub[N - 1];
  for (int i = 0; i < N; ++i) diag[i] = m[i * N + i];
  for (int i = 0; i < N - 1; ++i) sub[i] = m[(i + 1) * N + i];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) V[r * N + c] = (r == c) ? 1.0f : 0.0f;
  for (int k = N - 2; k >= 0; --k) {
    // applyHouseholderOnTheLeft on V[k+1.., k+1..] with essential = m[k+2.., k]
    const int rows = N - k - 1;
    const float tau = hCoeffs[k];
    float *M = &V[(k + 1) * N + (k + 1)];
    const float *ess = rows > 1 ? &m[(k + 2) * N + k] : &m[(k + 1) * N + k];  // (no essential part for k = N - 2: never read)
    if (rows == 1) {
      M[0] *= (1.0f - tau);
    } else if (tau != 0.0f) {
      for (int j = 0; j < rows; ++j) {
        float tmp = 0.0f;
        for (int i = 1; i < rows; ++i) tmp += ess[(i - 1) * N] * M[i * N + j];
        tmp += M[j];
        M[j] -= tau * tmp;
        for (int i = 1; i < rows; ++i) M[i * N + j] -= tau * ess[(i - 1) * N] * tmp;
      }
    }
  }
  // computeFromTridiagonal_impl
  int end = N - 1, start = 0, iter = 0;
  const float precision = 2.0f * FLT_EPSILON;
  while (end > 0) {
    for (int i = start; i < end; ++i)
      if (fabsf(sub[i]) <= (fabsf(diag[i]) + fabsf(diag[i + 1])) * precision ||
          fabsf(sub[i]) <= FLT_MIN)
        sub[i] = 0.0f;
    while (end > 0 && sub[end - 1] == 0.0f) end--;
    if (end <= 0) break;
    iter++;
    if (iter > 30 * N) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0f) start--;
    {
      const float td = (diag[end - 1] - diag[end]) * 0.5f;
      const float e = sub[end - 1];
      float mu = diag[end];
      if (td == 0.0f) {
        mu -= fabsf(e);
      } else if (e != 0.0f) {
        const float e2 = e * e;
        const float h = eigen_hypot(td, e);
        if (e2 == 0.0f) mu -= e / ((td + (td > 0.0f ? h : -h)) / e);
        else mu -= e2 / (td + (td > 0.0f ? h : -h));
      }
      float x = diag[start] - mu;
      float z = sub[start];
      for (int k = start; k < end; ++k) {
        float c, s;
        make_givens(x, z, c, s);
        const float sdk = s * diag[k] + c * sub[k];
        const float dkp1 = s * sub[k] + c * diag[k + 1];
        diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        sub[k] = c * sdk - s * dkp1;
        if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
        x = sub[k];
        if (k < end - 1) {
          z = -s * sub[k + 1];
          sub[k + 1] = c * sub[k + 1];
        }
        for (int i = 0; i < N; ++i) {
          const float xi = V[i * N + k], yi = V[i * N + k + 1];
          V[i * N + k] = c * xi - s * yi;
          V[i * N + k + 1] = s * xi + c * yi;
        }
      }
    }
  }
  for (int i = 0; i < N - 1; ++i) {
    int k = 0;
    float mn = diag[i];
    for (int j = 1; j < N - i; ++j)
      if (diag[i + j] < mn) { mn = diag[i + j]; k = j; }
    if (k > 0) {
      float tmp = diag[i]; diag[i] = diag[k + i]; diag[k + i] = tmp;
      for (int r = 0; r < N; ++r) {
        tmp = V[r * N + i]; V[r * N + i] = V[r * N + k + i]; V[r * N + k + i] = tmp;
      }
    }
  }
  for (int i = 0; i < N; ++i) evals[i] = diag[i] * scale;
}

__device__ static void inverse6_dyn(const float *A, float *Ainv) {
  constexpr int N = 6;
  float lu[N * N];
  int piv[N];
  for (int i = 0; i < N * N; ++i) lu[i] = A[i];
  for (int k = 0; k < N; ++k) {
    int p = k;
    float best = fabsf(lu[k * N + k]);
    for (int r = k + 1; r < N; ++r)
      if (fabsf(lu[r * N + k]) > best) { best = fabsf(lu[r * N + k]); p = r; }
    piv[k] = p;
    if (p != k)
      for (int c = 0; c < N; ++c) { const float t = lu[k * N + c]; lu[k * N + c] = lu[p * N + c]; lu[p * N + c] = t; }
    if (lu[k * N + k] != 0.0f)
      for (int r = k + 1; r < N; ++r) lu[r * N + k] /= lu[k * N + k];
    for (int r = k + 1; r < N; ++r)
      for (int c = k + 1; c < N; ++c) lu[r * N + c] -= lu[r * N + k] * lu[k * N + c];
  }
  for (int col = 0; col < N; ++col) {
    float y[N];
    for (int r = 0; r < N; ++r) y[r] = (r == col) ? 1.0f : 0.0f;
    for (int k = 0; k < N; ++k) { const float t = y[k]; y[k] = y[piv[k]]; y[piv[k]] = t; }
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < r; ++c) y[r] -= lu[r * N + c] * y[c];
    for (int r = N - 1; r >= 0; --r) {
      for (int c = r + 1; c < N; ++c) y[r] -= lu[r * N + c] * y[c];
      y[r] /= lu[r * N + r];
    }
    for (int r = 0; r < N; ++r) Ainv[r * N + col] = y[r];
  }
}

// ScanMatch.cpp:206-260 for one iteration, executed by the whole solve block:
//   wave 0           6x6 column-pivoted Householder QR solve, one column per lane
//   wave 1, lane 0   (first iteration) eigenvalues of A^T A for the degeneracy test,
//                    concurrently with the solve
//   wave 0           pose update, convergence test, six sin/cos pairs in six lanes,
//                    next rotation
// sA/sb: A^T A (row-major) and A^T b in LDS.  Contains block barriers: call from all
// threads of a block of >= 128 threads.
struct GnShared {
  float A[36];
  float b[6];
  float matP[36];
  int degenerate;
};

__device__ static void gn_step_block(GNState *st, GnShared &sh, float eig_thresh, float dr_abort,
                                     float dt_abort, bool nan_reset = false) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int iter = st->loop_iter;  // uniform; the reference's iterCount
  float x[6] = {0, 0, 0, 0, 0, 0};
  if (wave == 0) {
    float col[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i] = lane < 6 ? sh.A[i * 6 + lane] : (lane == 6 ? sh.b[i] : 0.0f);
    colpiv_qr_solve6_wave(col, lane, x);  // :209
    if (lane == 0) st->clk[2] = wall_clock64();
  } else if (wave == 1 && lane == 0 && iter == 0) {  // :211-233
    // The reference asks its eigensolver for all six eigenvalues and looks at the smallest: "is any below the threshold?".
    // That question has a cheap sufficient answer: A - (thresh + m) I positive definite (an LDL^T without a non-positive pivot,
    // Sylvester's law of inertia) means every eigenvalue is above thresh + m -- and with m = 1 % of the threshold plus
    // 1e-5 of the trace (the fp32 eigensolver the oracle restates is good to ~1e-7 of the largest eigenvalue) the fp32
    // eigenvalues are above the threshold too.  Only a system that fails this test -- degenerate, or within the margin -- pays
    // for the eigenvalues (tridiagonalisation + implicit QR in one lane: 19 of this launch's 30 us); its decision is then the
    // eigensolver's, as before.  (NaN fails every comparison and takes the eigensolver's path.)
    bool clear = true;
    {
      double B[36], tr = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) tr += (double)sh.A[i * 6 + i];
      const double shift = (double)eig_thresh * 1.01 + 1.0e-5 * tr;
#pragma unroll
      for (int i = 0; i < 36; ++i) B[i] = (double)sh.A[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) B[i * 6 + i] -= shift;
#pragma unroll
      for (int j = 0; j < 6; ++j) {  // LDL^T in place on the lower triangle: B[j][j] = pivot, B[i][j] = L[i][j] (i > j)
        double d = B[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= B[j * 6 + k] * B[j * 6 + k] * B[k * 6 + k];
        clear = clear && d > 0.0;
        B[j * 6 + j] = d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double v = B[i * 6 + j];
#pragma unroll
          for (int k = 0; k < j; ++k) v -= B[i * 6 + k] * B[j * 6 + k] * B[k * 6 + k];
          B[i * 6 + j] = v / d;
        }
      }
    }
    if (clear) {
      sh.degenerate = 0;
    } else {
      float A[36], E[6];
#pragma unroll
      for (int i = 0; i < 36; ++i) A[i] = sh.A[i];
      eig_sym6_values(A, E);
      sh.degenerate = E[0] < eig_thresh ? 1 : 0;  // ascending: any below <=> the smallest
    }
  }
  __syncthreads();
  if (wave != 0) return;
  int degenerate;
  if (iter == 0) {
    degenerate = sh.degenerate;
    if (degenerate) {  // rare: needs the eigenvectors (quirk Q2), single lane
      if (lane == 0) {
        float A[36], E[6], V[36], V2[36], Vinv[36];
        for (int i = 0; i < 36; ++i) A[i] = sh.A[i];
        eig_sym6_dyn(A, E, V);
        for (int i = 0; i < 36; ++i) V2[i] = V[i];
        for (int i = 0; i < 6; ++i) {
          if (E[i] < eig_thresh) {
            for (int j = 0; j < 6; ++j) V2[i * 6 + j] = 0.0f;  // row i
          } else
            break;
        }
        inverse6_dyn(V, Vinv);  // :234
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) {
            float s = 0.0f;
            for (int kk = 0; kk < 6; ++kk) s += Vinv[r * 6 + kk] * V2[kk * 6 + c];
            sh.matP[r * 6 + c] = s;
            st->matP[r * 6 + c] = s;
          }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) st->degenerate = degenerate;
  } else {
    degenerate = st->degenerate;
    if (degenerate && lane < 36) sh.matP[lane] = st->matP[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (degenerate) {  // :237-240
    float x2[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) x2[i] = x[i];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float s = 0.0f;
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) s += sh.matP[r * 6 + kk] * x2[kk];
      x[r] = s;
    }
  }
  float pose[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) pose[i] = st->pose[i] + x[i];  // :242-247
  if (nan_reset) {  // LaserOdometry.cpp:622-634
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (!isfinite(pose[i])) pose[i] = 0.0f;
  }
  // :249-253 (rad2deg(float) -> float, pow(float,int) -> double)
  const double kPi = 3.14159265358979323846;
  const double r0 = (double)(float)((double)x[0] * 180.0 / kPi);
  const double r1 = (double)(float)((double)x[1] * 180.0 / kPi);
  const double r2 = (double)(float)((double)x[2] * 180.0 / kPi);
  const float dR = (float)sqrt(r0 * r0 + r1 * r1 + r2 * r2);
  const double t0 = (double)(x[3] * 100), t1 = (double)(x[4] * 100), t2 = (double)(x[5] * 100);
  const float dT = (float)sqrt(t0 * t0 + t1 * t1 + t2 * t2);
  // next rotation: lanes 0..2 half angles, 3..5 full angles (the host uses
  // std::sin/std::cos(float); the double-precision functions rounded to float agree)
  const int ai = lane < 3 ? lane : (lane < 6 ? lane - 3 : 0);
  float ang = ai == 0 ? pose[0] : (ai == 1 ? pose[1] : pose[2]);
  if (lane < 3) ang = 0.5f * ang;
  const float sv = (float)sin((double)ang);
  const float cv = (float)cos((double)ang);
  float hs[3], hc[3], fs[3], fc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    hs[i] = __shfl(sv, i, 64);
    hc[i] = __shfl(cv, i, 64);
    fs[i] = __shfl(sv, 3 + i, 64);
    fc[i] = __shfl(cv, 3 + i, 64);
  }
  if (lane == 0) {
    float R[9], t[3], sc[6];
    sincos_to_Rt_sc(pose, hs, hc, fs, fc, R, t, sc);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      st->pose[i] = pose[i];
      st->x[i] = x[i];
      st->sc[i] = sc[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) st->R[i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) st->t[i] = t[i];
    st->delta_r = dR;
    st->delta_t = dT;
    st->iter += 1;
    if (dR < dr_abort && dT < dt_abort) {  // :257-260
      st->converged = 1;
      st->done = 1;
    }
  }
}

constexpr int SOLVE_THREADS = 1024;
constexpr int SOLVE_GROUPS = 32;  // 32 row groups x 32 columns: the summation order of the cross-block reduction

// Deterministic cross-block reduction in fp64 of one scan's block records: the thread that owns (group grp, column col)
// adds rows grp, grp + 32, ... in order (32 independent loads in flight per pass, so up to 1024 sweep blocks cost a single
// memory round trip); red[grp][col] receives the group's sum.  Called by THREADS = 32 * G threads, each owning 32 / G groups:
// the order inside a group -- the only order that matters -- does not depend on G (the solve kernel: 1024 threads, the fused
// tail of a sweep block: 256).  COHERENT: the records were written by other workgroups of the same launch (relaxed
// agent-scope atomic loads, served by the coherence point).
template <int THREADS, bool COHERENT>
LSLAM_DEV void reduce_partials(const float *partials, int nb, double (*red)[NCOL]) {
  constexpr int G = THREADS / NCOL;       // groups worked on at a time
  constexpr int NG = SOLVE_GROUPS / G;    // groups per thread (1 for the solve kernel, 4 for a sweep block)
  constexpr int CH = NG == 1 ? 32 : 16;   // rows of a group fetched together: NG * CH loads in flight per thread
  const int tid = threadIdx.x, col = tid & 31, g0 = tid >> 5;
  double s[NG];
#pragma unroll
  for (int gg = 0; gg < NG; ++gg) s[gg] = 0.0;
  for (int b0 = 0; b0 < nb; b0 += SOLVE_GROUPS * CH) {  // one pass for up to 32 * CH blocks
    float v[NG][CH];
#pragma unroll
    for (int gg = 0; gg < NG; ++gg)
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int b = b0 + (g0 + G * gg) + u * SOLVE_GROUPS;
        if (COHERENT) v[gg][u] = b < nb ? __hip_atomic_load(partials + (size_t)b * NCOL + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        else v[gg][u] = b < nb ? partials[(size_t)b * NCOL + col] : 0.0f;
      }
#pragma unroll
    for (int gg = 0; gg < NG; ++gg)
#pragma unroll
      for (int u = 0; u < CH; ++u) s[gg] += (double)v[gg][u];  // rows grp, grp + 32, ... in order
  }
#pragma unroll
  for (int gg = 0; gg < NG; ++gg) red[g0 + G * gg][col] = s[gg];
}

// ScanMatch.cpp:141-145 + :206-260 on the reduced sums tot[32] (LDS): bookkeeping, the too-few-rows guard, A^T A / A^T b,
// gn_step_block, loop counter.  Block barriers inside: all threads of a block of >= 128 threads.
__device__ static void solve_finish(GNState *st, const double *tot, GnShared &sh, int &go, const SolveParams &p) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    st->clk[1] = wall_clock64();
    st->sweeps += 1;
    const int n_rows = (int)tot[COL_ROWS];
    st->n_rows = n_rows;
    st->n_line = (int)tot[COL_LINE];
    st->n_plane = (int)tot[COL_PLANE];
    st->score = tot[COL_SCORE];
    go = 1;
    if (n_rows < p.min_rows) {  // ScanMatch.cpp:141-145 (break) / LaserOdometry.cpp:501-503 (continue)
      go = 0;
      if (p.too_few_continue) {
        st->loop_iter += 1;
        if (st->loop_iter >= p.max_iterations) st->done = 1;
      } else {
        st->too_few = 1;
        st->done = 1;
      }
    }
  }
  if (tid < 36) {  // symmetric A^T A from the 21 reduced upper-triangular sums
    const int r = tid / 6, c = tid % 6;
    const int i = r < c ? r : c, j = r < c ? c : r;
    sh.A[tid] = (float)tot[COL_ATA + (i * 6 - (i * (i - 1)) / 2) + (j - i)];
  }
  if (tid < 6) sh.b[tid] = (float)tot[COL_ATB + tid];
  __syncthreads();
  if (!go) return;
  gn_step_block(st, sh, p.eig_thresh, p.delta_r_abort, p.delta_t_abort, p.nan_reset != 0);
  if (tid == 0) {
    st->loop_iter += 1;
    if (st->loop_iter >= p.max_iterations) st->done = 1;
  }
  if (tid == 0) st->clk[3] = wall_clock64();
}

}  // namespace lslam
