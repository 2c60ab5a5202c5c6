// lslam_posegraph.hip -- SE(3) pose-graph Levenberg-Marquardt on gfx950 (fp64).
//
// Replaces pose_graph::SolverG2O (pose_graph/solver_g2o.cpp:51-95: add_se3_node,
// add_se3_edge, optimize), i.e. g2o's VertexSE3 / EdgeSE3 / "lm_var".  g2o itself is not
// under /root/reference (and not pinned by it): the conventions restated here are g2o's
// published ones -- update X <- X * fromVectorMQT(d), error toVectorMQT(Z^-1 Xi^-1 Xj),
// LM schedule of OptimizationAlgorithmLevenberg -- see oracle/posegraph_oracle.py.
//
// Data parallel structure
//   pg_edge_kernel      one lane per edge of this rank's shard: error, analytic Jacobians,
//                       J^T Omega J blocks, J^T Omega e, chi2 -> per-edge records
//   pg_assemble_kernel  deterministic gather of the records into the block system
//                       [diag blocks | off-diagonal blocks | b | chi2]  (fixed order, no atomics)
//   (all-reduce of that buffer across ranks: callback, RCCL via torch.distributed)
//   pg_expand / pg_precond / pg_spmv / pg_cg_*   replicated block-Jacobi PCG on (H + lambda I)
//   pg_update_kernel    X <- X * fromVectorMQT(dx)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/lslam_c.h"

namespace lslam {  // lslam_comm.hip
hipError_t comm_allreduce_f64(lslam_comm *comm, double *buf, size_t count, hipStream_t s);
hipError_t comm_allgatherv_f64(lslam_comm *comm, int n_lists, double *const *bufs, const int64_t *const *offs, hipStream_t s);
int comm_world(const lslam_comm *comm);
int comm_rank(const lslam_comm *comm);
const char *debug_env(const char *name);  // lslam_api.hip: a test hook's value, only in a process started with LSLAM_DEBUG_HOOKS=1
}

namespace {

#define PG_DEV __device__ __forceinline__

struct Q4 { double x, y, z, w; };
struct V3 { double x, y, z; };

PG_DEV Q4 qmul(const Q4 &a, const Q4 &b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
PG_DEV Q4 qconj(const Q4 &q) { return {-q.x, -q.y, -q.z, q.w}; }
PG_DEV void qrotmat(const Q4 &q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
PG_DEV V3 mulR(const double R[9], const V3 &v) {
  return {R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z,
          R[6] * v.x + R[7] * v.y + R[8] * v.z};
}
PG_DEV V3 mulRt(const double R[9], const V3 &v) {
  return {R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z,
          R[2] * v.x + R[5] * v.y + R[8] * v.z};
}
PG_DEV void skew(const V3 &v, double S[9]) {
  S[0] = 0; S[1] = -v.z; S[2] = v.y;
  S[3] = v.z; S[4] = 0; S[5] = -v.x;
  S[6] = -v.y; S[7] = v.x; S[8] = 0;
}

struct Pose { V3 t; Q4 q; };
PG_DEV Pose load_pose(const double *p) { return {{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}}; }

// e = toVectorMQT(Z^-1 * Xi^-1 * Xj); optionally the Jacobians w.r.t. the local updates.
// Ji/Jj row-major 6x6.
template <bool JAC>
PG_DEV void edge_error(const Pose &xi, const Pose &xj, const Pose &z, double e[6], double *Ji,
                       double *Jj) {
  double Ri[9], Rz[9];
  qrotmat(xi.q, Ri);
  qrotmat(z.q, Rz);
  const V3 d = {xj.t.x - xi.t.x, xj.t.y - xi.t.y, xj.t.z - xi.t.z};
  const V3 tb = mulRt(Ri, d);                                  // Xi^-1 Xj translation
  const Q4 qb = qmul(qconj(xi.q), xj.q);                       // Xi^-1 Xj rotation
  const V3 te = mulRt(Rz, {tb.x - z.t.x, tb.y - z.t.y, tb.z - z.t.z});
  Q4 qe = qmul(qconj(z.q), qb);
  const double nrm = sqrt(qe.x * qe.x + qe.y * qe.y + qe.z * qe.z + qe.w * qe.w);
  qe = {qe.x / nrm, qe.y / nrm, qe.z / nrm, qe.w / nrm};
  const double sgn = qe.w < 0 ? -1.0 : 1.0;
  qe = {sgn * qe.x, sgn * qe.y, sgn * qe.z, sgn * qe.w};
  e[0] = te.x; e[1] = te.y; e[2] = te.z;
  e[3] = qe.x; e[4] = qe.y; e[5] = qe.z;
  if (!JAC) return;
  // ---- vertex j: E' = E * Delta_j ------------------------------------------------
  double Re[9];
  qrotmat(qe, Re);
  double Sv[9];
  skew({qe.x, qe.y, qe.z}, Sv);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Jj[r * 6 + c] = Re[r * 3 + c];                                  // d te / d dt_j = R_e
      Jj[r * 6 + 3 + c] = 0.0;
      Jj[(3 + r) * 6 + c] = 0.0;
      Jj[(3 + r) * 6 + 3 + c] = (r == c ? qe.w : 0.0) + Sv[r * 3 + c];  // w_e I + [v_e]x
    }
  // ---- vertex i: E' = Z^-1 Delta_i^-1 Xi^-1 Xj ----------------------------------
  // translation: -Rz^T ; Rz^T * 2 [t_b]x
  double Stb[9];
  skew(tb, Stb);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Ji[r * 6 + c] = -Rz[c * 3 + r];
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += Rz[k * 3 + r] * 2.0 * Stb[k * 3 + c];
      Ji[r * 6 + 3 + c] = s;
      Ji[(3 + r) * 6 + c] = 0.0;
    }
  // rotation: q' = qz* (x) (1,-d) (x) qb ;  d q'_v / d d =
  //   wz (-wb I + [vb]x) - vz vb^T - [vz]x (-wb I + [vb]x), times the sign of q_e
  const V3 vb = {qb.x, qb.y, qb.z}, vz = {z.q.x, z.q.y, z.q.z};
  double Sb[9], Sz[9], M[9];
  skew(vb, Sb);
  skew(vz, Sz);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[r * 3 + c] = (r == c ? -qb.w : 0.0) + Sb[r * 3 + c];
  const double vzv[3] = {vz.x, vz.y, vz.z}, vbv[3] = {vb.x, vb.y, vb.z};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = z.q.w * M[r * 3 + c] - vzv[r] * vbv[c];
#pragma unroll
      for (int k = 0; k < 3; ++k) s -= Sz[r * 3 + k] * M[k * 3 + c];
      Ji[(3 + r) * 6 + 3 + c] = sgn * s / nrm;
    }
}

// Per-edge record: [Hii(36) bi(6) | Hjj(36) bj(6) | Hoff(36) | chi2] = 121 doubles
constexpr int REC = 121;

__global__ void pg_edge_kernel(const double *poses, const int32_t *ij, const double *meas,
                               const double *info, int e_begin, int e_end, int fixed, double *rec, double *chi) {
  const int e = e_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= e_end) return;
  const int i = ij[2 * e], j = ij[2 * e + 1];
  const Pose xi = load_pose(poses + 7 * i), xj = load_pose(poses + 7 * j), z = load_pose(meas + 7 * e);
  double er[6], Ji[36], Jj[36], Om[36];
  edge_error<true>(xi, xj, z, er, Ji, Jj);
#pragma unroll
  for (int k = 0; k < 36; ++k) Om[k] = info[(size_t)e * 36 + k];
  double *o = rec + (size_t)(e - e_begin) * REC;
  // A = J^T Omega
  double Ai[36], Aj[36];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double si = 0.0, sj = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        si += Ji[k * 6 + r] * Om[k * 6 + c];
        sj += Jj[k * 6 + r] * Om[k * 6 + c];
      }
      Ai[r * 6 + c] = si;
      Aj[r * 6 + c] = sj;
    }
  const bool fi = (i == fixed), fj = (j == fixed);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double hii = 0.0, hjj = 0.0, hij = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        hii += Ai[r * 6 + k] * Ji[k * 6 + c];
        hjj += Aj[r * 6 + k] * Jj[k * 6 + c];
        hij += Ai[r * 6 + k] * Jj[k * 6 + c];
      }
      o[r * 6 + c] = fi ? 0.0 : hii;
      o[42 + r * 6 + c] = fj ? 0.0 : hjj;
      // off-diagonal block is stored for the pair (min,max): transposed when i > j
      const double v = (fi || fj) ? 0.0 : hij;
      if (i < j) o[84 + r * 6 + c] = v; else o[84 + c * 6 + r] = v;
    }
    double bi = 0.0, bj = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      bi += Ai[r * 6 + k] * er[k];
      bj += Aj[r * 6 + k] * er[k];
    }
    o[36 + r] = fi ? 0.0 : -bi;
    o[42 + 36 + r] = fj ? 0.0 : -bj;
  }
  double c2 = 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += Om[r * 6 + c] * er[c];
    c2 += er[r] * s;
  }
  o[120] = c2;
  chi[e - e_begin] = c2;  // also compact: the sum over the edges then reads consecutive doubles instead of one per record
}

// chi2 only (trial evaluation): per-edge value
__global__ void pg_chi2_kernel(const double *poses, const int32_t *ij, const double *meas,
                               const double *info, int e_begin, int e_end, double *out) {
  const int e = e_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= e_end) return;
  const int i = ij[2 * e], j = ij[2 * e + 1];
  const Pose xi = load_pose(poses + 7 * i), xj = load_pose(poses + 7 * j), z = load_pose(meas + 7 * e);
  double er[6];
  edge_error<false>(xi, xj, z, er, nullptr, nullptr);
  double c2 = 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += info[(size_t)e * 36 + r * 6 + c] * er[c];
    c2 += er[r] * s;
  }
  out[e - e_begin] = c2;
}

// fixed-order sum of n doubles into *dst, single block of 1024 (four independent accumulators per thread keep four
// loads in flight: with one, the 25 000-edge chi2 sum was a 36 us chain of dependent strided loads)
constexpr int SUM_BLOCK = 1024;
__global__ __launch_bounds__(SUM_BLOCK) void pg_sum_kernel(const double *src, int n, int stride, double *dst) {
  __shared__ double sh[SUM_BLOCK];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = threadIdx.x;
  for (; k + 3 * SUM_BLOCK < n; k += 4 * SUM_BLOCK) {
    const double a0 = src[(size_t)k * stride], a1 = src[(size_t)(k + SUM_BLOCK) * stride];
    const double a2 = src[(size_t)(k + 2 * SUM_BLOCK) * stride], a3 = src[(size_t)(k + 3 * SUM_BLOCK) * stride];
    s0 += a0; s1 += a1; s2 += a2; s3 += a3;
  }
  for (; k < n; k += SUM_BLOCK) s0 += src[(size_t)k * stride];
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int w = SUM_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *dst = sh[0];
}

// Gather the per-edge records into the block system, fixed order (edge index order).
// vertex part: thread per (vertex, k<42): sum over incident shard edges.
__global__ void pg_assemble_vertex_kernel(const double *rec, const int32_t *v_ptr, const int32_t *v_adj,
                                          int n_v, int fixed, double *diag, double *b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = t / 42, k = t % 42;
  if (v >= n_v) return;
  double s = 0.0;
  for (int a = v_ptr[v]; a < v_ptr[v + 1]; ++a) {
    const int code = v_adj[a];  // (local edge << 1) | role
    s += rec[(size_t)(code >> 1) * REC + (code & 1) * 42 + k];
  }
  if (v == fixed) s = (k < 36 && (k / 6 == k % 6)) ? 1.0 : 0.0;  // identity row: dx_fixed = 0
  if (k < 36) diag[(size_t)v * 36 + k] = s; else b[(size_t)v * 6 + (k - 36)] = s;
}
__global__ void pg_assemble_off_kernel(const double *rec, const int32_t *o_ptr, const int32_t *o_adj,
                                       int n_off, double *off) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int blk = t / 36, k = t % 36;
  if (blk >= n_off) return;
  double s = 0.0;
  for (int a = o_ptr[blk]; a < o_ptr[blk + 1]; ++a) s += rec[(size_t)o_adj[a] * REC + 84 + k];
  off[(size_t)blk * 36 + k] = s;
}

// Full block-CSR copy of the symmetric system for the solver: entry a of row v refers to
// block `src` (diag: v; off: n_v + id), transposed or not.
__global__ void pg_expand_kernel(const double *sys, const int32_t *row_src, int n_entries, double lambda,
                                 double *vals) {
  // solver layout: six column planes, vals[c * n_items + entry * 6 + r] (n_items = 6 n_entries), so
  // that the product kernel's lanes (one per (entry, r)) read consecutive doubles
  const size_t n_items = (size_t)n_entries * 6;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_items * 6) return;
  const int c = (int)(t / n_items);
  const size_t item = t - (size_t)c * n_items;
  const int a = (int)(item / 6), r = (int)(item % 6);
  const int code = row_src[a];  // (block index in [diag|off] << 2) | transposed<<1 | is_diag
  const int blk = code >> 2;
  const bool tr = (code & 2) != 0, dg = (code & 1) != 0;
  double v = sys[(size_t)blk * 36 + (tr ? c * 6 + r : r * 6 + c)];
  if (dg && r == c) v += lambda;
  vals[t] = v;
}

// block-Jacobi preconditioner: inverse of the (damped) 6x6 diagonal blocks via Cholesky
__global__ void pg_precond_kernel(const double *vals, const int32_t *row_ptr, int n_v, size_t n_items,
                                  double *minv) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_v) return;
  const double *Ad = vals + (size_t)row_ptr[v] * 6;  // the diagonal entry is first in its row
  double A[36];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) A[i * 6 + j] = Ad[(size_t)j * n_items + i];
  double L[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) L[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = (i == j) ? sqrt(s > 1e-300 ? s : 1e-300) : s / L[j * 6 + j];
    }
  // inverse: solve L L^T X = I column by column
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
      y[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      double s = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * y[k];
      y[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) minv[(size_t)v * 36 + i * 6 + c] = y[i];
  }
}

// ---- PCG on the expanded block-CSR, one thread per scalar row ---------------------
// scal: [0] rz_old  [1] alpha-den (p.q)  [2] rz_new  [3] rr  [4] bb  [5] done flag (as double)
constexpr int CG_BLOCK = 128;
constexpr int CG_ROWS = 126;  // rows per block: a multiple of 6, so a vertex never straddles blocks

PG_DEV double block_sum(double v, double *sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int w = CG_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}
// Fixed-order sums: xor butterfly inside each wavefront (every lane ends with the same
// total), then the four wavefront totals in order.  M values at once.
template <int M, int BLOCK = CG_BLOCK>
PG_DEV void block_sum_m(double (&v)[M], double *sh /* [M * BLOCK / 64] */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int m = 0; m < M; ++m) v[m] += __shfl_xor(v[m], off);
  constexpr int W = BLOCK / 64;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int m = 0; m < M; ++m) sh[m * W + (threadIdx.x >> 6)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < M; ++m) {
    double s = sh[m * W];
#pragma unroll
    for (int w = 1; w < W; ++w) s += sh[m * W + w];
    v[m] = s;
  }
  __syncthreads();
}
// sums of M per-block partial arrays, identical in every thread of every block
template <int M, int BLOCK = CG_BLOCK>
PG_DEV void sum_partials_m(const double *const (&part)[M], int n, double (&out)[M], double *sh) {
#pragma unroll
  for (int m = 0; m < M; ++m) {
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += BLOCK) s += part[m][k];
    out[m] = s;
  }
  block_sum_m<M, BLOCK>(out, sh);
}

struct CgArgs {
  const double *vals;  // six planes of n_items
  const int32_t *row_ptr, *row_col, *row_of;  // block-CSR; row_of[e] = the vertex entry e belongs to
  const double *minv;
  const double *b;
  double *x, *r, *z;
  double *d;             // [n_items] per-(entry,row) products A_e[r,:] . p[col_e]
  double *p[2];          // search direction of iteration k lives in p[k & 1]
  double *part_pq;       // [n_pblocks]
  double *part_rz[2];    // r.z entering iteration k lives in part_rz[k & 1]
  double *part_rr;       // [n_blocks]
  double *scal;          // [3] rr  [4] bb  [5] done flag  [6] iterations done
  int n6, n_blocks, n_items, n_pblocks;
  int n_parts;           // entries of part_rz[] / part_rr: n_blocks + the coarse level's blocks (zero when it is off)
  double tol2;
  // row-sharded solve (lslam_pg_set_row_shard): this rank's items [item_begin, item_end) and row blocks from block_begin on;
  // the partial arrays then hold ONE entry each, the sum over all ranks (n_parts = n_pblocks = 1)
  int item_begin, item_end, block_begin;
};

// ---- second level of the preconditioner (see solve()) -----------------------------------------------------------
struct CoarseArgs {
  const double *poses;    // current estimate (7 per vertex)
  double *P;              // [n_v][36] prolongation blocks: delta_i = P_i xi_a
  double *Ac;             // [n_c][n_c] coarse matrix, inverted in place
  double *rc, *yc;        // [n_c]
  double *Rbuf, *Cbuf, *Bbuf;  // block Gauss-Jordan scratch: [6][n_c], [n_c][6], [36]
  const int32_t *cb_ptr, *cb_ent, *cb_ab;  // coarse blocks: fine entries of each, its (row, column) aggregate
  const int32_t *agg_of, *agg_ptr, *agg_mem;  // vertex -> aggregate; aggregate -> its members (at most PG_AGG_MAX), first = its origin
  int n_v, G, na, n_c, n_cb, n_cblk;
};

__global__ __launch_bounds__(CG_BLOCK) void pg_cg_init_kernel(CgArgs a) {
  __shared__ double sh[2 * CG_BLOCK / 64];
  const int row = blockIdx.x * CG_ROWS + threadIdx.x;
  double s[2] = {0.0, 0.0};
  if (threadIdx.x < CG_ROWS && row < a.n6) {
    const int v = row / 6, rr = row % 6;
    a.x[row] = 0.0;
    const double bi = a.b[row];
    a.r[row] = bi;
    double z = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) z += a.minv[(size_t)v * 36 + rr * 6 + c] * a.b[v * 6 + c];
    a.z[row] = z;
    s[0] = bi * z;
    s[1] = bi * bi;
  }
  block_sum_m<2>(s, sh);
  if (threadIdx.x == 0) {
    a.part_rz[0][blockIdx.x] = s[0];
    a.part_rr[blockIdx.x] = s[1];
  }
}
__global__ __launch_bounds__(CG_BLOCK) void pg_cg_init2_kernel(CgArgs a) {  // one block: |b|^2
  __shared__ double sh[CG_BLOCK / 64];
  const double *const parts[1] = {a.part_rr};
  double s[1];
  sum_partials_m<1>(parts, a.n_blocks, s, sh);
  if (threadIdx.x == 0) {
    a.scal[3] = s[0];
    a.scal[4] = s[0];
    a.scal[5] = (s[0] == 0.0) ? 1.0 : 0.0;
    a.scal[6] = 0.0;
  }
}

// Iteration k, first half: convergence test on the previous update, then the products
// d[e,r] = A_e[r,:] . p_k[col_e] with one lane per (entry, row) -- p_k = z + beta p_{k-1} is
// recomputed where it is needed, so no grid-wide barrier separates the direction update from
// the product -- and the partials of p_k . (A p_k) summed in item order.
constexpr int PROD_BLOCK = 256;
__global__ __launch_bounds__(PROD_BLOCK) void pg_cg_prod_kernel(CgArgs a, int k) {
  __shared__ double sh[3 * PROD_BLOCK / 64];
  if (a.scal[5] != 0.0) return;
  // operand loads first: they do not depend on beta, and the launch is latency bound
  const double *po = (k > 0) ? a.p[(k + 1) & 1] : a.z;  // k == 0: beta = 0, p_0 = z
  double *pn = a.p[k & 1];
  const int t = a.item_begin + blockIdx.x * PROD_BLOCK + threadIdx.x;
  const bool on = t < a.item_end;
  const int tt = on ? t : a.item_end - 1;
  const int e = tt / 6, r = tt - e * 6;
  const int v = a.row_of[e];
  const size_t c0 = (size_t)a.row_col[e] * 6;
  const int row = v * 6 + r;
  const bool diag = e == a.row_ptr[v];
  double A[6], zc[6], pc[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    A[c] = a.vals[(size_t)c * a.n_items + tt];
    zc[c] = a.z[c0 + c];
    pc[c] = po[c0 + c];
  }
  const double zr = a.z[row], pr0 = po[row];
  double beta = 0.0;
  if (k > 0) {
    const double *const parts[3] = {a.part_rz[k & 1], a.part_rz[(k + 1) & 1], a.part_rr};
    double s[3];
    sum_partials_m<3, PROD_BLOCK>(parts, a.n_parts, s, sh);
    if (s[2] <= a.tol2 * a.scal[4] || !(s[0] > 0.0)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.scal[3] = s[2];
        a.scal[5] = 1.0;
      }
      return;
    }
    beta = s[0] / s[1];
  }
  double pq[1] = {0.0};
  if (on) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += A[c] * (zc[c] + beta * pc[c]);
    a.d[t] = s;
    const double pr = zr + beta * pr0;
    if (diag) pn[row] = pr;  // the diagonal entry publishes p_k
    pq[0] = pr * s;
  }
  block_sum_m<1, PROD_BLOCK>(pq, sh);
  if (threadIdx.x == 0) a.part_pq[blockIdx.x] = pq[0];
}
// Iteration k, second half: q = row sums of d ; x += alpha p ; r -= alpha q ; z = M r ; partial r.z, r.r
__global__ __launch_bounds__(CG_BLOCK) void pg_cg_update_kernel(CgArgs a, int k) {
  __shared__ double sh[2 * CG_BLOCK / 64];
  __shared__ double rloc[CG_BLOCK];
  if (a.scal[5] != 0.0) return;
  const int row = (a.block_begin + blockIdx.x) * CG_ROWS + threadIdx.x;
  const bool on = threadIdx.x < CG_ROWS && row < a.n6;
  const int rowc = on ? row : 0;
  const int v = rowc / 6, rrow = rowc % 6;
  // everything that does not need alpha first (latency bound launch)
  double q = 0.0;  // (A p)[row]: this row's products, in entry order
  {
    const int e0 = a.row_ptr[v], e1 = a.row_ptr[v + 1];
    for (int e = e0; e < e1; e += 4) {
      double dv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) dv[j] = a.d[(size_t)((e + j < e1) ? e + j : e1 - 1) * 6 + rrow];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (e + j < e1) q += dv[j];
    }
  }
  const double xr = a.x[rowc], pr = a.p[k & 1][rowc], rr0 = a.r[rowc];
  double mi[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) mi[c] = a.minv[(size_t)v * 36 + rrow * 6 + c];
  double s0[1], s1[1];
  {
    const double *const p0[1] = {a.part_rz[k & 1]};
    const double *const p1[1] = {a.part_pq};
    sum_partials_m<1>(p0, a.n_parts, s0, sh);
    sum_partials_m<1>(p1, a.n_pblocks, s1, sh);
  }
  const double alpha = s0[0] / s1[0];
  double rnew = 0.0;
  if (on) {
    a.x[row] = xr + alpha * pr;
    rnew = rr0 - alpha * q;
    a.r[row] = rnew;
  }
  rloc[threadIdx.x] = rnew;
  __syncthreads();
  double o[2] = {0.0, 0.0};
  if (on) {
    const int base = (int)threadIdx.x - rrow;  // this vertex's 6 residuals sit in this block
    double z = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) z += mi[c] * rloc[base + c];
    a.z[row] = z;
    o[0] = rnew * z;
    o[1] = rnew * rnew;
  }
  block_sum_m<2>(o, sh);
  if (threadIdx.x == 0) {
    a.part_rz[(k + 1) & 1][blockIdx.x] = o[0];
    a.part_rr[blockIdx.x] = o[1];
    if (blockIdx.x == 0) a.scal[6] = (double)(k + 1);
  }
}
// one block: convergence test after iteration k-1 (what pg_cg_spmv_kernel(k) would decide),
// run before the host reads the flag
__global__ __launch_bounds__(CG_BLOCK) void pg_cg_check_kernel(CgArgs a, int k) {
  __shared__ double sh[2 * CG_BLOCK / 64];
  if (a.scal[5] != 0.0) return;
  const double *const parts[2] = {a.part_rz[k & 1], a.part_rr};
  double s[2];
  sum_partials_m<2>(parts, a.n_parts, s, sh);
  if (threadIdx.x == 0) {
    a.scal[3] = s[1];
    if (s[1] <= a.tol2 * a.scal[4] || !(s[0] > 0.0)) a.scal[5] = 1.0;
  }
}

// ---- row-sharded solve (large graphs on several GPUs; lslam_pg_set_row_shard) ---------------------------------------------
// Every rank owns a contiguous range of vertex rows: it multiplies, updates and preconditions only those, and the ranks meet
// twice per iteration -- one scalar (p . A p), then the vector z with r . z and r . r behind it -- through the same all-reduce
// the linearisation uses: each rank contributes its own rows of z to a zero-padded buffer, the sum IS the gathered vector.
// The direction p_k = z + beta p_(k-1) is recomputed by every rank for ALL rows from the gathered z (an axpy of 6 n values:
// cheaper than a second exchange); every rank sees the same reduced scalars, hence takes the same decisions.
__global__ void pg_rs_direction_kernel(CgArgs a, int k) {  // p_k for every row (what pg_cg_prod_kernel recomputes per column)
  if (a.scal[5] != 0.0) return;
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.n6) return;
  double beta = 0.0;
  if (k > 0) {
    const double rz = a.part_rz[k & 1][0], rz_old = a.part_rz[(k + 1) & 1][0], rr = a.part_rr[0];
    if (rr <= a.tol2 * a.scal[4] || !(rz > 0.0)) return;  // pg_cg_prod_kernel ends the solve on the same numbers
    beta = rz / rz_old;
  }
  const double *po = (k > 0) ? a.p[(k + 1) & 1] : a.z;
  a.p[k & 1][row] = a.z[row] + beta * (k > 0 ? po[row] : 0.0);
}
__global__ void pg_rs_zero_others_kernel(double *v, int n6, int row_begin, int row_end) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row < n6 && (row < row_begin || row >= row_end)) v[row] = 0.0;
}
__global__ void pg_rs_pack_kernel(const double *src, double *dst, int n6, int row_begin, int row_end) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row < n6) dst[row] = (row >= row_begin && row < row_end) ? src[row] : 0.0;
}
__global__ void pg_rs_tail_sum_kernel(double *sums /* [rz, rr] */, const double *tail /* per rank: rz, rr */, int world) {
  if (threadIdx.x < 2) {  // in rank order on every rank: the same bits everywhere
    double s = 0.0;
    for (int r = 0; r < world; ++r) s += tail[2 * r + threadIdx.x];
    sums[threadIdx.x] = s;
  }
}
__global__ void pg_rs_copy_kernel(const double *src, double *d0, double *d1, double *d2) {  // up to three scalars to their slots
  if (threadIdx.x == 0) {
    if (d0) *d0 = src[0];
    if (d1) *d1 = src[1];
    if (d2) *d2 = src[2];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Second level of the preconditioner: rigid motions of AGGREGATES of up to G keyframes that are close in the graph.
// A long trajectory bends almost for free -- moving a bundle of keyframes rigidly only strains the edges that leave
// it -- and those global modes are what block-Jacobi PCG needs thousands of iterations for once lambda is small.
// The aggregates are grown breadth-first over the graph's edges (odometry AND loop closures, lslam_pg_create), so a place
// that was visited on several laps -- keyframes tied together by loop edges -- falls into one aggregate; runs of
// consecutive keyframes instead (what this level used first) miss exactly those couplings and need six times the
// iterations on the bench graph (1 079 against 175 at lambda -> 0, numpy prototype and device agree).  The coarse
// unknown of aggregate a is a world-frame twist xi_a = (rho, phi) about the aggregate's first keyframe c_a; applied on
// the left it is, to first order, the local increment (g2o's right-multiplied [dt, dq]) delta_i = P_i xi_a of every
// keyframe i = (R_i, t_i) of the aggregate,  P_i = [ R_i^T   -R_i^T [t_i - c_a]x ;  0   R_i^T / 2 ].
// M^-1 = blockdiag(A)^-1 + P (P^T A P)^-1 P^T  (additive two-level Schwarz, symmetric positive definite).  The coarse
// matrix is dense (6 ceil(n/G) unknowns), inverted once per damped system by block Gauss-Jordan; per PCG iteration the
// level costs a restriction, a dense matrix-vector product and a prolongation.  No atomics: every sum has a fixed order,
// so the solve stays bit-reproducible and identical on every rank of a sharded run.
__global__ void pgc_P_kernel(CoarseArgs c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c.n_v) return;
  const Pose x = load_pose(c.poses + 7 * i);
  const Pose x0 = load_pose(c.poses + 7 * c.agg_mem[c.agg_ptr[c.agg_of[i]]]);
  double R[9];
  qrotmat(x.q, R);
  const double d[3] = {x.t.x - x0.t.x, x.t.y - x0.t.y, x.t.z - x0.t.z};
  // S = [d]x
  const double S[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
  double *P = c.P + (size_t)i * 36;
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 3; ++col) {
      const double rt = R[col * 3 + r];  // R^T[r][col]
      double m = 0.0;                    // (R^T S)[r][col]
      for (int k = 0; k < 3; ++k) m += R[k * 3 + r] * S[k * 3 + col];
      P[r * 6 + col] = rt;
      P[r * 6 + 3 + col] = -m;
      P[(3 + r) * 6 + col] = 0.0;
      P[(3 + r) * 6 + 3 + col] = 0.5 * rt;
    }
}

// A_c[a][b] = sum over the fine entries (i, j) with i in a, j in b of P_i^T A_ij P_j; one 64-thread block per coarse block
__global__ __launch_bounds__(64) void pgc_assemble_kernel(CoarseArgs c, const double *vals, const int32_t *row_of,
                                                          const int32_t *row_col, size_t n_items) {
  __shared__ double T[36];
  const int cb = blockIdx.x, t = threadIdx.x;
  const int r = t / 6, col = t % 6;  // t < 36
  double acc = 0.0;
  for (int q = c.cb_ptr[cb]; q < c.cb_ptr[cb + 1]; ++q) {
    const int e = c.cb_ent[q];
    const int i = row_of[e], j = row_col[e];
    if (t < 36) {  // T = A_e P_j
      double s = 0.0;
      for (int l = 0; l < 6; ++l) s += vals[(size_t)l * n_items + (size_t)e * 6 + r] * c.P[(size_t)j * 36 + l * 6 + col];
      T[t] = s;
    }
    __syncthreads();
    if (t < 36) {
      for (int k = 0; k < 6; ++k) acc += c.P[(size_t)i * 36 + k * 6 + r] * T[k * 6 + col];
    }
    __syncthreads();
  }
  if (t < 36) c.Ac[(size_t)(c.cb_ab[2 * cb] * 6 + r) * c.n_c + c.cb_ab[2 * cb + 1] * 6 + col] = acc;
}

// block Gauss-Jordan, pivot block k: B = A_kk^-1, R = B A_k*, C = A_*k  (one workgroup)
__global__ __launch_bounds__(256) void pgc_gj_pivot_kernel(CoarseArgs c, int k) {
  __shared__ double B[36];
  const int n = c.n_c, t = threadIdx.x;
  if (t == 0) {  // 6x6 inverse by Gauss-Jordan with partial pivoting
    double M[6][12];
    for (int r = 0; r < 6; ++r)
      for (int q = 0; q < 6; ++q) {
        M[r][q] = c.Ac[(size_t)(k * 6 + r) * n + k * 6 + q];
        M[r][6 + q] = r == q ? 1.0 : 0.0;
      }
    for (int p = 0; p < 6; ++p) {
      int best = p;
      for (int r = p + 1; r < 6; ++r)
        if (fabs(M[r][p]) > fabs(M[best][p])) best = r;
      if (best != p)
        for (int q = 0; q < 12; ++q) { const double tmp = M[p][q]; M[p][q] = M[best][q]; M[best][q] = tmp; }
      const double inv = 1.0 / M[p][p];
      for (int q = 0; q < 12; ++q) M[p][q] *= inv;
      for (int r = 0; r < 6; ++r)
        if (r != p) {
          const double f = M[r][p];
          for (int q = 0; q < 12; ++q) M[r][q] -= f * M[p][q];
        }
    }
    for (int r = 0; r < 6; ++r)
      for (int q = 0; q < 6; ++q) { B[r * 6 + q] = M[r][6 + q]; c.Bbuf[r * 6 + q] = M[r][6 + q]; }
  }
  __syncthreads();
  for (int j = t; j < n; j += 256) {
    double a[6];
    for (int r = 0; r < 6; ++r) {
      a[r] = c.Ac[(size_t)(k * 6 + r) * n + j];
      c.Cbuf[(size_t)j * 6 + r] = c.Ac[(size_t)j * n + k * 6 + r];
    }
    for (int r = 0; r < 6; ++r) {
      double s = 0.0;
      for (int q = 0; q < 6; ++q) s += B[r * 6 + q] * a[q];
      c.Rbuf[(size_t)r * n + j] = s;
    }
  }
}
// ... and the update of every entry with it
__global__ __launch_bounds__(256) void pgc_gj_update_kernel(CoarseArgs c, int k) {
  const int n = c.n_c;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * n) return;
  const int i = (int)(idx / n), j = (int)(idx % n);
  const bool ik = i / 6 == k, jk = j / 6 == k;
  double v;
  if (ik && jk) {
    v = c.Bbuf[(i - k * 6) * 6 + (j - k * 6)];
  } else if (ik) {
    v = c.Rbuf[(size_t)(i - k * 6) * n + j];
  } else if (jk) {
    double s = 0.0;
    for (int m = 0; m < 6; ++m) s += c.Cbuf[(size_t)i * 6 + m] * c.Bbuf[m * 6 + (j - k * 6)];
    v = -s;
  } else {
    double s = 0.0;
    for (int m = 0; m < 6; ++m) s += c.Cbuf[(size_t)i * 6 + m] * c.Rbuf[(size_t)m * n + j];
    v = c.Ac[idx] - s;
  }
  c.Ac[idx] = v;
}

// r_c = P^T r: one wavefront per aggregate
__global__ __launch_bounds__(64) void pgc_restrict_kernel(CoarseArgs c, const double *r) {
  const int a = blockIdx.x, j = threadIdx.x;
  const int m0 = c.agg_ptr[a], m1 = c.agg_ptr[a + 1];
  double w[6] = {0, 0, 0, 0, 0, 0};
  for (int jj = j; m0 + jj < m1; jj += 64) {
    const int i = c.agg_mem[m0 + jj];
    const double *P = c.P + (size_t)i * 36;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double rk = r[(size_t)i * 6 + k];
#pragma unroll
      for (int m = 0; m < 6; ++m) w[m] += P[k * 6 + m] * rk;
    }
  }
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    double s = w[m];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (j == 0) c.rc[a * 6 + m] = s;
  }
}
// y_c = A_c^-1 r_c and this block's share of r . (P y_c) = r_c . y_c; four rows per workgroup
__global__ __launch_bounds__(256) void pgc_mv_kernel(CoarseArgs c, double *part_rz_extra) {
  __shared__ double dots[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  double s = 0.0;
  if (row < c.n_c)
    for (int col = lane; col < c.n_c; col += 64) s += c.Ac[(size_t)row * c.n_c + col] * c.rc[col];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) {
    if (row < c.n_c) c.yc[row] = s;
    dots[w] = row < c.n_c ? s * c.rc[row] : 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) part_rz_extra[blockIdx.x] = ((dots[0] + dots[1]) + dots[2]) + dots[3];
}
// The two steps above in one launch (a PCG iteration is a chain of dependent launches, each worth ~7 us): one workgroup
// per aggregate computes ITS six rows of y_c = A_c^-1 r_c, its share r_c[a] . y_c[a] of r . z, and prolongs to its members.
__global__ __launch_bounds__(256) void pgc_mvp_kernel(CoarseArgs c, double *z, double *part_rz_extra) {
  __shared__ double sh[6 * 256 / 64];
  __shared__ double y6[6];
  const int a = blockIdx.x, t = threadIdx.x;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int col = t; col < c.n_c; col += 256) {
    const double rc = c.rc[col];
#pragma unroll
    for (int m = 0; m < 6; ++m) s[m] += c.Ac[(size_t)(a * 6 + m) * c.n_c + col] * rc;
  }
  block_sum_m<6, 256>(s, sh);  // every thread holds the six sums
  if (t == 0) {
    double dot = 0.0;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      y6[m] = s[m];
      dot += s[m] * c.rc[a * 6 + m];
    }
    part_rz_extra[a] = dot;
  }
  __syncthreads();
  const int m0 = c.agg_ptr[a], cnt = (c.agg_ptr[a + 1] - m0) * 6;
  for (int idx = t; idx < cnt; idx += 256) {
    const int i = c.agg_mem[m0 + idx / 6], r = idx % 6;
    const double *P = c.P + (size_t)i * 36 + r * 6;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < 6; ++m) acc += P[m] * y6[m];
    z[(size_t)i * 6 + r] += acc;
  }
}
// z += P y_c
__global__ void pgc_prolong_kernel(CoarseArgs c, double *z) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= c.n_v * 6) return;
  const int i = row / 6, r = row % 6;
  const double *P = c.P + (size_t)i * 36 + r * 6;
  const double *y = c.yc + (size_t)c.agg_of[i] * 6;
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < 6; ++m) s += P[m] * y[m];
  z[row] += s;
}


// ------------------------------------------------------------------------------------------------------------------
// The whole PCG solve in ONE persistent launch.  The multi-launch loop above spends a PCG iteration on four small
// dependent launches (~25-30 us: each kernel ramps up, reads its operands from L2 / Infinity Cache and drains; section 6
// of DESIGN.md); here one workgroup per AGGREGATE stays resident for the whole solve and keeps everything that does not
// change between iterations on chip -- its rows' matrix items in registers (6 doubles per (entry, row) item, PK_KREG
// items per thread), its rows of the block-Jacobi inverses and of P, in LDS its six rows of the coarse inverse -- and its
// rows' x, r, p, q in registers as well, so an iteration only moves the vectors other workgroups need and synchronises
// the grid TWICE:
//
//   phase 1   (exchange A: r.z / r.r partials and the columns' z of the previous phase 2)  beta -- every workgroup sums
//             the same partials in the same order: identical decisions everywhere, convergence included;  columns
//             p = z + beta p_old into LDS;  products of the workgroup's items, row sums q = (A p)[rows] in entry order;
//             publishes its partial p.q and the RESTRICTED q, P^T q
//   phase 2   (exchange B: p.q partials and P^T q of every aggregate)  alpha;  x += alpha p, r -= alpha q;  z = M^-1 r
//             with the block-Jacobi part local and the coarse part from  r_c(new) = r_c(old) - alpha P^T q  for ALL
//             aggregates (the same vector in every workgroup, so M^-1 is one symmetric operator; r_c(old) is the exact
//             restriction its owner published one iteration ago, so the recurrence never runs longer than one step), its own
//             six rows of A_c^-1 r_c, prolonged to its rows;  publishes z, the exact restriction of its new residual and
//             the partials r.z, r.r
//
// An exchange is barrier and data in one: what other workgroups read travels through relaxed agent-scope atomic stores /
// loads (written through to, and read from, the device's coherence point -- no cache write-back / invalidate: an acquire
// at agent scope empties the XCD's L2, after which every operand of the next phase, register spills included, came from
// the fabric: tools/ubench_gridbar.hip, 6.2 us per counter barrier with a published value against 2.7 us without the
// cache maintenance), and the values a phase waits for live in slots that hold a SENTINEL (a NaN payload no computation
// produces) until their owner publishes them: a reader polls the slots themselves, one round trip when everybody is
// on time, instead of a counter barrier (136 serialised atomics) followed by the loads.  Slots are double-buffered by
// iteration parity and reset by their owner one exchange after everybody has read them (a workgroup that has published
// exchange j+1 has finished reading exchange j); values that are not polled (p, the exact r_c) are complete before the
// polled ones are stored (an explicit s_waitcnt vmcnt(0) in every wavefront, then a workgroup barrier), so whoever sees the
// latter can read the former.
//
// Every sum has a fixed order: the solve is bit-reproducible and identical on every rank of a sharded run.  A spin limit
// raises an abort flag instead of hanging.  Used when the graph fits (one workgroup per aggregate co-resident, LDS for
// its columns and items); otherwise the multi-launch loop runs.
// ------------------------------------------------------------------------------------------------------------------
constexpr int PG_COARSE_MAX = 6144;        // coarse unknowns (1 024 aggregates): the dense inverse is 302 MB and O(n_c^3) beyond that is no preconditioner
constexpr int PK_BLOCK = 512;
constexpr int PG_AGG_MAX = 85;             // members of an aggregate: its 6 x 85 rows have a thread each in the persistent kernel
constexpr int PK_ROWS = 6 * PG_AGG_MAX;
constexpr int PK_W = PK_BLOCK / 64;
constexpr int PK_KREG = 9;   // matrix items per thread kept in registers (more are streamed from memory)
constexpr int PK_CM = 3;     // column values per thread whose p_old is fetched ahead of the poll
constexpr unsigned PK_SPIN_LIMIT = 1u << 20;
constexpr unsigned PK_SENT32 = 0xFFF8DEADu;  // both halves of the sentinel double (hipMemsetD32)
constexpr long long PK_SENT = (long long)(((unsigned long long)PK_SENT32 << 32) | PK_SENT32);

struct PkArgs {
  const double *vals, *minv, *b;
  double *x, *p0, *p1;
  const int32_t *agg_ptr, *agg_mem;     // aggregate -> member vertices
  const int32_t *bent_ptr, *bent, *blc; // aggregate -> its rows' entries (global entry id, local column), member by member
  const int32_t *bmptr;                 // per aggregate nm + 1 offsets: member -> its first local entry
  const int32_t *bcol_ptr, *bcol;       // aggregate -> the vertices its entries' columns refer to
  const double *P, *Ainv;
  double *rc0, *rc1;                    // exact restricted residual, by iteration parity
  double *zA, *rzA, *rrA;               // exchange A slots: [2][n6], [2][n_agg], [2][n_agg]
  double *pqB, *qcB;                    // exchange B slots: [2][n_agg], [2][n_c]
  double *scal;
  unsigned *bar;                        // [0] arrivals of the set-up barrier  [1] abort flag
  size_t n_items;
  int n6, n_agg, n_c, coarse, max_iter, lds_cols, lds_items;
  int debug_abort;                      // tests: workgroup 0 raises the abort flag at this iteration (-1: never)
  double tol2;
};

#ifdef LSLAM_PK_CLOCKS  // profiling build: where an iteration of the persistent kernel spends its time (workgroup 0)
#define PK_T(i) { const unsigned long long _n = wall_clock64(); pk_t[i] += _n - pk_last; pk_last = _n; }
#else
#define PK_T(i)
#endif
PG_DEV double pk_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PG_DEV void pk_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PG_DEV bool pk_is_sent(double v) { return __double_as_longlong(v) == PK_SENT; }
PG_DEV double pk_sent() { return __longlong_as_double(PK_SENT); }
PG_DEV bool pk_aborted(unsigned *bar) { return __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
PG_DEV void pk_abort(unsigned *bar) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// counter barrier of the set-up (once per solve)
PG_DEV bool pk_grid_barrier(unsigned *bar, unsigned target, int *ok_lds) {
  __builtin_amdgcn_s_waitcnt(0);  // this wavefront's published values are at the coherence point
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      if ((spins & 1023u) == 0 && (spins > PK_SPIN_LIMIT || pk_aborted(bar))) {
        pk_abort(bar);
        good = 0;
        break;
      }
    }
    *ok_lds = good;
  }
  __syncthreads();
  return *ok_lds != 0;
}

// Sums of up to eight per-thread values over the workgroup, fixed order, every thread ends with the totals.  Inside a
// wavefront a butterfly that HALVES the values it carries at each of its first three steps: partners lane ^ 1 split the
// eight values (the even lane adds both lanes' first four, the odd lane the last four), lane ^ 2 split the four, lane ^ 4
// the two -- 4 + 2 + 1 additions instead of 3 x 8 -- then lane ^ 8, ^ 16, ^ 32 on the one value left: lane l ends with the
// wavefront's sum of value 4 (l & 1) + 2 (l >> 1 & 1) + (l >> 2 & 1).  The moves are DPP quad permutes, ds_swizzle and one
// ds_bpermute; the eight wavefronts' sums meet in LDS and are added by a second, three-step butterfly.
// (__shfl_xor on doubles is two ds_bpermute round trips per step and value: 3 us for seven values.)
template <int CTRL>
PG_DEV double pk_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int PATTERN>
PG_DEV double pk_swz(double v) {  // ds_swizzle, bit mode: lane ^ (PATTERN >> 10) inside each half of the wavefront
  return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN), __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN));
}
template <int M>
PG_DEV void pk_reduce(double (&v)[M], double *sh) {
  static_assert(M <= 8, "at most eight values");
  const int lane = threadIdx.x & 63;
  double t8[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) t8[m] = m < M ? v[m] : 0.0;
  double t4[4], t2[2], t;
  {
    const bool hi = (lane & 1) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) t4[i] = (hi ? t8[4 + i] : t8[i]) + pk_dpp<0xB1>(hi ? t8[i] : t8[4 + i]);  // quad_perm [1,0,3,2]
  }
  {
    const bool hi = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) t2[i] = (hi ? t4[2 + i] : t4[i]) + pk_dpp<0x4E>(hi ? t4[i] : t4[2 + i]);  // quad_perm [2,3,0,1]
  }
  {
    const bool hi = (lane & 4) != 0;
    t = (hi ? t2[1] : t2[0]) + pk_swz<0x101F>(hi ? t2[0] : t2[1]);  // lane ^ 4
  }
  t += pk_swz<0x201F>(t);  // lane ^ 8
  t += pk_swz<0x401F>(t);  // lane ^ 16
  t += __shfl_xor(t, 32);  // lane ^ 32
  if (lane < 8) sh[(4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1)) * PK_W + (threadIdx.x >> 6)] = t;
  __syncthreads();
  // lane l takes the sum of wavefront l & 7 for value l >> 3; three butterfly steps over the wavefront index leave every
  // lane of the group with the value's total, broadcast from lane 8 m
  static_assert(PK_W == 8, "eight wavefronts");
  double u = sh[lane];
  u += pk_dpp<0xB1>(u);
  u += pk_dpp<0x4E>(u);
  u += pk_swz<0x101F>(u);
#pragma unroll
  for (int m = 0; m < M; ++m)
    v[m] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(u), 8 * m), __builtin_amdgcn_readlane(__double2loint(u), 8 * m));
  __syncthreads();
}

__global__ __launch_bounds__(PK_BLOCK) void pg_pcg_persistent_kernel(PkArgs a) {
#pragma clang fp contract(fast)  // fused multiply-adds here (the library is built with -ffp-contract=off for the fp32 parity paths)
  extern __shared__ double pk_lds[];
  double *pf = pk_lds;                           // [lds_cols * 6] the columns' p
  double *dl = pf + (size_t)a.lds_cols * 6;      // [lds_items] products
  double *rloc = dl + a.lds_items;               // [PK_ROWS] residuals of the rows
  double *rcs = rloc + PK_ROWS;                  // [n_c] restricted residual
  double *ainv = rcs + a.n_c;                    // [6 * n_c] this aggregate's rows of the coarse inverse
  double *sh = ainv + 6 * (size_t)a.n_c;         // [8 * PK_W] wavefront sums
  __shared__ int bar_ok;
  const int ag = blockIdx.x, tid = threadIdx.x;
  const int m0 = a.agg_ptr[ag], nm = a.agg_ptr[ag + 1] - m0, nrow = nm * 6;
  const int e0 = a.bent_ptr[ag], nit = (a.bent_ptr[ag + 1] - e0) * 6;
  const int c0 = a.bcol_ptr[ag], ncol6 = (a.bcol_ptr[ag + 1] - c0) * 6;
  const bool coarse = a.coarse != 0;

  // ---- what stays on chip for the whole solve ----
  // a thread's items come in units of three consecutive rows of one entry (unit u = tid + PK_BLOCK ju holds items
  // 3 u .. 3 u + 2): the three share the entry's column, fetched from LDS once
  static_assert(PK_KREG % 3 == 0, "items in units of three rows");
  double A[PK_KREG][6];
  int lcol[PK_KREG];
#pragma unroll
  for (int j = 0; j < PK_KREG; ++j) {
    const int i = 3 * (tid + PK_BLOCK * (j / 3)) + j % 3;
    const bool on = i < nit;
    const int le = on ? i / 6 : 0, r = on ? i - le * 6 : 0;
    const int e = on ? a.bent[e0 + le] : 0;
    lcol[j] = on ? a.blc[e0 + le] * 6 : 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) A[j][c] = on ? a.vals[(size_t)c * a.n_items + (size_t)e * 6 + r] : 0.0;
  }
  const bool isrow = tid < nrow;
  const int jm = tid / 6, rr = tid - jm * 6;
  const int v = isrow ? a.agg_mem[m0 + jm] : 0;
  const int row = v * 6 + rr;
  double mi[6], Pr[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    mi[c] = isrow ? a.minv[(size_t)v * 36 + rr * 6 + c] : 0.0;
    Pr[c] = (isrow && coarse) ? a.P[(size_t)v * 36 + rr * 6 + c] : 0.0;
  }
  const int le_b = isrow ? a.bmptr[m0 + ag + jm] : 0, le_e = isrow ? a.bmptr[m0 + ag + jm + 1] : 0;
  size_t gcol[PK_CM];  // the global (vertex, component) of the column values this thread fetches
#pragma unroll
  for (int j = 0; j < PK_CM; ++j) {
    const int i = tid + PK_BLOCK * j;
    const int ci = i < ncol6 ? i / 6 : 0;
    gcol[j] = i < ncol6 ? (size_t)a.bcol[c0 + ci] * 6 + (i - ci * 6) : 0;
  }
  if (coarse)
    for (int i = tid; i < 6 * a.n_c; i += PK_BLOCK) ainv[i] = a.Ainv[(size_t)ag * 6 * a.n_c + i];
  // y = this aggregate's six rows of A_c^-1 times rcs (both in LDS); every thread ends with the six sums
  auto coarse_rows = [&](double (&s)[6]) {
#pragma unroll
    for (int m = 0; m < 6; ++m) s[m] = 0.0;
    for (int col = tid; col < a.n_c; col += PK_BLOCK) {
      const double rcv = rcs[col];
#pragma unroll
      for (int m = 0; m < 6; ++m) s[m] += ainv[m * a.n_c + col] * rcv;
    }
    pk_reduce<6>(s, sh);
  };

  // ---- x = 0, r = b, z = M^-1 r ----
  double x = 0.0, r = isrow ? a.b[row] : 0.0, q = 0.0, p_own = 0.0, z_own = 0.0;
  if (tid < PK_ROWS) rloc[tid] = r;
  __syncthreads();
  {
    double zj = 0.0;
    if (isrow) {
#pragma unroll
      for (int c = 0; c < 6; ++c) zj += mi[c] * rloc[jm * 6 + c];
    }
    z_own = zj;
  }
  if (coarse) {
    double w[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) w[m] = Pr[m] * r;
    pk_reduce<6>(w, sh);
    if (tid == 0) {
#pragma unroll
      for (int m = 0; m < 6; ++m) pk_st(a.rc0 + ag * 6 + m, w[m]);
    }
    if (!pk_grid_barrier(a.bar, gridDim.x, &bar_ok)) return;
    for (int col = tid; col < a.n_c; col += PK_BLOCK) rcs[col] = pk_ld(a.rc0 + col);
    __syncthreads();
    double y[6];
    coarse_rows(y);
#pragma unroll
    for (int m = 0; m < 6; ++m) z_own += Pr[m] * y[m];
  }
  {  // exchange A of iteration 0
    if (isrow) pk_st(a.zA + row, z_own);
    double o[2] = {r * z_own, r * r};
    pk_reduce<2>(o, sh);
    if (tid == 0) {
      pk_st(a.rzA + ag, o[0]);
      pk_st(a.rrA + ag, o[1]);
    }
  }

  double rz_old = 0.0, bb = 0.0, rr_last = 0.0;
  int k = 0, done = 0;
#ifdef LSLAM_PK_CLOCKS
  unsigned long long pk_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pk_last = wall_clock64();
#endif
  for (;;) {
    const int g = k & 1;
    // ---- phase 1: exchange A ----
    const double *po = g ? a.p0 : a.p1;  // p of iteration k - 1 (complete: published before exchange B of k - 1)
    double *pn = g ? a.p1 : a.p0;
    double ppv[PK_CM];
#pragma unroll
    for (int j = 0; j < PK_CM; ++j) ppv[j] = (k > 0 && tid + PK_BLOCK * j < ncol6) ? pk_ld(po + gcol[j]) : 0.0;
    double S[2];
    {
      const double *zg = a.zA + (size_t)g * a.n6, *rzg = a.rzA + g * a.n_agg, *rrg = a.rrA + g * a.n_agg;
      for (unsigned spins = 0;; ++spins) {
        int bad = 0;
        S[0] = S[1] = 0.0;
        if (tid < a.n_agg) {
          S[0] = pk_ld(rzg + tid);
          S[1] = pk_ld(rrg + tid);
          bad |= (pk_is_sent(S[0]) || pk_is_sent(S[1])) ? 1 : 0;
        }
#pragma unroll
        for (int j = 0; j < PK_CM; ++j) {
          const int i = tid + PK_BLOCK * j;
          if (i < ncol6) {
            const double zz = pk_ld(zg + gcol[j]);
            pf[i] = zz;
            bad |= pk_is_sent(zz) ? 1 : 0;
          }
        }
        for (int i = tid + PK_BLOCK * PK_CM; i < ncol6; i += PK_BLOCK) {
          const int ci = i / 6;
          const double zz = pk_ld(zg + (size_t)a.bcol[c0 + ci] * 6 + (i - ci * 6));
          pf[i] = zz;
          bad |= pk_is_sent(zz) ? 1 : 0;
        }
        if (!__syncthreads_or(bad)) break;
        if ((spins & 255u) == 255u && (spins > PK_SPIN_LIMIT || pk_aborted(a.bar))) {
          pk_abort(a.bar);
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (k > 0 && tid == 0) {  // everybody has read exchange B of iteration k - 1: its slots are free again
      double *pq = a.pqB + (g ^ 1) * a.n_agg, *qc = a.qcB + (size_t)(g ^ 1) * a.n_c;
      pk_st(pq + ag, pk_sent());
#pragma unroll
      for (int m = 0; m < 6; ++m) pk_st(qc + ag * 6 + m, pk_sent());
    }
    pk_reduce<2>(S, sh);
    PK_T(0)
    if (k == 0) bb = S[1];
    rr_last = S[1];
    if (S[1] <= a.tol2 * bb || !(S[0] > 0.0)) {
      done = 1;
      break;
    }
    if (k >= a.max_iter) break;
    if (k == a.debug_abort && ag == 0 && tid == 0) pk_abort(a.bar);
    const double beta = k > 0 ? S[0] / rz_old : 0.0;
#pragma unroll
    for (int j = 0; j < PK_CM; ++j) {
      const int i = tid + PK_BLOCK * j;
      if (i < ncol6) pf[i] += beta * ppv[j];
    }
    if (k > 0)
      for (int i = tid + PK_BLOCK * PK_CM; i < ncol6; i += PK_BLOCK) {
        const int ci = i / 6;
        pf[i] += beta * pk_ld(po + (size_t)a.bcol[c0 + ci] * 6 + (i - ci * 6));
      }
    if (isrow) {
      p_own = z_own + beta * p_own;
      pk_st(pn + row, p_own);
    }
    __syncthreads();
    PK_T(1)
#pragma unroll
    for (int ju = 0; ju < PK_KREG / 3; ++ju) {
      const int i0 = 3 * (tid + PK_BLOCK * ju);
      if (i0 < nit) {  // (the items of an entry come in sixes: a unit is whole or absent)
        double pc[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) pc[c] = pf[lcol[3 * ju] + c];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < 6; ++c) s += A[3 * ju + k][c] * pc[c];
          dl[i0 + k] = s;
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // one unit's operands in flight, not all of them (register pressure)
    }
    for (int i = 3 * PK_BLOCK * (PK_KREG / 3) + tid; i < nit; i += PK_BLOCK) {  // an aggregate with more items than the registers hold
      const int le = i / 6, rw = i - le * 6;
      const int e = a.bent[e0 + le], lc = a.blc[e0 + le] * 6;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) s += a.vals[(size_t)c * a.n_items + (size_t)e * 6 + rw] * pf[lc + c];
      dl[i] = s;
    }
    __syncthreads();
    PK_T(2)
    q = 0.0;
    for (int le = le_b; le < le_e; ++le) q += dl[le * 6 + rr];
    {
      double w[7];
#pragma unroll
      for (int m = 0; m < 6; ++m) w[m] = Pr[m] * q;
      w[6] = p_own * q;
      // Everything this workgroup stored without a sentinel -- every wavefront's p, the exact r_c and the sentinel resets of the
      // previous phase -- must be at the coherence point before the polled slots below are filled: whoever sees those may read
      // the former.  The compiler's own wait ahead of a workgroup barrier covers LDS only (s_waitcnt lgkmcnt(0)), so every
      // wavefront waits for its outstanding vector-memory operations explicitly (they were issued a phase ago: no stall).
      __builtin_amdgcn_s_waitcnt(0);
      pk_reduce<7>(w, sh);  // (its workgroup barrier: nobody fills a slot before every wavefront has waited)
      if (tid == 0) {
        double *qc = a.qcB + (size_t)g * a.n_c;
#pragma unroll
        for (int m = 0; m < 6; ++m) pk_st(qc + ag * 6 + m, w[m]);
        pk_st(a.pqB + g * a.n_agg + ag, w[6]);
      }
    }
    PK_T(3)
    // ---- phase 2: exchange B ----
    const double *rco = g ? a.rc1 : a.rc0;  // exact r_c entering this iteration (complete: published before exchange A)
    double rcv[2] = {0.0, 0.0}, qcv[2] = {0.0, 0.0};
    if (coarse) {
#pragma unroll
      for (int jc = 0; jc < 2; ++jc) {
        const int col = tid + PK_BLOCK * jc;
        if (col < a.n_c) rcv[jc] = pk_ld(rco + col);
      }
    }
    double Spq[1];
    {
      const double *pq = a.pqB + g * a.n_agg, *qc = a.qcB + (size_t)g * a.n_c;
      for (unsigned spins = 0;; ++spins) {
        int bad = 0;
        Spq[0] = 0.0;
        if (tid < a.n_agg) {
          Spq[0] = pk_ld(pq + tid);
          bad |= pk_is_sent(Spq[0]) ? 1 : 0;
        }
        if (coarse) {
#pragma unroll
          for (int jc = 0; jc < 2; ++jc) {
            const int col = tid + PK_BLOCK * jc;
            if (col < a.n_c) {
              qcv[jc] = pk_ld(qc + col);
              bad |= pk_is_sent(qcv[jc]) ? 1 : 0;
            }
          }
          for (int col = tid + 2 * PK_BLOCK; col < a.n_c; col += PK_BLOCK) {
            const double t = pk_ld(qc + col);
            rcs[col] = t;
            bad |= pk_is_sent(t) ? 1 : 0;
          }
        }
        if (!__syncthreads_or(bad)) break;
        if ((spins & 255u) == 255u && (spins > PK_SPIN_LIMIT || pk_aborted(a.bar))) {
          pk_abort(a.bar);
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    PK_T(4)
    {  // everybody has read exchange A of this iteration: its slots are free again
      double *zg = a.zA + (size_t)g * a.n6;
      if (isrow) pk_st(zg + row, pk_sent());
      if (tid == 0) {
        pk_st(a.rzA + g * a.n_agg + ag, pk_sent());
        pk_st(a.rrA + g * a.n_agg + ag, pk_sent());
      }
    }
    pk_reduce<1>(Spq, sh);
    PK_T(5)
    const double alpha = S[0] / Spq[0];
    x += alpha * p_own;
    r -= alpha * q;
    if (tid < PK_ROWS) rloc[tid] = r;
    if (coarse) {
#pragma unroll
      for (int jc = 0; jc < 2; ++jc) {
        const int col = tid + PK_BLOCK * jc;
        if (col < a.n_c) rcs[col] = rcv[jc] - alpha * qcv[jc];
      }
      for (int col = tid + 2 * PK_BLOCK; col < a.n_c; col += PK_BLOCK) rcs[col] = pk_ld(rco + col) - alpha * rcs[col];
    }
    __syncthreads();
    PK_T(6)
    {
      double zj = 0.0;
      if (isrow) {
#pragma unroll
        for (int c = 0; c < 6; ++c) zj += mi[c] * rloc[jm * 6 + c];
      }
      z_own = zj;
    }
    if (coarse) {
      double y[6];
      coarse_rows(y);
#pragma unroll
      for (int m = 0; m < 6; ++m) z_own += Pr[m] * y[m];
    }
    PK_T(7)
    // the sentinel resets of exchange A's slots (start of this phase) are complete before this workgroup publishes anything
    // of exchange A of the next iteration; the z stores below are polled themselves and stay in flight behind it
    __builtin_amdgcn_s_waitcnt(0);
    if (isrow) pk_st(a.zA + (size_t)(g ^ 1) * a.n6 + row, z_own);  // exchange A of iteration k + 1
    {
      double o[8];
#pragma unroll
      for (int m = 0; m < 6; ++m) o[m] = Pr[m] * r;
      o[6] = r * z_own;
      o[7] = r * r;
      pk_reduce<8>(o, sh);
      if (tid == 0) {
        double *rcn = g ? a.rc0 : a.rc1;
#pragma unroll
        for (int m = 0; m < 6; ++m) pk_st(rcn + ag * 6 + m, o[m]);
        // (the exact r_c is read after exchange B of the next iteration, which this workgroup publishes several workgroup
        // barriers from here: complete by then)
        pk_st(a.rzA + (g ^ 1) * a.n_agg + ag, o[6]);
        pk_st(a.rrA + (g ^ 1) * a.n_agg + ag, o[7]);
      }
    }
    rz_old = S[0];
    PK_T(8)
    ++k;
  }
  if (isrow) a.x[row] = x;
  if (ag == 0 && tid == 0) {
    a.scal[3] = rr_last;
    a.scal[4] = bb;
    a.scal[5] = done ? 1.0 : 0.0;
    a.scal[6] = (double)k;
#ifdef LSLAM_PK_CLOCKS
    for (int i = 0; i < 12; ++i) a.scal[8 + i] += (double)pk_t[i];  // 100 MHz ticks, accumulated over the solves
#endif
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The coarse inverse in ONE persistent launch.  The two-launch-per-pivot loop above (pgc_gj_pivot_kernel: one workgroup,
// a serial 6 x 6 inverse, 29 us; pgc_gj_update_kernel: 7 us) costs 136 x 36 us = 4.9 ms per inverse on the bench graph,
// a quarter of the whole LM run once the PCG iterations are fused.  Here workgroup i owns block row i of the matrix
// (6 x n_c doubles, in registers, GJ_CREG columns per thread) for the whole elimination.  At step k the pivot's owner
// inverts A_kk (36 lanes, in-place Gauss-Jordan in LDS; the pivot blocks of an SPD matrix need no pivoting), scales its
// row, R = A_kk^-1 A_k*, and publishes it -- with A_kk^-1 itself in the pivot's own columns -- into slot k of a buffer
// that holds a sentinel until then; every other workgroup polls that slot for its own columns (barrier and data in one
// round trip, as in pg_pcg_persistent_kernel) and updates its row:  A_i* <- A_i* - A_ik R  (its pivot columns: -A_ik A_kk^-1).
// Same arithmetic as the launch-per-pivot form up to the 6 x 6 inverse's pivoting.
// ------------------------------------------------------------------------------------------------------------------
constexpr int GJ_BLOCK = 512;
constexpr int GJ_CREG = 4;
struct GjArgs {
  int debug_abort; // tests: workgroup 0 raises the abort flag at this pivot step (-1: never)
  double *Ac;      // [n_c][n_c], inverted in place
  double *slots;   // [na][6][n_c], sentinel-filled
  unsigned *bar;   // [1] abort flag
  int n_c, na;
};
__global__ __launch_bounds__(GJ_BLOCK) void pgc_gj_persistent_kernel(GjArgs g) {
#pragma clang fp contract(fast)
  __shared__ double Cb[36];
  __shared__ double Mb[36];
  const int i = blockIdx.x, tid = threadIdx.x, n = g.n_c;
  double A[GJ_CREG][6];
#pragma unroll
  for (int jc = 0; jc < GJ_CREG; ++jc) {
    const int col = tid + GJ_BLOCK * jc;
#pragma unroll
    for (int r = 0; r < 6; ++r) A[jc][r] = col < n ? g.Ac[(size_t)(i * 6 + r) * n + col] : 0.0;
  }
  for (int k = 0; k < g.na; ++k) {
    const int k6 = k * 6;
    double *slot = g.slots + (size_t)k * 6 * n;
    if (k == g.debug_abort && i == 0 && tid == 0) pk_abort(g.bar);
    if (i == k) {
#pragma unroll
      for (int jc = 0; jc < GJ_CREG; ++jc) {
        const int col = tid + GJ_BLOCK * jc;
        if (col >= k6 && col < k6 + 6) {
#pragma unroll
          for (int r = 0; r < 6; ++r) Mb[r * 6 + (col - k6)] = A[jc][r];
        }
      }
      __syncthreads();
      if (tid < 64) {  // in-place Gauss-Jordan on the 6 x 6 block, one lane per element (LDS serves a wavefront's accesses in order)
        volatile double *M = Mb;
        const int r = tid / 6, c = tid - r * 6;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          double nv = 0.0;
          if (tid < 36) {
            const double inv = 1.0 / M[p * 6 + p];
            const double mpc = M[p * 6 + c], mrp = M[r * 6 + p], mrc = M[r * 6 + c];
            if (r == p) nv = (c == p) ? inv : mpc * inv;
            else nv = (c == p) ? -(mrp * inv) : mrc - mrp * (mpc * inv);
          }
          __builtin_amdgcn_wave_barrier();
          if (tid < 36) M[tid] = nv;
          __builtin_amdgcn_wave_barrier();
        }
      }
      __syncthreads();
#pragma unroll
      for (int jc = 0; jc < GJ_CREG; ++jc) {
        const int col = tid + GJ_BLOCK * jc;
        if (col < n) {
          const bool ink = col >= k6 && col < k6 + 6;
          double Rv[6];
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) t += Mb[r * 6 + q] * A[jc][q];
            Rv[r] = ink ? Mb[r * 6 + (col - k6)] : t;
          }
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            A[jc][r] = Rv[r];
            pk_st(slot + (size_t)r * n + col, Rv[r]);
          }
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int jc = 0; jc < GJ_CREG; ++jc) {
        const int col = tid + GJ_BLOCK * jc;
        if (col >= k6 && col < k6 + 6) {
#pragma unroll
          for (int r = 0; r < 6; ++r) Cb[r * 6 + (col - k6)] = A[jc][r];
        }
      }
      double Rv[GJ_CREG][6];
      for (unsigned spins = 0;; ++spins) {
        int bad = 0;
#pragma unroll
        for (int jc = 0; jc < GJ_CREG; ++jc) {
          const int col = tid + GJ_BLOCK * jc;
#pragma unroll
          for (int m = 0; m < 6; ++m) {
            Rv[jc][m] = col < n ? pk_ld(slot + (size_t)m * n + col) : 0.0;
            bad |= pk_is_sent(Rv[jc][m]) ? 1 : 0;
          }
        }
        if (!__syncthreads_or(bad)) break;  // (its barrier also makes Cb visible)
        if ((spins & 255u) == 255u && (spins > PK_SPIN_LIMIT || pk_aborted(g.bar))) {
          pk_abort(g.bar);
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int jc = 0; jc < GJ_CREG; ++jc) {
        const int col = tid + GJ_BLOCK * jc;
        const bool ink = col >= k6 && col < k6 + 6;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double t = 0.0;
#pragma unroll
          for (int m = 0; m < 6; ++m) t += Cb[r * 6 + m] * Rv[jc][m];
          A[jc][r] = (ink ? 0.0 : A[jc][r]) - t;
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int jc = 0; jc < GJ_CREG; ++jc) {
    const int col = tid + GJ_BLOCK * jc;
    if (col < n) {
#pragma unroll
      for (int r = 0; r < 6; ++r) g.Ac[(size_t)(i * 6 + r) * n + col] = A[jc][r];
    }
  }
}

// X <- X * fromVectorMQT(dx)
__global__ void pg_update_kernel(const double *src, const double *dx, int n_v, int fixed, double *dst) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_v) return;
  const Pose x = load_pose(src + 7 * v);
  double d[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = (v == fixed) ? 0.0 : dx[v * 6 + k];
  const double w2 = 1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  Q4 dq = {d[3], d[4], d[5], sqrt(w2 > 0 ? w2 : 0.0)};
  if (w2 < 0) dq = {0, 0, 0, 1};
  double R[9];
  qrotmat(x.q, R);
  const V3 t = mulR(R, {d[0], d[1], d[2]});
  Q4 q = qmul(x.q, dq);
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  double *o = dst + 7 * v;
  o[0] = x.t.x + t.x; o[1] = x.t.y + t.y; o[2] = x.t.z + t.z;
  o[3] = q.x / n; o[4] = q.y / n; o[5] = q.z / n; o[6] = q.w / n;
}

__global__ void pg_dot_scale_kernel(const double *dx, const double *b, int n6, double lambda, double *part) {
  __shared__ double sh[CG_BLOCK];
  const int row = blockIdx.x * CG_ROWS + threadIdx.x;
  const double v = (threadIdx.x < CG_ROWS && row < n6) ? dx[row] * (lambda * dx[row] + b[row]) : 0.0;
  const double s = block_sum(v, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void pg_maxdiag_kernel(const double *diag, int n_v, int fixed, double *out) {
  __shared__ double sh[256];
  double m = 0.0;
  for (int k = threadIdx.x; k < n_v * 6; k += 256) {
    const int v = k / 6, r = k % 6;
    if (v != fixed) m = fmax(m, diag[(size_t)v * 36 + r * 6 + r]);
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

thread_local std::string g_pg_err;
#define PG_TRY(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      char _b[512];                                                                      \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      g_pg_err = _b;                                                                     \
      return LSLAM_ERR_HIP;                                                              \
    }                                                                                    \
  } while (0)

template <typename T>
hipError_t dev_upload(T **d, const std::vector<T> &h) {
  hipError_t e = hipMalloc((void **)d, std::max<size_t>(1, h.size()) * sizeof(T));
  if (e != hipSuccess) return e;
  if (!h.empty()) e = hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

}  // namespace

struct lslam_pg {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_v = 0, n_e = 0, fixed = 0, n_off = 0, n_entries = 0;
  int e_begin = 0, e_end = 0;
  lslam_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  lslam_comm *comm = nullptr;  // the library's RCCL communicator (not owned); used when no callback is set
  bool sharded() const { return allreduce != nullptr || comm != nullptr; }
  // in-place sum over the ranks, ordered on the stream
  int reduce(double *buf, size_t count) {
    if (allreduce) {
      if (hipStreamSynchronize(stream) != hipSuccess) return LSLAM_ERR_HIP;
      allreduce(allreduce_user, buf, count);  // contract: complete when it returns
      return LSLAM_OK;
    }
    return lslam::comm_allreduce_f64(comm, buf, count, stream) == hipSuccess ? LSLAM_OK : LSLAM_ERR_COMM;
  }
  // the row-sharded solve's exchange as an all-gather of the owned segments (half the bytes of the zero-padded all-reduce):
  // through the communicator when there is one, else through a callback of its own type (lslam_pg_set_row_gather)
  lslam_allgatherv_fn gatherv = nullptr;
  void *gatherv_user = nullptr;
  int gatherv_rank = -1, gatherv_world = 0;
  int rs_gathered = 0;
  int rs_agree_v0 = -2, rs_agree_v1 = -2, rs_agree_world = -1;  // the setting the ranks last agreed on ...
  bool rs_agree_gather = false;                                  // ... and what they agreed: every rank's rows are the canonical partition
  // rank / world of whichever transport can gather; false: only the all-reduce is available
  bool gather_transport(int *rank, int *world) const {
    if (gatherv) { *rank = gatherv_rank; *world = gatherv_world; return true; }
    if (!allreduce && comm) { *rank = lslam::comm_rank(comm); *world = lslam::comm_world(comm); return true; }
    return false;
  }
  int gather(int n_lists, double *const *bufs, const int64_t *const *offs, int world) {
    if (gatherv) {
      if (hipStreamSynchronize(stream) != hipSuccess) return LSLAM_ERR_HIP;
      for (int l = 0; l < n_lists; ++l) gatherv(gatherv_user, bufs[l], offs[l], world);  // contract: complete when it returns
      return LSLAM_OK;
    }
    return lslam::comm_allgatherv_f64(comm, n_lists, bufs, offs, stream) == hipSuccess ? LSLAM_OK : LSLAM_ERR_COMM;
  }
  // graph
  double *d_poses = nullptr, *d_trial = nullptr, *d_meas = nullptr, *d_info = nullptr;
  int32_t *d_ij = nullptr;
  std::vector<int32_t> h_ij;
  std::vector<int32_t> edge_block;  // per edge: off-diagonal block id
  std::vector<int32_t> off_pairs;   // [n_off][2]
  // shard structures
  double *d_rec = nullptr, *d_chi = nullptr;
  int32_t *d_vptr = nullptr, *d_vadj = nullptr, *d_optr = nullptr, *d_oadj = nullptr;
  // system buffer [diag | off | b | chi2]; owned unless supplied by the caller
  double *d_sys = nullptr;
  bool own_sys = true;
  // solver
  int32_t *d_row_ptr = nullptr, *d_row_col = nullptr, *d_row_src = nullptr, *d_row_of = nullptr;
  double *d_vals = nullptr, *d_minv = nullptr;
  double *d_x = nullptr, *d_r = nullptr, *d_z = nullptr, *d_p = nullptr, *d_q = nullptr;
  double *d_part = nullptr, *d_scal = nullptr, *d_tmp = nullptr;
  int n_cg_blocks = 0;
  // second level of the preconditioner (rigid motions of graph aggregates of at most `agg` keyframes)
  std::vector<int32_t> h_row_of, h_row_col;
  int agg = 64, n_agg = 0, n_c = 0, n_cb = 0, n_cblk = 0, n_parts = 0;
  int32_t *d_agg_of = nullptr, *d_agg_ptr = nullptr, *d_agg_mem = nullptr;
  double *d_P = nullptr, *d_Ac = nullptr, *d_rc = nullptr, *d_yc = nullptr, *d_gj = nullptr;
  int32_t *d_cb_ptr = nullptr, *d_cb_ent = nullptr, *d_cb_ab = nullptr;
  // environment switches, read when the graph is created (lslam_pg_create) -- never while it is optimised
  bool env_no_reuse = false, env_persistent_off = false, env_debug = false;
  int env_max_cg = 20000;
  double env_tol = 1e-8;
  int coarse_mode = -1;     // -1 automatic (switched on by a solve that needed many iterations), 0 off, 1 on
  bool coarse_on = false;
  int coarse_solves = 0;    // damped systems solved with the second level (statistics)
  // The coarse inverse (with the P it was built from) is a preconditioner, not part of the answer: it is kept across
  // damped solves while it still works -- rebuilt when lambda has moved by more than 10x since it was built, or when the
  // previous solve that used it needed over 1.3x the iterations of the first solve after it was built.  The rule only
  // looks at lambda and iteration counts, which are identical on every rank of a sharded run.
  bool coarse_valid = false;
  double coarse_lambda = 0.0;
  int coarse_fresh_iters = 0, coarse_last_iters = 0;
  int coarse_setups = 0;
  // persistent PCG kernel (pg_pcg_persistent_kernel): per-aggregate entry / column lists, its buffers, whether the graph fits
  int32_t *d_bent_ptr = nullptr, *d_bent = nullptr, *d_blc = nullptr, *d_bmptr = nullptr, *d_bcol_ptr = nullptr, *d_bcol = nullptr;
  double *d_pk = nullptr;      // [rc0 | rc1] then the exchange slots of pg_pcg_persistent_kernel
  unsigned *d_bar = nullptr;
  int pk_lds_cols = 0, pk_lds_items = 0;
  size_t pk_lds_bytes = 0;
  double *d_gjslots = nullptr; // pgc_gj_persistent_kernel's pivot-row slots [n_agg][6][n_c]
  int gj_fit = -1;             // as pk_fit, for the coarse inverse
  int pk_fit = -1;             // -1 not decided yet, 0 the multi-launch loop, 1 the persistent kernel
  int fused_solves = 0, total_solves = 0, pk_timeouts = 0;
  bool fell_back = false;      // a persistent kernel gave way to its launch loop during the last solve() (sharded runs: every rank must follow)
  // row-sharded solve (lslam_pg_set_row_shard): this rank's vertex rows [row_v0, row_v1); -1: the solve is replicated
  int row_v0 = -1, row_v1 = -1;
  std::vector<int32_t> h_row_ptr;
  int rs_solves = 0;
  // [diag | off | b | chi2 | fallback flag | exchange area of the row-sharded solve: 6 n_v + 8 + two partial sums per rank]: the system part is all-reduced
  // per linearisation, chi2 + flag per trial, the exchange area per PCG iteration of a row-sharded solve
  size_t core_doubles() const { return (size_t)n_v * 36 + (size_t)n_off * 36 + (size_t)n_v * 6 + 2; }
  static constexpr int RS_MAX_WORLD = 64;  // ranks whose partial sums the exchange area has room for (gather mode)
  size_t sys_doubles() const { return core_doubles() + (size_t)n_v * 6 + 8 + 2 * RS_MAX_WORLD; }
  double *xchg() const { return d_sys + core_doubles(); }
  double *diag() const { return d_sys; }
  double *off() const { return d_sys + (size_t)n_v * 36; }
  double *b() const { return d_sys + (size_t)n_v * 36 + (size_t)n_off * 36; }
  double *chi() const { return b() + (size_t)n_v * 6; }
};

namespace {

int build_shard(lslam_pg *pg, int e_begin, int e_end) {
  for (void *p : {(void *)pg->d_rec, (void *)pg->d_chi, (void *)pg->d_vptr, (void *)pg->d_vadj,
                  (void *)pg->d_optr, (void *)pg->d_oadj})
    if (p) (void)hipFree(p);
  pg->d_rec = pg->d_chi = nullptr;
  pg->d_vptr = pg->d_vadj = pg->d_optr = pg->d_oadj = nullptr;
  pg->e_begin = e_begin;
  pg->e_end = e_end;
  const int ne = e_end - e_begin;
  std::vector<int32_t> vptr(pg->n_v + 1, 0), optr(pg->n_off + 1, 0);
  for (int e = e_begin; e < e_end; ++e) {
    vptr[pg->h_ij[2 * e] + 1]++;
    vptr[pg->h_ij[2 * e + 1] + 1]++;
    optr[pg->edge_block[e] + 1]++;
  }
  for (int v = 0; v < pg->n_v; ++v) vptr[v + 1] += vptr[v];
  for (int k = 0; k < pg->n_off; ++k) optr[k + 1] += optr[k];
  std::vector<int32_t> vadj(2 * (size_t)ne), oadj((size_t)ne), vc(vptr.begin(), vptr.end() - 1),
      oc(optr.begin(), optr.end() - 1);
  for (int e = e_begin; e < e_end; ++e) {  // edge order = summation order
    const int le = e - e_begin;
    vadj[vc[pg->h_ij[2 * e]]++] = (le << 1) | 0;
    vadj[vc[pg->h_ij[2 * e + 1]]++] = (le << 1) | 1;
    oadj[oc[pg->edge_block[e]]++] = le;
  }
  PG_TRY(hipMalloc((void **)&pg->d_rec, std::max<size_t>(1, (size_t)ne) * REC * sizeof(double)));
  PG_TRY(hipMalloc((void **)&pg->d_chi, std::max<size_t>(1, (size_t)ne) * sizeof(double)));
  PG_TRY(dev_upload(&pg->d_vptr, vptr));
  PG_TRY(dev_upload(&pg->d_vadj, vadj));
  PG_TRY(dev_upload(&pg->d_optr, optr));
  PG_TRY(dev_upload(&pg->d_oadj, oadj));
  return LSLAM_OK;
}

// system of this shard's edges into pg->d_sys, then the all-reduce over ranks
int linearize(lslam_pg *pg, const double *poses) {
  const int ne = pg->e_end - pg->e_begin;
  if (ne > 0)
    hipLaunchKernelGGL(pg_edge_kernel, dim3((ne + 127) / 128), dim3(128), 0, pg->stream, poses, pg->d_ij,
                       pg->d_meas, pg->d_info, pg->e_begin, pg->e_end, pg->fixed, pg->d_rec, pg->d_chi);
  hipLaunchKernelGGL(pg_assemble_vertex_kernel, dim3((pg->n_v * 42 + 255) / 256), dim3(256), 0, pg->stream,
                     pg->d_rec, pg->d_vptr, pg->d_vadj, pg->n_v, pg->sharded() ? -1 : pg->fixed,
                     pg->diag(), pg->b());
  if (pg->n_off > 0)
    hipLaunchKernelGGL(pg_assemble_off_kernel, dim3((pg->n_off * 36 + 255) / 256), dim3(256), 0, pg->stream,
                       pg->d_rec, pg->d_optr, pg->d_oadj, pg->n_off, pg->off());
  hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, pg->d_chi, ne, 1, pg->chi());
  PG_TRY(hipGetLastError());
  if (pg->sharded()) {
    const int rc = pg->reduce(pg->d_sys, pg->core_doubles());
    if (rc) return rc;
    // identity block of the fixed vertex after the sum over ranks
    static const double I[36] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0,
                                 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1};
    if (pg->fixed >= 0)
      PG_TRY(hipMemcpyAsync(pg->diag() + (size_t)pg->fixed * 36, I, 36 * sizeof(double), hipMemcpyHostToDevice, pg->stream));
  }
  return LSLAM_OK;
}

int eval_chi2(lslam_pg *pg, const double *poses, double *out) {
  // the chi2 slot of the system buffer is free again once the current chi2 has been read
  // back, and it lives in the buffer the all-reduce hook knows how to address
  const int ne = pg->e_end - pg->e_begin;
  if (ne > 0)
    hipLaunchKernelGGL(pg_chi2_kernel, dim3((ne + 127) / 128), dim3(128), 0, pg->stream, poses, pg->d_ij,
                       pg->d_meas, pg->d_info, pg->e_begin, pg->e_end, pg->d_chi);
  hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, pg->d_chi, ne, 1, pg->chi());
  PG_TRY(hipGetLastError());
  if (pg->sharded()) {
    // chi2 and, next to it, whether this rank's persistent kernels gave way to their launch loops in the solve before:
    // one all-reduce carries both (lslam_pg_optimize makes every rank follow a rank that fell back)
    const double flag = pg->fell_back ? 1.0 : 0.0;
    PG_TRY(hipMemcpyAsync(pg->chi() + 1, &flag, sizeof(double), hipMemcpyHostToDevice, pg->stream));
    PG_TRY(hipStreamSynchronize(pg->stream));  // `flag` is a local
    const int rc = pg->reduce(pg->chi(), 2);
    if (rc) return rc;
  }
  if (!out) return LSLAM_OK;  // the caller reads pg->chi() itself, behind more work on the stream
  PG_TRY(hipMemcpyAsync(out, pg->chi(), sizeof(double), hipMemcpyDeviceToHost, pg->stream));
  PG_TRY(hipStreamSynchronize(pg->stream));
  return LSLAM_OK;
}

// (H + lambda I) dx = b by block-Jacobi PCG; returns iterations
int solve(lslam_pg *pg, double lambda, int max_cg, double tol, int *iters_out) {
  const int n6 = pg->n_v * 6;
  const size_t n_items = (size_t)pg->n_entries * 6;
  hipLaunchKernelGGL(pg_expand_kernel, dim3((unsigned)((n_items * 6 + 255) / 256)), dim3(256), 0, pg->stream,
                     pg->d_sys, pg->d_row_src, pg->n_entries, lambda, pg->d_vals);
  hipLaunchKernelGGL(pg_precond_kernel, dim3((pg->n_v + 63) / 64), dim3(64), 0, pg->stream, pg->d_vals,
                     pg->d_row_ptr, pg->n_v, n_items, pg->d_minv);
  CgArgs a;
  a.vals = pg->d_vals;
  a.row_ptr = pg->d_row_ptr;
  a.row_col = pg->d_row_col;
  a.row_of = pg->d_row_of;
  a.minv = pg->d_minv;
  a.b = pg->b();
  a.x = pg->d_x; a.r = pg->d_r; a.z = pg->d_z; a.d = pg->d_q;
  a.p[0] = pg->d_p;
  a.p[1] = pg->d_p + n6;
  a.part_rz[0] = pg->d_part;
  a.part_rz[1] = pg->d_part + pg->n_parts;
  a.part_rr = pg->d_part + 2 * pg->n_parts;
  a.part_pq = pg->d_part + 3 * pg->n_parts;
  a.n_parts = pg->n_parts;
  PG_TRY(hipMemsetAsync(pg->d_part, 0, 3 * (size_t)pg->n_parts * sizeof(double), pg->stream));  // the coarse level's slots
  a.n_items = (int)n_items;
  a.n_pblocks = (int)((n_items + PROD_BLOCK - 1) / PROD_BLOCK);
  a.scal = pg->d_scal;
  a.n6 = n6;
  a.n_blocks = pg->n_cg_blocks;
  a.tol2 = tol * tol;
  const dim3 g(pg->n_cg_blocks), blk(CG_BLOCK);
  // second level (see pgc_P_kernel): on when forced, or -- automatic -- once a solve of this graph needed many iterations
  const bool coarse = pg->coarse_mode == 1 || (pg->coarse_mode < 0 && pg->coarse_on);
  bool fresh_inverse = false;
  CoarseArgs c{};
  if (coarse) {
    c.poses = pg->d_poses;
    c.P = pg->d_P; c.Ac = pg->d_Ac; c.rc = pg->d_rc; c.yc = pg->d_yc;
    c.Rbuf = pg->d_gj; c.Cbuf = pg->d_gj + (size_t)6 * pg->n_c; c.Bbuf = pg->d_gj + (size_t)12 * pg->n_c;
    c.cb_ptr = pg->d_cb_ptr; c.cb_ent = pg->d_cb_ent; c.cb_ab = pg->d_cb_ab;
    c.agg_of = pg->d_agg_of; c.agg_ptr = pg->d_agg_ptr; c.agg_mem = pg->d_agg_mem;
    c.n_v = pg->n_v; c.G = pg->agg; c.na = pg->n_agg; c.n_c = pg->n_c; c.n_cb = pg->n_cb; c.n_cblk = pg->n_cblk;
    const size_t nn = (size_t)c.n_c * c.n_c;
    const bool no_reuse = pg->env_no_reuse;  // A/B switch (LSLAM_PG_NO_REUSE, read when the graph was created)
    bool rebuild = !pg->coarse_valid || no_reuse;
    if (!rebuild) {
      const double ratio = lambda > pg->coarse_lambda ? lambda / pg->coarse_lambda : pg->coarse_lambda / lambda;
      rebuild = !(ratio <= 10.0) || (pg->coarse_fresh_iters > 0 && 10 * pg->coarse_last_iters > 13 * pg->coarse_fresh_iters);
    }
    fresh_inverse = rebuild;
    if (!pg->d_Ac) {  // first solve with the second level: its dense matrix and scratch
      PG_TRY(hipMalloc((void **)&pg->d_Ac, nn * sizeof(double)));
      PG_TRY(hipMalloc((void **)&pg->d_gj, ((size_t)12 * pg->n_c + 36) * sizeof(double)));
      c.Ac = pg->d_Ac;
      c.Rbuf = pg->d_gj; c.Cbuf = pg->d_gj + (size_t)6 * pg->n_c; c.Bbuf = pg->d_gj + (size_t)12 * pg->n_c;
    }
    if (rebuild) {
    hipLaunchKernelGGL(pgc_P_kernel, dim3((pg->n_v + 127) / 128), dim3(128), 0, pg->stream, c);
    PG_TRY(hipMemsetAsync(c.Ac, 0, nn * sizeof(double), pg->stream));
    hipLaunchKernelGGL(pgc_assemble_kernel, dim3(c.n_cb), dim3(64), 0, pg->stream, c, pg->d_vals, pg->d_row_of, pg->d_row_col, n_items);
    if (pg->gj_fit < 0) {  // the persistent inverse when a workgroup per aggregate is co-resident and the row fits its registers
      pg->gj_fit = 0;
      const bool off = pg->env_persistent_off;
      hipDeviceProp_t prop;
      int per_cu = 0;
      if (!off && c.n_c <= GJ_CREG * GJ_BLOCK && hipGetDeviceProperties(&prop, pg->device) == hipSuccess &&
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pgc_gj_persistent_kernel, GJ_BLOCK, 0) == hipSuccess &&
          (long)per_cu * prop.multiProcessorCount >= c.na &&
          hipMalloc((void **)&pg->d_gjslots, (size_t)c.na * 6 * c.n_c * sizeof(double)) == hipSuccess)
        pg->gj_fit = 1;
      (void)hipGetLastError();
    }
    if (pg->gj_fit == 1) {
      GjArgs gj;
      gj.Ac = c.Ac; gj.slots = pg->d_gjslots; gj.bar = pg->d_bar; gj.n_c = c.n_c; gj.na = c.na;
      {
        const char *da = lslam::debug_env("LSLAM_DEBUG_GJ_ABORT");  // test hook (LSLAM_DEBUG_HOOKS=1)
        gj.debug_abort = da ? std::atoi(da) : -1;
      }
      PG_TRY(hipMemsetAsync(pg->d_bar, 0, 2 * sizeof(unsigned), pg->stream));
      PG_TRY(hipMemsetD32Async((hipDeviceptr_t)pg->d_gjslots, (int)PK_SENT32, (size_t)c.na * 6 * c.n_c * 2, pg->stream));
      void *gargs[] = {(void *)&gj};
      unsigned gbar[2] = {0, 0};
      if (hipLaunchCooperativeKernel((const void *)pgc_gj_persistent_kernel, dim3((unsigned)c.na), dim3(GJ_BLOCK), gargs, 0, pg->stream) != hipSuccess) {
        // no cooperative launch on this device / partition mode, or the runtime's residency count differs from the occupancy
        // query: not an error of the solve -- the launch loop inverts A_c (which is untouched)
        (void)hipGetLastError();
        gbar[1] = 1;
      } else {
        PG_TRY(hipMemcpyAsync(gbar, pg->d_bar, sizeof(gbar), hipMemcpyDeviceToHost, pg->stream));
        PG_TRY(hipStreamSynchronize(pg->stream));
      }
      if (gbar[1] != 0) {  // not all workgroups resident at once (see the PCG kernel's fallback): the launch loop inverts A_c,
                           // assembled once more (a kernel that timed out wrote nothing back, but nothing here relies on that)
        if (pg->env_debug) fprintf(stderr, "[lslam pg] persistent Gauss-Jordan kernel timed out: launch loop from here on\n");
        pg->gj_fit = 0;
        pg->pk_timeouts++;
        pg->fell_back = true;
        PG_TRY(hipMemsetAsync(c.Ac, 0, nn * sizeof(double), pg->stream));
        hipLaunchKernelGGL(pgc_assemble_kernel, dim3(c.n_cb), dim3(64), 0, pg->stream, c, pg->d_vals, pg->d_row_of, pg->d_row_col, n_items);
      }
    }
    if (pg->gj_fit != 1) {
    for (int k = 0; k < c.na; ++k) {
      hipLaunchKernelGGL(pgc_gj_pivot_kernel, dim3(1), dim3(256), 0, pg->stream, c, k);
      hipLaunchKernelGGL(pgc_gj_update_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, pg->stream, c, k);
    }
    }
    PG_TRY(hipGetLastError());
    pg->coarse_valid = true;
    pg->coarse_lambda = lambda;
    pg->coarse_fresh_iters = 0;
    pg->coarse_setups++;
    }
    pg->coarse_solves++;
  }
  a.item_begin = 0;
  a.item_end = (int)n_items;
  a.block_begin = 0;
  // ---- row-sharded solve: several GPUs share ONE solve of a large graph (beyond "replicated solve") ----------
  // Block-Jacobi PCG only (the dense second level is a single-device structure), launch per step; see pg_rs_direction_kernel.
  if (pg->sharded() && pg->row_v0 >= 0 && !coarse) {
    const int v0 = pg->row_v0, v1 = pg->row_v1;
    const int r0 = v0 * 6, r1 = v1 * 6;
    a.item_begin = pg->h_row_ptr[(size_t)v0] * 6;
    a.item_end = pg->h_row_ptr[(size_t)v1] * 6;
    a.block_begin = r0 / CG_ROWS;
    const int nb_rows = (r1 - a.block_begin * CG_ROWS + CG_ROWS - 1) / CG_ROWS;   // row blocks of this rank
    const int nb_items = (a.item_end - a.item_begin + PROD_BLOCK - 1) / PROD_BLOCK;  // item blocks of this rank
    double *X = pg->xchg();  // [z (6 n_v) | rz | rr | pq | ... (8) | per rank: its part of rz, rr]
    double *z_own = a.z;     // init runs replicated on the ordinary z
    // The exchange: an all-gather of the owned segments when the transport has one and the ranks' rows are the canonical
    // partition (every rank can then name every segment without asking); else the zero-padded all-reduce.
    int g_rank = 0, g_world = 1;
    bool gather = pg->gather_transport(&g_rank, &g_world) && g_world >= 1 && g_world <= lslam_pg::RS_MAX_WORLD;
    std::vector<int64_t> z_offs, s_offs;
    {
      // Gather or all-reduce is ONE decision of all ranks: a rank that gathered (grouped broadcasts) while another all-reduced
      // would hang the solve.  lslam_pg_set_row_shard accepts any whole-block range and a rank may have no gather transport at
      // all, so no rank can tell from what it holds what the others will do: EVERY rank of EVERY row-sharded solve sums a
      // "not from me" flag through the linearisation's transport -- 1 without a gather transport or with a range that is not
      // the canonical one -- and the ranks gather only if nobody raised it.  (Cached per rank this was itself a collective
      // that some ranks could skip: a rank whose setting had not changed did not enter it.  One scalar all-reduce per solve of
      // ~100 iterations with two collectives each costs nothing.)
      int cb = -1, ce = -1;
      if (gather) lslam_pg_row_shard_range(pg->n_v, g_rank, g_world, &cb, &ce);
      double flag = (gather && cb == v0 && ce == v1) ? 0.0 : 1.0;
      PG_TRY(hipMemcpyAsync(X + n6 + 7, &flag, sizeof(double), hipMemcpyHostToDevice, pg->stream));
      PG_TRY(hipStreamSynchronize(pg->stream));  // (flag is a local)
      int rc_a = pg->reduce(X + n6 + 7, 1);
      if (rc_a) return rc_a;
      PG_TRY(hipMemcpyAsync(&flag, X + n6 + 7, sizeof(double), hipMemcpyDeviceToHost, pg->stream));
      PG_TRY(hipStreamSynchronize(pg->stream));
      pg->rs_agree_gather = flag == 0.0;
    }
    if (gather) {
      gather = pg->rs_agree_gather;
      z_offs.resize((size_t)g_world + 1);
      s_offs.resize((size_t)g_world + 1);
      for (int r = 0; r <= g_world; ++r) {
        int b = pg->n_v, e = pg->n_v;
        if (r < g_world) lslam_pg_row_shard_range(pg->n_v, r, g_world, &b, &e);
        z_offs[(size_t)r] = (int64_t)b * 6;
        s_offs[(size_t)r] = 2 * (int64_t)r;
      }
    }
    double *tail = X + n6 + 8;
    double *const g_bufs[2] = {X, tail};
    const int64_t *const g_offs[2] = {z_offs.data(), s_offs.data()};
    // x = 0, r = b, z = M^-1 b and their sums, on every row by every rank: b is complete everywhere after the linearisation's
    // all-reduce, so this needs no exchange
    hipLaunchKernelGGL(pg_cg_init_kernel, g, blk, 0, pg->stream, a);
    hipLaunchKernelGGL(pg_cg_init2_kernel, dim3(1), blk, 0, pg->stream, a);
    hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, a.part_rz[0], pg->n_cg_blocks, 1, X + n6);
    hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, a.part_rr, pg->n_cg_blocks, 1, X + n6 + 1);
    hipLaunchKernelGGL(pg_rs_copy_kernel, dim3(1), dim3(64), 0, pg->stream, (const double *)(X + n6), a.part_rz[0], a.part_rr, (double *)nullptr);
    PG_TRY(hipMemcpyAsync(X, z_own, (size_t)n6 * sizeof(double), hipMemcpyDeviceToDevice, pg->stream));
    a.z = X;  // from here on z lives in the exchange area: own rows written by the update, the others arrive by the all-reduce
    a.n_parts = 1;
    a.n_pblocks = 1;
    const dim3 gall((n6 + 255) / 256), b256(256);
    double scal[8] = {0};
    int done_iters = 0;
    for (int it = 0; it < max_cg;) {
      const int chunk = std::min(50, max_cg - it);
      for (int k = it; k < it + chunk; ++k) {
        hipLaunchKernelGGL(pg_rs_direction_kernel, gall, b256, 0, pg->stream, a, k);
        a.n_pblocks = 1;
        hipLaunchKernelGGL(pg_cg_prod_kernel, dim3(std::max(nb_items, 1)), dim3(PROD_BLOCK), 0, pg->stream, a, k);
        hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, a.part_pq, nb_items, 1, X + n6 + 2);
        PG_TRY(hipGetLastError());
        int rc = pg->reduce(X + n6 + 2, 1);  // p . A p
        if (rc) return rc;
        hipLaunchKernelGGL(pg_rs_copy_kernel, dim3(1), dim3(64), 0, pg->stream, (const double *)(X + n6 + 2), a.part_pq, (double *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL(pg_cg_update_kernel, dim3(std::max(nb_rows, 1)), blk, 0, pg->stream, a, k);
        if (gather) {  // own rows of z are in place; own parts of r . z and r . r go to this rank's slot behind them
          hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, a.part_rz[(k + 1) & 1], nb_rows, 1, tail + 2 * g_rank);
          hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, a.part_rr, nb_rows, 1, tail + 2 * g_rank + 1);
          PG_TRY(hipGetLastError());
          rc = pg->gather(2, g_bufs, g_offs, g_world);
          if (rc) return rc;
          hipLaunchKernelGGL(pg_rs_tail_sum_kernel, dim3(1), dim3(64), 0, pg->stream, X + n6, (const double *)tail, g_world);
        } else {
          hipLaunchKernelGGL(pg_rs_zero_others_kernel, gall, b256, 0, pg->stream, X, n6, r0, r1);
          hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, a.part_rz[(k + 1) & 1], nb_rows, 1, X + n6);
          hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, a.part_rr, nb_rows, 1, X + n6 + 1);
          PG_TRY(hipGetLastError());
          rc = pg->reduce(X, (size_t)n6 + 2);  // z gathered, r . z and r . r summed
          if (rc) return rc;
        }
        hipLaunchKernelGGL(pg_rs_copy_kernel, dim3(1), dim3(64), 0, pg->stream, (const double *)(X + n6), a.part_rz[(k + 1) & 1], a.part_rr, (double *)nullptr);
      }
      it += chunk;
      hipLaunchKernelGGL(pg_cg_check_kernel, dim3(1), blk, 0, pg->stream, a, it);
      PG_TRY(hipMemcpyAsync(scal, pg->d_scal, sizeof(scal), hipMemcpyDeviceToHost, pg->stream));
      PG_TRY(hipStreamSynchronize(pg->stream));
      done_iters = (int)scal[6];
      if (scal[5] != 0.0) break;
    }
    // the solution: every rank's rows, gathered the same way
    hipLaunchKernelGGL(pg_rs_pack_kernel, gall, b256, 0, pg->stream, (const double *)pg->d_x, X, n6, r0, r1);
    PG_TRY(hipGetLastError());
    int rc = gather ? pg->gather(1, g_bufs, g_offs, g_world) : pg->reduce(X, (size_t)n6);
    if (rc) return rc;
    PG_TRY(hipMemcpyAsync(pg->d_x, X, (size_t)n6 * sizeof(double), hipMemcpyDeviceToDevice, pg->stream));
    *iters_out = done_iters;
    pg->total_solves++;
    pg->rs_solves++;
    pg->rs_gathered += gather ? 1 : 0;
    return LSLAM_OK;
  }
  // The persistent kernel when the graph fits: one workgroup per aggregate, all co-resident, LDS for its columns / items.
  if (pg->pk_fit < 0) {
    pg->pk_fit = 0;
    const bool off = pg->env_persistent_off;
    hipDeviceProp_t prop;
    int per_cu = 0;
    if (!off && pg->n_agg <= PK_BLOCK /* one thread per aggregate sums the partials */ && pg->pk_lds_bytes <= 150 * 1024 &&
        hipGetDeviceProperties(&prop, pg->device) == hipSuccess &&
        hipFuncSetAttribute((const void *)pg_pcg_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)pg->pk_lds_bytes) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pg_pcg_persistent_kernel, PK_BLOCK, pg->pk_lds_bytes) == hipSuccess &&
        (long)per_cu * prop.multiProcessorCount >= pg->n_agg)
      pg->pk_fit = 1;
    (void)hipGetLastError();
  }
  pg->total_solves++;
  if (pg->pk_fit == 1) {
    PkArgs k;
    k.vals = pg->d_vals; k.minv = pg->d_minv; k.b = pg->b();
    k.x = pg->d_x; k.p0 = pg->d_p; k.p1 = pg->d_p + n6;
    k.agg_ptr = pg->d_agg_ptr; k.agg_mem = pg->d_agg_mem;
    k.bent_ptr = pg->d_bent_ptr; k.bent = pg->d_bent; k.blc = pg->d_blc; k.bmptr = pg->d_bmptr;
    k.bcol_ptr = pg->d_bcol_ptr; k.bcol = pg->d_bcol;
    k.P = pg->d_P; k.Ainv = pg->d_Ac;
    k.rc0 = pg->d_pk; k.rc1 = pg->d_pk + pg->n_c;
    double *slots = pg->d_pk + 2 * (size_t)pg->n_c;
    const size_t n_slots = 2 * (size_t)n6 + 6 * (size_t)pg->n_agg + 2 * (size_t)pg->n_c;
    k.zA = slots; k.rzA = k.zA + 2 * (size_t)n6; k.rrA = k.rzA + 2 * pg->n_agg; k.pqB = k.rrA + 2 * pg->n_agg;
    k.qcB = k.pqB + 2 * pg->n_agg;
    k.scal = pg->d_scal; k.bar = pg->d_bar;
    k.n_items = n_items;
    k.n6 = n6; k.n_agg = pg->n_agg; k.n_c = pg->n_c; k.coarse = coarse ? 1 : 0; k.max_iter = max_cg;
    k.lds_cols = pg->pk_lds_cols; k.lds_items = pg->pk_lds_items;
    k.tol2 = tol * tol;
    {
      const char *da = lslam::debug_env("LSLAM_DEBUG_PG_ABORT");  // test hook (LSLAM_DEBUG_HOOKS=1)
      k.debug_abort = da ? std::atoi(da) : -1;
    }
    PG_TRY(hipMemsetAsync(pg->d_bar, 0, 2 * sizeof(unsigned), pg->stream));
    PG_TRY(hipMemsetD32Async((hipDeviceptr_t)slots, (int)PK_SENT32, 2 * n_slots, pg->stream));
    void *kargs[] = {(void *)&k};
    double scal[8] = {0};
    unsigned bar[2] = {0, 0};
    if (hipLaunchCooperativeKernel((const void *)pg_pcg_persistent_kernel, dim3((unsigned)pg->n_agg), dim3(PK_BLOCK), kargs,
                                   (unsigned)pg->pk_lds_bytes, pg->stream) != hipSuccess) {
      (void)hipGetLastError();  // as for the coarse inverse: a refused cooperative launch means the launch loop, not a failed solve
      bar[1] = 1;
    } else {
      PG_TRY(hipMemcpyAsync(scal, pg->d_scal, sizeof(scal), hipMemcpyDeviceToHost, pg->stream));
      PG_TRY(hipMemcpyAsync(bar, pg->d_bar, sizeof(bar), hipMemcpyDeviceToHost, pg->stream));
      PG_TRY(hipStreamSynchronize(pg->stream));
    }
    if (bar[1] == 0) {
      const int done_iters = (int)scal[6];
      *iters_out = done_iters;
      pg->fused_solves++;
      if (done_iters > 300) pg->coarse_on = true;
      if (coarse) {
        if (fresh_inverse) pg->coarse_fresh_iters = done_iters;
        pg->coarse_last_iters = done_iters;
      }
      return LSLAM_OK;
    }
    // An exchange ran into its spin limit: the workgroups were not all resident at once (another process's persistent
    // kernel on the same device can do that -- cooperative launches are not coordinated across processes).  Nothing was
    // written that the launch-per-step loop below reads; this graph stays on that loop from here on.
    if (pg->env_debug) fprintf(stderr, "[lslam pg] persistent PCG kernel timed out in a grid exchange: launch loop from here on\n");
    pg->pk_fit = 0;
    pg->pk_timeouts++;
    pg->fell_back = true;
  }
  auto coarse_correct = [&](int k) {  // z += P A_c^-1 P^T r and the matching share of r.z, entering iteration k + 1
    if (!coarse) return;
    hipLaunchKernelGGL(pgc_restrict_kernel, dim3(c.na), dim3(64), 0, pg->stream, c, (const double *)a.r);
    hipLaunchKernelGGL(pgc_mvp_kernel, dim3(c.na), dim3(256), 0, pg->stream, c, a.z, a.part_rz[(k + 1) & 1] + pg->n_cg_blocks);
  };
  hipLaunchKernelGGL(pg_cg_init_kernel, g, blk, 0, pg->stream, a);
  hipLaunchKernelGGL(pg_cg_init2_kernel, dim3(1), blk, 0, pg->stream, a);
  coarse_correct(-1);
  double scal[8] = {0};
  int done_iters = 0;
  for (int it = 0; it < max_cg;) {
    const int chunk = std::min(50, max_cg - it);
    for (int k = it; k < it + chunk; ++k) {
      hipLaunchKernelGGL(pg_cg_prod_kernel, dim3(a.n_pblocks), dim3(PROD_BLOCK), 0, pg->stream, a, k);
      hipLaunchKernelGGL(pg_cg_update_kernel, g, blk, 0, pg->stream, a, k);
      coarse_correct(k);
    }
    it += chunk;
    hipLaunchKernelGGL(pg_cg_check_kernel, dim3(1), blk, 0, pg->stream, a, it);
    PG_TRY(hipMemcpyAsync(scal, pg->d_scal, sizeof(scal), hipMemcpyDeviceToHost, pg->stream));
    PG_TRY(hipStreamSynchronize(pg->stream));
    done_iters = (int)scal[6];
    if (scal[5] != 0.0) break;
  }
  PG_TRY(hipGetLastError());
  *iters_out = done_iters;
  if (done_iters > 300) pg->coarse_on = true;  // ill-conditioned from here on: later solves of this graph take the second level
  if (coarse) {
    if (fresh_inverse) pg->coarse_fresh_iters = done_iters;
    pg->coarse_last_iters = done_iters;
  }
  return LSLAM_OK;
}

}  // namespace

extern "C" {

const char *lslam_pg_last_error(void) { return g_pg_err.c_str(); }

int lslam_pg_create(int device, int32_t n_v, const double *poses7, int32_t n_e, const int32_t *ij,
                    const double *meas7, const double *info36, int32_t fixed_vertex, lslam_pg **out) {
  if (!out || n_v <= 0 || n_e < 0 || !poses7 || (n_e && (!ij || !meas7 || !info36)) || fixed_vertex >= n_v) {
    g_pg_err = "bad pose-graph arguments";
    return LSLAM_ERR_INVALID;
  }
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    g_pg_err = "no HIP device available; this backend has no CPU fallback";
    return LSLAM_ERR_HIP;
  }
  PG_TRY(hipSetDevice(device));
  for (int e = 0; e < n_e; ++e)
    if (ij[2 * e] < 0 || ij[2 * e] >= n_v || ij[2 * e + 1] < 0 || ij[2 * e + 1] >= n_v || ij[2 * e] == ij[2 * e + 1]) {
      g_pg_err = "edge endpoint out of range";
      return LSLAM_ERR_INVALID;
    }
  lslam_pg *pg = new lslam_pg();
  struct Guard {  // a failure below frees what was built so far
    lslam_pg *p;
    ~Guard() { if (p) lslam_pg_destroy(p); }
  } guard{pg};
  pg->device = device;
  pg->n_v = n_v;
  pg->n_e = n_e;
  (void)lslam::debug_env("LSLAM_DEBUG_HOOKS");  // (the process-wide snapshot of the environment is taken no later than here)
  pg->env_no_reuse = std::getenv("LSLAM_PG_NO_REUSE") != nullptr;
  pg->env_persistent_off = std::getenv("LSLAM_PG_PERSISTENT") && std::atoi(std::getenv("LSLAM_PG_PERSISTENT")) == 0;
  pg->env_debug = std::getenv("LSLAM_DEBUG") != nullptr;
  if (const char *v = std::getenv("LSLAM_PG_MAX_CG")) pg->env_max_cg = std::atoi(v);
  if (const char *v = std::getenv("LSLAM_PG_TOL")) pg->env_tol = std::atof(v);
  pg->fixed = fixed_vertex;
  PG_TRY(hipStreamCreateWithFlags(&pg->stream, hipStreamNonBlocking));
  pg->h_ij.assign(ij, ij + 2 * (size_t)n_e);
  // off-diagonal block ids: distinct (min,max) pairs in first-appearance order of a sorted map
  std::map<std::pair<int, int>, int> ids;
  for (int e = 0; e < n_e; ++e) ids.emplace(std::minmax(ij[2 * e], ij[2 * e + 1]), 0);
  int nid = 0;
  for (auto &kv : ids) kv.second = nid++;
  pg->n_off = nid;
  pg->off_pairs.resize(2 * (size_t)nid);
  for (auto &kv : ids) {
    pg->off_pairs[2 * kv.second] = kv.first.first;
    pg->off_pairs[2 * kv.second + 1] = kv.first.second;
  }
  pg->edge_block.resize(n_e);
  for (int e = 0; e < n_e; ++e) pg->edge_block[e] = ids[std::minmax(ij[2 * e], ij[2 * e + 1])];
  // solver block-CSR: row v = [diag, then neighbours in block-id order]
  std::vector<std::vector<std::pair<int, int>>> rows(n_v);  // (col, code)
  for (int v = 0; v < n_v; ++v) rows[v].push_back({v, (v << 2) | 1});
  for (int k = 0; k < nid; ++k) {
    const int a = pg->off_pairs[2 * k], b = pg->off_pairs[2 * k + 1];
    rows[a].push_back({b, ((n_v + k) << 2) | 0});
    rows[b].push_back({a, ((n_v + k) << 2) | 2});
  }
  std::vector<int32_t> rptr(n_v + 1, 0), rcol, rsrc, rof;
  for (int v = 0; v < n_v; ++v) {
    rptr[v + 1] = rptr[v] + (int)rows[v].size();
    for (auto &pr : rows[v]) {
      rcol.push_back(pr.first);
      rsrc.push_back(pr.second);
      rof.push_back(v);
    }
  }
  pg->n_entries = (int)rcol.size();
  pg->h_row_ptr = rptr;
  pg->h_row_of = rof;
  pg->h_row_col = rcol;
  {  // coarse blocks: fine entries grouped by (aggregate of the row, aggregate of the column), in entry order
    if (const char *g = std::getenv("LSLAM_PG_AGG")) pg->agg = std::max(2, std::min(PG_AGG_MAX, std::atoi(g)));
    if (const char *m = std::getenv("LSLAM_PG_COARSE")) pg->coarse_mode = std::atoi(m);
    const int G = pg->agg;
    // aggregates: breadth-first from the lowest unassigned vertex over the solver rows (neighbours in block-id order,
    // i.e. deterministic), at most G members each
    std::vector<int32_t> agg_of((size_t)n_v, -1), agg_ptr(1, 0), agg_mem;
    const int Gg = std::getenv("LSLAM_PG_NO_MERGE") ? G : std::max(2, G - G / 8);  // growth limit (the merge below fills up to G)
    for (int seed = 0; seed < n_v; ++seed) {
      if (agg_of[(size_t)seed] >= 0) continue;
      const int a = (int)agg_ptr.size() - 1;
      const size_t first = agg_mem.size();
      agg_of[(size_t)seed] = a;
      agg_mem.push_back(seed);
      for (size_t head = first; head < agg_mem.size() && (int)(agg_mem.size() - first) < Gg; ++head) {
        const int u = agg_mem[head];
        for (int e = rptr[(size_t)u]; e < rptr[(size_t)u + 1] && (int)(agg_mem.size() - first) < Gg; ++e) {
          const int v = rcol[(size_t)e];
          if (agg_of[(size_t)v] < 0) {
            agg_of[(size_t)v] = a;
            agg_mem.push_back(v);
          }
        }
      }
      agg_ptr.push_back((int32_t)agg_mem.size());
    }
    // Breadth-first growth leaves pockets: a handful of keyframes cut off between two full aggregates (59 of the bench
    // graph's 136 aggregates had fewer than eight members).  Each would be a workgroup of the persistent solver, a
    // participant of every grid exchange and six coarse unknowns of little use: a pocket joins the smallest adjacent
    // aggregate that has room for it (the growth above stops at 7/8 of the limit to leave that room).
    if (!std::getenv("LSLAM_PG_NO_MERGE")) {
      const int n0 = (int)agg_ptr.size() - 1;
      std::vector<int> size(n0), into(n0);
      for (int a = 0; a < n0; ++a) { size[a] = agg_ptr[(size_t)a + 1] - agg_ptr[(size_t)a]; into[a] = a; }
      for (int a = 0; a < n0; ++a) {
        if (size[a] > std::max(1, G / 8)) continue;
        int best = -1;
        for (int m = agg_ptr[(size_t)a]; m < agg_ptr[(size_t)a + 1]; ++m) {
          const int u = agg_mem[(size_t)m];
          for (int e = rptr[(size_t)u]; e < rptr[(size_t)u + 1]; ++e) {
            int b = agg_of[(size_t)rcol[(size_t)e]];
            while (into[b] != b) b = into[b];
            if (b != a && size[b] + size[a] <= G && (best < 0 || size[b] < size[best] || (size[b] == size[best] && b < best))) best = b;
          }
        }
        if (best >= 0) { into[a] = best; size[best] += size[a]; size[a] = 0; }
      }
      std::vector<int> newid(n0, -1);
      int nn = 0;
      for (int a = 0; a < n0; ++a) if (into[a] == a) newid[a] = nn++;
      std::vector<std::vector<int32_t>> mem((size_t)nn);
      for (int a = 0; a < n0; ++a) {  // the receiving aggregate's members first (its first member stays its origin), pockets after them in order
        int b = a;
        while (into[b] != b) b = into[b];
        if (b == a)
          for (int m = agg_ptr[(size_t)a]; m < agg_ptr[(size_t)a + 1]; ++m) mem[(size_t)newid[a]].push_back(agg_mem[(size_t)m]);
      }
      for (int a = 0; a < n0; ++a) {
        int b = a;
        while (into[b] != b) b = into[b];
        if (b != a)
          for (int m = agg_ptr[(size_t)a]; m < agg_ptr[(size_t)a + 1]; ++m) mem[(size_t)newid[b]].push_back(agg_mem[(size_t)m]);
      }
      agg_ptr.assign(1, 0);
      agg_mem.clear();
      for (int a = 0; a < nn; ++a) {
        for (int32_t u : mem[(size_t)a]) { agg_of[(size_t)u] = a; agg_mem.push_back(u); }
        agg_ptr.push_back((int32_t)agg_mem.size());
      }
    }
    pg->n_agg = (int)agg_ptr.size() - 1;
    pg->n_c = 6 * pg->n_agg;
    pg->n_cblk = pg->n_agg;  // partial-sum slots of the coarse level: one per aggregate (pgc_mvp_kernel)
    PG_TRY(dev_upload(&pg->d_agg_of, agg_of));
    PG_TRY(dev_upload(&pg->d_agg_ptr, agg_ptr));
    PG_TRY(dev_upload(&pg->d_agg_mem, agg_mem));
    std::map<std::pair<int, int>, std::vector<int32_t>> cb;
    for (int e = 0; e < pg->n_entries; ++e) cb[{agg_of[(size_t)rof[(size_t)e]], agg_of[(size_t)rcol[(size_t)e]]}].push_back(e);
    std::vector<int32_t> cptr(1, 0), cent, cab;
    for (auto &kv : cb) {
      cab.push_back(kv.first.first);
      cab.push_back(kv.first.second);
      cent.insert(cent.end(), kv.second.begin(), kv.second.end());
      cptr.push_back((int32_t)cent.size());
    }
    pg->n_cb = (int)cb.size();
    PG_TRY(dev_upload(&pg->d_cb_ptr, cptr));
    PG_TRY(dev_upload(&pg->d_cb_ent, cent));
    PG_TRY(dev_upload(&pg->d_cb_ab, cab));
    PG_TRY(hipMalloc((void **)&pg->d_P, (size_t)n_v * 36 * sizeof(double)));
    PG_TRY(hipMalloc((void **)&pg->d_rc, (size_t)pg->n_c * sizeof(double)));
    PG_TRY(hipMalloc((void **)&pg->d_yc, (size_t)pg->n_c * sizeof(double)));
    // the dense coarse matrix (n_c^2 doubles) and the Gauss-Jordan scratch are allocated by the first solve that takes the
    // second level (solve()); graphs whose coarse matrix would exceed PG_COARSE_MAX unknowns never take it
    if (pg->n_c > PG_COARSE_MAX) {
      if (std::getenv("LSLAM_DEBUG"))
        fprintf(stderr, "[lslam pg] %d coarse unknowns > %d: the dense second level is not used for this graph (block-Jacobi PCG)\n", pg->n_c, PG_COARSE_MAX);
      pg->coarse_mode = 0;
    }
    {  // the persistent PCG kernel's view: per aggregate its rows' entries member by member, and a local numbering of
       // the column vertices those entries refer to (first appearance order)
      std::vector<int32_t> bent_ptr(1, 0), bent, blc, bmptr, bcol_ptr(1, 0), bcol;
      std::vector<int32_t> local((size_t)n_v, -1);
      int max_cols = 0, max_items = 0;
      for (int a = 0; a < pg->n_agg; ++a) {
        const size_t cfirst = bcol.size();
        const size_t efirst = bent.size();
        for (int m = agg_ptr[(size_t)a]; m < agg_ptr[(size_t)a + 1]; ++m) {
          const int u = agg_mem[(size_t)m];
          bmptr.push_back((int32_t)(bent.size() - efirst));
          for (int e = rptr[(size_t)u]; e < rptr[(size_t)u + 1]; ++e) {
            const int cv = rcol[(size_t)e];
            if (local[(size_t)cv] < 0) {
              local[(size_t)cv] = (int32_t)(bcol.size() - cfirst);
              bcol.push_back(cv);
            }
            bent.push_back(e);
            blc.push_back(local[(size_t)cv]);
          }
        }
        bmptr.push_back((int32_t)(bent.size() - efirst));
        for (size_t c = cfirst; c < bcol.size(); ++c) local[(size_t)bcol[c]] = -1;
        bent_ptr.push_back((int32_t)bent.size());
        bcol_ptr.push_back((int32_t)bcol.size());
        max_cols = std::max(max_cols, (int)(bcol.size() - cfirst));
        max_items = std::max(max_items, 6 * (int)(bent.size() - efirst));
      }
      pg->pk_lds_cols = max_cols;
      pg->pk_lds_items = max_items;
      pg->pk_lds_bytes = ((size_t)max_cols * 6 + (size_t)max_items + PK_ROWS + 7 * (size_t)pg->n_c + 8 * PK_W) * sizeof(double);
      if (std::getenv("LSLAM_DEBUG")) {
        int hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < pg->n_agg; ++a) hist[std::min(8, (agg_ptr[(size_t)a + 1] - agg_ptr[(size_t)a]) / 8)]++;
        fprintf(stderr, "[lslam pg] aggregate sizes (members / 8): %d %d %d %d %d %d %d %d %d\n", hist[0], hist[1], hist[2], hist[3],
                hist[4], hist[5], hist[6], hist[7], hist[8]);
      }
      if (std::getenv("LSLAM_DEBUG"))
        fprintf(stderr, "[lslam pg] %d aggregates, widest: %d columns, %d items; persistent-kernel LDS %zu bytes\n", pg->n_agg,
                max_cols, max_items, pg->pk_lds_bytes);
      PG_TRY(dev_upload(&pg->d_bent_ptr, bent_ptr));
      PG_TRY(dev_upload(&pg->d_bent, bent));
      PG_TRY(dev_upload(&pg->d_blc, blc));
      PG_TRY(dev_upload(&pg->d_bmptr, bmptr));
      PG_TRY(dev_upload(&pg->d_bcol_ptr, bcol_ptr));
      PG_TRY(dev_upload(&pg->d_bcol, bcol));
      // [rc0 | rc1] then the sentinel-initialised slots [zA 2 n6 | rzA 2 n_agg | rrA 2 n_agg | pqB 2 n_agg | qcB 2 n_c]
      PG_TRY(hipMalloc((void **)&pg->d_pk, (4 * (size_t)pg->n_c + 6 * (size_t)pg->n_agg + 12 * (size_t)n_v) * sizeof(double)));
      PG_TRY(hipMalloc((void **)&pg->d_bar, 2 * sizeof(unsigned)));
    }
  }
  std::vector<double> hp(poses7, poses7 + 7 * (size_t)n_v), hm(meas7, meas7 + 7 * (size_t)n_e),
      hi(info36, info36 + 36 * (size_t)n_e);
  PG_TRY(dev_upload(&pg->d_poses, hp));
  PG_TRY(dev_upload(&pg->d_trial, hp));
  PG_TRY(dev_upload(&pg->d_meas, hm));
  PG_TRY(dev_upload(&pg->d_info, hi));
  PG_TRY(dev_upload(&pg->d_ij, pg->h_ij));
  PG_TRY(dev_upload(&pg->d_row_ptr, rptr));
  PG_TRY(dev_upload(&pg->d_row_col, rcol));
  PG_TRY(dev_upload(&pg->d_row_src, rsrc));
  PG_TRY(dev_upload(&pg->d_row_of, rof));
  const size_t n6 = (size_t)n_v * 6;
  pg->n_cg_blocks = (int)((n6 + CG_ROWS - 1) / CG_ROWS);
  pg->n_parts = pg->n_cg_blocks + pg->n_cblk;
  PG_TRY(hipMalloc((void **)&pg->d_sys, pg->sys_doubles() * sizeof(double)));
  PG_TRY(hipMalloc((void **)&pg->d_vals, (size_t)pg->n_entries * 36 * sizeof(double)));
  PG_TRY(hipMalloc((void **)&pg->d_minv, (size_t)n_v * 36 * sizeof(double)));
  for (double **p : {&pg->d_x, &pg->d_r, &pg->d_z})
    PG_TRY(hipMalloc((void **)p, n6 * sizeof(double)));
  PG_TRY(hipMalloc((void **)&pg->d_q, (size_t)pg->n_entries * 6 * sizeof(double)));
  PG_TRY(hipMalloc((void **)&pg->d_p, 2 * n6 * sizeof(double)));
  PG_TRY(hipMalloc((void **)&pg->d_part, (3 * (size_t)pg->n_parts + ((size_t)pg->n_entries * 6 + PROD_BLOCK - 1) / PROD_BLOCK) * sizeof(double)));
  PG_TRY(hipMalloc((void **)&pg->d_scal, 32 * sizeof(double)));
  PG_TRY(hipMemset(pg->d_scal, 0, 32 * sizeof(double)));
  PG_TRY(hipMalloc((void **)&pg->d_tmp, 8 * sizeof(double)));
  int rc = build_shard(pg, 0, n_e);
  if (rc) return rc;
  guard.p = nullptr;
  *out = pg;
  return LSLAM_OK;
}

void lslam_pg_destroy(lslam_pg *pg) {
  if (!pg) return;
  (void)hipSetDevice(pg->device);
  if (pg->stream) (void)hipStreamSynchronize(pg->stream);
  for (void *p : {(void *)pg->d_poses, (void *)pg->d_trial, (void *)pg->d_meas, (void *)pg->d_info,
                  (void *)pg->d_ij, (void *)pg->d_rec, (void *)pg->d_chi, (void *)pg->d_vptr,
                  (void *)pg->d_vadj, (void *)pg->d_optr, (void *)pg->d_oadj, (void *)pg->d_row_ptr,
                  (void *)pg->d_row_col, (void *)pg->d_row_src, (void *)pg->d_row_of, (void *)pg->d_vals, (void *)pg->d_minv,
                  (void *)pg->d_P, (void *)pg->d_Ac, (void *)pg->d_rc, (void *)pg->d_yc, (void *)pg->d_gj, (void *)pg->d_cb_ptr,
                  (void *)pg->d_cb_ent, (void *)pg->d_cb_ab, (void *)pg->d_agg_of, (void *)pg->d_agg_ptr, (void *)pg->d_agg_mem,
                  (void *)pg->d_x, (void *)pg->d_r, (void *)pg->d_z, (void *)pg->d_p, (void *)pg->d_q,
                  (void *)pg->d_part, (void *)pg->d_scal, (void *)pg->d_tmp, (void *)pg->d_bent_ptr, (void *)pg->d_bent,
                  (void *)pg->d_blc, (void *)pg->d_bmptr, (void *)pg->d_bcol_ptr, (void *)pg->d_bcol, (void *)pg->d_pk,
                  (void *)pg->d_bar, (void *)pg->d_gjslots})
    if (p) (void)hipFree(p);
  if (pg->own_sys && pg->d_sys) (void)hipFree(pg->d_sys);
  if (pg->stream) (void)hipStreamDestroy(pg->stream);
  delete pg;
}

size_t lslam_pg_system_doubles(const lslam_pg *pg) { return pg ? pg->sys_doubles() : 0; }

void lslam_pg_row_shard_range(int32_t n_vertices, int32_t rank, int32_t world, int32_t *v_begin, int32_t *v_end) {
  // whole row blocks of the PCG kernels (21 vertices), as even as that allows
  const int per = CG_ROWS / 6;
  const int nblk = (n_vertices + per - 1) / per;
  const int b0 = (int)((long long)nblk * rank / world), b1 = (int)((long long)nblk * (rank + 1) / world);
  if (v_begin) *v_begin = std::min(n_vertices, b0 * per);
  if (v_end) *v_end = std::min(n_vertices, b1 * per);
}

int lslam_pg_set_row_shard(lslam_pg *pg, int32_t v_begin, int32_t v_end) {
  if (!pg) return LSLAM_ERR_INVALID;
  if (v_begin < 0 && v_end < 0) {  // back to the replicated solve
    pg->row_v0 = pg->row_v1 = -1;
    return LSLAM_OK;
  }
  if (v_begin < 0 || v_end < v_begin || v_end > pg->n_v || (v_begin % (CG_ROWS / 6)) != 0 ||
      (v_end != pg->n_v && (v_end % (CG_ROWS / 6)) != 0)) {
    g_pg_err = "row shard must be a range of whole 21-vertex row blocks (lslam_pg_row_shard_range)";
    return LSLAM_ERR_INVALID;
  }
  if (v_begin == v_end) {
    // more ranks than 21-vertex row blocks (e.g. 20 keyframes on 2 ranks): a rank without rows has nothing to multiply or
    // update, and the solve's launches are not written for that.  Such a graph is far too small to share: replicated solve.
    g_pg_err = "empty row shard: the graph has fewer 21-vertex row blocks than ranks -- use the replicated solve (lslam_pg_set_row_shard(pg, -1, -1))";
    return LSLAM_ERR_INVALID;
  }
  pg->row_v0 = v_begin;
  pg->row_v1 = v_end;
  return LSLAM_OK;
}

int lslam_pg_set_solve_tolerance(lslam_pg *pg, double rel_tol) {
  if (!pg || !(rel_tol > 0.0) || !(rel_tol < 1.0)) {
    g_pg_err = "solve tolerance must be in (0, 1)";
    return LSLAM_ERR_INVALID;
  }
  pg->env_tol = rel_tol;
  return LSLAM_OK;
}

int32_t lslam_pg_row_sharded_solves(const lslam_pg *pg) { return pg ? pg->rs_solves : 0; }
int32_t lslam_pg_row_gathered_solves(const lslam_pg *pg) { return pg ? pg->rs_gathered : 0; }

int lslam_pg_set_row_gather(lslam_pg *pg, lslam_allgatherv_fn fn, void *user, int32_t rank, int32_t world) {
  if (!pg) return LSLAM_ERR_INVALID;
  if (fn && (world < 1 || world > lslam_pg::RS_MAX_WORLD || rank < 0 || rank >= world)) {
    g_pg_err = "bad rank / world for the row gather";
    return LSLAM_ERR_INVALID;
  }
  pg->rs_agree_world = -1;  // another transport: the ranks agree again
  pg->gatherv = fn;
  pg->gatherv_user = user;
  pg->gatherv_rank = fn ? rank : -1;
  pg->gatherv_world = fn ? world : 0;
  return LSLAM_OK;
}

int lslam_pg_set_comm(lslam_pg *pg, lslam_comm *comm) {
  if (!pg) return LSLAM_ERR_INVALID;
  pg->comm = comm;
  pg->rs_agree_world = -1;  // another transport: the ranks agree again
  return LSLAM_OK;
}
int32_t lslam_pg_num_offdiag(const lslam_pg *pg) { return pg ? pg->n_off : 0; }

int lslam_pg_set_shard(lslam_pg *pg, int32_t e_begin, int32_t e_end, lslam_allreduce_fn fn, void *user,
                       double *system_buf) {
  if (!pg || e_begin < 0 || e_end < e_begin || e_end > pg->n_e) {
    g_pg_err = "bad shard";
    return LSLAM_ERR_INVALID;
  }
  PG_TRY(hipSetDevice(pg->device));
  pg->allreduce = fn;
  pg->allreduce_user = user;
  if (system_buf) {
    if (pg->own_sys && pg->d_sys) (void)hipFree(pg->d_sys);
    pg->d_sys = system_buf;
    pg->own_sys = false;
  }
  return build_shard(pg, e_begin, e_end);
}

int lslam_pg_linearize(lslam_pg *pg, double *diag_out, double *off_out, int32_t *off_ij_out, double *b_out,
                       double *chi2_out) {
  if (!pg) return LSLAM_ERR_INVALID;
  PG_TRY(hipSetDevice(pg->device));
  int rc = linearize(pg, pg->d_poses);
  if (rc) return rc;
  if (diag_out) PG_TRY(hipMemcpyAsync(diag_out, pg->diag(), (size_t)pg->n_v * 36 * 8, hipMemcpyDeviceToHost, pg->stream));
  if (off_out) PG_TRY(hipMemcpyAsync(off_out, pg->off(), (size_t)pg->n_off * 36 * 8, hipMemcpyDeviceToHost, pg->stream));
  if (b_out) PG_TRY(hipMemcpyAsync(b_out, pg->b(), (size_t)pg->n_v * 6 * 8, hipMemcpyDeviceToHost, pg->stream));
  if (chi2_out) PG_TRY(hipMemcpyAsync(chi2_out, pg->chi(), 8, hipMemcpyDeviceToHost, pg->stream));
  PG_TRY(hipStreamSynchronize(pg->stream));
  if (off_ij_out) std::memcpy(off_ij_out, pg->off_pairs.data(), pg->off_pairs.size() * sizeof(int32_t));
  return LSLAM_OK;
}

int lslam_pg_solve(lslam_pg *pg, double lambda, double *dx_out, int32_t *cg_iters) {
  if (!pg) return LSLAM_ERR_INVALID;
  PG_TRY(hipSetDevice(pg->device));
  int it = 0;
  int rc = solve(pg, lambda, 4000, 1e-10, &it);
  if (rc) return rc;
  if (dx_out) {
    PG_TRY(hipMemcpyAsync(dx_out, pg->d_x, (size_t)pg->n_v * 6 * 8, hipMemcpyDeviceToHost, pg->stream));
    PG_TRY(hipStreamSynchronize(pg->stream));
  }
  if (cg_iters) *cg_iters = it;
  return LSLAM_OK;
}

// Profiling tap (-DLSLAM_PK_CLOCKS builds): accumulated per-phase time of the persistent PCG kernel, 100 MHz ticks.
int lslam_pg_debug_clocks(lslam_pg *pg, double out[12]) {
  if (!pg || !out) return LSLAM_ERR_INVALID;
  PG_TRY(hipSetDevice(pg->device));
  PG_TRY(hipMemcpy(out, pg->d_scal + 8, 12 * sizeof(double), hipMemcpyDeviceToHost));
  return LSLAM_OK;
}

int lslam_pg_get_poses(lslam_pg *pg, double *poses7) {
  if (!pg || !poses7) return LSLAM_ERR_INVALID;
  PG_TRY(hipSetDevice(pg->device));
  PG_TRY(hipMemcpyAsync(poses7, pg->d_poses, (size_t)pg->n_v * 7 * 8, hipMemcpyDeviceToHost, pg->stream));
  PG_TRY(hipStreamSynchronize(pg->stream));
  return LSLAM_OK;
}

// SolverG2O::save (solver_g2o.cpp:97-100 -> g2o::OptimizableGraph::save): the graph in g2o's text
// format -- VERTEX_SE3:QUAT id x y z qx qy qz qw, FIX id, EDGE_SE3:QUAT i j x y z qx qy qz qw and
// the 21 upper-triangular information entries, row by row -- with the current estimates.  Numbers are
// written with 17 significant digits (g2o uses the stream default of 6; readers accept both).
int lslam_pg_save_g2o(lslam_pg *pg, const char *path) {
  if (!pg || !path) return LSLAM_ERR_INVALID;
  PG_TRY(hipSetDevice(pg->device));
  std::vector<double> poses((size_t)pg->n_v * 7), meas((size_t)pg->n_e * 7), info((size_t)pg->n_e * 36);
  PG_TRY(hipMemcpyAsync(poses.data(), pg->d_poses, poses.size() * 8, hipMemcpyDeviceToHost, pg->stream));
  if (pg->n_e) {
    PG_TRY(hipMemcpyAsync(meas.data(), pg->d_meas, meas.size() * 8, hipMemcpyDeviceToHost, pg->stream));
    PG_TRY(hipMemcpyAsync(info.data(), pg->d_info, info.size() * 8, hipMemcpyDeviceToHost, pg->stream));
  }
  PG_TRY(hipStreamSynchronize(pg->stream));
  std::ofstream ofs(path);
  if (!ofs) {
    g_pg_err = std::string("cannot open ") + path;
    return LSLAM_ERR_INVALID;
  }
  ofs << std::setprecision(17);
  for (int v = 0; v < pg->n_v; ++v) {
    ofs << "VERTEX_SE3:QUAT " << v;
    for (int k = 0; k < 7; ++k) ofs << ' ' << poses[(size_t)v * 7 + k];
    ofs << '\n';
    if (v == pg->fixed) ofs << "FIX " << v << '\n';
  }
  for (int e = 0; e < pg->n_e; ++e) {
    ofs << "EDGE_SE3:QUAT " << pg->h_ij[2 * e] << ' ' << pg->h_ij[2 * e + 1];
    for (int k = 0; k < 7; ++k) ofs << ' ' << meas[(size_t)e * 7 + k];
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) ofs << ' ' << info[(size_t)e * 36 + r * 6 + c];
    ofs << '\n';
  }
  return ofs.good() ? LSLAM_OK : LSLAM_ERR_INVALID;
}

// Reader for the same format (host only, no device needed).  Two-call pattern: with NULL arrays it
// returns the counts; vertex ids are mapped to 0..n-1 in order of appearance.
int lslam_g2o_read(const char *path, int32_t *n_vertices, double *poses7, int32_t *n_edges, int32_t *ij,
                   double *meas7, double *info36, int32_t *fixed_vertex) {
  if (!path || !n_vertices || !n_edges) return LSLAM_ERR_INVALID;
  std::ifstream ifs(path);
  if (!ifs) {
    g_pg_err = std::string("cannot open ") + path;
    return LSLAM_ERR_INVALID;
  }
  std::map<long, int> ids;
  int nv = 0, ne = 0, fixed = -1;
  const int cap_v = *n_vertices, cap_e = *n_edges;
  std::string line, tag;
  while (std::getline(ifs, line)) {
    std::istringstream ls(line);
    if (!(ls >> tag)) continue;
    if (tag == "VERTEX_SE3:QUAT") {
      long id;
      double v[7];
      if (!(ls >> id)) continue;
      for (int k = 0; k < 7; ++k) ls >> v[k];
      if (!ls) { g_pg_err = "malformed VERTEX_SE3:QUAT line"; return LSLAM_ERR_INVALID; }
      ids[id] = nv;
      if (poses7 && nv < cap_v) for (int k = 0; k < 7; ++k) poses7[(size_t)nv * 7 + k] = v[k];
      ++nv;
    } else if (tag == "FIX") {
      long id;
      if (ls >> id) { auto it = ids.find(id); if (it != ids.end() && fixed < 0) fixed = it->second; }
    } else if (tag == "EDGE_SE3:QUAT") {
      long a, b;
      double m[7], u[21];
      ls >> a >> b;
      for (int k = 0; k < 7; ++k) ls >> m[k];
      for (int k = 0; k < 21; ++k) ls >> u[k];
      if (!ls) { g_pg_err = "malformed EDGE_SE3:QUAT line"; return LSLAM_ERR_INVALID; }
      auto ia = ids.find(a), ib = ids.find(b);
      if (ia == ids.end() || ib == ids.end()) { g_pg_err = "edge refers to an unknown vertex"; return LSLAM_ERR_INVALID; }
      if (ne < cap_e) {
        if (ij) { ij[2 * ne] = ia->second; ij[2 * ne + 1] = ib->second; }
        if (meas7) for (int k = 0; k < 7; ++k) meas7[(size_t)ne * 7 + k] = m[k];
        if (info36) {
          int q = 0;
          for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) {
              info36[(size_t)ne * 36 + r * 6 + c] = u[q];
              info36[(size_t)ne * 36 + c * 6 + r] = u[q];
              ++q;
            }
        }
      }
      ++ne;
    }
  }
  *n_vertices = nv;
  *n_edges = ne;
  if (fixed_vertex) *fixed_vertex = fixed;
  return LSLAM_OK;
}

// g2o OptimizationAlgorithmLevenberg schedule (see oracle/posegraph_oracle.py)
namespace {
struct EventPair {  // destroyed on every exit path
  hipEvent_t a = nullptr, b = nullptr;
  ~EventPair() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};
}  // namespace

int lslam_pg_optimize(lslam_pg *pg, int32_t max_iters, lslam_pg_stats *st_out) {
  if (!pg) return LSLAM_ERR_INVALID;
  PG_TRY(hipSetDevice(pg->device));
  lslam_pg_stats st;
  std::memset(&st, 0, sizeof(st));
  EventPair ev;
  PG_TRY(hipEventCreate(&ev.a));
  PG_TRY(hipEventCreate(&ev.b));
  PG_TRY(hipEventRecord(ev.a, pg->stream));
  const int n6 = pg->n_v * 6;
  const int fused0 = pg->fused_solves;
  double lambda = -1.0, ni = 2.0;
  for (int it = 0; it < max_iters; ++it) {
    int rc = linearize(pg, pg->d_poses);
    if (rc) return rc;
    double cur;
    PG_TRY(hipMemcpyAsync(&cur, pg->chi(), 8, hipMemcpyDeviceToHost, pg->stream));
    if (lambda < 0) {
      hipLaunchKernelGGL(pg_maxdiag_kernel, dim3(1), dim3(256), 0, pg->stream, pg->diag(), pg->n_v, pg->fixed, pg->d_tmp + 1);
      PG_TRY(hipGetLastError());
      double md;
      PG_TRY(hipMemcpyAsync(&md, pg->d_tmp + 1, 8, hipMemcpyDeviceToHost, pg->stream));
      PG_TRY(hipStreamSynchronize(pg->stream));
      lambda = 1e-5 * md;
    }
    PG_TRY(hipStreamSynchronize(pg->stream));
    if (it == 0) st.chi2_initial = cur;
    double rho = 0.0;
    int qmax = 0;
    for (;;) {
      int cg = 0;
      // Every damped system is solved to a relative residual of 1e-8 (1e-10 until the PCG was fused; 1e-8 leaves every parity
      // and convergence test unchanged, 1e-6 -- g2o's own PCG default -- shows in the sharded-against-full comparison at 3e-8 m;
      // LSLAM_PG_TOL overrides), however many PCG iterations that takes: near the
      // optimum lambda shrinks, the system of a 5 000-keyframe chain reaches a condition number of ~1e7 and block-Jacobi
      // PCG needs thousands of iterations -- but a TRUNCATED solve leaves exactly the slow (global bending) modes
      // unresolved, and LM then crawls (measured: capped at 800 iterations the 5 k / 25 k graph is still 5 m from its
      // optimum after 2 000 LM iterations; uncapped it arrives in 342).  LSLAM_PG_MAX_CG overrides the cap for A/B runs.
      const int max_cg = pg->env_max_cg;
      const double cg_tol = pg->env_tol;
      pg->fell_back = false;
      rc = solve(pg, lambda, max_cg, cg_tol, &cg);
      if (rc) return rc;
      st.cg_iterations += cg;
      hipLaunchKernelGGL(pg_update_kernel, dim3((pg->n_v + 127) / 128), dim3(128), 0, pg->stream, pg->d_poses,
                         pg->d_x, pg->n_v, pg->fixed, pg->d_trial);
      PG_TRY(hipGetLastError());
      // the trial's chi2 and the gain denominator are enqueued together and read back after ONE wait
      double tmp, scale;
      rc = eval_chi2(pg, pg->d_trial, nullptr);  // enqueue only
      if (rc) return rc;
      hipLaunchKernelGGL(pg_dot_scale_kernel, dim3(pg->n_cg_blocks), dim3(CG_BLOCK), 0, pg->stream, pg->d_x,
                         pg->b(), n6, lambda, pg->d_part);
      hipLaunchKernelGGL(pg_sum_kernel, dim3(1), dim3(SUM_BLOCK), 0, pg->stream, pg->d_part, pg->n_cg_blocks, 1, pg->d_tmp + 2);
      PG_TRY(hipGetLastError());  // a failed launch must not turn into a stale read below
      double chi_flag[2] = {0.0, 0.0};
      PG_TRY(hipMemcpyAsync(chi_flag, pg->chi(), pg->sharded() ? 16 : 8, hipMemcpyDeviceToHost, pg->stream));
      PG_TRY(hipMemcpyAsync(&scale, pg->d_tmp + 2, 8, hipMemcpyDeviceToHost, pg->stream));
      PG_TRY(hipStreamSynchronize(pg->stream));
      tmp = chi_flag[0];
      if (pg->sharded() && chi_flag[1] > 0.0) {
        // Some rank's persistent kernel timed out (or was refused) in this trial and that rank solved with the launch loop: a
        // different summation order, dx equal to ~1e-8 instead of bit for bit -- and the replicated solve is only useful while
        // every rank takes the same LM decisions.  Every rank switches to the launch loops for good and repeats the trial
        // (one more chi2 all-reduce on every rank: the collectives stay matched).
        pg->pk_fit = 0;
        if (pg->gj_fit == 1) pg->gj_fit = 0;
        pg->coarse_valid = false;
        continue;
      }
      scale += 1e-3;
      rho = (cur - tmp) / scale;
      st.lm_trials++;
      if (rho > 0 && std::isfinite(tmp)) {
        double alpha = 1.0 - std::pow(2 * rho - 1, 3);
        alpha = std::min(alpha, 2.0 / 3.0);
        lambda *= std::max(1.0 / 3.0, alpha);
        ni = 2.0;
        std::swap(pg->d_poses, pg->d_trial);  // accept
        cur = tmp;
      } else {
        lambda *= ni;
        ni *= 2.0;
      }
      qmax++;
      if (!(rho < 0 && qmax < 10)) break;
    }
    st.iterations = it + 1;
    st.chi2_final = cur;
    st.lambda = lambda;
    if (qmax == 10 || rho == 0) break;
  }
  PG_TRY(hipEventRecord(ev.b, pg->stream));
  PG_TRY(hipStreamSynchronize(pg->stream));
  PG_TRY(hipEventElapsedTime(&st.gpu_ms_total, ev.a, ev.b));
  st.fused_solves = pg->fused_solves - fused0;
  if (st_out) *st_out = st;
  return LSLAM_OK;
}

int lslam_posegraph_optimize(int device, int32_t n_vertices, double *poses7, int32_t n_edges, const int32_t *ij,
                             const double *meas7, const double *info36, int32_t fixed_vertex, int32_t max_iters,
                             lslam_pg_stats *stats) {
  lslam_pg *pg = nullptr;
  int rc = lslam_pg_create(device, n_vertices, poses7, n_edges, ij, meas7, info36, fixed_vertex, &pg);
  if (rc < 0) return rc;
  rc = lslam_pg_optimize(pg, max_iters, stats);
  if (rc >= 0) {
    const int rc2 = lslam_pg_get_poses(pg, poses7);
    if (rc2 < 0) rc = rc2;
  }
  lslam_pg_destroy(pg);
  return rc;
}

}  // extern "C"
