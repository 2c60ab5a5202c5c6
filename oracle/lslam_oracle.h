/*
 * lslam_oracle.h -- CPU ORACLE for the L_SLAM scan-match hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (the-cooper-mapper_amd/csrc) never links, includes or calls
 * anything in this directory.
 *
 * It is a plain-C restatement of the reference's single-threaded CPU
 * arithmetic (citations are relative to /root/reference/L_SLAM/src/):
 *   - kd-tree build + kNN : util/nanoflann.hpp (v1.2.3) via util/nanoflann_pcl.h
 *   - line / plane fit      : util/feature_utils.h:108-204
 *   - residual coefficients : util/feature_utils.h:17-26,63-75,97-106
 *   - pose -> (R,t)         : util/transform_utils.h:288-299,308-311,476-482
 *   - Gauss-Newton loop     : scan_to_scan_match/ScanMatch.cpp:51-347
 *
 * PARITY PIN STATUS
 *   kNN (build + search): PINNED.  oracle/_ref/libref_nanoflann.so is the
 *     reference's own nanoflann.hpp compiled from /root/reference; tests check
 *     this restatement against it live (when present) and against the committed
 *     golden vectors in tests/golden/ (generated from it).
 *   fit / residual / Jacobian / solve: the reference code for these depends on
 *     Eigen (absent from /root/reference and from this image, version not pinned
 *     by the reference) and has no tests or golden vectors of its own, so this
 *     part is "PARITY UNPINNED": it restates the published Eigen 3.3 algorithms
 *     (SelfAdjointEigenSolver, ColPivHouseholderQR, Quaternion) and is checked
 *     by numpy/LAPACK fp64 cross-checks and analytic properties only.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off  (no FMA contraction: fp32 d^2, and
 * therefore kNN indices, must be platform independent).
 */
#ifndef LSLAM_ORACLE_H
#define LSLAM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- kd-tree (nanoflann v1.2.3 restatement) ---------------- */

typedef struct oracle_kdtree oracle_kdtree;

/* nanoflann.hpp:1270-1284 buildIndex (leaf_max_size 10, nanoflann.hpp:478-483).
 * pts: n points, stride_floats floats apart, xyz in the first three floats. */
oracle_kdtree *oracle_kdtree_build(const float *pts, size_t n, size_t stride_floats);
void oracle_kdtree_free(oracle_kdtree *t);

/* nanoflann_pcl.h:150-162 nearestKSearch -> nanoflann.hpp:1303-1323.
 * Returns the number of neighbours found (min(k, n)). */
int oracle_kdtree_knn(const oracle_kdtree *t, const float q[3], int k,
                      int32_t *idx_out, float *d2_out);

/* Introspection used by tests (topology comparison with the product builder). */
size_t oracle_kdtree_num_nodes(const oracle_kdtree *t);
int oracle_kdtree_max_depth(const oracle_kdtree *t);
/* Copies vind (the permuted index array, nanoflann.hpp:852). */
void oracle_kdtree_vind(const oracle_kdtree *t, int32_t *out);
/* Per node, preorder: kind(0 leaf,1 inner), a, b, divlow, divhigh, child2.
 * leaf: a=left b=right; inner: a=divfeat. */
void oracle_kdtree_node(const oracle_kdtree *t, size_t i, int32_t *kind, int32_t *a,
                        int32_t *b, float *divlow, float *divhigh, int32_t *child2);

/* ---------------- small dense algebra (Eigen 3.3 restatements) ---------- */

/* SelfAdjointEigenSolver<Matrix<float,N,N>>::compute (reads lower triangle;
 * ascending eigenvalues; eigenvectors in columns of V, row-major V[r*N+c]). */
void oracle_eig_sym3(const float A[9], float evals[3], float V[9]);
void oracle_eig_sym6(const float A[36], float evals[6], float V[36]);
/* ColPivHouseholderQR<Matrix<float,R,C>>::solve, row-major A. */
void oracle_qr_solve_5x3(const float A[15], const float b[5], float x[3]);
void oracle_qr_solve_6x6(const float A[36], const float b[6], float x[6]);
/* Matrix<float,6,6>::inverse() (PartialPivLU based for N>4). */
void oracle_inverse6(const float A[36], float Ainv[36]);

/* ---------------- geometry (feature_utils.h / transform_utils.h) -------- */

/* transform_utils.h:288-299 getTransformationTZYX: R = Rz*Ry*Rx via quaternions. */
void oracle_pose_to_Rt(const float pose[6], float R[9], float t[3]);
/* transform_utils.h:476-482 pointAssociateToMap with R,t from oracle_pose_to_Rt. */
void oracle_transform_point(const float R[9], const float t[3], const float p[3], float out[3]);
/* transform_utils.h:54-60,313-323: (R,t) -> pose (Euler extraction). */
void oracle_Rt_to_pose(const float R[9], const float t[3], float pose[6]);

/* feature_utils.h:108-154. returns 1 if a line was found. */
int oracle_find_line(const float *pts, size_t stride_floats, const int32_t idx[5],
                     float A[3], float B[3]);
/* feature_utils.h:63-75 (+17-26). coeff = (w*dir, w*d). returns w>0.1 */
int oracle_corner_coeff(const float A[3], const float B[3], const float X[3], float coeff[4]);
/* feature_utils.h:157-204. returns 1 if a plane was found. */
int oracle_find_plane(const float *pts, size_t stride_floats, const int32_t idx[5],
                      float max_distance, float plane[4]);
/* feature_utils.h:97-106. returns w>0.1 */
int oracle_surf_coeff(const float plane[4], const float X[3], float coeff[4]);

/* ScanMatch.cpp:185-203: Jacobian row (6) and rhs b for one matched point.
 * sc = {srx,crx,sry,cry,srz,crz}. */
/* {srx, crx, sry, cry, srz, crz}: the values Angle caches (util/Angle.h:17-18) */
void oracle_pose_sincos(const float pose[6], float sc[6]);
void oracle_jacobian_row(const float sc[6], const float p[3], const float coeff[4],
                         float row[6], float *b);

/* ---------------- Gauss-Newton scan match (ScanMatch.cpp:51-347) -------- */

typedef struct {
  int max_iterations;    /* ScanMatch.cpp:21  (10) */
  float delta_t_abort;   /* ScanMatch.cpp:22  (0.05; LaserMatcher.cpp:94 sets 0.1) */
  float delta_r_abort;   /* ScanMatch.cpp:22 */
  int use_score;         /* ScanMatch.cpp:23  (true; LaserMatcher.cpp:95 false) */
  int fine_score;        /* ScanMatch.cpp:32  (false) */
  double score_threshold;            /* ScanMatch.cpp:24 (800) */
  double match_percentage_threshold; /* ScanMatch.cpp:24 (0.4) */
} oracle_opts;

typedef struct {
  int status;        /* 0 ok(true), 1 too few ref, 2 not converged, 3 low score, 4 low percent */
  int iterations;    /* GN iterations executed (solve performed) */
  int n_line;        /* line_match_count of the last sweep */
  int n_plane;       /* plane_match_count of the last sweep */
  int n_rows;        /* laserCloudSelNum of the last sweep */
  int degenerate;
  int converged;
  float delta_r, delta_t;
  double score, percent;
  /* timing taps for the CPU baseline (seconds) */
  double t_build, t_sweep, t_solve;
  long long point_residuals; /* sum over sweeps of (Nc+Ns) */
  double score2, percent2;   /* _fineScore (ScanMatch.cpp:272-321): the re-sweep at the final pose gated on d2[0] < 0.02 / 0.05;
                                the reference only prints them; 0 unless fine_score && converged && use_score */
} oracle_stats;

void oracle_default_opts(oracle_opts *o);

/* One sweep (ScanMatch.cpp:97-204) at a fixed pose, for parity taps.
 * Outputs (any may be NULL): knn_idx[N*5], knn_d2[N*5], coeff[N*4], flags[N]
 * (bit0 = d2[4]<5, bit1 = fit found (counted as match), bit2 = row kept), and
 * sums[29] = {21 upper-tri AtA row-major, 6 Atb, n_rows, n_match} accumulated
 * sequentially in fp32 in row order (corner rows then surf rows). */
void oracle_sweep(const oracle_kdtree *tc, const float *map_c, const oracle_kdtree *ts,
                  const float *map_s, size_t map_stride, const float *qc, size_t nqc,
                  const float *qs, size_t nqs, size_t q_stride, const float pose[6],
                  int32_t *knn_idx, float *knn_d2, float *coeff, uint8_t *flags,
                  float sums[29]);

/* Full call.  Rebuilds both kd-trees (reference quirk Q4, ScanMatch.cpp:68-76). */
int oracle_scanmatch_scan(const float *map_c, size_t nc, const float *map_s, size_t ns,
                          size_t map_stride, const float *qc, size_t nqc, const float *qs,
                          size_t nqs, size_t q_stride, float pose[6],
                          const oracle_opts *opts, oracle_stats *stats);

/* Variant C: FeatureMap::scanMatchScan(Corner, Surf, Twist&) (util/FeatureMap.h:490-691):
 * the map is a grid of cubes (worldToCube, :475-487: round(p/size)+origin), each with its
 * own kd-tree; a scan point is searched only in the tree of the cube it falls into, cubes
 * with fewer than 5 points are skipped (:524,546); at most 10 iterations, thresholds
 * 0.05/0.05 (:515,685), no score gate, no return value (quirk Q6).
 * map clouds are partitioned into cubes in input order (pushCornerPoint, :188-196). */
typedef struct {
  float cube_size;    /* _worldCubeSize (50) */
  int32_t origin[3];  /* _cubeOriginWidth/Height/Depth */
  int32_t dims[3];    /* _cubeWidth/Height/Depth */
} oracle_cube_grid;

int oracle_scanmatch_cubes(const float *map_c, size_t nc, const float *map_s, size_t ns,
                           size_t map_stride, const oracle_cube_grid *grid, const float *qc,
                           size_t nqc, const float *qs, size_t nqs, size_t q_stride,
                           float pose[6], oracle_stats *stats);

/* Variant B: LaserOdometry::scanMatch (odometry/LaserOdometry.cpp:328-647) -- scan-to-scan.
 * Clouds are {x,y,z,intensity} with intensity = ring id + relative time (util/pcl_util.h:30-37);
 * last_corner/last_surf must be in scan order (the ring-window searches walk neighbouring
 * indices).  pose is the persistent `_transform` (in/out).  Returns iterations executed. */
typedef struct {
  int max_iterations;  /* LaserOdometry.cpp:24 (25) */
  float delta_t_abort; /* :25 (0.1) */
  float delta_r_abort; /* :25 (0.1) */
} oracle_odom_opts;

int oracle_odometry_match(const float *last_corner, size_t n_lc, const float *last_surf, size_t n_ls,
                          const float *sharp, size_t n_sharp, const float *flat, size_t n_flat,
                          size_t stride, float pose[6], const oracle_odom_opts *opts,
                          oracle_stats *stats);

/* LaserOdometry::transformToEnd (odometry/LaserOdometry.cpp:156-168): cloud {x,y,z,intensity}
 * in place. */
void oracle_transform_to_end(float *cloud, size_t n, size_t stride_floats, const float pose[6]);

/* One GN solve step given sums (ScanMatch.cpp:206-260).  iter==0 computes the
 * degeneracy projector into matP/degenerate (in/out state). Returns converged. */
int oracle_gn_step(const float AtA[36], const float Atb[6], int iter, float pose[6],
                   float matP[36], int *degenerate, float eig_thresh,
                   float delta_r_abort, float delta_t_abort, float x_out[6],
                   float *delta_r, float *delta_t);

/* ---------------- joint LiDAR + stereo term (BASELINE configs[4]) -------------------------
 * PARITY UNPINNED: the reference holds no code for a visual term (README.md:51-71 announces it).
 * Restated from the published arithmetic of ORB-SLAM2's pose-only stereo edge
 * (EdgeStereoSE3ProjectXYZOnlyPose / Optimizer::PoseOptimization); see include/lslam_c.h. */
typedef struct {
  float fx, fy, cx, cy, bf;
  float T_cl[12];
  float weight, huber_stereo, huber_mono;
  int32_t gate_outliers;
  float min_depth;
} oracle_stereo_cam;

typedef struct {
  const float *landmarks; /* [n][3] map frame */
  const float *obs;       /* [n][3] uL, v, uR (uR < 0: monocular) */
  const float *inv_sigma2; /* [n] or NULL */
  size_t n;
  oracle_stereo_cam cam;
} oracle_stereo;

/* The stereo rows at `pose`, accumulated sequentially in fp32 in observation order:
 * sums[29] = {21 upper-tri AtA, 6 Atb, rows, observations used}.  rows_out (optional,
 * [n][3][7]): the scaled rows [J | b] of every observation (zero when skipped). */
void oracle_stereo_sums(const oracle_stereo *s, const float pose[6], float sums[29], float *rows_out);

/* oracle_scanmatch_scan with the stereo rows added to every iteration's normal equations. */
int oracle_scanmatch_joint(const float *map_c, size_t nc, const float *map_s, size_t ns,
                           size_t map_stride, const float *qc, size_t nqc, const float *qs,
                           size_t nqs, size_t q_stride, const oracle_stereo *stereo, float pose[6],
                           const oracle_opts *opts, oracle_stats *stats, int *n_stereo_used);

#ifdef __cplusplus
}
#endif
#endif
