/*
 * lslam_oracle.c -- CPU ORACLE (test infrastructure only; see lslam_oracle.h).
 *
 * Plain-C restatement of the reference's scan-match arithmetic.  Every function
 * cites the reference lines it follows (paths relative to
 * /root/reference/L_SLAM/src/).  Where the reference calls into Eigen (absent
 * from /root/reference; version unpinned by the reference), the published
 * Eigen 3.3 algorithm is restated and the function says so.
 *
 * Compile with -ffp-contract=off.  All arithmetic is fp32 unless the reference
 * promotes to double (noted inline).
 */
#define _POSIX_C_SOURCE 199309L
#define ORACLE_PI 3.14159265358979323846 /* M_PI */
#include "lslam_oracle.h"
#ifdef ORACLE_OMP
#include <omp.h>
#endif

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ======================================================================== */
/* kd-tree: nanoflann v1.2.3 (util/nanoflann.hpp) restatement               */
/* ======================================================================== */

#define LEAF_MAX 10 /* nanoflann.hpp:478-483 KDTreeSingleIndexAdaptorParams */

typedef struct {
  int32_t leaf;          /* child1==child2==NULL, nanoflann.hpp:1438 */
  int32_t left, right;   /* leaf: vind range, nanoflann.hpp:833-835 */
  int32_t divfeat;       /* nanoflann.hpp:837 */
  float divlow, divhigh; /* nanoflann.hpp:838 */
  int32_t child1, child2;
} onode;

typedef struct { float low, high; } interval; /* nanoflann.hpp:845-847 */

struct oracle_kdtree {
  const float *pts;
  size_t n, stride;
  int32_t *vind;
  onode *nodes;
  size_t n_nodes, cap_nodes;
  interval root_bbox[3];
  int max_depth;
};

static inline float dataset_get(const oracle_kdtree *t, int32_t idx, int dim) {
  /* nanoflann_pcl.h:199-210 kdtree_get_pt */
  return t->pts[(size_t)idx * t->stride + (size_t)dim];
}

/* nanoflann.hpp:908-920 */
static void compute_min_max(const oracle_kdtree *t, const int32_t *ind, int32_t count,
                            int element, float *min_elem, float *max_elem) {
  *min_elem = dataset_get(t, ind[0], element);
  *max_elem = dataset_get(t, ind[0], element);
  for (int32_t i = 1; i < count; ++i) {
    float val = dataset_get(t, ind[i], element);
    if (val < *min_elem) *min_elem = val;
    if (val > *max_elem) *max_elem = val;
  }
}

/* nanoflann.hpp:1043-1078 planeSplit */
static void plane_split(const oracle_kdtree *t, int32_t *ind, const int32_t count,
                        int cutfeat, float cutval, int32_t *lim1, int32_t *lim2) {
  int32_t left = 0;
  int32_t right = count - 1;
  for (;;) {
    while (left <= right && dataset_get(t, ind[left], cutfeat) < cutval) ++left;
    while (right && left <= right && dataset_get(t, ind[right], cutfeat) >= cutval) --right;
    if (left > right || !right) break;
    int32_t tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
    ++left;
    --right;
  }
  *lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && dataset_get(t, ind[left], cutfeat) <= cutval) ++left;
    while (right && left <= right && dataset_get(t, ind[right], cutfeat) > cutval) --right;
    if (left > right || !right) break;
    int32_t tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
    ++left;
    --right;
  }
  *lim2 = left;
}

/* nanoflann.hpp:982-1031 middleSplit_ */
static void middle_split(const oracle_kdtree *t, int32_t *ind, int32_t count, int32_t *index,
                         int *cutfeat, float *cutval, const interval bbox[3]) {
  const float EPS = 0.00001f;
  float max_span = bbox[0].high - bbox[0].low;
  for (int i = 1; i < 3; ++i) {
    float span = bbox[i].high - bbox[i].low;
    if (span > max_span) max_span = span;
  }
  float max_spread = -1;
  *cutfeat = 0;
  for (int i = 0; i < 3; ++i) {
    float span = bbox[i].high - bbox[i].low;
    if (span > (1 - EPS) * max_span) {
      float min_elem, max_elem;
      compute_min_max(t, ind, count, i, &min_elem, &max_elem);
      float spread = max_elem - min_elem;
      if (spread > max_spread) {
        *cutfeat = i;
        max_spread = spread;
      }
    }
  }
  float split_val = (bbox[*cutfeat].low + bbox[*cutfeat].high) / 2;
  float min_elem, max_elem;
  compute_min_max(t, ind, count, *cutfeat, &min_elem, &max_elem);

  if (split_val < min_elem) *cutval = min_elem;
  else if (split_val > max_elem) *cutval = max_elem;
  else *cutval = split_val;

  int32_t lim1, lim2;
  plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);

  if (lim1 > count / 2) *index = lim1;
  else if (lim2 < count / 2) *index = lim2;
  else *index = count / 2;
}

static int32_t alloc_node(oracle_kdtree *t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (onode *)realloc(t->nodes, t->cap_nodes * sizeof(onode));
  }
  return (int32_t)t->n_nodes++;
}

/* nanoflann.hpp:931-980 divideTree.  Nodes are numbered in allocation order
 * (pool.allocate at function entry), i.e. preorder. */
static int32_t divide_tree(oracle_kdtree *t, const int32_t left, const int32_t right,
                           interval bbox[3], int depth) {
  int32_t ni = alloc_node(t);
  if (depth > t->max_depth) t->max_depth = depth;

  if ((right - left) <= (int32_t)LEAF_MAX) {
    t->nodes[ni].leaf = 1;
    t->nodes[ni].child1 = t->nodes[ni].child2 = -1;
    t->nodes[ni].left = left;
    t->nodes[ni].right = right;
    t->nodes[ni].divfeat = 0;
    t->nodes[ni].divlow = t->nodes[ni].divhigh = 0;
    for (int i = 0; i < 3; ++i) {
      bbox[i].low = dataset_get(t, t->vind[left], i);
      bbox[i].high = dataset_get(t, t->vind[left], i);
    }
    for (int32_t k = left + 1; k < right; ++k) {
      for (int i = 0; i < 3; ++i) {
        if (bbox[i].low > dataset_get(t, t->vind[k], i)) bbox[i].low = dataset_get(t, t->vind[k], i);
        if (bbox[i].high < dataset_get(t, t->vind[k], i)) bbox[i].high = dataset_get(t, t->vind[k], i);
      }
    }
  } else {
    int32_t idx;
    int cutfeat;
    float cutval;
    middle_split(t, t->vind + left, right - left, &idx, &cutfeat, &cutval, bbox);

    interval left_bbox[3], right_bbox[3];
    memcpy(left_bbox, bbox, sizeof(left_bbox));
    left_bbox[cutfeat].high = cutval;
    int32_t c1 = divide_tree(t, left, left + idx, left_bbox, depth + 1);

    memcpy(right_bbox, bbox, sizeof(right_bbox));
    right_bbox[cutfeat].low = cutval;
    int32_t c2 = divide_tree(t, left + idx, right, right_bbox, depth + 1);

    onode *nd = &t->nodes[ni]; /* re-fetch: realloc may have moved the array */
    nd->leaf = 0;
    nd->left = nd->right = 0;
    nd->divfeat = cutfeat;
    nd->child1 = c1;
    nd->child2 = c2;
    nd->divlow = left_bbox[cutfeat].high;
    nd->divhigh = right_bbox[cutfeat].low;

    for (int i = 0; i < 3; ++i) {
      bbox[i].low = left_bbox[i].low < right_bbox[i].low ? left_bbox[i].low : right_bbox[i].low;
      bbox[i].high = left_bbox[i].high > right_bbox[i].high ? left_bbox[i].high : right_bbox[i].high;
    }
  }
  return ni;
}

oracle_kdtree *oracle_kdtree_build(const float *pts, size_t n, size_t stride_floats) {
  oracle_kdtree *t = (oracle_kdtree *)calloc(1, sizeof(*t));
  t->pts = pts;
  t->n = n;
  t->stride = stride_floats;
  /* nanoflann.hpp:1397-1404 init_vind */
  t->vind = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
  for (size_t i = 0; i < n; ++i) t->vind[i] = (int32_t)i;
  if (n == 0) return t; /* nanoflann.hpp:1278-1279 */
  /* nanoflann.hpp:1406-1427 computeBoundingBox */
  for (int i = 0; i < 3; ++i) t->root_bbox[i].low = t->root_bbox[i].high = dataset_get(t, 0, i);
  for (size_t k = 1; k < n; ++k) {
    for (int i = 0; i < 3; ++i) {
      float v = dataset_get(t, (int32_t)k, i);
      if (v < t->root_bbox[i].low) t->root_bbox[i].low = v;
      if (v > t->root_bbox[i].high) t->root_bbox[i].high = v;
    }
  }
  divide_tree(t, 0, (int32_t)n, t->root_bbox, 1);
  return t;
}

void oracle_kdtree_free(oracle_kdtree *t) {
  if (!t) return;
  free(t->vind);
  free(t->nodes);
  free(t);
}

size_t oracle_kdtree_num_nodes(const oracle_kdtree *t) { return t->n_nodes; }
int oracle_kdtree_max_depth(const oracle_kdtree *t) { return t->max_depth; }
void oracle_kdtree_vind(const oracle_kdtree *t, int32_t *out) {
  memcpy(out, t->vind, t->n * sizeof(int32_t));
}
void oracle_kdtree_node(const oracle_kdtree *t, size_t i, int32_t *kind, int32_t *a, int32_t *b,
                        float *divlow, float *divhigh, int32_t *child2) {
  const onode *nd = &t->nodes[i];
  *kind = nd->leaf ? 0 : 1;
  *a = nd->leaf ? nd->left : nd->divfeat;
  *b = nd->leaf ? nd->right : 0;
  *divlow = nd->divlow;
  *divhigh = nd->divhigh;
  *child2 = nd->child2;
}

/* nanoflann.hpp:81-137 KNNResultSet */
typedef struct {
  int32_t *indices;
  float *dists;
  int capacity, count;
} knn_set;

static inline void knn_add(knn_set *r, float dist, int32_t index) {
  int i;
  for (i = r->count; i > 0; --i) {
    if (r->dists[i - 1] > dist) { /* strict: no NANOFLANN_FIRST_MATCH */
      if (i < r->capacity) {
        r->dists[i] = r->dists[i - 1];
        r->indices[i] = r->indices[i - 1];
      }
    } else
      break;
  }
  if (i < r->capacity) {
    r->dists[i] = dist;
    r->indices[i] = index;
  }
  if (r->count < r->capacity) r->count++;
}

/* nanoflann.hpp:364-372 L2_Simple_Adaptor::evalMetric: x,y,z accumulated in order */
static inline float eval_metric(const oracle_kdtree *t, const float *a, int32_t b_idx) {
  float result = 0.0f;
  for (int i = 0; i < 3; ++i) {
    const float diff = a[i] - dataset_get(t, b_idx, i);
    result += diff * diff;
  }
  return result;
}

/* nanoflann.hpp:1433-1497 searchLevel (epsError == 1) */
static void search_level(const oracle_kdtree *t, knn_set *rs, const float *vec, int32_t ni,
                         float mindistsq, float dists[3], const float epsError) {
  const onode *node = &t->nodes[ni];
  if (node->leaf) {
    float worst_dist = rs->dists[rs->capacity - 1]; /* cached once per leaf */
    for (int32_t i = node->left; i < node->right; ++i) {
      const int32_t index = t->vind[i];
      float dist = eval_metric(t, vec, index);
      if (dist < worst_dist) knn_add(rs, dist, t->vind[i]);
    }
    return;
  }
  int idx = node->divfeat;
  float val = vec[idx];
  float diff1 = val - node->divlow;
  float diff2 = val - node->divhigh;
  int32_t bestChild, otherChild;
  float cut_dist;
  if ((diff1 + diff2) < 0) {
    bestChild = node->child1;
    otherChild = node->child2;
    cut_dist = (val - node->divhigh) * (val - node->divhigh); /* accum_dist :374-377 */
  } else {
    bestChild = node->child2;
    otherChild = node->child1;
    cut_dist = (val - node->divlow) * (val - node->divlow);
  }
  search_level(t, rs, vec, bestChild, mindistsq, dists, epsError);

  float dst = dists[idx];
  mindistsq = mindistsq + cut_dist - dst;
  dists[idx] = cut_dist;
  if (mindistsq * epsError <= rs->dists[rs->capacity - 1])
    search_level(t, rs, vec, otherChild, mindistsq, dists, epsError);
  dists[idx] = dst;
}

int oracle_kdtree_knn(const oracle_kdtree *t, const float q[3], int k, int32_t *idx_out,
                      float *d2_out) {
  knn_set rs;
  rs.indices = idx_out;
  rs.dists = d2_out;
  rs.capacity = k;
  rs.count = 0;
  if (k) d2_out[k - 1] = FLT_MAX; /* nanoflann.hpp:92-98 init */
  if (t->n == 0) return 0;        /* nanoflann.hpp:1306-1307 */
  float epsError = 1 + 0.0f;      /* SearchParams eps=0, nanoflann.hpp:1313 */
  float dists[3] = {0, 0, 0};
  /* nanoflann.hpp:1080-1097 computeInitialDistances */
  float distsq = 0.0f;
  for (int i = 0; i < 3; ++i) {
    if (q[i] < t->root_bbox[i].low) {
      dists[i] = (q[i] - t->root_bbox[i].low) * (q[i] - t->root_bbox[i].low);
      distsq += dists[i];
    }
    if (q[i] > t->root_bbox[i].high) {
      dists[i] = (q[i] - t->root_bbox[i].high) * (q[i] - t->root_bbox[i].high);
      distsq += dists[i];
    }
  }
  search_level(t, &rs, q, 0, distsq, dists, epsError);
  return rs.count;
}

/* ======================================================================== */
/* Dense algebra: Eigen 3.3 restatements (Eigen is NOT under /root/reference) */
/* ======================================================================== */

/* Eigen/src/Jacobi/Jacobi.h JacobiRotation::makeGivens (real case) */
static void make_givens(float p, float q, float *c, float *s) {
  if (q == 0.0f) {
    *c = p < 0.0f ? -1.0f : 1.0f;
    *s = 0.0f;
  } else if (p == 0.0f) {
    *c = 0.0f;
    *s = q < 0.0f ? 1.0f : -1.0f;
  } else if (fabsf(p) > fabsf(q)) {
    float t = q / p;
    float u = sqrtf(1.0f + t * t);
    if (p < 0.0f) u = -u;
    *c = 1.0f / u;
    *s = -t * (*c);
  } else {
    float t = p / q;
    float u = sqrtf(1.0f + t * t);
    if (q < 0.0f) u = -u;
    *s = -1.0f / u;
    *c = -t * (*s);
  }
}

/* Eigen/src/Core/MathFunctionsImpl.h positive_real_hypot (numext::hypot in 3.3) */
static float eigen_hypot(float x, float y) {
  float ax = fabsf(x), ay = fabsf(y);
  float p = ax > ay ? ax : ay;
  if (p == 0.0f) return 0.0f;
  float q = ax > ay ? ay : ax;
  float qp = q / p;
  return p * sqrtf(1.0f + qp * qp);
}

/* Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h tridiagonal_qr_step.
 * Q is n x n row-major (Q[r*n+c]); rotation applied on the right to columns k,k+1. */
static void tridiagonal_qr_step(float *diag, float *subdiag, int start, int end, float *Q, int n) {
  float td = (diag[end - 1] - diag[end]) * 0.5f;
  float e = subdiag[end - 1];
  float mu = diag[end];
  if (td == 0.0f) {
    mu -= fabsf(e);
  } else if (e != 0.0f) {
    const float e2 = e * e;
    const float h = eigen_hypot(td, e);
    if (e2 == 0.0f)
      mu -= e / ((td + (td > 0.0f ? h : -h)) / e);
    else
      mu -= e2 / (td + (td > 0.0f ? h : -h));
  }
  float x = diag[start] - mu;
  float z = subdiag[start];
  for (int k = start; k < end; ++k) {
    float c, s;
    make_givens(x, z, &c, &s);
    float sdk = s * diag[k] + c * subdiag[k];
    float dkp1 = s * subdiag[k] + c * diag[k + 1];
    diag[k] = c * (c * diag[k] - s * subdiag[k]) - s * (c * subdiag[k] - s * diag[k + 1]);
    diag[k + 1] = s * sdk + c * dkp1;
    subdiag[k] = c * sdk - s * dkp1;
    if (k > start) subdiag[k - 1] = c * subdiag[k - 1] - s * z;
    x = subdiag[k];
    if (k < end - 1) {
      z = -s * subdiag[k + 1];
      subdiag[k + 1] = c * subdiag[k + 1];
    }
    if (Q) {
      /* q.applyOnTheRight(k,k+1,rot): x_i' = c x_i - s y_i ; y_i' = s x_i + c y_i */
      for (int i = 0; i < n; ++i) {
        float xi = Q[i * n + k], yi = Q[i * n + k + 1];
        Q[i * n + k] = c * xi - s * yi;
        Q[i * n + k + 1] = s * xi + c * yi;
      }
    }
  }
}

/* SelfAdjointEigenSolver.h computeFromTridiagonal_impl (maxIterations = 30) + sort */
static void eig_from_tridiagonal(float *diag, float *subdiag, float *Q, int n) {
  int end = n - 1, start = 0, iter = 0;
  const float considerAsZero = FLT_MIN;
  const float precision = 2.0f * FLT_EPSILON;
  while (end > 0) {
    for (int i = start; i < end; ++i)
      if (fabsf(subdiag[i]) <= (fabsf(diag[i]) + fabsf(diag[i + 1])) * precision ||
          fabsf(subdiag[i]) <= considerAsZero)
        subdiag[i] = 0.0f;
    while (end > 0 && subdiag[end - 1] == 0.0f) end--;
    if (end <= 0) break;
    iter++;
    if (iter > 30 * n) break;
    start = end - 1;
    while (start > 0 && subdiag[start - 1] != 0.0f) start--;
    tridiagonal_qr_step(diag, subdiag, start, end, Q, n);
  }
  /* selection sort ascending, swapping eigenvector columns */
  for (int i = 0; i < n - 1; ++i) {
    int k = 0;
    float m = diag[i];
    for (int j = 1; j < n - i; ++j)
      if (diag[i + j] < m) { m = diag[i + j]; k = j; }
    if (k > 0) {
      float tmp = diag[i]; diag[i] = diag[k + i]; diag[k + i] = tmp;
      for (int r = 0; r < n; ++r) {
        tmp = Q[r * n + i]; Q[r * n + i] = Q[r * n + k + i]; Q[r * n + k + i] = tmp;
      }
    }
  }
}

/* SelfAdjointEigenSolver<Matrix3f>::compute with the 3x3 real tridiagonalization
 * specialisation (Tridiagonalization.h tridiagonalization_inplace_selector<M,3,false>). */
void oracle_eig_sym3(const float A[9], float evals[3], float V[9]) {
  float m[9];
  /* mat = lower triangle of A; scale to [-1,1] */
  m[0] = A[0]; m[1] = 0; m[2] = 0;
  m[3] = A[3]; m[4] = A[4]; m[5] = 0;
  m[6] = A[6]; m[7] = A[7]; m[8] = A[8];
  float scale = 0.0f;
  for (int i = 0; i < 9; ++i) if (fabsf(m[i]) > scale) scale = fabsf(m[i]);
  if (scale == 0.0f) scale = 1.0f;
  m[0] /= scale; m[3] /= scale; m[4] /= scale; m[6] /= scale; m[7] /= scale; m[8] /= scale;

  float diag[3], subdiag[2];
  const float tol = FLT_MIN;
  diag[0] = m[0];
  float v1norm2 = m[6] * m[6];
  if (v1norm2 <= tol) {
    diag[1] = m[4];
    diag[2] = m[8];
    subdiag[0] = m[3];
    subdiag[1] = m[7];
    V[0] = 1; V[1] = 0; V[2] = 0; V[3] = 0; V[4] = 1; V[5] = 0; V[6] = 0; V[7] = 0; V[8] = 1;
  } else {
    float beta = sqrtf(m[3] * m[3] + v1norm2);
    float invBeta = 1.0f / beta;
    float m01 = m[3] * invBeta;
    float m02 = m[6] * invBeta;
    float q = 2.0f * m01 * m[7] + m02 * (m[8] - m[4]);
    diag[1] = m[4] + m02 * q;
    diag[2] = m[8] - m02 * q;
    subdiag[0] = beta;
    subdiag[1] = m[7] - m01 * q;
    V[0] = 1; V[1] = 0;   V[2] = 0;
    V[3] = 0; V[4] = m01; V[5] = m02;
    V[6] = 0; V[7] = m02; V[8] = -m01;
  }
  eig_from_tridiagonal(diag, subdiag, V, 3);
  for (int i = 0; i < 3; ++i) evals[i] = diag[i] * scale;
}

/* Householder.h makeHouseholder: v = [c0; tail] (length len).  On return
 * tail := essential, *tau, *beta. */
static void make_householder(float *c0, float *tail, int tail_len, int tail_stride,
                             float *tau, float *beta) {
  float tailSqNorm = 0.0f;
  for (int i = 0; i < tail_len; ++i) tailSqNorm += tail[i * tail_stride] * tail[i * tail_stride];
  const float tol = FLT_MIN;
  if (tail_len == 0) tailSqNorm = 0.0f;
  if (tailSqNorm <= tol) {
    *tau = 0.0f;
    *beta = *c0;
    for (int i = 0; i < tail_len; ++i) tail[i * tail_stride] = 0.0f;
  } else {
    float b = sqrtf((*c0) * (*c0) + tailSqNorm);
    if (*c0 >= 0.0f) b = -b;
    float denom = *c0 - b;
    for (int i = 0; i < tail_len; ++i) tail[i * tail_stride] = tail[i * tail_stride] / denom;
    *tau = (b - *c0) / b;
    *beta = b;
  }
}

/* Householder.h applyHouseholderOnTheLeft on block M (rows x cols, row-major with
 * leading dimension ld); essential has rows-1 entries spaced ess_stride. */
static void apply_householder_left(float *M, int rows, int cols, int ld, const float *essential,
                                   int ess_stride, float tau) {
  if (rows == 1) {
    for (int j = 0; j < cols; ++j) M[j] *= (1.0f - tau);
  } else if (tau != 0.0f) {
    for (int j = 0; j < cols; ++j) {
      float tmp = 0.0f;
      for (int i = 1; i < rows; ++i) tmp += essential[(i - 1) * ess_stride] * M[i * ld + j];
      tmp += M[j];
      M[j] -= tau * tmp;
      for (int i = 1; i < rows; ++i) M[i * ld + j] -= tau * essential[(i - 1) * ess_stride] * tmp;
    }
  }
}

/* SelfAdjointEigenSolver<Matrix<float,6,6>>::compute, generic path:
 * Tridiagonalization.h tridiagonalization_inplace (Householder) then QR steps. */
void oracle_eig_sym6(const float A[36], float evals[6], float V[36]) {
  enum { N = 6 };
  float m[N * N];
  float scale = 0.0f;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      m[r * N + c] = (c <= r) ? A[r * N + c] : 0.0f;
      if (fabsf(m[r * N + c]) > scale) scale = fabsf(m[r * N + c]);
    }
  if (scale == 0.0f) scale = 1.0f;
  for (int r = 0; r < N; ++r)
    for (int c = 0; c <= r; ++c) m[r * N + c] /= scale;

  float hCoeffs[N - 1];
  for (int i = 0; i < N - 1; ++i) {
    int rem = N - i - 1;
    float h, beta;
    /* matA.col(i).tail(rem).makeHouseholderInPlace(h, beta) */
    /* (an empty tail for i = N - 2: its address is never formed -- row N does not exist) */
    make_householder(&m[(i + 1) * N + i], rem > 1 ? &m[(i + 2) * N + i] : &m[(i + 1) * N + i], rem - 1, N, &h, &beta);
    m[(i + 1) * N + i] = 1.0f;
    /* hCoeffs.tail(rem) = (A22.selfadjointView<Lower>() * (h * v)) */
    float v[N], p[N];
    for (int a = 0; a < rem; ++a) v[a] = m[(i + 1 + a) * N + i];
    for (int a = 0; a < rem; ++a) {
      float acc = 0.0f;
      for (int b = 0; b < rem; ++b) {
        int r = i + 1 + (a > b ? a : b), c = i + 1 + (a > b ? b : a);
        acc += m[r * N + c] * (h * v[b]);
      }
      p[a] = acc;
    }
    /* hCoeffs.tail += (h * -0.5 * (hCoeffs.tail . v)) * v */
    float dot = 0.0f;
    for (int a = 0; a < rem; ++a) dot += p[a] * v[a];
    float alpha = h * -0.5f * dot;
    for (int a = 0; a < rem; ++a) p[a] += alpha * v[a];
    /* A22.selfadjointView<Lower>().rankUpdate(v, p, -1): A -= v p^T + p v^T */
    for (int a = 0; a < rem; ++a)
      for (int b = 0; b <= a; ++b)
        m[(i + 1 + a) * N + (i + 1 + b)] -= (v[a] * p[b] + p[a] * v[b]);
    m[(i + 1) * N + i] = beta;
    hCoeffs[i] = h;
  }
  float diag[N], subdiag[N - 1];
  for (int i = 0; i < N; ++i) diag[i] = m[i * N + i];
  for (int i = 0; i < N - 1; ++i) subdiag[i] = m[(i + 1) * N + i];
  /* Q = HouseholderSequence(m, hCoeffs).setLength(N-1).setShift(1) evaluated */
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) V[r * N + c] = (r == c) ? 1.0f : 0.0f;
  for (int k = N - 2; k >= 0; --k) {
    int corner = N - k - 1;
    /* essential vector k: column k, rows k+2..N-1 */
    apply_householder_left(&V[(k + 1) * N + (k + 1)], corner, corner, N, corner > 1 ? &m[(k + 2) * N + k] : &m[(k + 1) * N + k], N,
                           hCoeffs[k]);
  }
  eig_from_tridiagonal(diag, subdiag, V, N);
  for (int i = 0; i < N; ++i) evals[i] = diag[i] * scale;
}

/* ColPivHouseholderQR.h computeInPlace + _solve_impl (Eigen 3.3), rows<=6, cols<=6.
 * A row-major rows x cols (copied). */
static void colpiv_qr_solve(int rows, int cols, const float *Ain, const float *b, float *x) {
  float qr[36], hC[6], normsU[6], normsD[6], c[6];
  int trans[6], perm[6];
  for (int i = 0; i < rows * cols; ++i) qr[i] = Ain[i];
  int size = rows < cols ? rows : cols;
  for (int k = 0; k < cols; ++k) {
    float s = 0.0f;
    for (int i = 0; i < rows; ++i) s += qr[i * cols + k] * qr[i * cols + k];
    normsD[k] = sqrtf(s);
    normsU[k] = normsD[k];
  }
  float maxn = normsU[0];
  for (int k = 1; k < cols; ++k) if (normsU[k] > maxn) maxn = normsU[k];
  float th = maxn * FLT_EPSILON;
  float threshold_helper = (th * th) / (float)rows;
  float norm_downdate_threshold = sqrtf(FLT_EPSILON);
  int nonzero_pivots = size;
  float maxpivot = 0.0f;
  for (int k = 0; k < size; ++k) {
    int big = 0;
    float bigv = normsU[k];
    for (int j = 1; j < cols - k; ++j)
      if (normsU[k + j] > bigv) { bigv = normsU[k + j]; big = j; }
    float biggest_col_sq_norm = bigv * bigv;
    big += k;
    if (nonzero_pivots == size && biggest_col_sq_norm < threshold_helper * (float)(rows - k))
      nonzero_pivots = k;
    trans[k] = big;
    if (k != big) {
      for (int i = 0; i < rows; ++i) {
        float t = qr[i * cols + k]; qr[i * cols + k] = qr[i * cols + big]; qr[i * cols + big] = t;
      }
      float t = normsU[k]; normsU[k] = normsU[big]; normsU[big] = t;
      t = normsD[k]; normsD[k] = normsD[big]; normsD[big] = t;
    }
    float beta;
    make_householder(&qr[k * cols + k], rows - k - 1 > 0 ? &qr[(k + 1) * cols + k] : &qr[k * cols + k], rows - k - 1, cols, &hC[k], &beta);
    qr[k * cols + k] = beta;
    if (fabsf(beta) > maxpivot) maxpivot = fabsf(beta);
    if (cols - k - 1 > 0)
      apply_householder_left(&qr[k * cols + k + 1], rows - k, cols - k - 1, cols,
                             rows - k - 1 > 0 ? &qr[(k + 1) * cols + k] : &qr[k * cols + k], cols, hC[k]);
    for (int j = k + 1; j < cols; ++j) {
      if (normsU[j] != 0.0f) {
        float temp = fabsf(qr[k * cols + j]) / normsU[j];
        temp = (1.0f + temp) * (1.0f - temp);
        temp = temp < 0.0f ? 0.0f : temp;
        float r = normsU[j] / normsD[j];
        float temp2 = temp * (r * r);
        if (temp2 <= norm_downdate_threshold) {
          float s = 0.0f;
          for (int i = k + 1; i < rows; ++i) s += qr[i * cols + j] * qr[i * cols + j];
          normsD[j] = sqrtf(s);
          normsU[j] = normsD[j];
        } else {
          normsU[j] *= sqrtf(temp);
        }
      }
    }
  }
  for (int k = 0; k < cols; ++k) perm[k] = k;
  for (int k = 0; k < size; ++k) {
    int t = perm[k]; perm[k] = perm[trans[k]]; perm[trans[k]] = t;
  }
  /* solve */
  if (nonzero_pivots == 0) {
    for (int i = 0; i < cols; ++i) x[i] = 0.0f;
    return;
  }
  for (int i = 0; i < rows; ++i) c[i] = b[i];
  for (int k = 0; k < nonzero_pivots; ++k)
    apply_householder_left(&c[k], rows - k, 1, 1, rows - k - 1 > 0 ? &qr[(k + 1) * cols + k] : &qr[k * cols + k], cols, hC[k]);
  /* upper-triangular solve, column-oriented (Eigen triangular_solve_vector, ColMajor) */
  for (int i = nonzero_pivots - 1; i >= 0; --i) {
    c[i] /= qr[i * cols + i];
    for (int r = 0; r < i; ++r) c[r] -= c[i] * qr[r * cols + i];
  }
  for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = c[i];
  for (int i = nonzero_pivots; i < cols; ++i) x[perm[i]] = 0.0f;
}

void oracle_qr_solve_5x3(const float A[15], const float b[5], float x[3]) {
  colpiv_qr_solve(5, 3, A, b, x);
}
void oracle_qr_solve_6x6(const float A[36], const float b[6], float x[6]) {
  colpiv_qr_solve(6, 6, A, b, x);
}

/* Matrix<float,6,6>::inverse(): Eigen uses PartialPivLU for sizes > 4. */
void oracle_inverse6(const float A[36], float Ainv[36]) {
  enum { N = 6 };
  float lu[N * N];
  int piv[N];
  memcpy(lu, A, sizeof(lu));
  for (int k = 0; k < N; ++k) {
    int p = k;
    float best = fabsf(lu[k * N + k]);
    for (int r = k + 1; r < N; ++r)
      if (fabsf(lu[r * N + k]) > best) { best = fabsf(lu[r * N + k]); p = r; }
    piv[k] = p;
    if (p != k)
      for (int c = 0; c < N; ++c) { float t = lu[k * N + c]; lu[k * N + c] = lu[p * N + c]; lu[p * N + c] = t; }
    if (lu[k * N + k] != 0.0f) {
      for (int r = k + 1; r < N; ++r) lu[r * N + k] /= lu[k * N + k];
    }
    for (int r = k + 1; r < N; ++r)
      for (int c = k + 1; c < N; ++c) lu[r * N + c] -= lu[r * N + k] * lu[k * N + c];
  }
  for (int col = 0; col < N; ++col) {
    float y[N];
    for (int r = 0; r < N; ++r) y[r] = (r == col) ? 1.0f : 0.0f;
    for (int k = 0; k < N; ++k) { float t = y[k]; y[k] = y[piv[k]]; y[piv[k]] = t; }
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < r; ++c) y[r] -= lu[r * N + c] * y[c];
    for (int r = N - 1; r >= 0; --r) {
      for (int c = r + 1; c < N; ++c) y[r] -= lu[r * N + c] * y[c];
      y[r] /= lu[r * N + r];
    }
    for (int r = 0; r < N; ++r) Ainv[r * N + col] = y[r];
  }
}

/* ======================================================================== */
/* Geometry                                                                 */
/* ======================================================================== */

typedef struct { float w, x, y, z; } quat;

/* Eigen Quaternion(AngleAxis): w = cos(a/2), vec = sin(a/2)*axis */
static quat quat_axis(float angle, int axis) {
  float ha = 0.5f * angle;
  quat q;
  q.w = cosf(ha);
  float s = sinf(ha);
  q.x = axis == 0 ? s : 0.0f;
  q.y = axis == 1 ? s : 0.0f;
  q.z = axis == 2 ? s : 0.0f;
  return q;
}
/* Eigen quat_product (generic, non-SIMD form) */
static quat quat_mul(quat a, quat b) {
  quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

/* transform_utils.h:288-299 getTransformationTZYX + :308-311 convertTransform:
 * q = AngleAxis(rz,Z)*AngleAxis(ry,Y)*AngleAxis(rx,X); R = q.toRotationMatrix(). */
void oracle_pose_to_Rt(const float pose[6], float R[9], float t[3]) {
  quat q = quat_mul(quat_mul(quat_axis(pose[2], 2), quat_axis(pose[1], 1)), quat_axis(pose[0], 0));
  const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
  t[0] = pose[3]; t[1] = pose[4]; t[2] = pose[5];
}

/* transform_utils.h:476-482: po = it * pi  (= R*p + t, row dot products in x,y,z order) */
void oracle_transform_point(const float R[9], const float t[3], const float p[3], float out[3]) {
  out[0] = ((R[0] * p[0] + R[1] * p[1]) + R[2] * p[2]) + t[0];
  out[1] = ((R[3] * p[0] + R[4] * p[1]) + R[5] * p[2]) + t[1];
  out[2] = ((R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]) + t[2];
}

/* transform_utils.h:54-60 getEulerAngles + :313-323 */
void oracle_Rt_to_pose(const float R[9], const float t[3], float pose[6]) {
  pose[0] = atan2f(R[7], R[8]);
  pose[1] = asinf(-R[6]);
  pose[2] = atan2f(R[3], R[0]);
  pose[3] = t[0]; pose[4] = t[1]; pose[5] = t[2];
}

/* feature_utils.h:108-154 findLine */
int oracle_find_line(const float *pts, size_t stride, const int32_t idx[5], float A[3], float B[3]) {
  float c[3] = {0, 0, 0};
  for (int j = 0; j < 5; ++j) {
    const float *p = pts + (size_t)idx[j] * stride;
    c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
  }
  c[0] /= 5.0f; c[1] /= 5.0f; c[2] /= 5.0f; /* _lineCentroid /= 5.0 */
  float M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 5; ++j) {
    const float *p = pts + (size_t)idx[j] * stride;
    float a0 = p[0] - c[0], a1 = p[1] - c[1], a2 = p[2] - c[2];
    M[0] += a0 * a0; /* (0,0) */
    M[3] += a0 * a1; /* (1,0) */
    M[6] += a0 * a2; /* (2,0) */
    M[4] += a1 * a1; /* (1,1) */
    M[7] += a1 * a2; /* (2,1) */
    M[8] += a2 * a2; /* (2,2) */
  }
  for (int i = 0; i < 9; ++i) M[i] /= 5.0f;
  float D[3], V[9];
  oracle_eig_sym3(M, D, V);
  if (D[2] > 5 * D[1]) {
    float v[3] = {V[2], V[5], V[8]}; /* col(2) */
    for (int i = 0; i < 3; ++i) {
      A[i] = c[i] - v[i] * 0.1f;
      B[i] = c[i] + v[i] * 0.1f;
    }
    return 1;
  }
  return 0;
}

static inline void cross3(const float a[3], const float b[3], float o[3]) {
  /* Eigen cross(): (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0) */
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline float norm3(const float a[3]) { return sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); }

/* feature_utils.h:17-26 getLinePointDistance + :63-75 getCornerFeatureCoefficients */
int oracle_corner_coeff(const float A[3], const float B[3], const float X[3], float coeff[4]) {
  float XB[3] = {X[0] - B[0], X[1] - B[1], X[2] - B[2]};
  float XA[3] = {X[0] - A[0], X[1] - A[1], X[2] - A[2]};
  float n[3];
  cross3(XB, XA, n);
  float nn = norm3(n);
  float AB[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]};
  float lengthAB = norm3(AB);
  float BA[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
  float mn[3] = {-n[0], -n[1], -n[2]};
  float cr[3];
  cross3(mn, BA, cr);
  float den = nn * lengthAB;
  float dir[3] = {cr[0] / den, cr[1] / den, cr[2] / den};
  float distance = nn / lengthAB;
  /* feature_utils.h:70 `1 - 0.9f * fabs(distance)`: `fabs` is unqualified and non-dependent inside a template of namespace
   * lidar_slam with only <cmath> in sight, so it binds to ::fabs(double) at the template's definition (g++ 5 .. 11 probed:
   * sizeof(fabs(float)) == 8; `using std::fabs` in ScanMatch.cpp:16 comes later and does not reach it) -- the expression is
   * evaluated in double and rounded once on the assignment to float */
  float weight = (float)(1 - (double)0.9f * fabs((double)distance));
  coeff[0] = dir[0] * weight;
  coeff[1] = dir[1] * weight;
  coeff[2] = dir[2] * weight;
  coeff[3] = distance * weight;
  return weight > 0.1;
}

/* feature_utils.h:157-204 findPlane */
int oracle_find_plane(const float *pts, size_t stride, const int32_t idx[5], float max_distance,
                      float plane[4]) {
  float Am[15], bm[5] = {-1, -1, -1, -1, -1}, x[3];
  float c[3] = {0, 0, 0};
  for (int j = 0; j < 5; ++j) {
    const float *p = pts + (size_t)idx[j] * stride;
    c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
    Am[j * 3 + 0] = p[0]; Am[j * 3 + 1] = p[1]; Am[j * 3 + 2] = p[2];
  }
  c[0] /= 5.0f; c[1] /= 5.0f; c[2] /= 5.0f;
  colpiv_qr_solve(5, 3, Am, bm, x);
  plane[0] = x[0]; plane[1] = x[1]; plane[2] = x[2]; plane[3] = 0;
  /* Vector4f::norm() with w = 0 */
  float norm = sqrtf(((plane[0] * plane[0] + plane[1] * plane[1]) + plane[2] * plane[2]) + plane[3] * plane[3]);
  plane[0] /= norm; plane[1] /= norm; plane[2] /= norm; plane[3] /= norm;
  plane[3] = -((plane[0] * c[0] + plane[1] * c[1]) + plane[2] * c[2]);
  for (int j = 0; j < 5; ++j) {
    const float *p = pts + (size_t)idx[j] * stride;
    float distance = ((plane[0] * p[0] + plane[1] * p[1]) + plane[2] * p[2]) + plane[3];
    if (fabsf(distance) > max_distance) return 0;
  }
  return 1;
}

/* feature_utils.h:97-106 getSurfaceFeatureCoefficients (0.9 is a double literal;
 * sqrt() of the float norm evaluated in double). */
int oracle_surf_coeff(const float plane[4], const float X[3], float coeff[4]) {
  float distance = ((plane[0] * X[0] + plane[1] * X[1]) + plane[2] * X[2]) + plane[3];
  float xn = norm3(X);
  float weight = (float)(1 - 0.9 * (double)fabsf(distance) / sqrt((double)xn));
  coeff[0] = plane[0] * weight;
  coeff[1] = plane[1] * weight;
  coeff[2] = plane[2] * weight;
  coeff[3] = distance * weight;
  return weight > 0.1;
}

/* ScanMatch.cpp:185-203, verbatim operator precedence (quirk Q1 in arz). */
void oracle_jacobian_row(const float sc[6], const float p[3], const float coeff[4], float row[6],
                         float *b) {
  const float srx = sc[0], crx = sc[1], sry = sc[2], cry = sc[3], srz = sc[4], crz = sc[5];
  const float px = p[0], py = p[1], pz = p[2];
  const float cx = coeff[0], cy = coeff[1], cz = coeff[2];
  float arx = ((crz * sry * crx + srz * srx) * py + (srz * crx - crz * sry * srx) * pz) * cx +
              ((srz * sry * crx - crz * srx) * py - (srz * sry * srx + crz * crx) * pz) * cy +
              (cry * crx * py - cry * srx * pz) * cz;
  float ary = (-crz * sry * px + crz * cry * srx * py + crz * cry * crx * pz) * cx +
              (-srz * sry * px + srz * cry * srx * py + srz * cry * crx * pz) * cy +
              (-cry * px - sry * srx * py - sry * crx * pz) * cz;
  float arz = (-srz * cry * px - (srz * sry * srx + crz * crx) * py + (crz * srx - srz * sry * crx) * pz) * cx +
              (crz * cry * px + (crz * sry * srx - srz * crx) * py + crz * sry * crx + srz * srx * pz) * cy +
              0 * cz;
  row[0] = arx; row[1] = ary; row[2] = arz;
  row[3] = cx; row[4] = cy; row[5] = cz;
  *b = -coeff[3];
}

/* ======================================================================== */
/* Gauss-Newton loop                                                        */
/* ======================================================================== */

void oracle_default_opts(oracle_opts *o) {
  o->max_iterations = 10;
  o->delta_t_abort = 0.05f;
  o->delta_r_abort = 0.05f;
  o->use_score = 1;
  o->fine_score = 0;
  o->score_threshold = 800;
  o->match_percentage_threshold = 0.4;
}

static void pose_sincos(const float pose[6], float sc[6]) {
  /* Angle.h:17-18: cached std::cos/std::sin(float) */
  sc[0] = sinf(pose[0]); sc[1] = cosf(pose[0]);
  sc[2] = sinf(pose[1]); sc[3] = cosf(pose[1]);
  sc[4] = sinf(pose[2]); sc[5] = cosf(pose[2]);
}

void oracle_pose_sincos(const float pose[6], float sc[6]) { pose_sincos(pose, sc); }

typedef struct {
  /* sequential fp32 accumulators, rows in reference order */
  float AtA[36];
  float Atb[6];
  int n_rows, n_line, n_plane;
  double score;
} sweep_acc;

static void acc_row(sweep_acc *a, const float row[6], float b) {
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) a->AtA[i * 6 + j] += row[i] * row[j];
    a->Atb[i] += row[i] * b;
  }
}

/* One query point of ScanMatch.cpp:97-114 (corner) / :116-132 (surf). */
static void sweep_point(const oracle_kdtree *tree, const float *map, size_t map_stride, int is_surf,
                        const float *p, const float R[9], const float t[3], const float sc[6],
                        float d2_gate_first, /* <0: gate on d2[4]<5.0; else d2[0]<gate (fine score) */
                        sweep_acc *acc, int32_t *idx_out, float *d2_out, float *coeff_out,
                        uint8_t *flag_out) {
  float sel[3];
  int32_t idx[5];
  float d2[5];
  float coeff[4] = {0, 0, 0, 0};
  uint8_t flag = 0;
  oracle_transform_point(R, t, p, sel);
  oracle_kdtree_knn(tree, sel, 5, idx, d2);
  int gate = d2_gate_first < 0 ? (d2[4] < 5.0) : (d2[0] < d2_gate_first);
  if (gate) {
    flag |= 1;
    if (!is_surf) {
      float A[3], B[3];
      if (oracle_find_line(map, map_stride, idx, A, B)) {
        flag |= 2;
        if (oracle_corner_coeff(A, B, sel, coeff)) flag |= 4;
        acc->n_line++;
      }
    } else {
      float plane[4];
      if (oracle_find_plane(map, map_stride, idx, 0.2f, plane)) {
        flag |= 2;
        if (oracle_surf_coeff(plane, sel, coeff)) flag |= 4;
        acc->n_plane++;
      }
    }
  }
  if (flag & 4) {
    float row[6], b;
    oracle_jacobian_row(sc, p, coeff, row, &b);
    acc_row(acc, row, b);
    acc->n_rows++;
    acc->score += (double)expf(-fabsf(coeff[3])); /* ScanMatch.cpp:42-49 getScore */
  }
  if (idx_out) memcpy(idx_out, idx, sizeof(idx));
  if (d2_out) memcpy(d2_out, d2, sizeof(d2));
  if (coeff_out) memcpy(coeff_out, coeff, sizeof(coeff));
  if (flag_out) *flag_out = flag;
}

static void sweep_all(const oracle_kdtree *tc, const float *map_c, const oracle_kdtree *ts,
                      const float *map_s, size_t map_stride, const float *qc, size_t nqc,
                      const float *qs, size_t nqs, size_t q_stride, const float pose[6],
                      float gate_c, float gate_s, sweep_acc *acc, int32_t *knn_idx, float *knn_d2,
                      float *coeff, uint8_t *flags) {
  float R[9], t[3], sc[6];
  memset(acc, 0, sizeof(*acc));
  oracle_pose_to_Rt(pose, R, t);
  pose_sincos(pose, sc);
#ifdef ORACLE_OMP
  /* NOT reference behaviour (its hot path is single-threaded): the "all host cores" upper bound of
   * SURVEY 8d.  Points are split into contiguous chunks, one accumulator per thread, combined in
   * thread order (the sums differ from the sequential ones in the last bits). */
  {
    const int nth = omp_get_max_threads();
    sweep_acc *part = (sweep_acc *)calloc((size_t)nth, sizeof(sweep_acc));
    const long long total = (long long)(nqc + nqs);
#pragma omp parallel
    {
      sweep_acc local; /* on the thread's own stack: no false sharing */
      sweep_acc *a = &local;
      memset(a, 0, sizeof(*a));
#pragma omp for schedule(static) nowait
      for (long long k = 0; k < total; ++k) {
        const size_t o = (size_t)k;
        if (o < nqc)
          sweep_point(tc, map_c, map_stride, 0, qc + o * q_stride, R, t, sc, gate_c, a,
                      knn_idx ? knn_idx + 5 * o : NULL, knn_d2 ? knn_d2 + 5 * o : NULL,
                      coeff ? coeff + 4 * o : NULL, flags ? flags + o : NULL);
        else
          sweep_point(ts, map_s, map_stride, 1, qs + (o - nqc) * q_stride, R, t, sc, gate_s, a,
                      knn_idx ? knn_idx + 5 * o : NULL, knn_d2 ? knn_d2 + 5 * o : NULL,
                      coeff ? coeff + 4 * o : NULL, flags ? flags + o : NULL);
      }
      part[omp_get_thread_num()] = local;
    }
    for (int th = 0; th < nth; ++th) {
      for (int i = 0; i < 36; ++i) acc->AtA[i] += part[th].AtA[i];
      for (int i = 0; i < 6; ++i) acc->Atb[i] += part[th].Atb[i];
      acc->n_rows += part[th].n_rows;
      acc->n_line += part[th].n_line;
      acc->n_plane += part[th].n_plane;
      acc->score += part[th].score;
    }
    free(part);
    return;
  }
#endif
  for (size_t i = 0; i < nqc; ++i)
    sweep_point(tc, map_c, map_stride, 0, qc + i * q_stride, R, t, sc, gate_c, acc,
                knn_idx ? knn_idx + 5 * i : NULL, knn_d2 ? knn_d2 + 5 * i : NULL,
                coeff ? coeff + 4 * i : NULL, flags ? flags + i : NULL);
  for (size_t i = 0; i < nqs; ++i) {
    size_t o = nqc + i;
    sweep_point(ts, map_s, map_stride, 1, qs + i * q_stride, R, t, sc, gate_s, acc,
                knn_idx ? knn_idx + 5 * o : NULL, knn_d2 ? knn_d2 + 5 * o : NULL,
                coeff ? coeff + 4 * o : NULL, flags ? flags + o : NULL);
  }
}

void oracle_sweep(const oracle_kdtree *tc, const float *map_c, const oracle_kdtree *ts,
                  const float *map_s, size_t map_stride, const float *qc, size_t nqc,
                  const float *qs, size_t nqs, size_t q_stride, const float pose[6],
                  int32_t *knn_idx, float *knn_d2, float *coeff, uint8_t *flags, float sums[29]) {
  sweep_acc acc;
  sweep_all(tc, map_c, ts, map_s, map_stride, qc, nqc, qs, nqs, q_stride, pose, -1.0f, -1.0f, &acc,
            knn_idx, knn_d2, coeff, flags);
  if (sums) {
    int k = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) sums[k++] = acc.AtA[i * 6 + j];
    for (int i = 0; i < 6; ++i) sums[k++] = acc.Atb[i];
    sums[27] = (float)acc.n_rows;
    sums[28] = (float)(acc.n_line + acc.n_plane);
  }
}

/* ScanMatch.cpp:206-260 */
int oracle_gn_step(const float AtA[36], const float Atb[6], int iter, float pose[6], float matP[36],
                   int *degenerate, float eig_thresh, float delta_r_abort, float delta_t_abort,
                   float x_out[6], float *delta_r, float *delta_t) {
  float x[6];
  colpiv_qr_solve(6, 6, AtA, Atb, x); /* :209 */
  if (iter == 0) {                    /* :211-235 */
    float E[6], V[36], V2[36], Vinv[36];
    oracle_eig_sym6(AtA, E, V);
    memcpy(V2, V, sizeof(V2));
    *degenerate = 0;
    for (int i = 0; i < 6; ++i) {
      if (E[i] < eig_thresh) {
        for (int j = 0; j < 6; ++j) V2[i * 6 + j] = 0; /* row i (quirk Q2) */
        *degenerate = 1;
      } else
        break;
    }
    oracle_inverse6(V, Vinv);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) {
        float s = 0.0f;
        for (int k = 0; k < 6; ++k) s += Vinv[r * 6 + k] * V2[k * 6 + c];
        matP[r * 6 + c] = s;
      }
  }
  if (*degenerate) { /* :237-240 */
    float x2[6];
    memcpy(x2, x, sizeof(x2));
    for (int r = 0; r < 6; ++r) {
      float s = 0.0f;
      for (int k = 0; k < 6; ++k) s += matP[r * 6 + k] * x2[k];
      x[r] = s;
    }
  }
  for (int i = 0; i < 6; ++i) pose[i] = pose[i] + x[i]; /* :242-247 (Angle.h:29) */
  /* :249-253: rad2deg(float) -> float (math_utils.h:23), pow(float,int) -> double */
  double r0 = (double)(float)((double)x[0] * 180.0 / ORACLE_PI);
  double r1 = (double)(float)((double)x[1] * 180.0 / ORACLE_PI);
  double r2 = (double)(float)((double)x[2] * 180.0 / ORACLE_PI);
  float dR = (float)sqrt(r0 * r0 + r1 * r1 + r2 * r2);
  double t0 = (double)(x[3] * 100), t1 = (double)(x[4] * 100), t2 = (double)(x[5] * 100);
  float dT = (float)sqrt(t0 * t0 + t1 * t1 + t2 * t2);
  if (x_out) memcpy(x_out, x, sizeof(x));
  if (delta_r) *delta_r = dR;
  if (delta_t) *delta_t = dT;
  return (dR < delta_r_abort && dT < delta_t_abort);
}

/* ---- stereo reprojection rows (no reference code: PARITY UNPINNED; include/lslam_c.h states the
 * arithmetic, ORB-SLAM2's pose-only stereo edge) ------------------------------------------- */
static void stereo_dR(const float sc[6], float dRx[9], float dRy[9], float dRz[9]) {
  /* derivatives of R = Rz(rz) Ry(ry) Rx(rx) (transform_utils.h:288-299) by rx, ry, rz */
  const float srx = sc[0], crx = sc[1], sry = sc[2], cry = sc[3], srz = sc[4], crz = sc[5];
  dRx[0] = 0.0f; dRx[1] = crz * sry * crx + srz * srx;  dRx[2] = srz * crx - crz * sry * srx;
  dRx[3] = 0.0f; dRx[4] = srz * sry * crx - crz * srx;  dRx[5] = -(srz * sry * srx) - crz * crx;
  dRx[6] = 0.0f; dRx[7] = cry * crx;                    dRx[8] = -(cry * srx);
  dRy[0] = -(crz * sry); dRy[1] = crz * cry * srx; dRy[2] = crz * cry * crx;
  dRy[3] = -(srz * sry); dRy[4] = srz * cry * srx; dRy[5] = srz * cry * crx;
  dRy[6] = -cry;         dRy[7] = -(sry * srx);    dRy[8] = -(sry * crx);
  dRz[0] = -(srz * cry); dRz[1] = -(srz * sry * srx) - crz * crx; dRz[2] = crz * srx - srz * sry * crx;
  dRz[3] = crz * cry;    dRz[4] = crz * sry * srx - srz * crx;    dRz[5] = crz * sry * crx + srz * srx;
  dRz[6] = 0.0f; dRz[7] = 0.0f; dRz[8] = 0.0f;
}

static void mat3T_vec(const float M[9], const float d[3], float out[3]) { /* M^T d */
  out[0] = M[0] * d[0] + M[3] * d[1] + M[6] * d[2];
  out[1] = M[1] * d[0] + M[4] * d[1] + M[7] * d[2];
  out[2] = M[2] * d[0] + M[5] * d[1] + M[8] * d[2];
}

/* One observation: rows[3][7] = scaled [J | b]; returns the number of rows (0, 2 or 3). */
static int stereo_rows(const oracle_stereo_cam *c, const float R[9], const float t[3], const float dRx[9],
                       const float dRy[9], const float dRz[9], const float Xw[3], const float ob[3],
                       float inv_sigma2, float rows[3][7]) {
  memset(rows, 0, sizeof(float) * 21);
  const float d[3] = {Xw[0] - t[0], Xw[1] - t[1], Xw[2] - t[2]};
  float p[3], G[6][3]; /* G[k] = d p / d theta_k (lidar frame) */
  mat3T_vec(R, d, p);
  mat3T_vec(dRx, d, G[0]);
  mat3T_vec(dRy, d, G[1]);
  mat3T_vec(dRz, d, G[2]);
  for (int k = 0; k < 3; ++k) { /* d p / d t_k = -R^T e_k = -(row k of R) */
    G[3 + k][0] = -R[3 * k + 0];
    G[3 + k][1] = -R[3 * k + 1];
    G[3 + k][2] = -R[3 * k + 2];
  }
  const float *T = c->T_cl;
  const float x = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
  const float y = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
  const float z = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
  if (!(z > c->min_depth)) return 0;
  const int mono = ob[2] < 0.0f;
  const float iz = 1.0f / z;
  const float uL = c->fx * x * iz + c->cx, v = c->fy * y * iz + c->cy, uR = uL - c->bf * iz;
  const float e0 = uL - ob[0], e1 = v - ob[1], e2 = mono ? 0.0f : uR - ob[2];
  const float chi2 = (e0 * e0 + e1 * e1 + e2 * e2) * inv_sigma2;
  const float delta = mono ? c->huber_mono : c->huber_stereo;
  if (c->gate_outliers && chi2 > delta * delta) return 0;
  const float rchi = sqrtf(chi2);
  const float wh = rchi <= delta ? 1.0f : delta / rchi;
  const float s = sqrtf(c->weight * inv_sigma2 * wh);
  for (int k = 0; k < 6; ++k) {
    const float gx = T[0] * G[k][0] + T[1] * G[k][1] + T[2] * G[k][2];
    const float gy = T[4] * G[k][0] + T[5] * G[k][1] + T[6] * G[k][2];
    const float gz = T[8] * G[k][0] + T[9] * G[k][1] + T[10] * G[k][2];
    const float ju = c->fx * iz * (gx - x * iz * gz);
    const float jv = c->fy * iz * (gy - y * iz * gz);
    rows[0][k] = s * ju;
    rows[1][k] = s * jv;
    rows[2][k] = mono ? 0.0f : s * (ju + c->bf * iz * iz * gz);
  }
  rows[0][6] = -(s * e0);
  rows[1][6] = -(s * e1);
  rows[2][6] = mono ? 0.0f : -(s * e2);
  return mono ? 2 : 3;
}

static int stereo_accumulate(const oracle_stereo *s, const float pose[6], sweep_acc *acc, float *rows_out) {
  float R[9], t[3], sc[6], dRx[9], dRy[9], dRz[9];
  oracle_pose_to_Rt(pose, R, t);
  pose_sincos(pose, sc);
  stereo_dR(sc, dRx, dRy, dRz);
  int used = 0;
  for (size_t i = 0; i < s->n; ++i) {
    float rows[3][7];
    const int nr = stereo_rows(&s->cam, R, t, dRx, dRy, dRz, s->landmarks + 3 * i, s->obs + 3 * i,
                               s->inv_sigma2 ? s->inv_sigma2[i] : 1.0f, rows);
    if (rows_out) memcpy(rows_out + 21 * i, rows, sizeof(float) * 21);
    for (int r = 0; r < nr; ++r) acc_row(acc, rows[r], rows[r][6]);
    acc->n_rows += nr;
    used += nr > 0;
  }
  return used;
}

void oracle_stereo_sums(const oracle_stereo *s, const float pose[6], float sums[29], float *rows_out) {
  sweep_acc acc;
  memset(&acc, 0, sizeof(acc));
  const int used = stereo_accumulate(s, pose, &acc, rows_out);
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) sums[k++] = acc.AtA[i * 6 + j];
  for (int i = 0; i < 6; ++i) sums[k++] = acc.Atb[i];
  sums[27] = (float)acc.n_rows;
  sums[28] = (float)used;
}

static int scanmatch_impl(const float *map_c, size_t nc, const float *map_s, size_t ns,
                          size_t map_stride, const float *qc, size_t nqc, const float *qs,
                          size_t nqs, size_t q_stride, float pose_io[6], const oracle_opts *opts,
                          oracle_stats *st, const oracle_stereo *stereo, int *n_stereo_used);

int oracle_scanmatch_scan(const float *map_c, size_t nc, const float *map_s, size_t ns,
                          size_t map_stride, const float *qc, size_t nqc, const float *qs,
                          size_t nqs, size_t q_stride, float pose_io[6], const oracle_opts *opts,
                          oracle_stats *st) {
  return scanmatch_impl(map_c, nc, map_s, ns, map_stride, qc, nqc, qs, nqs, q_stride, pose_io, opts, st, NULL,
                        NULL);
}

int oracle_scanmatch_joint(const float *map_c, size_t nc, const float *map_s, size_t ns,
                           size_t map_stride, const float *qc, size_t nqc, const float *qs,
                           size_t nqs, size_t q_stride, const oracle_stereo *stereo, float pose_io[6],
                           const oracle_opts *opts, oracle_stats *st, int *n_stereo_used) {
  return scanmatch_impl(map_c, nc, map_s, ns, map_stride, qc, nqc, qs, nqs, q_stride, pose_io, opts, st,
                        stereo, n_stereo_used);
}

static int scanmatch_impl(const float *map_c, size_t nc, const float *map_s, size_t ns,
                          size_t map_stride, const float *qc, size_t nqc, const float *qs,
                          size_t nqs, size_t q_stride, float pose_io[6], const oracle_opts *opts,
                          oracle_stats *st, const oracle_stereo *stereo, int *n_stereo_used) {
  oracle_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof(*st));
  if (nc < 50 || ns < 100) { /* :57-61 */
    st->status = 1;
    return 0;
  }
  float pose[6];
  memcpy(pose, pose_io, sizeof(pose));
  double t0 = now_s();
  oracle_kdtree *tc = oracle_kdtree_build(map_c, nc, map_stride); /* :75 */
  oracle_kdtree *ts = oracle_kdtree_build(map_s, ns, map_stride); /* :76 */
  st->t_build = now_s() - t0;

  int converge = 0, degenerate = 0;
  float matP[36];
  memset(matP, 0, sizeof(matP));
  sweep_acc acc;
  memset(&acc, 0, sizeof(acc));
  for (int iter = 0; iter < opts->max_iterations; ++iter) { /* :91 */
    double ts0 = now_s();
    sweep_all(tc, map_c, ts, map_s, map_stride, qc, nqc, qs, nqs, q_stride, pose, -1.0f, -1.0f,
              &acc, NULL, NULL, NULL, NULL);
    st->t_sweep += now_s() - ts0;
    st->point_residuals += (long long)(nqc + nqs);
    if (stereo && stereo->n) { /* the stereo rows join the same A^T A / A^T b */
      const int used = stereo_accumulate(stereo, pose, &acc, NULL);
      if (n_stereo_used) *n_stereo_used = used;
    }
    st->n_line = acc.n_line;
    st->n_plane = acc.n_plane;
    st->n_rows = acc.n_rows;
    if (acc.n_rows < 50) break; /* :141-145 */
    double tv0 = now_s();
    converge = oracle_gn_step(acc.AtA, acc.Atb, iter, pose, matP, &degenerate, 100.0f,
                              opts->delta_r_abort, opts->delta_t_abort, NULL, &st->delta_r,
                              &st->delta_t);
    st->t_solve += now_s() - tv0;
    st->iterations = iter + 1;
    if (converge) break; /* :257-260 */
  }
  st->degenerate = degenerate;
  st->converged = converge;
  memcpy(pose_io, pose, sizeof(pose)); /* pose is always written back :324,331,338,343 */

  int ok = 0;
  if (converge && opts->use_score) { /* :263-341 */
    double score = acc.score;        /* coeffSel of the last sweep */
    double match_count = acc.n_line + acc.n_plane;
    float percent = (float)(match_count / (double)(nqc + nqs));
    if (opts->fine_score) { /* :272-321: recomputed and printed, not used for the gate */
      sweep_acc acc2;
      sweep_all(tc, map_c, ts, map_s, map_stride, qc, nqc, qs, nqs, q_stride, pose, 0.02f, 0.05f,
                &acc2, NULL, NULL, NULL, NULL);
      st->score2 = acc2.score;                                                           /* :317 */
      st->percent2 = (float)((double)(acc2.n_line + acc2.n_plane) / (double)(nqc + nqs)); /* :318-319 */
    }
    st->score = score;
    st->percent = percent;
    if (score < opts->score_threshold) st->status = 3;
    else if (percent < opts->match_percentage_threshold) st->status = 4;
    else { st->status = 0; ok = 1; }
  } else {
    st->status = 2; /* :342-346 */
  }
  oracle_kdtree_free(tc);
  oracle_kdtree_free(ts);
  return ok;
}


/* ======================================================================== */
/* Variant C: per-cube kd-trees (util/FeatureMap.h:465-691)                 */
/* ======================================================================== */

typedef struct {
  float *pts;          /* packed 4 floats per point, cube by cube */
  oracle_kdtree **tree; /* per cube (NULL when empty) */
  size_t *count, *first;
  int n_cubes;
} cube_set;

/* worldToCube + isIndexValid + toIndex (FeatureMap.h:475-487,102-108,146-148) */
static int cube_index(const oracle_cube_grid *g, float x, float y, float z) {
  int gi = (int)(roundf(x / g->cube_size) + (float)g->origin[0]);
  int gj = (int)(roundf(y / g->cube_size) + (float)g->origin[1]);
  int gk = (int)(roundf(z / g->cube_size) + (float)g->origin[2]);
  if (0 <= gi && gi < g->dims[0] && 0 <= gj && gj < g->dims[1] && 0 <= gk && gk < g->dims[2])
    return gi + gj * g->dims[0] + gk * g->dims[0] * g->dims[1];
  return -1;
}

static void cube_set_build(cube_set *cs, const oracle_cube_grid *g, const float *map, size_t n, size_t stride) {
  cs->n_cubes = g->dims[0] * g->dims[1] * g->dims[2];
  cs->count = (size_t *)calloc((size_t)cs->n_cubes, sizeof(size_t));
  cs->first = (size_t *)calloc((size_t)cs->n_cubes + 1, sizeof(size_t));
  cs->tree = (oracle_kdtree **)calloc((size_t)cs->n_cubes, sizeof(oracle_kdtree *));
  cs->pts = (float *)malloc((n ? n : 1) * 4 * sizeof(float));
  int *idx = (int *)malloc((n ? n : 1) * sizeof(int));
  for (size_t i = 0; i < n; ++i) {  /* pushCornerPoint / pushSurfPoint, :188-204 */
    idx[i] = cube_index(g, map[i * stride], map[i * stride + 1], map[i * stride + 2]);
    if (idx[i] >= 0) cs->count[idx[i]]++;
  }
  for (int c = 0; c < cs->n_cubes; ++c) cs->first[c + 1] = cs->first[c] + cs->count[c];
  size_t *fill = (size_t *)calloc((size_t)cs->n_cubes, sizeof(size_t));
  for (size_t i = 0; i < n; ++i) {
    if (idx[i] < 0) continue;
    float *d = cs->pts + 4 * (cs->first[idx[i]] + fill[idx[i]]++);
    d[0] = map[i * stride]; d[1] = map[i * stride + 1]; d[2] = map[i * stride + 2]; d[3] = 0;
  }
  for (int c = 0; c < cs->n_cubes; ++c)
    if (cs->count[c]) cs->tree[c] = oracle_kdtree_build(cs->pts + 4 * cs->first[c], cs->count[c], 4);
  free(fill);
  free(idx);
}

static void cube_set_free(cube_set *cs) {
  for (int c = 0; c < cs->n_cubes; ++c) oracle_kdtree_free(cs->tree[c]);
  free(cs->tree); free(cs->count); free(cs->first); free(cs->pts);
}

int oracle_scanmatch_cubes(const float *map_c, size_t nc, const float *map_s, size_t ns,
                           size_t map_stride, const oracle_cube_grid *grid, const float *qc,
                           size_t nqc, const float *qs, size_t nqs, size_t q_stride,
                           float pose_io[6], oracle_stats *st) {
  oracle_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof(*st));
  cube_set cc, cs;
  cube_set_build(&cc, grid, map_c, nc, map_stride);
  cube_set_build(&cs, grid, map_s, ns, map_stride);
  float pose[6];
  memcpy(pose, pose_io, sizeof(pose));
  int converge = 0, degenerate = 0;
  float matP[36];
  memset(matP, 0, sizeof(matP));
  sweep_acc acc;
  for (int iter = 0; iter < 10; ++iter) { /* FeatureMap.h:515 */
    float R[9], t[3], sc[6];
    memset(&acc, 0, sizeof(acc));
    oracle_pose_to_Rt(pose, R, t);
    pose_sincos(pose, sc);
    for (int type = 0; type < 2; ++type) {
      const cube_set *set = type ? &cs : &cc;
      const float *q = type ? qs : qc;
      const size_t nq = type ? nqs : nqc;
      for (size_t i = 0; i < nq; ++i) {
        float sel[3];
        oracle_transform_point(R, t, q + i * q_stride, sel);
        int idx = cube_index(grid, sel[0], sel[1], sel[2]); /* :523,545 */
        if (idx < 0) continue;
        if (set->count[idx] < 5) continue;                  /* :524,546 */
        sweep_point(set->tree[idx], set->pts + 4 * set->first[idx], 4, type, q + i * q_stride, R, t, sc,
                    -1.0f, &acc, NULL, NULL, NULL, NULL);
      }
    }
    st->point_residuals += (long long)(nqc + nqs);
    st->n_line = acc.n_line;
    st->n_plane = acc.n_plane;
    st->n_rows = acc.n_rows;
    if (acc.n_rows < 50) break; /* :571-574 */
    converge = oracle_gn_step(acc.AtA, acc.Atb, iter, pose, matP, &degenerate, 100.0f, 0.05f, 0.05f, NULL,
                              &st->delta_r, &st->delta_t);
    st->iterations = iter + 1;
    if (converge) break; /* :685-688 */
  }
  st->degenerate = degenerate;
  st->converged = converge;
  st->status = converge ? 0 : 2;
  memcpy(pose_io, pose, sizeof(pose)); /* :690 */
  cube_set_free(&cc);
  cube_set_free(&cs);
  return converge;
}


/* ======================================================================== */
/* Variant B: LaserOdometry::scanMatch (odometry/LaserOdometry.cpp:328-647) */
/* ======================================================================== */

/* LaserOdometry.cpp:135-142 transformToStart: s = 10*frac(intensity); Twist t = _transform*s
 * (Twist.h:28-35, Angle.h:45-49); it = getTransformationTZYX(t); po = it * pi */
static void odom_transform_to_start(const float pose[6], const float *pi, float po[3]) {
  float s = 10 * (pi[3] - (int)pi[3]);
  float ps[6];
  for (int k = 0; k < 3; ++k) ps[k] = pose[k] * s;        /* rot * scale */
  for (int k = 3; k < 6; ++k) ps[k] = pose[k] * s;        /* pos * scale */
  float R[9], t[3];
  oracle_pose_to_Rt(ps, R, t);
  oracle_transform_point(R, t, pi, po);
}

/* LaserOdometry.cpp:156-168 transformToEnd(CloudI): every point is de-skewed to the sweep start
 * (transformToStart) and then moved to the sweep end by the inverse of the full transform
 * (Eigen Isometry inverse: R^T, -R^T t). */
void oracle_transform_to_end(float *cloud, size_t n, size_t stride_floats, const float pose[6]) {
  float R[9], t[3], Ri[9], ti[3];
  oracle_pose_to_Rt(pose, R, t);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Ri[r * 3 + c] = R[c * 3 + r];
  for (int r = 0; r < 3; ++r) ti[r] = (-Ri[r * 3] * t[0] + -Ri[r * 3 + 1] * t[1]) + -Ri[r * 3 + 2] * t[2];
  for (size_t i = 0; i < n; ++i) {
    float *p = cloud + i * stride_floats, a[3], b[3];
    odom_transform_to_start(pose, p, a);
    oracle_transform_point(Ri, ti, a, b);
    p[0] = b[0]; p[1] = b[1]; p[2] = b[2];
  }
}

/* math_utils.h:47-54 calcSquaredDiff(a, b) */
static inline float sq_diff(const float *a, const float *b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return dx * dx + dy * dy + dz * dz;
}

/* feature_utils.h:42-61 getCornerFeatureCoefficients(A,B,X,iteration,coeff) */
static int odom_corner_coeff(const float *A, const float *B, const float *X, int iter, float coeff[4]) {
  float XB[3] = {X[0] - B[0], X[1] - B[1], X[2] - B[2]};
  float XA[3] = {X[0] - A[0], X[1] - A[1], X[2] - A[2]};
  float n[3];
  cross3(XB, XA, n);
  float nn = norm3(n);
  float AB[3] = {A[0] - B[0], A[1] - B[1], A[2] - B[2]};
  float lengthAB = norm3(AB);
  float BA[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
  float mn[3] = {-n[0], -n[1], -n[2]};
  float cr[3];
  cross3(mn, BA, cr);
  float den = nn * lengthAB;
  float dir[3] = {cr[0] / den, cr[1] / den, cr[2] / den};
  float distance = nn / lengthAB;
  float weight = 1.0;
  if (iter >= 5) weight = (float)(1 - 1.8 * (double)fabsf(distance));
  coeff[0] = dir[0] * weight;
  coeff[1] = dir[1] * weight;
  coeff[2] = dir[2] * weight;
  coeff[3] = distance * weight;
  return ((double)weight > 0.1 && distance != 0);
}

/* feature_utils.h:28-40 getSurfacePointDistance + :77-95 getSurfaceFeatureCoefficients(A,B,C,X,it) */
static int odom_surf_coeff(const float *A, const float *B, const float *C, const float *X, int iter,
                           float coeff[4]) {
  float BA[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
  float CA[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
  float nrm[3];
  cross3(BA, CA, nrm);
  float z = (nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2];
  if (z > 0.0f) { /* Eigen normalize() */
    float l = sqrtf(z);
    nrm[0] /= l; nrm[1] /= l; nrm[2] /= l;
  }
  float XA[3] = {X[0] - A[0], X[1] - A[1], X[2] - A[2]};
  float distance = (XA[0] * nrm[0] + XA[1] * nrm[1]) + XA[2] * nrm[2];
  float AX[3] = {A[0] - X[0], A[1] - X[1], A[2] - X[2]};
  float cosv = distance / norm3(nrm) / norm3(AX);
  if (cosv < 0) { nrm[0] *= -1.0f; nrm[1] *= -1.0f; nrm[2] *= -1.0f; }
  distance = fabsf(distance);
  float weight = 1;
  if (iter >= 5) weight = (float)(1 - 1.8 * (double)fabsf(distance) / sqrt((double)norm3(X)));
  coeff[0] = weight * nrm[0];
  coeff[1] = weight * nrm[1];
  coeff[2] = weight * nrm[2];
  coeff[3] = weight * distance;
  return ((double)weight > 0.1 && distance != 0);
}

int oracle_odometry_match(const float *lc, size_t n_lc, const float *ls, size_t n_ls, const float *sharp,
                          size_t n_sharp, const float *flat, size_t n_flat, size_t stride, float pose[6],
                          const oracle_odom_opts *opts, oracle_stats *st) {
  oracle_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof(*st));
  if (!(n_lc > 10 && n_ls > 100)) return 0; /* :337 */
  oracle_kdtree *tc = oracle_kdtree_build(lc, n_lc, stride);
  oracle_kdtree *ts = oracle_kdtree_build(ls, n_ls, stride);
  int *c1 = (int *)calloc(n_sharp + 1, sizeof(int)), *c2 = (int *)calloc(n_sharp + 1, sizeof(int));
  int *s1 = (int *)calloc(n_flat + 1, sizeof(int)), *s2 = (int *)calloc(n_flat + 1, sizeof(int)),
      *s3 = (int *)calloc(n_flat + 1, sizeof(int));
  int degenerate = 0;
  float matP[36];
  memset(matP, 0, sizeof(matP));
  int iters = 0;
  for (int iter = 0; iter < opts->max_iterations; ++iter) {
    sweep_acc acc;
    memset(&acc, 0, sizeof(acc));
    float sc[6];
    /* rows are accumulated in push order: sharp points then flat points */
    for (size_t i = 0; i < n_sharp; ++i) {
      const float *pi = sharp + i * stride;
      float sel[3];
      odom_transform_to_start(pose, pi, sel);
      if (iter % 5 == 0) { /* :358-408 */
        int32_t idx;
        float d2;
        oracle_kdtree_knn(tc, sel, 1, &idx, &d2);
        int closest = -1, min2 = -1;
        if (d2 < 25) {
          closest = idx;
          int scan = (int)lc[(size_t)closest * stride + 3];
          float minD2 = 25;
          /* quirk Q5: the loop bound is the QUERY count (:370); never read past the cloud */
          for (int j = closest + 1; j < (int)n_sharp && j < (int)n_lc; j++) {
            if ((int)lc[(size_t)j * stride + 3] > scan + 2.5) break;
            float d = sq_diff(lc + (size_t)j * stride, sel);
            if ((int)lc[(size_t)j * stride + 3] > scan) {
              if (d < minD2) { minD2 = d; min2 = j; }
            }
          }
          for (int j = closest - 1; j >= 0; j--) {
            if ((int)lc[(size_t)j * stride + 3] < scan - 2.5) break;
            float d = sq_diff(lc + (size_t)j * stride, sel);
            if ((int)lc[(size_t)j * stride + 3] < scan) {
              if (d < minD2) { minD2 = d; min2 = j; }
            }
          }
        }
        c1[i] = closest;
        c2[i] = min2;
      }
      if (c2[i] >= 0) { /* :409-418 */
        float coeff[4];
        if (odom_corner_coeff(lc + (size_t)c1[i] * stride, lc + (size_t)c2[i] * stride, sel, iter, coeff)) {
          float row[6], b;
          pose_sincos(pose, sc);
          oracle_jacobian_row(sc, pi, coeff, row, &b);
          b = (float)(-0.05 * (double)coeff[3]); /* :575 */
          acc_row(&acc, row, b);
          acc.n_rows++;
          acc.n_line++;
        }
      }
    }
    for (size_t i = 0; i < n_flat; ++i) {
      const float *pi = flat + i * stride;
      float sel[3];
      odom_transform_to_start(pose, pi, sel);
      if (iter % 5 == 0) { /* :424-483 */
        int32_t idx;
        float d2;
        oracle_kdtree_knn(ts, sel, 1, &idx, &d2);
        int closest = -1, min2 = -1, min3 = -1;
        if (d2 < 25) {
          closest = idx;
          int scan = (int)ls[(size_t)closest * stride + 3];
          float minD2 = 25, minD3 = 25;
          for (int j = closest + 1; j < (int)n_flat && j < (int)n_ls; j++) { /* Q5: :434 */
            if ((int)ls[(size_t)j * stride + 3] > scan + 2.5) break;
            float d = sq_diff(ls + (size_t)j * stride, sel);
            if ((int)ls[(size_t)j * stride + 3] <= scan) {
              if (d < minD2) { minD2 = d; min2 = j; }
            } else {
              if (d < minD3) { minD3 = d; min3 = j; }
            }
          }
          for (int j = closest - 1; j >= 0; j--) {
            if ((int)ls[(size_t)j * stride + 3] < scan - 2.5) break;
            float d = sq_diff(ls + (size_t)j * stride, sel);
            if ((int)ls[(size_t)j * stride + 3] >= scan) {
              if (d < minD2) { minD2 = d; min2 = j; }
            } else {
              if (d < minD3) { minD3 = d; min3 = j; }
            }
          }
        }
        s1[i] = closest;
        s2[i] = min2;
        s3[i] = min3;
      }
      if (s2[i] >= 0 && s3[i] >= 0) { /* :485-496 */
        float coeff[4];
        if (odom_surf_coeff(ls + (size_t)s1[i] * stride, ls + (size_t)s2[i] * stride,
                            ls + (size_t)s3[i] * stride, sel, iter, coeff)) {
          float row[6], b;
          pose_sincos(pose, sc);
          oracle_jacobian_row(sc, pi, coeff, row, &b);
          b = (float)(-0.05 * (double)coeff[3]);
          acc_row(&acc, row, b);
          acc.n_rows++;
          acc.n_plane++;
        }
      }
    }
    st->point_residuals += (long long)(n_sharp + n_flat);
    st->n_rows = acc.n_rows;
    st->n_line = acc.n_line;
    st->n_plane = acc.n_plane;
    iters = iter + 1;
    if (acc.n_rows < 10) continue; /* :501-503 */
    int conv = oracle_gn_step(acc.AtA, acc.Atb, iter, pose, matP, &degenerate, 10.0f, opts->delta_r_abort,
                              opts->delta_t_abort, NULL, &st->delta_r, &st->delta_t); /* :581-640, lambda<10 :596 */
    st->iterations++;
    for (int k = 0; k < 6; ++k)
      if (!isfinite(pose[k])) pose[k] = 0.0f; /* :622-634 */
    if (conv) { st->converged = 1; break; }   /* :642-644 */
  }
  st->degenerate = degenerate;
  free(c1); free(c2); free(s1); free(s2); free(s3);
  oracle_kdtree_free(tc);
  oracle_kdtree_free(ts);
  return iters;
}
