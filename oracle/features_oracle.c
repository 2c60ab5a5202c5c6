/*
 * features_oracle.c -- CPU ORACLE for the feature-extraction front end (second
 * half).  TEST INFRASTRUCTURE ONLY (see lslam_oracle.h).
 *
 * Restates, single-threaded and in the reference's order of operations, odometry/
 * ScanRegistration.cpp (citations relative to /root/reference/L_SLAM/src/odometry/):
 *   extractFeatures        ScanRegistration.cpp:190-425
 *   setRegionBuffersFor    :427-469  (curvature over +-curvatureRegion points, stable merge sort)
 *   setScanBuffersFor      :471-531  (blind / occlusion / broken-edge marks per scan ring)
 *   markAsPicked           :533-555
 *   pointClassify          :557-687  (two one-sided 6-point line fits, SelfAdjointEigenSolver)
 * with the helpers of util/math_utils.h:46-52,76-92 and util/pcl_util.h:30-37 (toXYZI: the output
 * intensity is the input point's `curvature` field = ring id + relative time).
 *
 * Input is the ring-sorted full-resolution cloud MultiScanRegistration::process builds
 * (MultiScanRegistration.cpp:178-190) with its per-ring [first, last] index ranges; building that
 * cloud from raw driver packets (ring from the vertical angle, sweep start/end orientation, IMU
 * de-skew) is not restated.
 *
 * PARITY PIN STATUS: "parity unpinned" -- the code depends on Eigen (SelfAdjointEigenSolver,
 * restated in lslam_oracle.c) and PCL (VoxelGrid, restated in fmap_oracle.c), both absent here; the
 * reference has no tests or golden vectors for it.
 */
#include "features_oracle.h"

#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "fmap_oracle.h"
#include "lslam_oracle.h"

#define OR_PI 3.14159265358979323846 /* M_PI */

/* PointLabel, ScanRegistration.h:22-42 */
enum {
  L_UNKNOW = 6, L_SURF_PICKED_NEAR = 3, L_CORNER_SHARP = 1, L_SURFACE_LESS_FLAT = 0, L_SURFACE_FLAT = -1,
  L_ONESIDE_FLAT = 5, L_EDGE_BROKEN = -2, L_NEAR_BLOCK = -3, L_BLIND_BLOCK = -4, L_MESSY = 9
};

void oracle_reg_default_params(oracle_reg_params *p) {
  /* RegistrationParams ctor defaults, ScanRegistration.h:49-57 / ScanRegistration.cpp:14-30 */
  p->n_feature_regions = 6;
  p->curvature_region = 5;
  p->max_corner_sharp = 2;
  p->max_surface_flat = 4;
  p->less_flat_filter_size = 0.2f;
  p->surface_curvature_threshold = 0.02f;
  const float deg = 0.5f;                                /* blindDegreeThreshold */
  const float rad = (float)(deg * OR_PI / 180.0);         /* deg2rad(float), math_utils.h:37 */
  p->blind_threshold = (float)cos((double)rad);          /* cos(float) -> double -> float member */
}

typedef struct { const float *c; size_t s; } cloud_v;
static inline const float *P(const cloud_v *c, size_t i) { return c->c + i * c->s; }

static inline float sqdiff(const float *a, const float *b) { /* math_utils.h:46-52 */
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return dx * dx + dy * dy + dz * dz;
}
static inline float pdist(const float *p) { return sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); } /* :76-78 */
static inline float cos_angle(const float *a, const float *b) {                                      /* :86-92 */
  const float ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  const float dis = pdist(a) * pdist(b);
  return ab / dis;
}
static void fill_n(int *p, int n, int v) { for (int i = 0; i < n; ++i) p[i] = v; }

/* ScanRegistration.cpp:471-531 */
static void set_scan_buffers(const cloud_v *cl, size_t start, size_t end, const oracle_reg_params *cfg, int *picked) {
  const int cr = cfg->curvature_region;
  const size_t scan_size = end - start + 1;
  for (size_t i = 0; i < scan_size; ++i) picked[i] = 0;
  for (int i = 0; i < cr; ++i)
    if (cos_angle(P(cl, start + i), P(cl, start + i + 1)) < cfg->blind_threshold) fill_n(&picked[i], cr + 1, L_BLIND_BLOCK);
  for (int i = 0; i < cr; ++i)
    if (cos_angle(P(cl, end - i), P(cl, end - i - 1)) < cfg->blind_threshold)
      fill_n(&picked[end - i - start - cr], cr + 1, L_BLIND_BLOCK);
  for (size_t i = start + cr; i < end - cr; ++i) {
    const float *prev = P(cl, i - 1), *pt = P(cl, i), *next = P(cl, i + 1);
    const float diff_next = sqdiff(next, pt);
    if (cos_angle(pt, next) < cfg->blind_threshold) {
      fill_n(&picked[i - start - cr + 1], cr * 2, L_BLIND_BLOCK);
      continue;
    }
    if (diff_next > 1.0) {
      const float depth1 = pdist(pt), depth2 = pdist(next);
      const float diff_prev = sqdiff(prev, pt);
      if (depth1 > depth2) {
        if (picked[i - start + 1] > L_NEAR_BLOCK && diff_prev / diff_next < 0.2) picked[i - start + 1] = L_EDGE_BROKEN;
        fill_n(&picked[i - start - cr + 1], cr, L_NEAR_BLOCK);
      } else {
        if (picked[i - start] > L_NEAR_BLOCK && diff_prev / diff_next < 0.2) picked[i - start] = L_EDGE_BROKEN;
        fill_n(&picked[i - start + 1], cr, L_NEAR_BLOCK);
      }
    }
  }
}

/* one half of pointClassify (:566-606 with sign=-1: points idx-0..idx-cr; :607-650 with sign=+1:
 * points idx+cr..idx+0, accumulated in that order) */
static int one_sided_line(const cloud_v *cl, size_t idx, int cr, int sign, float v[3]) {
  float c[3] = {0.f, 0.f, 0.f};
  for (int q = 0; q <= cr; ++q) {
    const int j = sign < 0 ? q : cr - q;  /* sign<0: idx-0, idx-1, ...; sign>0: idx+cr, ..., idx+0 */
    const float *p = P(cl, sign < 0 ? idx - (size_t)j : idx + (size_t)j);
    for (int d = 0; d < 3; ++d) c[d] += p[d];
  }
  for (int d = 0; d < 3; ++d) c[d] /= (float)(cr + 1);
  float A[9] = {0};
  for (int q = 0; q <= cr; ++q) {
    const int j = sign < 0 ? q : cr - q;
    const float *p = P(cl, sign < 0 ? idx - (size_t)j : idx + (size_t)j);
    const float a[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    A[0] += a[0] * a[0];
    A[3] += a[0] * a[1];
    A[6] += a[0] * a[2];
    A[4] += a[1] * a[1];
    A[7] += a[1] * a[2];
    A[8] += a[2] * a[2];
  }
  for (int k = 0; k < 9; ++k) A[k] /= (float)(cr + 1);
  float D[3], V[9];
  oracle_eig_sym3(A, D, V);
  if (!(D[2] > 100 * D[1] && D[2] > 10000 * D[0])) return 0;
  v[0] = V[2]; v[1] = V[5]; v[2] = V[8];
  const float vn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  for (int q = 0; q <= cr; ++q) {
    const int j = sign < 0 ? q : cr - q;
    const float *p = P(cl, sign < 0 ? idx - (size_t)j : idx + (size_t)j);
    const float a[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
    const float cx = a[1] * v[2] - a[2] * v[1], cy = a[2] * v[0] - a[0] * v[2], cz = a[0] * v[1] - a[1] * v[0];
    const float distance = sqrtf(cx * cx + cy * cy + cz * cz) / vn;
    if (fabs((double)distance) > 0.08) return 0;
  }
  return 1;
}

/* ScanRegistration.cpp:557-687 */
int oracle_point_classify(const float *cloud, size_t stride_floats, size_t idx, int curvature_region) {
  cloud_v cl = {cloud, stride_floats};
  float v1[3], v2[3];
  const int line1 = one_sided_line(&cl, idx, curvature_region, -1, v1);
  const int line2 = one_sided_line(&cl, idx, curvature_region, +1, v2);
  if (line1 && line2) {
    const float ab = v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2];
    const float dis = sqrtf(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]) * sqrtf(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
    const double diff = (double)(ab / dis);
    if (diff < cos(175.0 * OR_PI / 180.0) || diff > cos(5.0 * OR_PI / 180.0)) return L_SURFACE_FLAT;
    if (diff > cos(135.0 * OR_PI / 180.0) && diff < cos(45.0 * OR_PI / 180.0)) return L_CORNER_SHARP;
  }
  return (line1 || line2) ? L_ONESIDE_FLAT : L_MESSY;
}

typedef struct { float *p; size_t n; } out_cloud;
static void push(out_cloud *o, const float *pt, size_t cf) { /* toXYZI: intensity = curvature field */
  float *d = o->p + 4 * o->n++;
  d[0] = pt[0]; d[1] = pt[1]; d[2] = pt[2]; d[3] = pt[cf];
}

/* stable ascending merge sort on curvature with ties kept in index order (:151-186: `<=`) */
static void merge_sort(size_t *a, int first, int last, const float *curv, size_t *tmp) {
  if (first >= last) return;
  const int mid = (first + last) / 2;
  merge_sort(a, first, mid, curv, tmp);
  merge_sort(a, mid + 1, last, curv, tmp);
  int i = first, j = mid + 1, k = 0;
  while (i <= mid && j <= last) tmp[k++] = (curv[a[i]] <= curv[a[j]]) ? a[i++] : a[j++];
  while (i <= mid) tmp[k++] = a[i++];
  while (j <= last) tmp[k++] = a[j++];
  for (int q = 0; q < k; ++q) a[first + q] = tmp[q];
}

void oracle_extract_features(const float *cloud, size_t n_points, size_t stride_floats, size_t curvature_field,
                             const int32_t *scan_ranges, size_t n_scans, const oracle_reg_params *cfg,
                             float *sharp, float *less_sharp, float *flat, float *less_flat, size_t counts[4],
                             float *curvature_out, int8_t *picked_out, int8_t *label_out) {
  cloud_v cl = {cloud, stride_floats};
  out_cloud o_sharp = {sharp, 0}, o_less_sharp = {less_sharp, 0}, o_flat = {flat, 0}, o_less_flat = {less_flat, 0};
  const int cr = cfg->curvature_region, nf = cfg->n_feature_regions;
  int *picked = (int *)malloc(sizeof(int) * (n_points + 16));
  float *curv = (float *)malloc(sizeof(float) * (n_points + 16));
  int *rlabel = (int *)malloc(sizeof(int) * (n_points + 16));
  size_t *sorted = (size_t *)malloc(sizeof(size_t) * (n_points + 16));
  size_t *tmp = (size_t *)malloc(sizeof(size_t) * (n_points + 16));
  float *scan_less = (float *)malloc(sizeof(float) * 4 * (n_points + 16));
  float *scan_ds = (float *)malloc(sizeof(float) * 4 * (n_points + 16));
  if (curvature_out) for (size_t i = 0; i < n_points; ++i) curvature_out[i] = 0.0f;
  if (picked_out) memset(picked_out, 0, n_points);
  if (label_out) memset(label_out, L_UNKNOW, n_points);
  for (size_t s = 0; s < n_scans; ++s) {
    const size_t start = (size_t)scan_ranges[2 * s], end = (size_t)scan_ranges[2 * s + 1];
    if (end <= start + 2 * (size_t)cr) continue;  /* :205-207 */
    out_cloud o_scan = {scan_less, 0};
    set_scan_buffers(&cl, start, end, cfg, picked);
    if (picked_out) for (size_t i = start; i <= end; ++i) picked_out[i] = (int8_t)picked[i - start];
    for (int j = 0; j < nf; ++j) {
      const size_t sp = ((start + cr) * (size_t)(nf - j) + (end - cr) * (size_t)j) / (size_t)nf;
      const size_t ep = ((start + cr) * (size_t)(nf - 1 - j) + (end - cr) * (size_t)(j + 1)) / (size_t)nf - 1;
      if (ep <= sp) continue;
      const size_t rs = ep - sp + 1;
      /* setRegionBuffersFor, :427-469 */
      const float w = -2 * cr;
      for (size_t i = sp, r = 0; i <= ep; ++i, ++r) {
        const float *p = P(&cl, i);
        float dx = w * p[0], dy = w * p[1], dz = w * p[2];
        for (int q = 1; q <= cr; ++q) {
          const float *a = P(&cl, i + q), *b = P(&cl, i - q);
          dx += a[0] + b[0];
          dy += a[1] + b[1];
          dz += a[2] + b[2];
        }
        curv[r] = dx * dx + dy * dy + dz * dz;
        sorted[r] = r;
        rlabel[r] = L_UNKNOW;
        if (curvature_out) curvature_out[i] = curv[r];
      }
      merge_sort(sorted, 0, (int)rs - 1, curv, tmp);
      for (size_t r = 0; r < rs; ++r) sorted[r] += sp;
      /* flat surface features, :268-284 */
      int surf_picked = 0;
      for (size_t k = 0; k < rs && surf_picked < cfg->max_surface_flat; ++k) {
        const size_t idx = sorted[k], scan_idx = idx - start, ridx = idx - sp;
        if (picked[scan_idx] != L_SURF_PICKED_NEAR && curv[ridx] < cfg->surface_curvature_threshold) {
          ++surf_picked;
          rlabel[ridx] = L_SURFACE_FLAT;
          push(&o_flat, P(&cl, idx), curvature_field);
          picked[scan_idx] = L_SURF_PICKED_NEAR; /* markAsPicked, :533-555 */
          for (int q = 1; q <= cr; ++q) picked[scan_idx + q] = L_SURF_PICKED_NEAR;
          for (int q = 1; q <= cr; ++q) picked[scan_idx - q] = L_SURF_PICKED_NEAR;
        }
      }
      /* less flat + broken edges, :286-302 */
      for (size_t k = 0; k < rs; ++k) {
        const size_t idx = sp + k, scan_idx = idx - start;
        if (curv[k] < cfg->surface_curvature_threshold) {
          push(&o_scan, P(&cl, idx), curvature_field);
          if (rlabel[k] != L_SURFACE_FLAT) rlabel[k] = L_SURFACE_LESS_FLAT;
        }
        if (picked[scan_idx] == L_EDGE_BROKEN) {
          push(&o_sharp, P(&cl, idx), curvature_field);
          push(&o_less_sharp, P(&cl, idx), curvature_field);
          rlabel[k] = L_CORNER_SHARP;
        }
      }
      /* classified features in descending curvature, :304-354 */
      int corner_picked = 0;
      surf_picked = 0;
      for (size_t k = rs; k > 0;) {
        const size_t idx = sorted[--k], scan_idx = idx - start, ridx = idx - sp;
        if (curv[ridx] < cfg->surface_curvature_threshold) break;
        const int lab = oracle_point_classify(cloud, stride_floats, idx, cr);
        if (lab == L_SURFACE_FLAT) {
          rlabel[ridx] = L_SURFACE_FLAT;
          if (surf_picked < cfg->max_surface_flat) ++surf_picked;
          push(&o_scan, P(&cl, idx), curvature_field);
        } else if (lab == L_CORNER_SHARP) {
          if (picked[scan_idx] > L_EDGE_BROKEN) {
            rlabel[ridx] = L_CORNER_SHARP;
            if (corner_picked < cfg->max_corner_sharp) {
              ++corner_picked;
              push(&o_sharp, P(&cl, idx), curvature_field);
            }
            push(&o_less_sharp, P(&cl, idx), curvature_field);
          }
        } else if (lab == L_ONESIDE_FLAT) {
          rlabel[ridx] = L_ONESIDE_FLAT;
          if (surf_picked < cfg->max_surface_flat) {
            ++surf_picked;
            push(&o_flat, P(&cl, idx), curvature_field);
          }
          push(&o_scan, P(&cl, idx), curvature_field);
        }
      }
      if (label_out) for (size_t r = 0; r < rs; ++r) label_out[sp + r] = (int8_t)rlabel[r];
    }
    /* :398-407: VoxelGrid(lessFlatFilterSize) of this ring's less-flat points */
    const size_t m = oracle_voxel_grid(scan_less, o_scan.n, 4, cfg->less_flat_filter_size, scan_ds);
    memcpy(o_less_flat.p + 4 * o_less_flat.n, scan_ds, m * 4 * sizeof(float));
    o_less_flat.n += m;
  }
  counts[0] = o_sharp.n;
  counts[1] = o_less_sharp.n;
  counts[2] = o_flat.n;
  counts[3] = o_less_flat.n;
  free(picked); free(curv); free(rlabel); free(sorted); free(tmp); free(scan_less); free(scan_ds);
}

/* MultiScanRegistration::process (MultiScanRegistration.cpp:94-190) without the IMU branch
 * (hasIMUData() == false): axis swap, validity, ring from the vertical angle (linear mapper,
 * MultiScanRegistration.h:57-87), sweep-relative time from the horizontal angle with the
 * half-passed logic, per-ring clouds in arrival order.  out: {x, y, z, curvature = ring + relTime}. */
size_t oracle_multiscan_register(const float *in, size_t n, size_t stride_floats, float lower_deg, float upper_deg,
                                 int n_rings, float scan_period, float *out, int32_t *ranges) {
  const float factor = (n_rings - 1) / (upper_deg - lower_deg); /* :63 */
  int32_t *ring = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
  float *tmp = (float *)malloc(sizeof(float) * 4 * (n + 1));
  size_t *count = (size_t *)calloc((size_t)n_rings + 1, sizeof(size_t));
  const float *first = in, *last = in + (n - 1) * stride_floats;
  float start_ori = -atan2f(first[1], first[0]);
  float end_ori = -atan2f(last[1], last[0]) + 2 * (float)OR_PI;
  if (end_ori - start_ori > 3 * OR_PI) end_ori -= 2 * OR_PI;
  else if (end_ori - start_ori < OR_PI) end_ori += 2 * OR_PI;
  int half_passed = 0;
  for (size_t i = 0; i < n; ++i) {
    const float *p = in + i * stride_floats;
    const float x = p[1], y = p[2], z = p[0]; /* :127-129 */
    ring[i] = -1;
    if (!isfinite(x) || !isfinite(y) || !isfinite(z)) continue;
    if (x * x + y * y + z * z < 0.0001) continue;
    const float angle = atanf(y / sqrtf(x * x + z * z));
    const int scan_id = (int)(((angle * 180 / OR_PI) - lower_deg) * factor + 0.5);
    if (scan_id >= n_rings || scan_id < 0) continue;
    float ori = -atan2f(x, z);
    if (!half_passed) {
      if (ori < start_ori - OR_PI / 2) ori += 2 * OR_PI;
      else if (ori > start_ori + OR_PI * 3 / 2) ori -= 2 * OR_PI;
      if (ori - start_ori > OR_PI) half_passed = 1;
    } else {
      ori += 2 * OR_PI;
      if (ori < end_ori - OR_PI * 3 / 2) ori += 2 * OR_PI;
      else if (ori > end_ori + OR_PI / 2) ori -= 2 * OR_PI;
    }
    const float rel_time = scan_period * (ori - start_ori) / (end_ori - start_ori);
    ring[i] = scan_id;
    tmp[4 * i] = x; tmp[4 * i + 1] = y; tmp[4 * i + 2] = z;
    tmp[4 * i + 3] = scan_id + rel_time;
    count[scan_id]++;
  }
  size_t total = 0;
  size_t *fill = (size_t *)malloc(sizeof(size_t) * ((size_t)n_rings + 1));
  for (int r = 0; r < n_rings; ++r) { /* :180-190 */
    fill[r] = total;
    ranges[2 * r] = (int32_t)total;
    total += count[r];
    ranges[2 * r + 1] = total > 0 ? (int32_t)total - 1 : 0;
  }
  for (size_t i = 0; i < n; ++i)
    if (ring[i] >= 0) memcpy(out + 4 * fill[ring[i]]++, tmp + 4 * i, 4 * sizeof(float));
  free(ring); free(tmp); free(count); free(fill);
  return total;
}
