/*
 * fmap_oracle.c -- CPU ORACLE for map maintenance.  TEST INFRASTRUCTURE ONLY
 * (see lslam_oracle.h): nothing under the-cooper-mapper_amd/ links or calls this.
 *
 * Restates, single-threaded and in the reference's order of operations:
 *   - FeatureMap<PointXYZI>  util/FeatureMap.h:55-91 (ctor), :188-230 (push/add), :232-254 (update),
 *     :256-265 (getSurroundFeature), :267-286 (getFullMap), :288-306 (downsizeValidCloud),
 *     :307-352 (computeActiveAera), :353-377 (shift, including its swap-chain behaviour),
 *     :475-487 (worldToCube)
 *   - pcl::transformPointCloud (called at util/transform_utils.h:459): p' = R p + t per point
 *   - pcl::VoxelGrid<PointXYZI>::applyFilter with the defaults the reference leaves untouched
 *     (downsample_all_data = true, min_points_per_voxel = 0, is_dense input).
 *
 * PARITY PIN STATUS: "parity unpinned" for the VoxelGrid part.  PCL is not under /root/reference
 * and not installed; the version is not pinned by the reference.  What is restated is PCL's
 * published algorithm (voxel index = floor(p * (1/leaf)) - min_b, points grouped by the linear
 * index idx = i + j*div_x + k*div_x*div_y, output in ascending idx, one centroid per voxel over
 * x, y, z AND intensity).  PCL sorts the (idx, point) pairs with std::sort, which is not stable:
 * the order in which the points of one voxel are summed is unspecified there.  This restatement
 * (and the device code) sum them in input order; centroids can differ from a PCL build in the
 * last fp32 bits, membership and output order cannot.
 */
#include "fmap_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float *p; /* xyzi */
  size_t n, cap;
} cloud_t;

static void cloud_push(cloud_t *c, const float q[4]) {
  if (c->n == c->cap) {
    c->cap = c->cap ? 2 * c->cap : 16;
    c->p = (float *)realloc(c->p, c->cap * 4 * sizeof(float));
  }
  memcpy(c->p + 4 * c->n, q, 4 * sizeof(float));
  c->n++;
}
static void cloud_clear(cloud_t *c) { c->n = 0; }
static void cloud_free(cloud_t *c) {
  free(c->p);
  c->p = NULL;
  c->n = c->cap = 0;
}

/* ---------------- pcl::VoxelGrid::applyFilter ---------------- */

typedef struct {
  int idx;
  unsigned pos;
} vg_pair;

static int vg_cmp(const void *a, const void *b) {
  const vg_pair *x = (const vg_pair *)a, *y = (const vg_pair *)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0); /* input order inside a voxel */
}

size_t oracle_voxel_grid(const float *in, size_t n, size_t stride_floats, float leaf, float *out) {
  if (n == 0) return 0;
  const float inv = 1.0f / leaf; /* inverse_leaf_size_ = 1 / leaf_size_ */
  float mn[3], mx[3];
  for (int d = 0; d < 3; ++d) mn[d] = mx[d] = in[d];
  for (size_t i = 1; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      const float v = in[i * stride_floats + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  /* overflow guard of applyFilter: too many voxels -> the input is returned unfiltered */
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
                  dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (long long)INT_MAX) {
    for (size_t i = 0; i < n; ++i) memcpy(out + 4 * i, in + i * stride_floats, 4 * sizeof(float));
    return n;
  }
  int min_b[3], max_b[3], div_b[3];
  for (int d = 0; d < 3; ++d) {
    min_b[d] = (int)floorf(mn[d] * inv);
    max_b[d] = (int)floorf(mx[d] * inv);
    div_b[d] = max_b[d] - min_b[d] + 1;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  vg_pair *pr = (vg_pair *)malloc(n * sizeof(vg_pair));
  for (size_t i = 0; i < n; ++i) {
    const float *p = in + i * stride_floats;
    const int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
    const int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
    const int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
    pr[i].idx = i0 * mul[0] + i1 * mul[1] + i2 * mul[2];
    pr[i].pos = (unsigned)i;
  }
  qsort(pr, n, sizeof(vg_pair), vg_cmp);
  size_t m = 0;
  for (size_t a = 0; a < n;) {
    size_t b = a;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    while (b < n && pr[b].idx == pr[a].idx) {
      const float *p = in + (size_t)pr[b].pos * stride_floats;
      for (int d = 0; d < 4; ++d) s[d] += p[d];
      ++b;
    }
    const float cnt = (float)(b - a);
    for (int d = 0; d < 4; ++d) out[4 * m + d] = s[d] / cnt;
    ++m;
    a = b;
  }
  free(pr);
  return m;
}

/* ---------------- FeatureMap ---------------- */

struct oracle_fmap {
  int W, H, D, ncube;
  int origin[3];
  int cur[3];
  float cube_size, valid_dist;
  float leaf_corner, leaf_surf, leaf_map;
  cloud_t **cube[2]; /* arrays of POINTERS: shift() swaps the pointers (FeatureMap.h:364-367) */
  int *valid;
  size_t n_valid;
};

static int is_valid(const oracle_fmap *f, int i, int j, int k) {
  return 0 <= i && i < f->W && 0 <= j && j < f->H && 0 <= k && k < f->D;
}
static int to_index(const oracle_fmap *f, int i, int j, int k) { return i + j * f->W + k * f->W * f->H; }

oracle_fmap *oracle_fmap_create(int w, int h, int d) {
  oracle_fmap *f = (oracle_fmap *)calloc(1, sizeof(*f));
  f->W = w;
  f->H = h;
  f->D = d;
  f->ncube = w * h * d;
  /* FeatureMap.h:60-62: origin = round(--size / 2.0) */
  f->origin[0] = (int)round((w - 1) / 2.0);
  f->origin[1] = (int)round((h - 1) / 2.0);
  f->origin[2] = (int)round((d - 1) / 2.0);
  f->cube_size = 50.0f;
  f->valid_dist = 150.0f;
  f->leaf_corner = 0.2f;
  f->leaf_surf = 0.2f;
  f->leaf_map = 0.6f;
  for (int t = 0; t < 2; ++t) {
    f->cube[t] = (cloud_t **)calloc((size_t)f->ncube, sizeof(cloud_t *));
    for (int c = 0; c < f->ncube; ++c) f->cube[t][c] = (cloud_t *)calloc(1, sizeof(cloud_t));
  }
  f->valid = (int *)malloc(sizeof(int) * (size_t)f->ncube);
  return f;
}

void oracle_fmap_free(oracle_fmap *f) {
  if (!f) return;
  for (int t = 0; t < 2; ++t) {
    for (int c = 0; c < f->ncube; ++c) {
      cloud_free(f->cube[t][c]);
      free(f->cube[t][c]);
    }
    free(f->cube[t]);
  }
  free(f->valid);
  free(f);
}

void oracle_fmap_setup_filter_size(oracle_fmap *f, float corner, float surf, float map) {
  f->leaf_corner = corner;
  f->leaf_surf = surf;
  f->leaf_map = map;
}
void oracle_fmap_setup_cube_size(oracle_fmap *f, float s) { f->cube_size = s; }
void oracle_fmap_setup_valid_distance(oracle_fmap *f, float d) { f->valid_dist = d; }
void oracle_fmap_origin(const oracle_fmap *f, int32_t out[3]) {
  for (int d = 0; d < 3; ++d) out[d] = f->origin[d];
}

/* FeatureMap.h:475-487 */
static int world_to_cube(const oracle_fmap *f, const float p[3], int g[3]) {
  for (int d = 0; d < 3; ++d) g[d] = (int)(roundf(p[d] / f->cube_size) + (float)f->origin[d]);
  return is_valid(f, g[0], g[1], g[2]);
}

/* FeatureMap.h:353-377 */
static void shift(oracle_fmap *f, int di, int dj, int dk) {
  if (di == 0 && dj == 0 && dk == 0) return;
  for (int i = 0; i < f->W; ++i)
    for (int j = 0; j < f->H; ++j)
      for (int k = 0; k < f->D; ++k) {
        const int oi = i - di, oj = j - dj, ok = k - dk;
        const int a = to_index(f, i, j, k);
        if (is_valid(f, oi, oj, ok)) {
          const int b = to_index(f, oi, oj, ok);
          for (int t = 0; t < 2; ++t) {
            cloud_t *tmp = f->cube[t][a];
            f->cube[t][a] = f->cube[t][b];
            f->cube[t][b] = tmp;
          }
        } else {
          cloud_clear(f->cube[0][a]);
          cloud_clear(f->cube[1][a]);
        }
      }
}

/* FeatureMap.h:307-352 */
static void compute_active_area(oracle_fmap *f, const float pos[3]) {
  f->n_valid = 0;
  const int win = (int)ceil(f->valid_dist / f->cube_size);
  for (int i = f->cur[0] - win; i <= f->cur[0] + win; ++i)
    for (int j = f->cur[1] - win; j <= f->cur[1] + win; ++j)
      for (int k = f->cur[2] - win; k <= f->cur[2] + win; ++k) {
        if (!is_valid(f, i, j, k)) continue;
        const float cx = f->cube_size * (float)(i - f->origin[0]);
        const float cy = f->cube_size * (float)(j - f->origin[1]);
        const float cz = f->cube_size * (float)(k - f->origin[2]);
        int in_fov = 0;
        for (int ii = -1; ii <= 1 && !in_fov; ii += 2)
          for (int jj = -1; jj <= 1 && !in_fov; jj += 2)
            for (int kk = -1; kk <= 1 && !in_fov; kk += 2) {
              const float x = (float)((double)cx + (double)f->cube_size / 2.0 * ii);
              const float y = (float)((double)cy + (double)f->cube_size / 2.0 * jj);
              const float z = (float)((double)cz + (double)f->cube_size / 2.0 * kk);
              const float ddx = pos[0] - x, ddy = pos[1] - y, ddz = pos[2] - z;
              const float sq = ddx * ddx + ddy * ddy + ddz * ddz;
              if (sqrt((double)sq) < (double)f->valid_dist) in_fov = 1;
            }
        if (in_fov) f->valid[f->n_valid++] = to_index(f, i, j, k);
      }
}

/* FeatureMap.h:232-254 */
void oracle_fmap_update(oracle_fmap *f, const float pos[3]) {
  int g[3];
  world_to_cube(f, pos, g);
  const int PAD = 3;
  const int lim[3] = {f->W, f->H, f->D};
  int ng[3];
  for (int d = 0; d < 3; ++d) {
    int v = g[d] > PAD ? g[d] : PAD;
    const int hi = lim[d] - PAD - 1;
    ng[d] = v < hi ? v : hi;
  }
  shift(f, ng[0] - g[0], ng[1] - g[1], ng[2] - g[2]);
  for (int d = 0; d < 3; ++d) {
    f->origin[d] += ng[d] - g[d];
    f->cur[d] = ng[d];
  }
  compute_active_area(f, pos);
}

size_t oracle_fmap_valid_cubes(const oracle_fmap *f, int32_t *out, size_t cap) {
  for (size_t i = 0; i < f->n_valid && i < cap; ++i) out[i] = f->valid[i];
  return f->n_valid;
}

/* FeatureMap.h:288-306 */
static void downsize_valid(oracle_fmap *f) {
  for (size_t v = 0; v < f->n_valid; ++v) {
    const int c = f->valid[v];
    for (int t = 0; t < 2; ++t) {
      cloud_t *src = f->cube[t][c];
      if (src->n == 0) continue; /* VoxelGrid of an empty cloud is an empty cloud */
      float *tmp = (float *)malloc(src->n * 4 * sizeof(float));
      const size_t m = oracle_voxel_grid(src->p, src->n, 4, t == 0 ? f->leaf_corner : f->leaf_surf, tmp);
      memcpy(src->p, tmp, m * 4 * sizeof(float));
      src->n = m;
      free(tmp);
    }
  }
}

/* FeatureMap.h:218-230 + :188-217 ; T = row-major 4x4 of the Isometry3f */
void oracle_fmap_add_feature_cloud(oracle_fmap *f, const float *corner, size_t nc, const float *surf,
                                   size_t ns, size_t stride_floats, const float T[16]) {
  const float *src[2] = {corner, surf};
  const size_t cnt[2] = {nc, ns};
  for (int t = 0; t < 2; ++t)
    for (size_t i = 0; i < cnt[t]; ++i) {
      const float *p = src[t] + i * stride_floats;
      float q[4];
      for (int r = 0; r < 3; ++r) q[r] = ((T[4 * r] * p[0] + T[4 * r + 1] * p[1]) + T[4 * r + 2] * p[2]) + T[4 * r + 3];
      q[3] = p[3];
      int g[3];
      if (world_to_cube(f, q, g)) cloud_push(f->cube[t][to_index(f, g[0], g[1], g[2])], q);
    }
  downsize_valid(f);
}

/* FeatureMap.h:256-265 */
size_t oracle_fmap_get_surround(const oracle_fmap *f, int which, float *out, size_t cap) {
  size_t n = 0;
  for (size_t v = 0; v < f->n_valid; ++v) {
    const cloud_t *c = f->cube[which][f->valid[v]];
    if (out && c->n && n + c->n <= cap) memcpy(out + 4 * n, c->p, c->n * 4 * sizeof(float));
    n += c->n;
  }
  return n;
}

size_t oracle_fmap_cube_count(const oracle_fmap *f, int which, int cube) { return f->cube[which][cube]->n; }

/* FeatureMap.h:267-286 */
size_t oracle_fmap_get_full_map(const oracle_fmap *f, float *out, size_t cap) {
  size_t n = 0;
  for (int c = 0; c < f->ncube; ++c)
    for (int t = 0; t < 2; ++t) {
      const cloud_t *src = f->cube[t][c];
      if (src->n == 0) continue;
      float *tmp = (float *)malloc(src->n * 4 * sizeof(float));
      const size_t m = oracle_voxel_grid(src->p, src->n, 4, f->leaf_map, tmp);
      if (out && n + m <= cap) memcpy(out + 4 * n, tmp, m * 4 * sizeof(float));
      n += m;
      free(tmp);
    }
  return n;
}
