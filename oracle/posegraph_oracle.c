/*
 * posegraph_oracle.c -- CPU ORACLE / CPU BASELINE of the SE(3) pose-graph Levenberg-Marquardt.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under the-cooper-mapper_amd/ includes, links or calls this file.  Only tests/ and
 * bench.py's `cpu_baseline` leg load it (oracle/libposegraph_oracle.so).
 *
 * PARITY UNPINNED.  The reference hands this arithmetic to g2o (/root/reference/L_SLAM/src/pose_graph/solver_g2o.cpp:5,16,
 * 51-95: VertexSE3, EdgeSE3, "lm_var", the csparse solver library forced at :5), which is neither under /root/reference
 * nor pinned by it nor installed here.  What is restated is g2o's PUBLISHED method, the way its CSparse linear solver runs it
 * on one core -- so that the GPU solver has a compiled direct solver timed beside it instead of a numpy script:
 *
 *   * conventions exactly as oracle/posegraph_oracle.py (which documents them and is this file's own check: the two
 *     agree to ~1e-9 on H, b, chi2 and to 1e-6 m over whole LM runs, tests/test_oracle_posegraph_c.py):
 *     X <- X * fromVectorMQT(d); e = toVectorMQT(Z^-1 Xi^-1 Xj); chi2 = sum e^T Omega e; LM schedule of
 *     OptimizationAlgorithmLevenberg (lambda0 = 1e-5 max diag H, rho test, x1/3..2/3 / x nu doubling, <= 10 trials);
 *     first vertex fixed (solver_g2o.cpp:55-59); information matrices as pose_graph/graph.cpp:279-288,333-339 builds them;
 *   * ANALYTIC Jacobians of the edge error (g2o's EdgeSE3 uses analytic ones too);
 *   * the damped normal equations (H + lambda I) dx = b solved by a SPARSE BLOCK CHOLESKY: the 6x6-block matrix is
 *     reordered by reverse Cuthill-McKee (a profile-reducing order: what a chain-with-loop-closures graph wants) and
 *     factored in its envelope (Jennings' profile method), every block row a dense 6 x 6w panel so that the inner
 *     kernel is a 6 x 6w by 6w x 6 product of contiguous rows; forward / backward substitution on the same panels.
 *     g2o + CSparse use an AMD order and a column factorisation: same factor up to the order, same class of cost.
 *
 * Plain C99, fp64, one thread.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  int iterations;      /* LM iterations performed */
  int trials;          /* damped solves (accepted + rejected) */
  int status;          /* 0 ok, -1 factorisation failed (matrix not positive definite) */
  int bandwidth;       /* widest block row of the envelope */
  long long env_blocks; /* 6x6 blocks held by the factor */
  double chi2_initial, chi2_final, lambda;
  double t_linearize, t_factor, t_solve, t_chi2, t_order, t_total; /* seconds */
  double factor_flops; /* multiply-adds x 2 of one factorisation */
} pgo_stats;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- SE(3) helpers; quaternions [x, y, z, w] ---------------------------------------------------------------------- */
static void qmul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
static void qconj(const double q[4], double o[4]) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
static void qmat(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static void skew(const double v[3], double S[9]) {
  S[0] = 0; S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2]; S[4] = 0; S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}

/* e = toVectorMQT(Z^-1 Xi^-1 Xj) and, when Ji != NULL, de/d(delta_i), de/d(delta_j) (row-major 6x6) for the
 * right-multiplied updates X <- X * fromVectorMQT(delta).  With T_b = Xi^-1 Xj = (t_b, q_b), E = Z^-1 T_b = (t_e, q_e):
 *   delta_j:  E' = E * D            -> d t_e = R_e dt,          d q_e.v = (w_e I + [v_e]x) dq
 *   delta_i:  E' = Z^-1 D^-1 T_b    -> d t_e = -Rz^T dt + Rz^T 2 [t_b]x dq   (D^-1 ~ (-dt, -dq), rotation angle 2 dq)
 *                                      d q_e = qz* (x) (-dq, 1) (x) q_b, linear in dq
 * q_e is normalised and its sign chosen so that w_e >= 0 (EdgeSE3::computeError via toVectorMQT). */
static void edge_error(const double *xi, const double *xj, const double *z, double e[6], double *Ji, double *Jj) {
  double Ri[9], Rz[9], qic[4], qzc[4], qb[4], qe[4];
  qmat(xi + 3, Ri);
  qmat(z + 3, Rz);
  const double d[3] = {xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2]};
  double tb[3], te[3];
  for (int r = 0; r < 3; ++r) tb[r] = Ri[r] * d[0] + Ri[3 + r] * d[1] + Ri[6 + r] * d[2]; /* Ri^T d */
  qconj(xi + 3, qic);
  qmul(qic, xj + 3, qb);
  const double u[3] = {tb[0] - z[0], tb[1] - z[1], tb[2] - z[2]};
  for (int r = 0; r < 3; ++r) te[r] = Rz[r] * u[0] + Rz[3 + r] * u[1] + Rz[6 + r] * u[2]; /* Rz^T (t_b - t_z) */
  qconj(z + 3, qzc);
  qmul(qzc, qb, qe);
  const double nrm = sqrt(qe[0] * qe[0] + qe[1] * qe[1] + qe[2] * qe[2] + qe[3] * qe[3]);
  const double sgn = qe[3] < 0 ? -1.0 : 1.0;
  for (int k = 0; k < 4; ++k) qe[k] = sgn * qe[k] / nrm;
  e[0] = te[0]; e[1] = te[1]; e[2] = te[2];
  e[3] = qe[0]; e[4] = qe[1]; e[5] = qe[2];
  if (!Ji) return;
  double Re[9], Sv[9], Stb[9], Sb[9], Sz[9], M[9];
  qmat(qe, Re);
  skew(qe, Sv);
  memset(Ji, 0, 36 * sizeof(double));
  memset(Jj, 0, 36 * sizeof(double));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      Jj[r * 6 + c] = Re[r * 3 + c];
      Jj[(3 + r) * 6 + 3 + c] = (r == c ? qe[3] : 0.0) + Sv[r * 3 + c];
    }
  skew(tb, Stb);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      Ji[r * 6 + c] = -Rz[c * 3 + r];
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += Rz[k * 3 + r] * 2.0 * Stb[k * 3 + c];
      Ji[r * 6 + 3 + c] = s;
    }
  /* vector part of qz* (x) (-dq, 1) (x) qb  =  -( wz (wb I - [vb]x) + vz vb^T + [vz]x (wb I - [vb]x) ) dq + const */
  skew(qb, Sb);
  skew(z + 3, Sz);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) M[r * 3 + c] = (r == c ? -qb[3] : 0.0) + Sb[r * 3 + c];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = z[6] * M[r * 3 + c] - z[3 + r] * qb[c];
      for (int k = 0; k < 3; ++k) s -= Sz[r * 3 + k] * M[k * 3 + c];
      Ji[(3 + r) * 6 + 3 + c] = sgn * s / nrm;
    }
}

static double edge_chi2(const double e[6], const double *Om) {
  double c2 = 0.0;
  for (int r = 0; r < 6; ++r) {
    double s = 0.0;
    for (int c = 0; c < 6; ++c) s += Om[r * 6 + c] * e[c];
    c2 += e[r] * s;
  }
  return c2;
}

double pgo_chi2(int n_e, const double *poses, const int32_t *ij, const double *meas, const double *info) {
  double c2 = 0.0;
  for (int k = 0; k < n_e; ++k) {
    double e[6];
    edge_error(poses + 7 * ij[2 * k], poses + 7 * ij[2 * k + 1], meas + 7 * k, e, NULL, NULL);
    c2 += edge_chi2(e, info + 36 * (size_t)k);
  }
  return c2;
}

/* X_v <- X_v * fromVectorMQT(dx_v), free vertices only; quaternion renormalised (as posegraph_oracle.oplus) */
static void oplus(int n_v, const double *poses, const double *dx, int fixed, double *out) {
  for (int v = 0; v < n_v; ++v) {
    const double *p = poses + 7 * v, *d = dx + 6 * v;
    double *o = out + 7 * v;
    if (v == fixed) { memcpy(o, p, 7 * sizeof(double)); continue; }
    const double w2 = 1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    double dq[4] = {d[3], d[4], d[5], sqrt(w2 > 0 ? w2 : 0.0)};
    if (w2 < 0) { dq[0] = dq[1] = dq[2] = 0.0; dq[3] = 1.0; }
    double R[9];
    qmat(p + 3, R);
    for (int r = 0; r < 3; ++r) o[r] = p[r] + R[r * 3] * d[0] + R[r * 3 + 1] * d[1] + R[r * 3 + 2] * d[2];
    qmul(p + 3, dq, o + 3);
    const double n = sqrt(o[3] * o[3] + o[4] * o[4] + o[5] * o[5] + o[6] * o[6]);
    for (int k = 3; k < 7; ++k) o[k] /= n;
  }
}

/* ---- reverse Cuthill-McKee on the vertex graph -------------------------------------------------------------------- */
typedef struct { int n; int *ptr, *adj, *deg; } graph_t;

static void graph_build(graph_t *g, int n_v, int n_e, const int32_t *ij) {
  g->n = n_v;
  g->ptr = (int *)calloc((size_t)n_v + 1, sizeof(int));
  g->deg = (int *)calloc((size_t)n_v, sizeof(int));
  for (int k = 0; k < n_e; ++k)
    if (ij[2 * k] != ij[2 * k + 1]) { g->ptr[ij[2 * k] + 1]++; g->ptr[ij[2 * k + 1] + 1]++; }
  for (int v = 0; v < n_v; ++v) g->ptr[v + 1] += g->ptr[v];
  g->adj = (int *)malloc(sizeof(int) * (size_t)(g->ptr[n_v] > 0 ? g->ptr[n_v] : 1));
  int *fill = (int *)calloc((size_t)n_v, sizeof(int));
  for (int k = 0; k < n_e; ++k) {
    const int a = ij[2 * k], b = ij[2 * k + 1];
    if (a == b) continue;
    g->adj[g->ptr[a] + fill[a]++] = b;
    g->adj[g->ptr[b] + fill[b]++] = a;
  }
  for (int v = 0; v < n_v; ++v) g->deg[v] = g->ptr[v + 1] - g->ptr[v]; /* multi-edges counted: only a sort key */
  free(fill);
}
static void graph_free(graph_t *g) { free(g->ptr); free(g->adj); free(g->deg); }

static const int *g_deg_for_sort;
static int cmp_deg(const void *a, const void *b) {
  const int x = *(const int *)a, y = *(const int *)b;
  if (g_deg_for_sort[x] != g_deg_for_sort[y]) return g_deg_for_sort[x] < g_deg_for_sort[y] ? -1 : 1;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* breadth-first levels from s over unvisited (mark[v] == 0 or == stamp) vertices; returns the last vertex reached */
static int bfs_far(const graph_t *g, int s, int *mark, int stamp, int *queue, int *n_out, int *depth_out) {
  int head = 0, tail = 0;
  queue[tail++] = s;
  mark[s] = stamp;
  int *lvl = (int *)calloc((size_t)g->n, sizeof(int));
  int last = s;
  while (head < tail) {
    const int v = queue[head++];
    last = v;
    for (int a = g->ptr[v]; a < g->ptr[v + 1]; ++a) {
      const int w = g->adj[a];
      if (mark[w] != stamp && mark[w] <= 0) { mark[w] = stamp; lvl[w] = lvl[v] + 1; queue[tail++] = w; }
    }
  }
  /* among the vertices of the last level take the one of lowest degree */
  const int dl = lvl[last];
  for (int k = tail - 1; k >= 0 && lvl[queue[k]] == dl; --k)
    if (g->deg[queue[k]] < g->deg[last]) last = queue[k];
  *n_out = tail;
  *depth_out = dl;
  free(lvl);
  return last;
}

/* perm[new] = old */
static void rcm_order(int n_v, int n_e, const int32_t *ij, int *perm) {
  graph_t g;
  graph_build(&g, n_v, n_e, ij);
  int *mark = (int *)calloc((size_t)n_v, sizeof(int)); /* 0 unvisited, -k probing stamp, 1 ordered */
  int *queue = (int *)malloc(sizeof(int) * (size_t)n_v);
  int n_done = 0, stamp = -1;
  g_deg_for_sort = g.deg;
  for (int s0 = 0; s0 < n_v; ++s0) {
    if (mark[s0] == 1) continue;
    /* pseudo-peripheral start of this component: a few far-vertex sweeps */
    int s = s0, depth = -1, cnt = 0;
    for (int pass = 0; pass < 4; ++pass) {
      int d;
      const int far = bfs_far(&g, s, mark, stamp, queue, &cnt, &d);
      for (int k = 0; k < cnt; ++k) mark[queue[k]] = 0;
      --stamp;
      if (d <= depth) break;
      depth = d;
      s = far;
    }
    /* Cuthill-McKee from s, neighbours in order of increasing degree */
    int head = n_done, tail = n_done;
    perm[tail++] = s;
    mark[s] = 1;
    while (head < tail) {
      const int v = perm[head++];
      const int t0 = tail;
      for (int a = g.ptr[v]; a < g.ptr[v + 1]; ++a) {
        const int w = g.adj[a];
        if (mark[w] != 1) { mark[w] = 1; perm[tail++] = w; }
      }
      qsort(perm + t0, (size_t)(tail - t0), sizeof(int), cmp_deg);
    }
    n_done = tail;
  }
  for (int a = 0, b = n_v - 1; a < b; ++a, --b) { const int t = perm[a]; perm[a] = perm[b]; perm[b] = t; }
  free(mark);
  free(queue);
  graph_free(&g);
}

/* ---- envelope (profile) storage of the permuted block matrix ------------------------------------------------------ */
/* block row i (permuted index) holds columns first[i]..i as a dense 6 x 6 w panel, row-major: entry (r, 6 (j - first) + c) */
typedef struct {
  int n;
  int *perm, *inv, *first;
  size_t *off; /* panel start (in doubles) */
  int *width;
  double *A;   /* assembled H (lower envelope) */
  double *L;   /* factor of H + lambda I */
  size_t n_doubles;
  long long env_blocks;
  int bandwidth;
  double flops;
} env_t;

static void env_build(env_t *E, int n_v, int n_e, const int32_t *ij) {
  E->n = n_v;
  E->perm = (int *)malloc(sizeof(int) * (size_t)n_v);
  E->inv = (int *)malloc(sizeof(int) * (size_t)n_v);
  E->first = (int *)malloc(sizeof(int) * (size_t)n_v);
  E->width = (int *)malloc(sizeof(int) * (size_t)n_v);
  E->off = (size_t *)malloc(sizeof(size_t) * ((size_t)n_v + 1));
  rcm_order(n_v, n_e, ij, E->perm);
  for (int k = 0; k < n_v; ++k) E->inv[E->perm[k]] = k;
  for (int k = 0; k < n_v; ++k) E->first[k] = k;
  for (int k = 0; k < n_e; ++k) {
    const int a = E->inv[ij[2 * k]], b = E->inv[ij[2 * k + 1]];
    const int hi = a > b ? a : b, lo = a > b ? b : a;
    if (lo < E->first[hi]) E->first[hi] = lo;
  }
  E->off[0] = 0;
  E->env_blocks = 0;
  E->bandwidth = 0;
  E->flops = 0.0;
  for (int i = 0; i < n_v; ++i) {
    E->width[i] = i - E->first[i] + 1;
    E->off[i + 1] = E->off[i] + (size_t)36 * (size_t)E->width[i];
    E->env_blocks += E->width[i];
    if (E->width[i] > E->bandwidth) E->bandwidth = E->width[i];
  }
  /* multiply-adds of the factorisation: block (i, j) costs 216 per common predecessor column */
  for (int i = 0; i < n_v; ++i)
    for (int j = E->first[i]; j <= i; ++j) {
      const int k0 = E->first[i] > E->first[j] ? E->first[i] : E->first[j];
      E->flops += 2.0 * 216.0 * (double)(j - k0 + 1);
    }
  E->n_doubles = E->off[n_v];
  E->A = (double *)malloc(sizeof(double) * E->n_doubles);
  E->L = (double *)malloc(sizeof(double) * E->n_doubles);
}
static void env_free(env_t *E) {
  free(E->perm); free(E->inv); free(E->first); free(E->width); free(E->off); free(E->A); free(E->L);
}
/* pointer to entry (0, 0) of block (i, j), j in [first[i], i]; the panel's row stride is 6 * width[i] */
static double *env_blk(const env_t *E, double *base, int i, int j) { return base + E->off[i] + (size_t)6 * (size_t)(j - E->first[i]); }

/* ---- linearisation into the envelope ------------------------------------------------------------------------------ */
static double linearize(const env_t *E, int n_v, const double *poses, int n_e, const int32_t *ij, const double *meas,
                        const double *info, int fixed, double *b /* [6 n_v], original vertex order */) {
  memset(E->A, 0, sizeof(double) * E->n_doubles);
  memset(b, 0, sizeof(double) * 6 * (size_t)n_v);
  double chi2 = 0.0;
  for (int k = 0; k < n_e; ++k) {
    const int vi = ij[2 * k], vj = ij[2 * k + 1];
    double e[6], Ji[36], Jj[36], Ai[36], Aj[36];
    const double *Om = info + 36 * (size_t)k;
    edge_error(poses + 7 * vi, poses + 7 * vj, meas + 7 * k, e, Ji, Jj);
    chi2 += edge_chi2(e, Om);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) {
        double si = 0.0, sj = 0.0;
        for (int m = 0; m < 6; ++m) { si += Ji[m * 6 + r] * Om[m * 6 + c]; sj += Jj[m * 6 + r] * Om[m * 6 + c]; }
        Ai[r * 6 + c] = si; /* Ji^T Omega */
        Aj[r * 6 + c] = sj;
      }
    const int pi = E->inv[vi], pj = E->inv[vj];
    const size_t si_ = (size_t)6 * (size_t)E->width[pi], sj_ = (size_t)6 * (size_t)E->width[pj];
    double *Dii = env_blk(E, E->A, pi, pi), *Djj = env_blk(E, E->A, pj, pj);
    double *Off = pi > pj ? env_blk(E, E->A, pi, pj) : env_blk(E, E->A, pj, pi); /* lower triangle: (max, min) */
    const size_t so = pi > pj ? si_ : sj_;
    for (int r = 0; r < 6; ++r) {
      double bi = 0.0, bj = 0.0;
      for (int c = 0; c < 6; ++c) {
        double hii = 0.0, hjj = 0.0, hij = 0.0;
        for (int m = 0; m < 6; ++m) {
          hii += Ai[r * 6 + m] * Ji[m * 6 + c];
          hjj += Aj[r * 6 + m] * Jj[m * 6 + c];
          hij += Ai[r * 6 + m] * Jj[m * 6 + c];
        }
        if (vi != fixed) Dii[(size_t)r * si_ + c] += hii;
        if (vj != fixed) Djj[(size_t)r * sj_ + c] += hjj;
        if (vi != fixed && vj != fixed) {
          if (pi > pj) Off[(size_t)r * so + c] += hij; /* block (i, j) = Ji^T Om Jj */
          else Off[(size_t)c * so + r] += hij;         /* block (j, i) = its transpose */
        }
        bi += Ai[r * 6 + c] * e[c];
        bj += Aj[r * 6 + c] * e[c];
      }
      if (vi != fixed) b[6 * vi + r] -= bi;
      if (vj != fixed) b[6 * vj + r] -= bj;
    }
  }
  if (fixed >= 0 && fixed < n_v) { /* identity row: dx_fixed = 0 */
    const int pf = E->inv[fixed];
    double *D = env_blk(E, E->A, pf, pf);
    const size_t s = (size_t)6 * (size_t)E->width[pf];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) D[(size_t)r * s + c] = r == c ? 1.0 : 0.0;
  }
  return chi2;
}

/* ---- block envelope Cholesky: L L^T = A + lambda I (lambda not added to the fixed vertex's identity block) --------- */
static int factorize(env_t *E, double lambda, int fixed) {
  const int n = E->n;
  memcpy(E->L, E->A, sizeof(double) * E->n_doubles);
  const int pf = (fixed >= 0 && fixed < n) ? E->inv[fixed] : -1;
  for (int i = 0; i < n; ++i) {
    const int fi = E->first[i];
    const size_t si = (size_t)6 * (size_t)E->width[i];
    double *Pi = E->L + E->off[i];
    if (i != pf)
      for (int r = 0; r < 6; ++r) Pi[(size_t)r * si + 6 * (size_t)(i - fi) + r] += lambda;
    for (int j = fi; j <= i; ++j) {
      const int fj = E->first[j];
      const size_t sj = (size_t)6 * (size_t)E->width[j];
      const double *Pj = E->L + E->off[j];
      const int k0 = fi > fj ? fi : fj;
      const size_t len = (size_t)6 * (size_t)(j - k0); /* scalar columns k0..j-1 shared by rows i and j */
      const double *ai = Pi + 6 * (size_t)(k0 - fi), *aj = Pj + 6 * (size_t)(k0 - fj);
      double *S = Pi + 6 * (size_t)(j - fi);
      if (len) { /* 6 x len by len x 6: row r of panel i against the six rows of panel j, six vector accumulators */
        const double *y0 = aj, *y1 = aj + sj, *y2 = aj + 2 * sj, *y3 = aj + 3 * sj, *y4 = aj + 4 * sj, *y5 = aj + 5 * sj;
        for (int r = 0; r < 6; ++r) {
          const double *x = ai + (size_t)r * si;
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0;
#pragma omp simd reduction(+ : s0, s1, s2, s3, s4, s5)
          for (size_t q = 0; q < len; ++q) {
            const double xv = x[q];
            s0 += xv * y0[q]; s1 += xv * y1[q]; s2 += xv * y2[q];
            s3 += xv * y3[q]; s4 += xv * y4[q]; s5 += xv * y5[q];
          }
          const double sv[6] = {s0, s1, s2, s3, s4, s5};
          for (int c = 0; c < (j < i ? 6 : r + 1); ++c) S[(size_t)r * si + c] -= sv[c];
        }
      }
      if (j < i) { /* L_ij = S L_jj^-T : rows of S against the lower-triangular L_jj */
        const double *Ljj = Pj + 6 * (size_t)(j - fj);
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) {
            double s = S[(size_t)r * si + c];
            for (int m = 0; m < c; ++m) s -= S[(size_t)r * si + m] * Ljj[(size_t)c * sj + m];
            S[(size_t)r * si + c] = s / Ljj[(size_t)c * sj + c];
          }
      } else { /* diagonal block: in-place dense Cholesky of the lower triangle, upper part zeroed */
        for (int r = 0; r < 6; ++r) {
          for (int c = 0; c <= r; ++c) {
            double s = S[(size_t)r * si + c];
            for (int m = 0; m < c; ++m) s -= S[(size_t)r * si + m] * S[(size_t)c * si + m];
            if (r == c) {
              if (!(s > 0.0)) return -1;
              S[(size_t)r * si + c] = sqrt(s);
            } else {
              S[(size_t)r * si + c] = s / S[(size_t)c * si + c];
            }
          }
          for (int c = r + 1; c < 6; ++c) S[(size_t)r * si + c] = 0.0;
        }
      }
    }
  }
  return 0;
}

/* dx (original vertex order) = (L L^T)^-1 b */
static void solve(const env_t *E, const double *b, double *dx, double *work /* [6 n] */) {
  const int n = E->n;
  for (int i = 0; i < n; ++i) memcpy(work + 6 * (size_t)i, b + 6 * (size_t)E->perm[i], 6 * sizeof(double));
  for (int i = 0; i < n; ++i) { /* forward: L y = b */
    const int fi = E->first[i];
    const size_t si = (size_t)6 * (size_t)E->width[i];
    const double *Pi = E->L + E->off[i];
    const double *yk = work + 6 * (size_t)fi;
    const size_t len = (size_t)6 * (size_t)(i - fi);
    for (int r = 0; r < 6; ++r) {
      const double *x = Pi + (size_t)r * si;
      double s = work[6 * (size_t)i + r];
      for (size_t q = 0; q < len; ++q) s -= x[q] * yk[q];
      for (int m = 0; m < r; ++m) s -= x[len + m] * work[6 * (size_t)i + m];
      work[6 * (size_t)i + r] = s / x[len + r];
    }
  }
  for (int i = n - 1; i >= 0; --i) { /* backward: L^T x = y, row panel i scattered into the rows above */
    const int fi = E->first[i];
    const size_t si = (size_t)6 * (size_t)E->width[i];
    const double *Pi = E->L + E->off[i];
    const size_t len = (size_t)6 * (size_t)(i - fi);
    double *xk = work + 6 * (size_t)fi;
    for (int r = 5; r >= 0; --r) {
      const double *x = Pi + (size_t)r * si;
      double s = work[6 * (size_t)i + r];
      for (int m = r + 1; m < 6; ++m) s -= Pi[(size_t)m * si + len + r] * work[6 * (size_t)i + m];
      work[6 * (size_t)i + r] = s / x[len + r];
    }
    for (int r = 0; r < 6; ++r) {
      const double *x = Pi + (size_t)r * si;
      const double v = work[6 * (size_t)i + r];
      for (size_t q = 0; q < len; ++q) xk[q] -= x[q] * v;
    }
  }
  for (int i = 0; i < n; ++i) memcpy(dx + 6 * (size_t)E->perm[i], work + 6 * (size_t)i, 6 * sizeof(double));
}

/* ---- taps ---------------------------------------------------------------------------------------------------------- */
/* H's diagonal blocks [n_v][36], b [n_v][6] and chi2 at `poses` (vertex order): compared with posegraph_oracle.linearize */
int pgo_linearize(int n_v, const double *poses, int n_e, const int32_t *ij, const double *meas, const double *info,
                  int fixed, double *diag_out, double *b_out, double *chi2_out) {
  env_t E;
  env_build(&E, n_v, n_e, ij);
  double *b = (double *)malloc(sizeof(double) * 6 * (size_t)n_v);
  const double c2 = linearize(&E, n_v, poses, n_e, ij, meas, info, fixed, b);
  if (chi2_out) *chi2_out = c2;
  if (b_out) memcpy(b_out, b, sizeof(double) * 6 * (size_t)n_v);
  if (diag_out)
    for (int v = 0; v < n_v; ++v) {
      const int p = E.inv[v];
      const double *D = env_blk(&E, E.A, p, p);
      const size_t s = (size_t)6 * (size_t)E.width[p];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) diag_out[36 * (size_t)v + r * 6 + c] = D[(size_t)r * s + c]; /* lower triangle */
      for (int r = 0; r < 6; ++r)
        for (int c = r + 1; c < 6; ++c) diag_out[36 * (size_t)v + r * 6 + c] = D[(size_t)c * s + r];
    }
  free(b);
  env_free(&E);
  return 0;
}

/* one damped solve (H + lambda I) dx = b at `poses` */
int pgo_solve(int n_v, const double *poses, int n_e, const int32_t *ij, const double *meas, const double *info, int fixed,
              double lambda, double *dx_out) {
  env_t E;
  env_build(&E, n_v, n_e, ij);
  double *b = (double *)malloc(sizeof(double) * 6 * (size_t)n_v), *w = (double *)malloc(sizeof(double) * 6 * (size_t)n_v);
  linearize(&E, n_v, poses, n_e, ij, meas, info, fixed, b);
  const int rc = factorize(&E, lambda, fixed);
  if (rc == 0) solve(&E, b, dx_out, w);
  free(b); free(w);
  env_free(&E);
  return rc;
}

/* ---- SparseOptimizer::optimize(max_iters) with "lm_var" (solver_g2o.cpp:16,79-95) ---------------------------------- */
int pgo_optimize(int n_v, double *poses /* [n_v][7] in/out */, int n_e, const int32_t *ij, const double *meas,
                 const double *info, int fixed, int max_iters, pgo_stats *st) {
  pgo_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof(*st));
  const double t_all = now_s();
  env_t E;
  double t0 = now_s();
  env_build(&E, n_v, n_e, ij);
  st->t_order = now_s() - t0;
  st->env_blocks = E.env_blocks;
  st->bandwidth = E.bandwidth;
  st->factor_flops = E.flops;
  const size_t n6 = 6 * (size_t)n_v;
  double *b = (double *)malloc(sizeof(double) * n6), *dx = (double *)malloc(sizeof(double) * n6);
  double *work = (double *)malloc(sizeof(double) * n6), *trial = (double *)malloc(sizeof(double) * 7 * (size_t)n_v);
  double lambda = -1.0, ni = 2.0;
  int rc = 0;
  for (int it = 0; it < max_iters; ++it) {
    t0 = now_s();
    const double cur = linearize(&E, n_v, poses, n_e, ij, meas, info, fixed, b);
    st->t_linearize += now_s() - t0;
    if (it == 0) st->chi2_initial = cur;
    st->chi2_final = cur;
    if (lambda < 0) { /* 1e-5 x the largest diagonal entry of H over the free vertices */
      double dmax = 0.0;
      for (int v = 0; v < n_v; ++v) {
        if (v == fixed) continue;
        const int p = E.inv[v];
        const double *D = env_blk(&E, E.A, p, p);
        const size_t s = (size_t)6 * (size_t)E.width[p];
        for (int r = 0; r < 6; ++r) if (D[(size_t)r * s + r] > dmax) dmax = D[(size_t)r * s + r];
      }
      lambda = 1e-5 * dmax;
    }
    double rho = 0.0;
    int qmax = 0;
    for (;;) {
      t0 = now_s();
      rc = factorize(&E, lambda, fixed);
      st->t_factor += now_s() - t0;
      if (rc) break;
      t0 = now_s();
      solve(&E, b, dx, work);
      st->t_solve += now_s() - t0;
      st->trials++;
      t0 = now_s();
      oplus(n_v, poses, dx, fixed, trial);
      const double tmp = pgo_chi2(n_e, trial, ij, meas, info);
      st->t_chi2 += now_s() - t0;
      double scale = 1e-3;
      for (size_t k = 0; k < n6; ++k) scale += dx[k] * (lambda * dx[k] + b[k]);
      rho = (cur - tmp) / scale;
      if (rho > 0 && isfinite(tmp)) {
        double alpha = 1.0 - pow(2 * rho - 1, 3);
        if (alpha > 2.0 / 3.0) alpha = 2.0 / 3.0;
        lambda *= alpha > 1.0 / 3.0 ? alpha : 1.0 / 3.0;
        ni = 2.0;
        memcpy(poses, trial, sizeof(double) * 7 * (size_t)n_v);
        st->chi2_final = tmp;
      } else {
        lambda *= ni;
        ni *= 2.0;
      }
      ++qmax;
      if (!(rho < 0 && qmax < 10)) break;
    }
    if (rc) break;
    st->iterations = it + 1;
    if (qmax == 10 || rho == 0) break;
  }
  st->lambda = lambda;
  st->status = rc;
  st->t_total = now_s() - t_all;
  free(b); free(dx); free(work); free(trial);
  env_free(&E);
  return rc;
}
