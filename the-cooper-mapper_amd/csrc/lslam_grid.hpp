// lslam_grid.hpp -- exact 5-NN without the tree walk: a dense cell grid over the map, a lane-uniform 3 x 3 x 3 probe and a
// completeness proof; the kd-tree search stays as the fallback for the queries the probe cannot prove.
//
// What the reference fixes is the RESULT of KdTreeFLANN::nearestKSearch(p, 5, ...) (util/nanoflann_pcl.h:150-162 ->
// nanoflann.hpp:1303-1323,1433-1497): the five map points with the smallest L2_Simple distance (x -> y -> z accumulation,
// nanoflann.hpp:364-372), ascending, ties in visit order.  Any procedure that evaluates the same fp32 distances and can PROVE
// that (a) no map point outside the ones it looked at can be among the five and (b) the six smallest distances are pairwise
// distinct returns the same five indices in the same order -- without replaying nanoflann's traversal.  A query for which
// either proof fails (sparse neighbourhood, exact distance tie) is handed to knn5_search, which replays it.
//
// The map is VoxelGrid output (at most one centroid per 0.2 / 0.4 m voxel and cube, util/FeatureMap.h:289-306), so a cell of
// c ~ 0.6 m holds a handful of points and the 27 cells around a query a few dozen: fewer candidates than the ~25 inner nodes +
// ~44 leaf slots of the bounded tree search, and -- the point -- the same instruction stream for every lane: no stack, no
// divergent descent, one candidate per lane and round.
//
// Layout in HBM (built by lslam_grid.hip):
//   pts[n]            the map points sorted by cell, {x, y, z, bitcast(original index)}
//   cell_start[N + 1] cell -> first point, cells numbered x-fastest: cell = ix + nx (iy + ny iz); the three x-neighbours of a
//                     cell row are ONE contiguous run of pts, so the 27 cells are nine runs (two loads each)
// Cell of a coordinate v on axis a: floor(fl(fl(v - org[a]) * inv_c)), the same two fp32 operations for map points and queries.
// The grid covers the map's bounding box plus GRID_MARGIN_CELLS(c) empty cells on every side, so that a query whose cell is
// not an interior one is farther than sqrt(5) m from every map point (the acceptance gate of ScanMatch.cpp:102,120).
#pragma once

#include "lslam_device.hpp"

namespace lslam {

struct CellGrid {
  const uint32_t *cell_start;
  const float4 *pts;
  float org[3];
  float inv_c, c;
  int32_t nx, ny, nz;
  int32_t n_pts;
};

#ifndef LSLAM_GRID_BALL
#define LSLAM_GRID_BALL 0  // 1: the x-extent of every row clipped to the bound's BALL instead of its box -- fewer candidates (wave maximum 38 -> 30 rounds in late sweeps on the dumped workload), measured SLOWER: 1.228e10 against 1.250e10 on the same box (nine square roots and ~90 more set-up instructions per probe cost more than the rounds they save)
#endif
#ifndef LSLAM_GRID_ASM_LOOP
#define LSLAM_GRID_ASM_LOOP 1  // 0: the candidate loop as the compiler makes it (A/B switch)
#endif
constexpr float GRID_CELL_DEFAULT = 0.6f;
constexpr int GRID_MAX_DIM = 2048;        // cells per axis (the rounding slack below is sized for it)
constexpr size_t GRID_MAX_CELLS = (size_t)1 << 26;  // cells per table: two dense 4-byte tables rewritten by every build (lslam_grid.hip)
// Rounding of the cell coordinate u(v) = fl(fl(v - o) * inv_c): relative error <= 2^-23 on a value < GRID_MAX_DIM, so two
// points whose cell coordinates differ by w are at least (w - GRID_U_SLACK) cells apart on that axis.
constexpr float GRID_U_SLACK = 2.0e-3f;
constexpr float GRID_CLIP_MARGIN_MIN = 3.0e-3f;  // [m] smallest padding of a clip radius (covers GRID_U_SLACK * c for c <= 1 m)
// nanoflann is an exact fp32 5-NN only up to the rounding of its OWN pruning bound: searchLevel updates the lower bound of a
// far branch incrementally, `mindistsq + cut_dist - dists[idx]` (nanoflann.hpp:1485), two fp32 roundings per far step, and the
// split values are point coordinates (:971-972), so a point can sit on the corner of its subtree's box.  Along a path the
// bound only grows, so after F far steps it exceeds the exact box distance by at most ~1.5 F ulps: a point whose distance is
// within that BELOW the current fifth can be pruned by nanoflann although it is closer -- and the tree walk of
// lslam_device.hpp replays exactly that, the grid cannot.  (With the query outside the box in ONE dimension only -- the
// common case -- the bound is the cut distance itself, exact.)  Every comparison of the proof that separates the fifth
// distance from a point NOT among the five therefore carries a relative margin; a point that fails it is unproven and goes to
// the tree walk.  Where a tree exists the margin costs nothing and is the worst case: 100 ulps = 1.5 x 66 far steps, deeper
// than any tree the builder makes (a few points in 10^5 more are listed).  The wide probe of a map WITHOUT trees
// (sweep_wide_kernel) has no tree walk to hand a point to -- a refusal there builds the trees and repeats the call, 2.2 ms
// on a 1.5 ms mapping frame -- so there the margin is an OPTION (lslam_opts.ab_switches & LSLAM_AB_WIDE_NF_MARGIN:
// GRID_NF_PRUNE_SLACK_WIDE = 8 ulps, five adversarially rounded far steps in a row onto a box corner; 1-2 % of frames pay).
// Off, a fifth / sixth pair that close is ordered by its exact distances.  How often nanoflann itself deviates from the
// exact fp32 five was measured against the reference's own nanoflann on lattice maps jittered by a few ulps
// (tools/nanoflann_exactness.py): 0 of 1 423 985 tie-free queries, 0 of the 37 139 among them whose fifth and sixth
// distances were within 16 ulps.
constexpr float GRID_NF_PRUNE_SLACK = 1.2e-5f;
// a candidate's KEY distance (knn5_grid's loop: one rounded square and two fused multiply-adds) against its exact fp32 distance
// (five roundings): they differ by at most ~3 ulps; a bound taken from a key is shrunk by 8 ulps
constexpr float GRID_KEY_SLACK = 1.0e-6f;
constexpr float GRID_NF_PRUNE_SLACK_WIDE = 9.6e-7f;
inline int grid_margin_cells(float c) { return (int)(2.2361f / c) + 3; }

// a candidate's place in its lane's row table, carried in the low mantissa bits of its key: row slot (9 rows) and offset
constexpr uint32_t GRID_ROW_BITS = 6, GRID_ID_BITS = 10;
constexpr uint32_t GRID_ID_MASK = (1u << GRID_ID_BITS) - 1u, GRID_ROW_MAX = (1u << GRID_ROW_BITS) - 1u;
LSLAM_DEV uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

#ifdef LSLAM_EXP_SECTION_CLOCK  // TIMING EXPERIMENT (same results): where a wavefront of sweep_grid_kernel spends its life, section by section
struct SecClock {
  unsigned long long last, acc[10];
  unsigned long long cand_sum, cand_rounds;  // candidates of the wavefront's lanes; 64 x its slowest lane's (the loop's rounds x 64)
};
// (the clobber keeps memory operations on their side of a section's end; loads are waited for where their values are used)
#define LSLAM_TICK_AT(S, n)                                                                   \
  do {                                                                                        \
    unsigned long long t_;                                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    (S).acc[n] += t_ - (S).last;                                                              \
    (S).last = t_;                                                                            \
  } while (0)
#define LSLAM_TICK_P(P, n) do { if (P) LSLAM_TICK_AT(*(P), n); } while (0)
#define LSLAM_SEC_PARAM , SecClock *scp = nullptr
#else
#define LSLAM_TICK_P(P, n) do { } while (0)
#define LSLAM_SEC_PARAM
#endif

// knn5_grid's verdict
enum : int {
  GRID_UNPROVEN = 0,  // the five returned are not proven to be nanoflann's answer: search the tree
  GRID_PROVEN = 1,    // d[], p[] are nanoflann's answer (p: positions in G.pts)
  GRID_FAR = 2        // the query is farther than sqrt(5) m from every map point; d[] = FLT_MAX, p[] = -1
};

// Exact 5-NN of (qx, qy, qz) among G.pts by a probe of the 3 x 3 x 3 cells around the query, one query per lane; EVERY lane of
// the wavefront must call it (`on` false: the lane has no query and only keeps the loop's rounds company).
//
//   bound      an upper bound of the fifth neighbour's squared distance known in advance (FLT_MAX: none).  Cell rows and
//              cells farther than sqrt(bound) + clip_margin from the query on some axis are not looked at.
//   rows       this lane's row table in LDS: entry k at rows[2 * k * BLOCK], rows[2 * k * BLOCK + 1] (9 entries)
//   lb6        (out) a lower bound of the squared distance of every map point NOT among the five returned -- the sixth
//              smallest candidate distance, the guaranteed radius of the probe and the clip radius, whichever is smallest
//
// Proof obligations (GRID_PROVEN):
//   * the query's cell is an interior cell, so all 27 cells exist;
//   * d[4] < rg2, rg = c (1 + min over the axes of the distance to the nearer cell wall, in cells) less the rounding slack:
//     a point outside the 27 cells differs by at least that much on one axis.  Its computed distance can be smaller than its
//     true one by a few ulps only: rg2 carries a 1e-5 relative pad;
//   * points in rows or cells that were clipped are farther than sqrt(bound) (padded), hence farther than five known points;
//   * d[0] < d[1] < ... < d[4] < (sixth smallest candidate distance) (1 - GRID_NF_PRUNE_SLACK): no tie that nanoflann's visit
//     order would have decided, and no sixth point close enough for the rounding of nanoflann's pruning bound to matter.
// R: rings of cells around the query's cell -- 1: the 27-cell probe of pass 1 (nine runs of three cells); 2: 125 cells
// (twenty-five runs of five), for the points the 27-cell probe could not prove: guaranteed radius c (2 + wall) instead of
// c (1 + wall), rows clipped to the BALL of the caller's bound (which such a point always has: the fifth distance the first
// probe saw).  rows: 2 (2R + 1)^2 words per lane.
template <int BLOCK, int R = 1>
LSLAM_DEV int knn5_grid(const CellGrid &G, const bool on, const float qx, const float qy, const float qz, const float bound,
                        const float clip_margin, lds_u32 *rows, float (&d)[5], int (&p)[5], float &lb6 LSLAM_SEC_PARAM) {
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    d[i] = FLT_MAX;
    p[i] = -1;
  }
  const float ux = __fmul_rn(__fsub_rn(qx, G.org[0]), G.inv_c);
  const float uy = __fmul_rn(__fsub_rn(qy, G.org[1]), G.inv_c);
  const float uz = __fmul_rn(__fsub_rn(qz, G.org[2]), G.inv_c);
  // interior cell: 1 <= i <= n - 2 on every axis (NaN fails every comparison)
  constexpr int W = 2 * R + 1, NR = W * W;          // rows per probe
  constexpr uint32_t IDB = R == 1 ? GRID_ID_BITS : GRID_ID_BITS + 1;  // id = (row slot << GRID_ROW_BITS) | offset in the row
  constexpr uint32_t IDM = (1u << IDB) - 1u;
  static_assert(NR - 1 <= (int)(IDM >> GRID_ROW_BITS), "row slots must fit the id");
  const bool inr = ux >= (float)R && ux < (float)(G.nx - R) && uy >= (float)R && uy < (float)(G.ny - R) && uz >= (float)R && uz < (float)(G.nz - R);
  const bool alive0 = on && inr;
  const float fx0 = floorf(ux), fy0 = floorf(uy), fz0 = floorf(uz);
  const float ex = ux - fx0, ey = uy - fy0, ez = uz - fz0;  // position inside the cell, [0, 1)
  const float wall = fminf(fminf(fminf(ex, 1.0f - ex), fminf(ey, 1.0f - ey)), fminf(ez, 1.0f - ez));
  const float rg = G.c * (((float)R - GRID_U_SLACK) + wall);
  const float rg2 = (rg * rg) * (1.0f - (1.0e-5f + GRID_NF_PRUNE_SLACK));
  // clip box in cell coordinates
  float clip_lo2 = FLT_MAX;
  float xlo = fx0 - (float)R, xhi = fx0 + (float)R, ylo = fy0 - (float)R, yhi = fy0 + (float)R, zlo = fz0 - (float)R, zhi = fz0 + (float)R;
  if (bound < 1.0e30f) {
    const float rb = sqrtf(bound) * (1.0f + 1.0e-5f) + clip_margin;
    const float rbc = rb * G.inv_c;
    xlo = fmaxf(xlo, floorf(ux - rbc)); xhi = fminf(xhi, floorf(ux + rbc));
    ylo = fmaxf(ylo, floorf(uy - rbc)); yhi = fminf(yhi, floorf(uy + rbc));
    zlo = fmaxf(zlo, floorf(uz - rbc)); zhi = fminf(zhi, floorf(uz + rbc));
    const float cl = rb - GRID_U_SLACK * G.c;  // a clipped point is at least this far away
    clip_lo2 = (cl * cl) * (1.0f - 1.0e-5f);
  }
  // The nine runs; all eighteen cell_start loads in flight together.  With a bound the x-extent of a row is clipped to the
  // BALL of radius rb, not its box: a row whose nearest wall is g cells away (in y and z) is scanned over
  // [ux - w, ux + w], w = sqrt(rbc^2 - g^2), and not at all when g > rbc.  g is taken a rounding slack short and w a hair
  // long, so a point outside the scanned cells is still farther than rb - GRID_U_SLACK c (the box version's claim).
  const bool ball = (R > 1 || LSLAM_GRID_BALL) && bound < 1.0e30f;
  const float rbc2 = ball ? (sqrtf(bound) * (1.0f + 1.0e-5f) + clip_margin) * G.inv_c : 0.0f;
  const float rbcs = rbc2 * rbc2;
  const int iy = alive0 ? (int)fy0 : R, iz = alive0 ? (int)fz0 : R;
  uint32_t rs[NR], re[NR];
  bool rowon[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int dy = (r % W) - R, dz = (r / W) - R;
    float rx0 = xlo, rx1 = xhi;
    bool on_r = true;
    if (ball) {
      // distance (in cells) from the query to the nearer wall of the row, a rounding slack short
      const float gy = dy == 0 ? 0.0f : fmaxf((dy < 0 ? ey + (float)(-dy - 1) : (1.0f - ey) + (float)(dy - 1)) - 1.0e-3f, 0.0f);
      const float gz = dz == 0 ? 0.0f : fmaxf((dz < 0 ? ez + (float)(-dz - 1) : (1.0f - ez) + (float)(dz - 1)) - 1.0e-3f, 0.0f);
      const float w2 = rbcs - (gy * gy + gz * gz);
      on_r = w2 > 0.0f;
      const float w = __builtin_amdgcn_sqrtf(fmaxf(w2, 0.0f)) * (1.0f + 1.0e-5f) + 1.0e-4f;
      rx0 = fmaxf(rx0, floorf(ux - w));
      rx1 = fminf(rx1, floorf(ux + w));
    }
    rowon[r] = on_r;
    const int jy = iy + dy, jz = iz + dz;
    const int base = G.nx * (jy + G.ny * jz);
    const int ix0 = alive0 ? (int)rx0 : R, ix1 = alive0 ? (int)rx1 : R;
    rs[r] = G.cell_start[base + ix0];
    re[r] = G.cell_start[base + ix1 + 1];
  }
  // non-empty, unclipped runs, compacted into this lane's LDS table
  int nrow = 0;
  bool row_overflow = false;
#ifdef LSLAM_EXP_SECTION_CLOCK
  uint32_t cand_lane = 0;
#endif
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float jy = fy0 + (float)((r % W) - R), jz = fz0 + (float)((r / W) - R);
    const bool use = alive0 && rowon[r] && jy >= ylo && jy <= yhi && jz >= zlo && jz <= zhi && re[r] > rs[r];
#ifdef LSLAM_EXP_SECTION_CLOCK
    cand_lane += use ? re[r] - rs[r] : 0u;
#endif
    if (use) {
      rows[2 * nrow * BLOCK] = rs[r];
      rows[(2 * nrow + 1) * BLOCK] = re[r];
      ++nrow;
    }
    row_overflow = row_overflow || (use && re[r] - rs[r] > GRID_ROW_MAX + 1u);  // more candidates than an id can count (never seen on voxel maps)
  }
  // The candidate loop.  One candidate per lane and round; what a round keeps is ONE 32-bit key per candidate: the squared
  // distance with its low GRID_ID_BITS mantissa bits replaced by the candidate's place in this lane's row table (row slot,
  // offset in the row).  Squared distances are non-negative, so the keys order like the distances (to 2^-13 relative), and
  // the six smallest keys are kept by six integer min / med3 operations -- no index selects, no per-candidate branches.
  // After the loop the (at most) six survivors are fetched again, their exact distances recomputed with the search's own
  // arithmetic and sorted; everybody else is at least T6 = the sixth key's truncated distance away (a key is never above
  // the exact distance: truncation rounds towards zero), which is what the proof below needs.
  uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, k3 = 0xFFFFFFFFu, k4 = 0xFFFFFFFFu, k5 = 0xFFFFFFFFu;
  uint32_t cur = 0, end = 0, id = 0;
  int k = 0;
  // (a lane with an over-long run is unproven whatever it finds: it sits the loop out -- its candidate ids would overflow
  // into the row-slot bits, and a survivor's place must never be decoded from such a key)
  bool alive = nrow > 0 && !row_overflow;
  if (row_overflow) nrow = 0;
  if (alive) {
    cur = rows[0];
    end = rows[BLOCK];
    k = 1;
  }
  // the row after the current one waits in registers: a lane that runs out of its row takes it with two selects, and the
  // table is read again (by every lane, its own column) only in rounds in which some lane did so.  Row overflow (a run of
  // more than GRID_ROW_MAX + 1 candidates) was decided when the table was written.
  uint32_t ncur = 0, nend = 0;
  if (nrow > 1) {
    ncur = rows[2 * BLOCK];
    nend = rows[3 * BLOCK];
  }
#ifdef LSLAM_EXP_SECTION_CLOCK
  if (scp) {
    uint32_t cs = cand_lane, cm = cand_lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cs += __shfl_xor(cs, o, 64);
      cm = max(cm, (uint32_t)__shfl_xor(cm, o, 64));
    }
    scp->cand_sum += __builtin_amdgcn_readfirstlane(cs);
    scp->cand_rounds += 64u * __builtin_amdgcn_readfirstlane(cm);
  }
#endif
  LSLAM_TICK_P(scp, 1);  // probe set-up: cell coordinates, clip box, eighteen cell-table loads, the row table
#if LSLAM_GRID_ASM_LOOP
  // (The loop's distance is a KEY, not the reference's distance: dx dx, then two fused multiply-adds -- six instructions instead
  // of the eight of L2_Simple's rounded squares and sums.  It differs from the exact fp32 distance by at most a few ulps
  // (three roundings against five), which the proof accounts for where it turns the sixth key into a bound: GRID_KEY_SLACK.
  // The exact distances of the six survivors are re-evaluated with the reference's arithmetic below.)
  // The loop by hand: 19 vector instructions per candidate + 11 in rounds in which a lane changes rows (left to the compiler
  // the same loop carried eleven register copies and a three-deep exec-mask nest per round), software-pipelined one
  // candidate deep: the point of candidate i + 1 is requested -- which takes knowing where candidate i + 1 is, i.e. the row
  // bookkeeping of candidate i -- before the distance of candidate i is evaluated, so a round's load latency hides behind
  // the wavefront's own work as well as the other wavefronts'.  Two register sets (A, B) alternate.  Positions are kept as
  // BYTE offsets into G.pts (x 16) inside the loop.
  static_assert(BLOCK == 256, "the row table's strides are written into the loop's LDS instruction");
  {
    register float pax asm("v2");
    register float pay asm("v3");
    register float paz asm("v4");
    register float pbx asm("v6");
    register float pby asm("v7");
    register float pbz asm("v8");
    register uint32_t l0 asm("v10");
    register uint32_t l1 asm("v11");
    uint32_t curb = cur << 4, endb = end << 4, ncurb = ncur << 4, nendb = nend << 4;
    unsigned long long am = __builtin_amdgcn_ballot_w64(alive);
    unsigned long long adv, tmp;
    uint32_t t0, t1, key, ida, idb;
    const uint32_t rowaddr = (uint32_t)(uintptr_t)rows;
    const uint32_t keep = ~IDM;
// (a lane that has run out of candidates carries the id 0xFFFFFFFF: `(dist & keep) | id` is then the empty key, no select)
#ifdef LSLAM_EXP_LOAD2   // TIMING EXPERIMENT ONLY (wrong results): 8 bytes per candidate through the L1 instead of 12
#define LSLAM_GRID_LOADOP "global_load_dwordx2 "
#define LSLAM_GRID_PA "v[2:3]"
#define LSLAM_GRID_PB "v[6:7]"
#else
#define LSLAM_GRID_LOADOP "global_load_dwordx3 "
#define LSLAM_GRID_PA "v[2:4]"
#define LSLAM_GRID_PB "v[6:8]"
#endif
#ifdef LSLAM_EXP_LOAD_TWICE  // TIMING EXPERIMENT ONLY (same results): every candidate requested twice -- twice the bytes through the L1
#define LSLAM_GRID_VMW "s_waitcnt vmcnt(2)\n\t"
#define LSLAM_GRID_LOAD_AGAIN(P, A) LSLAM_GRID_LOADOP P ", " A ", %[base]\n\t"
#else
#define LSLAM_GRID_VMW "s_waitcnt vmcnt(1)\n\t"
#define LSLAM_GRID_LOAD_AGAIN(P, A)
#endif
#define LSLAM_GRID_ISSUE(ID, P)                       \
  "v_cndmask_b32 " ID ", -1, %[id], %[am]\n\t"           \
  "v_cndmask_b32 %[t0], 0, %[curb], %[am]\n\t"           \
  LSLAM_GRID_LOADOP P ", %[t0], %[base]\n\t" LSLAM_GRID_LOAD_AGAIN(P, "%[t0]")
#ifdef LSLAM_EXP_PROCESS_TWICE  // TIMING EXPERIMENT ONLY: the key distance evaluated twice (six more vector instructions per candidate)
#define LSLAM_GRID_EXTRA(PX, PY, PZ)                     \
  "v_sub_f32 %[t0], %[qx], " PX "\n\t"                   \
  "v_mul_f32 %[t0], %[t0], %[t0]\n\t"                    \
  "v_sub_f32 %[t1], %[qy], " PY "\n\t"                   \
  "v_fmac_f32 %[t0], %[t1], %[t1]\n\t"                   \
  "v_sub_f32 %[t1], %[qz], " PZ "\n\t"                   \
  "v_fmac_f32 %[t0], %[t1], %[t1]\n\t"
#else
#define LSLAM_GRID_EXTRA(PX, PY, PZ)
#endif
#define LSLAM_GRID_ADVANCE(L)                                        \
  "v_add_u32 %[curb], 16, %[curb]\n\t"                              \
  "v_add_u32 %[id], 1, %[id]\n\t"                                   \
  "v_cmp_eq_u32 %[adv], %[curb], %[endb]\n\t"                       \
  "s_and_b64 %[adv], %[adv], %[am]\n\t"                             \
  "s_cbranch_scc0 L_grid_noadv" L "_%=\n\t"                         \
  "v_cmp_lt_i32 vcc, %[k], %[nrow]\n\t"                             \
  "s_andn2_b64 %[tmp], %[adv], vcc\n\t"                             \
  "s_andn2_b64 %[am], %[am], %[tmp]\n\t"                            \
  "v_cndmask_b32 %[curb], %[curb], %[ncurb], %[adv]\n\t"            \
  "v_cndmask_b32 %[endb], %[endb], %[nendb], %[adv]\n\t"            \
  "v_lshlrev_b32 %[t0], 6, %[k]\n\t"                                \
  "v_cndmask_b32 %[id], %[id], %[t0], %[adv]\n\t"                   \
  "v_addc_co_u32 %[k], %[tmp], 0, %[k], %[adv]\n\t"                 \
  "v_min_i32 %[t0], %[nlast], %[k]\n\t"                             \
  "v_lshl_add_u32 %[t0], %[t0], 11, %[rowaddr]\n\t"                 \
  "ds_read2st64_b32 v[10:11], %[t0] offset1:4\n\t"                    \
  "s_waitcnt lgkmcnt(0)\n\t"                                        \
  "v_lshlrev_b32 %[ncurb], 4, %[l0]\n\t"                            \
  "v_lshlrev_b32 %[nendb], 4, %[l1]\n"                               \
  "L_grid_noadv" L "_%=:\n\t"
#define LSLAM_GRID_PROCESS(ID, PX, PY, PZ)            \
  LSLAM_GRID_EXTRA(PX, PY, PZ)                           \
  "v_sub_f32 %[t0], %[qx], " PX "\n\t"                   \
  "v_mul_f32 %[t0], %[t0], %[t0]\n\t"                    \
  "v_sub_f32 %[t1], %[qy], " PY "\n\t"                   \
  "v_fmac_f32 %[t0], %[t1], %[t1]\n\t"                   \
  "v_sub_f32 %[t1], %[qz], " PZ "\n\t"                   \
  "v_fmac_f32 %[t0], %[t1], %[t1]\n\t"                   \
  "v_and_or_b32 %[key], %[t0], %[keep], " ID "\n\t"      \
  "v_med3_u32 %[k5], %[k4], %[k5], %[key]\n\t"           \
  "v_med3_u32 %[k4], %[k3], %[k4], %[key]\n\t"           \
  "v_med3_u32 %[k3], %[k2], %[k3], %[key]\n\t"           \
  "v_med3_u32 %[k2], %[k1], %[k2], %[key]\n\t"           \
  "v_med3_u32 %[k1], %[k0], %[k1], %[key]\n\t"           \
  "v_min_u32 %[k0], %[k0], %[key]\n\t"
    asm volatile(
        "s_cmp_eq_u64 %[am], 0\n\t"
        "s_cbranch_scc1 L_grid_done_%=\n\t"
        LSLAM_GRID_ISSUE("%[ida]", LSLAM_GRID_PA)
        LSLAM_GRID_ADVANCE("0")
        "L_grid_loop_%=:\n\t"
        "s_cmp_eq_u64 %[am], 0\n\t"
        "s_cbranch_scc1 L_grid_lasta_%=\n\t"
        LSLAM_GRID_ISSUE("%[idb]", LSLAM_GRID_PB)
        LSLAM_GRID_ADVANCE("1")
        LSLAM_GRID_VMW
        LSLAM_GRID_PROCESS("%[ida]", "%[pax]", "%[pay]", "%[paz]")
        "s_cmp_eq_u64 %[am], 0\n\t"
        "s_cbranch_scc1 L_grid_lastb_%=\n\t"
        LSLAM_GRID_ISSUE("%[ida]", LSLAM_GRID_PA)
        LSLAM_GRID_ADVANCE("2")
        LSLAM_GRID_VMW
        LSLAM_GRID_PROCESS("%[idb]", "%[pbx]", "%[pby]", "%[pbz]")
        "s_branch L_grid_loop_%=\n"
        "L_grid_lasta_%=:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        LSLAM_GRID_PROCESS("%[ida]", "%[pax]", "%[pay]", "%[paz]")
        "s_branch L_grid_done_%=\n"
        "L_grid_lastb_%=:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        LSLAM_GRID_PROCESS("%[idb]", "%[pbx]", "%[pby]", "%[pbz]")
        "L_grid_done_%=:\n\t"
        : [k0] "+v"(k0), [k1] "+v"(k1), [k2] "+v"(k2), [k3] "+v"(k3), [k4] "+v"(k4), [k5] "+v"(k5), [curb] "+v"(curb), [endb] "+v"(endb),
          [ncurb] "+v"(ncurb), [nendb] "+v"(nendb), [id] "+v"(id), [k] "+v"(k), [am] "+s"(am), [adv] "=&s"(adv), [tmp] "=&s"(tmp),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [key] "=&v"(key), [ida] "=&v"(ida), [idb] "=&v"(idb),
          [pax] "=&v"(pax), [pay] "=&v"(pay), [paz] "=&v"(paz), [pbx] "=&v"(pbx), [pby] "=&v"(pby), [pbz] "=&v"(pbz), [l0] "=&v"(l0), [l1] "=&v"(l1)
        : [qx] "v"(qx), [qy] "v"(qy), [qz] "v"(qz), [nrow] "v"(nrow), [rowaddr] "v"(rowaddr), [base] "s"(G.pts), [keep] "s"(keep),
          [nlast] "n"(NR - 1)
        : "vcc", "scc", "memory");
#undef LSLAM_GRID_ISSUE
#undef LSLAM_GRID_EXTRA
#undef LSLAM_GRID_VMW
#undef LSLAM_GRID_LOAD_AGAIN
#undef LSLAM_GRID_LOADOP
#undef LSLAM_GRID_PA
#undef LSLAM_GRID_PB
#undef LSLAM_GRID_ADVANCE
#undef LSLAM_GRID_PROCESS
  }
#else
  while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {
    const float4 pt = G.pts[alive ? cur : 0u];
    const float dist = dist2_xyz(qx, qy, qz, pt);
    uint32_t key = (__float_as_uint(dist) & ~IDM) | id;
    key = alive ? key : 0xFFFFFFFFu;
    // six smallest keys, ascending, updated in place from the top: slot i becomes med3(old slot i-1, old slot i, key)
    asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k5) : "v"(k4), "v"(key));
    asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k4) : "v"(k3), "v"(key));
    asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k3) : "v"(k2), "v"(key));
    asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k2) : "v"(k1), "v"(key));
    asm("v_med3_u32 %0, %1, %0, %2" : "+v"(k1) : "v"(k0), "v"(key));
    k0 = min(k0, key);
    cur += 1u;
    id += 1u;
    const bool adv = alive && cur == end;
    if (__builtin_amdgcn_ballot_w64(adv) != 0ull) {  // wave-uniform
      const bool more = k < nrow;
      alive = alive && !(adv && !more);
      cur = adv ? ncur : cur;
      end = adv ? nend : end;
      id = adv ? (uint32_t)k << GRID_ROW_BITS : id;
      k += adv ? 1 : 0;
      const int kn = k < NR - 1 ? k : NR - 1;  // (the last entry may never have been written: only read, never used, in that case)
      ncur = rows[2 * kn * BLOCK];
      nend = rows[(2 * kn + 1) * BLOCK];
    }
  }
#endif
  LSLAM_TICK_P(scp, 2);  // the candidate loop
  // the survivors: place in the row table -> position in G.pts -> exact distance
  const uint32_t ks[6] = {k0, k1, k2, k3, k4, k5};
  uint32_t pos[6];
  float4 sp[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const bool have = ks[j] != 0xFFFFFFFFu;
    const uint32_t slot = (ks[j] & IDM) >> GRID_ROW_BITS, off = ks[j] & GRID_ROW_MAX;
    pos[j] = have ? rows[2 * slot * BLOCK] + off : 0u;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) sp[j] = G.pts[pos[j]];
  float e[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) e[j] = (ks[j] != 0xFFFFFFFFu) ? dist2_xyz(qx, qy, qz, sp[j]) : FLT_MAX;
  // The keys order the distances to 2^-13 relative: nearly always the exact distances of the six survivors ascend in key
  // order, and then the first five ARE the answer in nanoflann's order.  A wavefront with a lane for which they do not
  // (two of the six within 2^-13 of each other) sorts by the search's own insert.
  float lb = e[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    d[j] = e[j];
    p[j] = (ks[j] != 0xFFFFFFFFu) ? (int)pos[j] : -1;
  }
  const bool ascending = e[0] <= e[1] && e[1] <= e[2] && e[2] <= e[3] && e[3] <= e[4] && e[4] <= e[5];
  if (__builtin_amdgcn_ballot_w64(!ascending) != 0ull) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      d[j] = FLT_MAX;
      p[j] = -1;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) knn_insert_sorted(d, p, e[j], (ks[j] != 0xFFFFFFFFu) ? (int)pos[j] : -1);
    lb = fmaxf(e[5], d[4]);  // whoever of the six is left out
    knn_insert_sorted(d, p, e[5], (int)pos[5]);
  }
  // everybody who is not a survivor is at least the sixth key's truncated distance away; the sixth exact distance less
  // nanoflann's pruning slack (FLT_MAX stays FLT_MAX: no sixth candidate at all)
  const float t6 = (k5 != 0xFFFFFFFFu) ? __uint_as_float(k5 & ~IDM) * (1.0f - GRID_KEY_SLACK) : FLT_MAX;
  lb = fminf(lb < 1.0e30f ? lb * (1.0f - GRID_NF_PRUNE_SLACK) : lb, t6);
  if (row_overflow) lb = 0.0f;
  lb6 = fminf(fminf(lb, rg2), clip_lo2);
  if (!inr) {
    lb6 = 0.0f;
    // outside the interior cells: beyond the margin, i.e. farther than the gate from the whole map -- or not a number
    const bool num = (ux == ux) && (uy == uy) && (uz == uz);
    return num ? GRID_FAR : GRID_UNPROVEN;
  }
  const bool distinct = d[0] < d[1] && d[1] < d[2] && d[2] < d[3] && d[3] < d[4];
  return (distinct && d[4] < lb6) ? GRID_PROVEN : GRID_UNPROVEN;
}

}  // namespace lslam
