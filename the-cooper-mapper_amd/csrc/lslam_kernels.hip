// lslam_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
//   sweep_kernel   one lane per scan point: transform -> exact kd-tree 5-NN ->
//                  line/plane fit -> residual coefficient -> Jacobian row ->
//                  27 normal-equation terms, reduced per block
//                  (ScanMatch.cpp:97-204 + the products of :206-208)
//   solve_kernel   deterministic cross-block reduction, 6x6 solve, degeneracy
//                  projection, pose update, convergence test (ScanMatch.cpp:141-145,
//                  206-260); writes the next pose for the following sweep, so the
//                  whole Gauss-Newton loop stays on the device
//   knn5_kernel    parity tap: nearestKSearch(p, 5, ...) (nanoflann_pcl.h:150-162)
//
// Compile with -ffp-contract=off (see lslam_device.hpp).
#include <hip/hip_ext.h>

#include "lslam_internal.hpp"
#include "lslam_packet.hpp"
#include "lslam_odom_dev.hpp"
#include "lslam_solve_dev.hpp"

namespace lslam {

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b%8).  Remap
// so that every XCD works on one contiguous range of scan points: neighbouring
// scan points walk the same kd-tree nodes and leaves, which then stay in that
// XCD's private 4 MiB L2.  Bijective for any grid size.
LSLAM_DEV int xcd_remap(int b, int nb) {
  const int xcd = b & 7;
  const int q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// ---------------------------------------------------------------------------
// J^T J by MFMA (jtj_mode 1): the wave's 64 rows [J | b] (7 of 16 columns used)
// are staged through LDS and contracted with v_mfma_f32_16x16x4_f32, 4 scan
// points per instruction, 16 instructions per wave.  For J^T J the A operand
// (16 x 4: A[i][k] = J[k][i]) and the B operand (4 x 16: B[k][j] = J[k][j]) hold
// the SAME value in lane l = 16*k + i, so one ds_read feeds both.  The MFMA is an
// exact fp32 fma chain in k order (no reduced precision).
// ---------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

// worldToCube + isIndexValid + toIndex (util/FeatureMap.h:475-487,102-108,146-148)
LSLAM_DEV int cube_tree_of(const CubeGridDev &g, float x, float y, float z) {
  const int gi = (int)(roundf(x / g.cube_size) + (float)g.origin[0]);
  const int gj = (int)(roundf(y / g.cube_size) + (float)g.origin[1]);
  const int gk = (int)(roundf(z / g.cube_size) + (float)g.origin[2]);
  if (0 <= gi && gi < g.dims[0] && 0 <= gj && gj < g.dims[1] && 0 <= gk && gk < g.dims[2])
    return g.cell_tree[gi + gj * g.dims[0] + gk * g.dims[0] * g.dims[1]];
  return -1;
}

// Upper bound of the fifth neighbour's squared distance at `sel` from what the previous sweep left (grid sweep): the point
// was at pq.xyz then and its five neighbours were within sqrt(pq.w) of it, so they are within sqrt(pq.w) + |sel - pq.xyz| now
// (triangle inequality; padded by 1e-5 relative, far above the rounding of the three square roots and the distances'
// own).  pq.w = FLT_MAX (no five inside the gate last time): no bound.
LSLAM_DEV float grid_carried_bound(const float4 &pq, const float (&sel)[3]) {
  const float ex = sel[0] - pq.x, ey = sel[1] - pq.y, ez = sel[2] - pq.z;
  const float delta = sqrtf((ex * ex + ey * ey) + ez * ez);
  const float r = (sqrtf(pq.w) + delta) * (1.0f + 1.0e-5f) + 1.0e-6f;
  const float b = r * r;
  return pq.w < 1.0e30f ? b : FLT_MAX;
}

// ScanMatch.cpp:102-139 for one point whose five neighbours are known: the acceptance gate, findLine / findPlane on the five
// (fetched from P, the array the neighbour ids p[] index), the coefficient, the Jacobian row; the optional per-point taps.
// Shared by the tree sweep (P = the tree's permuted points) and the grid sweep (P = the cell-sorted points).
LSLAM_DEV void point_residual(const SweepArgs &a, const BlockDesc &bd, const bool is_surf, const float4 *P, const float4 &q,
                              const float (&sel)[3], const float (&d)[5], const int (&p)[5], const float (&sc)[6],
                              float (&row)[6], float &rb, float &kept, float &matched, float &score) {
  float coeff[4] = {0, 0, 0, 0};
  unsigned flag = 0;
  float4 nb[5];
  // ScanMatch.cpp:102,120; the _fineScore re-sweep gates on the nearest neighbour instead (:282,302)
  const bool gate = a.fine_gate_c >= 0.0f ? d[0] < (is_surf ? a.fine_gate_s : a.fine_gate_c) : d[4] < 5.0f;
  if (gate) {
    flag |= 1u;
#pragma unroll
    for (int j = 0; j < 5; ++j) nb[j] = P[p[j]];
    if (!is_surf) {
      float A[3], B[3];
      if (find_line(nb, A, B)) {  // ScanMatch.cpp:105-112
        flag |= 2u;
        if (corner_coeff(A, B, sel, coeff)) flag |= 4u;
      }
    } else {
      float plane[4];
#ifdef LSLAM_EXP_NO_FIT  // TIMING EXPERIMENT ONLY (wrong results): no plane fit -- what the fit costs
      plane[0] = 0.0f; plane[1] = 0.0f; plane[2] = 1.0f; plane[3] = -nb[0].z;
      if (nb[1].x < 1.0e30f) {
#else
      if (find_plane(nb, 0.2f, plane)) {  // ScanMatch.cpp:122-130
#endif
        flag |= 2u;
        if (surf_coeff(plane, sel, coeff)) flag |= 4u;
      }
    }
  }
#ifdef LSLAM_FIT_TWICE  // profiling only: the fit and the coefficient once more with no effect -> their share of the kernel time
  if (gate) {
    float4 nb2[5];
    float off = 0.0f;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int j = 0; j < 5; ++j) nb2[j] = make_float4(nb[j].x + off, nb[j].y, nb[j].z, nb[j].w);
    float c2[4] = {0, 0, 0, 0};
    bool any2 = false;
    if (!is_surf) {
      float A[3], B[3];
      if (find_line(nb2, A, B)) any2 = corner_coeff(A, B, sel, c2);
    } else {
      float plane[4];
      if (find_plane(nb2, 0.2f, plane)) any2 = surf_coeff(plane, sel, c2);
    }
    if (any2 && c2[3] == off + 1e30f) coeff[3] = c2[0];  // never
  }
#endif
  if (flag & 2u) matched = 1.0f;
  if (flag & 4u) {
    jacobian_row(sc, q.x, q.y, q.z, coeff, row, rb);
    kept = 1.0f;
    score = expf(-fabsf(coeff[3]));
  }
  if (a.flags_out) {  // parity taps
    const int gi = bd.out_base + __float_as_int(q.w);  // caller's index of this point
    a.flags_out[gi] = (uint8_t)flag;
    if (a.coeff_out) a.coeff_out[gi] = make_float4(coeff[0], coeff[1], coeff[2], coeff[3]);
    if (a.idx_out) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        a.idx_out[gi * 5 + j] = p[j] >= 0 ? __float_as_int(P[p[j]].w) : -1;
        a.d2_out[gi * 5 + j] = d[j];
      }
    }
  }
}

// The block's 27 normal-equation sums and counters from its lanes' rows (ScanMatch.cpp:206-208 products): per-wave contraction
// (MFMA f32 16x16x4 through `stage`, eight word rows of BLOCK lanes that the wavefront owns, or VALU + wave shuffles), then a
// fixed-order sum over the block's waves.  ADD: onto the record another launch left (pass 2 of a two-pass sweep).
// PAD: words added to the staging rows' stride of BLOCK.  With the stride a multiple of the 64 banks the eight lanes that fetch
// the eight rows' entries of one point for an MFMA step share a bank (an eight-way conflict in every one of the sixteen steps);
// with PAD = 4 the 32 words of a step lie in 32 banks.  Same words to the same lanes: same bits.  Only for a caller whose `stage`
// is free of other wavefronts' data (a padded row reaches into the neighbouring wavefronts' columns): the grid sweep, behind
// its workgroup barrier; the tree sweeps stage in the columns of their own traversal stacks, without one.
template <int BLOCK, bool FUSE, bool ADD, int PAD = 0>
LSLAM_DEV void block_accumulate(const int jtj_mode, const bool is_surf, const float (&row)[6], const float rb, const float kept,
                                const float matched, const float score, uint32_t *stage, float (*red)[NCOL], float *partial_out) {
  constexpr int NWAVE = BLOCK / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // ---- normal equations: per-wave reduction --------------------------------
  float v[NCOL];
#pragma unroll
  for (int i = 0; i < NCOL; ++i) v[i] = 0.0f;

#ifdef LSLAM_EXP_NO_ACC  // TIMING EXPERIMENT ONLY (wrong results): no contraction -- what the per-wave J^T J costs
  if (false) {
#else
  if (jtj_mode == 1) {
#endif
    // stage [J | b] rows; rows of rejected points are zero
    constexpr int ST = BLOCK + PAD;
    float *jr = reinterpret_cast<float *>(stage) + wave * 64;  // [c * ST + p]
#pragma unroll
    for (int c = 0; c < 6; ++c) jr[c * ST + lane] = row[c];
    jr[6 * ST + lane] = rb;
    jr[7 * ST + lane] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int i16 = lane & 15, k4 = lane >> 4;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float op = (i16 < 8) ? jr[i16 * ST + 4 * s + k4] : 0.0f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(op, op, acc, 0, 0, 0);
    }
    // C/D layout: col = lane&15, row = (lane>>4)*4 + reg.  Entry (r,c), r<=c<7.
    // Scatter the 27 needed entries back to column slots through LDS.
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int rr = k4 * 4 + r4, cc = i16;
      if (rr < 6 && cc < 7 && cc >= rr) {
        int col;
        if (cc == 6) col = COL_ATB + rr;
        else col = COL_ATA + (rr * 6 - (rr * (rr - 1)) / 2) + (cc - rr);
        red[wave][col] = acc[r4];
      }
    }
    // counters still go through the shuffle reduction
    const float s_rows = wave_sum(kept), s_match = wave_sum(matched), s_score = wave_sum(score);
    if (lane == 0) {
      red[wave][COL_ROWS] = s_rows;
      red[wave][COL_LINE] = is_surf ? 0.0f : s_match;
      red[wave][COL_PLANE] = is_surf ? s_match : 0.0f;
      red[wave][COL_SCORE] = s_score;
      red[wave][31] = 0.0f;
    }
  } else {
    // ScanMatch.cpp:206-208 products, then gfx950 wave-shuffle reduction
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) v[k++] = row[i] * row[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[COL_ATB + i] = row[i] * rb;
    v[COL_ROWS] = kept;
    v[COL_LINE] = is_surf ? 0.0f : matched;
    v[COL_PLANE] = is_surf ? matched : 0.0f;
    v[COL_SCORE] = score;
#pragma unroll
    for (int i = 0; i < 31; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NCOL; ++i) red[wave][i] = v[i];
    }
  }
  __syncthreads();
  // LDS-staged per-block accumulation: fixed order over the block's waves
  if (tid < NCOL) {
    float s = red[0][tid];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) s += red[w][tid];
    // fused solve: the record is read by another workgroup of THIS launch (possibly on another XCD) -- written through to
    // the device's coherence point; otherwise by the next launch
    if (FUSE) __hip_atomic_store(partial_out + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (ADD) partial_out[tid] += s;  // onto the record pass 1 left (this thread alone touches the word)
    else partial_out[tid] = s;
  }
}

#ifdef LSLAM_EXP_PASS2_CLASS
#ifndef LSLAM_EXP_PASS2_SPLIT
#define LSLAM_EXP_PASS2_SPLIT 1.0
#endif
__device__ unsigned long long g_pass2_hist[8];  // listed points by the bound their tree search starts from: < 0.5, < 1, < 2, < 4.99 m^2, the gate
#endif
#ifndef LSLAM_SHALLOW_OCC
#define LSLAM_SHALLOW_OCC 5  // wavefronts per SIMD the shallow-stack variant is compiled for (96 VGPRs, 25 KB LDS per workgroup)
#endif
// PACKET: the 5-NN search is the wave-cooperative one of lslam_packet.hpp (scalar loads, no per-lane
// traversal stack: LDS_DEPTH = 4 only provides the eight staging rows of the MFMA contraction); lanes
// that see an exact distance tie redo their search with nanoflann's traversal (stack in HBM).
// One block of one sweep: the body of sweep_kernel, and of an iteration of gn_persistent_kernel.  `st` is the scan's
// state -- in HBM (STATE_LDS = false: R, t, sc arrive by scalar loads) or a workgroup's LDS copy (true); `red` and
// `stack_lds` are the caller's LDS; the block's 32 sums go to partial_out[0..32).
// CERT (the bounded production sweep over whole-map trees, two launches): 1 = the pass over every point, which carries the
// neighbour lists of points that have hardly moved over by certificate and lists the others; 2 = the pass over the listed
// points (sweep_queue_kernel: `q_item` is this lane's point or -1, the sums are ADDED to partial_out).  See below.
// GRIDQ (with CERT = 2): 0 = the certificate sweep's second pass; 1 = the grid sweep's (tree search bounded by what pass 1
// hands over, no certificate bookkeeping); 2 = the grid sweep's on a map without trees (neighbours given by sweep_wide_kernel).
// A template argument, not a run-time flag: each second pass is compiled without the others' search code (the one kernel for
// all three needed 96 VGPRs + 92 bytes of scratch per lane).
template <int BLOCK, bool OVF, bool CUBES, int LDS_DEPTH, bool PACKET, bool STATE_LDS, bool FUSE = false, int CERT = 0, int GRIDQ = 0>
LSLAM_DEV void sweep_body(const SweepArgs &a, const int jtj_mode, const int lb, const BlockDesc &bd, const GNState *st,
                          uint32_t *stack_lds, float (*red)[NCOL], float *partial_out, const int prev_valid, const int q_item = -1) {
  constexpr int NWAVE = BLOCK / 64;
  static_assert(2 * LDS_DEPTH >= 8, "staging needs eight word rows of the stack");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint64_t dbg_t0 = a.dbg ? __builtin_readcyclecounter() : 0;
  uint64_t dbg_t1 = 0, dbg_t2 = 0;
  const bool is_surf = bd.is_surf != 0;
  const int qi = CERT == 2 ? q_item : bd.first + tid;
  bool active = CERT == 2 ? q_item >= 0 : tid < bd.count;

  // pose of this iteration: R, t, sc are 18 consecutive floats of the scan's GNState.  The compiler cannot prove the
  // state invariant (the solve kernel writes it between launches) and would fetch it with vector loads into 18 VGPRs
  // that then sit in the register file through the whole search: fetched with scalar loads they live in SGPRs.
  float R[9], t[3], sc[6];
  if (STATE_LDS) {  // the workgroup's own copy: wave-uniform values, moved to SGPRs
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st->R[i])));
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st->t[i])));
#pragma unroll
    for (int i = 0; i < 6; ++i) sc[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st->sc[i])));
  } else {
    static_assert(offsetof(GNState, R) == 24 && offsetof(GNState, t) == 60 && offsetof(GNState, sc) == 72, "GNState layout");
    typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x16_t w;
    u32x2_t w2;
    asm volatile("s_load_dwordx16 %0, %2, 0x18\n\ts_load_dwordx2 %1, %2, 0x58\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(w), "=&s"(w2)
                 : "s"(st)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = __uint_as_float(w[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = __uint_as_float(w[9 + i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) sc[i] = __uint_as_float(w[12 + i]);
    sc[4] = __uint_as_float(w2[0]);
    sc[5] = __uint_as_float(w2[1]);
  }

  float row[6] = {0, 0, 0, 0, 0, 0};
  float rb = 0.0f;
  float kept = 0.0f, matched = 0.0f, score = 0.0f;

  // PACKET: the search is a wave-level operation -- every lane of the wavefront takes part (lanes without
  // a point carry a dummy query and are masked out of every decision)
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  float sel[3] = {0.f, 0.f, 0.f};
  float d[5];
  int p[5];
  TreeView T;
  if (PACKET) {
    if (active) q = a.q[qi];
    sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
    sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
    sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
    T = is_surf ? a.ts : a.tc;  // block-uniform
    float cap = FLT_MAX;
    if (a.bounded) {
      cap = 5.0f * (1.0f + 1e-5f);
      if (prev_valid && active) {
        float u = 0.0f;
        bool all = true;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int pp = a.prev_nb[(size_t)qi * 5 + j];
          all = all && pp >= 0 && pp < T.n_pts;
          if (pp >= 0 && pp < T.n_pts) u = fmaxf(u, dist2_xyz(sel[0], sel[1], sel[2], T.pts[pp]));
        }
        if (all) cap = fminf(cap, u * (1.0f + 1e-5f) + 1e-12f);
      }
    }
    bool tie;
#ifdef LSLAM_PACKET_STATS
    PacketStats ps_unused = {0, 0, 0, 0};
    knn5_packet(T, sel[0], sel[1], sel[2], active, cap, d, p, tie, ps_unused);
#else
    knn5_packet(T, sel[0], sel[1], sel[2], active, cap, d, p, tie);
#endif
    if (a.bounded) tie = tie && d[4] < 5.0f;  // beyond the gate the five are not used (ScanMatch.cpp:102,120)
    if (__any(tie)) {  // exact distance ties: nanoflann's visit order decides -- its own traversal, stack in HBM
      if (tie) {
        KdStack<BLOCK, true, 0> stk;
        stk.lds = nullptr;
        stk.ovf = a.stack_ovf + ((size_t)lb * BLOCK + tid);
        stk.ovf_stride = (size_t)a.nb_total * BLOCK;
        knn5_search<BLOCK, true, 0>(T, sel[0], sel[1], sel[2], d, p, stk);
      }
    }
    if (a.bounded && active) {
#pragma unroll
      for (int j = 0; j < 5; ++j) a.prev_nb[(size_t)qi * 5 + j] = p[j];
    }
  }

  // ---------------------------------------------------------------------------------------------------------------------
  // CERT: "the five are still the five".  From the second sweep of a loop a point knows its previous five neighbours, where
  // it was then (q_prev) and a lower bound lb6 on the squared distance from q_prev of every OTHER map point
  // (knn5_search<TRACK>).  It has moved by delta = |q - q_prev|.  Every other point is now at least sqrt(lb6) - delta away
  // (triangle inequality); if the farthest of the five, at its NEW distance, is strictly closer than that, the five nearest
  // neighbours are the same five -- proven, not searched -- and nanoflann's answer is those five in ascending order of their
  // new distances (recomputed with the search's own arithmetic; five distinct distances: an exact tie goes to the search,
  // whose visit order decides).  Late iterations move a scan by millimetres: most points certify.
  //   A wavefront is only as fast as its slowest lane and a workgroup holds its LDS until its last wavefront retires, so a
  // search left in a workgroup of certified points costs what sixty-four do.  Hence two launches.  Pass 1 (this one, every
  // point): a scan whose last update was small enough for certificates to stand a chance (block-uniform test on the state's
  // delta_r / delta_t) tests every point, runs the residuals of the certified ones and LISTS the others -- one byte per
  // point, the lane number, at a fixed place per workgroup; other scans are searched right here as without certificates.
  // Pass 2 (sweep_queue_kernel): one workgroup per group of CERT_GROUP consecutive pass-1 workgroups of the same scan and
  // feature type concatenates their lists and searches the listed points with every lane busy, its sums added to the
  // group's first record.  No atomics, fixed places, fixed order: the sums -- hence the poses -- are the same bits in every
  // run.  Results equal searching every point (tests/test_gpu_stack_shapes.py holds the two against each other and both
  // against the oracle); LSLAM_KNN_CERT=0 searches every point in one launch.
  // ---------------------------------------------------------------------------------------------------------------------
  bool do_search = true;  // this lane's neighbours come from a search below
  // grid sweep (pass 2 of it: the points sweep_grid_kernel could not prove): what is carried from sweep to sweep is where the
  // point was and how far its fifth neighbour (prev_q: x, y, z, d2[4]) -- no neighbour ids, no certificates
  static_assert(GRIDQ == 0 || CERT == 2, "GRIDQ is a mode of the second pass");
  constexpr bool grid = GRIDQ != 0;
  bool track = CERT == 2 && !grid && a.prev_q != nullptr;  // ... which keeps the bound the next sweep's certificate needs
  if (GRIDQ == 3) {
    // The grid sweep's SECOND probe: this lane's point is one the 27-cell probe could not prove (q_item).  125 cells now --
    // guaranteed radius c (2 + wall) -- clipped to the ball of what is known about the point: the fifth distance the first probe
    // saw, the carried bound, the gate.  Proven: the residual chain below.  Still unproven (a sparser neighbourhood yet, an
    // exact tie): listed once more, at this work item's fixed place, for the tree search.
    static_assert(GRIDQ != 3 || (2 * LDS_DEPTH >= 50 && BLOCK == 256), "the second probe's row table: 25 runs per lane");
    __shared__ int wave_needy2[NWAVE];
    CellGrid G;
    G.cell_start = is_surf ? a.ks.cell_start : a.kc.cell_start;
    G.pts = is_surf ? a.ks.pts : a.kc.pts;
#pragma unroll
    for (int i = 0; i < 3; ++i) G.org[i] = is_surf ? a.ks.org[i] : a.kc.org[i];
    G.inv_c = is_surf ? a.ks.inv_c : a.kc.inv_c;
    G.c = is_surf ? a.ks.c : a.kc.c;
    G.nx = is_surf ? a.ks.nx : a.kc.nx;
    G.ny = is_surf ? a.ks.ny : a.kc.ny;
    G.nz = is_surf ? a.ks.nz : a.kc.nz;
    G.n_pts = is_surf ? a.ks.n_pts : a.kc.n_pts;
    if (active) q = a.q[qi];
    sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
    sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
    sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
    float bound2 = a.bounded ? 5.0f * (1.0f + 1e-5f) : FLT_MAX;
    if (active) {
      if (a.bounded && prev_valid) bound2 = fminf(bound2, grid_carried_bound(a.prev_q[qi], sel));
      if (a.grid_hint) bound2 = fminf(bound2, a.grid_hint[qi]);
    }
    // A point that is known to have five neighbours within two cells is worth the wider probe (the ball it scans is no
    // bigger than the first probe's block); one that is not -- the first probe saw fewer than five, or its fifth far out --
    // would scan all 125 cells, hundreds of candidates, with its wavefront waiting: the tree search is the tool for those.
    const float worth = (2.0f * G.c) * (2.0f * G.c);
    const bool probe = active && bound2 < worth;
    float lb6u;
    int verdict = knn5_grid<BLOCK, 2>(G, probe, sel[0], sel[1], sel[2], bound2, a.grid_clip_margin, (lds_u32 *)(stack_lds + tid), d, p, lb6u);
    if (verdict == GRID_FAR && !a.bounded) verdict = GRID_UNPROVEN;
    const bool needy2 = active && (!probe || verdict == GRID_UNPROVEN);
    const unsigned long long m2 = __ballot(needy2);
    if (lane == 0) wave_needy2[wave] = __popcll(m2);
    __syncthreads();
    int off2 = 0, total2 = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) {
      const int c2 = wave_needy2[w];
      off2 += w < wave ? c2 : 0;
      total2 += c2;
    }
    if (needy2) {
      a.need2_list[(size_t)lb * BLOCK + off2 + __popcll(m2 & ((1ull << lane) - 1ull))] = (uint16_t)(qi - bd.first);
      if (a.grid_hint && probe && d[4] < 1.0e30f) a.grid_hint[qi] = fminf(a.grid_hint[qi], d[4] * (1.0f + 1e-5f) + 1e-12f);
    }
    if (tid == 0) {
      a.need2_cnt[lb] = (uint16_t)total2;
      if (a.cert_stats) atomicAdd(a.cert_stats + 2, (unsigned long long)total2);  // debug tap: points left to the tree search
    }
    if (active && !needy2 && a.bounded)
      a.prev_q[qi] = make_float4(sel[0], sel[1], sel[2], (verdict == GRID_PROVEN && d[4] < 5.0f) ? d[4] : FLT_MAX);
    active = active && !needy2;
    do_search = false;
    __syncthreads();  // the row table is dead: its words become the staging rows of the contraction
  }
  if (CERT == 1) {
    static_assert(CERT != 1 || (!PACKET && !CUBES && !STATE_LDS && BLOCK <= 256), "certificate pass: whole-map lane search, byte lists");
    // the last update of this scan, as the largest displacement of a point within CERT_RANGE_M of the sensor [m]
    const float dr_deg = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st->delta_r)));
    const float dt_cm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st->delta_t)));
    const float disp = prev_valid ? dt_cm * 0.01f + dr_deg * (0.017453292f * CERT_RANGE_M) : FLT_MAX;
    const bool try_cert = disp < a.cert_try_m;
    // a scan still moving by more than cert_track_m will not test certificates after its next update either: it is searched
    // without the bookkeeping of the bound (which costs ~10 % of a search), and says so per point (prev_lb = 0: no certificate)
    track = disp < a.cert_track_m;
    if (try_cert) {  // block-uniform
      __shared__ int wave_needy[NWAVE];
      const float4 *pts = is_surf ? a.ts.pts : a.tc.pts;
      const int n_pts = is_surf ? a.ts.n_pts : a.tc.n_pts;
      bool certified = false;
      if (active) {
        q = a.q[qi];
        sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
        sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
        sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
        int pp[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) pp[j] = a.prev_nb[(size_t)qi * 5 + j];
        float4 pq = a.prev_q[qi];
        pq.w = a.prev_lb[qi];
        float4 pv[5];
        bool all = true;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const bool ok = pp[j] >= 0 && pp[j] < n_pts;
          all = all && ok;
          pv[j] = pts[ok ? pp[j] : 0];
        }
        float dj[5], u = 0.0f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          dj[j] = dist2_xyz(sel[0], sel[1], sel[2], pv[j]);
          u = fmaxf(u, dj[j]);
        }
        const float ex = sel[0] - pq.x, ey = sel[1] - pq.y, ez = sel[2] - pq.z;
        const float delta = sqrtf((ex * ex + ey * ey) + ez * ez);
        const float sl = sqrtf(pq.w) * 0.99999f - delta * 1.00001f - 1e-6f;  // every other point is farther than this
        bool distinct = true;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
          for (int j = i + 1; j < 5; ++j) distinct = distinct && dj[i] != dj[j];
        certified = all && distinct && sqrtf(u) * 1.00001f < sl;
        if (certified) {  // the previous five, in ascending order of their new distances
#pragma unroll
          for (int j = 0; j < 5; ++j) { d[j] = FLT_MAX; p[j] = -1; }
#pragma unroll
          for (int j = 0; j < 5; ++j) knn_insert_sorted(d, p, dj[j], pp[j]);
          // nothing is written: the five (as a set), the position of the last SEARCH and the bound taken there stay what
          // the next certificate is measured against -- the moves since add up in delta, and shrink geometrically
        }
      }
      const bool needy = active && !certified;
      const unsigned long long m = __ballot(needy);
      if (lane == 0) wave_needy[wave] = __popcll(m);
      __syncthreads();
      int off = 0, total = 0;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) {
        const int c = wave_needy[w];
        off += w < wave ? c : 0;
        total += c;
      }
      if (needy) a.need_list[(size_t)lb * BLOCK + off + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)tid;
      if (tid == 0) {
        a.need_cnt[lb] = (uint16_t)total;
        if (a.cert_stats) {  // debug tap: points left to pass 2 / points of certificate-testing workgroups
          atomicAdd(a.cert_stats, (unsigned long long)total);
          atomicAdd(a.cert_stats + 1, (unsigned long long)bd.count);
        }
      }
      active = certified;
      do_search = false;
    } else if (tid == 0) {
      a.need_cnt[lb] = 0;
    }
  }

  if (active) {
    if (!PACKET && do_search && GRIDQ != 3) {
    q = a.q[qi];
    // util/transform_utils.h:476-482 pointAssociateToMap: it * p
    sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
    sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
    sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
    }

    bool searched = true;
    if (!PACKET) {
    if (CUBES) {  // per-lane tree: the cube the transformed point falls into (FeatureMap.h:523-526)
      const int tree = cube_tree_of(is_surf ? a.gs : a.gc, sel[0], sel[1], sel[2]);
      searched = tree >= 0;
      T = (is_surf ? a.gs.trees : a.gc.trees)[searched ? tree : 0];
    } else {  // block-uniform choice of tree
      T.nodes = is_surf ? a.ts.nodes : a.tc.nodes;
      T.pn = is_surf ? a.ts.pn : a.tc.pn;
      T.pts = is_surf ? a.ts.pts : a.tc.pts;
      T.n_pts = is_surf ? a.ts.n_pts : a.tc.n_pts;
      T.n_nodes = is_surf ? a.ts.n_nodes : a.tc.n_nodes;
      T.root_ref = is_surf ? a.ts.root_ref : a.tc.root_ref;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        T.bb_lo[i] = is_surf ? a.ts.bb_lo[i] : a.tc.bb_lo[i];
        T.bb_hi[i] = is_surf ? a.ts.bb_hi[i] : a.tc.bb_hi[i];
      }
    }
    if (CUBES && !searched) T.n_pts = 0;  // knn5_search returns at once, d[4] stays FLT_MAX
    if (GRIDQ == 2) {  // the wide probe has been here: its five, no search (and no tree to search)
      do_search = false;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        d[j] = a.wide_d[(size_t)qi * 5 + j];
        p[j] = a.wide_p[(size_t)qi * 5 + j];
      }
      q = a.q[qi];
      sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
      sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
      sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
      if (a.bounded) a.prev_q[qi] = make_float4(sel[0], sel[1], sel[2], (p[4] >= 0 && d[4] < 5.0f) ? d[4] : FLT_MAX);
    }
    if (GRIDQ != 2 && GRIDQ != 3 && do_search) {
    KdStack<BLOCK, OVF, LDS_DEPTH> stk;
    stk.lds = (lds_u32 *)(stack_lds + tid);
    stk.ovf = OVF ? a.stack_ovf + ((size_t)lb * BLOCK + tid) : nullptr;  // lb: unique per workgroup of a launch, < nb_total
    stk.ovf_stride = (size_t)a.nb_total * BLOCK;
    // Bound on the 5th neighbour's distance (production loop only; the taps run nanoflann's
    // unbounded search).  Points whose 5th neighbour is not closer than sqrt(5) m are rejected
    // by the gate below anyway (ScanMatch.cpp:102,120), and the five neighbours found in the
    // previous sweep -- any five map points -- bound the new distance from above.  Both are
    // padded by 1e-5 relative, far above the rounding of the traversal's mindistsq.
    float bound = FLT_MAX;
    if (a.bounded) {
      bound = 5.0f * (1.0f + 1e-5f);
      if (prev_valid && T.n_pts > 0 && grid) {
        bound = fminf(bound, grid_carried_bound(a.prev_q[qi], sel));
      } else if (grid) {
      } else if (prev_valid && T.n_pts > 0) {
        // the five indices, then the five points, all in flight together (a guarded load per neighbour would be a chain
        // of ten dependent round trips at the head of every wavefront): an invalid index reads point 0 and is ignored
        int pp[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) pp[j] = a.prev_nb[(size_t)qi * 5 + j];
        float4 pv[5];
        bool all = true;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const bool ok = pp[j] >= 0 && pp[j] < T.n_pts;
          all = all && ok;
          pv[j] = T.pts[ok ? pp[j] : 0];
        }
        float u = 0.0f;
#pragma unroll
        for (int j = 0; j < 5; ++j) u = fmaxf(u, dist2_xyz(sel[0], sel[1], sel[2], pv[j]));
        if (all) bound = fminf(bound, u * (1.0f + 1e-5f) + 1e-12f);
      }
    }
    // grid sweep: the probe that could not PROVE its five has still SEEN five (usually): their fifth distance bounds the true
    // one from above, and a search bounded that tightly visits a fraction of what one bounded by the gate alone does
    if (grid && a.grid_hint) bound = fminf(bound, a.grid_hint[qi]);
#ifdef LSLAM_TRAVERSAL_STATS
    TravStats ts = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    knn5_search<BLOCK, OVF, LDS_DEPTH>(T, sel[0], sel[1], sel[2], d, p, stk, ts, bound);
    if (a.bounded) {
#pragma unroll
      for (int j = 0; j < 5; ++j) a.prev_nb[(size_t)qi * 5 + j] = p[j];
    }
    if (a.dbg) {  // per-lane stats: [N][8] after the per-wave stamps
      uint64_t *o = a.dbg + (size_t)a.nb_total * NWAVE * 4 + ((size_t)lb * BLOCK + tid) * 8;
      o[0] = ts.t_desc; o[1] = ts.t_leaf; o[2] = ts.t_pop; o[3] = ts.n_node | ((uint64_t)ts.n_cull << 32);
      o[4] = ts.n_leaf | ((uint64_t)ts.n_hit << 32); o[5] = ts.n_pop | ((uint64_t)ts.n_cand << 32); o[6] = ((uint64_t)ts.n_popit << 32) | ts.n_take; o[7] = ts.t_take | (1ull << 63);
    }
#else
    if (CERT && !grid && track) {  // block-uniform
      float slb = 0.0f;
      knn5_search<BLOCK, OVF, LDS_DEPTH, true>(T, sel[0], sel[1], sel[2], d, p, stk, bound, &slb);
      // the bound is only worth keeping when the five are the true five: all found, inside the gate the bounded search is exact in
      a.prev_q[qi] = make_float4(sel[0], sel[1], sel[2], 0.0f);
      a.prev_lb[qi] = (p[4] >= 0 && d[4] < 5.0f) ? slb * 0.9999f : 0.0f;
    } else {
#ifdef LSLAM_EXP_PASS2_CLASS  // TIMING EXPERIMENT ONLY (wrong results): the second pass's searches by how tightly they are bounded -- 1: only
      // the lanes with a bound below LSLAM_EXP_PASS2_SPLIT search, 2: only the others; 0: all (the histogram of the bounds only)
      if (grid) {
        if (LSLAM_EXP_PASS2_CLASS == 0) {  // (the histogram's atomics cost a tenth of the sweep: not in the timing variants)
          const float edges[5] = {0.5f, 1.0f, 2.0f, 4.99f, 1.0e30f};
#pragma unroll
          for (int b = 0; b < 5; ++b) {
            const unsigned long long m = __ballot(active && bound < edges[b] && (b == 0 || bound >= edges[b - 1]));
            if (lane == 0 && m) atomicAdd(&g_pass2_hist[b], (unsigned long long)__popcll(m));
          }
        }
        const bool tight = bound < (float)(LSLAM_EXP_PASS2_SPLIT);
        if (LSLAM_EXP_PASS2_CLASS == 1 ? tight : (LSLAM_EXP_PASS2_CLASS == 2 ? !tight : true))
          knn5_search<BLOCK, OVF, LDS_DEPTH>(T, sel[0], sel[1], sel[2], d, p, stk, bound);
      } else
#endif
      knn5_search<BLOCK, OVF, LDS_DEPTH, false, (GRIDQ == 1 && LDS_DEPTH <= 16) ? 1 : 0>(T, sel[0], sel[1], sel[2], d, p, stk, bound);
      // no bound kept: no certificate for this point in the next sweep (the first sweep of a certificate loop is this kernel
      // WITHOUT the certificate code -- launch_sweep -- which is 4 % faster at searching than the one with it)
      if ((CERT || !CUBES) && a.prev_lb) a.prev_lb[qi] = 0.0f;
    }
    if (grid) {
      if (a.bounded) a.prev_q[qi] = make_float4(sel[0], sel[1], sel[2], (p[4] >= 0 && d[4] < 5.0f) ? d[4] : FLT_MAX);
    } else if (a.bounded) {
#pragma unroll
      for (int j = 0; j < 5; ++j) a.prev_nb[(size_t)qi * 5 + j] = p[j];
    }
#endif
    }  // do_search
    }  // !PACKET
    (void)searched;
    if (a.dbg) dbg_t1 = __builtin_readcyclecounter();

    // (GRIDQ == 2 with trees -- a.grid == 1: the neighbours sweep_refill_kernel left are positions in the tree's point array)
    point_residual(a, bd, is_surf, ((GRIDQ == 2 && a.grid == 2) || GRIDQ == 3) ? (is_surf ? a.ks.pts : a.kc.pts) : T.pts, q, sel, d, p, sc, row, rb, kept, matched, score);
  }

  if (a.dbg) dbg_t2 = __builtin_readcyclecounter();
  block_accumulate<BLOCK, FUSE, CERT == 2>(jtj_mode, is_surf, row, rb, kept, matched, score, stack_lds, red, partial_out);
  if (a.dbg && lane == 0) {  // per-wave phase stamps (shader clock)
    uint64_t *o = a.dbg + ((size_t)lb * NWAVE + wave) * 4;
    o[0] = dbg_t0;
    o[1] = dbg_t1;
    o[2] = dbg_t2;
    o[3] = __builtin_readcyclecounter();
  }
}

__global__ __launch_bounds__(SOLVE_THREADS) void solve_kernel(SolveArgs a) {
  GNState *st = a.states + blockIdx.x;  // one block per scan of the batch
  if (a.reduce_only == 2 ? !st->converged : st->done) return;
  const ProbBlocks pb = a.probs[blockIdx.x];
  const float *partials = a.partials + (size_t)pb.first_block * NCOL;
  const int nb = pb.n_blocks;
  __shared__ double red[SOLVE_GROUPS][NCOL];
  __shared__ double tot[NCOL];
  __shared__ GnShared sh;
  __shared__ int go;
  const int tid = threadIdx.x, col = tid & 31, grp = tid >> 5;
  if (tid == 0) st->clk[0] = wall_clock64();
  if (a.ext_sums) {  // sharded points: the sums were reduced per rank and all-reduced by the caller
    if (tid < NCOL) {
      const double v = a.ext_sums[(size_t)blockIdx.x * NCOL + tid];
      tot[tid] = v;
      st->sums[tid] = v;
    }
    __syncthreads();
  } else {
    reduce_partials<SOLVE_THREADS, false>(partials, nb, red);
    if (a.partials2) {  // stereo blocks of the joint system: same pattern, after the LiDAR blocks
      double s = red[grp][col];  // (this thread's own group)
      for (int b = grp; b < a.n_blocks2; b += SOLVE_GROUPS) s += (double)a.partials2[(size_t)b * NCOL + col];
      red[grp][col] = s;
    }
    __syncthreads();
    if (tid < NCOL) {
      double v = 0.0;
  #pragma unroll
      for (int g = 0; g < SOLVE_GROUPS; ++g) v += red[g][tid];
      tot[tid] = v;
      st->sums[tid] = v;
      if (a.sums_out) a.sums_out[(size_t)blockIdx.x * NCOL + tid] = v;
    }
    __syncthreads();
  }
  if (a.reduce_only) return;
  SolveParams p;
  p.max_iterations = a.max_iterations;
  p.min_rows = a.min_rows;
  p.too_few_continue = a.too_few_continue;
  p.nan_reset = a.nan_reset;
  p.delta_r_abort = a.delta_r_abort;
  p.delta_t_abort = a.delta_t_abort;
  p.eig_thresh = a.eig_thresh;
  solve_finish(st, tot, sh, go, p);
}

// PACKET: the 5-NN search is the wave-cooperative one of lslam_packet.hpp (see sweep_body).
// FUSE: the block that retires a scan's last record of the launch runs that scan's reduction + solve (SweepTail).
// CERT = 1: pass 1 of the certificate sweep (sweep_body); sweep_queue_kernel is pass 2.
template <int BLOCK, bool OVF, bool CUBES, int LDS_DEPTH, bool PACKET = false, bool FUSE = false, int CERT = 0>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(LDS_DEPTH <= 16 ? LSLAM_SHALLOW_OCC : 2))) void sweep_kernel(SweepArgs a, int jtj_mode) {
  const int lb = xcd_remap(blockIdx.x, a.nb_total);
  const BlockDesc bd = a.blocks[lb];
  GNState *st = const_cast<GNState *>(a.states) + bd.prob;
  // this scan's loop already ended (ScanMatch.cpp:144,259); the _fineScore re-sweep visits the converged scans only
  if (a.fine_gate_c >= 0.0f ? !st->converged : st->done) return;
  __shared__ float red[BLOCK / 64][NCOL];
  // The MFMA staging of [J | b] (8 floats per point) lives in the wavefront's OWN traversal-stack
  // slots -- word row c, lane slot p -- which are dead once its 5-NN searches are over: no extra LDS,
  // no cross-wavefront hazard (the shallow variant then fits six workgroups per CU instead of four).
  __shared__ uint32_t stack_lds[2 * LDS_DEPTH * BLOCK];
  sweep_body<BLOCK, OVF, CUBES, LDS_DEPTH, PACKET, false, FUSE, CERT>(a, jtj_mode, lb, bd, st, stack_lds, red, a.partials + (size_t)lb * NCOL, a.prev_valid);
  if (FUSE) {
    static_assert(!FUSE || (2 * LDS_DEPTH * BLOCK * 4 >= (int)((SOLVE_GROUPS + 1) * NCOL * sizeof(double) + sizeof(GnShared) + 16)), "the tail's LDS lives in the stack");
    __shared__ int last;
    // every wavefront's record stores are at the coherence point before the ticket is taken
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const ProbBlocks pb = a.tail.probs[bd.prob];
    if (threadIdx.x == 0) {
      // acq_rel: the records published above are ordered before the ticket, and the last block's reads of the others' after it
      const int ticket = __hip_atomic_fetch_add(a.tail.count + bd.prob, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      last = ticket == pb.n_blocks - 1;
      if (last) __hip_atomic_store(a.tail.count + bd.prob, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
    }
    __syncthreads();
    if (!last) return;
    // the traversal stack is dead: the reduction's 32 x 32 doubles, the totals and the solve's 6 x 6 system live in it
    double (*redd)[NCOL] = reinterpret_cast<double (*)[NCOL]>(stack_lds);
    double *tot = &redd[SOLVE_GROUPS][0];
    GnShared &sh = *reinterpret_cast<GnShared *>(tot + NCOL);
    int &go = *reinterpret_cast<int *>(reinterpret_cast<char *>(&sh) + sizeof(GnShared));
    const int tid = threadIdx.x;
    if (tid == 0) st->clk[0] = wall_clock64();
    reduce_partials<BLOCK, true>(a.tail.partials_abs + (size_t)pb.first_block * NCOL, pb.n_blocks, redd);
    __syncthreads();
    if (tid < NCOL) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < SOLVE_GROUPS; ++g) v += redd[g][tid];
      tot[tid] = v;
      st->sums[tid] = v;
    }
    __syncthreads();
    solve_finish(st, tot, sh, go, a.tail.sp);
  }
}

// Pass 2 of the certificate sweep (sweep_body), in two launches.
// cert_plan_kernel, one wavefront per group: how many chunks of BLOCK listed points the group's pass-1 workgroups left (none for
// a scan whose loop has ended), and that many work items (group, chunk) appended to the work list.  The order of the list
// depends on the order of the atomics; nothing else does -- an item's sums go to a place of its own.
// level 0: the first-level lists (need_cnt); when second-level lists exist their counts are zeroed here, so that a block no
// work item of the second probe is made for reads as empty.  level 1: the second-level lists (need2_cnt).
LSLAM_DEV void grid_prefix_block(const SweepArgs &a, int *part);
__global__ __launch_bounds__(256) void cert_plan_kernel(SweepArgs a, CertPlan plan, int level, int with_prefix) {
  // a map without trees (a.grid == 2): the exclusive prefix of the list lengths that sweep_wide_kernel needs rides in one more
  // workgroup of this launch instead of a launch of its own (grid_prefix_kernel: five launches per sweep of a mapping frame)
  if (with_prefix && blockIdx.x == gridDim.x - 1) {
    __shared__ int part[256];
    grid_prefix_block(a, part);
    return;
  }
  // one WAVEFRONT per group: its lanes share the group's list lengths (a group is up to CERT_GROUP workgroups: one thread
  // adding them up one after the other was a chain of that many loads)
  const int g = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the counters of the NEXT plan (they alternate)
    *plan.count_next = 0;
    *plan.ticket_next = 0;
    if (plan.ticket2_next) *plan.ticket2_next = 0;
  }
  // debug tap of the grid sweep (lslam_opts.debug_stats): points listed for pass 2 / points swept, in total and by feature type
  // and sweep of the loop.  Counted HERE, from the counts pass 1 left -- per workgroup of this (tiny) launch a histogram in
  // LDS, then one atomic per non-empty slot -- so that a run with the tap on times the same sweep kernel as one without
  const bool tap = a.cert_stats != nullptr && a.grid && !level;  // launch-uniform
  __shared__ unsigned int hist[CERT_STATS_WORDS];
  if (tap) {
    if (threadIdx.x < CERT_STATS_WORDS) hist[threadIdx.x] = 0;
    __syncthreads();
  }
  bool on = g < a.n_groups;  // wave-uniform
  GroupDesc gd{};
  if (on) {
    gd = a.groups[g];
    // the scans this sweep works on: the running loops -- or, for the _fineScore re-sweep, the converged ones (sweep_kernel)
    if (a.fine_gate_c >= 0.0f ? !a.states[gd.prob].converged : a.states[gd.prob].done) on = false;
  }
  if (on) {
    const int fb = gd.first_block - a.group_block_base;
    int total = 0, swept = 0;
    for (int k = lane; k < gd.n_blocks; k += 64) {
      total += level ? (int)a.need2_cnt[fb + k] : (int)a.need_cnt[fb + k];
      if (!level && a.need2_cnt) a.need2_cnt[fb + k] = 0;
      if (tap) swept += a.blocks[fb + k].count;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      total += __shfl_xor(total, o, 64);
      swept += __shfl_xor(swept, o, 64);
    }
    if (tap && lane == 0) {
      const int slot = CERT_STATS_BY_SWEEP + 2 * ((a.blocks[fb].is_surf ? GRID_STATS_SWEEPS : 0) + min(a.states[gd.prob].sweeps, GRID_STATS_SWEEPS - 1));
      atomicAdd(&hist[0], (unsigned)total);
      atomicAdd(&hist[1], (unsigned)swept);
      atomicAdd(&hist[slot], (unsigned)total);
      atomicAdd(&hist[slot + 1], (unsigned)swept);
    }
    const int nch = (total + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
    if (nch > 0) {
      int at = 0;
      if (lane == 0) at = atomicAdd(plan.count, nch);
      at = __shfl(at, 0, 64);
      for (int c = lane; c < nch; c += 64) plan.work[at + c] = g * CERT_GROUP + c;
    }
  }
  if (tap) {
    __syncthreads();
    if (threadIdx.x < CERT_STATS_WORDS && hist[threadIdx.x]) atomicAdd(a.cert_stats + threadIdx.x, (unsigned long long)hist[threadIdx.x]);
  }
}

// sweep_queue_kernel, a resident grid that deals the work items round-robin: item (g, c) searches points [c BLOCK, (c + 1)
// BLOCK) of the concatenation of the lists of group g's pass-1 workgroups, and ADDS its sums to the record of pass-1
// workgroup first_block + c (written one launch earlier; nobody else adds to it).  A grid of one workgroup per POSSIBLE item
// -- as many as the sweep itself has, nearly all of them leaving at once -- cost 0.25 ms per launch, more than the last
// sweeps of a batch themselves.  The argument block is re-read through a laundered pointer in every turn: left to itself the
// compiler carries the sweep's invariants across the loop in registers it does not have (124 bytes of scratch per lane).
// LIST2: the work items are chunks of the SECOND-level lists (need2_list: point offsets from the group's first point).
template <int BLOCK, bool OVF, int LDS_DEPTH, int GRIDQ = 0, bool LIST2 = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(GRIDQ == 3 ? 3 : (LDS_DEPTH <= 16 ? LSLAM_SHALLOW_OCC : 2)))) void sweep_queue_kernel(const SweepArgs a_in, const int jtj_mode_in, const CertPlan plan) {
  __shared__ float red[BLOCK / 64][NCOL];
  __shared__ uint32_t stack_lds[2 * LDS_DEPTH * BLOCK];
  __shared__ int next_w;
  __shared__ int pre_lds[CERT_GROUP + 1];
  __shared__ int pre_wave[BLOCK / 64];
  const int n_work = *plan.count;
  for (;;) {
    // dealt by ticket, not round-robin: an item is anything between a handful of points and BLOCK of them
    if (threadIdx.x == 0) next_w = atomicAdd(plan.ticket, 1);
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(next_w);
    if (w >= n_work) break;
    SweepArgs a = a_in;
    int jtj_mode = jtj_mode_in;
    asm volatile("" : "+s"(a.q), "+s"(a.blocks), "+s"(a.states), "+s"(a.partials), "+s"(a.prev_nb), "+s"(a.prev_q), "+s"(a.prev_lb), "+s"(jtj_mode));
    asm volatile("" : "+s"(a.tc.nodes), "+s"(a.tc.pts), "+s"(a.ts.nodes), "+s"(a.ts.pts), "+s"(a.stack_ovf), "+s"(a.need_list), "+s"(a.need_cnt), "+s"(a.groups));
    asm volatile("" : "+s"(a.need2_list), "+s"(a.need2_cnt), "+s"(a.grid_hint));
    const int item_id = __builtin_amdgcn_readfirstlane(plan.work[w]);
    const int g = item_id / CERT_GROUP, c = item_id % CERT_GROUP;
    GroupDesc gd = a.groups[g];
    gd.first_block = __builtin_amdgcn_readfirstlane(gd.first_block);  // wave-uniform by construction: into scalar registers
    gd.n_blocks = __builtin_amdgcn_readfirstlane(gd.n_blocks);
    gd.prob = __builtin_amdgcn_readfirstlane(gd.prob);
    const int fb = gd.first_block - a.group_block_base;
    // where the lists of the group's workgroups begin in their concatenation: one length per lane of the first wavefront, an
    // inclusive scan, the prefix in LDS (with groups of sixteen it was seventeen scalar registers and a chain of selects; with
    // sixty-four -- the grid sweep lists 1 - 3 % of a workgroup's points, and an item is a workgroup's worth of lanes -- it is not)
    static_assert(CERT_GROUP <= BLOCK && (CERT_GROUP & (CERT_GROUP - 1)) == 0, "one list length per thread; the search below halves");
    {
      const int k = (int)threadIdx.x, ln = k & 63, wv = k >> 6;
      int run = k < gd.n_blocks ? (LIST2 ? (int)a.need2_cnt[fb + k] : (int)a.need_cnt[fb + k]) : 0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(run, o, 64);
        if (ln >= o) run += u;
      }
      if (CERT_GROUP > 64) {  // the wavefronts' totals, then every wavefront adds what lies before it
        if (ln == 63) pre_wave[wv] = run;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) run += w < wv ? pre_wave[w] : 0;
      }
      if (k < CERT_GROUP) pre_lds[k + 1] = run;
      if (k == 0) pre_lds[0] = 0;
    }
    __syncthreads();
    const int total = pre_lds[CERT_GROUP];
    BlockDesc bd = a.blocks[fb];
    bd.prob = gd.prob;
    bd.is_surf = __builtin_amdgcn_readfirstlane(bd.is_surf);
    bd.out_base = __builtin_amdgcn_readfirstlane(bd.out_base);
    GNState *st = const_cast<GNState *>(a.states) + gd.prob;
    const int i = c * BLOCK + (int)threadIdx.x;
    int item = -1;
    if (i < total) {
      int k = 0;  // the last workgroup whose list begins at or before entry i
#pragma unroll
      for (int step = CERT_GROUP / 2; step >= 1; step >>= 1)
        if (pre_lds[k + step] <= i) k += step;
      const int p0 = pre_lds[k];
      if (LIST2) item = a.blocks[fb].first + (int)a.need2_list[(size_t)(fb + k) * BLOCK + (i - p0)];
      else item = a.blocks[fb + k].first + (int)a.need_list[(size_t)(fb + k) * BLOCK + (i - p0)];
    }
    sweep_body<BLOCK, OVF, false, LDS_DEPTH, false, false, false, 2, GRIDQ>(a, jtj_mode, fb + c, bd, st, stack_lds, red, a.partials + (size_t)(fb + c) * NCOL, a.prev_valid, item);  // (the certificate sweep's second pass only runs with prev_valid set; the grid sweep's runs in a loop's first sweep too)
    __syncthreads();  // red and the stack columns are free again
  }
}

// The grid sweep's second pass as TWO launches (throughput-bound batches on whole-map trees): this one only SEARCHES -- with
// persistent lanes (knn5_search_refill): a workgroup takes REFILL_CHUNKS consecutive items of the plan, i.e. up to that many
// workgroups' worth of one group's listed points, as one pool; a lane whose walk has ended hands its five in (wide_d / wide_p,
// by point) and takes the pool's next point, so a wavefront is no longer as slow as the longest of its first 64 walks with the
// lanes that drew short ones idle.  sweep_queue_kernel<.., 2> then runs the residual chain of the same items, chunk by chunk,
// its sums where the one-launch form puts them: same bits.
#ifndef LSLAM_REFILL_CHUNKS
#define LSLAM_REFILL_CHUNKS 4
#endif
#ifndef LSLAM_REFILL_OCC
#define LSLAM_REFILL_OCC LSLAM_SHALLOW_OCC
#endif
constexpr int REFILL_CHUNKS = LSLAM_REFILL_CHUNKS;
template <int BLOCK, bool OVF, int LDS_DEPTH>
struct RefillSrc {
  const SweepArgs &a;
  const float (&R)[9];
  const float (&t)[3];
  const TreeView &T;
  const int *pre;   // LDS: where the lists of the group's workgroups begin
  int *pool_next;   // LDS
  int fb, i1, prev_valid, qi;
  LSLAM_DEV bool next(float &qx, float &qy, float &qz, float &bound) {
    const int i = atomicAdd(pool_next, 1);
    if (i >= i1) return false;
    int k = 0;
#pragma unroll
    for (int step = CERT_GROUP / 2; step >= 1; step >>= 1)
      if (pre[k + step] <= i) k += step;
    qi = a.blocks[fb + k].first + (int)a.need_list[(size_t)(fb + k) * BLOCK + (i - pre[k])];
    const float4 q = a.q[qi];
    // util/transform_utils.h:476-482 pointAssociateToMap (sweep_body's statement)
    qx = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
    qy = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
    qz = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
    bound = 5.0f * (1.0f + 1e-5f);  // (this kernel runs the bounded production sweep only)
    if (prev_valid && T.n_pts > 0) {
      const float sel[3] = {qx, qy, qz};
      bound = fminf(bound, grid_carried_bound(a.prev_q[qi], sel));
    }
    if (a.grid_hint) bound = fminf(bound, a.grid_hint[qi]);
    return true;
  }
  LSLAM_DEV void emit(const float (&d)[5], const int (&p)[5]) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      a.wide_d[(size_t)qi * 5 + j] = d[j];
      a.wide_p[(size_t)qi * 5 + j] = p[j];
    }
  }
};
template <int BLOCK, bool OVF, int LDS_DEPTH>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(LSLAM_REFILL_OCC))) void sweep_refill_kernel(const SweepArgs a, const CertPlan plan) {
  __shared__ uint32_t stack_lds[2 * LDS_DEPTH * BLOCK];
  __shared__ int next_w, pool_next;
  __shared__ int pre_lds[CERT_GROUP + 1];
  __shared__ int pre_wave[BLOCK / 64];
  const int n_work = *plan.count;
  for (;;) {
    if (threadIdx.x == 0) next_w = atomicAdd(plan.ticket2, REFILL_CHUNKS);
    __syncthreads();
    int w = __builtin_amdgcn_readfirstlane(next_w);
    if (w >= n_work) break;
    const int w_end = min(w + REFILL_CHUNKS, n_work);
    while (w < w_end) {  // (wave-uniform) runs of consecutive chunks of one group: nearly always all of them
      const int id0 = __builtin_amdgcn_readfirstlane(plan.work[w]);
      int run = 1;
      while (w + run < w_end && __builtin_amdgcn_readfirstlane(plan.work[w + run]) == id0 + run) ++run;
      const int g = id0 / CERT_GROUP, c0 = id0 % CERT_GROUP;
      GroupDesc gd = a.groups[g];
      gd.first_block = __builtin_amdgcn_readfirstlane(gd.first_block);
      gd.n_blocks = __builtin_amdgcn_readfirstlane(gd.n_blocks);
      gd.prob = __builtin_amdgcn_readfirstlane(gd.prob);
      const int fb = gd.first_block - a.group_block_base;
      {  // where the group's lists begin (sweep_queue_kernel's scan)
        const int k = (int)threadIdx.x, ln = k & 63, wv = k >> 6;
        int sum = k < gd.n_blocks ? (int)a.need_cnt[fb + k] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int u = __shfl_up(sum, o, 64);
          if (ln >= o) sum += u;
        }
        if (CERT_GROUP > 64) {
          if (ln == 63) pre_wave[wv] = sum;
          __syncthreads();
#pragma unroll
          for (int ww = 0; ww < BLOCK / 64; ++ww) sum += ww < wv ? pre_wave[ww] : 0;
        }
        if (k < CERT_GROUP) pre_lds[k + 1] = sum;
        if (k == 0) {
          pre_lds[0] = 0;
          pool_next = c0 * BLOCK;
        }
      }
      __syncthreads();
      const int total = pre_lds[CERT_GROUP];
      const bool is_surf = __builtin_amdgcn_readfirstlane(a.blocks[fb].is_surf) != 0;
      const GNState *st = a.states + gd.prob;
      float R[9], t[3];
      {
        static_assert(offsetof(GNState, R) == 24 && offsetof(GNState, t) == 60, "GNState layout");
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
        u32x8_t w8;
        u32x4_t w4;
        asm volatile("s_load_dwordx8 %0, %2, 0x18\n\ts_load_dwordx4 %1, %2, 0x38\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w8), "=&s"(w4) : "s"(st) : "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) R[i] = __uint_as_float(w8[i]);
        R[8] = __uint_as_float(w4[0]);
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = __uint_as_float(w4[1 + i]);
      }
      TreeView T;
      T.nodes = is_surf ? a.ts.nodes : a.tc.nodes;
      T.pn = is_surf ? a.ts.pn : a.tc.pn;
      T.pts = is_surf ? a.ts.pts : a.tc.pts;
      T.n_pts = is_surf ? a.ts.n_pts : a.tc.n_pts;
      T.n_nodes = is_surf ? a.ts.n_nodes : a.tc.n_nodes;
      T.root_ref = is_surf ? a.ts.root_ref : a.tc.root_ref;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        T.bb_lo[i] = is_surf ? a.ts.bb_lo[i] : a.tc.bb_lo[i];
        T.bb_hi[i] = is_surf ? a.ts.bb_hi[i] : a.tc.bb_hi[i];
      }
      KdStack<BLOCK, OVF, LDS_DEPTH> stk;
      stk.lds = (lds_u32 *)(stack_lds + threadIdx.x);
      stk.ovf = OVF ? a.stack_ovf + ((size_t)(fb + c0) * BLOCK + threadIdx.x) : nullptr;  // (fb + c0: this workgroup's alone while it runs)
      stk.ovf_stride = (size_t)a.nb_total * BLOCK;
      RefillSrc<BLOCK, OVF, LDS_DEPTH> src{a, R, t, T, pre_lds, &pool_next, fb, min((c0 + run) * BLOCK, total), a.prev_valid, 0};
      knn5_search_refill<BLOCK, OVF, LDS_DEPTH>(T, stk, src);
      __syncthreads();  // pre_lds, pool_next and the stack columns are free again
      w += run;
    }
  }
}

// start/stop (optional) time exactly this dispatch on its own stream: the events are
// attached to the kernel's AQL packet, no extra barrier packets are enqueued.
hipError_t launch_sweep(const SweepArgs &a, int jtj_mode, hipStream_t s, hipEvent_t start,
                        hipEvent_t stop, int *variant, bool *cert_launched) {
  if (variant) *variant = -1;
  if (cert_launched) *cert_launched = false;
  if (a.nb_total <= 0) return hipSuccess;
  const dim3 g(a.nb_total), b(SWEEP_BLOCK);
  const bool cubes = a.gc.trees != nullptr;
#ifndef LSLAM_SHALLOW_DEPTH
#define LSLAM_SHALLOW_DEPTH 12
#endif
  constexpr int SHALLOW = LSLAM_SHALLOW_DEPTH;  // LDS levels of the production (bounded) sweep
  // A launch with more wavefronts than two per SIMD (256 CUs x 4 SIMDs) is throughput bound:
  // take the shallow-stack kernel (4 workgroups per CU).  A smaller launch is latency bound
  // and every wavefront is resident anyway: keep the whole stack in LDS.
  const bool many_waves = (long)a.nb_total * (SWEEP_BLOCK / 64) > 2 * 1024;
  const bool deep_tree = a.deep_tree != 0;
  // The stack shape is a matter of speed only -- both kernels hold the same entries, the shallow one
  // keeps levels >= 12 in HBM -- so a caller may force either (LSLAM_STACK_DEEP / _SHALLOW in the search
  // mode, LSLAM_FORCE_STACK in the environment): the parity tests run every comparison against the
  // oracle through both, whatever the size of their launch.
  const bool shallow_ok = a.stack_ovf != nullptr && !cubes;
#ifdef LSLAM_TRAVERSAL_STATS
  const bool cert = false;
#else
  // pass 1 of the certificate sweep (sweep_body); a loop's first sweep has nothing to certify and nothing to keep a bound for
  const bool cert = a.prev_q != nullptr && a.need_cnt != nullptr && a.bounded && !cubes && a.prev_valid;
#endif
  const bool shallow = shallow_ok && (a.stack_mode == SWEEP_STACK_SHALLOW ||
                                      (a.stack_mode == SWEEP_STACK_AUTO && a.bounded && (many_waves || deep_tree)));
  int v;
  if (a.packet && !cubes && a.stack_ovf && a.tc.pn && a.ts.pn) {
    v = SWEEP_VARIANT_PACKET;
    hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, true, false, 4, true>), g, b, 0, s, start, stop, 0, a, jtj_mode);
  } else if (shallow) {
    v = SWEEP_VARIANT_SHALLOW;
    if (cert) hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, true, false, SHALLOW, false, false, 1>), g, b, 0, s, start, stop, 0, a, jtj_mode);
    else hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, true, false, SHALLOW>), g, b, 0, s, start, stop, 0, a, jtj_mode);
  } else if (cubes && a.stack_ovf && deep_tree) {
    v = SWEEP_VARIANT_CUBES_OVF;
    hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, true, true, KD_STACK_LDS>), g, b, 0, s, start, stop, 0, a, jtj_mode);
  } else if (cubes) {
    v = SWEEP_VARIANT_CUBES;
    hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, false, true, KD_STACK_LDS>), g, b, 0, s, start, stop, 0, a, jtj_mode);
  } else if (a.stack_ovf && deep_tree) {
    if (a.tail.count) {
      v = SWEEP_VARIANT_DEEP_FUSED;
      hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, true, false, KD_STACK_LDS, false, true>), g, b, 0, s, start, stop, 0, a, jtj_mode);
    } else {
      v = SWEEP_VARIANT_DEEP_OVF;
      if (cert) hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, true, false, KD_STACK_LDS, false, false, 1>), g, b, 0, s, start, stop, 0, a, jtj_mode);
      else hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, true, false, KD_STACK_LDS>), g, b, 0, s, start, stop, 0, a, jtj_mode);
    }
  } else if (a.tail.count) {  // latency-bound launch: the solve rides in its tail
    v = SWEEP_VARIANT_DEEP_FUSED;
    hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, false, false, KD_STACK_LDS, false, true>), g, b, 0, s, start, stop, 0, a, jtj_mode);
  } else {
    v = SWEEP_VARIANT_DEEP;
    if (cert) hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, false, false, KD_STACK_LDS, false, false, 1>), g, b, 0, s, start, stop, 0, a, jtj_mode);
    else hipExtLaunchKernelGGL((sweep_kernel<SWEEP_BLOCK, false, false, KD_STACK_LDS>), g, b, 0, s, start, stop, 0, a, jtj_mode);
  }
  if (variant) *variant = v;
  // pass 1 of the certificate sweep ran iff one of the three instantiations that have one was taken with `cert`
  if (cert_launched) *cert_launched = cert && (v == SWEEP_VARIANT_SHALLOW || v == SWEEP_VARIANT_DEEP_OVF || v == SWEEP_VARIANT_DEEP);
  return hipGetLastError();
}

// the planner of a second pass on its own (a map without trees plans BEFORE the wide probe, whose prefix rides in the same launch)
hipError_t launch_sweep_plan(const SweepArgs &a, hipStream_t s, const CertPlan &plan, int level, bool with_prefix) {
  if (a.n_groups <= 0) return hipSuccess;
  hipLaunchKernelGGL(cert_plan_kernel, dim3((a.n_groups + 3) / 4 + (with_prefix ? 1 : 0)), dim3(256), 0, s, a, plan, level, with_prefix ? 1 : 0);
  return hipGetLastError();
}

// the grid sweep's second pass in its two-launch form (sweep_refill_kernel, then the residual chain of the same items): the
// shallow-stack shape on whole-map trees only
hipError_t launch_sweep_refill(const SweepArgs &a, int jtj_mode, hipStream_t s, hipEvent_t stop, const CertPlan &plan) {
  if (a.n_groups <= 0) return hipSuccess;
  constexpr int SHALLOW = LSLAM_SHALLOW_DEPTH;
  hipLaunchKernelGGL(cert_plan_kernel, dim3((a.n_groups + 3) / 4), dim3(256), 0, s, a, plan, 0, 0);
  const long possible = a.active_blocks ? std::max<long>(a.n_active, 1) : (long)a.nb_total;
  const dim3 b(SWEEP_BLOCK);
  hipLaunchKernelGGL((sweep_refill_kernel<SWEEP_BLOCK, true, SHALLOW>), dim3((unsigned)std::min<long>((possible + REFILL_CHUNKS - 1) / REFILL_CHUNKS, 256 * LSLAM_REFILL_OCC)), b, 0, s, a, plan);
  hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, SHALLOW, 2>), dim3((unsigned)std::min<long>(possible, 256 * 5)), b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
  return hipGetLastError();
}

hipError_t launch_sweep_queue(const SweepArgs &a, int jtj_mode, hipStream_t s, hipEvent_t stop, int variant, const CertPlan &plan, int level, bool planned) {
  if (a.n_groups <= 0) return hipSuccess;
  if (!planned) hipLaunchKernelGGL(cert_plan_kernel, dim3((a.n_groups + 3) / 4), dim3(256), 0, s, a, plan, level, 0);
  // as many workgroups as stay resident (256 CUs x five of the shallow kernel, two of the deep ones), never more than items possible
  constexpr int SHALLOW = LSLAM_SHALLOW_DEPTH;
  const long possible = a.active_blocks ? std::max<long>(a.n_active, 1) : (long)a.nb_total;  // (a work item is at most one per pass-1 workgroup that ran)
  const dim3 b(SWEEP_BLOCK);
  if (a.grid == 1 && a.need2_cnt && level == 0) {  // the grid sweep's second probe (no tree search in it: its LDS is the 25-run row table)
    const dim3 g((unsigned)std::min<long>(possible, 256 * 3));
    hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, 25, 3>), g, b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
    return hipGetLastError();
  }
  if (a.grid == 1 && a.need2_cnt && level == 1) {  // ... and the tree search of what it listed
    if (variant == SWEEP_VARIANT_SHALLOW) {
      hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, SHALLOW, 1, true>), dim3((unsigned)std::min<long>(possible, 256 * 5)), b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
    } else if (variant == SWEEP_VARIANT_DEEP_OVF) {
      hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, KD_STACK_LDS, 1, true>), dim3((unsigned)std::min<long>(possible, 256 * 2)), b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
    } else if (variant == SWEEP_VARIANT_DEEP) {
      hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, false, KD_STACK_LDS, 1, true>), dim3((unsigned)std::min<long>(possible, 256 * 2)), b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
    } else {
      return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  if (a.grid == 2) {  // neighbours given (a map without trees): no search, no stack -- the shallow shape's LDS is plenty
    const dim3 g((unsigned)std::min<long>(possible, 256 * 5));
    hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, SHALLOW, 2>), g, b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
  } else if (variant == SWEEP_VARIANT_SHALLOW) {
    const dim3 g((unsigned)std::min<long>(possible, 256 * 5));
    if (a.grid) hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, SHALLOW, 1>), g, b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
    else hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, SHALLOW>), g, b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
  } else if (variant == SWEEP_VARIANT_DEEP_OVF) {
    const dim3 g((unsigned)std::min<long>(possible, 256 * 2));
    if (a.grid) hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, KD_STACK_LDS, 1>), g, b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
    else hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, true, KD_STACK_LDS>), g, b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
  } else if (variant == SWEEP_VARIANT_DEEP) {
    const dim3 g((unsigned)std::min<long>(possible, 256 * 2));
    if (a.grid) hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, false, KD_STACK_LDS, 1>), g, b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
    else hipExtLaunchKernelGGL((sweep_queue_kernel<SWEEP_BLOCK, false, KD_STACK_LDS>), g, b, 0, s, nullptr, stop, 0, a, jtj_mode, plan);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The grid sweep: pass 1 of a two-pass sweep whose 5-NN search is the cell-grid probe of lslam_grid.hpp.  One lane per scan
// point as in sweep_kernel, but no traversal stack and no divergent descent: every lane of a wavefront runs the same
// candidate loop.  A point whose five neighbours the probe PROVES goes through the residual chain right here; the others
// (sparse neighbourhoods, exact distance ties: a few per cent) are listed the way the certificate sweep lists its searches --
// one byte per point at a fixed place per workgroup -- and sweep_queue_kernel searches the tree for them, every lane busy,
// its sums added to this workgroup's record.  Fixed places, fixed order: the same bits in every run.
// ---------------------------------------------------------------------------
#ifndef LSLAM_GRID_OCC
#define LSLAM_GRID_OCC 7  // wavefronts per SIMD the grid sweep is compiled for: 72 VGPRs + 36 B of scratch (6: 80 VGPRs + 8 B, 1.6 % slower since round 6 -- it had been the faster one before the loop was hand-scheduled; 5: 88 VGPRs, 5 % slower; 8: 64 VGPRs + 96 B of scratch, 8 % slower)
#endif
#ifndef LSLAM_STAGE_PAD
#define LSLAM_STAGE_PAD 0
#endif
#ifdef LSLAM_EXP_SECTION_CLOCK
__device__ unsigned long long g_section_clock[12 * 64];  // [section][place]: s_memtime ticks; 9 = wavefronts reporting, 10 / 11 = candidates, 64 x rounds
#endif
LSLAM_DEV void wide_probe(const CellGrid &G, const float (&sel)[3], float r2, int lane, float nf_slack, float (&d)[5], int (&p)[5], bool &num, bool &bad);

// WIDE (A/B, LSLAM_AB_WIDE_IN_PLACE; a map without kd-trees, a launch too small to fill the chip -- the mapping node's frame):
// a point the probe cannot prove is resolved on the spot by its own wavefront (wide_probe: every cell within the fifth
// distance the probe saw), and the residual chain then runs ONCE for all 64 lanes.  One launch per sweep instead of five, no
// lists; the workgroup's sums are formed exactly as the lane search's sweep forms them.  Slower all the same (0.84 ms against
// 0.49 per scan match of a frame): the wavefront's unproven points are probed one after the other, ~15 us each.
// (Tried for single-scan launches and dropped: the candidate loop row after row with four points in flight instead of the
// lock-step one-candidate-per-round loop -- same bits, no faster, 27 - 32 us against 23 - 31: a launch of one wavefront per
// SIMD is a chain of latencies of which the loop is only one.)
// FITC: the instantiation with the fit cache (LSLAM_AB_FIT_CACHE) -- a template argument, not a run-time branch: with the cache's
// code in it the kernel everybody runs was 3.5 % slower (16 bytes of scratch, twelve more scalar registers, a longer program)
template <int BLOCK, bool WIDE = false, bool FITC = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WIDE ? 4 : LSLAM_GRID_OCC))) void sweep_grid_kernel(SweepArgs a, int jtj_mode) {
  constexpr int NWAVE = BLOCK / 64;
  const int lb = a.active_blocks ? a.active_blocks[xcd_remap(blockIdx.x, a.n_active)] : xcd_remap(blockIdx.x, a.nb_total);
  const BlockDesc bd = a.blocks[lb];
  const GNState *st = a.states + bd.prob;
  if (a.fine_gate_c >= 0.0f ? !st->converged : st->done) return;  // (cert_plan_kernel skips the groups of such scans)
  __shared__ float red[NWAVE][NCOL];
  __shared__ uint32_t rows_lds[18 * BLOCK];  // nine (first, end) runs per lane; afterwards the MFMA staging rows
  __shared__ int wave_needy[NWAVE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_surf = bd.is_surf != 0;
  const int qi = bd.first + tid;
  const bool active = tid < bd.count;
#ifdef LSLAM_EXP_SECTION_CLOCK
  SecClock lslam_sc;
  SecClock *const scp = &lslam_sc;
  {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lslam_sc.last)::"memory");
#pragma unroll
    for (int i = 0; i < 10; ++i) lslam_sc.acc[i] = 0;
    lslam_sc.cand_sum = lslam_sc.cand_rounds = 0;
  }
#define LSLAM_SEC_ARG , scp
#else
#define LSLAM_SEC_ARG
#endif

  float R[9], t[3], sc[6];
  {
    static_assert(offsetof(GNState, R) == 24 && offsetof(GNState, t) == 60 && offsetof(GNState, sc) == 72, "GNState layout");
    typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x16_t w;
    u32x2_t w2;
    asm volatile("s_load_dwordx16 %0, %2, 0x18\n\ts_load_dwordx2 %1, %2, 0x58\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(w), "=&s"(w2)
                 : "s"(st)
                 : "memory");
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = __uint_as_float(w[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = __uint_as_float(w[9 + i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) sc[i] = __uint_as_float(w[12 + i]);
    sc[4] = __uint_as_float(w2[0]);
    sc[5] = __uint_as_float(w2[1]);
  }
  CellGrid G;  // block-uniform choice
  G.cell_start = is_surf ? a.ks.cell_start : a.kc.cell_start;
  G.pts = is_surf ? a.ks.pts : a.kc.pts;
#pragma unroll
  for (int i = 0; i < 3; ++i) G.org[i] = is_surf ? a.ks.org[i] : a.kc.org[i];
  G.inv_c = is_surf ? a.ks.inv_c : a.kc.inv_c;
  G.c = is_surf ? a.ks.c : a.kc.c;
  G.nx = is_surf ? a.ks.nx : a.kc.nx;
  G.ny = is_surf ? a.ks.ny : a.kc.ny;
  G.nz = is_surf ? a.ks.nz : a.kc.nz;
  G.n_pts = is_surf ? a.ks.n_pts : a.kc.n_pts;

  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) q = a.q[qi];
  float sel[3];
  // util/transform_utils.h:476-482 pointAssociateToMap
  sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
  sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
  sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
  // the bound of the production loop: the acceptance gate, and the previous sweep's five seen from the new position
  float bound = FLT_MAX;
  if (a.bounded) {
    bound = 5.0f * (1.0f + 1e-5f);
    if (a.prev_valid && active) bound = fminf(bound, grid_carried_bound(a.prev_q[qi], sel));
  }
  float d[5], lb6;
  int p[5];
#ifdef LSLAM_EXP_SETUP_TWICE  // TIMING EXPERIMENT ONLY (same results): the probe's eighteen cell-table loads issued once more, one cell row higher
  {
    const float ux = (sel[0] - G.org[0]) * G.inv_c, uy = (sel[1] - G.org[1]) * G.inv_c, uz = (sel[2] - G.org[2]) * G.inv_c;
    const bool inr = ux >= 2.0f && ux < (float)(G.nx - 2) && uy >= 2.0f && uy < (float)(G.ny - 2) && uz >= 2.0f && uz < (float)(G.nz - 3);
    const int ix = inr ? (int)ux : 2, iy = inr ? (int)uy : 2, iz = inr ? (int)uz + 1 : 2;
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int base = G.nx * ((iy + r % 3 - 1) + G.ny * (iz + r / 3 - 1));
      acc += G.cell_start[base + ix - 1] ^ G.cell_start[base + ix + 2];
    }
    asm volatile("" ::"v"(acc));
  }
#endif
#ifdef LSLAM_EXP_SLEEP  // TIMING EXPERIMENT ONLY (same results): every wavefront parked for 64 x LSLAM_EXP_SLEEP cycles -- what a microsecond of latency costs
  __builtin_amdgcn_s_sleep(LSLAM_EXP_SLEEP);
#endif
  LSLAM_TICK_P(scp, 0);  // block and state fetched, the point, its transform, the carried bound
  int verdict = knn5_grid<BLOCK>(G, active, sel[0], sel[1], sel[2], bound, a.grid_clip_margin, (lds_u32 *)(rows_lds + tid), d, p, lb6 LSLAM_SEC_ARG);
  LSLAM_TICK_P(scp, 3);  // the six survivors fetched again, exact distances, order, proof
  // beyond the gate nothing is looked up (ScanMatch.cpp:102,120); the taps and the _fineScore re-sweep want nanoflann's answer
  if (verdict == GRID_FAR && !a.bounded) verdict = GRID_UNPROVEN;
  bool needy = active && verdict == GRID_UNPROVEN;
  bool resolved_wide = false;
  if (WIDE) {
    unsigned long long todo = __ballot(needy);
    if (a.cert_stats && lane == 0) atomicAdd(a.cert_stats, (unsigned long long)__popcll(todo));
    if (a.cert_stats && tid == 0) atomicAdd(a.cert_stats + 1, (unsigned long long)bd.count);
    const float my_hint = d[4] < 1.0e30f ? d[4] * (1.0f + 1e-5f) + 1e-12f : FLT_MAX;  // five map points the probe SAW lie within this
    bool any_bad = false;
    while (todo) {  // wave-uniform
      const int L = __builtin_ctzll(todo);
      todo &= todo - 1;
      const float s3[3] = {__shfl(sel[0], L, 64), __shfl(sel[1], L, 64), __shfl(sel[2], L, 64)};
      const float r2 = fminf(5.0f * (1.0f + 1e-5f), __shfl(my_hint, L, 64));
      float wd[5];
      int wp[5];
      bool num, bad;
      wide_probe(G, s3, r2, lane, a.wide_nf_slack, wd, wp, num, bad);
      any_bad = any_bad || bad;
      if (lane == L) {
#pragma unroll
        for (int j = 0; j < 5; ++j) { d[j] = wd[j]; p[j] = wp[j]; }
      }
    }
    // an exact tie (only nanoflann's traversal can order it): the host builds the trees and repeats the call
    if (any_bad && lane == 0) atomicOr(&const_cast<GNState *>(st)->pad, 1);
    resolved_wide = needy;
    needy = false;
  }
  if (!WIDE) {  // the list of pass 2 (as the certificate sweep's pass 1 writes it)
    const unsigned long long m = __ballot(needy);
    if (lane == 0) wave_needy[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) {
      const int c = wave_needy[w];
      off += w < wave ? c : 0;
      total += c;
    }
    if (needy) {
      a.need_list[(size_t)lb * BLOCK + off + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)tid;
      // what pass 2's search may take for granted: five map points within this distance (padded like every bound)
      if (a.grid_hint) a.grid_hint[qi] = d[4] < 1.0e30f ? d[4] * (1.0f + 1e-5f) + 1e-12f : FLT_MAX;
#ifdef LSLAM_EXP_COUNT_NOHINT  // EXPERIMENT: how many listed points go to the tree search without a bound from the probe (it saw fewer than five)
      if (a.cert_stats && !(d[4] < 1.0e30f)) atomicAdd(a.cert_stats + 2, 1ull);
      if (a.cert_stats && d[4] < 1.0e30f && d[4] >= 5.0f) atomicAdd(a.cert_stats + 3, 1ull);
#endif
    }
    if (tid == 0) a.need_cnt[lb] = (uint16_t)total;  // (the debug tap counts from these, in cert_plan_kernel)
  }
  const bool has = active && !needy;
  float row[6] = {0, 0, 0, 0, 0, 0};
  float rb = 0.0f, kept = 0.0f, matched = 0.0f, score = 0.0f;
  if (has) {
    // what the next sweep's bound is taken from: where the point is now and how far its fifth neighbour
    const bool fifth_known = resolved_wide ? (p[4] >= 0 && d[4] < 5.0f) : (verdict == GRID_PROVEN && d[4] < 5.0f);
    if (a.bounded) a.prev_q[qi] = make_float4(sel[0], sel[1], sel[2], fifth_known ? d[4] : FLT_MAX);
  }
  // ---- the plane fit of a surf block, with the fit cache (lslam_opts.ab_switches & LSLAM_AB_FIT_CACHE) ---------------------
  // findPlane is a pure function of the five neighbours IN ORDER (feature_utils.h:157-204), and late in a Gauss-Newton loop
  // most points keep theirs from one sweep to the next (84 % from the third to the fourth sweep of the bench's loops, 97 % to
  // the fifth; 1 % and 23 % before: tools/nb_change_stats.py).  From the sweep with index fit_from_sweep - 1 on, a proven point
  // stores its five neighbours (positions in the cell-sorted map) and its plane; from fit_from_sweep on it compares, and the
  // workgroup runs the fit only for the points whose five changed -- COMPACTED through LDS into as few wavefronts as they
  // fill, because a lone changed lane would otherwise cost its wavefront the whole fit (917 instructions).  Same bits: a
  // cached plane is the plane the same five points gave.
  bool fit_done = false;
  if (FITC && !WIDE && a.fit_ids != nullptr && is_surf && a.bounded && a.fine_gate_c < 0.0f && a.flags_out == nullptr) {  // block-uniform
    const int sweep_ix = st->sweeps;
    const bool fc_store = sweep_ix >= a.fit_from_sweep - 1, fc_use = sweep_ix >= a.fit_from_sweep && a.prev_valid;
    if (fc_store) {
      fit_done = true;
      const size_t nf = (size_t)a.n_fit;
      const bool gate = has && d[4] < 5.0f;  // ScanMatch.cpp:120
      bool same = false;
      if (fc_use && gate) {
        same = true;
#pragma unroll
        for (int j = 0; j < 5; ++j) same = same && a.fit_ids[j * nf + qi] == p[j];
      }
      const bool need_fit = gate && !same;
      float plane[4] = {0.f, 0.f, 0.f, 0.f};
      bool plane_ok = false;
      // slots of the points that need a fit, in lane order
      const unsigned long long mf = __ballot(need_fit);
      __syncthreads();  // (the list's counters above have been read by everybody: wave_needy is free again)
      if (lane == 0) wave_needy[wave] = __popcll(mf);
      __syncthreads();
      int foff = 0, ftotal = 0;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) {
        const int c = wave_needy[w];
        foff += w < wave ? c : 0;
        ftotal += c;
      }
      if (fc_use && ftotal <= BLOCK / 2) {  // block-uniform: few enough to pay for the detour
        int32_t *slot_p = reinterpret_cast<int32_t *>(rows_lds);               // [BLOCK][6]: five positions, the owner's thread
        float *slot_f = reinterpret_cast<float *>(rows_lds) + 6 * BLOCK;       // [BLOCK][5]: the plane, found
        const int myslot = foff + __popcll(mf & ((1ull << lane) - 1ull));
        if (need_fit) {
#pragma unroll
          for (int j = 0; j < 5; ++j) slot_p[myslot * 6 + j] = p[j];
          slot_p[myslot * 6 + 5] = tid;
        }
        __syncthreads();
        if (tid < ftotal) {  // the first ceil(ftotal / 64) wavefronts
          float4 nb5[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) nb5[j] = G.pts[slot_p[tid * 6 + j]];
          float pl[4];
          const bool ok = find_plane(nb5, 0.2f, pl);
          const int owner_q = bd.first + slot_p[tid * 6 + 5];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            slot_f[tid * 5 + j] = pl[j];
            a.fit_val[j * nf + owner_q] = pl[j];
          }
          slot_f[tid * 5 + 4] = ok ? 1.0f : 0.0f;
          a.fit_val[4 * nf + owner_q] = ok ? 1.0f : 0.0f;
        }
        if (same) {
#pragma unroll
          for (int j = 0; j < 4; ++j) plane[j] = a.fit_val[j * nf + qi];
          plane_ok = a.fit_val[4 * nf + qi] != 0.0f;
        }
        __syncthreads();
        if (need_fit) {
#pragma unroll
          for (int j = 0; j < 4; ++j) plane[j] = slot_f[myslot * 5 + j];
          plane_ok = slot_f[myslot * 5 + 4] != 0.0f;
        }
      } else if (gate) {  // everybody fits for itself (the sweep that fills the cache; a sweep after a large step)
        float4 nb5[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) nb5[j] = G.pts[p[j]];
        plane_ok = find_plane(nb5, 0.2f, plane);
        if (!same) {
#pragma unroll
          for (int j = 0; j < 4; ++j) a.fit_val[j * nf + qi] = plane[j];
          a.fit_val[4 * nf + qi] = plane_ok ? 1.0f : 0.0f;
        }
      }
      if (active && !same) {  // what the next sweep compares with: the five of a proven point inside the gate, else "none"
#pragma unroll
        for (int j = 0; j < 5; ++j) a.fit_ids[j * nf + qi] = gate ? p[j] : -1;
      }
      if (gate && plane_ok) {  // ScanMatch.cpp:122-139
        matched = 1.0f;
        float coeff[4];
        if (surf_coeff(plane, sel, coeff)) {
          jacobian_row(sc, q.x, q.y, q.z, coeff, row, rb);
          kept = 1.0f;
          score = expf(-fabsf(coeff[3]));
        }
      }
    }
  }
  LSLAM_TICK_P(scp, 4);  // the list of pass 2 (one workgroup barrier), the carried state written
  if (has && !fit_done) point_residual(a, bd, is_surf, G.pts, q, sel, d, p, sc, row, rb, kept, matched, score);
  LSLAM_TICK_P(scp, 5);  // the residual chain: five neighbours fetched, findLine / findPlane, coefficient, Jacobian row, score
  __syncthreads();  // every wavefront is done with its row table: the staging rows may be written
  LSLAM_TICK_P(scp, 7);  // waiting for the workgroup's slowest wavefront
  block_accumulate<BLOCK, false, false, LSLAM_STAGE_PAD>(jtj_mode, is_surf, row, rb, kept, matched, score, rows_lds, red, a.partials + (size_t)lb * NCOL);
#ifdef LSLAM_EXP_SECTION_CLOCK
  LSLAM_TICK_P(scp, 8);  // staging, the MFMA contraction, the workgroup's record
  if ((lb & 15) == 0 && lane == 0) {  // one workgroup in sixteen reports (atomics on 64 places per section)
    const int place = (lb >> 4) & 63;
#pragma unroll
    for (int i = 0; i < 9; ++i) atomicAdd(&g_section_clock[i * 64 + place], lslam_sc.acc[i]);
    atomicAdd(&g_section_clock[9 * 64 + place], 1ull);
    atomicAdd(&g_section_clock[10 * 64 + place], lslam_sc.cand_sum);
    atomicAdd(&g_section_clock[11 * 64 + place], lslam_sc.cand_rounds);
  }
#endif
#undef LSLAM_SEC_ARG
}

// The workgroups of the scans whose loop is still running, in block order: late in a batch's loops most scans have converged,
// and a launch of every workgroup of every scan -- 430 000 for the bench's 960 scans, nearly all of them leaving at once --
// costs 0.17 ms for the sweep kernel alone, 0.3 ms per trailing iteration with the second pass and the solve: 4 % of a step.
// Every workgroup scans the scans' block counts for itself (a few thousand loads), then the workgroups share the WRITING of the
// list -- entry i belongs to the scan whose run holds it, found by a search in the prefix in LDS.  (One workgroup whose
// threads wrote their own scans' runs one entry after the other took 45 us per iteration for the bench's 430 000 entries.)
constexpr int COMPACT_WGS = 64;
__global__ __launch_bounds__(1024) void compact_active_kernel(const GNState *states, const ProbBlocks *probs, int n_prob, int32_t block_base,
                                                              int32_t *active_blocks, int32_t *count_out) {
  __shared__ int incl[1024];
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n_prob + 1023) / 1024;
  const int p0 = min(n_prob, tid * per), p1 = min(n_prob, p0 + per);
  int sum = 0;
  for (int p = p0; p < p1; ++p) sum += states[p].done ? 0 : probs[p].n_blocks;
  int run = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(run, o, 64);
    if (lane >= o) run += u;
  }
  if (lane == 63) wsum[wave] = run;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int c = wsum[w];
    before += w < wave ? c : 0;
    total += c;
  }
  incl[tid] = before + run;  // blocks of the scans of threads 0 .. tid
  __syncthreads();
  for (int i = blockIdx.x * 1024 + tid; i < total; i += gridDim.x * 1024) {
    int lo = 0, hi = 1023;  // the first thread whose inclusive count exceeds i
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (incl[mid] > i) hi = mid; else lo = mid + 1;
    }
    int at = lo ? incl[lo - 1] : 0;
    const int q0 = min(n_prob, lo * per), q1 = min(n_prob, q0 + per);
    for (int p = q0; p < q1; ++p) {  // (per = 1 for up to 1 024 scans)
      const int nb = states[p].done ? 0 : probs[p].n_blocks;
      if (i < at + nb) {
        active_blocks[i] = probs[p].first_block - block_base + (i - at);
        break;
      }
      at += nb;
    }
  }
  if (blockIdx.x == 0 && tid == 0) *count_out = total;
}
hipError_t launch_compact_active(const GNState *states, const ProbBlocks *probs, int n_prob, int32_t block_base, int32_t *active_blocks,
                                 int32_t *count_out, hipStream_t s) {
  hipLaunchKernelGGL(compact_active_kernel, dim3(COMPACT_WGS), dim3(1024), 0, s, states, probs, n_prob, block_base, active_blocks, count_out);
  return hipGetLastError();
}

hipError_t launch_sweep_grid(const SweepArgs &a, int jtj_mode, hipStream_t s, hipEvent_t start, hipEvent_t stop, bool resolve_in_place) {
  if (a.nb_total <= 0) return hipSuccess;
  if (a.active_blocks && !resolve_in_place) {  // the batch's grid sweep over the running scans' workgroups only
    if (a.n_active <= 0) return hipSuccess;
    if (a.fit_ids) hipExtLaunchKernelGGL((sweep_grid_kernel<SWEEP_BLOCK, false, true>), dim3(a.n_active), dim3(SWEEP_BLOCK), 0, s, start, stop, 0, a, jtj_mode);
    else hipExtLaunchKernelGGL((sweep_grid_kernel<SWEEP_BLOCK>), dim3(a.n_active), dim3(SWEEP_BLOCK), 0, s, start, stop, 0, a, jtj_mode);
    return hipGetLastError();
  }
  if (resolve_in_place) hipExtLaunchKernelGGL((sweep_grid_kernel<SWEEP_BLOCK, true>), dim3(a.nb_total), dim3(SWEEP_BLOCK), 0, s, start, stop, 0, a, jtj_mode);
  else if (a.fit_ids) hipExtLaunchKernelGGL((sweep_grid_kernel<SWEEP_BLOCK, false, true>), dim3(a.nb_total), dim3(SWEEP_BLOCK), 0, s, start, stop, 0, a, jtj_mode);
  else hipExtLaunchKernelGGL((sweep_grid_kernel<SWEEP_BLOCK>), dim3(a.nb_total), dim3(SWEEP_BLOCK), 0, s, start, stop, 0, a, jtj_mode);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The grid sweep on a map WITHOUT kd-trees (deferred: lslam_map_defer_trees; the mapping node's per-frame map).  The points
// pass 1 could not prove -- a few per cent of a few thousand -- are resolved by the grid alone:
//   grid_prefix_kernel   exclusive prefix of the per-workgroup list lengths (one block), so that
//   sweep_wide_kernel    can give every listed point ONE WAVEFRONT: the 64 lanes share the rows of every cell within the
//                        point's bound (the fifth distance pass 1 saw, else the acceptance gate), each lane keeps the six
//                        smallest keys of its rows, six rounds of wave-minimum pick the six smallest of all, their exact
//                        distances are sorted, and the proof is the 27-cell probe's with the bound's ball as the covered
//                        region.  The five go to wide_d / wide_p; sweep_queue_kernel runs the residual chain on them.
// An exact distance tie among the six -- the one thing only nanoflann's traversal can order -- raises GNState::pad: the host
// builds the trees and repeats the call through them.
// ---------------------------------------------------------------------------
LSLAM_DEV void grid_prefix_block(const SweepArgs &a, int *part) {  // one workgroup of 256 threads
  const int tid = threadIdx.x, nb = a.nb_total;
  const int per = (nb + 255) / 256;
  const int b0 = min(nb, tid * per), b1 = min(nb, b0 + per);
  int sum = 0;
  for (int b = b0; b < b1; ++b) {
    const GNState &st = a.states[a.blocks[b].prob];
    const bool on = a.fine_gate_c >= 0.0f ? st.converged != 0 : st.done == 0;
    sum += on ? (int)a.need_cnt[b] : 0;
  }
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // inclusive scan
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
  for (int b = b0; b < b1; ++b) {
    a.wide_off[b] = run;
    const GNState &st = a.states[a.blocks[b].prob];
    const bool on = a.fine_gate_c >= 0.0f ? st.converged != 0 : st.done == 0;
    run += on ? (int)a.need_cnt[b] : 0;
  }
  if (tid == 255) a.wide_off[nb] = part[255];
}

LSLAM_DEV uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
  return v;
}

constexpr int WIDE_ROWS_PER_LANE = 4;  // up to 256 cell rows around a point (cells of >= 0.3 m inside the sqrt(5) m gate)

// One wavefront, one point (wave-uniform arguments): every cell within sqrt(r2) of sel is scanned, 64 rows at a time; the five
// nearest with their exact distances come back in every lane.  bad: the answer needs nanoflann's visit order (a tie), or the
// kernel is not sized for the cells.
LSLAM_DEV void wide_probe(const CellGrid &G, const float (&sel)[3], float r2, int lane, float nf_slack, float (&d)[5], int (&p)[5], bool &num, bool &bad) {
  const float rb = sqrtf(r2) * (1.0f + 1.0e-5f) + GRID_CLIP_MARGIN_MIN;
  const float rbc = rb * G.inv_c;
  const float ux = __fmul_rn(__fsub_rn(sel[0], G.org[0]), G.inv_c);
  const float uy = __fmul_rn(__fsub_rn(sel[1], G.org[1]), G.inv_c);
  const float uz = __fmul_rn(__fsub_rn(sel[2], G.org[2]), G.inv_c);
  num = (ux == ux) && (uy == uy) && (uz == uz);
  // cells [lo, hi] per axis, clamped to the table (cells outside it hold nothing)
  const int xlo = (int)fminf(fmaxf(floorf(ux - rbc), 0.0f), (float)(G.nx - 1)), xhi = (int)fminf(fmaxf(floorf(ux + rbc), 0.0f), (float)(G.nx - 1));
  const int ylo = (int)fminf(fmaxf(floorf(uy - rbc), 0.0f), (float)(G.ny - 1)), yhi = (int)fminf(fmaxf(floorf(uy + rbc), 0.0f), (float)(G.ny - 1));
  const int zlo = (int)fminf(fmaxf(floorf(uz - rbc), 0.0f), (float)(G.nz - 1)), zhi = (int)fminf(fmaxf(floorf(uz + rbc), 0.0f), (float)(G.nz - 1));
  const int ny = yhi - ylo + 1, nz = zhi - zlo + 1;
  const int nrows = num ? ny * nz : 0;
  bool unresolved = nrows > 64 * WIDE_ROWS_PER_LANE;  // (cells smaller than the kernel is sized for)
  uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, k3 = 0xFFFFFFFFu, k4 = 0xFFFFFFFFu, k5 = 0xFFFFFFFFu;
  uint32_t row_s[WIDE_ROWS_PER_LANE], row_id0[WIDE_ROWS_PER_LANE + 1];
  uint32_t id = 0;
#pragma unroll
  for (int k = 0; k < WIDE_ROWS_PER_LANE; ++k) {
    const int r = lane + 64 * k;
    row_s[k] = 0;
    row_id0[k] = id;
    if (r < nrows && !unresolved) {
      const int jy = ylo + r % ny, jz = zlo + r / ny;
      const int base = G.nx * (jy + G.ny * jz);
      const uint32_t s0 = G.cell_start[base + xlo], e0 = G.cell_start[base + xhi + 1];
      row_s[k] = s0;
      // four points in flight per step: the loads do not depend on one another, only the six-key insert is a chain
      for (uint32_t j = s0; j < e0; j += 4) {
        float4 pt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pt[u] = G.pts[min(j + (uint32_t)u, e0 - 1u)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j + (uint32_t)u < e0) {
            const float dist = dist2_xyz(sel[0], sel[1], sel[2], pt[u]);
            const uint32_t key = (__float_as_uint(dist) & ~GRID_ID_MASK) | (id & GRID_ID_MASK);
            k5 = umed3(k4, k5, key); k4 = umed3(k3, k4, key); k3 = umed3(k2, k3, key);
            k2 = umed3(k1, k2, key); k1 = umed3(k0, k1, key); k0 = min(k0, key);
            ++id;
          }
        }
      }
    }
  }
  row_id0[WIDE_ROWS_PER_LANE] = id;
  unresolved = unresolved || __any(id > GRID_ID_MASK + 1u);  // a lane saw more candidates than an id can count
  // the six smallest keys of the wavefront, smallest first: winner = the lowest lane that holds the minimum
  // A lane keeps its SIX smallest keys only.  What it dropped is at least its sixth key away (truncated) -- which `rest`
  // below covers as long as the lane still holds a key after the six rounds, and does not when all six winners were its own:
  // then the bound is the sixth winner's key itself (dropped: that lane saw more than six candidates)
  int pos[6];
  uint32_t dropped = 0xFFFFFFFFu;
  int dry_lane = -1;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const uint32_t m = wave_min_u32(k0);
    const unsigned long long who = __ballot(k0 == m && m != 0xFFFFFFFFu);
    const int w = who ? __builtin_ctzll(who) : 0;
    int mine = -1;
    bool dry = false;
    if (who && lane == w) {
      const uint32_t cid = k0 & GRID_ID_MASK;
#pragma unroll
      for (int k = 0; k < WIDE_ROWS_PER_LANE; ++k)
        if (cid >= row_id0[k] && cid < row_id0[k + 1]) mine = (int)(row_s[k] + (cid - row_id0[k]));
      k0 = k1; k1 = k2; k2 = k3; k3 = k4; k4 = k5; k5 = 0xFFFFFFFFu;
      dry = j == 5 && k0 == 0xFFFFFFFFu && id > 6u;
    }
    pos[j] = who ? __shfl(mine, w, 64) : -1;
    if (j == 5 && __any(dry)) {
      dropped = m;
      dry_lane = w;
    }
  }
  // exact distances, sorted by the search's own insert (every lane computes the same)
  float e6 = FLT_MAX;
#pragma unroll
  for (int j = 0; j < 5; ++j) { d[j] = FLT_MAX; p[j] = -1; }
  float lb = FLT_MAX;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float x = pos[j] >= 0 ? dist2_xyz(sel[0], sel[1], sel[2], G.pts[pos[j]]) : FLT_MAX;
    if (j < 5) knn_insert_sorted(d, p, x, pos[j]);
    else e6 = x;
  }
  // The lane that ran dry: its keys order its candidates only to 2^-13, so among candidates of one such bucket the six it KEPT
  // need not be its six nearest, and what it dropped is bounded only by its sixth key.  Refusing the point on that bound
  // (d[4] against the sixth key's truncated distance) built the trees of the whole map in every fourth frame of the mapping
  // node's loop.  The dry lane looks at its rows once more instead, EXACTLY: its six nearest candidates by the search's own
  // distance replace the six winners (every other lane's candidates are at least the smallest key those lanes still hold
  // away -- `rest` below -- which is not below the sixth winner's key), and nothing it dropped is nearer than the sixth of them.
  if (dry_lane >= 0) {  // wave-uniform, rare
    float ed[6] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX};
    int ep[6] = {-1, -1, -1, -1, -1, -1};
    if (lane == dry_lane) {
#pragma unroll
      for (int k = 0; k < WIDE_ROWS_PER_LANE; ++k) {
        const uint32_t s0 = row_s[k], cnt = row_id0[k + 1] - row_id0[k];
        for (uint32_t j = 0; j < cnt; ++j) {
          float x = dist2_xyz(sel[0], sel[1], sel[2], G.pts[s0 + j]);
          int xp = (int)(s0 + j);
#pragma unroll
          for (int m = 0; m < 6; ++m) {  // sorted insert, earlier candidates first among equals (an exact tie is refused below)
            const bool lt = x < ed[m];
            const float td = ed[m];
            const int tp = ep[m];
            ed[m] = lt ? x : td;
            ep[m] = lt ? xp : tp;
            x = lt ? td : x;
            xp = lt ? tp : xp;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      ed[m] = __shfl(ed[m], dry_lane, 64);
      ep[m] = __shfl(ep[m], dry_lane, 64);
    }
#pragma unroll
    for (int m = 0; m < 5; ++m) { d[m] = ed[m]; p[m] = ep[m]; }
    e6 = ed[5];
    pos[5] = ep[5];
    dropped = 0xFFFFFFFFu;
  }
  const uint32_t rest = min(wave_min_u32(k0), dropped);  // every other candidate's truncated distance is at least this
  lb = fmaxf(e6, d[4]);
  knn_insert_sorted(d, p, e6, pos[5]);
  if (lb < 1.0e30f) lb *= 1.0f - nf_slack;  // (lslam_grid.hpp: the rounding of nanoflann's own pruning bound; 0 unless LSLAM_AB_WIDE_NF_MARGIN)
  if (rest != 0xFFFFFFFFu) lb = fminf(lb, __uint_as_float(rest & ~GRID_ID_MASK));
  // covered: every cell a point within (rb - slack) of the query can lie in has been scanned
  const float cov = rb - GRID_U_SLACK * G.c;
  const float cov2 = (cov * cov) * (1.0f - 1.0e-5f);
  const bool distinct = d[0] < d[1] && d[1] < d[2] && d[2] < d[3] && d[3] < d[4];
  // resolved: the five are proven, or there provably are not five inside the gate (the result is not used then: ScanMatch.cpp:102,120)
  const bool five_proven = distinct && d[4] < lb && d[4] < cov2;
  const bool none_inside = !(d[4] < 5.0f) && cov2 >= 5.0f * (1.0f + 1e-6f) && lb >= 5.0f;
  bad = num && (unresolved || !(five_proven || none_inside));
  if (!num) {
#pragma unroll
    for (int j = 0; j < 5; ++j) { d[j] = FLT_MAX; p[j] = -1; }
  }
}

__global__ __launch_bounds__(256) void sweep_wide_kernel(SweepArgs a) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);  // wave-uniform
  const int nb = a.nb_total;
  if (item >= a.wide_off[nb]) return;
  int lo_b = 0, hi_b = nb;  // the workgroup whose list holds the item: wide_off[b] <= item < wide_off[b + 1]
  while (hi_b - lo_b > 1) {
    const int mid = (lo_b + hi_b) >> 1;
    if (a.wide_off[mid] <= item) lo_b = mid; else hi_b = mid;
  }
  const int b = __builtin_amdgcn_readfirstlane(lo_b);
  const BlockDesc bd = a.blocks[b];
  const int qi = bd.first + (int)a.need_list[(size_t)b * SWEEP_BLOCK + (item - a.wide_off[b])];
  GNState *st = const_cast<GNState *>(a.states) + bd.prob;
  const bool is_surf = bd.is_surf != 0;
  const CellGrid &G = is_surf ? a.ks : a.kc;
  const float4 q = a.q[qi];
  float sel[3];
  sel[0] = ((st->R[0] * q.x + st->R[1] * q.y) + st->R[2] * q.z) + st->t[0];
  sel[1] = ((st->R[3] * q.x + st->R[4] * q.y) + st->R[5] * q.z) + st->t[1];
  sel[2] = ((st->R[6] * q.x + st->R[7] * q.y) + st->R[8] * q.z) + st->t[2];
  // the ball that must be covered: five map points are known to lie within sqrt(r2) (pass 1 saw them), else the gate
  float r2 = 5.0f * (1.0f + 1e-5f);
  if (a.grid_hint) r2 = fminf(r2, a.grid_hint[qi]);
  float d[5];
  int p[5];
  bool num, bad;
  wide_probe(G, sel, r2, lane, a.wide_nf_slack, d, p, num, bad);
  if (bad) {
    if (lane == 0) atomicOr(&st->pad, 1);
  }
  if (lane < 5) {
    float dv = d[0];
    int pv = p[0];
#pragma unroll
    for (int j = 1; j < 5; ++j) {
      dv = lane == j ? d[j] : dv;
      pv = lane == j ? p[j] : pv;
    }
    a.wide_d[(size_t)qi * 5 + lane] = dv;
    a.wide_p[(size_t)qi * 5 + lane] = pv;
  }
}

hipError_t launch_sweep_wide(const SweepArgs &a, hipStream_t s) {  // (after launch_sweep_plan(..., with_prefix): wide_off is there)
  if (a.nb_total <= 0) return hipSuccess;
  // one wavefront per point that CAN be listed (every point of the launch): the ones beyond the listed total leave at once
  hipLaunchKernelGGL(sweep_wide_kernel, dim3((unsigned)a.nb_total * (SWEEP_BLOCK / 4)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The whole Gauss-Newton loop of ONE scan in one persistent launch.  As launches an iteration is a sweep
// (44-55 us for 115 200 points), a solve launch (13 us: 1024 threads ramp up to reduce 450 x 32 partials
// and run a 6 x 6 QR), two launch-to-launch gaps and, after the last iteration, a spare pair that finds
// the loop ended: ~79 us per iteration, 0.32 ms of device time per scanMatchScan of four.  Here every workgroup of the
// sweep stays resident (450 of them at two per CU), keeps ITS OWN copy of the scan's state in LDS and,
// after its block of the sweep, takes part in one grid-wide exchange of the blocks' 32 sums -- slots that
// hold a sentinel until their owner stores them, polled by the readers themselves (the mechanism of
// pg_pcg_persistent_kernel, lslam_posegraph.hip; three generations, a slot is reset one exchange after
// everybody has read it) -- reduces them in the solve kernel's order and runs the solve REPLICATED: every
// workgroup computes the same next pose from the same numbers, nothing travels back.  Bit for bit the
// launch loop's result (same sums in the same order, same solve; tests/test_gpu_parity.py holds the two
// against each other).  For a single resident scan against shallow trees without the stereo term,
// the sharded exchange or per-launch profiling; a spin limit raises an abort flag instead of hanging
// and the caller falls back to the launch loop.  MEASURED NO FASTER than the launch loop (327 us against
// 315 us per four-iteration loop: the exchanges and the replicated solve cost what the solve launch and
// its gaps do) and therefore off unless LSLAM_PERSISTENT_GN=1; kept as the A/B switch of that result.
// ---------------------------------------------------------------------------
constexpr uint32_t GNP_SENT = 0xFFF8DEADu;  // a NaN payload no sum produces
constexpr long long GNP_SENT64 = (long long)(((unsigned long long)GNP_SENT << 32) | GNP_SENT);
constexpr unsigned GNP_SPIN_LIMIT = 1u << 20;
LSLAM_DEV float gnp_ld(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
LSLAM_DEV void gnp_st(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(SWEEP_BLOCK) __attribute__((amdgpu_waves_per_eu(2))) void gn_persistent_kernel(SweepArgs a, int jtj_mode, GnLoopArgs g) {
  constexpr int BLOCK = SWEEP_BLOCK, LDS_DEPTH = KD_STACK_LDS;
  __shared__ float red[BLOCK / 64][NCOL];
  __shared__ uint32_t stack_lds[2 * LDS_DEPTH * BLOCK];
  __shared__ GNState lst;       // this workgroup's copy of the scan's state
  __shared__ float part[NCOL];
  __shared__ double tot[NCOL];
  __shared__ GnShared sh;
  __shared__ int go;
  __shared__ double redd[32][NCOL];
  const int tid = threadIdx.x;
  const int lb = xcd_remap(blockIdx.x, a.nb_total);
  const BlockDesc bd = a.blocks[lb];
  const int nb = a.nb_total;
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.states);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&lst);
    for (int i = tid; i < (int)(sizeof(GNState) / 4); i += BLOCK) dst[i] = src[i];
  }
  __syncthreads();
  for (int it = 0; it < g.max_iterations && !lst.done; ++it) {
    sweep_body<BLOCK, false, false, LDS_DEPTH, false, true>(a, jtj_mode, lb, bd, &lst, stack_lds, red, part, (a.bounded && it > 0) ? 1 : 0);
    __syncthreads();
    // Exchange in two stages, the solve kernel's summation order kept: the blocks' 32 sums go to slots (two generations);
    // workgroup `grp` < 32 adds rows grp, grp + 32, ... of them in fp64 and publishes the group's sums (three generations);
    // every workgroup adds the 32 groups in order.  (One stage -- every workgroup reading all 450 x 32 sums through the
    // coherence point, 26 MB per iteration -- cost more than the launches it replaced.)
    float *slot = g.slots + (size_t)(it & 1) * nb * NCOL;
    double *gsl = g.gslots + (size_t)(it % 3) * 32 * NCOL;
    if (tid < NCOL) gnp_st(slot + (size_t)lb * NCOL + tid, part[tid]);
    if (lb < 32) {
      float *pv = reinterpret_cast<float *>(stack_lds);  // [<= 16][NCOL] this group's rows (the traversal stack is dead by now)
      const int nrow = lb < nb ? (nb - lb + 31) / 32 : 0;
      for (unsigned spins = 0;; ++spins) {
        int bad = 0;
        for (int i = tid; i < nrow * NCOL; i += BLOCK) {
          const float v = gnp_ld(slot + (size_t)(lb + 32 * (i / NCOL)) * NCOL + (i % NCOL));
          pv[i] = v;
          bad |= (__float_as_uint(v) == GNP_SENT) ? 1 : 0;
        }
        if (!__syncthreads_or(bad)) break;
        if ((spins & 255u) == 255u &&
            (spins > GNP_SPIN_LIMIT || __hip_atomic_load(g.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
          __hip_atomic_store(g.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (tid < NCOL) {
        double sgrp = 0.0;
        for (int r = 0; r < nrow; ++r) sgrp += (double)pv[r * NCOL + tid];
        __hip_atomic_store(gsl + lb * NCOL + tid, sgrp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    for (unsigned spins = 0;; ++spins) {
      int bad = 0;
#pragma unroll 4
      for (int i = tid; i < 32 * NCOL; i += BLOCK) {
        const double v = __hip_atomic_load(gsl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (&redd[0][0])[i] = v;
        bad |= (__double_as_longlong(v) == GNP_SENT64) ? 1 : 0;
      }
      if (!__syncthreads_or(bad)) break;
      if ((spins & 255u) == 255u &&
          (spins > GNP_SPIN_LIMIT || __hip_atomic_load(g.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        __hip_atomic_store(g.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    // all 32 group sums are there: every reducer has read every block's sums of this iteration, and every workgroup
    // has finished reading the group sums of the iteration before the last
    if (tid < NCOL) gnp_st(slot + (size_t)lb * NCOL + tid, __uint_as_float(GNP_SENT));
    if (it > 0 && lb < 32 && tid < NCOL)
      __hip_atomic_store(g.gslots + (size_t)((it - 1) % 3) * 32 * NCOL + lb * NCOL + tid, __longlong_as_double(GNP_SENT64),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < NCOL) {
      double v = 0.0;
#pragma unroll 8
      for (int gi = 0; gi < 32; ++gi) v += redd[gi][tid];
      tot[tid] = v;
      lst.sums[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {  // solve_kernel's bookkeeping (ScanMatch.cpp:141-145)
      lst.sweeps += 1;
      const int n_rows = (int)tot[COL_ROWS];
      lst.n_rows = n_rows;
      lst.n_line = (int)tot[COL_LINE];
      lst.n_plane = (int)tot[COL_PLANE];
      lst.score = tot[COL_SCORE];
      go = 1;
      if (n_rows < g.min_rows) {
        go = 0;
        lst.too_few = 1;
        lst.done = 1;
      }
    }
    if (tid < 36) {
      const int r = tid / 6, c = tid % 6;
      const int i = r < c ? r : c, j = r < c ? c : r;
      sh.A[tid] = (float)tot[COL_ATA + (i * 6 - (i * (i - 1)) / 2) + (j - i)];
    }
    if (tid < 6) sh.b[tid] = (float)tot[COL_ATB + tid];
    __syncthreads();
    if (go) {
      gn_step_block(&lst, sh, g.eig_thresh, g.delta_r_abort, g.delta_t_abort, false);
      if (tid == 0) {
        lst.loop_iter += 1;
        if (lst.loop_iter >= g.max_iterations) lst.done = 1;
      }
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    uint32_t *dst = reinterpret_cast<uint32_t *>(g.state_out);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&lst);
    for (int i = tid; i < (int)(sizeof(GNState) / 4); i += BLOCK) dst[i] = src[i];
  }
}

int gn_persistent_capacity(int device) {  // workgroups of gn_persistent_kernel the device holds at once
  hipDeviceProp_t prop;
  int per_cu = 0;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gn_persistent_kernel, SWEEP_BLOCK, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return per_cu * prop.multiProcessorCount;
}

hipError_t launch_gn_persistent(const SweepArgs &a, int jtj_mode, const GnLoopArgs &g, hipStream_t s) {
  // An ordinary launch: the caller has checked that the grid fits the device at once, and the kernel's spin limit turns
  // the case it does not (another process on the device) into a fallback instead of a hang.  hipLaunchCooperativeKernel
  // adds tens of microseconds per launch on this runtime -- more than the launches this kernel saves.
  const bool coop = env_once().gnp_coop;  // A/B switch
  if (coop) {
    SweepArgs aa = a;
    int jm = jtj_mode;
    GnLoopArgs gg = g;
    void *args[] = {(void *)&aa, (void *)&jm, (void *)&gg};
    return hipLaunchCooperativeKernel((const void *)gn_persistent_kernel, dim3((unsigned)a.nb_total), dim3(SWEEP_BLOCK), args, 0, s);
  }
  hipLaunchKernelGGL(gn_persistent_kernel, dim3((unsigned)a.nb_total), dim3(SWEEP_BLOCK), 0, s, a, jtj_mode, g);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Variant B: LaserOdometry::scanMatch (odometry/LaserOdometry.cpp:328-647), one
// iteration per launch, one lane per sharp / flat point.
// ---------------------------------------------------------------------------
// LaserOdometry::transformToEnd (LaserOdometry.cpp:156-168): de-skew every point to the sweep start
// (transformToStart, :135-142), then move it to the sweep end with the inverse of the full transform.
__global__ void odom_to_end_kernel(float4 *pts, int n, const float *pose6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float pose[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) pose[k] = pose6[k];
  float R[9], t[3], scd[6];
  pose_to_Rt_sc(pose, R, t, scd, DevSinCosF());
  float ti[3];  // Eigen Isometry inverse: R^T, (-R^T) t
#pragma unroll
  for (int r = 0; r < 3; ++r) ti[r] = ((-R[r] * t[0]) + (-R[3 + r] * t[1])) + (-R[6 + r] * t[2]);
  const float4 q = pts[i];
  const float s = 10 * (q.w - (int)q.w);
  float ps[6], Rs[9], ts[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) ps[k] = pose[k] * s;
  pose_to_Rt_sc(ps, Rs, ts, scd, DevSinCosF());
  const float a0 = ((Rs[0] * q.x + Rs[1] * q.y) + Rs[2] * q.z) + ts[0];
  const float a1 = ((Rs[3] * q.x + Rs[4] * q.y) + Rs[5] * q.z) + ts[1];
  const float a2 = ((Rs[6] * q.x + Rs[7] * q.y) + Rs[8] * q.z) + ts[2];
  float4 o;
  o.x = ((R[0] * a0 + R[3] * a1) + R[6] * a2) + ti[0];
  o.y = ((R[1] * a0 + R[4] * a1) + R[7] * a2) + ti[1];
  o.z = ((R[2] * a0 + R[5] * a1) + R[8] * a2) + ti[2];
  o.w = q.w;
  pts[i] = o;
}

hipError_t launch_odom_to_end(float4 *pts, int n, const float *d_pose6, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(odom_to_end_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, d_pose6);
  return hipGetLastError();
}

__global__ __launch_bounds__(256, 2) void odom_sweep_kernel(OdomArgs a) {
  const GNState *st = a.state;
  if (st->done) return;
  constexpr int BLOCK = 256, NWAVE = 4;
  __shared__ float red[NWAVE][NCOL];
  __shared__ uint32_t stack_lds[2 * KD_STACK_LDS * BLOCK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lb = blockIdx.x;
  const bool is_flat = lb >= a.nb_sharp;
  const int li = (is_flat ? lb - a.nb_sharp : lb) * BLOCK + tid;
  const int nq = is_flat ? a.n_flat : a.n_sharp;
  const bool active = li < nq;
  const int qi = is_flat ? a.n_sharp + li : li;
  const int iter = st->loop_iter;
  const int nall = a.n_sharp + a.n_flat;
  float row[6] = {0, 0, 0, 0, 0, 0};
  float rb = 0.0f, kept = 0.0f;
  if (active) {
    const float4 q = is_flat ? a.qf[li] : a.q[li];
    // transformToStart (:135-142): s = 10*frac(intensity); t = _transform * s
    const float s = 10 * (q.w - (int)q.w);
    float ps[6], R[9], t[3], scd[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) ps[k] = st->pose[k] * s;
    pose_to_Rt_sc(ps, R, t, scd, DevSinCosF());
    float sel[3];
    sel[0] = ((R[0] * q.x + R[1] * q.y) + R[2] * q.z) + t[0];
    sel[1] = ((R[3] * q.x + R[4] * q.y) + R[5] * q.z) + t[1];
    sel[2] = ((R[6] * q.x + R[7] * q.y) + R[8] * q.z) + t[2];
    const float4 *org = is_flat ? a.os : a.oc;
    const int n_org = is_flat ? a.n_os : a.n_oc;
    int i1 = a.ind[qi], i2 = a.ind[nall + qi], i3 = a.ind[2 * nall + qi];
    if (iter % 5 == 0 && a.mode != 2) {  // :357 / :423
      TreeView T = is_flat ? a.ts : a.tc;
      float d[5];
      int p[5];
      KdStack<BLOCK, false, KD_STACK_LDS> stk;
      stk.lds = (lds_u32 *)(stack_lds + tid);
      stk.ovf = nullptr;
      stk.ovf_stride = 0;
#ifdef LSLAM_TRAVERSAL_STATS
      TravStats ts_unused = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      knn5_search<BLOCK, false, KD_STACK_LDS>(T, sel[0], sel[1], sel[2], d, p, stk, ts_unused);
#else
      knn5_search<BLOCK, false, KD_STACK_LDS>(T, sel[0], sel[1], sel[2], d, p, stk);  // top-1 of the 5 == nearestKSearch(.,1)
#endif
      i1 = -1; i2 = -1; i3 = -1;
      if (a.mode == 1) {  // the ring-window searches run in odom_window_kernel, one wavefront per query
        if (d[0] < 25.0f) i1 = __float_as_int(T.pts[p[0]].w);
        a.ind[qi] = i1;
        a.sel[qi] = make_float4(sel[0], sel[1], sel[2], 0.0f);
      } else if (d[0] < 25.0f) {
        i1 = __float_as_int(T.pts[p[0]].w);
        const int scan = (int)org[i1].w;
        float m2 = 25.0f, m3 = 25.0f;
        for (int j = i1 + 1; j < nq && j < n_org; ++j) {  // quirk Q5: bound is the query count
          const float4 o = org[j];
          if ((double)(int)o.w > scan + 2.5) break;
          const float dd = sq_diff3(o, sel);
          if (!is_flat) {
            if ((int)o.w > scan && dd < m2) { m2 = dd; i2 = j; }
          } else if ((int)o.w <= scan) {
            if (dd < m2) { m2 = dd; i2 = j; }
          } else {
            if (dd < m3) { m3 = dd; i3 = j; }
          }
        }
        for (int j = i1 - 1; j >= 0; --j) {
          const float4 o = org[j];
          if ((double)(int)o.w < scan - 2.5) break;
          const float dd = sq_diff3(o, sel);
          if (!is_flat) {
            if ((int)o.w < scan && dd < m2) { m2 = dd; i2 = j; }
          } else if ((int)o.w >= scan) {
            if (dd < m2) { m2 = dd; i2 = j; }
          } else {
            if (dd < m3) { m3 = dd; i3 = j; }
          }
        }
      }
      if (a.mode != 1) {
        a.ind[qi] = i1;
        a.ind[nall + qi] = i2;
        a.ind[2 * nall + qi] = i3;
      }
    }
    float coeff[4];
    bool ok = false;
    if (!is_flat) {
      if (i2 >= 0) ok = odom_corner_coeff(org[i1], org[i2], sel, iter, coeff);
    } else {
      if (i2 >= 0 && i3 >= 0) ok = odom_surf_coeff(org[i1], org[i2], org[i3], sel, iter, coeff);
    }
    if (ok) {
      float sc[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) sc[k] = st->sc[k];
      jacobian_row(sc, q.x, q.y, q.z, coeff, row, rb);
      rb = (float)(-0.05 * (double)coeff[3]);  // :575
      kept = 1.0f;
    }
  }
  if (a.mode == 1) return;  // correspondence refresh only; the residual pass follows as its own launch
  float v[NCOL];
#pragma unroll
  for (int i = 0; i < NCOL; ++i) v[i] = 0.0f;
  int kk = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) v[kk++] = row[i] * row[j];
#pragma unroll
  for (int i = 0; i < 6; ++i) v[COL_ATB + i] = row[i] * rb;
  v[COL_ROWS] = kept;
  v[COL_LINE] = is_flat ? 0.0f : kept;
  v[COL_PLANE] = is_flat ? kept : 0.0f;
#pragma unroll
  for (int i = 0; i < 31; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NCOL; ++i) red[wave][i] = v[i];
  }
  __syncthreads();
  if (tid < NCOL) {
    float sacc = red[0][tid];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) sacc += red[w][tid];
    a.partials[(size_t)lb * NCOL + tid] = sacc;
  }
}

// The ring-window searches of LaserOdometry::scanMatch (:366-403 sharp, :430-477 flat), one
// WAVEFRONT per query instead of one lane: the reference walks forwards (to the query count, quirk
// Q5) and backwards from the nearest neighbour until the ring id leaves scan +- 2.5, keeping the
// first strictly smaller squared distance per category.  The wavefront takes the window 64 candidates
// at a time; "first minimum in scan order" = minimum distance, lowest position on ties, compared
// strictly with the running minimum -- the same choice.
__global__ __launch_bounds__(64) void odom_window_kernel(OdomArgs a) {
  if (a.state->done) return;
  const int qi = blockIdx.x, lane = threadIdx.x;
  const int nall = a.n_sharp + a.n_flat;
  const bool is_flat = qi >= a.n_sharp;
  const int nq = is_flat ? a.n_flat : a.n_sharp;
  const float4 *org = is_flat ? a.os : a.oc;
  const int n_org = is_flat ? a.n_os : a.n_oc;
  const int i1 = a.ind[qi];
  int i2 = -1, i3 = -1;
  if (i1 >= 0) {
    const float4 sq = a.sel[qi];
    const float sel[3] = {sq.x, sq.y, sq.z};
    odom_window_walk(org, n_org, nq, is_flat, i1, sel, lane, i2, i3);
  }
  if (lane == 0) {
    a.ind[nall + qi] = i2;
    a.ind[2 * nall + qi] = i3;
  }
}

hipError_t launch_odom_window(const OdomArgs &a, hipStream_t s) {
  const int n = a.n_sharp + a.n_flat;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(odom_window_kernel, dim3(n), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_odom_sweep(const OdomArgs &a, hipStream_t s) {
  if (a.nb_total <= 0) return hipSuccess;
  hipLaunchKernelGGL(odom_sweep_kernel, dim3(a.nb_total), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_solve(const SolveArgs &a, hipStream_t s) {
  if (a.n_prob <= 0) return hipSuccess;
  hipLaunchKernelGGL(solve_kernel, dim3(a.n_prob), dim3(SOLVE_THREADS), 0, s, a);
  return hipGetLastError();
}

__global__ __launch_bounds__(128) void gn_step_tap_kernel(GNState *st, const float *AtA,
                                                          const float *Atb, float dr, float dt,
                                                          float eig_thresh) {
  __shared__ GnShared sh;
  const int tid = threadIdx.x;
  if (tid < 36) sh.A[tid] = AtA[tid];
  if (tid < 6) sh.b[tid] = Atb[tid];
  if (tid < 36) sh.matP[tid] = st->matP[tid];
  __syncthreads();
  gn_step_block(st, sh, eig_thresh, dr, dt);
}

hipError_t launch_gn_step_tap(GNState *st, const float *AtA, const float *Atb, float dr, float dt,
                              float eig_thresh, hipStream_t s) {
  hipLaunchKernelGGL(gn_step_tap_kernel, dim3(1), dim3(128), 0, s, st, AtA, Atb, dr, dt,
                     eig_thresh);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// parity tap: nearestKSearch(p, 5, idx, d2) for nq points already in map frame
// ---------------------------------------------------------------------------
template <bool OVF>
__global__ __launch_bounds__(128, 2) void knn5_kernel(TreeView T, const float4 *q, int nq,
                                                      int32_t *idx, float *d2,
                                                      uint32_t *stack_ovf) {
  __shared__ uint32_t stack_lds[2 * KD_STACK_LDS * 128];
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int i = lb * 128 + threadIdx.x;
  if (i >= nq) return;
  const float4 qq = q[i];
  float d[5];
  int p[5];
  KdStack<128, OVF, KD_STACK_LDS> stk;
  stk.lds = (lds_u32 *)(stack_lds + threadIdx.x);
  stk.ovf = OVF ? stack_ovf + ((size_t)blockIdx.x * 128 + threadIdx.x) : nullptr;
  stk.ovf_stride = (size_t)gridDim.x * 128;
#ifdef LSLAM_TRAVERSAL_STATS
  TravStats ts = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  knn5_search<128, OVF, KD_STACK_LDS>(T, qq.x, qq.y, qq.z, d, p, stk, ts);
#else
  knn5_search<128, OVF, KD_STACK_LDS>(T, qq.x, qq.y, qq.z, d, p, stk);
#endif
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    idx[i * 5 + j] = p[j] >= 0 ? __float_as_int(T.pts[p[j]].w) : -1;
    d2[i * 5 + j] = d[j];
  }
}

// the same tap through the packet search: 64 consecutive queries per wavefront (their order is the
// caller's: scattered queries make slow packets, the answer does not depend on it); tie lanes are redone
// with knn5_search on the HBM stack
__global__ __launch_bounds__(256) void knn5_packet_kernel(TreeView T, const float4 *q, int nq, int32_t *idx, float *d2,
                                                          uint32_t *stack_ovf, int32_t *n_tie) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < nq;
  const float4 qq = on ? q[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  float d[5];
  int p[5];
  bool tie;
#ifdef LSLAM_PACKET_STATS
  PacketStats ps_unused = {0, 0, 0, 0};
  knn5_packet(T, qq.x, qq.y, qq.z, on, FLT_MAX, d, p, tie, ps_unused);
#else
  knn5_packet(T, qq.x, qq.y, qq.z, on, FLT_MAX, d, p, tie);
#endif
  if (__any(tie)) {
    if (tie) {
      if (n_tie) atomicAdd(n_tie, 1);
      KdStack<256, true, 0> stk;
      stk.lds = nullptr;
      stk.ovf = stack_ovf + i;
      stk.ovf_stride = (size_t)gridDim.x * 256;
      knn5_search<256, true, 0>(T, qq.x, qq.y, qq.z, d, p, stk);
    }
  }
  if (!on) return;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    idx[i * 5 + j] = p[j] >= 0 ? __float_as_int(T.pts[p[j]].w) : -1;
    d2[i * 5 + j] = d[j];
  }
}

#ifdef LSLAM_PACKET_STATS
// profiling build only (tools/packet_stats.py): per wavefront {nodes, leaves, inserts, pops, ties, cycles}
__global__ __launch_bounds__(256) void packet_stats_kernel(TreeView T, const float4 *q, int nq, unsigned *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < nq;
  const float4 qq = on ? q[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  float d[5];
  int p[5];
  bool tie;
  PacketStats ps = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  knn5_packet(T, qq.x, qq.y, qq.z, on, 5.0f, d, p, tie, ps);
  const unsigned long long t1 = __builtin_readcyclecounter();
  const int nt = __popcll(__ballot(tie && d[4] < 5.0f));
  if ((threadIdx.x & 63) == 0) {
    unsigned *o = out + (size_t)(i >> 6) * 8;
    o[0] = ps.nodes; o[1] = ps.leaves; o[2] = ps.inserts; o[3] = ps.pops; o[4] = nt; o[5] = (unsigned)(t1 - t0);
    o[6] = (unsigned)__float_as_uint(d[4]); o[7] = 0;
  }
}
hipError_t launch_packet_stats(const TreeView &T, const float4 *q, int nq, unsigned *out, hipStream_t s) {
  hipLaunchKernelGGL(packet_stats_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, T, q, nq, out);
  return hipGetLastError();
}
#endif

hipError_t launch_knn5_packet(const TreeView &T, const float4 *q, int nq, int32_t *idx, float *d2, uint32_t *stack_ovf,
                              int32_t *n_tie, hipStream_t s) {
  if (nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(knn5_packet_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, T, q, nq, idx, d2, stack_ovf, n_tie);
  return hipGetLastError();
}

// the same tap through the grid probe (lslam_grid.hpp); queries it cannot prove are searched in the tree right here (stack
// in HBM) and counted
__global__ __launch_bounds__(256) void knn5_grid_kernel(CellGrid G, TreeView T, const float4 *q, int nq, int32_t *idx, float *d2,
                                                        uint32_t *stack_ovf, int32_t *n_unproven) {
  __shared__ uint32_t rows_lds[18 * 256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool on = i < nq;
  const float4 qq = on ? q[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  float d[5], lb6;
  int p[5];
  const int verdict = knn5_grid<256>(G, on, qq.x, qq.y, qq.z, FLT_MAX, 0.0f, (lds_u32 *)(rows_lds + threadIdx.x), d, p, lb6);
  const bool redo = on && verdict != GRID_PROVEN;
  int orig[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) orig[j] = (!redo && on && p[j] >= 0) ? __float_as_int(G.pts[p[j]].w) : -1;
  if (__any(redo)) {
    if (redo) {
      if (n_unproven) atomicAdd(n_unproven, 1);
      KdStack<256, true, 0> stk;
      stk.lds = nullptr;
      stk.ovf = stack_ovf + i;
      stk.ovf_stride = (size_t)gridDim.x * 256;
      knn5_search<256, true, 0>(T, qq.x, qq.y, qq.z, d, p, stk);
#pragma unroll
      for (int j = 0; j < 5; ++j) orig[j] = p[j] >= 0 ? __float_as_int(T.pts[p[j]].w) : -1;
    }
  }
  if (!on) return;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    idx[i * 5 + j] = orig[j];
    d2[i * 5 + j] = d[j];
  }
}

hipError_t launch_knn5_grid(const CellGrid &G, const TreeView &T, const float4 *q, int nq, int32_t *idx, float *d2,
                            uint32_t *stack_ovf, int32_t *n_unproven, hipStream_t s) {
  if (nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(knn5_grid_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, G, T, q, nq, idx, d2, stack_ovf, n_unproven);
  return hipGetLastError();
}

// parity tap of the wide probe (a map without kd-trees): one wavefront per query, every cell within the acceptance gate
__global__ __launch_bounds__(256) void knn5_wide_kernel(CellGrid G, const float4 *q, int nq, float nf_slack, int32_t *idx, float *d2, uint8_t *undecided) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // wave-uniform
  if (i >= nq) return;
  const float4 qq = q[i];
  const float sel[3] = {qq.x, qq.y, qq.z};
  float d[5];
  int p[5];
  bool num, bad;
  wide_probe(G, sel, 5.0f * (1.0f + 1e-5f), lane, nf_slack, d, p, num, bad);
  if (lane < 5) {
    float dv = d[0];
    int pv = p[0];
#pragma unroll
    for (int j = 1; j < 5; ++j) {
      dv = lane == j ? d[j] : dv;
      pv = lane == j ? p[j] : pv;
    }
    idx[(size_t)i * 5 + lane] = pv >= 0 ? __float_as_int(G.pts[pv].w) : -1;
    d2[(size_t)i * 5 + lane] = dv;
  }
  if (lane == 0) undecided[i] = (bad || !num) ? 1 : 0;
}
hipError_t launch_knn5_wide(const CellGrid &G, const float4 *q, int nq, float nf_slack, int32_t *idx, float *d2, uint8_t *undecided, hipStream_t s) {
  if (nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(knn5_wide_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, G, q, nq, nf_slack, idx, d2, undecided);
  return hipGetLastError();
}

hipError_t launch_knn5(const TreeView &T, const float4 *q, int nq, int32_t *idx, float *d2,
                       uint32_t *stack_ovf, hipStream_t s) {
  if (nq <= 0) return hipSuccess;
  if (stack_ovf)
    hipLaunchKernelGGL(knn5_kernel<true>, dim3((nq + 127) / 128), dim3(128), 0, s, T, q, nq, idx,
                       d2, stack_ovf);
  else
    hipLaunchKernelGGL(knn5_kernel<false>, dim3((nq + 127) / 128), dim3(128), 0, s, T, q, nq, idx,
                       d2, stack_ovf);
  return hipGetLastError();
}

}  // namespace lslam

#ifdef LSLAM_EXP_SECTION_CLOCK
// (experiment build only, not part of the ABI) out[12]: s_memtime ticks per section of sweep_grid_kernel summed over the reporting
// wavefronts (one workgroup in sixteen), out[9] = how many reported; reset != 0 zeroes the counters afterwards
extern "C" int lslam_debug_section_clock(uint64_t *out, int reset) {
  static unsigned long long h[12 * 64];
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(lslam::g_section_clock), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 12; ++i) {
    out[i] = 0;
    for (int k = 0; k < 64; ++k) out[i] += h[i * 64 + k];
  }
  if (reset) {
    for (auto &v : h) v = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(lslam::g_section_clock), h, sizeof(h)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

#ifdef LSLAM_EXP_PASS2_CLASS
extern "C" int lslam_debug_pass2_hist(uint64_t *out8, int reset) {  // (experiment build only, not part of the ABI)
  unsigned long long h[8];
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(lslam::g_pass2_hist), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 8; ++i) out8[i] = h[i];
  if (reset) {
    for (auto &v : h) v = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(lslam::g_pass2_hist), h, sizeof(h)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
