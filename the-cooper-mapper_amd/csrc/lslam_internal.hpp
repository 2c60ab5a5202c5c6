// lslam_internal.hpp -- structures shared by the host API and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "lslam_device.hpp"
#include "lslam_grid.hpp"

// Timing experiments (tools/build_variant.sh, tools/ab_variants.sh) compile the product's own kernels with LSLAM_EXP_* macros,
// several of which give WRONG RESULTS on purpose (a fit skipped, a pass dropped, a load halved).  Such a library identifies
// itself: lslam_abi_version() answers with the NEGATIVE version, so every caller that checks the ABI (the C++ mirrors, the
// ctypes binding) refuses it -- unless the process says LSLAM_ALLOW_EXPERIMENT_BUILD=1, as the A/B scripts do.
#if defined(LSLAM_EXP_COUNT_NOHINT) || defined(LSLAM_EXP_LOAD2) || defined(LSLAM_EXP_LOAD_TWICE) || defined(LSLAM_EXP_NO_ACC) || \
    defined(LSLAM_EXP_NO_FIT) || defined(LSLAM_EXP_NO_PASS2) || defined(LSLAM_EXP_PROCESS_TWICE) || defined(LSLAM_EXP_SETUP_TWICE) || \
    defined(LSLAM_EXP_SECTION_CLOCK) || defined(LSLAM_EXP_SLEEP) || defined(LSLAM_EXP_PASS2_CLASS)
#define LSLAM_EXPERIMENT_BUILD 1
#else
#define LSLAM_EXPERIMENT_BUILD 0
#endif

namespace lslam {

// Columns of one block's partial-sum record (and of the reduced sums).
enum : int {
  COL_ATA = 0,      // 21 upper-triangular entries of A^T A, row-major
  COL_ATB = 21,     // 6 entries of A^T b
  COL_ROWS = 27,    // rows kept (laserCloudSelNum)
  COL_LINE = 28,    // line_match_count   (corner blocks only)
  COL_PLANE = 29,   // plane_match_count  (surf blocks only)
  COL_SCORE = 30,   // sum exp(-|res|) over kept rows (ScanMatch.cpp:42-49)
  COL_STEREO = 31,  // stereo observations used (joint LiDAR + stereo system; zero in LiDAR blocks)
  NCOL = 32
};

// Device-resident Gauss-Newton state: the loop of ScanMatch.cpp:91-261 never
// returns to the host between iterations.
struct GNState {
  float pose[6];   // Twist: rot_x rot_y rot_z pos_x pos_y pos_z
  float R[9];      // rotation of the current pose (row-major)
  float t[3];
  float sc[6];     // srx crx sry cry srz crz (util/Angle.h cached values)
  float matP[36];  // degeneracy projector (ScanMatch.cpp:234)
  float x[6];      // last update
  float delta_r, delta_t;
  int32_t iter;        // solves performed
  int32_t loop_iter;   // loop counter of the reference's for(iterCount...) (== iter except in
                       // variant B, where an iteration with too few rows is skipped, not ended)
  int32_t done;        // loop has ended; later launches exit at once
  int32_t converged;   // ScanMatch.cpp:257-260
  int32_t degenerate;  // ScanMatch.cpp:222-233
  int32_t too_few;     // ScanMatch.cpp:141-145
  int32_t n_line, n_plane, n_rows;
  int32_t sweeps;      // sweeps that did work
  int32_t pad;
  double score;        // of the last sweep
  double sums[NCOL];   // last reduced sums (parity tap)
  uint64_t clk[8];     // solve-kernel phase stamps, 100 MHz wall clock (profiling tap)
};

// One sweep launch covers every scan of the batch.  A block works on one scan and one
// feature type (corner blocks run findLine, surf blocks findPlane: no divergence
// between the two fits inside a block).
struct BlockDesc {
  int32_t prob;      // scan (problem) index
  int32_t first;     // first point of this block in SweepArgs::q
  int32_t count;     // points in this block (<= SWEEP_BLOCK)
  int32_t is_surf;
  int32_t out_base;  // tap outputs: index of this scan's corner (or surf) point 0
  int32_t pad[3];
};

struct ProbBlocks {
  int32_t first_block, n_blocks;
};

// Certificate sweep (sweep_body): pass-2 workgroup g works for pass-1 workgroups [first_block, first_block + n_blocks)
struct GroupDesc {
  int32_t first_block, n_blocks;
  int32_t prob;  // the scan (an index into the launch's states)
  int32_t pad;
};
// work list of pass 2: items group * CERT_GROUP + chunk; two sets of counters that alternate from plan to plan (each zeroes the other's)
struct CertPlan {
  int32_t *work;
  int32_t *count, *count_next;
  int32_t *ticket, *ticket_next;  // the next item to deal (sweep_queue_kernel)
  int32_t *ticket2 = nullptr, *ticket2_next = nullptr;  // ... of a second consumer of the same list (sweep_refill_kernel), or null
};
#ifndef LSLAM_CERT_GROUP
#define LSLAM_CERT_GROUP 256
#endif
constexpr int CERT_GROUP = LSLAM_CERT_GROUP;  // (16 until round 6: the certificate sweep lists a third of a workgroup's points, the grid sweep 1 - 3 %)
// SweepArgs::cert_stats: [0] points left to pass 2, [1] points swept by the workgroups that could leave some, [2] (certificate
// sweep) points left to the tree search; from CERT_STATS_BY_SWEEP on, the grid sweep's [0], [1] once more by feature type
// (corner, surf) and sweep of the loop (the last slot: that sweep and every later one): [type][sweep][listed, swept]
constexpr int GRID_STATS_SWEEPS = 8;
constexpr int CERT_STATS_BY_SWEEP = 8;
constexpr int CERT_STATS_WORDS = CERT_STATS_BY_SWEEP + 2 * 2 * GRID_STATS_SWEEPS;
constexpr float CERT_TRY_M_DEFAULT = 0.05f;    // SweepArgs::cert_try_m (LSLAM_CERT_TRY_M overrides)
constexpr float CERT_TRACK_M_DEFAULT = 1.0f;   // SweepArgs::cert_track_m (LSLAM_CERT_TRACK_M)
constexpr float CERT_RANGE_M = 40.0f;  // lever arm that turns a rotation update into a displacement (sweep_body's try test)

// Variant C (util/FeatureMap.h:465-691): the map as a grid of cubes, one kd-tree per cube.
// cell_tree[toIndex(i,j,k)] = index into `trees`, or -1 for a cube with fewer than 5 points.
struct CubeGridDev {
  float cube_size;
  int32_t origin[3];
  int32_t dims[3];
  const int32_t *cell_tree;
  const TreeView *trees;
};

// The solve fused into the tail of a sweep launch (ScanMatch.cpp:206-260 right behind :97-204): the block that retires a
// scan's LAST partial record runs the cross-block reduction, the 6x6 solve and the pose update for that scan, so a
// Gauss-Newton iteration of a latency-bound (single-scan) launch is ONE launch instead of a sweep, a 13 us solve launch and
// the gap between them.  count == nullptr: no fusion (the solve kernel follows as its own launch).
struct SolveParams {
  int32_t max_iterations;
  int32_t min_rows;          // 50 (ScanMatch.cpp:142); 10 in LaserOdometry.cpp:501
  int32_t too_few_continue;  // variant B: `continue` instead of `break` (LaserOdometry.cpp:501-503)
  int32_t nan_reset;         // variant B: LaserOdometry.cpp:622-634
  float delta_r_abort, delta_t_abort;
  float eig_thresh;          // 100 (ScanMatch.cpp:223); 10 in LaserOdometry.cpp:596
};
struct SweepTail {
  int32_t *count;            // [n_prob] blocks of the scan that have stored their record in this launch (zero between launches)
  const ProbBlocks *probs;   // [n_prob] absolute block range of each scan
  float *partials_abs;       // the records of block 0 (SweepArgs::partials is the chunk's base)
  SolveParams sp;
};

struct SweepArgs {
  TreeView tc, ts;
  CubeGridDev gc, gs;  // used instead of tc/ts by the per-cube kernels
  // cell grids of the whole-map trees (lslam_grid.hpp).  grid != 0: the sweep is the grid sweep -- sweep_grid_kernel proves
  // the five neighbours of most points by a 27-cell probe and lists the others for sweep_queue_kernel's tree search; the
  // neighbour ids carried from sweep to sweep (prev_nb) are then positions in kc.pts / ks.pts
  CellGrid kc, ks;
  int32_t grid;
  float grid_clip_margin;  // [m] padding of the clip radius sqrt(bound) of a bounded grid search
  // grid == 2: the map has no kd-trees (deferred, lslam_map_defer_trees).  The points pass 1 lists are resolved by
  // sweep_wide_kernel -- one wavefront per point scans every cell within the point's bound -- which leaves their five
  // neighbours in wide_d / wide_p (positions in kc.pts / ks.pts); sweep_queue_kernel then only runs their residual chain.  A
  // point whose answer needs nanoflann's visit order (an exact distance tie) raises GNState::pad of its scan: the caller builds
  // the trees and runs the call again through them.
  // the fit cache of the grid sweep (sweep_grid_kernel; null: off): per scan point the five neighbours its last fit was made
  // from (positions in ks.pts, -1: none) and the plane with its verdict, both as five planes of n_fit words
  int32_t *fit_ids;        // [5][n_fit]
  float *fit_val;          // [5][n_fit]  plane[0..3], found (0 / 1)
  int32_t n_fit;
  int32_t fit_from_sweep;  // sweep index (0 = a loop's first) from which cached fits are used; they are stored one sweep earlier
  float wide_nf_slack;     // relative margin of the wide probe's fifth-against-sixth test (lslam_grid.hpp GRID_NF_PRUNE_SLACK_WIDE, or 0)
  float *wide_d;           // [points][5]
  int32_t *wide_p;         // [points][5]
  int32_t *wide_off;       // [nb_total + 1] exclusive prefix of the listed points per pass-1 workgroup (grid_prefix_kernel)
  // the grid sweep's second level: the points pass 1 listed get a second, wider probe (125 cells, clipped to the ball of what
  // the first probe saw) by sweep_queue_kernel<..., 3>; what THAT cannot prove is listed again -- same shape as the first
  // lists, entries = point offset from the group's first point -- and only those few go to the tree search
  uint16_t *need2_list;    // [nb_total][SWEEP_BLOCK]
  uint16_t *need2_cnt;     // [nb_total]
  float *grid_hint;        // [points] grid sweep, pass 1 -> pass 2: an upper bound of the fifth neighbour's squared distance of a
                           // point the probe could not prove (the fifth smallest distance it saw, FLT_MAX if it saw fewer than five)
  const float4 *q;  // scan points of all scans, sensor frame, Morton order within a scan
                    // and type, {x,y,z,bitcast(original index)}
  const BlockDesc *blocks;  // [nb_total]
  int32_t nb_total;
  // the grid sweep of a batch late in its loops: only the workgroups of scans whose loop is still running are launched --
  // active_blocks[i] = index into `blocks` of the i-th of them (compact_active_kernel), n_active of them (null / 0: all nb_total)
  const int32_t *active_blocks;
  int32_t n_active;
  const GNState *states;  // [n_prob]
  float *partials;        // [nb_total][NCOL]
  uint32_t *stack_ovf;    // traversal-stack overflow (null unless a tree is deeper than 33)
  int32_t *prev_nb;       // [points][5] neighbour positions found by the previous sweep
  float4 *prev_q;         // [points] (or null) where the point was in the previous sweep (map frame); with prev_lb, a lower bound of the
                          // squared distance from there of every map point outside its five neighbours, the certificate of sweep_body
  float *prev_lb;         // [points] ... that bound (0: none)
  int32_t prev_valid;     // prev_nb holds positions of the current trees
  int32_t bounded;        // 1: production loop (bounded search), 0: taps (nanoflann's plain search)
  int32_t deep_tree;      // a tree is deeper than KD_STACK_LDS+1: the LDS-only kernels cannot be used
  int32_t packet;         // 1: wave-cooperative packet search (lslam_packet.hpp); needs stack_ovf and the trees' PNodes
  int32_t stack_mode;     // SWEEP_STACK_*: which traversal-stack shape launch_sweep takes (AUTO: by launch size)
  // _fineScore re-sweep (ScanMatch.cpp:272-321): >= 0 -> the acceptance gate is d2[0] < fine_gate_c (corner blocks) /
  // fine_gate_s (surf blocks) instead of d2[4] < 5, and the launch works on the scans whose loop CONVERGED (their `done`
  // flag is set) instead of the ones still running
  float fine_gate_c, fine_gate_s;
  SweepTail tail;
  // the certificate sweep's second pass (sweep_body, sweep_queue_kernel); need_cnt null: no certificates
  uint8_t *need_list;              // [nb_total][SWEEP_BLOCK] lanes of pass-1 workgroup b whose point is left to pass 2
  uint16_t *need_cnt;              // [nb_total] how many
  const GroupDesc *groups;         // [n_groups] runs of <= CERT_GROUP consecutive workgroups of one scan and feature type
  int32_t n_groups;
  int32_t group_block_base;        // GroupDesc::first_block counts from the context's first block, `blocks` from this one
  float cert_try_m;                // a scan tests certificates when its last update moved its points by less than this [m]
  float cert_track_m;              // ... and its searches keep the bound for the next sweep's certificates when by less than this
  unsigned long long *cert_stats;  // (or null) debug tap: [0] points left to pass 2, [1] points of certificate-testing workgroups
  // optional per-point taps (all NULL in the production loop)
  int32_t *idx_out;    // [N][5] original map indices
  float *d2_out;       // [N][5]
  float4 *coeff_out;   // [N]
  uint8_t *flags_out;  // [N]
  uint64_t *dbg;       // optional [nb_total][waves][4] shader-clock stamps (profiling tap)
};

struct SolveArgs {
  GNState *states;            // [n_prob]; block p of the solve grid handles scan p
  const float *partials;
  const ProbBlocks *probs;    // [n_prob] block range of each scan
  int32_t n_prob;
  int32_t reduce_only;  // 1: only reduce partials into state->sums (tap; first half of a sharded iteration)
                        // 2: the same for the scans whose loop converged (the _fineScore re-sweep)
  const double *ext_sums;  // non-null: [n_prob][32] sums already reduced (and summed over ranks); skip the reduction
  const float *partials2;  // non-null (single scan only): block records of the stereo term, added after the LiDAR ones
  int32_t n_blocks2;
  int32_t max_iterations;
  float delta_r_abort, delta_t_abort;
  float eig_thresh;  // 100 (ScanMatch.cpp:223); 10 in LaserOdometry.cpp:596
  int32_t min_rows;          // 50 (ScanMatch.cpp:142); 10 in LaserOdometry.cpp:501
  int32_t too_few_continue;  // variant B: `continue` instead of `break` (LaserOdometry.cpp:501-503)
  int32_t nan_reset;         // variant B: LaserOdometry.cpp:622-634
  double *sums_out;          // reduce_only: also write the [n_prob][32] sums here (the all-reduce buffer)
};

// gn_persistent_kernel: the whole Gauss-Newton loop of one resident scan in one cooperative launch (lslam_kernels.hip)
struct GnLoopArgs {
  float *slots;       // [2][nb_total][NCOL] the blocks' sums; exchange slots, filled with the sentinel 0xFFF8DEAD before the launch
  double *gslots;     // [3][32][NCOL] the 32 row groups' sums (same sentinel in both halves)
  unsigned *bar;      // [1] abort flag
  GNState *state_out; // where workgroup 0 leaves the final state (SweepArgs::states holds the initial one)
  int32_t max_iterations, min_rows;
  float delta_r_abort, delta_t_abort, eig_thresh;
};
int gn_persistent_capacity(int device);
hipError_t launch_gn_persistent(const SweepArgs &a, int jtj_mode, const GnLoopArgs &g, hipStream_t s);

// Stereo reprojection rows of the joint system (lslam_stereo.hip; include/lslam_c.h lslam_stereo_cam).
struct StereoCam {
  float fx, fy, cx, cy, bf;
  float T_cl[12];
  float weight, huber_stereo, huber_mono;
  int32_t gate_outliers;
  float min_depth;
};
struct StereoArgs {
  const float4 *landmarks;  // {X, Y, Z, inv_sigma2}, map frame
  const float4 *obs;        // {uL, v, uR (< 0: monocular), -}
  int32_t n;
  StereoCam cam;
  const GNState *state;
  float *partials;          // [stereo_blocks(n)][NCOL]
};
int stereo_blocks(int n);
hipError_t launch_stereo(const StereoArgs &a, hipStream_t s);

// Variant B (LaserOdometry::scanMatch): one launch = one iteration over sharp + flat points.
struct OdomArgs {
  TreeView tc, ts;              // kd-trees of the last corner / surface clouds
  const float4 *oc, *os;        // the same clouds in scan order {x,y,z,intensity}
  int32_t n_oc, n_os;
  const float4 *q, *qf;         // sharp points, flat points {x,y,z,intensity}
  int32_t n_sharp, n_flat;
  int32_t nb_sharp, nb_total;   // blocks: [0,nb_sharp) sharp
  int32_t *ind;                 // [3][n_sharp+n_flat] cached correspondences (:357-408,:423-483)
  float4 *sel;                  // [n_sharp+n_flat] de-skewed query of the last correspondence refresh
  int32_t mode;                 // 0: search inline (one lane per query); 1: nearest neighbour only, store
                                // it and the de-skewed query; 2: cached correspondences only
  const GNState *state;
  float *partials;
};
hipError_t launch_odom_sweep(const OdomArgs &a, hipStream_t s);
hipError_t launch_odom_window(const OdomArgs &a, hipStream_t s);
hipError_t launch_odom_to_end(float4 *pts, int n, const float *d_pose6, hipStream_t s);

#ifndef LSLAM_SWEEP_BLOCK
#define LSLAM_SWEEP_BLOCK 256
#endif
constexpr int SWEEP_BLOCK = LSLAM_SWEEP_BLOCK;

// launchers (lslam_kernels.hip)
// Traversal-stack shape of a sweep launch (include/lslam_c.h LSLAM_STACK_*) and the instantiation launch_sweep took
// (reported through `variant`, counted per context: lslam_debug_sweep_launches).
enum : int { SWEEP_STACK_AUTO = 0, SWEEP_STACK_DEEP = 1, SWEEP_STACK_SHALLOW = 2 };
enum : int {
  SWEEP_VARIANT_DEEP = 0,       // sweep_kernel<256, false, false, 32>: whole stack in LDS
  SWEEP_VARIANT_DEEP_OVF = 1,   // sweep_kernel<256, true, false, 32>: trees deeper than 33 levels
  SWEEP_VARIANT_SHALLOW = 2,    // sweep_kernel<256, true, false, 12>: the batch (bench) kernel
  SWEEP_VARIANT_CUBES = 3,      // sweep_kernel<256, false, true, 32>
  SWEEP_VARIANT_CUBES_OVF = 4,  // sweep_kernel<256, true, true, 32>
  SWEEP_VARIANT_PACKET = 5,     // sweep_kernel<256, true, false, 4, true>
  SWEEP_VARIANT_PERSISTENT = 6, // gn_persistent_kernel
  SWEEP_VARIANT_DEEP_FUSED = 7, // sweep_kernel<256, *, false, 32> with the solve in its tail (single scans)
  SWEEP_VARIANT_GRID = 8,       // sweep_grid_kernel<256> (+ sweep_queue_kernel for the points it could not prove)
  SWEEP_VARIANT_GRID_WIDE = 9,  // sweep_grid_kernel<256, true>: a map without trees, a small launch -- unproven points resolved in place
  SWEEP_N_VARIANTS = 10
};
hipError_t launch_sweep_grid(const SweepArgs &a, int jtj_mode, hipStream_t s, hipEvent_t start, hipEvent_t stop, bool resolve_in_place = false);
hipError_t launch_sweep_wide(const SweepArgs &a, hipStream_t s);  // sweep_wide_kernel (its prefix: launch_sweep_plan with_prefix)
hipError_t launch_knn5_wide(const CellGrid &G, const float4 *q, int nq, float nf_slack, int32_t *idx, float *d2, uint8_t *undecided, hipStream_t s);
hipError_t launch_knn5_grid(const CellGrid &G, const TreeView &T, const float4 *q, int nq, int32_t *idx, float *d2,
                            uint32_t *stack_ovf, int32_t *n_unproven, hipStream_t s);

// Host side of a cell grid (lslam_grid.hip): owns the device arrays of one CellGrid.
struct GridDev {
  CellGrid view{};
  size_t n_cells = 0;
  float4 *pts = nullptr, *tmp_pts = nullptr;
  int32_t *err = nullptr;
  uint32_t *cell_start = nullptr, *count = nullptr, *coarse = nullptr;
  size_t cap_pts = 0, cap_tmp = 0, cap_cell = 0, cap_count = 0, cap_coarse = 0, cap_err = 0;
  bool count_clean = false;  // every cell of `count` is zero (each build leaves it so)
  template <typename T>
  static hipError_t reserve(T *&p, size_t &cap, size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  // status: 0 built (or nothing to build: view.cell_start stays null), 1 non-finite point, 2 map too large for the grid
  hipError_t build(const TreeView &T, float cell, hipStream_t s, int *status);
  // ... from n points {x, y, z, bitcast(original index)} in any order and their bounding box (a map whose trees are deferred)
  hipError_t build(const float4 *src, int n, const float lo[3], const float hi[3], float cell, hipStream_t s, int *status, bool wait = true);
  void release();
};
hipError_t grid_bbox2(const float4 *const pts[2], const int n[2], uint32_t *d_box12, float lo[2][3], float hi[2][3], hipStream_t s);
hipError_t grid_unsort(const CellGrid &G, float4 *out, hipStream_t s);
hipError_t launch_compact_active(const GNState *states, const ProbBlocks *probs, int n_prob, int32_t block_base, int32_t *active_blocks,
                                 int32_t *count_out, hipStream_t s);
hipError_t launch_sweep_plan(const SweepArgs &a, hipStream_t s, const CertPlan &plan, int level, bool with_prefix);
hipError_t launch_sweep_queue(const SweepArgs &a, int jtj_mode, hipStream_t s, hipEvent_t stop, int variant, const CertPlan &plan, int level = 0, bool planned = false);
hipError_t launch_sweep_refill(const SweepArgs &a, int jtj_mode, hipStream_t s, hipEvent_t stop, const CertPlan &plan);
hipError_t launch_sweep(const SweepArgs &a, int jtj_mode, hipStream_t s,
                        hipEvent_t start = nullptr, hipEvent_t stop = nullptr, int *variant = nullptr, bool *cert_launched = nullptr);
hipError_t launch_solve(const SolveArgs &a, hipStream_t s);
hipError_t launch_knn5(const TreeView &T, const float4 *q, int nq, int32_t *idx, float *d2,
                       uint32_t *stack_ovf, hipStream_t s);
hipError_t launch_knn5_packet(const TreeView &T, const float4 *q, int nq, int32_t *idx, float *d2, uint32_t *stack_ovf,
                              int32_t *n_tie, hipStream_t s);
#ifdef LSLAM_PACKET_STATS
hipError_t launch_packet_stats(const TreeView &T, const float4 *q, int nq, unsigned *out, hipStream_t s);
#endif
// words of overflow stack needed for n_threads lanes
// (level e of the stack lives in rows 2(e - LDS_DEPTH), 2(e - LDS_DEPTH) + 1; a traversal of a tree of depth d stacks at
// most d entries; without a depth the size covers the deepest tree the library accepts and any LDS depth)
inline size_t stack_ovf_words(size_t n_threads, int tree_depth = KD_STACK_MAX) {
  const int levels = tree_depth + 2 < KD_STACK_MAX ? tree_depth + 2 : KD_STACK_MAX;
  return 2 * (size_t)levels * n_threads;
}
hipError_t launch_gn_step_tap(GNState *st, const float *AtA, const float *Atb, float dr, float dt,
                              float eig_thresh, hipStream_t s);

// device kd-tree builder (lslam_treebuild.hip): same tree, built in HBM
// d_own_box (node_cap x 6 floats, or null): the tight box of every inner node's points, from which build_packet_nodes makes
// the packet search's nodes when that search is asked for
hipError_t build_kdtree_device(float4 *d_pts, int32_t n, KdNode *d_nodes, float *d_own_box, int32_t node_cap,
                               hipStream_t stream, TreeView *view, int *depth, size_t *n_leaves,
                               int *fallback);

hipError_t build_packet_nodes(const TreeView &view, const float *d_own_box, PNode *d_pn, hipStream_t stream);
hipError_t build_kdforest_device(float4 *d_pts, int32_t n_total, const int32_t *roots_lr, int T, KdNode *d_nodes,
                                 PNode *d_pn, int32_t node_cap, hipStream_t stream, TreeView *views, int *max_depth,
                                 size_t *n_leaves, int *fallback);
void treebuild_release_scratch(hipStream_t s);  // frees the per-stream build scratch

// lslam_api.hip internals used by lslam_fmap.hip (map maintenance)
}  // namespace lslam
struct lslam_ctx;
struct lslam_comm;
// A sweep's four feature lists in HBM (include/lslam_c.h lslam_fset_*): sixteen header slots (what the extraction kernels write:
// [0..3] the lists' sizes, [4] error), then four slices of `cap` points {x, y, z, intensity} -- sharp, less-sharp, flat, less-flat.
struct lslam_fset {
  int device = 0;
  float4 *buf = nullptr;
  size_t cap = 0;  // points per slice
  size_t counts[4] = {0, 0, 0, 0};
  float4 *h_stage = nullptr;  // page-locked staging of lslam_fset_upload (grow-only)
  size_t h_stage_cap = 0;
  float4 *list(int k) const { return buf + 16 + (size_t)k * cap; }
};
namespace lslam {
hipError_t fset_reserve(lslam_fset *fs, size_t points_per_list);  // lslam_odom.hip: grows (contents are lost)
// lslam_comm.hip: in-place fp64 SUM over the ranks of `comm`, enqueued on `s`
hipError_t comm_allreduce_f64(lslam_comm *comm, double *buf, size_t count, hipStream_t s);
hipError_t comm_allgatherv_f64(lslam_comm *comm, int n_lists, double *const *bufs, const int64_t *const *offs, hipStream_t s);
int comm_world(const lslam_comm *comm);
int comm_rank(const lslam_comm *comm);
// box_lo / box_hi (optional): the two clouds' bounding boxes, if the caller has them (the deferred-tree map set then needs no
// pass over the points and no wait of its own)
int map_set_device(lslam_ctx *ctx, const float4 *d_corner, size_t n_corner, const float4 *d_surf, size_t n_surf,
                   const float (*box_lo)[3] = nullptr, const float (*box_hi)[3] = nullptr);
int cubemap_set_device(lslam_ctx *ctx, const float4 *d_corner, size_t nc, const std::vector<int32_t> &roots_c,
                       const std::vector<int32_t> &cells_c, const float4 *d_surf, size_t ns,
                       const std::vector<int32_t> &roots_s, const std::vector<int32_t> &cells_s, float cube_size,
                       const int32_t origin[3], const int32_t dims[3]);
// variant-C map whose trees are owned by the caller (lslam_fmap's per-cube forest): views / cell tables are uploaded,
// the trees themselves are referenced, not copied
int cubemap_set_views(lslam_ctx *ctx, const std::vector<TreeView> &views_c, const std::vector<int32_t> &cells_c, size_t nc,
                      int depth_c, const std::vector<TreeView> &views_s, const std::vector<int32_t> &cells_s, size_t ns, int depth_s,
                      float cube_size, const int32_t origin[3], const int32_t dims[3]);
void cubemap_drop_views(lslam_ctx *ctx);  // the owner of such trees goes away
void set_error(const char *msg);
// lslam_odom.hip
void odom_ctx_gone(lslam_ctx *ctx);

// small accessors for translation units that work on a context (lslam_icp.hip)
hipStream_t ctx_stream(lslam_ctx *ctx);
hipStream_t ctx_stream2(lslam_ctx *ctx);  // made on first use; nullptr on failure
TreeView ctx_tree_view(lslam_ctx *ctx, int which);   // 0 corner, 1 surf tree of the resident map
void ctx_invalidate_map(lslam_ctx *ctx);              // the resident trees belong to the caller's own call from here on
int ctx_scratch(lslam_ctx *ctx, size_t n_float4, size_t n_double, float4 **pts, double **dbl);  // grow-only scratch
int ctx_stack_ovf_if_deep(lslam_ctx *ctx, size_t n_threads, uint32_t **out);  // null unless a tree is deeper than the LDS stack
int ctx_device(const lslam_ctx *ctx);
bool ctx_alive(const lslam_ctx *ctx);

// The process environment, read ONCE: when the first context (or pose graph) is created.  No entry point reads the
// environment while it runs -- another thread's setenv would race with it.  A/B switches of measurements and the values tests
// steer failure paths with live here; the per-call test hooks (LSLAM_DEBUG_*: node-slot divisor, spin limits, forced aborts,
// LSLAM_HUGE_MIN) go through debug_env, which looks at the environment only in a process started with LSLAM_DEBUG_HOOKS=1.
struct EnvOnce {
  bool hooks = false;            // LSLAM_DEBUG_HOOKS=1
  bool debug = false;            // LSLAM_DEBUG
  bool unbounded_knn = false;    // LSLAM_UNBOUNDED_KNN
  bool no_morton = false, host_morton = false;  // LSLAM_NO_MORTON, LSLAM_HOST_MORTON
  bool odom_inline = false;      // LSLAM_ODOM_INLINE_SEARCH
  bool odom_trees = false;       // LSLAM_ODOM_TREES=1: A/B switch -- lslam_odometry_match through kd-trees, one launch per step (rounds 1-5)
  bool gnp_coop = false;         // LSLAM_GNP_COOPERATIVE
  bool tiny_phase_off = false;   // LSLAM_TINY_PHASE=0
  bool no_reg_nodes = false;     // LSLAM_NO_REG_NODES
  bool no_level_build = false;   // LSLAM_NO_LEVEL_BUILD
  bool fmap_timing = false;      // LSLAM_FMAP_TIMING
  bool fmap_one_stream = false;  // LSLAM_FMAP_ONE_STREAM: addFeatureCloud's two feature types one after the other on the context's stream (A/B)
  int fx_helpers = 3;                  // LSLAM_FX_HELPERS=0..7: helper workgroups per ring for pointClassify (fx_ring_kernel); 0 = A/B: none
  bool grid_one_stream = false;        // LSLAM_GRID_ONE_STREAM=1: A/B switch -- a deferred map's two cell grids one after the other on the context's stream
  bool small_sort = false;             // LSLAM_SMALL_SORT=1: A/B switch -- a frame's sorts by lslam_sort.hip instead of rocprim::radix_sort_pairs (measured slower: see there)
  bool fmap_measured_extents = false;  // LSLAM_FMAP_MEASURED_EXTENTS: A/B switch -- a map rebuild reads every point for its cube's extremes (fm_minmax_kernel) instead of taking the cube's nominal box (fm_base_kernel)
};
const EnvOnce &env_once();
const char *debug_env(const char *name);  // nullptr unless the process runs with LSLAM_DEBUG_HOOKS=1 and `name` is set

// lslam_fmap.hip: pcl::VoxelGrid per segment (see there)
int voxel_filter_segments(hipStream_t s, const float4 *in_pts, const int32_t *in_seg, size_t n, int nseg, float leaf,
                          float4 *out_pts, int32_t *out_seg, size_t *n_out, bool filter = true, uint32_t *done = nullptr);

// lslam_sort.hip: (64-bit key, 32-bit value) pairs ascending by key, equal keys by value -- the values must be distinct among
// equal keys (every caller passes input positions: a stable sort by key).  For n <= SMALL_SORT_MAX; tmp: small_sort_tmp_bytes(n)
// bytes of device memory (none for a single tile), in / out must not overlap
constexpr int SMALL_SORT_TILE = 4096, SMALL_SORT_MAX_TILES = 32;
constexpr size_t SMALL_SORT_MAX = (size_t)SMALL_SORT_MAX_TILES * SMALL_SORT_TILE;
size_t small_sort_tmp_bytes(size_t n);
hipError_t small_sort_pairs(hipStream_t s, const uint64_t *k_in, uint64_t *k_out, const uint32_t *v_in, uint32_t *v_out, size_t n,
                            void *tmp);

// lslam_scanprep.hip: Morton ordering of the resident scans on the device
struct ScanPrep;
ScanPrep *scanprep_create();
void scanprep_destroy(ScanPrep *sp);
hipError_t scanprep_order(ScanPrep *sp, hipStream_t s, const float4 *h_pts, size_t n, const int32_t *h_seg_off,
                          int nseg, float4 *d_out);  // false once lslam_ctx_destroy ran
}  // namespace lslam
