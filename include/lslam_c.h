/*
 * lslam_c.h -- C ABI of the MI355X-native nonlinear-least-squares backend for L_SLAM.
 *
 * This is the drop-in boundary for the reference's scan-match hot path.  The
 * reference has no FFI seam; its seam is the C++ class lidar_slam::ScanMatch
 * (scan_to_scan_match/ScanMatch.h:21-61), called from
 *   odometry/LaserMatcher.cpp:327-331      (LaserMatcher::optimizeTransform)
 *   pose_graph/graph.cpp:185-190           (Graph::getFinalFeatureMap)
 *   pose_graph/loop_detector.hpp:206-223   (LoopDetector::matching_nearest)
 * Every entry point below names the reference interface it replaces.  All paths
 * are relative to /root/reference/L_SLAM/src/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types cross the ABI.
 *   - Clouds are caller-owned host arrays of points `stride_bytes` apart, x,y,z
 *     as three consecutive floats at offset 0 (pcl::PointXYZI: stride 32;
 *     packed float4: 16; packed xyz: 12).  The library copies what it needs to
 *     HBM; the caller's buffers are only read during the call.
 *   - pose is {rot_x, rot_y, rot_z, pos_x, pos_y, pos_z} = the reference's
 *     Twist (util/Twist.h:13-36); p_map = Rz(rz)*Ry(ry)*Rx(rx)*p + pos
 *     (util/transform_utils.h:288-299).
 *   - Functions return lslam_status and never throw.  One call in flight per
 *     ctx (the reference's ScanMatch is not re-entrant either,
 *     ScanMatch.h:63-85); different ctx may be used from different threads/GPUs.
 *   - The library owns all device memory and its HIP stream.
 *   - There is NO CPU fallback: without a usable HIP device every compute entry
 *     point returns LSLAM_ERR_HIP.
 */
#ifndef LSLAM_C_H
#define LSLAM_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lslam_ctx lslam_ctx;

typedef enum {
  /* outcomes of a scan match (ScanMatch::scanMatchScan returns bool; the
   * reasons for `false` are distinguished here) */
  LSLAM_OK = 0,               /* returned true (ScanMatch.cpp:340) */
  LSLAM_TOO_FEW_REF = 1,      /* ScanMatch.cpp:57-61, pose untouched */
  LSLAM_NOT_CONVERGED = 2,    /* ScanMatch.cpp:342-346 (also when use_score is off) */
  LSLAM_LOW_SCORE = 3,        /* ScanMatch.cpp:323-328 */
  LSLAM_LOW_PERCENT = 4,      /* ScanMatch.cpp:330-335 */
  LSLAM_TOO_FEW_MATCHES = 5,  /* ScanMatch.cpp:141-145 break, then :342-346 */
  /* errors */
  LSLAM_ERR_INVALID = -1,     /* bad argument */
  LSLAM_ERR_HIP = -2,         /* HIP runtime error / no device (see lslam_last_error) */
  LSLAM_ERR_NO_MAP = -3,      /* scan match requested before lslam_map_set */
  LSLAM_ERR_NO_SCAN = -4,     /* lslam_scanmatch_run before lslam_scan_set */
  LSLAM_ERR_TREE_DEPTH = -5,  /* kd-tree deeper than the device traversal stack */
  LSLAM_ERR_TREE_BUILD = -6,  /* the device kd-tree builder hit a structure limit (lslam_last_error says which);
                                 nothing is rebuilt on the host -- the call fails */
  LSLAM_ERR_COMM = -7         /* RCCL communicator error (lslam_comm_*) */
} lslam_status;

/* Mirrors ScanMatch's constructor defaults and setters (ScanMatch.cpp:21-33,
 * ScanMatch.h:21-34). */
typedef struct {
  int32_t max_iterations;            /* ScanMatch(maxIterations = 10) */
  float delta_t_abort;               /* setConvergeThreshold, default 0.05 */
  float delta_r_abort;               /* setConvergeThreshold, default 0.05 */
  int32_t use_score;                 /* setUseCore, default 1 */
  int32_t fine_score;                /* setFineScore, default 0: after a converged loop, one more sweep at the final pose gated
                                        on d2[0] < 0.02 (corner) / 0.05 (surf) -> stats.score2 / percent2 (ScanMatch.cpp:272-321;
                                        printed by the reference, never part of the return value) */
  double score_threshold;            /* setScoreThreshold, default 800 */
  double match_percentage_threshold; /* setPercentThreshold, default 0.4 */
  /* backend knobs (no reference counterpart) */
  int32_t jtj_mode;  /* 0: VALU + wave-shuffle reduction of J^T J; 1: MFMA f32 16x16x4 */
  int32_t profile;   /* 1: bracket every sweep launch with HIP events (stats.gpu_ms_sweep) */
  int32_t scans_in_flight; /* lslam_scanmatch_run_batch: resident scans matched together by one sequence of
                              launches; more scans are taken in chunks of this size (0: up to 128; a scan
                              in flight costs ~1.3 MB of scratch in HBM for a 115 200-point scan) */
  int32_t search_mode;     /* how the 5-NN search maps to the GPU: LSLAM_SEARCH_AUTO / _LANE / _PACKET / _GRID, ORed with LSLAM_STACK_* */
  int32_t knn_cert;        /* certificate sweep (neighbour lists carried over, without a search, where a point provably kept its
                              five neighbours): 1 (default) where it pays -- throughput-bound batches; 0 never: every 5-NN search
                              of every sweep is executed; 2 whatever the size of the launch (tests) */
  float cert_try_m;        /* a scan tests certificates when its last update moved its points by less than this [m] (0.05) */
  float cert_track_m;      /* ... and its searches keep the bound a certificate needs when by less than this [m] (1.0) */
  float grid_cell;         /* LSLAM_SEARCH_GRID: edge of a grid cell [m]; 0 = the default (0.6) */
  int32_t debug_stats;     /* 1: count the points the certificate / grid sweeps leave to their second pass (lslam_debug_cert_stats) */
  int32_t ab_switches;     /* LSLAM_AB_*: measured-and-kept alternatives, off by default (DESIGN.md has the numbers) */
} lslam_opts;
/* Environment overrides of the fields above -- LSLAM_KNN_CERT, LSLAM_CERT_TRY_M, LSLAM_CERT_TRACK_M, LSLAM_GRID_CELL,
 * LSLAM_SEARCH=lane|packet|grid, LSLAM_FORCE_STACK=deep|shallow|auto, LSLAM_PERSISTENT_GN=1, LSLAM_FUSED_SOLVE=1,
 * LSLAM_DEBUG_CERT_STATS=1 -- are read ONCE per context, in lslam_ctx_create; the A/B switches of the builders and the loop
 * (LSLAM_UNBOUNDED_KNN, LSLAM_NO_MORTON, LSLAM_HOST_MORTON, LSLAM_TINY_PHASE, LSLAM_NO_REG_NODES, LSLAM_NO_LEVEL_BUILD, ...)
 * once per process, when the first context is created; a pose graph's (LSLAM_PG_*) when it is created.  No entry point reads
 * the environment while it runs.  The hooks tests use to force failure paths (LSLAM_DEBUG_NODE_CAP_DIV,
 * LSLAM_DEBUG_SPIN_LIMIT, LSLAM_DEBUG_PG_ABORT, LSLAM_DEBUG_GJ_ABORT, LSLAM_HUGE_MIN) are the exception, and exist only in a
 * process started with LSLAM_DEBUG_HOOKS=1 (tests/conftest.py sets it); without it they are never looked at. */
enum { LSLAM_AB_PERSISTENT_GN = 1, /* one resident scan: the whole Gauss-Newton loop as one persistent launch */
       LSLAM_AB_FUSED_SOLVE = 2,   /* latency-bound launches: the 6x6 solve in the tail of the sweep launch */
       LSLAM_AB_SECOND_PROBE = 4,  /* grid sweep: the points the 27-cell probe cannot prove get a second, 125-cell probe (clipped to
                                      the ball of what the first saw) before the tree search -- exact, measured slower
                                      (1.16e10 against 1.25e10 point-residuals/s: the points that need more than the first probe
                                      are the ones a wide probe is slow for too) */
       LSLAM_AB_WIDE_IN_PLACE = 8, /* a map without kd-trees (lslam_map_defer_trees), a launch of at most two wavefronts per SIMD:
                                      the points the probe cannot prove are resolved inside the probe's own launch, wavefront by
                                      wavefront (one launch per sweep instead of five; the sums are then formed exactly as the
                                      lane search's sweep forms them).  Exact, measured slower: a frame's scan match 0.84 ms
                                      against 0.49 -- a wavefront works its unproven points off one after the other, the
                                      separate wide-probe launch gives each its own wavefront */
       LSLAM_AB_FIT_CACHE = 32,    /* grid sweep of a batch: from a loop's third sweep on a surf point whose five neighbours are the
                                      ones of the sweep before reuses that sweep's plane (findPlane is a pure function of the five
                                      in order); the points whose five changed are compacted through LDS so that the fit runs on as
                                      few wavefronts as they fill.  Same bits.  DESIGN 4 has the measurement */
       LSLAM_AB_NO_COMPACT = 64,   /* a batch through the grid sweep: launch every workgroup of every scan in every sweep (round 4's
                                      form) instead of only those of the scans whose loop is still running */
       LSLAM_AB_REFILL = 128,      /* grid sweep of a batch: the second pass as two launches -- the listed points searched by persistent
                                      lanes (a lane whose tree walk has ended hands its five in and takes the next point of a
                                      pool of four workgroups' worth), their residual chain in the next.  Same bits.  Measured
                                      slower: 1.405e10 against 1.461e10, no faster even in a loop's first sweep, where work is
                                      plentiful -- the lanes a wavefront loses are lost inside every round of the walk (descents
                                      and stack pops of different lengths), not at its end */
       LSLAM_AB_WIDE_NF_MARGIN = 16 /* a map without kd-trees: a point whose fifth and sixth distances are within 8 ulps of each
                                      other counts as undecidable too (as an exact tie does): the trees are built and the call
                                      repeated.  Off: such a pair is ordered by its exact fp32 distances -- nanoflann's order
                                      unless its own pruning bound (nanoflann.hpp:1485, two roundings per far step) rounds
                                      across the pair: not observed in 37 139 near-tie queries built to provoke it against the
                                      reference's nanoflann (tools/nanoflann_exactness.py).  On: ~1-2 % of a mapping node's
                                      frames build the trees after all (+2.2 ms each).  Where trees exist (batches, the headline)
                                      the margin is always on and wider (100 ulps): a refused point just takes the tree walk */ };

/* 5-NN search implementations (same answer, bit for bit):
 *   LANE    one query per lane, nanoflann's traversal with an explicit per-lane stack
 *   PACKET  one wavefront walks the tree once for its 64 (Morton-neighbouring) queries: nodes and leaves
 *           arrive by scalar loads, lanes test them against their own query (csrc/lslam_packet.hpp)
 *   GRID    no tree walk for most queries: a dense cell grid over the map (cells of lslam_opts.grid_cell metres), a probe of
 *           the 27 cells around the query, and a PROOF that the five smallest distances found are the five nearest map
 *           points with no tie among the six smallest (csrc/lslam_grid.hpp); a query the proof fails for -- a sparse
 *           neighbourhood, an exact tie -- is searched by LANE.  Whole-map trees only
 *   AUTO    the faster one on MI355X: LANE (the packet search trades the divergent gathers for about twice
 *           the vector ALU work and measured slower; DESIGN.md has the numbers) */
enum { LSLAM_SEARCH_AUTO = 0, LSLAM_SEARCH_LANE = 1, LSLAM_SEARCH_PACKET = 2, LSLAM_SEARCH_GRID = 3 };
/* Traversal-stack shape of the LANE search, ORed into a search mode (same answer, bit for bit -- the shapes hold the
 * same entries; only where they live differs):
 *   DEEP     all 32 levels of a lane's stack in LDS (two workgroups per CU): what a latency-bound single-scan launch takes
 *   SHALLOW  12 levels in LDS, deeper ones in an HBM overflow area (five wavefronts per SIMD): what a launch of more than
 *            2 048 wavefronts -- a batch, the bench -- takes
 *   AUTO     by the size of the launch
 * LSLAM_FORCE_STACK=deep|shallow|auto in the environment (read by lslam_ctx_create) overrides the bits.  The parity tests run every oracle
 * comparison through both shapes; lslam_debug_sweep_launches says which kernel really ran. */
enum { LSLAM_STACK_AUTO = 0, LSLAM_STACK_DEEP = 0x100, LSLAM_STACK_SHALLOW = 0x200 };
/* lslam_sweep_ex only, ORed into LSLAM_SEARCH_GRID: run the tap's sweep the way the production loop runs a LATER sweep of a
 * Gauss-Newton loop -- bounded by the acceptance gate and by what the previous sweep of the loop carried over per point (where
 * the point was, how far its fifth neighbour: the probe's rows and cells are clipped to that bound, the second pass's tree
 * search starts from it).  The carried state is the one the last lslam_scanmatch_run* on this resident scan left (run it
 * with search_mode LSLAM_SEARCH_GRID and max_iterations = k: the tap at the pose it returned is then sweep k + 1 of the
 * loop, kernel for kernel).  Beyond the gate (d2[4] >= 5) nothing is looked up by the reference (ScanMatch.cpp:102,120) and
 * idx_out / d2_out of such a point are not nanoflann's; flags and coefficients are the reference's for every point. */
/* LSLAM_SWEEP_FIRST: the loop's FIRST sweep, kernel for kernel -- bounded by the acceptance gate only, nothing carried. */
enum { LSLAM_SWEEP_CARRIED = 0x400, LSLAM_SWEEP_FIRST = 0x800 };

/* Per-call statistics (the counters the reference prints, ScanMatch.cpp:35-40,
 * 143,269, plus timing taps). */
typedef struct {
  int32_t status;          /* same value the call returned */
  int32_t iterations;      /* GN iterations whose 6x6 solve ran */
  int32_t n_line;          /* line_match_count of the last sweep */
  int32_t n_plane;         /* plane_match_count of the last sweep */
  int32_t n_rows;          /* laserCloudSelNum of the last sweep */
  int32_t degenerate;      /* isDegenerate (ScanMatch.cpp:222-233) */
  int32_t converged;       /* ScanMatch.cpp:257-260 */
  float delta_r, delta_t;  /* of the last solve (deg, cm) */
  double score, percent;   /* ScanMatch.cpp:265-268 (valid when converged && use_score) */
  int64_t point_residuals; /* (Nc+Ns) x sweeps executed on the device */
  int32_t sweeps;          /* sweep launches that did work (not early-exited) */
  int32_t sweep_launches;  /* sweep launches timed (profile=1) */
  float gpu_ms_total;      /* HIP events around the whole device-resident GN loop */
  float gpu_ms_sweep;      /* sum of sweep-kernel durations (profile=1), else 0 */
  double score2, percent2; /* ScanMatch.cpp:317-319 (opts.fine_score && converged && use_score, else 0) */
} lslam_stats;

typedef struct {
  uint64_t n_corner, n_surf;         /* map sizes */
  uint32_t nodes_corner, nodes_surf; /* kd-tree node counts */
  int32_t depth_corner, depth_surf;  /* kd-tree depths */
  float build_ms;                    /* tree build wall time inside lslam_map_set */
  float upload_ms;                   /* H2D copy time inside lslam_map_set */
  int32_t built_on_device;           /* always 1: the HIP builder is the only one */
  int32_t build_attempts;            /* node-slot array sizes tried (1: the first, 8n/3 slots, fitted) */
} lslam_map_info;

/* ---- lifecycle --------------------------------------------------------- */

/* Creates a context on HIP device `device` with its own stream.
 * Replaces: construction of a lidar_slam::ScanMatch member
 * (odometry/LaserMatcher.h _scan_match; pose_graph/loop_detector.hpp:269). */
int lslam_ctx_create(int device, lslam_ctx **out);
void lslam_ctx_destroy(lslam_ctx *ctx);
/* Last error text for this thread ("" if none). */
const char *lslam_last_error(void);
/* Fills opts with the reference defaults (ScanMatch.cpp:21-33) -- EVERY byte of the struct: call it before setting fields. */
void lslam_default_opts(lslam_opts *opts);
/* The structs of this ABI grow from release to release (lslam_opts: 72 -> 80 bytes in round 4).  A caller compiled against
 * another header would hand the library a struct of the wrong size: compare before the first call --
 *   assert(lslam_abi_version() == LSLAM_ABI_VERSION && lslam_sizeof_opts() == sizeof(lslam_opts));
 * (the C++ mirrors do, and refuse to start otherwise). */
/* 5 (round 5): no struct changed; new entry points (lslam_debug_grid_stats, lslam_debug_knn5_wide, lslam_debug_sort_pairs), new bits (LSLAM_SWEEP_FIRST /
 * _CARRIED, LSLAM_AB_FIT_CACHE, LSLAM_AB_WIDE_NF_MARGIN) -- a program built against this header needs a library that has them. */
/* 6 (round 6): no struct changed; new entry points (lslam_fset_*, lslam_extract_features_dev, lslam_odom_*, lslam_map_epoch). */
#define LSLAM_ABI_VERSION 6
int lslam_abi_version(void);
size_t lslam_sizeof_opts(void);
size_t lslam_sizeof_stats(void);

/* ---- map (reference clouds) -------------------------------------------- */

/* Uploads the reference corner/surf clouds and builds both kd-trees.
 * Replaces: kdtreeCorner.setInputCloud / kdtreeSurf.setInputCloud,
 * ScanMatch.cpp:68-76 (util/nanoflann_pcl.h:141-148 -> nanoflann.hpp:1270-1284).
 * The map stays resident and may be reused by any number of scan matches
 * (the reference rebuilds it on every call -- quirk Q4). */
int lslam_map_set(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf,
                  size_t n_surf, size_t stride_bytes);
int lslam_map_info_get(const lslam_ctx *ctx, lslam_map_info *info);
/* Which map is resident: a number that changes whenever ANY entry point replaces or drops the context's map (lslam_map_set,
 * lslam_cubemap_set, lslam_fmap_surround_to_map / _to_cubemap, lslam_odometry_match_trees, lslam_icp_align, a failed tree build
 * ...), 0 while there is none.  A caller that skips an upload because "my clouds are resident" (ScanMatch::setReferenceEpoch,
 * include/lslam_scan_match.hpp) records it after its map set and compares before every reuse. */
uint64_t lslam_map_epoch(const lslam_ctx *ctx);

/* Deferred kd-trees.  on != 0: a map set from now on -- one that is already in HBM (lslam_fmap_surround_to_map: the mapping
 * node's per-frame map, LaserMatcher.cpp:303-331, where the reference rebuilds both trees every frame -- quirk Q4) or host
 * clouds (lslam_map_set: ScanMatch::scanMatchScan's, uploaded as usual) -- gets its cell grids at once (bounding box, cell
 * counts, one scan, placement: about a sixth of a tree build) and its kd-trees only when something needs them.  A scan match of one small scan against such a map runs on the grids alone -- the 27-cell probe with
 * its proof, and for the points it cannot prove one wavefront each over every cell within their bound -- and gives nanoflann's
 * neighbours exactly as the tree search does; the one case the grids cannot decide, an exact distance tie among a point's six
 * nearest, makes the call build the trees and run again through them (with lslam_opts.ab_switches & LSLAM_AB_WIDE_NF_MARGIN a
 * fifth / sixth pair within 8 ulps does too: see there for what that guards against and what it costs).  Every other entry point that touches the trees (the
 * taps, LSLAM_SEARCH_LANE / _PACKET, batches, the sharded loop, lslam_icp_align, ...) builds them first.  Results do not
 * depend on the setting.  lslam_map_info reports depth 0 / 0 nodes while the trees are pending. */
int lslam_map_defer_trees(lslam_ctx *ctx, int32_t on);

/* Variant C -- the map as FeatureMap keeps it (util/FeatureMap.h): a grid of dims[0] x dims[1]
 * x dims[2] cubes of cube_size metres (50), cube of a point = round(p/cube_size) + origin
 * (worldToCube, :475-487); every cube gets its own kd-tree (as _kdtreeCorner/_kdtreeSurf,
 * :71-72,438,453).  After this call the scan-match entry points behave like
 * FeatureMap::scanMatchScan (:490-691): a scan point is searched only in the tree of the cube
 * it falls into, cubes with fewer than 5 points are skipped, and there is no
 * "reference cloud too few" guard.  The caller passes the reference's settings through
 * lslam_opts (max_iterations 10, thresholds 0.05/0.05, use_score 0).  lslam_map_set
 * switches back to the whole-map trees of ScanMatch::scanMatchScan. */
int lslam_cubemap_set(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf,
                      size_t n_surf, size_t stride_bytes, float cube_size, const int32_t origin[3],
                      const int32_t dims[3]);

/* ---- scan (query clouds) ------------------------------------------------ */

/* Uploads the scan's corner/surf feature clouds (CornerCloud / SurfCloud of
 * ScanMatch.cpp:53-54) so that lslam_scanmatch_run works on HBM-resident data. */
int lslam_scan_set(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf,
                   size_t n_surf, size_t stride_bytes);

/* Batch form: n_scans independent scans (e.g. keyframes) made resident together; they
 * are matched against the same resident map by ONE sequence of kernel launches, which
 * is what fills a 256-CU GPU (one 64-ring scan is only ~1800 wavefronts).
 * Replaces the per-keyframe loop of Graph::getFinalFeatureMap (pose_graph/graph.cpp:171-197)
 * and the per-candidate loop of LoopDetector::matching_nearest
 * (pose_graph/loop_detector.hpp:166-226). */
int lslam_scan_set_batch(lslam_ctx *ctx, int32_t n_scans, const void *const *corner,
                         const size_t *n_corner, const void *const *surf, const size_t *n_surf,
                         size_t stride_bytes);

/* ---- Gauss-Newton scan match ------------------------------------------- */

/* The GN loop of ScanMatch::scanMatchScan(..., Twist&), ScanMatch.cpp:78-347,
 * on the resident map and scan.  pose is in/out and is always written back
 * except for LSLAM_TOO_FEW_REF and errors (ScanMatch.cpp:324,331,338,343). */
int lslam_scanmatch_run(lslam_ctx *ctx, float pose[6], const lslam_opts *opts, lslam_stats *stats);

/* The same loop for every resident scan of a batch, each with its own pose, iteration
 * count and convergence: poses[n_scans*6] in/out, stats[n_scans] (may be NULL).
 * gpu_ms_* in every stats entry are those of the whole batch.  Returns LSLAM_OK if
 * every scan returned true, else the first non-OK outcome (errors are negative). */
int lslam_scanmatch_run_batch(lslam_ctx *ctx, int32_t n_scans, float *poses,
                              const lslam_opts *opts, lslam_stats *stats);

/* = lslam_scan_set + lslam_scanmatch_run.  Replaces scanMatchScan against a map
 * that is already resident (the FeatureMap::scanMatchScan usage, util/FeatureMap.h:490-691). */
int lslam_scanmatch_scan(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf,
                         size_t n_surf, size_t stride_bytes, float pose[6],
                         const lslam_opts *opts, lslam_stats *stats);

/* = lslam_map_set + lslam_scan_set + lslam_scanmatch_run: the exact drop-in for
 * bool ScanMatch::scanMatchScan(refCorner, refSurf, Corner, Surf, Twist&),
 * ScanMatch.cpp:51-347, including the per-call tree rebuild. */
int lslam_scanmatch_full(lslam_ctx *ctx, const void *ref_corner, size_t n_ref_corner,
                         const void *ref_surf, size_t n_ref_surf, size_t ref_stride_bytes,
                         const void *corner, size_t n_corner, const void *surf, size_t n_surf,
                         size_t stride_bytes, float pose[6], const lslam_opts *opts,
                         lslam_stats *stats);

/* Variant B -- scan-to-scan odometry: void LaserOdometry::scanMatch()
 * (odometry/LaserOdometry.cpp:328-647): nearest neighbour + ring-window correspondences
 * refreshed every 5th iteration, motion-interpolated de-skew (transformToStart, :135-142),
 * weights from iteration 5 on, b = -0.05 d, eigenvalue threshold 10, NaN reset.
 * Clouds are {x,y,z,intensity} (intensity = ring id + relative time; at byte 16 for
 * stride >= 32 as in pcl::PointXYZI, else at byte 12); last_corner / last_surf
 * (_lastCornerCloud / _lastSurfaceCloud) must be in scan order.  pose is the persistent
 * `_transform` (in/out).  Defaults of the reference: 25 iterations, 0.1 / 0.1 (:24-25).
 * Returns LSLAM_OK when the loop converged (:642-644), LSLAM_NOT_CONVERGED otherwise,
 * LSLAM_TOO_FEW_REF when the guard of :337 fails. */
int lslam_odometry_match(lslam_ctx *ctx, const void *last_corner, size_t n_last_corner,
                         const void *last_surf, size_t n_last_surf, const void *sharp, size_t n_sharp,
                         const void *flat, size_t n_flat, size_t stride_bytes, float pose[6],
                         int32_t max_iterations, float delta_t_abort, float delta_r_abort,
                         lslam_stats *stats);

/* The same match with nearestKSearch answered by kd-trees of the two last clouds (built in the call) and one launch per step:
 * the implementation lslam_odometry_match had before the hashed grids of csrc/lslam_odom.hip.  It is what a sweep with an exact
 * distance tie is redone through (nanoflann's visit order decides there) and the A/B partner of the grid path: same result, bit
 * for bit, on tie-free clouds (tests/test_gpu_odom.py).  LSLAM_ODOM_TREES=1 in the environment routes lslam_odometry_match here. */
int lslam_odometry_match_trees(lslam_ctx *ctx, const void *last_corner, size_t n_last_corner,
                               const void *last_surf, size_t n_last_surf, const void *sharp, size_t n_sharp,
                               const void *flat, size_t n_flat, size_t stride_bytes, float pose[6],
                               int32_t max_iterations, float delta_t_abort, float delta_r_abort,
                               lslam_stats *stats);

/* LaserOdometry::transformToEnd (odometry/LaserOdometry.cpp:156-168): every point of a host
 * cloud ({x,y,z} + intensity = ring + relTime at byte 12 of 16-byte points, byte 16 of PointXYZI)
 * is de-skewed to the sweep start and moved to the sweep end, in place. */
int lslam_transform_to_end(lslam_ctx *ctx, void *cloud, size_t n, size_t stride_bytes, const float pose[6]);

/* Isometry3f <-> Twist conversion used by the Isometry overloads
 * (ScanMatch.cpp:349-360; util/transform_utils.h:308-323,54-60).  T is a
 * row-major 4x4. Host-side helpers, no device work. */
void lslam_isometry_to_pose(const float T[16], float pose[6]);
/* The odometry-prior merge that precedes the scan match in the mapping node,
 * transformAssociate(Lold, Lnew, Wold, Wnew): Wnew = Wold * Lold^-1 * Lnew
 * (util/transform_utils.h:502-507, called from LaserMatcher::transformMerge,
 * odometry/LaserMatcher.cpp:333-340).  Row-major 4x4 matrices, host helper. */
void lslam_transform_associate(const float Lold[16], const float Lnew[16], const float Wold[16],
                               float Wnew[16]);
void lslam_pose_to_isometry(const float pose[6], float T[16]);

/* ---- parity / debug taps ------------------------------------------------ */

/* Exact 5-NN of nq query points (already in the map frame) in the resident
 * corner (which_map=0) or surf (1) tree.  Replaces
 * KdTreeFLANN::nearestKSearch(p, 5, idx, d2), util/nanoflann_pcl.h:150-162.
 * idx_out[nq*5] are indices into the cloud passed to lslam_map_set (int32,
 * bit-exact with nanoflann), d2_out[nq*5] squared distances ascending. */
int lslam_knn5(lslam_ctx *ctx, int which_map, const void *queries, size_t nq,
               size_t stride_bytes, int32_t *idx_out, float *d2_out);

/* One sweep (ScanMatch.cpp:97-204) over the resident scan at a fixed pose.
 * Any output may be NULL.  Point order: corner queries then surf queries.
 *   idx_out[N*5], d2_out[N*5]  kNN of the transformed point
 *   coeff_out[N*4]             (w*dir|w*n, w*d) -- coeffSel
 *   flags_out[N]               bit0 d2[4]<5, bit1 fit found, bit2 row kept
 *   sums_out[30]               21 upper-tri A^T A (row-major) | 6 A^T b |
 *                              n_rows | n_line+n_plane | score */
int lslam_sweep(lslam_ctx *ctx, const float pose[6], int32_t jtj_mode, int32_t *idx_out,
                float *d2_out, float *coeff_out, uint8_t *flags_out, float *sums_out);

/* The two taps above with the search implementation chosen explicitly (LSLAM_SEARCH_*); n_ties (may be
 * NULL) receives the number of queries whose answer needed nanoflann's visit order (exact distance ties). */
int lslam_knn5_ex(lslam_ctx *ctx, int which_map, const void *queries, size_t nq, size_t stride_bytes,
                  int32_t search_mode, int32_t *idx_out, float *d2_out, int32_t *n_ties);
int lslam_sweep_ex(lslam_ctx *ctx, const float pose[6], int32_t jtj_mode, int32_t search_mode, int32_t *idx_out,
                   float *d2_out, float *coeff_out, uint8_t *flags_out, float *sums_out);
/* Parity tap of the search a map WITHOUT kd-trees is matched through (lslam_map_defer_trees): the wide probe -- one wavefront
 * per query over every cell of the resident cell grid within the sqrt(5) m acceptance gate -- for nq queries in the map
 * frame.  Builds no tree.  idx_out[nq*5] / d2_out[nq*5] as lslam_knn5 (-1 / FLT_MAX where the gate holds fewer than five
 * points); undecided_out[nq] = 1 where the grids cannot prove nanoflann's answer (an exact distance tie among the six nearest;
 * with nf_margin != 0 also a fifth / sixth pair within 8 ulps, LSLAM_AB_WIDE_NF_MARGIN): a scan match would build the trees. */
int lslam_debug_knn5_wide(lslam_ctx *ctx, int which_map, const void *queries, size_t nq, size_t stride_bytes, int32_t nf_margin,
                          int32_t *idx_out, float *d2_out, uint8_t *undecided_out);

/* Parity tap of the sort the per-frame map maintenance runs on (csrc/lslam_sort.hip; n <= 131 072): keys_out / values_out =
 * the n (key, value) pairs ascending by key, equal keys in ascending order of their values (distinct among equal keys: with
 * values = input positions a stable sort by key -- pcl::VoxelGrid's index order, the Morton order of the resident scans). */
int lslam_debug_sort_pairs(lslam_ctx *ctx, const uint64_t *keys, const uint32_t *values, size_t n, uint64_t *keys_out,
                           uint32_t *values_out);

/* The names SURVEY.md 8(b) gave these entry points before they were built, kept as exported aliases:
 *   lslam_residuals        = lslam_sweep with the MFMA contraction: coeff_out[N*4], valid_out[N] (the flag bits of
 *                            lslam_sweep), JtJ27_out[27] = the 21 upper-triangular A^T A sums then the 6 A^T b sums
 *   lslam_scanmatch_batch  = lslam_scanmatch_run_batch
 *   lslam_posegraph_optimize (further down) = lslam_pg_create + lslam_pg_optimize + lslam_pg_get_poses + destroy */
int lslam_residuals(lslam_ctx *ctx, const float pose[6], float *coeff_out, uint8_t *valid_out, float *JtJ27_out);
int lslam_scanmatch_batch(lslam_ctx *ctx, int32_t n_problems, float *poses, const lslam_opts *opts, lslam_stats *stats);

/* One solve/update step (ScanMatch.cpp:206-260) run by the device solve kernel
 * on caller-provided normal equations.  matP/degenerate are in/out state. */
int lslam_gn_step(lslam_ctx *ctx, const float AtA[36], const float Atb[6], int32_t iter,
                  float pose[6], float matP[36], int32_t *degenerate, float delta_r_abort,
                  float delta_t_abort, float x_out[6], float *delta_r, float *delta_t,
                  int32_t *converged);

/* ---- map maintenance ----------------------------------------
 * Replaces lidar_slam::FeatureMap<PointXYZI> (util/FeatureMap.h) for the steps either side of
 * the scan match: the cube grid, addFeatureCloud + per-cube pcl::VoxelGrid, the active area and
 * the surround concatenation -- all resident in HBM, so that the map never returns to the host
 * between frames.  Clouds handed in are {x,y,z} at offset 0 with the intensity at byte 12 of
 * 16-byte points or byte 16 of pcl::PointXYZI (32 bytes); clouds handed out are packed
 * {x,y,z,intensity}.  One in-flight call per ctx, like everything else on a ctx. */
typedef struct lslam_fmap lslam_fmap;

/* FeatureMap(cubeWidth, cubeHeight, cubeDepth), FeatureMap.h:55-68 (origin = round((size-1)/2),
 * cube 50 m, valid distance 150 m, leaves 0.2/0.2/0.6). */
int lslam_fmap_create(lslam_ctx *ctx, int32_t cube_width, int32_t cube_height, int32_t cube_depth,
                      lslam_fmap **out);
void lslam_fmap_destroy(lslam_fmap *fm);
int lslam_fmap_setup_filter_size(lslam_fmap *fm, float corner, float surf, float map);     /* :72-76 */
int lslam_fmap_setup_world_origin(lslam_fmap *fm, int32_t ox, int32_t oy, int32_t oz);     /* :78-83 */
int lslam_fmap_setup_world_cube_size(lslam_fmap *fm, float size);                          /* :85-87 */
int lslam_fmap_setup_lidar_valid_distance(lslam_fmap *fm, float dist);                     /* :89-91 */
/* update(sensorPose), FeatureMap.h:232-254: clamp the sensor's cube 3 cubes inside the grid,
 * shift() the cube contents (with the reference's swap-chain behaviour, :353-377), move the
 * origin, recompute the active area (:307-352). */
int lslam_fmap_update(lslam_fmap *fm, const float sensor_xyz[3]);
/* addFeatureCloud(corner, surf, tf), FeatureMap.h:218-230: transform by the row-major 4x4 T, push
 * every point into its cube (worldToCube, :475-487; points outside the grid are dropped), then
 * downsizeValidCloud (:288-306): VoxelGrid every cube of the active area. */
int lslam_fmap_add_feature_cloud(lslam_fmap *fm, const void *corner, size_t n_corner, const void *surf,
                                 size_t n_surf, size_t stride_bytes, const float T[16]);
/* The same without the wait at its end, for a node whose sweep ends with addFeatureCloud (LaserMapping::process,
 * LaserMapping.cpp:349-353): the clouds are copied out of the caller's memory and everything is enqueued; the rebuild is
 * waited for and committed at the head of the NEXT call on this map, whichever it is (or by lslam_fmap_wait), so the node's
 * next sweep is being prepared while the map is rebuilt.  An error of the deferred half is that next call's error. */
int lslam_fmap_add_feature_cloud_begin(lslam_fmap *fm, const void *corner, size_t n_corner, const void *surf,
                                       size_t n_surf, size_t stride_bytes, const float T[16]);
int lslam_fmap_wait(lslam_fmap *fm);
/* getSurroundFeature, FeatureMap.h:256-265: the active cubes' clouds, concatenated in
 * _cubeValidInd order.  counts first, then the copy to the host ... */
int lslam_fmap_surround_counts(lslam_fmap *fm, size_t *n_corner, size_t *n_surf);
int lslam_fmap_get_surround(lslam_fmap *fm, float *corner_xyzi, size_t cap_corner, float *surf_xyzi,
                            size_t cap_surf);
/* ... or, without leaving HBM: the surround becomes the ctx's map (what
 * LaserMatcher::prepareFeatureSurround + ScanMatch.cpp:68-76 do through the host), kd-trees
 * built on the device. */
int lslam_fmap_surround_to_map(lslam_fmap *fm);
/* The same, handing back the surround's sizes it reads anyway (getSurroundFeature's two clouds): a caller that only wants to
 * know whether there is anything to match against (LaserMatcher.cpp:303-331) need not wait for lslam_fmap_surround_counts first. */
int lslam_fmap_surround_to_map_counts(lslam_fmap *fm, size_t *n_corner, size_t *n_surf);
/* ... or as a variant-C map: one kd-tree per cube of the active area (cubes with fewer than 5
 * points skipped, FeatureMap.h:524,546), all built in one go on the device; scan points are then
 * matched against the tree of the cube they fall into (FeatureMap::scanMatchScan, :490-691). */
int lslam_fmap_to_cubemap(lslam_fmap *fm);
/* The per-cube trees persist between calls: lslam_fmap_to_cubemap rebuilds only the trees of cubes whose cloud changed
 * since their tree was built (addFeatureCloud marks the cubes that received points; shifts and loads mark all).
 * Counts of the last call: trees built in it / trees kept from earlier calls. */
int lslam_fmap_cubemap_stats(lslam_fmap *fm, int64_t *trees_built, int64_t *trees_reused);
/* How lslam_fmap_add_feature_cloud's rebuilds of the point arrays went so far (per feature type and call).  The current points
 * are an earlier rebuild's output, hence in (cube, voxel) key order: only the new points are sorted and merged in (`merged`).
 * The order is checked on the device; when it does not hold -- a cube that has just become active gets voxel keys it did not
 * have, a centroid can round onto a voxel wall -- the whole array is sorted as before (`resorted`).  Same arrays either way. */
int lslam_fmap_rebuild_stats(lslam_fmap *fm, int64_t *merged, int64_t *resorted);
/* Forget the cached trees (the next lslam_fmap_to_cubemap rebuilds the whole active area). */
int lslam_fmap_cubemap_invalidate(lslam_fmap *fm);
/* getFullMap, FeatureMap.h:267-286: per cube, VoxelGrid(map leaf) of corner then surf. */
int lslam_fmap_get_full_map(lslam_fmap *fm, float *out_xyzi, size_t cap, size_t *n_out);
/* saveCloudToFiles / loadCloudFromFiles, FeatureMap.h:378-462: one binary PCD (fields x y z
 * intensity) per non-empty (cube, type) named <count>.pcd and index.txt with lines
 * "count type i j k size"; loading runs every loaded cube through its type's VoxelGrid and
 * replaces that cube's content.  ascii and binary PCDs are read, binary_compressed is not. */
int lslam_fmap_save(lslam_fmap *fm, const char *directory);
int lslam_fmap_load(lslam_fmap *fm, const char *directory);
/* introspection: grid origin, _cubeValidInd, points held per type; any output may be NULL */
int lslam_fmap_info(lslam_fmap *fm, int32_t origin[3], int32_t *n_valid, int32_t *valid_out, size_t cap,
                    size_t *n_corner_total, size_t *n_surf_total);
/* pcl::VoxelGrid<PointXYZI>::filter with a cubic leaf on one cloud (LaserMatcher.cpp:289-301,
 * ScanMatch.cpp:362-398 scanMatchLocal): one centroid {x,y,z,intensity} per occupied voxel, in
 * ascending voxel index.  Points of a voxel are summed in input order (PCL: unspecified). */
int lslam_voxel_grid(lslam_ctx *ctx, const void *cloud, size_t n, size_t stride_bytes, float leaf,
                     float *out_xyzi, size_t cap, size_t *n_out);
/* Two clouds with the same leaf in one pass -- prepareFeatureFrame's corner and surface clouds (LaserMatcher.cpp:289-301, two
 * VoxelGrid objects there): each cloud is filtered on its own (own min_b, own "leaf too small" guard), one upload, one wait,
 * one download instead of two of each.  Bit for bit what two lslam_voxel_grid calls give. */
int lslam_voxel_grid2(lslam_ctx *ctx, const void *cloud_a, size_t n_a, const void *cloud_b, size_t n_b, size_t stride_bytes,
                      float leaf, float *out_a_xyzi, size_t cap_a, size_t *n_out_a, float *out_b_xyzi, size_t cap_b, size_t *n_out_b);

/* ---- feature extraction front end ------------------------------
 * Replaces ScanRegistration::extractFeatures (odometry/ScanRegistration.cpp:190-425, with
 * setScanBuffersFor :471-531, setRegionBuffersFor :427-469, markAsPicked :533-555 and
 * pointClassify :557-687) on the ring-sorted full-resolution cloud MultiScanRegistration::process
 * builds (MultiScanRegistration.cpp:178-190).  Building that cloud from raw driver packets is the
 * caller's (ring from the vertical angle, IMU de-skew). */
typedef struct lslam_reg_params {     /* RegistrationParams, ScanRegistration.h:45-112 */
  int32_t n_feature_regions;          /* 6; 1 .. 512 (the device keeps the sort words of a ring's regions in LDS) */
  int32_t curvature_region;           /* 5 */
  int32_t max_corner_sharp;           /* 2 */
  int32_t max_surface_flat;           /* 4 */
  float less_flat_filter_size;        /* 0.2 */
  float surface_curvature_threshold;  /* 0.02 */
  float blind_threshold;              /* cos(deg2rad(blindDegreeThreshold = 0.5)) */
  int32_t reserved;
} lslam_reg_params;
void lslam_reg_default_params(lslam_reg_params *p);
/* cloud: n_points points, {x,y,z} at offset 0 and, at intensity_offset_bytes, the float copied to
 * the outputs' intensity (toXYZI, util/pcl_util.h:30-37: the `curvature` field = ring id +
 * relative time).  scan_ranges: n_scans x {first, last} inclusive index ranges (a ring of more
 * than 2560 points is refused).  sharp / less_sharp / flat / less_flat: room for n_points
 * {x,y,z,intensity} each (any may be NULL); counts = their sizes.  Optional taps, n_points each:
 * curvature (0 outside the feature regions), _scanNeighborPicked right after setScanBuffersFor,
 * final region label (PointLabel values, 6 = UNKNOW). */
int lslam_extract_features(lslam_ctx *ctx, const void *cloud, size_t n_points, size_t stride_bytes,
                           size_t intensity_offset_bytes, const int32_t *scan_ranges, size_t n_scans,
                           const lslam_reg_params *params, float *sharp, float *less_sharp, float *flat,
                           float *less_flat, size_t counts[4], float *curvature_out, int8_t *picked_out,
                           int8_t *label_out);

/* MultiScanRegistration::process (odometry/MultiScanRegistration.cpp:94-190) without the IMU
 * branch: raw driver cloud {x,y,z} -> ring-sorted cloud {x', y', z', ring + relTime} in the
 * registration's swapped axes (x' = y, y' = z, z' = x) with its per-ring {first, last} ranges --
 * the input of lslam_extract_features (intensity offset 12).  Linear ring mapper
 * (MultiScanRegistration.h:57-87: VLP-16 = -15..15 deg / 16, HDL-32 = -30.67..10.67 / 32).  atan /
 * atan2 are evaluated on the device: ring ids equal the reference's except for points within an
 * ulp of a ring boundary, relTime agrees to ~1e-7. */
int lslam_multiscan_register(lslam_ctx *ctx, const void *cloud, size_t n_points, size_t stride_bytes,
                             float lower_deg, float upper_deg, int32_t n_rings, float scan_period,
                             float *out_xyzc, size_t cap, size_t *n_out, int32_t *ranges_out);

/* ---- the odometry node resident on the device ("fully on device ... feeds the path without a CPU hop") ----
 *
 * A sweep's four feature clouds stay in HBM between the registration node and the odometry node: lslam_extract_features_dev
 * leaves them in an lslam_fset (the /laser_cloud_sharp, /laser_cloud_less_sharp, /laser_cloud_flat, /laser_cloud_less_flat
 * messages of ScanRegistration::publishResult, odometry/ScanRegistration.cpp, without the trip through the host), and
 * lslam_odom_process runs LaserOdometry::process (odometry/LaserOdometry.cpp:288-326) on them:
 *   first sweep     the less-sharp / less-flat clouds become _lastCornerCloud / _lastSurfaceCloud (:295-303)
 *   afterwards      scanMatch (:328-647) against the last clouds when they hold > 10 / > 100 points (:337) with the persistent
 *                   _transform as the initial guess, _Tsum = _Tsum * _transform (:649-653), transformToEnd of the less-sharp /
 *                   less-flat clouds (:312-313), which become the last clouds (:315-316)
 * with nothing but the 6 + 16 floats of the result (and, if asked for, the two last clouds the mapping node subscribes to)
 * crossing PCIe.  The nearest neighbour of :359 / :425 (nearestKSearch(pointSel, 1, ...), nanoflann_pcl.h:150-162) is found
 * through two hashed cell grids per last cloud (1 m cells, then 5.02 m cells: their 27-cell probe covers the 5 m gate of
 * :363,429 completely) with the proof of csrc/lslam_grid.hpp restated for k = 1: the smallest fp32 distance among the
 * candidates is nanoflann's answer when it is below the probe's guaranteed radius and no second point has the same distance;
 * an exact tie -- the one thing only nanoflann's visit order decides -- makes the call build the kd-trees after all and run
 * through them (lslam_odom_stats.tree_fallbacks counts those).  The grids are built by the same launch sequence that moves the
 * clouds to the sweep end: no kd-tree is built per sweep.  Same result as lslam_odometry_match + lslam_transform_to_end on
 * the same clouds, bit for bit (tests/test_gpu_odom.py). */
typedef struct lslam_fset lslam_fset;
typedef struct lslam_odom lslam_odom;
/* A feature set lives on the context's device; its buffers grow on demand.  It is complete when the call that fills it
 * returns, and free for the next fill once the lslam_odom_process that consumed it has returned (a program with the two nodes
 * on two threads rotates a few of them). */
int lslam_fset_create(lslam_ctx *ctx, lslam_fset **out);
void lslam_fset_destroy(lslam_fset *fs);
/* counts[4]: points in sharp, less-sharp, flat, less-flat */
int lslam_fset_counts(const lslam_fset *fs, size_t counts[4]);
/* Host clouds {x,y,z,intensity} (stride / intensity offset as lslam_odometry_match) into a feature set: the entry for callers
 * whose registration runs elsewhere, and for tests. */
int lslam_fset_upload(lslam_ctx *ctx, lslam_fset *fs, const void *sharp, size_t n_sharp, const void *less_sharp,
                      size_t n_less_sharp, const void *flat, size_t n_flat, const void *less_flat, size_t n_less_flat,
                      size_t stride_bytes);
/* One list back to the host, packed {x,y,z,intensity} (which: 0 sharp, 1 less-sharp, 2 flat, 3 less-flat). */
int lslam_fset_download(lslam_ctx *ctx, const lslam_fset *fs, int32_t which, float *out_xyzi, size_t cap, size_t *n_out);
/* lslam_extract_features with the four lists left in HBM (same kernels, same lists bit for bit). */
int lslam_extract_features_dev(lslam_ctx *ctx, const void *cloud, size_t n_points, size_t stride_bytes,
                               size_t intensity_offset_bytes, const int32_t *scan_ranges, size_t n_scans,
                               const lslam_reg_params *params, lslam_fset *out, size_t counts[4]);

typedef struct {
  int32_t matched;         /* 0: the first sweep, or the last clouds were too small (:337): no scan match ran */
  int32_t tree_fallbacks;  /* scan matches of this node so far that were redone through kd-trees (an exact distance tie) */
  int32_t searches;        /* correspondence refreshes of this sweep's loop (every fifth iteration, :357,:423) */
  int32_t reserved;
  uint64_t sweeps;         /* sweeps processed by this node so far */
  size_t n_last_corner, n_last_surf; /* the last clouds after this sweep */
} lslam_odom_stats;
/* LaserOdometry(scanPeriod, maxIterations = 25), _deltaTAbort = _deltaRAbort = 0.1 (LaserOdometry.cpp:24-25). */
int lslam_odom_create(lslam_ctx *ctx, int32_t max_iterations, float delta_t_abort, float delta_r_abort, lslam_odom **out);
void lslam_odom_destroy(lslam_odom *od);
/* LaserOdometry::process.  transform[6] (out) = _transform after the sweep, Tsum[16] (out) = _Tsum, row-major; stats (may be
 * NULL) as lslam_odometry_match fills it (zero when nothing was matched), ostats (may be NULL) the node's own.  last_corner /
 * last_surf (may be NULL): room for cap_corner / cap_surf packed {x,y,z,intensity} points -- the two clouds
 * LaserOdometry::publishResult sends to the mapping node (/laser_cloud_corner_last, /laser_cloud_surf_last), copied out behind
 * the same wait as the result.  Returns LSLAM_OK / LSLAM_NOT_CONVERGED as lslam_odometry_match, LSLAM_TOO_FEW_REF when nothing
 * was matched, a negative status on errors. */
int lslam_odom_process(lslam_odom *od, lslam_fset *fs, float transform[6], float Tsum[16], lslam_stats *stats,
                       lslam_odom_stats *ostats, float *last_corner, size_t cap_corner, float *last_surf, size_t cap_surf);
/* The last clouds as they are in HBM now (any pointer may be NULL; counts in lslam_odom_stats). */
int lslam_odom_last_clouds(lslam_odom *od, float *last_corner, size_t cap_corner, float *last_surf, size_t cap_surf);
/* Publishing without a copy on the host: with n_buffers > 0 every lslam_odom_process also leaves the two last clouds in one of
 * n_buffers page-locked buffers of the node, taken in turn, and lslam_odom_last_view hands out where (packed {x,y,z,intensity};
 * NULL / 0 before the first sweep).  A view stays valid until n_buffers further sweeps have been processed -- size the ring
 * for the sweeps the consumer (the mapping node) may lag behind -- or until the next lslam_odom_set_publish / _destroy.
 * n_buffers = 0 (the default) switches it off. */
int lslam_odom_set_publish(lslam_odom *od, int32_t n_buffers);
int lslam_odom_last_view(lslam_odom *od, const float **last_corner, size_t *n_corner, const float **last_surf, size_t *n_surf);
/* Profiling tap (a node made in a process with LSLAM_DEBUG_HOOKS=1 and LSLAM_ODOM_SEARCH_TAP=1): per query of the node's LAST
 * search launch four words -- 10 ns ticks, candidates looked at for the nearest neighbour, for the ring categories, bit 0 / 1 a
 * coarse-level pass in the former / the latter.  Returns the number of queries copied (0: tap off). */
int lslam_debug_odom_search(lslam_odom *od, uint32_t *out, size_t cap_queries);
/* Start again from the first sweep (keeps the buffers). */
int lslam_odom_reset(lslam_odom *od);

/* ---- coarse alignment of a loop-closure candidate ------------------------------
 * Replaces LoopDetector::corseMatching (pose_graph/loop_detector.hpp:232-255), i.e.
 * pcl::IterativeClosestPoint<PointXYZI, PointXYZI> with default settings: point-to-point ICP of `source`
 * onto `target` from the initial guess T (row-major 4x4, in/out = getFinalTransformation()).  PCL is not
 * part of the reference tree: PARITY UNPINNED (csrc/lslam_icp.hip states the restated defaults;
 * oracle/icp_oracle.py is the independent CPU statement).  max_iterations <= 0: 10; transformation_epsilon
 * 0 and max_correspondence_distance <= 0 (unlimited) are PCL's defaults.  *converged = hasConverged(),
 * *fitness = getFitnessScore().  An empty target returns converged = 0 (:233-235).  The context's resident
 * map is replaced by the target's kd-tree (like lslam_odometry_match). */
int lslam_icp_align(lslam_ctx *ctx, const void *target, size_t n_target, const void *source, size_t n_source,
                    size_t stride_bytes, float T[16], int32_t max_iterations, double transformation_epsilon,
                    double max_correspondence_distance, double *fitness, int32_t *converged, int32_t *iterations);

/* ---- SE(3) pose-graph Levenberg-Marquardt ------------------------------------
 * Replaces pose_graph::SolverG2O (pose_graph/solver_g2o.cpp:51-95): add_se3_node /
 * add_se3_edge build the arrays passed to lslam_pg_create, optimize() becomes
 * lslam_pg_optimize.  g2o conventions (VertexSE3 / EdgeSE3 / "lm_var"): a pose is
 * {tx,ty,tz, qx,qy,qz,qw}; an edge e has vertices ij[2e], ij[2e+1], measurement
 * meas7[7e..] and a row-major 6x6 information matrix over [translation, rotation]
 * (pose_graph/graph.cpp:279-288: diag(0.8,0.4,0.8,1,2,1) for odometry, :333-339: 2*I
 * for loop closures); vertex `fixed_vertex` is held fixed (solver_g2o.cpp:55-59).  fp64.
 *
 * The damped system is solved by preconditioned CG (block Jacobi + a coarse level of rigid-body motions of graph
 * aggregates, csrc/lslam_posegraph.hip) instead of g2o's sparse Cholesky: same optimum (tests/test_posegraph_bench_fixture.py).
 * When a workgroup per aggregate fits the device at once (the 5 000-keyframe bench graph does: 88 aggregates) the whole PCG
 * loop of a damped solve, and the inverse of the coarse matrix, each run as ONE persistent cooperative launch
 * (lslam_pg_stats.fused_solves counts them); larger graphs take a launch-per-step loop with the same arithmetic.  A
 * cooperative launch wants its workgroups resident together; if they are not (another process's persistent kernel on the
 * same device) a grid exchange runs into its spin limit, the kernel raises an abort flag instead of hanging, the solve is
 * redone by the launch-per-step loop and the graph stays on that loop (tested through a debug hook).
 *
 * Multi-GPU (one process per GPU): every rank creates the same graph and takes an edge range
 * (lslam_pg_set_shard); the block system [diagonal blocks | off-diagonal blocks | b | chi2 | flag] is summed
 * across ranks once per linearisation -- by the library's RCCL communicator on the solver's stream
 * (lslam_pg_set_comm, further up) or through the callback of lslam_pg_set_shard for hosts with their own
 * transport; the damped solve is replicated and bit-identical on every rank. */
typedef struct lslam_pg lslam_pg;

typedef struct {
  int32_t iterations;     /* LM iterations (SparseOptimizer::optimize return value) */
  int32_t lm_trials;      /* damped solves, accepted + rejected */
  int32_t cg_iterations;  /* total preconditioned-CG iterations */
  int32_t status;
  double chi2_initial, chi2_final, lambda;
  float gpu_ms_total;
  int32_t fused_solves;   /* damped solves whose whole PCG loop ran in the persistent kernel (the rest took the
                             launch-per-step loop: graph too large for one workgroup per aggregate to be co-resident) */
} lslam_pg_stats;

/* In-place SUM over all ranks of `count` doubles at DEVICE address `buf`; must have
 * completed when it returns. */
typedef void (*lslam_allreduce_fn)(void *user, double *buf, size_t count);

/* ---- collectives: RCCL inside the library (north star: "RCCL all-reduce over xGMI of
 * the block Hessian") ------------------------------------------------------------------------
 * One process per GPU.  Rank 0 makes an id (lslam_comm_unique_id), the host program carries its 128
 * bytes to the other ranks (MPI, a file, torch.distributed ...), every rank creates a communicator
 * on its device and attaches it to a context (lslam_ctx_set_comm) and / or a pose graph
 * (lslam_pg_set_comm).  The sharded paths then enqueue ncclAllReduce on the library's own stream
 * between the producing and the consuming kernel: no host round trip per iteration.  librccl is
 * dlopen'ed on first use.  The lslam_allreduce_fn callbacks below remain for hosts that bring their
 * own transport (and for tests on one GPU, where RCCL refuses two ranks on one device). */
#define LSLAM_COMM_ID_BYTES 128
typedef struct lslam_comm lslam_comm;
int lslam_comm_unique_id(uint8_t id[LSLAM_COMM_ID_BYTES]);
int lslam_comm_create(int device, const uint8_t id[LSLAM_COMM_ID_BYTES], int32_t rank, int32_t world,
                      lslam_comm **out);
void lslam_comm_destroy(lslam_comm *comm);
/* ncclGetVersion of the librccl the library loaded (e.g. 22606 = 2.26.6) */
int lslam_comm_version(int32_t *version);
/* rank and rank count as the RCCL communicator itself reports them (ncclCommUserRank, ncclCommCount) */
int lslam_comm_info(const lslam_comm *comm, int32_t *rank, int32_t *world);
/* In-place SUM of `count` doubles at DEVICE address buf over all ranks, enqueued on hip_stream. */
int lslam_comm_allreduce_f64(lslam_comm *comm, double *device_buf, size_t count, void *hip_stream);
/* Attach (or, with NULL, detach) a communicator: lslam_scanmatch_run_sharded then needs no callback.
 * The communicator must outlive its use by the context. */
int lslam_ctx_set_comm(lslam_ctx *ctx, lslam_comm *comm);

/* ONE scan's points sharded over the ranks (the reference has no such seam: it is the
 * data-parallel form of ScanMatch.cpp:97-209 -- every point's row is independent given the
 * pose, the only coupling is the sum A^T A, A^T b and the counters).  Each rank holds its
 * contiguous shard of the corner and surf points (lslam_scan_set) and the whole map; per
 * Gauss-Newton iteration the rank's 32 fp64 sums (21 upper-triangular A^T A, 6 A^T b, rows,
 * line matches, plane matches, score, spare) are copied to xchg32 (DEVICE, 32 doubles, caller
 * owned -- e.g. a torch tensor the hook can all-reduce over RCCL), fn sums them over the
 * ranks in place, and every rank runs the same 6x6 solve on the same numbers.  The summation
 * order differs from the single-GPU loop: poses agree to ~1e-6, not bit for bit.
 * With a communicator attached (lslam_ctx_set_comm) fn and xchg32 may be NULL: the sums are then
 * all-reduced by RCCL on the library's stream and the whole loop stays device-resident. */
int lslam_scanmatch_run_sharded(lslam_ctx *ctx, float pose[6], const lslam_opts *opts,
                                lslam_allreduce_fn fn, void *user, double *xchg32,
                                lslam_stats *stats);

/* ---- joint LiDAR + stereo term (BASELINE configs[4]) ------------------
 *
 * The reference only announces this (README.md:51-71: ORB-SLAM2 on a ZED stereo camera, "planning
 * to extend the LOAM module to integrate Lidar and Visual SLAM methods"): it holds no code for a
 * visual term, so there is nothing to cite line by line and PARITY IS UNPINNED.  The
 * arithmetic restated here (and in oracle/lslam_oracle.c) is the published one of ORB-SLAM2's
 * pose-only stereo edge (g2o EdgeStereoSE3ProjectXYZOnlyPose as used by
 * Optimizer::PoseOptimization): a landmark X_w (map frame) seen at (uL, v, uR) projects as
 *   X_c = R_cl * (R^T (X_w - t)) + t_cl          (R, t: the Twist being optimised, sensor -> map)
 *   uL' = fx x/z + cx,  v' = fy y/z + cy,  uR' = uL' - bf/z
 * error e = (uL'-uL, v'-v, uR'-uR), chi2 = |e|^2 * inv_sigma2, Huber weight with
 * delta = sqrt(7.815) (sqrt(5.991) for a monocular observation, uR < 0: two rows), optional
 * outlier gate chi2 > delta^2.  Its rows, scaled by sqrt(weight * inv_sigma2 * w_huber), are added
 * to the LOAM rows of ScanMatch.cpp:134-204 in the SAME 6x6 A^T A / A^T b of every Gauss-Newton
 * iteration (Jacobian taken with respect to the same six Twist parameters), followed by the same
 * solve / degeneracy / convergence steps (ScanMatch.cpp:206-260). */
typedef struct {
  float fx, fy, cx, cy;   /* rectified pinhole intrinsics */
  float bf;               /* stereo baseline times fx (ORB-SLAM2's mbf) */
  float T_cl[12];         /* camera-from-lidar rigid transform, row-major 3x4 [R_cl | t_cl] */
  float weight;           /* lambda: scale of the stereo block against the LiDAR block */
  float huber_stereo;     /* sqrt(7.815) */
  float huber_mono;       /* sqrt(5.991) */
  int32_t gate_outliers;  /* 1: skip observations whose chi2 exceeds delta^2 at the current pose */
  float min_depth;        /* observations with camera z <= min_depth are skipped */
} lslam_stereo_cam;
void lslam_stereo_default_cam(lslam_stereo_cam *cam);
/* Makes n observations resident on the context: landmarks_xyz[n][3] (map frame),
 * obs[n][3] = (uL, v, uR; uR < 0: monocular), inv_sigma2[n] (NULL: all 1).  From then on
 * lslam_scanmatch_run / _scan / _run_sharded of a SINGLE resident scan assemble the joint
 * system (under _run_sharded each rank holds its own shard of the observations; the stereo sums
 * travel in the same 32-double all-reduce).  n = 0 removes the term. */
int lslam_stereo_set(lslam_ctx *ctx, const float *landmarks_xyz, const float *obs, const float *inv_sigma2,
                     size_t n, const lslam_stereo_cam *cam);
int lslam_stereo_clear(lslam_ctx *ctx);
/* Parity tap: the stereo term alone at `pose`: sums32 = {21 upper-triangular A^T A, 6 A^T b,
 * rows, 0, 0, 0, observations used} (fp64 reduction of the per-block fp32 partials). */
int lslam_stereo_sums(lslam_ctx *ctx, const float pose[6], double sums32[32]);

int lslam_pg_create(int device, int32_t n_vertices, const double *poses7, int32_t n_edges,
                    const int32_t *ij, const double *meas7, const double *info36,
                    int32_t fixed_vertex, lslam_pg **out);
void lslam_pg_destroy(lslam_pg *pg);
const char *lslam_pg_last_error(void);
/* Edge shard [e_begin, e_end) of this rank; fn == NULL: single GPU.  system_buf: optional
 * caller-owned DEVICE buffer of lslam_pg_system_doubles() doubles to assemble into. */
int lslam_pg_set_shard(lslam_pg *pg, int32_t e_begin, int32_t e_end, lslam_allreduce_fn fn,
                       void *user, double *system_buf);
/* The same with the library's own RCCL communicator instead of a callback (NULL detaches). */
int lslam_pg_set_comm(lslam_pg *pg, lslam_comm *comm);
size_t lslam_pg_system_doubles(const lslam_pg *pg);
/* ONE solve shared by the ranks (large graphs: the replicated solve above does not get faster with more GPUs).  Every rank also
 * takes a range of vertex ROWS [v_begin, v_end) -- whole 21-vertex row blocks, lslam_pg_row_shard_range gives an even
 * partition -- and the damped system is then solved by a row-sharded block-Jacobi PCG: each rank multiplies, updates and
 * preconditions its own rows; per iteration the ranks exchange one scalar (p . A p) and the vector z with r . z, r . r behind it
 * (each rank's rows in a zero-padded buffer of 6 n + 2 doubles: its all-reduce IS the gather; or a true all-gather, see
 * lslam_pg_set_row_gather below), through the same transport as
 * the linearisation (callback or RCCL communicator; the buffer is the tail of the system buffer).  Same iterates as the
 * single-process block-Jacobi solve up to the order of the sums.  The dense second level of the preconditioner is a
 * single-device structure: it is not used (and does not switch itself on) in this mode; with LSLAM_PG_COARSE=1 the solve
 * stays replicated.  (-1, -1) returns to the replicated solve. */
void lslam_pg_row_shard_range(int32_t n_vertices, int32_t rank, int32_t world, int32_t *v_begin, int32_t *v_end);
int lslam_pg_set_row_shard(lslam_pg *pg, int32_t v_begin, int32_t v_end);
int32_t lslam_pg_row_sharded_solves(const lslam_pg *pg); /* damped solves that took the row-sharded form so far */
/* The exchange of z as an ALL-GATHER of the owned segments instead of the zero-padded all-reduce (half the bytes on the wire,
 * no zeroing pass): taken whenever every rank's rows are lslam_pg_row_shard_range's partition -- each rank can then name every
 * segment without asking -- and the transport can gather.  The RCCL communicator (lslam_pg_set_comm) can: one ncclBroadcast per
 * segment rooted at its owner, all in one group = one launch; each rank's parts of r . z and r . r travel in the same group
 * and are summed in rank order on every rank (same bits everywhere).  A host with its own transport registers a second
 * callback type: in-place, segment r of buf = doubles [offsets[r], offsets[r + 1]), valid on rank r on entry and on every
 * rank on return; complete when it returns.  Ranges that are not the canonical partition, or no gather transport: the
 * all-reduce form.  Which form is taken is ONE decision of all ranks, taken by every row-sharded solve: each rank sums a
 * "not from me" flag over the ranks -- raised without a gather transport or with a range that is not the canonical one -- (one
 * scalar all-reduce through the linearisation's transport) and follows the result: a rank cannot see the others' ranges or
 * transports, and every rank takes part in that sum whatever its own setting.  At most 64 ranks. */
typedef void (*lslam_allgatherv_fn)(void *user, double *buf, const int64_t *offsets, int32_t world);
int lslam_pg_set_row_gather(lslam_pg *pg, lslam_allgatherv_fn fn, void *user, int32_t rank, int32_t world);
int32_t lslam_pg_row_gathered_solves(const lslam_pg *pg); /* ... of which exchanged by all-gather */
int32_t lslam_pg_num_offdiag(const lslam_pg *pg);
/* Relative residual |r| / |b| at which a damped solve's PCG stops.  Default 1e-8: to the LM schedule the solves are then what
 * g2o's direct factorisation ("lm_var", solver_g2o.cpp:16) gives it -- the trajectory of iterates is the oracle's.  A looser
 * value is an inexact Levenberg-Marquardt: fewer PCG iterations per solve, a different trajectory (accept / reject decisions
 * move), the same optimum -- on the bench graph 1e-3 ends at the same chi2 to 1e-6 relative and the same keyframe positions
 * to 1e-4 m in a comparable number of iterations at about twice the rate (tests/test_posegraph_bench_fixture.py; bench.py
 * reports it next to the default, never instead of it). */
int lslam_pg_set_solve_tolerance(lslam_pg *pg, double rel_tol);
/* SolverG2O::optimize (solver_g2o.cpp:79-95): up to max_iters LM iterations. */
int lslam_pg_optimize(lslam_pg *pg, int32_t max_iters, lslam_pg_stats *stats);
int lslam_pg_get_poses(lslam_pg *pg, double *poses7);
/* One-call form: poses7 is in/out; single GPU. */
int lslam_posegraph_optimize(int device, int32_t n_vertices, double *poses7, int32_t n_edges, const int32_t *ij,
                             const double *meas7, const double *info36, int32_t fixed_vertex, int32_t max_iters,
                             lslam_pg_stats *stats);
/* SolverG2O::save (solver_g2o.cpp:97-100): the graph with its current estimates in g2o's text
 * format (VERTEX_SE3:QUAT / FIX / EDGE_SE3:QUAT with the 21 upper-triangular information
 * entries) -- the route to cross-check this solver against an external g2o. */
int lslam_pg_save_g2o(lslam_pg *pg, const char *path);
/* Reader for that format (host only).  *n_vertices / *n_edges: in = capacity of the arrays, out =
 * what the file holds (call with NULL arrays to size them); vertex ids are renumbered 0..n-1 in
 * order of appearance; *fixed_vertex = first FIX id or -1. */
int lslam_g2o_read(const char *path, int32_t *n_vertices, double *poses7, int32_t *n_edges, int32_t *ij,
                   double *meas7, double *info36, int32_t *fixed_vertex);
/* Parity taps: the assembled system at the current estimate (diag[n_v*36],
 * off[n_off*36] with its (i<j) pairs off_ij[n_off*2], b[n_v*6], chi2), and one damped
 * solve (H + lambda I) dx = b. Any output may be NULL. */
int lslam_pg_linearize(lslam_pg *pg, double *diag_out, double *off_out, int32_t *off_ij_out,
                       double *b_out, double *chi2_out);
int lslam_pg_solve(lslam_pg *pg, double lambda, double *dx_out, int32_t *cg_iters);

/* Device handle taps for harnesses that time on the library's stream. */
void *lslam_stream(lslam_ctx *ctx); /* hipStream_t */
/* Sweep launches of this context so far, per kernel instantiation: [0] whole stack in LDS, [1] the same with the HBM
 * overflow (trees deeper than 33 levels), [2] the shallow-stack batch kernel sweep_kernel<256,true,false,12>, [3] per-cube
 * trees, [4] per-cube trees with overflow, [5] packet search, [6] persistent Gauss-Newton kernel, [7] the whole-stack kernel with
 * the 6x6 solve fused into its tail (the Gauss-Newton loop of single scans: one launch per iteration). */
void lslam_debug_sweep_launches(lslam_ctx *ctx, uint64_t counts[8]);
/* ... and of the grid sweep (sweep_grid_kernel; LSLAM_SEARCH_GRID) */
uint64_t lslam_debug_grid_launches(lslam_ctx *ctx);
/* ... of which the single-launch form for a map without trees (sweep_grid_kernel<256, true>) */
uint64_t lslam_debug_grid_wide_launches(lslam_ctx *ctx);
/* cells of the resident map's two cell tables (corner, surf); 0 while a type has no grid */
void lslam_debug_grid_cells(lslam_ctx *ctx, uint64_t out[2]);
/* out[0] maps set with deferred trees, out[1] of those whose trees were built after all, out[2] 1 while the resident map's are pending */
void lslam_debug_lazy_trees(lslam_ctx *ctx, uint64_t out[3]);
/* Debug tap of the certificate sweep (csrc/lslam_kernels.hip sweep_body): out[2] = second-pass launches of this
 * context since its creation; out[0] = points the certificate-testing workgroups left to the second pass and out[1] = points
 * of those workgroups, counted only when the process runs with LSLAM_DEBUG_CERT_STATS=1 (two atomics per workgroup). */
void lslam_debug_cert_stats(lslam_ctx *ctx, uint64_t out[3]);
/* ... and the grid sweep's (lslam_opts.debug_stats = 1 during the runs): out[((type * 8) + sweep) * 2 + {0, 1}] = points the
 * probe could not prove (left to the tree search) / points swept, by feature type (0 corner, 1 surf) and sweep of the
 * Gauss-Newton loop (0 = a loop's first sweep; slot 7 = the eighth and every later one).  Counted by the planner of the second
 * pass from the lists' lengths: the sweep kernel itself runs the same code with the tap on. */
void lslam_debug_grid_stats(lslam_ctx *ctx, uint64_t out[32]);
/* Test tap: what the certificate sweep carries per resident scan point after a lslam_scanmatch_run* that ran it -- the
 * map-frame position of the point's last SEARCH (q_xyz0: 4 floats per point, the fourth unused) and the lower bound taken
 * there of the squared distance of every map point outside its five neighbours (lb; 0: none).  Resident order: per scan its
 * corner points, then its surf points (each in the library's own order).  Returns the number of points copied (at most
 * cap_points) or a negative status. */
int lslam_debug_cert_state(lslam_ctx *ctx, float *q_xyz0, float *lb, size_t cap_points);

#ifdef __cplusplus
}
#endif
#endif /* LSLAM_C_H */
