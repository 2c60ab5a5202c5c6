// lslam_api.hip -- host side of the C ABI declared in include/lslam_c.h.
//
// Owns the HIP stream, the HBM-resident map (two kd-trees), the resident scan and
// the device Gauss-Newton state; enqueues the sweep/solve kernels of one
// scanMatchScan call back to back with no host round trip in between.
#include "../../include/lslam_c.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>

#include "lslam_internal.hpp"

using namespace lslam;

namespace {

thread_local std::string g_err;
std::mutex g_live_mu;
std::set<const lslam_ctx *> g_live;  // contexts that exist (dependants check before touching a stream)

void set_err(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return LSLAM_ERR_HIP;                                                        \
    }                                                                              \
  } while (0)

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  bool borrowed = false;  // p points into another allocation (adopt): never freed here, dropped by the next reserve
  void adopt(T *ptr, size_t n) {
    if (p && !borrowed) (void)hipFree(p);
    p = ptr;
    cap = n;
    borrowed = true;
  }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p && !borrowed) (void)hipFree(p);
    borrowed = false;
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p && !borrowed) (void)hipFree(p);
    borrowed = false;
    p = nullptr;
    cap = 0;
  }
};

struct DevTree {
  DevBuf<KdNode> nodes;
  DevBuf<PNode> pn;  // packet-search nodes, same slots: made on demand (ensure_packet_nodes)
  DevBuf<float> own_box;  // [slots][6] tight boxes recorded by the build (whole-map trees only)
  DevBuf<float4> pts;
  TreeView view{};
  int depth = 0;
  int cap_attempt = 0;  // node-slot multiplier that last fitted this tree (device build)
};

// A helper thread that lives as long as its context: the corner tree of a map is built from it while the calling thread
// builds the surf tree.  (A std::thread per lslam_map_set cost ~100 us before its first launch -- creation plus the HIP
// runtime's per-thread set-up -- and with the level phase at seven launches per level the corner tree, started that late, had
// become the end of the build.)
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, busy = false, quit = false;
  void start() {
    if (th.joinable()) return;
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        std::function<void()> f = std::move(job);
        has_job = false;
        lk.unlock();
        f();
        lk.lock();
        busy = false;
        cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> f) {
    start();
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(f);
    has_job = true;
    busy = true;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return !busy; });
  }
  void stop() {
    if (!th.joinable()) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      cv.notify_all();
    }
    th.join();
  }
};

struct HostSinCos {
  void operator()(float a, float &s, float &c) const {
    s = std::sin(a);  // util/Angle.h:17-18 std::sin/std::cos(float)
    c = std::cos(a);
  }
};

}  // namespace

struct lslam_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // the corner tree is built beside the surf tree; the corner map's cell grid beside the surf map's
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // ... forked from and joined to `stream` by these
  ScanPrep *scanprep = nullptr;
  Worker worker;  // builds the corner tree beside the surf tree
  int cube_sides_on_device = 0;  // cube-map sides built by the device forest builder (of the last set)
  DevTree tc, ts;
  // variant C: per-cube trees (shared node/point arrays in tc/ts, one TreeView per cube)
  bool cube_mode = false;
  CubeGridDev gc{}, gs{};
  DevBuf<int32_t> cell_c, cell_s;
  DevBuf<TreeView> views_c, views_s;
  bool have_map = false;
  bool have_scan = false;
  lslam_map_info info{};
  // resident scans (a batch of independent scan-match problems against the same map)
  DevBuf<float4> q;             // all points, per scan: corner block-run then surf
  DevBuf<BlockDesc> blocks;
  DevBuf<ProbBlocks> probs;
  std::vector<BlockDesc> h_blocks;
  std::vector<ProbBlocks> h_probs;
  std::vector<int32_t> nqc, nqs;  // per scan
  int32_t n_prob = 0;
  int32_t nb_total = 0;
  size_t n_points = 0;
  DevBuf<float> partials;
  DevBuf<char> scan_tables;    // [blocks | probs | groups] of the resident scans: blocks / probs / groups point into it
  std::vector<char> h_tables;  // ... its host image
  DevBuf<int32_t> tail_count;  // per resident scan: the fused solve's ticket counter (zero between launches)
  DevBuf<uint32_t> stack_ovf;  // only allocated for trees deeper than KD_STACK_LDS+1
  DevBuf<int32_t> prev_nb;     // neighbours of the previous sweep, per resident scan point
  DevBuf<float4> prev_q;       // ... and where the point was then,
  DevBuf<float> prev_lb;       // ... with the bound the certificate needs (sweep_body)
  DevBuf<uint8_t> need_list;   // certificate sweep: the points pass 1 leaves to pass 2 (SweepArgs)
  DevBuf<uint16_t> need_cnt;
  DevBuf<uint16_t> need2_list, need2_cnt;  // grid sweep: what its second probe leaves to the tree search (SweepArgs)
  DevBuf<GroupDesc> groups;
  DevBuf<int32_t> cert_work;   // [blocks] work list of pass 2 (CertPlan)
  DevBuf<int32_t> cert_count;  // [2] items + [2] tickets
  std::vector<GroupDesc> h_groups;
  std::vector<int32_t> h_prob_group0;  // [n_prob + 1] first group of every scan
  uint64_t queue_launches = 0;
  DevBuf<int32_t> active_blocks;  // [nb_total] the grid sweep of a batch: block indices of the scans still running, per chunk at its block range
  int32_t *h_active = nullptr;    // pinned: [n_chunks] how many
  size_t h_active_cap = 0;
  DevBuf<int32_t> d_active_cnt;
  DevBuf<int32_t> fit_ids;     // the grid sweep's fit cache (LSLAM_AB_FIT_CACHE): [5][n_points] neighbour positions ...
  DevBuf<float> fit_val;       // ... and [5][n_points] plane + verdict
  DevBuf<unsigned long long> cert_stats;  // LSLAM_DEBUG_CERT_STATS=1: [searched, swept] counters of the certificate path
  bool prev_valid = false;
  bool grid_state_valid = false;  // prev_q holds what a grid-sweep run of the resident scan against the resident map left (LSLAM_SWEEP_CARRIED)
  lslam_comm *comm = nullptr;  // RCCL communicator of the sharded-points path (not owned)
  DevBuf<double> xchg;         // its exchange buffer
  GNState *d_state = nullptr;   // [state_cap]
  GNState *h_state = nullptr;   // pinned, [state_cap]
  int32_t state_cap = 0;
  // tap buffers
  DevBuf<int32_t> t_idx;
  DevBuf<float> t_d2;
  DevBuf<float4> t_coeff;
  DevBuf<uint8_t> t_flags;
  DevBuf<float4> t_q;
  DevBuf<float> t_small;  // AtA/Atb upload for the gn_step tap
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> sweep_ev;
  int iter_hint = 4;  // size of the first batch of enqueued GN iterations
  int iter_last = -1, iter_same = 0;  // the last call's iteration count, and for how many calls in a row it has been that
  // gn_persistent_kernel (one resident scan: the whole loop in one cooperative launch)
  DevBuf<float> gnp_slots;
  DevBuf<unsigned> gnp_bar;
  int gnp_cap = -1;       // workgroups the device holds at once (-1: not asked yet)
  bool gnp_ok = true;     // false once an exchange timed out: the launch loop from then on
  int gnp_runs = 0;
  // pinned staging area for the scan clouds a caller hands over (packed here, copied from here)
  float4 *h_stage = nullptr;
  size_t h_stage_cap = 0;
  bool stage_busy = false;  // a copy out of h_stage was enqueued and no wait on the stream has happened since
  // ... and for the two map clouds of lslam_map_set (one per tree: they are packed on two threads)
  float4 *h_map_stage[2] = {nullptr, nullptr};
  size_t h_map_cap[2] = {0, 0};
  // variant B (lslam_odometry_match): clouds, correspondences (grow-only, reused across sweeps)
  DevBuf<float4> od_oc, od_os, od_q, od_sel;
  DevBuf<int32_t> od_ind;
  int od_iter_hint = 6;
  // stereo term of the joint system (lslam_stereo_set)
  DevBuf<float4> st_lm, st_obs;
  DevBuf<float> st_partials;
  DevBuf<ProbBlocks> st_noblocks;  // {0, 0}: a problem with no LiDAR blocks (lslam_stereo_sums)
  int32_t n_stereo = 0;
  StereoCam st_cam{};
  uint64_t sweep_variants[SWEEP_N_VARIANTS] = {0};  // sweep launches per kernel instantiation (lslam_debug_sweep_launches)
  // cell grids over the whole-map trees (lslam_grid.hpp), built the first time the grid search is asked for on a map
  GridDev kc, ks;
  uint64_t map_epoch = 0;      // bumped whenever the resident trees change
  uint64_t grid_epoch = ~0ull; // the map the grids were built from
  float grid_cell = 0.0f;      // ... and their cell size
  int grid_status = 0;         // why they could not be built (GridDev::build), 0: fine
  // deferred kd-trees (lslam_map_defer_trees): a map handed over on the device gets its cell grids at once and its trees on
  // first need -- a tap, the lane search, a query whose neighbours need nanoflann's visit order (an exact distance tie)
  bool defer_trees = false;    // the caller's wish
  bool trees_pending = false;  // the resident map has grids but no trees yet
  DevBuf<uint32_t> bbox6;      // bounding-box reduction scratch
  DevBuf<int32_t> wide_p;      // [points][5] / [points][5]: neighbours the wide probe found for the points the 27-cell probe
  DevBuf<float> wide_d;        //   could not prove (grid sweep without trees)
  DevBuf<int32_t> wide_off;    // [blocks + 1] exclusive prefix of need_cnt, [blocks + 1 .. +2) {total, need-tree flag}
  uint64_t lazy_sets = 0, lazy_builds = 0;  // maps set with deferred trees / trees then built after all (lslam_debug_lazy_trees)
  // environment overrides of lslam_opts fields, read ONCE when the context is made (never inside a call)
  int env_knn_cert = -1;       // LSLAM_KNN_CERT (-1: not set)
  float env_cert_try_m = -1.0f, env_cert_track_m = -1.0f, env_grid_cell = -1.0f;  // LSLAM_CERT_TRY_M, LSLAM_CERT_TRACK_M, LSLAM_GRID_CELL
  int env_fit_from_sweep = 0;  // LSLAM_FIT_FROM_SWEEP: A/B -- the sweep of a loop from which cached fits are used (default 3)
  float env_grid_cell_corner = -1.0f;  // LSLAM_GRID_CELL_CORNER: A/B switch -- another cell edge for the corner map's grid (the line-like cloud)
  int env_search = -1;         // LSLAM_SEARCH=lane|packet|grid
  int env_debug_cert_stats = 0;  // LSLAM_DEBUG_CERT_STATS
  int env_force_stack = -1;    // LSLAM_FORCE_STACK=deep|shallow|auto -> SWEEP_STACK_*
  int env_ab = 0;              // LSLAM_PERSISTENT_GN=1, LSLAM_FUSED_SOLVE=1 -> LSLAM_AB_* bits
  int ab_now = 0;              // lslam_opts.ab_switches | env_ab of the call that is running (sweep_launch has no options at hand)
  bool env_debug = false;      // LSLAM_DEBUG
};

namespace {

// strided cloud -> packed float4 {x,y,z,w}
void pack_cloud(const void *src, size_t n, size_t stride_bytes, std::vector<float4> &out) {
  out.resize(n);
  const char *p = static_cast<const char *>(src);
  for (size_t i = 0; i < n; ++i) {
    float xyz[3];
    std::memcpy(xyz, p + i * stride_bytes, sizeof(xyz));
    out[i] = make_float4(xyz[0], xyz[1], xyz[2], 0.0f);
  }
}

// Scan points are processed one per lane; lanes of a wave run in lockstep, so a wave is
// as slow as its most expensive traversal.  Ordering the scan along a Morton curve
// (0.25 m cells) makes the 64 points of a wave spatial neighbours that walk the same
// part of the kd-tree.  The original index travels in .w: per-point outputs are
// written back in the caller's order, only the order of summation changes.
inline uint32_t spread10(uint32_t v) {
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

void morton_order(std::vector<float4> &pts) {
  const size_t n = pts.size();
  std::vector<std::pair<uint32_t, uint32_t>> key(n);
  for (size_t i = 0; i < n; ++i) {
    auto q = [](float v) {
      float c = v * 4.0f + 512.0f;  // 0.25 m cells, +-128 m
      c = c < 0.0f ? 0.0f : (c > 1023.0f ? 1023.0f : c);
      return (uint32_t)c;
    };
    key[i] = {spread10(q(pts[i].x)) | (spread10(q(pts[i].y)) << 1) | (spread10(q(pts[i].z)) << 2),
              (uint32_t)i};
  }
  std::sort(key.begin(), key.end());
  std::vector<float4> out(n);
  for (size_t i = 0; i < n; ++i) {
    float4 v = pts[key[i].second];
    v.w = __builtin_bit_cast(float, key[i].second);
    out[i] = v;
  }
  pts.swap(out);
}

void init_state(GNState &s, const float pose[6]) {
  std::memset(&s, 0, sizeof(s));
  for (int i = 0; i < 6; ++i) s.pose[i] = pose[i];
  pose_to_Rt_sc(pose, s.R, s.t, s.sc, HostSinCos());
}

void fill_sweep_args(lslam_ctx *ctx, SweepArgs &a) {
  a.tc = ctx->tc.view;
  a.ts = ctx->ts.view;
  a.gc = CubeGridDev{};
  a.gs = CubeGridDev{};
  if (ctx->cube_mode) {
    a.gc = ctx->gc;
    a.gs = ctx->gs;
  }
  a.kc = CellGrid{};
  a.ks = CellGrid{};
  a.grid = 0;
  a.grid_clip_margin = GRID_CLIP_MARGIN_MIN;
  a.wide_nf_slack = 0.0f;
  a.active_blocks = nullptr;
  a.n_active = 0;
  a.fit_ids = nullptr;
  a.fit_val = nullptr;
  a.n_fit = 0;
  a.fit_from_sweep = 0;
  a.grid_hint = nullptr;
  a.need2_list = nullptr;
  a.need2_cnt = nullptr;
  a.wide_d = nullptr;
  a.wide_p = nullptr;
  a.wide_off = nullptr;
  a.q = ctx->q.p;
  a.blocks = ctx->blocks.p;
  a.nb_total = ctx->nb_total;
  a.states = ctx->d_state;
  a.partials = ctx->partials.p;
  a.stack_ovf = nullptr;
  a.prev_nb = ctx->prev_nb.p;
  a.prev_q = nullptr;
  a.prev_lb = nullptr;
  a.prev_valid = 0;
  a.bounded = 0;
  a.deep_tree = (ctx->tc.depth > KD_STACK_LDS + 1 || ctx->ts.depth > KD_STACK_LDS + 1) ? 1 : 0;
  a.packet = 0;
  a.stack_mode = SWEEP_STACK_AUTO;
  a.fine_gate_c = a.fine_gate_s = -1.0f;
  a.tail = SweepTail{};
  a.cert_stats = nullptr;
  a.need_list = nullptr;
  a.need_cnt = nullptr;
  a.groups = nullptr;
  a.n_groups = 0;
  a.group_block_base = 0;
  a.cert_try_m = 0.0f;
  a.cert_track_m = 0.0f;
  a.idx_out = nullptr;
  a.d2_out = nullptr;
  a.coeff_out = nullptr;
  a.flags_out = nullptr;
  a.dbg = nullptr;
}

// LSLAM_SEARCH_AUTO = the per-lane search.  The packet search (lslam_packet.hpp) removes the divergent gathers
// that bound the per-lane kernel, but pays ~2x the VALU work (every lane tests the union of the packet's nodes
// and leaves, every insert runs when ANY lane needs it): measured on MI355X it is SLOWER -- 0.81 against 0.45 ms
// per 2.6 M-point launch, 277 against 59 us for a single 115 200-point scan (incoherent far-range packets make
// a long tail) -- so it stays an explicit choice (LSLAM_SEARCH_PACKET, or LSLAM_SEARCH=packet for A/B runs).
// The packet search's nodes of the resident whole-map trees, made the first time that search is asked for on this map.
int ensure_packet_nodes(lslam_ctx *ctx) {
  if (!ctx->have_map || ctx->cube_mode) return LSLAM_OK;
  for (DevTree *dt : {&ctx->tc, &ctx->ts}) {
    if (dt->view.pn || !dt->own_box.p) continue;
    HIP_TRY(dt->pn.reserve((size_t)std::max(dt->view.n_nodes, 1)));  // (a tree whose root is a leaf has no inner node)
    HIP_TRY(build_packet_nodes(dt->view, dt->own_box.p, dt->pn.p, ctx->stream));
    dt->view.pn = dt->pn.p;
  }
  return LSLAM_OK;
}

// The cell grids of the resident whole-map trees (lslam_grid.hpp), made the first time the grid search is asked for on this
// map with this cell size.  LSLAM_OK with ctx->kc.view.cell_start == nullptr: no grid for this map (ctx->grid_status).
int ensure_grid(lslam_ctx *ctx, float cell) {
  if (!ctx->have_map || ctx->cube_mode) return LSLAM_OK;
  if (!(cell > 0.0f)) cell = GRID_CELL_DEFAULT;
  if (ctx->grid_epoch == ctx->map_epoch && ctx->grid_cell == cell) return LSLAM_OK;
  ctx->grid_epoch = ctx->map_epoch;
  ctx->grid_cell = cell;
  int st_c = 0, st_s = 0;
  HIP_TRY(ctx->kc.build(ctx->tc.view, ctx->env_grid_cell_corner > 0.0f ? ctx->env_grid_cell_corner : cell, ctx->stream, &st_c));
  HIP_TRY(ctx->ks.build(ctx->ts.view, cell, ctx->stream, &st_s));
  ctx->grid_status = st_c ? st_c : st_s;
  if (ctx->grid_status || !ctx->kc.view.cell_start || !ctx->ks.view.cell_start) {  // both or none
    ctx->kc.view = CellGrid{};
    ctx->ks.view = CellGrid{};
  }
  return LSLAM_OK;
}

int resolve_search_mode(const lslam_ctx *ctx, int32_t requested) {
  if (ctx->env_search >= 0) requested = ctx->env_search;
  requested &= 0xFF;  // the LSLAM_STACK_* bits are resolve_stack_mode's
  if (ctx->cube_mode) return LSLAM_SEARCH_LANE;
  if (requested == LSLAM_SEARCH_PACKET) {  // its nodes exist only once somebody has asked for it
    if (ensure_packet_nodes(const_cast<lslam_ctx *>(ctx)) != LSLAM_OK || !ctx->tc.view.pn || !ctx->ts.view.pn) return LSLAM_SEARCH_LANE;
    return requested;
  }
  if (requested == LSLAM_SEARCH_GRID) return LSLAM_SEARCH_GRID;  // (the caller builds the grids: ensure_grid)
  return LSLAM_SEARCH_LANE;
}

// Traversal-stack shape asked for: LSLAM_STACK_* bits of a search mode, overridden by LSLAM_FORCE_STACK=deep|shallow|auto
// of the environment the context was made in.
int resolve_stack_mode(const lslam_ctx *ctx, int32_t search_mode) {
  if (ctx->env_force_stack >= 0) return ctx->env_force_stack;
  return (search_mode & LSLAM_STACK_SHALLOW) ? SWEEP_STACK_SHALLOW : ((search_mode & LSLAM_STACK_DEEP) ? SWEEP_STACK_DEEP : SWEEP_STACK_AUTO);
}

// launch_sweep + the per-context count of the instantiation it took
hipError_t sweep_launch(lslam_ctx *ctx, const SweepArgs &a, int jtj_mode, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                        int *variant = nullptr) {
  int v = -1;
  CertPlan plan;
  plan.work = ctx->cert_work.p;
  plan.count = ctx->cert_count.p + (ctx->queue_launches & 1);
  plan.count_next = ctx->cert_count.p + ((ctx->queue_launches + 1) & 1);
  plan.ticket = plan.count + 2;
  plan.ticket_next = plan.count_next + 2;
  plan.ticket2 = plan.count + 4;
  plan.ticket2_next = plan.count_next + 4;
  if (a.grid) {  // the grid sweep: probe + proof for every point, then the tree search for the points it listed; timed as one
    if (a.nb_total <= 0 || a.n_groups <= 0 || !a.need_cnt) return a.nb_total <= 0 ? hipSuccess : hipErrorInvalidValue;
    // A/B (off; lslam_opts.ab_switches & LSLAM_AB_WIDE_IN_PLACE): a map without trees and a launch of at most two wavefronts per
    // SIMD as ONE launch -- each wavefront resolves its own unproven points by the wide probe before the residual chain
    // (sweep_grid_kernel<.., true>).  Measured slower than the five launches below: see include/lslam_c.h.
    if (a.grid == 2 && (long)a.nb_total * (SWEEP_BLOCK / 64) <= 2 * 1024 && (ctx->ab_now & LSLAM_AB_WIDE_IN_PLACE)) {
      hipError_t e1w = launch_sweep_grid(a, jtj_mode, ctx->stream, e0, e1, true);
      ctx->sweep_variants[SWEEP_VARIANT_GRID_WIDE]++;
      if (variant) *variant = SWEEP_VARIANT_GRID_WIDE;
      return e1w;
    }
    hipError_t e = launch_sweep_grid(a, jtj_mode, ctx->stream, e0, nullptr);
    v = SWEEP_VARIANT_GRID;
    ctx->sweep_variants[v]++;
    if (variant) *variant = v;
    if (e != hipSuccess) return e;
    bool planned = false;
    if (a.grid == 2) {  // no trees: the listed points' neighbours come from the wide probe, the queue only runs their residual chain
      e = launch_sweep_plan(a, ctx->stream, plan, 0, true);  // the second pass's plan and the wide probe's prefix in one launch
      if (e != hipSuccess) return e;
      planned = true;
      e = launch_sweep_wide(a, ctx->stream);
      if (e != hipSuccess) return e;
    }
    const int pass2 = a.stack_ovf ? (a.deep_tree ? SWEEP_VARIANT_DEEP_OVF : SWEEP_VARIANT_SHALLOW) : SWEEP_VARIANT_DEEP;
    if (a.grid == 1 && a.need2_cnt) {  // second probe, then the tree search of what IT could not prove: two plans, two queues
      e = launch_sweep_queue(a, jtj_mode, ctx->stream, nullptr, pass2, plan, 0);
      ctx->queue_launches++;
      if (e != hipSuccess) return e;
      CertPlan plan2;
      plan2.work = ctx->cert_work.p;
      plan2.count = ctx->cert_count.p + (ctx->queue_launches & 1);
      plan2.count_next = ctx->cert_count.p + ((ctx->queue_launches + 1) & 1);
      plan2.ticket = plan2.count + 2;
      plan2.ticket_next = plan2.count_next + 2;
      e = launch_sweep_queue(a, jtj_mode, ctx->stream, e1, pass2, plan2, 1);
      ctx->queue_launches++;
      return e;
    }
#ifdef LSLAM_EXP_NO_PASS2  // TIMING EXPERIMENT ONLY (wrong results): the listed points are dropped -- what the second pass costs
    if (e1) return hipEventRecord(e1, ctx->stream);
    return hipSuccess;
#endif
    // A/B (off; lslam_opts.ab_switches & LSLAM_AB_REFILL sets wide_d / wide_p): whole-map trees, the bounded production sweep of a
    // throughput-bound batch -- the listed points are searched by persistent lanes in a launch of their own, their residual
    // chain runs in the next (sweep_refill_kernel).  Same bits, measured slower
    if (a.grid == 1 && pass2 == SWEEP_VARIANT_SHALLOW && a.bounded && a.wide_d && a.wide_p && !a.flags_out && a.fine_gate_c < 0.0f && !planned) {
      e = launch_sweep_refill(a, jtj_mode, ctx->stream, e1, plan);
      ctx->queue_launches++;
      return e;
    }
    e = launch_sweep_queue(a, jtj_mode, ctx->stream, e1, pass2, plan, 0, planned);
    ctx->queue_launches++;
    return e;
  }
  // certificate sweep: from the second sweep of a loop a second launch searches the points pass 1 listed (sweep_body); the
  // pair is timed as one
  // (n_groups > 0: the two counter pairs of the plan alternate per LAUNCHED plan -- each zeroes the other's -- so a sweep
  // with nothing to plan must not advance them)
  bool cert_launched = false;
  const bool two_pass = a.prev_q && a.need_cnt && a.bounded && a.prev_valid && !a.tail.count && !a.gc.trees && a.n_groups > 0 && a.nb_total > 0;
  hipError_t e = launch_sweep(a, jtj_mode, ctx->stream, e0, two_pass ? nullptr : e1, &v, &cert_launched);
  if (v >= 0 && v < SWEEP_N_VARIANTS) ctx->sweep_variants[v]++;
  if (variant) *variant = v;
  if (e == hipSuccess && two_pass) {
    if (cert_launched) {  // (a build whose launch_sweep never takes the certificate instantiation lists nothing)
      e = launch_sweep_queue(a, jtj_mode, ctx->stream, e1, v, plan);
      ctx->queue_launches++;
    } else if (e1) {
      e = hipEventRecord(e1, ctx->stream);
    }
  }
  return e;
}

int ensure_states(lslam_ctx *ctx, int32_t n) {
  if (n <= ctx->state_cap) return LSLAM_OK;
  if (ctx->d_state) (void)hipFree(ctx->d_state);
  if (ctx->h_state) (void)hipHostFree(ctx->h_state);
  ctx->d_state = nullptr;
  ctx->h_state = nullptr;
  ctx->state_cap = 0;
  HIP_TRY(hipMalloc((void **)&ctx->d_state, sizeof(GNState) * (size_t)n));
  HIP_TRY(hipHostMalloc((void **)&ctx->h_state, sizeof(GNState) * (size_t)n, hipHostMallocDefault));
  ctx->state_cap = n;
  return LSLAM_OK;
}

// Trees deeper than the LDS part of the traversal stack need the global overflow area.
int ensure_stack_ovf(lslam_ctx *ctx, size_t n_threads, uint32_t **out) {
  *out = nullptr;
  if (ctx->tc.depth <= KD_STACK_LDS + 1 && ctx->ts.depth <= KD_STACK_LDS + 1) return LSLAM_OK;
  HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words(n_threads)));
  *out = ctx->stack_ovf.p;
  return LSLAM_OK;
}

int ensure_trees(lslam_ctx *ctx);

// Every entry point starts here.  An entry point that can work on a map whose kd-trees are still deferred says so; every
// other one gets them built first.
int check_ctx(lslam_ctx *ctx, bool trees_may_be_pending = false) {
  if (!ctx) {
    set_err("null ctx");
    return LSLAM_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->trees_pending && !trees_may_be_pending) return ensure_trees(ctx);
  return LSLAM_OK;
}

}  // namespace

namespace lslam {
const EnvOnce &env_once() {
  static const EnvOnce e = [] {
    EnvOnce v;
    auto on = [](const char *n) { return std::getenv(n) != nullptr; };
    auto num = [](const char *n, int dflt) { const char *s = std::getenv(n); return s ? std::atoi(s) : dflt; };
    v.hooks = num("LSLAM_DEBUG_HOOKS", 0) == 1;
    v.debug = on("LSLAM_DEBUG");
    v.unbounded_knn = on("LSLAM_UNBOUNDED_KNN");
    v.no_morton = on("LSLAM_NO_MORTON");
    v.fmap_one_stream = on("LSLAM_FMAP_ONE_STREAM");
    v.host_morton = on("LSLAM_HOST_MORTON");
    v.odom_inline = on("LSLAM_ODOM_INLINE_SEARCH");
    v.odom_trees = on("LSLAM_ODOM_TREES");
    v.gnp_coop = on("LSLAM_GNP_COOPERATIVE");
    v.tiny_phase_off = on("LSLAM_TINY_PHASE") && num("LSLAM_TINY_PHASE", 1) == 0;
    v.no_reg_nodes = on("LSLAM_NO_REG_NODES");
    v.no_level_build = on("LSLAM_NO_LEVEL_BUILD");
    v.fmap_timing = on("LSLAM_FMAP_TIMING");
    v.fmap_measured_extents = on("LSLAM_FMAP_MEASURED_EXTENTS");
    v.small_sort = on("LSLAM_SMALL_SORT");
    v.grid_one_stream = on("LSLAM_GRID_ONE_STREAM");
    if (on("LSLAM_FX_HELPERS")) v.fx_helpers = std::max(0, std::min(7, (int)num("LSLAM_FX_HELPERS", 3)));
    return v;
  }();
  return e;
}
const char *debug_env(const char *name) { return env_once().hooks ? std::getenv(name) : nullptr; }
}  // namespace lslam

extern "C" {

const char *lslam_last_error(void) { return g_err.c_str(); }

int lslam_abi_version(void) {
#if LSLAM_EXPERIMENT_BUILD
  const char *ok = std::getenv("LSLAM_ALLOW_EXPERIMENT_BUILD");  // (a timing build of tools/build_variant.sh: not a product)
  if (!(ok && ok[0] == '1')) return -LSLAM_ABI_VERSION;
#endif
  return LSLAM_ABI_VERSION;
}
size_t lslam_sizeof_opts(void) { return sizeof(lslam_opts); }
size_t lslam_sizeof_stats(void) { return sizeof(lslam_stats); }

void lslam_default_opts(lslam_opts *o) {
  if (!o) return;
  // EVERY byte: a caller's struct on the stack is garbage until this call, and a field this function forgot (debug_stats and
  // ab_switches were, for half a round: C++ callers ran whichever A/B variants their stack happened to spell) must read 0
  std::memset(o, 0, sizeof(*o));
  o->max_iterations = 10;  // ScanMatch.h:36
  o->delta_t_abort = 0.05f;
  o->delta_r_abort = 0.05f;  // ScanMatch.cpp:22
  o->use_score = 1;          // :23
  o->fine_score = 0;         // :32
  o->score_threshold = 800;  // :24
  o->match_percentage_threshold = 0.4;
  o->jtj_mode = 1;  // MFMA J^T J: measured >= the VALU path (profiles/), same sums to 1e-5
  o->profile = 0;
  o->scans_in_flight = 0;
  o->search_mode = LSLAM_SEARCH_AUTO;
  o->knn_cert = 1;
  o->cert_try_m = CERT_TRY_M_DEFAULT;
  o->cert_track_m = CERT_TRACK_M_DEFAULT;
  o->grid_cell = 0.0f;  // GRID_CELL_DEFAULT
  o->debug_stats = 0;
  o->ab_switches = 0;
}

int lslam_ctx_create(int device, lslam_ctx **out) {
  if (!out) {
    set_err("null out");
    return LSLAM_ERR_INVALID;
  }
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_err("no HIP device available (%s); this backend has no CPU fallback",
            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return LSLAM_ERR_HIP;
  }
  if (device < 0 || device >= count) {
    set_err("device %d out of range [0,%d)", device, count);
    return LSLAM_ERR_INVALID;
  }
  HIP_TRY(hipSetDevice(device));
  lslam_ctx *ctx = new lslam_ctx();
  ctx->device = device;
  HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  // (stream2 is made when something first forks onto it -- ctx_stream2: a context that only registers sweeps or runs the odometry
  // node never does, and every stream is one more claimant of the device's few hardware queues)
  HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  {
    int rc = ensure_states(ctx, 1);
    if (rc) return rc;
  }
  HIP_TRY(hipEventCreate(&ctx->ev0));
  HIP_TRY(hipEventCreate(&ctx->ev1));
  // Environment overrides of lslam_opts fields (A/B runs of a binary one cannot pass options to): read here, once per
  // context -- never inside a call, where another thread's setenv would race with it.  (The process-wide switches: env_once.)
  (void)lslam::env_once();
  if (const char *v = std::getenv("LSLAM_KNN_CERT")) ctx->env_knn_cert = std::atoi(v);
  if (const char *v = std::getenv("LSLAM_CERT_TRY_M")) ctx->env_cert_try_m = (float)std::atof(v);
  if (const char *v = std::getenv("LSLAM_CERT_TRACK_M")) ctx->env_cert_track_m = (float)std::atof(v);
  if (const char *v = std::getenv("LSLAM_GRID_CELL")) ctx->env_grid_cell = (float)std::atof(v);
  if (const char *v = std::getenv("LSLAM_GRID_CELL_CORNER")) ctx->env_grid_cell_corner = (float)std::atof(v);
  if (const char *v = std::getenv("LSLAM_FIT_FROM_SWEEP")) ctx->env_fit_from_sweep = std::atoi(v);
  if (const char *v = std::getenv("LSLAM_DEBUG_CERT_STATS")) ctx->env_debug_cert_stats = std::atoi(v);
  if (const char *v = std::getenv("LSLAM_FORCE_STACK")) {
    if (!std::strcmp(v, "deep")) ctx->env_force_stack = SWEEP_STACK_DEEP;
    else if (!std::strcmp(v, "shallow")) ctx->env_force_stack = SWEEP_STACK_SHALLOW;
    else if (!std::strcmp(v, "auto")) ctx->env_force_stack = SWEEP_STACK_AUTO;
  }
  if (const char *v = std::getenv("LSLAM_PERSISTENT_GN")) ctx->env_ab |= std::atoi(v) == 1 ? LSLAM_AB_PERSISTENT_GN : 0;
  if (const char *v = std::getenv("LSLAM_FUSED_SOLVE")) ctx->env_ab |= std::atoi(v) == 1 ? LSLAM_AB_FUSED_SOLVE : 0;
  if (const char *v = std::getenv("LSLAM_AB_SWITCHES")) ctx->env_ab |= std::atoi(v);  // any LSLAM_AB_* bits, for A/B runs of unmodified programs
  ctx->env_debug = lslam::env_once().debug;
  if (const char *v = std::getenv("LSLAM_SEARCH")) {
    if (!std::strcmp(v, "lane")) ctx->env_search = LSLAM_SEARCH_LANE;
    else if (!std::strcmp(v, "packet")) ctx->env_search = LSLAM_SEARCH_PACKET;
    else if (!std::strcmp(v, "grid")) ctx->env_search = LSLAM_SEARCH_GRID;
  }
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.insert(ctx);
  }
  *out = ctx;
  return LSLAM_OK;
}

void lslam_ctx_destroy(lslam_ctx *ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.erase(ctx);
  }
  ctx->worker.stop();
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  lslam::odom_ctx_gone(ctx);  // the hidden odometry node of lslam_odometry_match (lslam_odom.hip)
  ctx->tc.nodes.release(); ctx->tc.pts.release(); ctx->tc.pn.release(); ctx->tc.own_box.release();
  ctx->ts.nodes.release(); ctx->ts.pts.release(); ctx->ts.pn.release(); ctx->ts.own_box.release();
  ctx->cell_c.release(); ctx->cell_s.release(); ctx->views_c.release(); ctx->views_s.release();
  ctx->prev_nb.release();
  ctx->prev_q.release();
  ctx->prev_lb.release();
  ctx->need_list.release();
  ctx->need_cnt.release();
  ctx->need2_list.release();
  ctx->need2_cnt.release();
  ctx->groups.release();
  ctx->scan_tables.release();
  ctx->cert_work.release();
  ctx->cert_count.release();
  ctx->kc.release();
  ctx->ks.release();
  ctx->bbox6.release(); ctx->wide_p.release(); ctx->wide_d.release(); ctx->wide_off.release();
  ctx->xchg.release();
  ctx->active_blocks.release(); ctx->d_active_cnt.release(); ctx->fit_ids.release(); ctx->fit_val.release();
  ctx->cert_stats.release();
  ctx->gnp_slots.release(); ctx->gnp_bar.release();
  ctx->q.release(); ctx->blocks.release(); ctx->probs.release(); ctx->partials.release(); ctx->stack_ovf.release();
  ctx->t_idx.release(); ctx->t_d2.release(); ctx->t_coeff.release(); ctx->t_flags.release();
  ctx->t_q.release(); ctx->t_small.release();
  if (ctx->d_state) (void)hipFree(ctx->d_state);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  for (int k = 0; k < 2; ++k)
    if (ctx->h_map_stage[k]) (void)hipHostFree(ctx->h_map_stage[k]);
  if (ctx->h_state) (void)hipHostFree(ctx->h_state);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (hipEvent_t e : ctx->sweep_ev) (void)hipEventDestroy(e);
  scanprep_destroy(ctx->scanprep);
  treebuild_release_scratch(ctx->stream);
  if (ctx->stream2) treebuild_release_scratch(ctx->stream2);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->h_active) (void)hipHostFree(ctx->h_active);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

void *lslam_stream(lslam_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

void lslam_debug_cert_stats(lslam_ctx *ctx, uint64_t out[3]) {  // out[0], out[1] need lslam_opts.debug_stats (or LSLAM_DEBUG_CERT_STATS=1 when the context was made) during the runs
  out[0] = out[1] = out[2] = 0;
  if (!ctx) return;
  if (ctx->cert_stats.p && hipMemcpy(out, ctx->cert_stats.p, 16, hipMemcpyDeviceToHost) != hipSuccess) out[0] = out[1] = 0;
  out[2] = ctx->queue_launches;
}

// the grid sweep's share of listed points by feature type and sweep of the loop (include/lslam_c.h)
void lslam_debug_grid_stats(lslam_ctx *ctx, uint64_t out[32]) {
  static_assert(2 * 2 * GRID_STATS_SWEEPS == 32, "lslam_debug_grid_stats' out[32]");
  for (int i = 0; i < 32; ++i) out[i] = 0;
  if (!ctx || !ctx->cert_stats.p) return;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
#ifdef LSLAM_EXP_COUNT_NOHINT  // (experiment build: the raw words from 0, the experiment's counters are words 2 and 3)
      hipMemcpy(out, ctx->cert_stats.p, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess)
#else
      hipMemcpy(out, ctx->cert_stats.p + CERT_STATS_BY_SWEEP, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess)
#endif
    for (int i = 0; i < 32; ++i) out[i] = 0;
}

// what the certificate sweep carries per resident scan point (resident order: per scan its corner points, then its surf points)
int lslam_debug_cert_state(lslam_ctx *ctx, float *q_xyz0, float *lb, size_t cap_points) {
  if (!ctx || !ctx->prev_q.p || !ctx->prev_lb.p) return LSLAM_ERR_INVALID;
  const size_t n = std::min(cap_points, ctx->n_points);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return LSLAM_ERR_HIP;
  if (n && q_xyz0 && hipMemcpy(q_xyz0, ctx->prev_q.p, n * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) return LSLAM_ERR_HIP;
  if (n && lb && hipMemcpy(lb, ctx->prev_lb.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return LSLAM_ERR_HIP;
  return (int)n;
}

int lslam_map_defer_trees(lslam_ctx *ctx, int32_t on) {
  int rc = check_ctx(ctx, true);
  if (rc) return rc;
  ctx->defer_trees = on != 0;
  return LSLAM_OK;
}

void lslam_debug_lazy_trees(lslam_ctx *ctx, uint64_t out[3]) {
  out[0] = ctx ? ctx->lazy_sets : 0;
  out[1] = ctx ? ctx->lazy_builds : 0;
  out[2] = ctx ? (ctx->trees_pending ? 1 : 0) : 0;
}

uint64_t lslam_debug_grid_launches(lslam_ctx *ctx) { return ctx ? ctx->sweep_variants[SWEEP_VARIANT_GRID] + ctx->sweep_variants[SWEEP_VARIANT_GRID_WIDE] : 0; }
uint64_t lslam_debug_grid_wide_launches(lslam_ctx *ctx) { return ctx ? ctx->sweep_variants[SWEEP_VARIANT_GRID_WIDE] : 0; }
void lslam_debug_grid_cells(lslam_ctx *ctx, uint64_t out[2]) {
  out[0] = ctx && ctx->kc.view.cell_start ? (uint64_t)ctx->kc.n_cells : 0;
  out[1] = ctx && ctx->ks.view.cell_start ? (uint64_t)ctx->ks.n_cells : 0;
}

void lslam_debug_sweep_launches(lslam_ctx *ctx, uint64_t counts[8]) {
  for (int i = 0; i < 8; ++i) counts[i] = ctx ? ctx->sweep_variants[i] : 0;
}

int lslam_ctx_set_comm(lslam_ctx *ctx, lslam_comm *comm) {
  if (!ctx) return LSLAM_ERR_INVALID;
  ctx->comm = comm;
  return LSLAM_OK;
}

// Parity tap: download a resident tree (inner nodes as 4 words each, permuted points).
int lslam_debug_tree_dump(lslam_ctx *ctx, int which, uint32_t *nodes_out, size_t node_cap,
                          float *pts_out, size_t pts_cap, uint32_t *root_ref, int32_t *n_nodes) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (!ctx->have_map) return LSLAM_ERR_NO_MAP;
  const DevTree &dt = which ? ctx->ts : ctx->tc;
  if ((size_t)dt.view.n_nodes > node_cap || (size_t)dt.view.n_pts > pts_cap) return LSLAM_ERR_INVALID;
  if (dt.view.n_nodes)
    HIP_TRY(hipMemcpyAsync(nodes_out, dt.nodes.p, (size_t)dt.view.n_nodes * sizeof(KdNode),
                           hipMemcpyDeviceToHost, ctx->stream));
  if (dt.view.n_pts)
    HIP_TRY(hipMemcpyAsync(pts_out, dt.pts.p, (size_t)dt.view.n_pts * sizeof(float4), hipMemcpyDeviceToHost,
                           ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *root_ref = dt.view.root_ref;
  *n_nodes = dt.view.n_nodes;
  return LSLAM_OK;
}

// Profiling tap: phase stamps of the last solve kernel (100 MHz ticks).
void lslam_debug_solve_clocks(lslam_ctx *ctx, uint64_t out[8]) {
  for (int i = 0; i < 8; ++i) out[i] = ctx->h_state->clk[i];
}

// Profiling tap (not part of the drop-in surface): one sweep at `pose` with per-wave
// shader-clock stamps {start, after kNN, after fit, end}; out[n_waves*4], returns n_waves.
int lslam_debug_sweep_clocks(lslam_ctx *ctx, const float pose[6], int32_t jtj_mode,
                             uint64_t *out, size_t out_cap_waves) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (!ctx->have_map || !ctx->have_scan || ctx->n_prob != 1) return LSLAM_ERR_NO_MAP;
  init_state(*ctx->h_state, pose);
  HIP_TRY(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(GNState), hipMemcpyHostToDevice, ctx->stream));
  SweepArgs sa;
  fill_sweep_args(ctx, sa);
  rc = ensure_stack_ovf(ctx, (size_t)sa.nb_total * SWEEP_BLOCK, &sa.stack_ovf);
  if (rc) return rc;
  const size_t nw = (size_t)sa.nb_total * (SWEEP_BLOCK / 64);
  if (nw > out_cap_waves) return LSLAM_ERR_INVALID;
  uint64_t *d = nullptr;
  size_t words = nw * 4;
#ifdef LSLAM_TRAVERSAL_STATS
  words += (size_t)sa.nb_total * SWEEP_BLOCK * 8;  // per-lane traversal statistics follow
  if (nw * 4 + (size_t)sa.nb_total * SWEEP_BLOCK * 8 > out_cap_waves * 4) return LSLAM_ERR_INVALID;
#endif
  HIP_TRY(hipMalloc((void **)&d, words * sizeof(uint64_t)));
  HIP_TRY(hipMemsetAsync(d, 0, words * sizeof(uint64_t), ctx->stream));
  sa.dbg = d;
  const bool unbounded_dbg = env_once().unbounded_knn;
  sa.bounded = unbounded_dbg ? 0 : 1;
  sa.prev_valid = ctx->prev_valid ? 1 : 0;
  if (sa.bounded) ctx->prev_valid = true;
  HIP_TRY(sweep_launch(ctx, sa, jtj_mode));
  HIP_TRY(hipMemcpyAsync(out, d, words * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  (void)hipFree(d);
  return (int)nw;
}

}  // extern "C"

namespace {
// Body of lslam_map_set.  dev_corner/dev_surf non-null: the clouds are already in HBM as
// float4 {x, y, z, bitcast(index)} (map maintenance hands the surround over without a host hop);
// corner/surf are then ignored.  There is one builder, the device one (lslam_treebuild.hip): when it
// reports a structure limit the call FAILS with the reason -- nothing is rebuilt elsewhere.
//   1  node-slot array too small: retried here with 8n/3 and 8n slots (8n always fits: a group of
//      eight slots holds at least one inner node and a tree of n points has fewer than n of them)
//   2  watchdog of the persistent phase-A kernel (an idle workgroup saw unfinished nodes for ~1 s):
//      cannot happen by construction -- a workgroup that holds a node never waits for another one --
//      it only bounds a hang if that reasoning were ever broken; reported as LSLAM_ERR_TREE_BUILD
//   3  more levels of >1536-point nodes than the level driver / node queue is sized for: the tree is
//      deeper than the 64 levels the device traversal stack holds, i.e. LSLAM_ERR_TREE_DEPTH anyway
//   4  more than 64 pending siblings on one path of a wavefront-local subtree: likewise too deep
int tree_build_failed(int limit, size_t n_points) {
  if (limit == 3 || limit == 4) {
    set_err("kd-tree of %zu points is deeper than the device traversal stack (%d levels)", n_points, KD_STACK_MAX);
    return LSLAM_ERR_TREE_DEPTH;
  }
  set_err("device kd-tree build of %zu points hit structure limit %d (1 node slots, 2 queue watchdog)", n_points, limit);
  return LSLAM_ERR_TREE_BUILD;
}

int map_set_impl(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf, size_t n_surf,
                 size_t stride_bytes, const float4 *dev_corner, const float4 *dev_surf, bool may_defer = true,
                 const float (*box_lo)[3] = nullptr, const float (*box_hi)[3] = nullptr) {
  int rc = check_ctx(ctx, true);  // (a map whose trees were never needed is simply replaced)
  if (rc) return rc;
  ctx->trees_pending = false;
  const bool from_dev = dev_corner != nullptr || dev_surf != nullptr;
  if (!from_dev && (stride_bytes < 12 || (stride_bytes & 3) || (n_corner && !corner) || (n_surf && !surf))) {
    set_err("bad cloud arguments (stride %zu)", stride_bytes);
    return LSLAM_ERR_INVALID;
  }
  if (n_corner >= KD_MAX_POINTS || n_surf >= KD_MAX_POINTS) {
    set_err("map too large for 27-bit leaf references");
    return LSLAM_ERR_INVALID;
  }
  ctx->have_map = false;
  ctx->map_epoch++;
  ctx->cube_mode = false;
  ctx->prev_valid = false;
  ctx->grid_state_valid = false;
  const double t0 = now_ms();
  double t1 = t0, t2 = t0;
  size_t nodes_c = 0, nodes_s = 0;
  int attempts_used = 1;
  // ---- deferred trees (lslam_map_defer_trees): a map that is already in HBM gets its cell grids now -- bounding box, key
  // sort, cell table: a third of a tree build -- and its kd-trees when something needs them (ensure_trees).  Only for maps the
  // grid can take and the scan-match guard accepts; anything else is built eagerly below.
  // {x, y, z, bitcast(index)} of a host cloud packed straight into pinned memory, a chunk at a time, each chunk's DMA running
  // while the next one is packed
  auto upload_host = [&](int k, const void *host, size_t n, float4 *dst, hipStream_t st) -> hipError_t {
    if (n > ctx->h_map_cap[k]) {
      if (ctx->h_map_stage[k]) (void)hipHostFree(ctx->h_map_stage[k]);
      ctx->h_map_stage[k] = nullptr;
      ctx->h_map_cap[k] = 0;
      const size_t want = n + n / 4 + 1024;
      hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_map_stage[k]), want * sizeof(float4), hipHostMallocDefault);
      if (e != hipSuccess) return e;
      ctx->h_map_cap[k] = want;
    }
    float4 *stage = ctx->h_map_stage[k];
    const char *sp = static_cast<const char *>(host);
    constexpr size_t CHUNK = 1u << 17;
    for (size_t off = 0; off < n; off += CHUNK) {
      const size_t end = std::min(n, off + CHUNK);
      for (size_t i = off; i < end; ++i) {
        float xyz[3];
        std::memcpy(xyz, sp + i * stride_bytes, sizeof(xyz));
        stage[i] = make_float4(xyz[0], xyz[1], xyz[2], __builtin_bit_cast(float, (uint32_t)i));
      }
      hipError_t e = hipMemcpyAsync(dst + off, stage + off, (end - off) * sizeof(float4), hipMemcpyHostToDevice, st);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  };
  bool host_uploaded = false;  // the host clouds are in tc.pts / ts.pts already (a deferred attempt that the grid refused)
  if (ctx->defer_trees && may_defer && n_corner >= 50 && n_surf >= 100) {
    const float cell = ctx->env_grid_cell > 0.0f ? ctx->env_grid_cell : (ctx->grid_cell > 0.0f ? ctx->grid_cell : GRID_CELL_DEFAULT);
    HIP_TRY(ctx->bbox6.reserve(12));
    GridDev *gd[2] = {&ctx->kc, &ctx->ks};
    DevTree *trees[2] = {&ctx->tc, &ctx->ts};
    const float4 *dev_src[2] = {dev_corner, dev_surf};
    const int counts[2] = {(int)n_corner, (int)n_surf};
    if (!from_dev) {  // a host map (ScanMatch::scanMatchScan's clouds): uploaded once, then as a device map
      const void *host_src[2] = {corner, surf};
      for (int k = 0; k < 2; ++k) {
        HIP_TRY(trees[k]->pts.reserve((size_t)counts[k]));
        HIP_TRY(upload_host(k, host_src[k], (size_t)counts[k], trees[k]->pts.p, ctx->stream));
        dev_src[k] = trees[k]->pts.p;
      }
      host_uploaded = true;
    }
    float lo[2][3], hi[2][3];
    if (box_lo && box_hi) {  // (lslam_fmap_surround_to_map: the gather took the boxes along)
      std::memcpy(lo, box_lo, sizeof(lo));
      std::memcpy(hi, box_hi, sizeof(hi));
    } else {
      HIP_TRY(grid_bbox2(dev_src, counts, ctx->bbox6.p, lo, hi, ctx->stream));  // the one host round trip of the map set
    }
    // The two grids are independent chains of short launches (count, scan, scatter, rank: 55 us for the corner map, 100 for
    // the surf map of a mapping frame): the corner map's runs on the second stream beside the surf map's, forked and joined by
    // events -- nothing waits on the host.
    int st = 0, st2[2] = {0, 0};
    const bool two_streams = !env_once().grid_one_stream;
    if (two_streams && !ctx_stream2(ctx)) return LSLAM_ERR_HIP;
    hipStream_t s0 = two_streams ? ctx->stream2 : ctx->stream;
    if (two_streams) {
      HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
      HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
    {
      const hipError_t e0 = gd[0]->build(dev_src[0], counts[0], lo[0], hi[0], cell, s0, &st2[0], false);
      const hipError_t e1 = gd[1]->build(dev_src[1], counts[1], lo[1], hi[1], cell, ctx->stream, &st2[1], false);
      if (two_streams) {
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));  // (joined whatever happened: the second stream must not run on into the next call)
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
      }
      HIP_TRY(e0);
      HIP_TRY(e1);
    }
    for (int k = 0; k < 2; ++k) {
      if (!st) st = st2[k];
      if (!st && !gd[k]->view.cell_start) st = 3;
      TreeView v{};
      v.n_pts = (int32_t)counts[k];
      for (int a = 0; a < 3; ++a) {
        v.bb_lo[a] = lo[k][a];
        v.bb_hi[a] = hi[k][a];
      }
      trees[k]->view = v;  // no nodes, no points: whoever needs them goes through ensure_trees
      trees[k]->depth = 0;
    }
    if (!st) {
      ctx->info.n_corner = n_corner;
      ctx->info.n_surf = n_surf;
      ctx->info.nodes_corner = ctx->info.nodes_surf = 0;
      ctx->info.depth_corner = ctx->info.depth_surf = 0;
      ctx->info.build_ms = (float)(now_ms() - t0);
      ctx->info.upload_ms = 0.0f;
      ctx->info.built_on_device = 1;
      ctx->info.build_attempts = 0;  // no tree build attempted yet
      ctx->have_map = true;
      ctx->map_epoch++;
      ctx->grid_epoch = ctx->map_epoch;
      ctx->grid_cell = cell;
      ctx->grid_status = 0;
      ctx->trees_pending = true;
      ctx->lazy_sets++;
      return LSLAM_OK;
    }
    ctx->kc.view = CellGrid{};
    ctx->ks.view = CellGrid{};
    ctx->grid_epoch = ~0ull;
  }
  {
    // ---- device build: upload {x,y,z,index}, build both trees in HBM --------------------
    t1 = now_ms();
    DevTree *trees[2] = {&ctx->tc, &ctx->ts};
    const void *host_src[2] = {corner, surf};
    const float4 *dev_src[2] = {dev_corner, dev_surf};
    const size_t counts[2] = {n_corner, n_surf};
    size_t *ncount[2] = {&nodes_c, &nodes_s};
    // the two trees are independent: build the corner tree on a second stream from a second
    // host thread (the builder synchronises its stream a few times) while this one builds surf
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int fb[2] = {0, 0};
    hipError_t errs[2] = {hipSuccess, hipSuccess};
    auto build_one = [&](int k, hipStream_t st) {
      DevTree &dt = *trees[k];
      const size_t n = counts[k];
      (void)hipSetDevice(ctx->device);
      if ((errs[k] = dt.pts.reserve(n ? n : 1)) != hipSuccess) return;
      // Node slots come in groups of 8 (one cache line per 3-level treelet); how full the
      // groups get depends on the shape of the tree (a cloud of vertical lines leaves most
      // of them nearly empty), so grow the slot array until the build fits: 8n/3, 16n/3, 8n --
      // starting from what fitted this tree last time.  (Slots are taken in order, so a generous array costs a memset and
      // address space, not traffic: the first guess is the one a voxel map's surface-shaped trees need, 43 B per point.)
      int fallback = 0;
      size_t n_leaves = 0;
      for (int attempt = dt.cap_attempt; attempt < 3; ++attempt) {
        const size_t mult[3] = {8, 16, 24};
        size_t cap = ((mult[attempt] * n / 3 + 64) + 7) & ~(size_t)7;
        if (const char *dv = debug_env("LSLAM_DEBUG_NODE_CAP_DIV"))  // tests: force the retry / failure paths
          cap = std::max<size_t>(16, (cap / (size_t)std::max(1, atoi(dv))) & ~(size_t)7);
        if ((errs[k] = dt.nodes.reserve(cap)) != hipSuccess) return;
        if ((errs[k] = dt.own_box.reserve(cap * 6)) != hipSuccess) return;
        if (n && from_dev) {
          errs[k] = hipMemcpyAsync(dt.pts.p, dev_src[k], n * sizeof(float4), hipMemcpyDeviceToDevice, st);
        } else if (n && (attempt > dt.cap_attempt || host_uploaded)) {  // the packed cloud is in the pinned staging already
          errs[k] = hipMemcpyAsync(dt.pts.p, ctx->h_map_stage[k], n * sizeof(float4), hipMemcpyHostToDevice, st);
        } else if (n) {
          errs[k] = upload_host(k, host_src[k], n, dt.pts.p, st);
        }
        if (errs[k] != hipSuccess) return;
        errs[k] = build_kdtree_device(dt.pts.p, (int32_t)n, dt.nodes.p, dt.own_box.p, (int32_t)cap, st, &dt.view, &dt.depth,
                                      &n_leaves, &fallback);
        if (errs[k] != hipSuccess) return;
        if (fallback != 1) {
          if (!fallback) dt.cap_attempt = attempt;
          break;
        }
      }
      fb[k] = fallback;
      *ncount[k] = (size_t)dt.view.n_nodes / 8 * 7 + n_leaves;  // approximate node count
    };
    {
      if (!ctx_stream2(ctx)) return LSLAM_ERR_HIP;
      ctx->worker.submit([&] { build_one(0, ctx->stream2); });
      build_one(1, ctx->stream);
      ctx->worker.wait();
    }
    for (int k = 0; k < 2; ++k) {
      HIP_TRY(errs[k]);
      attempts_used = std::max(attempts_used, trees[k]->cap_attempt + 1);
      if (fb[k]) return tree_build_failed(fb[k], counts[k]);
    }
    t2 = now_ms();
  }
  if (ctx->tc.depth > KD_STACK_MAX || ctx->ts.depth > KD_STACK_MAX) {
    set_err("kd-tree depth %d/%d exceeds device stack %d", ctx->tc.depth, ctx->ts.depth, KD_STACK_MAX);
    return LSLAM_ERR_TREE_DEPTH;
  }
  ctx->info.n_corner = n_corner;
  ctx->info.n_surf = n_surf;
  ctx->info.nodes_corner = (uint32_t)nodes_c;
  ctx->info.nodes_surf = (uint32_t)nodes_s;
  ctx->info.depth_corner = ctx->tc.depth;
  ctx->info.depth_surf = ctx->ts.depth;
  ctx->info.build_ms = (float)(t2 - t1);   // [pack | upload + build]
  ctx->info.upload_ms = (float)(t1 - t0);
  ctx->info.built_on_device = 1;  // there is no other builder
  ctx->info.build_attempts = attempts_used;
  ctx->have_map = true;
  ctx->map_epoch++;
  return LSLAM_OK;
}
}  // namespace

namespace {
// The kd-trees of a map that was set with deferred trees, now: the points come back out of the cell grids in their original
// order, the builder runs as it would have at map-set time.  The grids stay what they are (same points, same indices).
int ensure_trees(lslam_ctx *ctx) {
  if (!ctx->trees_pending) return LSLAM_OK;
  ctx->trees_pending = false;  // (a failure below leaves a map without trees: have_map goes false)
  // a HIP error below leaves a map WITHOUT trees: it must not stay "set" (every later call would search what is not there)
#define ET_TRY(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ctx->have_map = false;                                                                  \
      ctx->map_epoch++;                                                                       \
      set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);     \
      return LSLAM_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)
  const double t0 = now_ms();
  DevTree *trees[2] = {&ctx->tc, &ctx->ts};
  GridDev *gd[2] = {&ctx->kc, &ctx->ks};
  size_t nodes[2] = {0, 0};
  int attempts_used = 1;
  for (int k = 0; k < 2; ++k) {
    DevTree &dt = *trees[k];
    const size_t n = (size_t)gd[k]->view.n_pts;
    ET_TRY(dt.pts.reserve(n ? n : 1));
    int fallback = 0;
    size_t n_leaves = 0;
    for (int attempt = dt.cap_attempt; attempt < 3; ++attempt) {
      const size_t mult[3] = {8, 16, 24};
      const size_t cap = ((mult[attempt] * n / 3 + 64) + 7) & ~(size_t)7;
      ET_TRY(dt.nodes.reserve(cap));
      ET_TRY(dt.own_box.reserve(cap * 6));
      ET_TRY(grid_unsort(gd[k]->view, dt.pts.p, ctx->stream));
      ET_TRY(build_kdtree_device(dt.pts.p, (int32_t)n, dt.nodes.p, dt.own_box.p, (int32_t)cap, ctx->stream, &dt.view, &dt.depth,
                                  &n_leaves, &fallback));
      if (fallback != 1) {
        if (!fallback) dt.cap_attempt = attempt;
        break;
      }
    }
    attempts_used = std::max(attempts_used, dt.cap_attempt + 1);
    if (fallback) {
      ctx->have_map = false;
      ctx->map_epoch++;
      return tree_build_failed(fallback, n);
    }
    nodes[k] = (size_t)dt.view.n_nodes / 8 * 7 + n_leaves;
    dt.view.pn = nullptr;
  }
  if (ctx->tc.depth > KD_STACK_MAX || ctx->ts.depth > KD_STACK_MAX) {
    ctx->have_map = false;
    ctx->map_epoch++;
    set_err("kd-tree depth %d/%d exceeds device stack %d", ctx->tc.depth, ctx->ts.depth, KD_STACK_MAX);
    return LSLAM_ERR_TREE_DEPTH;
  }
  ctx->info.nodes_corner = (uint32_t)nodes[0];
  ctx->info.nodes_surf = (uint32_t)nodes[1];
  ctx->info.depth_corner = ctx->tc.depth;
  ctx->info.depth_surf = ctx->ts.depth;
  ctx->info.build_ms += (float)(now_ms() - t0);
  ctx->info.build_attempts = attempts_used;
  ctx->lazy_builds++;
  return LSLAM_OK;
#undef ET_TRY
}
}  // namespace

namespace {
int build_cube_side_device(lslam_ctx *ctx, DevTree &dt, const float4 *src, bool src_on_device, size_t n_pts,
                           const std::vector<int32_t> &roots_lr, const std::vector<int32_t> &cell_tree, float cube_size,
                           const int32_t origin[3], const int32_t dims[3], DevBuf<int32_t> &cells_d,
                           DevBuf<TreeView> &views_d, CubeGridDev &grid, int *max_depth, size_t *n_nodes, int *fallback);
}  // namespace

namespace lslam {
int map_set_device(lslam_ctx *ctx, const float4 *d_corner, size_t n_corner, const float4 *d_surf, size_t n_surf,
                   const float (*box_lo)[3], const float (*box_hi)[3]) {
  return map_set_impl(ctx, nullptr, n_corner, nullptr, n_surf, sizeof(float4), d_corner ? d_corner : d_surf,
                      d_surf ? d_surf : d_corner, true, box_lo, box_hi);
}
// variant C map straight from device arrays (map maintenance): per side the cubes' clouds back to
// back with .w = index inside the cube, their ranges, and the cell -> tree table
int cubemap_set_device(lslam_ctx *ctx, const float4 *d_corner, size_t nc, const std::vector<int32_t> &roots_c,
                       const std::vector<int32_t> &cells_c, const float4 *d_surf, size_t ns,
                       const std::vector<int32_t> &roots_s, const std::vector<int32_t> &cells_s, float cube_size,
                       const int32_t origin[3], const int32_t dims[3]) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  ctx->have_map = false;
  ctx->map_epoch++;
  ctx->prev_valid = false;
  ctx->grid_state_valid = false;
  const double t0 = now_ms();
  int dc = 0, ds = 0, fb = 0;
  size_t nn_c = 0, nn_s = 0;
  rc = build_cube_side_device(ctx, ctx->tc, d_corner, true, nc, roots_c, cells_c, cube_size, origin, dims, ctx->cell_c,
                              ctx->views_c, ctx->gc, &dc, &nn_c, &fb);
  if (rc) return rc;
  if (!fb)
    rc = build_cube_side_device(ctx, ctx->ts, d_surf, true, ns, roots_s, cells_s, cube_size, origin, dims, ctx->cell_s,
                                ctx->views_s, ctx->gs, &ds, &nn_s, &fb);
  if (rc) return rc;
  if (fb) return tree_build_failed(fb, nc + ns);
  if (dc > KD_STACK_MAX || ds > KD_STACK_MAX) {
    set_err("kd-tree depth %d/%d exceeds device stack %d", dc, ds, KD_STACK_MAX);
    return LSLAM_ERR_TREE_DEPTH;
  }
  ctx->info = lslam_map_info{};
  ctx->info.n_corner = nc;
  ctx->info.n_surf = ns;
  ctx->info.nodes_corner = (uint32_t)nn_c;
  ctx->info.nodes_surf = (uint32_t)nn_s;
  ctx->info.depth_corner = dc;
  ctx->info.depth_surf = ds;
  ctx->info.build_ms = (float)(now_ms() - t0);
  ctx->info.built_on_device = 1;
  ctx->cube_mode = true;
  ctx->have_map = true;
  ctx->map_epoch++;
  return LSLAM_OK;
}
int cubemap_set_views(lslam_ctx *ctx, const std::vector<TreeView> &views_c, const std::vector<int32_t> &cells_c, size_t nc,
                      int depth_c, const std::vector<TreeView> &views_s, const std::vector<int32_t> &cells_s, size_t ns, int depth_s,
                      float cube_size, const int32_t origin[3], const int32_t dims[3]) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  ctx->have_map = false;
  ctx->map_epoch++;
  ctx->prev_valid = false;
  ctx->grid_state_valid = false;
  if (depth_c > KD_STACK_MAX || depth_s > KD_STACK_MAX) {
    set_err("kd-tree depth %d/%d exceeds device stack %d", depth_c, depth_s, KD_STACK_MAX);
    return LSLAM_ERR_TREE_DEPTH;
  }
  const std::vector<TreeView> *views[2] = {&views_c, &views_s};
  const std::vector<int32_t> *cells[2] = {&cells_c, &cells_s};
  DevBuf<TreeView> *vd[2] = {&ctx->views_c, &ctx->views_s};
  DevBuf<int32_t> *cd[2] = {&ctx->cell_c, &ctx->cell_s};
  CubeGridDev *grid[2] = {&ctx->gc, &ctx->gs};
  for (int t = 0; t < 2; ++t) {
    HIP_TRY(vd[t]->reserve(views[t]->size() + 1));
    HIP_TRY(cd[t]->reserve(cells[t]->size() + 1));
    if (!views[t]->empty())
      HIP_TRY(hipMemcpyAsync(vd[t]->p, views[t]->data(), views[t]->size() * sizeof(TreeView), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(cd[t]->p, cells[t]->data(), cells[t]->size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    grid[t]->cube_size = cube_size;
    for (int d = 0; d < 3; ++d) { grid[t]->origin[d] = origin[d]; grid[t]->dims[d] = dims[d]; }
    grid[t]->cell_tree = cd[t]->p;
    grid[t]->trees = vd[t]->p;
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));  // the tables are the caller's locals
  ctx->tc.view = TreeView{};
  ctx->ts.view = TreeView{};
  ctx->tc.depth = depth_c;
  ctx->ts.depth = depth_s;
  ctx->info = lslam_map_info{};
  ctx->info.n_corner = nc;
  ctx->info.n_surf = ns;
  ctx->info.nodes_corner = (uint32_t)views_c.size();
  ctx->info.nodes_surf = (uint32_t)views_s.size();
  ctx->info.depth_corner = depth_c;
  ctx->info.depth_surf = depth_s;
  ctx->info.built_on_device = 1;
  ctx->info.build_attempts = 1;
  ctx->cube_mode = true;
  ctx->have_map = true;
  ctx->map_epoch++;
  return LSLAM_OK;
}
void cubemap_drop_views(lslam_ctx *ctx) {
  if (ctx->cube_mode) {
    ctx->have_map = false;
    ctx->map_epoch++;
    ctx->cube_mode = false;
  }
}
void set_error(const char *msg) { set_err("%s", msg); }
hipStream_t ctx_stream(lslam_ctx *ctx) { return ctx->stream; }
hipStream_t ctx_stream2(lslam_ctx *ctx) {  // the context's second stream, made on first use (nullptr + the error text on failure)
  if (!ctx->stream2) {
    (void)hipSetDevice(ctx->device);
    const hipError_t e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    if (e != hipSuccess) {
      ctx->stream2 = nullptr;
      set_err("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
    }
  }
  return ctx->stream2;
}
TreeView ctx_tree_view(lslam_ctx *ctx, int which) { return which ? ctx->ts.view : ctx->tc.view; }
void ctx_invalidate_map(lslam_ctx *ctx) { ctx->have_map = false; ctx->map_epoch++; ctx->have_scan = false; }
int ctx_scratch(lslam_ctx *ctx, size_t n_float4, size_t n_double, float4 **pts, double **dbl) {
  HIP_TRY(ctx->t_q.reserve(n_float4 ? n_float4 : 1));
  HIP_TRY(ctx->xchg.reserve(n_double ? n_double : 1));
  *pts = ctx->t_q.p;
  *dbl = ctx->xchg.p;
  return LSLAM_OK;
}
int ctx_stack_ovf_if_deep(lslam_ctx *ctx, size_t n_threads, uint32_t **out) { return ensure_stack_ovf(ctx, n_threads, out); }
int ctx_device(const lslam_ctx *ctx) { return ctx ? ctx->device : -1; }
bool ctx_alive(const lslam_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return g_live.count(ctx) != 0;
}
}  // namespace lslam

extern "C" {

int lslam_map_set(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf, size_t n_surf,
                  size_t stride_bytes) {
  return map_set_impl(ctx, corner, n_corner, surf, n_surf, stride_bytes, nullptr, nullptr);
}

namespace {

// One side (corner or surf) of a cube map built on the device: `src` holds the cubes' clouds back to
// back (cube after cube, .w = index inside the cube's cloud), roots_lr the range of every cube that
// gets a tree, cell_tree[cell] = index into roots_lr or -1.  All trees share dt.nodes / dt.pts.
// *fallback != 0: the device build gave up (caller uses the host builder).
int build_cube_side_device(lslam_ctx *ctx, DevTree &dt, const float4 *src, bool src_on_device, size_t n_pts,
                           const std::vector<int32_t> &roots_lr, const std::vector<int32_t> &cell_tree, float cube_size,
                           const int32_t origin[3], const int32_t dims[3], DevBuf<int32_t> &cells_d,
                           DevBuf<TreeView> &views_d, CubeGridDev &grid, int *max_depth, size_t *n_nodes, int *fallback) {
  const int T = (int)(roots_lr.size() / 2);
  *fallback = 0;
  *max_depth = 0;
  *n_nodes = 0;
  HIP_TRY(dt.pts.reserve(n_pts + 16));
  HIP_TRY(cells_d.reserve(cell_tree.size()));
  HIP_TRY(views_d.reserve((size_t)T + 1));
  std::vector<TreeView> views((size_t)T);
  size_t n_leaves = 0;
  for (int attempt = 0; attempt < 3 && T > 0; ++attempt) {
    const size_t mult[3] = {2, 8, 24};
    const size_t cap = ((mult[attempt] * n_pts / 3 + 64 + 8 * (size_t)T) + 7) & ~(size_t)7;
    HIP_TRY(dt.nodes.reserve(cap));
    if (n_pts)
      HIP_TRY(hipMemcpyAsync(dt.pts.p, src, n_pts * sizeof(float4), src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                             ctx->stream));
    HIP_TRY(build_kdforest_device(dt.pts.p, (int32_t)n_pts, roots_lr.data(), T, dt.nodes.p, nullptr, (int32_t)cap, ctx->stream,
                                  views.data(), max_depth, &n_leaves, fallback));
    if (*fallback != 1) break;
  }
  if (*fallback) return LSLAM_OK;
  HIP_TRY(hipMemcpyAsync(cells_d.p, cell_tree.data(), cell_tree.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  if (T) HIP_TRY(hipMemcpyAsync(views_d.p, views.data(), (size_t)T * sizeof(TreeView), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  grid.cube_size = cube_size;
  for (int d = 0; d < 3; ++d) { grid.origin[d] = origin[d]; grid.dims[d] = dims[d]; }
  grid.cell_tree = cells_d.p;
  grid.trees = views_d.p;
  dt.depth = *max_depth;
  dt.view = TreeView{};
  *n_nodes = T ? (size_t)views[0].n_nodes / 8 * 7 + n_leaves : 0;
  return LSLAM_OK;
}

// Partition one cloud into cubes (pushCornerPoint / pushSurfPoint, util/FeatureMap.h:188-204:
// input order is kept inside a cube), build one tree per cube with >= 5 points and upload
// everything into shared node / point arrays.
int build_cube_trees(lslam_ctx *ctx, const void *cloud, size_t n, size_t stride_bytes, float cube_size,
                     const int32_t origin[3], const int32_t dims[3], DevTree &dt, DevBuf<int32_t> &cells_d,
                     DevBuf<TreeView> &views_d, CubeGridDev &grid, int *max_depth, size_t *n_nodes) {
  std::vector<float4> pts;
  pack_cloud(cloud, n, stride_bytes, pts);
  const size_t n_cells = (size_t)dims[0] * dims[1] * dims[2];
  std::vector<int32_t> cell_of(n), count(n_cells, 0);
  for (size_t i = 0; i < n; ++i) {
    const int gi = (int)(std::round(pts[i].x / cube_size) + (float)origin[0]);
    const int gj = (int)(std::round(pts[i].y / cube_size) + (float)origin[1]);
    const int gk = (int)(std::round(pts[i].z / cube_size) + (float)origin[2]);
    const bool ok = 0 <= gi && gi < dims[0] && 0 <= gj && gj < dims[1] && 0 <= gk && gk < dims[2];
    cell_of[i] = ok ? gi + gj * dims[0] + gk * dims[0] * dims[1] : -1;
    if (ok) count[cell_of[i]]++;
  }
  std::vector<size_t> first(n_cells + 1, 0);
  for (size_t c = 0; c < n_cells; ++c) first[c + 1] = first[c] + (size_t)count[c];
  std::vector<float4> sorted(first[n_cells]);
  std::vector<size_t> fill(first.begin(), first.end() - 1);
  for (size_t i = 0; i < n; ++i)
    if (cell_of[i] >= 0) sorted[fill[cell_of[i]]++] = pts[i];
  // all cube trees at once on the device
  std::vector<int32_t> roots_lr, cell_tree_dev(n_cells, -1);
  for (size_t c = 0; c < n_cells; ++c) {
    if (count[c] < 5) continue;  // FeatureMap.h:524,546
    cell_tree_dev[c] = (int32_t)(roots_lr.size() / 2);
    roots_lr.push_back((int32_t)first[c]);
    roots_lr.push_back((int32_t)first[c + 1]);
    for (int32_t k = 0; k < count[c]; ++k) sorted[first[c] + (size_t)k].w = __builtin_bit_cast(float, k);
  }
  int fallback = 0;
  int rc = build_cube_side_device(ctx, dt, sorted.data(), false, sorted.size(), roots_lr, cell_tree_dev, cube_size, origin,
                                  dims, cells_d, views_d, grid, max_depth, n_nodes, &fallback);
  if (rc) return rc;
  if (fallback) return tree_build_failed(fallback, sorted.size());
  ctx->cube_sides_on_device++;
  return LSLAM_OK;
}

}  // namespace

int lslam_cubemap_set(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf, size_t n_surf,
                      size_t stride_bytes, float cube_size, const int32_t origin[3], const int32_t dims[3]) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (stride_bytes < 12 || (stride_bytes & 3) || (n_corner && !corner) || (n_surf && !surf) || !origin || !dims ||
      !(cube_size > 0) || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0 ||
      (size_t)dims[0] * dims[1] * dims[2] > (1u << 26) || n_corner >= KD_MAX_POINTS || n_surf >= KD_MAX_POINTS) {
    set_err("bad cube map arguments");
    return LSLAM_ERR_INVALID;
  }
  ctx->have_map = false;
  ctx->map_epoch++;
  ctx->prev_valid = false;
  ctx->grid_state_valid = false;
  const double t0 = now_ms();
  int dc = 0, ds = 0;
  size_t nc_nodes = 0, ns_nodes = 0;
  ctx->cube_sides_on_device = 0;
  rc = build_cube_trees(ctx, corner, n_corner, stride_bytes, cube_size, origin, dims, ctx->tc, ctx->cell_c,
                        ctx->views_c, ctx->gc, &dc, &nc_nodes);
  if (rc) return rc;
  rc = build_cube_trees(ctx, surf, n_surf, stride_bytes, cube_size, origin, dims, ctx->ts, ctx->cell_s,
                        ctx->views_s, ctx->gs, &ds, &ns_nodes);
  if (rc) return rc;
  if (dc > KD_STACK_MAX || ds > KD_STACK_MAX) {
    set_err("kd-tree depth %d/%d exceeds device stack %d", dc, ds, KD_STACK_MAX);
    return LSLAM_ERR_TREE_DEPTH;
  }
  ctx->info = lslam_map_info{};
  ctx->info.n_corner = n_corner;
  ctx->info.n_surf = n_surf;
  ctx->info.nodes_corner = (uint32_t)nc_nodes;
  ctx->info.nodes_surf = (uint32_t)ns_nodes;
  ctx->info.depth_corner = dc;
  ctx->info.depth_surf = ds;
  ctx->info.build_ms = (float)(now_ms() - t0);
  ctx->info.built_on_device = ctx->cube_sides_on_device == 2 ? 1 : 0;
  ctx->cube_mode = true;
  ctx->have_map = true;
  ctx->map_epoch++;
  return LSLAM_OK;
}

uint64_t lslam_map_epoch(const lslam_ctx *ctx) {
  if (!ctx) return 0;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (!g_live.count(ctx)) return 0;
  }
  return ctx->have_map ? ctx->map_epoch : 0;
}

int lslam_map_info_get(const lslam_ctx *ctx, lslam_map_info *info) {
  if (!ctx || !info) return LSLAM_ERR_INVALID;
  if (!ctx->have_map) return LSLAM_ERR_NO_MAP;
  *info = ctx->info;
  return LSLAM_OK;
}

int lslam_scan_set_batch(lslam_ctx *ctx, int32_t n_scans, const void *const *corner,
                         const size_t *n_corner, const void *const *surf, const size_t *n_surf,
                         size_t stride_bytes) {
  int rc = check_ctx(ctx, true);
  if (rc) return rc;
  if (n_scans <= 0 || !corner || !n_corner || !surf || !n_surf || stride_bytes < 12 ||
      (stride_bytes & 3)) {
    set_err("bad scan arguments");
    return LSLAM_ERR_INVALID;
  }
  size_t total = 0;
  for (int32_t p = 0; p < n_scans; ++p) {
    if ((n_corner[p] && !corner[p]) || (n_surf[p] && !surf[p])) {
      set_err("null cloud in scan %d", p);
      return LSLAM_ERR_INVALID;
    }
    total += n_corner[p] + n_surf[p];
  }
  if (total > 0x3FFFFFFFu) {
    set_err("batch too large");
    return LSLAM_ERR_INVALID;
  }
  ctx->have_scan = false;
  if (ctx->stage_busy) {  // the previous call's copy out of the pinned staging area may still be in flight
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stage_busy = false;
  }
  // the clouds are packed straight into pinned memory: one pass over the caller's points, and the
  // H2D copy is a real asynchronous DMA instead of a staged pageable copy
  if (total > ctx->h_stage_cap) {
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->h_stage_cap = 0;
    const size_t cap = total + total / 4 + 1024;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stage), cap * sizeof(float4), hipHostMallocDefault));
    ctx->h_stage_cap = cap;
  }
  float4 *all = ctx->h_stage;
  size_t n_all = 0;
  ctx->h_blocks.clear();
  ctx->h_groups.clear();
  ctx->h_prob_group0.clear();
  ctx->h_probs.assign((size_t)n_scans, ProbBlocks{0, 0});
  ctx->nqc.assign((size_t)n_scans, 0);
  ctx->nqs.assign((size_t)n_scans, 0);
  std::vector<float4> tmp;
  // LSLAM_NO_MORTON: caller order (profiling A/B); LSLAM_HOST_MORTON: order on the host (A/B, and the
  // definition the device ordering is tested against)
  const bool no_morton = env_once().no_morton;
  const bool host_morton = env_once().host_morton;
  const bool dev_morton = !no_morton && !host_morton;
  std::vector<int32_t> seg_off;
  int32_t out_base = 0;
  for (int32_t p = 0; p < n_scans; ++p) {
    ctx->h_probs[(size_t)p].first_block = (int32_t)ctx->h_blocks.size();
    for (int type = 0; type < 2; ++type) {
      const void *src = type ? surf[p] : corner[p];
      const size_t cnt = type ? n_surf[p] : n_corner[p];
      const int32_t base = (int32_t)n_all;
      float4 *dst = all + n_all;
      if (host_morton && !no_morton) {
        pack_cloud(src, cnt, stride_bytes, tmp);
        morton_order(tmp);
        std::copy(tmp.begin(), tmp.end(), dst);
      } else {
        const char *sp = static_cast<const char *>(src);
        for (size_t i = 0; i < cnt; ++i) {  // {x, y, z, bitcast(original index)}
          float xyz[3];
          std::memcpy(xyz, sp + i * stride_bytes, sizeof(xyz));
          dst[i] = make_float4(xyz[0], xyz[1], xyz[2], __builtin_bit_cast(float, (uint32_t)i));
        }
      }
      seg_off.push_back(base);
      for (size_t off = 0; off < cnt; off += SWEEP_BLOCK) {
        BlockDesc bd{};
        bd.prob = p;
        bd.first = base + (int32_t)off;
        bd.count = (int32_t)std::min<size_t>(SWEEP_BLOCK, cnt - off);
        bd.is_surf = type;
        bd.out_base = out_base + (type ? (int32_t)n_corner[p] : 0);
        ctx->h_blocks.push_back(bd);
      }
      n_all += cnt;
    }
    ctx->h_probs[(size_t)p].n_blocks = (int32_t)ctx->h_blocks.size() - ctx->h_probs[(size_t)p].first_block;
    {  // groups of the certificate sweep's second pass: runs of one feature type
      ctx->h_prob_group0.push_back((int32_t)ctx->h_groups.size());
      const int32_t b0 = ctx->h_probs[(size_t)p].first_block, b1 = (int32_t)ctx->h_blocks.size();
      for (int32_t b = b0; b < b1;) {
        int32_t e = b + 1;
        while (e < b1 && e - b < CERT_GROUP && ctx->h_blocks[(size_t)e].is_surf == ctx->h_blocks[(size_t)b].is_surf) ++e;
        ctx->h_groups.push_back(GroupDesc{b, e - b, p, 0});
        b = e;
      }
    }
    ctx->nqc[(size_t)p] = (int32_t)n_corner[p];
    ctx->nqs[(size_t)p] = (int32_t)n_surf[p];
    out_base += (int32_t)(n_corner[p] + n_surf[p]);
  }
  seg_off.push_back((int32_t)n_all);
  ctx->h_prob_group0.push_back((int32_t)ctx->h_groups.size());
  const size_t nb = ctx->h_blocks.size();
  HIP_TRY(ctx->q.reserve(total ? total : 1));
  // the scan's three tables -- workgroups, scans, second-pass groups -- in ONE device allocation behind ONE upload (three
  // allocations and three copies were three launches of a mapping frame's scan match)
  const size_t n_groups_tab = ctx->h_groups.empty() ? 1 : ctx->h_groups.size();
  const size_t off_probs = ((nb ? nb : 1) * sizeof(BlockDesc) + 15) & ~(size_t)15;
  const size_t off_groups = (off_probs + (size_t)n_scans * sizeof(ProbBlocks) + 15) & ~(size_t)15;
  const size_t tab_bytes = off_groups + n_groups_tab * sizeof(GroupDesc);
  if (tab_bytes > ctx->scan_tables.cap) {  // the three views point into the allocation reserve() is about to free
    ctx->blocks.release();
    ctx->probs.release();
    ctx->groups.release();
  }
  HIP_TRY(ctx->scan_tables.reserve(tab_bytes));
  ctx->blocks.adopt(reinterpret_cast<BlockDesc *>(ctx->scan_tables.p), nb ? nb : 1);
  ctx->probs.adopt(reinterpret_cast<ProbBlocks *>(ctx->scan_tables.p + off_probs), (size_t)n_scans);
  ctx->groups.adopt(reinterpret_cast<GroupDesc *>(ctx->scan_tables.p + off_groups), n_groups_tab);
  HIP_TRY(ctx->partials.reserve((nb ? nb : 1) * NCOL));
  HIP_TRY(ctx->prev_nb.reserve((total ? total : 1) * 5));
  HIP_TRY(ctx->prev_q.reserve(total ? total : 1));
  HIP_TRY(ctx->prev_lb.reserve(total ? total : 1));
  HIP_TRY(ctx->need_list.reserve((nb ? nb : 1) * SWEEP_BLOCK));
  HIP_TRY(ctx->need_cnt.reserve(nb ? nb : 1));
  HIP_TRY(ctx->need2_list.reserve((nb ? nb : 1) * SWEEP_BLOCK));
  HIP_TRY(ctx->need2_cnt.reserve(nb ? nb : 1));
  HIP_TRY(ctx->cert_work.reserve(nb ? nb : 1));
  if (!ctx->cert_count.p) {
    HIP_TRY(ctx->cert_count.reserve(6));  // items [2], tickets [2], the refill search's tickets [2]
    HIP_TRY(hipMemsetAsync(ctx->cert_count.p, 0, 6 * sizeof(int32_t), ctx->stream));
  }
  HIP_TRY(ctx->tail_count.reserve((size_t)n_scans));  // (zeroed where the fused solve is switched on: run_batch_impl)
  ctx->prev_valid = false;
  ctx->grid_state_valid = false;
  rc = ensure_states(ctx, n_scans);
  if (rc) return rc;
  if (total && dev_morton) {
    if (!ctx->scanprep) ctx->scanprep = scanprep_create();
    HIP_TRY(scanprep_order(ctx->scanprep, ctx->stream, all, total, seg_off.data(), (int)seg_off.size() - 1,
                           ctx->q.p));
  } else if (total) {
    HIP_TRY(hipMemcpyAsync(ctx->q.p, all, total * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  }
  ctx->h_tables.assign(tab_bytes, 0);
  if (nb) std::memcpy(ctx->h_tables.data(), ctx->h_blocks.data(), nb * sizeof(BlockDesc));
  std::memcpy(ctx->h_tables.data() + off_probs, ctx->h_probs.data(), (size_t)n_scans * sizeof(ProbBlocks));
  if (!ctx->h_groups.empty()) std::memcpy(ctx->h_tables.data() + off_groups, ctx->h_groups.data(), ctx->h_groups.size() * sizeof(GroupDesc));
  HIP_TRY(hipMemcpyAsync(ctx->scan_tables.p, ctx->h_tables.data(), tab_bytes, hipMemcpyHostToDevice, ctx->stream));
  // No wait here: everything that uses the scan is ordered behind these copies on the same stream, the block / range tables
  // are pageable (consumed when hipMemcpyAsync returns) and the pinned staging area is guarded by `stage_busy` -- so that
  // lslam_scanmatch_scan reaches the host once per call, when the loop's result comes back.
  ctx->stage_busy = true;
  ctx->n_prob = n_scans;
  ctx->nb_total = (int32_t)nb;
  ctx->n_points = total;
  ctx->have_scan = true;
  return LSLAM_OK;
}

int lslam_scan_set(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf,
                   size_t n_surf, size_t stride_bytes) {
  if ((n_corner && !corner) || (n_surf && !surf)) {
    set_err("bad scan arguments");
    return LSLAM_ERR_INVALID;
  }
  return lslam_scan_set_batch(ctx, 1, &corner, &n_corner, &surf, &n_surf, stride_bytes);
}

namespace {
// Shared body of lslam_scanmatch_run_batch and lslam_scanmatch_run_sharded.  With `fn` set the
// resident scan is this rank's shard of ONE scan's points: every iteration reduces the local
// partials, hands the 32 fp64 sums to `fn` (sum over ranks, in place, on `xchg`) and then
// every rank runs the same solve on the same numbers.
int run_batch_impl(lslam_ctx *ctx, int32_t n_scans, float *poses, const lslam_opts *opts_in,
                   lslam_stats *stats, lslam_allreduce_fn fn, void *user, double *xchg, bool use_comm = false) {
  int rc = check_ctx(ctx, true);
  if (rc) return rc;
  // A map whose kd-trees are deferred (lslam_map_defer_trees) is matched through its cell grids alone when the call is a
  // latency-bound one on the plain loop -- a mapping frame: one scan of a few thousand points; anything else builds the trees now.
  bool lazy = false;
  if (ctx->trees_pending) {
    const lslam_opts *oi = opts_in;
    const int sm = ctx->env_search >= 0 ? ctx->env_search : (oi ? (oi->search_mode & 0xFF) : LSLAM_SEARCH_AUTO);
    lazy = ctx->have_map && ctx->have_scan && !ctx->cube_mode && !fn && !use_comm && n_scans == ctx->n_prob &&
           (sm == LSLAM_SEARCH_AUTO || sm == LSLAM_SEARCH_GRID) && !(oi && oi->fine_score && oi->use_score) &&
           (long)ctx->nb_total * (SWEEP_BLOCK / 64) <= 2 * 1024 && ctx->kc.view.cell_start && ctx->ks.view.cell_start;
    if (!lazy) {
      rc = ensure_trees(ctx);
      if (rc) return rc;
    }
  }
  if (!poses || n_scans <= 0) {
    set_err("null poses");
    return LSLAM_ERR_INVALID;
  }
  lslam_opts o;
  if (opts_in) o = *opts_in; else lslam_default_opts(&o);
  if (stats) std::memset(stats, 0, sizeof(lslam_stats) * (size_t)n_scans);
  auto fail_all = [&](int code) {
    if (stats) for (int32_t p = 0; p < n_scans; ++p) stats[p].status = code;
    return code;
  };
  if (!ctx->have_map) { set_err("no map set"); return fail_all(LSLAM_ERR_NO_MAP); }
  if (!ctx->have_scan) { set_err("no scan set"); return fail_all(LSLAM_ERR_NO_SCAN); }
  if (n_scans != ctx->n_prob) {
    set_err("batch size %d does not match the %d resident scans", n_scans, ctx->n_prob);
    return fail_all(LSLAM_ERR_INVALID);
  }
  // ScanMatch.cpp:57-61 (variant C, FeatureMap::scanMatchScan, has no such guard)
  if (!ctx->cube_mode && (ctx->info.n_corner < 50 || ctx->info.n_surf < 100)) return fail_all(LSLAM_TOO_FEW_REF);
  const int max_it = o.max_iterations < 0 ? 0 : o.max_iterations;
  // Every call starts cold: the neighbour lists a previous call left behind belong to another pose (or
  // another scan) and must not bound this call's first sweep -- a real call on a new scan has none.
  // From the second sweep on the bound comes from the first sweep of THIS loop.
  ctx->prev_valid = false;
  ctx->grid_state_valid = false;

  for (int32_t p = 0; p < n_scans; ++p) {
    init_state(ctx->h_state[p], poses + 6 * p);
    if (max_it == 0) ctx->h_state[p].done = 1;
  }
  HIP_TRY(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(GNState) * (size_t)n_scans,
                         hipMemcpyHostToDevice, ctx->stream));
  const int search = resolve_search_mode(ctx, o.search_mode);  // (may make the packet search's nodes: before the views are copied)
  SweepArgs sa;
  fill_sweep_args(ctx, sa);
  sa.packet = search == LSLAM_SEARCH_PACKET ? 1 : 0;
  sa.stack_mode = resolve_stack_mode(ctx, o.search_mode);
  // the production sweep keeps a shallow stack in LDS: it always gets the overflow area (sized per
  // chunk below; the sharded path has one resident scan)
  const bool sharded = fn != nullptr || use_comm;
  if (sharded) {
    HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words((size_t)sa.nb_total * SWEEP_BLOCK)));
    sa.stack_ovf = ctx->stack_ovf.p;
    if (!xchg) {  // the library's own exchange buffer
      HIP_TRY(ctx->xchg.reserve(NCOL));
      xchg = ctx->xchg.p;
    }
  }
  SolveArgs so{};
  so.states = ctx->d_state;
  so.partials = ctx->partials.p;
  so.probs = ctx->probs.p;
  so.n_prob = n_scans;
  so.reduce_only = 0;
  so.ext_sums = nullptr;
  so.max_iterations = max_it;
  so.delta_r_abort = o.delta_r_abort;
  so.delta_t_abort = o.delta_t_abort;
  so.eig_thresh = 100.0f;  // ScanMatch.cpp:223
  so.min_rows = 50;        // ScanMatch.cpp:142
  so.too_few_continue = 0;
  so.nan_reset = 0;
  // joint LiDAR + stereo system: the stereo blocks' records join the reduction of every iteration
  StereoArgs sta{};
  if (ctx->n_stereo > 0) {
    if (n_scans != 1) {
      set_err("the stereo term needs a single resident scan (%d resident)", n_scans);
      return fail_all(LSLAM_ERR_INVALID);
    }
    sta.landmarks = ctx->st_lm.p;
    sta.obs = ctx->st_obs.p;
    sta.n = ctx->n_stereo;
    sta.cam = ctx->st_cam;
    sta.state = ctx->d_state;
    sta.partials = ctx->st_partials.p;
    so.partials2 = ctx->st_partials.p;
    so.n_blocks2 = stereo_blocks(ctx->n_stereo);
  }

  auto sweep_events = [&](int launch, hipEvent_t *e0, hipEvent_t *e1) -> hipError_t {
    *e0 = *e1 = nullptr;
    if (!o.profile) return hipSuccess;
    while ((int)ctx->sweep_ev.size() < 2 * (launch + 1)) {
      hipEvent_t e;
      hipError_t rc_e = hipEventCreate(&e);
      if (rc_e != hipSuccess) return rc_e;
      ctx->sweep_ev.push_back(e);
    }
    *e0 = ctx->sweep_ev[2 * launch];
    *e1 = ctx->sweep_ev[2 * launch + 1];
    return hipSuccess;
  };
  // The loop is device-resident: sweep/solve pairs are enqueued back to back and a
  // finished loop turns the remaining launches into immediate exits.  To avoid paying
  // for many such exits the first batch is sized from the previous call's iteration
  // count (+1 spare); only if a loop is still running after it does the host look at
  // the states (one round trip) and enqueue two more iterations at a time.
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  const uint64_t grid_launches_before = ctx->sweep_variants[SWEEP_VARIANT_GRID];
  int launched = 0;
  int batch = ctx->iter_hint < 1 ? 1 : ctx->iter_hint;
  double total_points = -1.0;  // sharded: points of the whole scan (sum over ranks)
  auto exchange = [&]() -> int {  // sharded: sum xchg[0..32) over the ranks; ordered on the library's stream
    if (fn) {
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      fn(user, xchg, NCOL);  // returns with the sum visible to this stream
    } else {
      if (comm_allreduce_f64(ctx->comm, xchg, NCOL, ctx->stream) != hipSuccess) return LSLAM_ERR_COMM;
    }
    return LSLAM_OK;
  };
  if (sharded) {
    // xchg[32]: the local point count first (one exchange per call), then the sums per iteration
    const bool unbounded = env_once().unbounded_knn;
    sa.bounded = (ctx->cube_mode || unbounded) ? 0 : 1;
    double cnt[NCOL] = {0};
    cnt[0] = (double)ctx->nqc[0] + (double)ctx->nqs[0];
    HIP_TRY(hipMemcpyAsync(xchg, cnt, sizeof(cnt), hipMemcpyHostToDevice, ctx->stream));
    rc = exchange();
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(cnt, xchg, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    total_points = cnt[0];
    // With the library's communicator the loop is device-resident like the single-GPU one: sweep ->
    // per-rank reduction straight into the exchange buffer -> ncclAllReduce -> replicated solve, `batch`
    // iterations enqueued before the host looks.  Every rank sees the same sums, hence the same `done`
    // flag, hence enqueues the same number of collectives.  A callback transport (fn) needs the host
    // between the two halves of every iteration.
    while (launched < max_it) {
      const int todo = fn ? 1 : std::min(batch, max_it - launched);
      for (int b = 0; b < todo; ++b) {
        sa.prev_valid = (sa.bounded && launched > 0) ? 1 : 0;
        hipEvent_t e0, e1;
        HIP_TRY(sweep_events(launched, &e0, &e1));
        HIP_TRY(sweep_launch(ctx, sa, o.jtj_mode, e0, e1));
        HIP_TRY(launch_stereo(sta, ctx->stream));
        so.reduce_only = 1;
        so.ext_sums = nullptr;
        so.sums_out = xchg;
        HIP_TRY(launch_solve(so, ctx->stream));
        rc = exchange();
        if (rc) return rc;
        so.reduce_only = 0;
        so.ext_sums = xchg;
        so.sums_out = nullptr;
        HIP_TRY(launch_solve(so, ctx->stream));
        ++launched;
      }
      HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (ctx->h_state[0].done) break;  // identical on every rank: same sums, same solve
      batch = 2;
    }
  }
  // ---- device-resident loops -----------------------------------------------------------------------
  // The resident scans are matched `in_flight` at a time (a keyframe re-matching pass holds hundreds
  // of scans, pose_graph/graph.cpp:171-197): every chunk is its own sequence of sweep/solve launches
  // over its block range, so the traversal-stack overflow area and the wavefront count of a launch stay
  // bounded.  All chunks' first `batch` iterations are enqueued back to back before the host looks once.
  int n_launches = 0;
  // One resident scan: the whole loop in ONE persistent launch (gn_persistent_kernel) when every block of the sweep can
  // be resident at once and nothing needs the launches in between (per-launch profiling, the stereo term, the exchange of
  // a sharded run, per-cube trees, the packet search, trees deeper than the LDS stack).  OFF by default
  // (lslam_opts.ab_switches & LSLAM_AB_PERSISTENT_GN turns it on): bit-identical results, but measured no faster than the launch loop -- 327 us
  // against 315 us of device time per four-iteration scanMatchScan of 115 200 points; the two grid exchanges and the
  // replicated solve of an iteration cost what the solve launch and its two gaps do.
  bool gnp_done = false;
  {
    const bool gnp_off = !((o.ab_switches | ctx->env_ab) & LSLAM_AB_PERSISTENT_GN);
    const bool unbounded = env_once().unbounded_knn;
    if (!sharded && !lazy && n_scans == 1 && !gnp_off && ctx->gnp_ok && !o.profile && ctx->n_stereo == 0 && !ctx->cube_mode &&
        !sa.packet && sa.stack_mode != SWEEP_STACK_SHALLOW && max_it > 0 && ctx->tc.depth <= KD_STACK_LDS + 1 && ctx->ts.depth <= KD_STACK_LDS + 1 &&
        sa.nb_total > 0 && sa.nb_total <= 512) {
      if (ctx->gnp_cap < 0) ctx->gnp_cap = gn_persistent_capacity(ctx->device);
      if (sa.nb_total <= ctx->gnp_cap) {
        // [2][nb][32] floats, then [3][32][32] doubles (8-byte aligned: the float part is a multiple of 64 words)
        const size_t n_slot = 2 * (size_t)sa.nb_total * NCOL + 2 * 3 * 32 * NCOL;
        HIP_TRY(ctx->gnp_slots.reserve(n_slot));
        HIP_TRY(ctx->gnp_bar.reserve(2));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ctx->gnp_slots.p, (int)0xFFF8DEADu, n_slot, ctx->stream));
        HIP_TRY(hipMemsetAsync(ctx->gnp_bar.p, 0, 2 * sizeof(unsigned), ctx->stream));
        SweepArgs sp = sa;
        sp.bounded = unbounded ? 0 : 1;
        sp.stack_ovf = nullptr;
        GnLoopArgs gl{};
        gl.slots = ctx->gnp_slots.p;
        gl.gslots = reinterpret_cast<double *>(ctx->gnp_slots.p + 2 * (size_t)sa.nb_total * NCOL);
        gl.bar = ctx->gnp_bar.p;
        gl.state_out = ctx->d_state;
        gl.max_iterations = max_it;
        gl.min_rows = so.min_rows;
        gl.delta_r_abort = so.delta_r_abort;
        gl.delta_t_abort = so.delta_t_abort;
        gl.eig_thresh = so.eig_thresh;
        HIP_TRY(launch_gn_persistent(sp, o.jtj_mode, gl, ctx->stream));
        ctx->sweep_variants[SWEEP_VARIANT_PERSISTENT]++;
        HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
        unsigned gbar[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(gbar, ctx->gnp_bar.p, sizeof(gbar), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (gbar[1] == 0) {
          gnp_done = true;
          ctx->gnp_runs++;
          launched = ctx->h_state[0].sweeps;
        } else {  // an exchange ran into its spin limit (the workgroups were not all resident): the launch loop, from the start
          if (ctx->env_debug) fprintf(stderr, "[lslam] persistent GN kernel timed out in its grid exchange: launch loop from here on\n");
          ctx->gnp_ok = false;
          init_state(ctx->h_state[0], poses);
          HIP_TRY(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(GNState), hipMemcpyHostToDevice, ctx->stream));
        }
      }
    }
  }
  if (!sharded && !gnp_done) {
    const int in_flight = o.scans_in_flight > 0 ? std::min<int>(o.scans_in_flight, n_scans) : std::min<int>(n_scans, 128);
    const int n_chunks = (n_scans + in_flight - 1) / in_flight;
    const bool unbounded = env_once().unbounded_knn;  // A/B switch
    sa.bounded = (ctx->cube_mode || unbounded) ? 0 : 1;  // per-cube positions are tree-relative
    // The grid sweep (LSLAM_SEARCH_GRID): cell grids over the resident trees, made on first use.  A map the grid cannot take
    // (non-finite points, an extent beyond the grid's limits) is searched by the tree as before.
    // AUTO takes it for throughput-bound batches (launch_sweep's own test: more than two wavefronts per SIMD in a launch of
    // the first chunk): a launch that fits the device at once is latency-bound, and the grid sweep is two launches more
    bool want_grid = search == LSLAM_SEARCH_GRID;
    if (!want_grid && ctx->env_search < 0 && (o.search_mode & 0xFF) == LSLAM_SEARCH_AUTO && !ctx->cube_mode && !sa.packet) {
      const int p1 = std::min(n_scans, in_flight);
      const long nb0 = (long)ctx->h_probs[(size_t)p1 - 1].first_block + ctx->h_probs[(size_t)p1 - 1].n_blocks;
      want_grid = nb0 * (SWEEP_BLOCK / 64) > 2 * 1024;
    }
    if (lazy && !sa.bounded) {  // (LSLAM_UNBOUNDED_KNN: an A/B switch of the tree search)
      rc = ensure_trees(ctx);
      if (rc) return rc;
      lazy = false;
    }
    if (lazy) {  // the grids the map was set with; no trees: the listed points go through the wide probe
      sa.kc = ctx->kc.view;
      sa.ks = ctx->ks.view;
      sa.grid = 2;
      sa.grid_clip_margin = GRID_CLIP_MARGIN_MIN;
      sa.wide_nf_slack = ((o.ab_switches | ctx->env_ab) & LSLAM_AB_WIDE_NF_MARGIN) ? GRID_NF_PRUNE_SLACK_WIDE : 0.0f;
      HIP_TRY(ctx->wide_d.reserve(std::max<size_t>(ctx->n_points, 1) * 5));
      HIP_TRY(ctx->wide_p.reserve(std::max<size_t>(ctx->n_points, 1) * 5));
      HIP_TRY(ctx->wide_off.reserve((size_t)std::max(ctx->nb_total, 1) + 2));
      sa.wide_d = ctx->wide_d.p;
      sa.wide_p = ctx->wide_p.p;
      sa.wide_off = ctx->wide_off.p;
    } else if (want_grid && sa.bounded) {
      rc = ensure_grid(ctx, ctx->env_grid_cell > 0.0f ? ctx->env_grid_cell : o.grid_cell);
      if (rc) return rc;
      if (ctx->kc.view.cell_start && ctx->ks.view.cell_start) {
        sa.kc = ctx->kc.view;
        sa.ks = ctx->ks.view;
        sa.grid = 1;
        if ((o.ab_switches | ctx->env_ab) & LSLAM_AB_REFILL) {  // A/B (off): the second pass's two-launch form -- the five of every listed point between its launches
          HIP_TRY(ctx->wide_d.reserve(std::max<size_t>(ctx->n_points, 1) * 5));
          HIP_TRY(ctx->wide_p.reserve(std::max<size_t>(ctx->n_points, 1) * 5));
          sa.wide_d = ctx->wide_d.p;
          sa.wide_p = ctx->wide_p.p;
        }
        sa.grid_clip_margin = GRID_CLIP_MARGIN_MIN;
      }
    }
    ctx->ab_now = o.ab_switches | ctx->env_ab;
    const bool grid_on = sa.grid != 0;
    const bool no_probe2 = ((o.ab_switches | ctx->env_ab) & LSLAM_AB_SECOND_PROBE) == 0;  // A/B (off): a second, wider probe before the tree search
    // neighbour lists carried from sweep to sweep by certificate where a point has hardly moved (sweep_body): lslam_opts.knn_cert
    // = 0 searches every point in every sweep, 2 takes the certificate sweep whatever the size of the launch (tests); the
    // environment's LSLAM_KNN_CERT / LSLAM_CERT_TRY_M / LSLAM_CERT_TRACK_M, read when the context was made, override the options
    const int cert_mode = ctx->env_knn_cert >= 0 ? ctx->env_knn_cert : o.knn_cert;
    const bool no_cert = cert_mode == 0 || sa.grid, force_cert = cert_mode == 2;
    sa.prev_q = (sa.bounded && !no_cert && !sa.packet) ? ctx->prev_q.p : nullptr;
    if (sa.prev_q) {
      sa.prev_lb = ctx->prev_lb.p;
      sa.cert_try_m = ctx->env_cert_try_m >= 0.0f ? ctx->env_cert_try_m : (o.cert_try_m > 0.0f ? o.cert_try_m : CERT_TRY_M_DEFAULT);
      sa.cert_track_m = ctx->env_cert_track_m >= 0.0f ? ctx->env_cert_track_m : (o.cert_track_m > 0.0f ? o.cert_track_m : CERT_TRACK_M_DEFAULT);
    }
    if (grid_on) {  // the grid sweep carries (position, fifth distance) per point in prev_q; no certificates
      sa.prev_q = ctx->prev_q.p;
      sa.prev_lb = nullptr;
      sa.grid_hint = ctx->prev_lb.p;  // (the certificate sweep's per-point array, free in this mode)
      if (sa.grid == 1 && ((o.ab_switches | ctx->env_ab) & LSLAM_AB_FIT_CACHE)) {  // the fit cache (sweep_grid_kernel)
        const size_t nf = std::max<size_t>(ctx->n_points, 1);
        HIP_TRY(ctx->fit_ids.reserve(5 * nf));
        HIP_TRY(ctx->fit_val.reserve(5 * nf));
        sa.fit_ids = ctx->fit_ids.p;
        sa.fit_val = ctx->fit_val.p;
        sa.n_fit = (int32_t)nf;
        sa.fit_from_sweep = ctx->env_fit_from_sweep > 0 ? ctx->env_fit_from_sweep : 3;
      }
    }
    if ((sa.prev_q || sa.grid) && (ctx->env_debug_cert_stats || o.debug_stats)) {
      if (!ctx->cert_stats.p) {
        HIP_TRY(ctx->cert_stats.reserve(CERT_STATS_WORDS));
        HIP_TRY(hipMemsetAsync(ctx->cert_stats.p, 0, CERT_STATS_WORDS * sizeof(unsigned long long), ctx->stream));
      }
      sa.cert_stats = ctx->cert_stats.p;
    }
    std::vector<int> done_iters((size_t)n_chunks, 0);    // iterations enqueued per chunk
    std::vector<char> finished((size_t)n_chunks, 0);
    // lslam_opts.ab_switches & LSLAM_AB_FUSED_SOLVE: the solve rides in the tail of the sweep launch whenever that
    // launch is a latency-bound one (launch_sweep takes the whole-stack kernel: single scans, small batches) -- the block that
    // retires a scan's last record reduces and solves, one launch per Gauss-Newton iteration.  Bit-identical to the solve
    // kernel as its own launch (tests/test_gpu_stack_shapes.py) and MEASURED NO FASTER, so off by default: per working
    // iteration of a 115 200-point scan 83 us against 47 + 13 us of kernels plus a ~3 us gap (rocprofv3 --kernel-trace);
    // 0.270 against 0.237 ms of device time per three-iteration loop, 0.244 against 0.229 ms for a 4 835-point scan.  The
    // tail is the solve's own dependent chain (reduction round trip, 6 x 6 QR, pose update: ~13 us) run by ONE workgroup
    // that first had to finish its share of the sweep, reading the records through the coherence point; the launch it saves
    // costs less than that.  Not with the stereo term (its records come from a launch of their own).
    const bool no_fuse = !((o.ab_switches | ctx->env_ab) & LSLAM_AB_FUSED_SOLVE);
    if (!no_fuse && ctx->n_stereo == 0 && !sa.grid) {
      // the ticket counters: zero between launches (the last block of a launch resets its scan's); once per call here, in case
      // an earlier call ended in an error half way
      HIP_TRY(hipMemsetAsync(ctx->tail_count.p, 0, sizeof(int32_t) * (size_t)n_scans, ctx->stream));
      sa.tail.count = ctx->tail_count.p;
      sa.tail.probs = ctx->probs.p;
      sa.tail.partials_abs = ctx->partials.p;
      sa.tail.sp.max_iterations = so.max_iterations;
      sa.tail.sp.min_rows = so.min_rows;
      sa.tail.sp.too_few_continue = so.too_few_continue;
      sa.tail.sp.nan_reset = so.nan_reset;
      sa.tail.sp.delta_r_abort = so.delta_r_abort;
      sa.tail.sp.delta_t_abort = so.delta_t_abort;
      sa.tail.sp.eig_thresh = so.eig_thresh;
    }
    bool cert_counters_reset = false;
    // a batch through the grid sweep launches, from a loop's second sweep on, only the workgroups of the scans still running
    const bool compact = sa.grid == 1 && n_scans >= 4 && !(((o.ab_switches | ctx->env_ab) & LSLAM_AB_NO_COMPACT));
    std::vector<int32_t> active_n((size_t)n_chunks, 0);
    auto enqueue = [&](int c, int iters) -> int {
      const int p0 = c * in_flight, p1 = std::min(n_scans, p0 + in_flight);
      const int32_t fb = ctx->h_probs[(size_t)p0].first_block;
      const int32_t lb = ctx->h_probs[(size_t)p1 - 1].first_block + ctx->h_probs[(size_t)p1 - 1].n_blocks;
      SweepArgs sc = sa;
      sc.blocks = ctx->blocks.p + fb;
      sc.nb_total = lb - fb;
      sc.partials = ctx->partials.p + (size_t)fb * NCOL;
      if (compact && done_iters[(size_t)c] > 0) {  // only the workgroups of the scans still running (compact_active_kernel, below)
        sc.active_blocks = ctx->active_blocks.p + fb;
        sc.n_active = active_n[(size_t)c];
      }
      // throughput-bound launches only (launch_sweep's own test: more wavefronts than two per SIMD): a launch that fits the
      // device at once ends when its slowest wavefront does, certificates or not, and the second pass is two launches more
      // per iteration (measured on single scans: 0.29 against 0.26 ms per loop).  LSLAM_KNN_CERT=2 takes it regardless (tests)
      if ((sc.grid || sc.prev_q) && !sc.tail.count && (sc.grid || force_cert || (long)sc.nb_total * (SWEEP_BLOCK / 64) > 2 * 1024)) {
        if (!cert_counters_reset) {  // once per call: whatever an earlier call that ended in an error left in the plan's counters
          HIP_TRY(hipMemsetAsync(ctx->cert_count.p, 0, 6 * sizeof(int32_t), ctx->stream));
          cert_counters_reset = true;
        }
        sc.need_list = ctx->need_list.p + (size_t)fb * SWEEP_BLOCK;
        sc.need_cnt = ctx->need_cnt.p + fb;
        if (sc.grid == 1 && !no_probe2) {
          sc.need2_list = ctx->need2_list.p + (size_t)fb * SWEEP_BLOCK;
          sc.need2_cnt = ctx->need2_cnt.p + fb;
        }
        sc.groups = ctx->groups.p + ctx->h_prob_group0[(size_t)p0];
        sc.n_groups = ctx->h_prob_group0[(size_t)p1] - ctx->h_prob_group0[(size_t)p0];
        sc.group_block_base = fb;
      }
      SolveArgs soc = so;
      soc.states = ctx->d_state + p0;
      soc.probs = ctx->probs.p + p0;
      soc.n_prob = p1 - p0;
      for (int b = 0; b < iters; ++b) {
        // the first sweep of a loop is bounded by the acceptance gate only, later ones also by the
        // neighbours the previous sweep of THIS loop found
        sc.prev_valid = (sc.bounded && done_iters[(size_t)c] > 0) ? 1 : 0;
        hipEvent_t e0, e1;
        HIP_TRY(sweep_events(n_launches, &e0, &e1));
        int variant = -1;
        HIP_TRY(sweep_launch(ctx, sc, o.jtj_mode, e0, e1, &variant));
        ++n_launches;
        HIP_TRY(launch_stereo(sta, ctx->stream));
        if (variant != SWEEP_VARIANT_DEEP_FUSED) HIP_TRY(launch_solve(soc, ctx->stream));
        ++done_iters[(size_t)c];
      }
      return LSLAM_OK;
    };
    // overflow area of the shallow LDS stack: sized for the largest chunk
    {
      int32_t max_nb = 0;
      for (int c = 0; c < n_chunks; ++c) {
        const int p0 = c * in_flight, p1 = std::min(n_scans, p0 + in_flight);
        max_nb = std::max(max_nb, ctx->h_probs[(size_t)p1 - 1].first_block + ctx->h_probs[(size_t)p1 - 1].n_blocks -
                                      ctx->h_probs[(size_t)p0].first_block);
      }
      HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words((size_t)std::max(max_nb, 1) * SWEEP_BLOCK, std::max(ctx->tc.depth, ctx->ts.depth))));
      sa.stack_ovf = ctx->stack_ovf.p;
    }
    if (compact) {
      // One iteration at a time, and between two of them the host learns how many workgroups are left (4 bytes per chunk, one
      // wait of ~30 us against sweeps of milliseconds): the next sweep is launched over THOSE, not over every workgroup of
      // every scan.  The states come back once, at the end.
      HIP_TRY(ctx->active_blocks.reserve((size_t)std::max(ctx->nb_total, 1)));
      HIP_TRY(ctx->d_active_cnt.reserve((size_t)n_chunks));
      if (ctx->h_active_cap < (size_t)n_chunks) {
        if (ctx->h_active) (void)hipHostFree(ctx->h_active);
        ctx->h_active = nullptr;
        ctx->h_active_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&ctx->h_active, sizeof(int32_t) * (size_t)n_chunks, hipHostMallocDefault));
        ctx->h_active_cap = (size_t)n_chunks;
      }
      for (;;) {
        bool any = false;
        for (int c = 0; c < n_chunks; ++c) {
          if (finished[(size_t)c]) continue;
          if (done_iters[(size_t)c] >= max_it) { finished[(size_t)c] = 1; continue; }
          rc = enqueue(c, 1);
          if (rc) return rc;
          const int p0 = c * in_flight, p1 = std::min(n_scans, p0 + in_flight);
          const int32_t fb = ctx->h_probs[(size_t)p0].first_block;
          HIP_TRY(launch_compact_active(ctx->d_state + p0, ctx->probs.p + p0, p1 - p0, fb, ctx->active_blocks.p + fb, ctx->d_active_cnt.p + c, ctx->stream));
          any = true;
        }
        if (!any) break;
        launched = *std::max_element(done_iters.begin(), done_iters.end());
        HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->h_active, ctx->d_active_cnt.p, sizeof(int32_t) * (size_t)n_chunks, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        bool all_done = true;
        for (int c = 0; c < n_chunks; ++c) {
          if (finished[(size_t)c]) continue;
          active_n[(size_t)c] = ctx->h_active[c];
          if (active_n[(size_t)c] <= 0 || done_iters[(size_t)c] >= max_it) finished[(size_t)c] = 1;
          all_done = all_done && finished[(size_t)c];
        }
        if (all_done) break;
      }
      HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState) * (size_t)n_scans, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
    } else
    for (;;) {
      bool any = false;
      for (int c = 0; c < n_chunks; ++c) {
        if (finished[(size_t)c]) continue;
        const int iters = std::min(batch, max_it - done_iters[(size_t)c]);
        if (iters <= 0) { finished[(size_t)c] = 1; continue; }
        rc = enqueue(c, iters);
        if (rc) return rc;
        any = true;
      }
      launched = *std::max_element(done_iters.begin(), done_iters.end());
      HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState) * (size_t)n_scans,
                             hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      bool all_done = true;
      for (int c = 0; c < n_chunks; ++c) {
        const int p0 = c * in_flight, p1 = std::min(n_scans, p0 + in_flight);
        bool cd = true;
        for (int p = p0; p < p1; ++p) cd = cd && ctx->h_state[p].done;
        if (cd || done_iters[(size_t)c] >= max_it) finished[(size_t)c] = 1;
        all_done = all_done && finished[(size_t)c];
      }
      if (all_done || !any) break;
      batch = 2;
    }
  }
  // ---- _fineScore (ScanMatch.cpp:272-321) ---------------------------------------------------------------------------
  // After a converged loop with the score gate on, the reference sweeps once more at the FINAL pose, accepting a point when
  // its nearest neighbour is within sqrt(0.02) m (corner) / sqrt(0.05) m (surf) instead of the fifth within sqrt(5) m, and
  // prints score2 / percent2; neither enters the return value.  One more launch over the converged scans (unbounded search:
  // the bound of the loop's sweeps assumes the d2[4] < 5 gate), their sums reduced by the solve kernel's first half.
  std::vector<double> score2((size_t)n_scans, 0.0), match2((size_t)n_scans, 0.0);
  if (o.fine_score && o.use_score && max_it > 0) {
    bool any_conv = false;
    for (int32_t p = 0; p < n_scans; ++p) any_conv = any_conv || ctx->h_state[p].converged;
    if (any_conv) {
      SweepArgs sf = sa;
      sf.tail = SweepTail{};  // this pass only reduces
      sf.bounded = 0;
      sf.prev_valid = 0;
      sf.fine_gate_c = 0.02f;  // :282
      sf.fine_gate_s = 0.05f;  // :302
      const int in_flight = sharded ? n_scans : (o.scans_in_flight > 0 ? std::min<int>(o.scans_in_flight, n_scans) : std::min<int>(n_scans, 128));
      const int n_chunks = (n_scans + in_flight - 1) / in_flight;
      int32_t max_nb = 1;
      for (int c = 0; c < n_chunks; ++c) {
        const int p0 = c * in_flight, p1 = std::min(n_scans, p0 + in_flight);
        max_nb = std::max(max_nb, ctx->h_probs[(size_t)p1 - 1].first_block + ctx->h_probs[(size_t)p1 - 1].n_blocks - ctx->h_probs[(size_t)p0].first_block);
      }
      HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words((size_t)max_nb * SWEEP_BLOCK, std::max(ctx->tc.depth, ctx->ts.depth))));
      sf.stack_ovf = ctx->stack_ovf.p;
      for (int c = 0; c < n_chunks; ++c) {
        const int p0 = c * in_flight, p1 = std::min(n_scans, p0 + in_flight);
        const int32_t fb = ctx->h_probs[(size_t)p0].first_block;
        const int32_t lb = ctx->h_probs[(size_t)p1 - 1].first_block + ctx->h_probs[(size_t)p1 - 1].n_blocks;
        SweepArgs sc = sf;
        sc.blocks = ctx->blocks.p + fb;
        sc.nb_total = lb - fb;
        sc.partials = ctx->partials.p + (size_t)fb * NCOL;
        if (sc.grid) {  // the grid sweep's second pass (as enqueue() sets it up)
          if (sharded) {
            sc.grid = 0;
          } else {
            sc.need_list = ctx->need_list.p + (size_t)fb * SWEEP_BLOCK;
            sc.need_cnt = ctx->need_cnt.p + fb;
            sc.groups = ctx->groups.p + ctx->h_prob_group0[(size_t)p0];
            sc.n_groups = ctx->h_prob_group0[(size_t)p1] - ctx->h_prob_group0[(size_t)p0];
            sc.group_block_base = fb;
          }
        }
        SolveArgs soc = so;
        soc.states = ctx->d_state + p0;
        soc.probs = ctx->probs.p + p0;
        soc.n_prob = p1 - p0;
        soc.reduce_only = 2;
        soc.ext_sums = nullptr;
        soc.partials2 = nullptr;  // LiDAR rows only
        soc.n_blocks2 = 0;
        soc.sums_out = sharded ? xchg : nullptr;
        HIP_TRY(sweep_launch(ctx, sc, o.jtj_mode));
        HIP_TRY(launch_solve(soc, ctx->stream));
      }
      double xs[NCOL] = {0};
      if (sharded) {  // every rank converged together (same sums, same solve): every rank is here
        rc = exchange();
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(xs, xchg, sizeof(xs), hipMemcpyDeviceToHost, ctx->stream));
      }
      HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
      HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState) * (size_t)n_scans, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      for (int32_t p = 0; p < n_scans; ++p) {
        if (!ctx->h_state[p].converged) continue;
        const double *sm = sharded ? xs : ctx->h_state[p].sums;
        score2[(size_t)p] = sm[COL_SCORE];
        match2[(size_t)p] = sm[COL_LINE] + sm[COL_PLANE];
      }
    }
  }
  ctx->grid_state_valid = !sharded && !gnp_done && ctx->sweep_variants[SWEEP_VARIANT_GRID] > grid_launches_before && n_scans == 1;
  ctx->stage_busy = false;  // the stream has been waited for since the scan was set
  if (lazy) {  // a point's answer needed nanoflann's visit order (sweep_wide_kernel): the trees after all, and the call again
    bool need_tree = false;
    for (int32_t p = 0; p < n_scans; ++p) need_tree = need_tree || ctx->h_state[p].pad != 0;
    if (need_tree) {
      rc = ensure_trees(ctx);
      if (rc) return rc;
      return run_batch_impl(ctx, n_scans, poses, opts_in, stats, fn, user, xchg, use_comm);
    }
  }
  int max_sweeps = 0, max_iter = 0;
  for (int32_t p = 0; p < n_scans; ++p) {
    max_sweeps = std::max(max_sweeps, ctx->h_state[p].sweeps);
    max_iter = std::max(max_iter, ctx->h_state[p].iter);
  }
  // the spare iteration (five immediate exits of 4 - 5 us each on the grid path) is dropped once three calls in a row ran the
  // same number of iterations -- a mapping node's frames do; a loop that then needs one more costs one more round trip
  ctx->iter_same = (max_iter == ctx->iter_last) ? std::min(ctx->iter_same + 1, 1000) : 0;
  ctx->iter_last = max_iter;
  ctx->iter_hint = max_iter + (ctx->iter_same >= 2 ? 0 : 1);

  float gpu_ms_total = 0.f, gpu_ms_sweep = 0.f;
  int sweep_launches = 0;
  HIP_TRY(hipEventElapsedTime(&gpu_ms_total, ctx->ev0, ctx->ev1));
  if (o.profile) {
    // every sweep launch of this call, the trailing ones that found all scans converged included
    // (a few microseconds each): the same population rocprofv3's per-kernel average is taken over
    for (int it = 0; it < (sharded ? launched : n_launches); ++it) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, ctx->sweep_ev[2 * it], ctx->sweep_ev[2 * it + 1]));
      gpu_ms_sweep += ms;
      ++sweep_launches;
    }
  }
  int worst = LSLAM_OK;
  for (int32_t p = 0; p < n_scans; ++p) {
    const GNState &g = ctx->h_state[p];
    for (int i = 0; i < 6; ++i) poses[6 * p + i] = g.pose[i];  // always written back
    const size_t npts = (size_t)ctx->nqc[(size_t)p] + (size_t)ctx->nqs[(size_t)p];
    int status;
    double score = 0.0, percent = 0.0, s2 = 0.0, pc2 = 0.0;
    if (g.converged && o.use_score) {  // ScanMatch.cpp:263-341
      score = g.score;
      const double match_count = (double)g.n_line + (double)g.n_plane;
      percent = (float)(match_count / (total_points >= 0.0 ? total_points : (double)npts));
      if (o.fine_score) {  // :317-319
        s2 = score2[(size_t)p];
        pc2 = (float)(match2[(size_t)p] / (total_points >= 0.0 ? total_points : (double)npts));
      }
      if (score < o.score_threshold) status = LSLAM_LOW_SCORE;
      else if (percent < o.match_percentage_threshold) status = LSLAM_LOW_PERCENT;
      else status = LSLAM_OK;
    } else if (g.too_few) {
      status = LSLAM_TOO_FEW_MATCHES;
    } else {
      status = LSLAM_NOT_CONVERGED;
    }
    if (status != LSLAM_OK && worst == LSLAM_OK) worst = status;
    if (stats) {
      lslam_stats &st = stats[p];
      st.status = status;
      st.iterations = g.iter;
      st.n_line = g.n_line;
      st.n_plane = g.n_plane;
      st.n_rows = g.n_rows;
      st.degenerate = g.degenerate;
      st.converged = g.converged;
      st.delta_r = g.delta_r;
      st.delta_t = g.delta_t;
      st.score = score;
      st.percent = percent;
      st.score2 = s2;
      st.percent2 = pc2;
      st.sweeps = g.sweeps;
      st.point_residuals = (int64_t)g.sweeps * (int64_t)npts;
      st.gpu_ms_total = gpu_ms_total;    // whole batch
      st.gpu_ms_sweep = gpu_ms_sweep;    // whole batch
      st.sweep_launches = sweep_launches;
    }
  }
  return n_scans == 1 ? (stats ? stats[0].status : worst) : worst;
}
}  // namespace

// ---- stereo term of the joint system (include/lslam_c.h; no reference code: parity unpinned) -----
void lslam_stereo_default_cam(lslam_stereo_cam *c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->fx = c->fy = 700.0f;  // a ZED-class rectified pair at 1280x720 (README.md:53)
  c->cx = 640.0f;
  c->cy = 360.0f;
  c->bf = 0.12f * 700.0f;
  c->T_cl[0] = c->T_cl[5] = c->T_cl[10] = 1.0f;
  c->weight = 1e-4f;
  c->huber_stereo = 2.7955322f;  // sqrt(7.815)
  c->huber_mono = 2.4476519f;    // sqrt(5.991)
  c->gate_outliers = 0;
  c->min_depth = 0.1f;
}

int lslam_stereo_clear(lslam_ctx *ctx) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  ctx->n_stereo = 0;
  return LSLAM_OK;
}

int lslam_stereo_set(lslam_ctx *ctx, const float *landmarks_xyz, const float *obs, const float *inv_sigma2,
                     size_t n, const lslam_stereo_cam *cam) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (n == 0) return lslam_stereo_clear(ctx);
  if (!landmarks_xyz || !obs || !cam || n >= (size_t)1 << 30) {
    set_err("bad stereo arguments");
    return LSLAM_ERR_INVALID;
  }
  if (!(cam->fx > 0.0f) || !(cam->fy > 0.0f) || !(cam->weight >= 0.0f) || !(cam->huber_stereo > 0.0f) ||
      !(cam->huber_mono > 0.0f)) {
    set_err("bad stereo camera (fx, fy, huber deltas must be positive, weight non-negative)");
    return LSLAM_ERR_INVALID;
  }
  std::vector<float4> lm(n), ob(n);
  for (size_t i = 0; i < n; ++i) {
    lm[i] = make_float4(landmarks_xyz[3 * i], landmarks_xyz[3 * i + 1], landmarks_xyz[3 * i + 2],
                        inv_sigma2 ? inv_sigma2[i] : 1.0f);
    ob[i] = make_float4(obs[3 * i], obs[3 * i + 1], obs[3 * i + 2], 0.0f);
  }
  HIP_TRY(ctx->st_lm.reserve(n));
  HIP_TRY(ctx->st_obs.reserve(n));
  HIP_TRY(ctx->st_partials.reserve((size_t)stereo_blocks((int)n) * NCOL));
  HIP_TRY(hipMemcpyAsync(ctx->st_lm.p, lm.data(), n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->st_obs.p, ob.data(), n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));  // lm, ob are locals
  StereoCam &c = ctx->st_cam;
  c.fx = cam->fx; c.fy = cam->fy; c.cx = cam->cx; c.cy = cam->cy; c.bf = cam->bf;
  std::memcpy(c.T_cl, cam->T_cl, sizeof(c.T_cl));
  c.weight = cam->weight;
  c.huber_stereo = cam->huber_stereo;
  c.huber_mono = cam->huber_mono;
  c.gate_outliers = cam->gate_outliers;
  c.min_depth = cam->min_depth;
  ctx->n_stereo = (int32_t)n;
  return LSLAM_OK;
}

int lslam_stereo_sums(lslam_ctx *ctx, const float pose[6], double sums32[32]) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (!pose || !sums32) { set_err("null argument"); return LSLAM_ERR_INVALID; }
  if (ctx->n_stereo <= 0) { set_err("no stereo observations set"); return LSLAM_ERR_INVALID; }
  rc = ensure_states(ctx, 1);
  if (rc) return rc;
  init_state(*ctx->h_state, pose);
  HIP_TRY(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(GNState), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx->st_noblocks.reserve(1));
  const ProbBlocks none{0, 0};
  HIP_TRY(hipMemcpyAsync(ctx->st_noblocks.p, &none, sizeof(none), hipMemcpyHostToDevice, ctx->stream));
  StereoArgs sta{};
  sta.landmarks = ctx->st_lm.p;
  sta.obs = ctx->st_obs.p;
  sta.n = ctx->n_stereo;
  sta.cam = ctx->st_cam;
  sta.state = ctx->d_state;
  sta.partials = ctx->st_partials.p;
  HIP_TRY(launch_stereo(sta, ctx->stream));
  SolveArgs so{};
  so.states = ctx->d_state;
  so.partials = ctx->st_partials.p;  // unused: the problem has no LiDAR blocks
  so.probs = ctx->st_noblocks.p;
  so.n_prob = 1;
  so.reduce_only = 1;
  so.partials2 = ctx->st_partials.p;
  so.n_blocks2 = stereo_blocks(ctx->n_stereo);
  HIP_TRY(launch_solve(so, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));  // `none` is a local too
  for (int i = 0; i < NCOL; ++i) sums32[i] = ctx->h_state->sums[i];
  return LSLAM_OK;
}

int lslam_scanmatch_run_batch(lslam_ctx *ctx, int32_t n_scans, float *poses, const lslam_opts *opts,
                              lslam_stats *stats) {
  return run_batch_impl(ctx, n_scans, poses, opts, stats, nullptr, nullptr, nullptr);
}

int lslam_scanmatch_run_sharded(lslam_ctx *ctx, float pose[6], const lslam_opts *opts, lslam_allreduce_fn fn,
                                void *user, double *xchg32, lslam_stats *stats) {
  if (!pose || (fn && !xchg32) || (!fn && !(ctx && ctx->comm))) {
    set_err("run_sharded needs a pose and either an attached communicator (lslam_ctx_set_comm) or an all-reduce hook "
            "with a 32-double device buffer");
    return LSLAM_ERR_INVALID;
  }
  if (ctx && ctx->n_prob != 1) {
    set_err("run_sharded works on one resident scan shard (lslam_scan_set)");
    return LSLAM_ERR_INVALID;
  }
  lslam_stats local;
  lslam_stats *st = stats ? stats : &local;
  const int rc = run_batch_impl(ctx, 1, pose, opts, st, fn, user, xchg32, fn == nullptr);
  return rc < 0 ? rc : st->status;
}

int lslam_scanmatch_run(lslam_ctx *ctx, float pose[6], const lslam_opts *opts, lslam_stats *stats) {
  if (!pose) {
    set_err("null pose");
    return LSLAM_ERR_INVALID;
  }
  lslam_stats local;
  lslam_stats *st = stats ? stats : &local;
  const int rc = lslam_scanmatch_run_batch(ctx, 1, pose, opts, st);
  return rc < 0 ? rc : st->status;
}

int lslam_scanmatch_scan(lslam_ctx *ctx, const void *corner, size_t n_corner, const void *surf,
                         size_t n_surf, size_t stride_bytes, float pose[6],
                         const lslam_opts *opts, lslam_stats *stats) {
  int rc = lslam_scan_set(ctx, corner, n_corner, surf, n_surf, stride_bytes);
  if (rc) return rc;
  return lslam_scanmatch_run(ctx, pose, opts, stats);
}

int lslam_scanmatch_full(lslam_ctx *ctx, const void *ref_corner, size_t n_ref_corner,
                         const void *ref_surf, size_t n_ref_surf, size_t ref_stride_bytes,
                         const void *corner, size_t n_corner, const void *surf, size_t n_surf,
                         size_t stride_bytes, float pose[6], const lslam_opts *opts,
                         lslam_stats *stats) {
  // ScanMatch.cpp:57-61 comes before the trees are built
  if (n_ref_corner < 50 || n_ref_surf < 100) {
    if (stats) {
      std::memset(stats, 0, sizeof(*stats));
      stats->status = LSLAM_TOO_FEW_REF;
    }
    return LSLAM_TOO_FEW_REF;
  }
  int rc = lslam_map_set(ctx, ref_corner, n_ref_corner, ref_surf, n_ref_surf, ref_stride_bytes);
  if (rc) return rc;
  return lslam_scanmatch_scan(ctx, corner, n_corner, surf, n_surf, stride_bytes, pose, opts, stats);
}

// Variant B: LaserOdometry::scanMatch (odometry/LaserOdometry.cpp:328-647) through kd-trees of the last clouds, one launch per
// step: the implementation of rounds 1-5.  lslam_odometry_match (lslam_odom.hip) searches hashed cell grids instead and comes
// here only for a sweep with an exact distance tie, where nanoflann's visit order decides (and under LSLAM_ODOM_TREES=1, the A/B
// switch of the two).
}  // extern "C"
namespace lslam {
int odometry_match_trees(lslam_ctx *ctx, const void *last_corner, size_t n_lc, const void *last_surf, size_t n_ls, const void *sharp,
                         size_t n_sharp, const void *flat, size_t n_flat, size_t stride_bytes, float pose[6], int32_t max_iterations,
                         float delta_t_abort, float delta_r_abort, lslam_stats *stats);
}
int lslam::odometry_match_trees(lslam_ctx *ctx, const void *last_corner, size_t n_lc, const void *last_surf,
                         size_t n_ls, const void *sharp, size_t n_sharp, const void *flat, size_t n_flat,
                         size_t stride_bytes, float pose[6], int32_t max_iterations, float delta_t_abort,
                         float delta_r_abort, lslam_stats *stats) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (stride_bytes < 16 || (stride_bytes & 3) || !pose || (n_lc && !last_corner) || (n_ls && !last_surf) ||
      (n_sharp && !sharp) || (n_flat && !flat) || n_lc >= KD_MAX_POINTS || n_ls >= KD_MAX_POINTS) {
    set_err("bad odometry arguments (clouds need x,y,z,intensity: stride >= 16)");
    return LSLAM_ERR_INVALID;
  }
  lslam_stats local;
  lslam_stats &st = stats ? *stats : local;
  std::memset(&st, 0, sizeof(st));
  if (!(n_lc > 10 && n_ls > 100)) {  // :337
    st.status = LSLAM_TOO_FEW_REF;
    return LSLAM_TOO_FEW_REF;
  }
  auto pack4 = [&](const void *src, size_t n, std::vector<float4> &out) {
    out.resize(n);
    const char *p = static_cast<const char *>(src);
    const size_t ioff = stride_bytes >= 32 ? 16 : 12;  // pcl::PointXYZI keeps intensity at byte 16
    for (size_t i = 0; i < n; ++i) {
      float xyz[3], w;
      std::memcpy(xyz, p + i * stride_bytes, 12);
      std::memcpy(&w, p + i * stride_bytes + ioff, 4);
      out[i] = make_float4(xyz[0], xyz[1], xyz[2], w);
    }
  };
  std::vector<float4> lc, ls, q, qf;
  pack4(last_corner, n_lc, lc);
  pack4(last_surf, n_ls, ls);
  pack4(sharp, n_sharp, q);
  pack4(flat, n_flat, qf);
  q.insert(q.end(), qf.begin(), qf.end());
  // the map slot of the context holds the two kd-trees of the last clouds (device build, host
  // builder as its fallback -- the same path as lslam_map_set)
  rc = map_set_impl(ctx, lc.data(), n_lc, ls.data(), n_ls, sizeof(float4), nullptr, nullptr, false);  // (the trees are used right here)
  if (rc) return rc;
  ctx->have_map = false;
  ctx->map_epoch++;  // these trees belong to this call, not to a resident map
  if (ctx->tc.depth > KD_STACK_LDS + 1 || ctx->ts.depth > KD_STACK_LDS + 1) {
    set_err("kd-tree deeper than %d", KD_STACK_LDS + 1);
    return LSLAM_ERR_TREE_DEPTH;
  }
  const size_t nq = q.size();
  DevBuf<float4> &d_oc = ctx->od_oc, &d_os = ctx->od_os, &d_q = ctx->od_q, &d_sel = ctx->od_sel;
  DevBuf<int32_t> &d_ind = ctx->od_ind;
  HIP_TRY(d_sel.reserve(nq + 1));
  HIP_TRY(d_oc.reserve(n_lc + 1));
  HIP_TRY(d_os.reserve(n_ls + 1));
  HIP_TRY(d_q.reserve(nq + 1));
  HIP_TRY(d_ind.reserve(3 * nq + 1));
  HIP_TRY(hipMemcpyAsync(d_oc.p, lc.data(), n_lc * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_os.p, ls.data(), n_ls * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  if (nq) HIP_TRY(hipMemcpyAsync(d_q.p, q.data(), nq * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemsetAsync(d_ind.p, 0xFF, (3 * nq + 1) * sizeof(int32_t), ctx->stream));
  OdomArgs oa{};
  oa.tc = ctx->tc.view;
  oa.ts = ctx->ts.view;
  oa.oc = d_oc.p;
  oa.os = d_os.p;
  oa.n_oc = (int32_t)n_lc;
  oa.n_os = (int32_t)n_ls;
  oa.q = d_q.p;
  oa.qf = d_q.p + n_sharp;
  oa.n_sharp = (int32_t)n_sharp;
  oa.n_flat = (int32_t)n_flat;
  oa.nb_sharp = (int32_t)((n_sharp + 255) / 256);
  oa.nb_total = oa.nb_sharp + (int32_t)((n_flat + 255) / 256);
  oa.ind = d_ind.p;
  oa.sel = d_sel.p;
  oa.mode = 0;
  oa.state = ctx->d_state;
  HIP_TRY(ctx->partials.reserve((size_t)(oa.nb_total ? oa.nb_total : 1) * NCOL));
  oa.partials = ctx->partials.p;
  ProbBlocks pb{0, oa.nb_total};
  HIP_TRY(ctx->probs.reserve(1));
  HIP_TRY(hipMemcpyAsync(ctx->probs.p, &pb, sizeof(pb), hipMemcpyHostToDevice, ctx->stream));
  init_state(*ctx->h_state, pose);
  const int max_it = max_iterations < 0 ? 0 : max_iterations;
  if (max_it == 0 || oa.nb_total == 0) ctx->h_state->done = 1;
  HIP_TRY(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(GNState), hipMemcpyHostToDevice, ctx->stream));
  SolveArgs so{};
  so.states = ctx->d_state;
  so.partials = ctx->partials.p;
  so.probs = ctx->probs.p;
  so.n_prob = 1;
  so.max_iterations = max_it;
  so.delta_r_abort = delta_r_abort;
  so.delta_t_abort = delta_t_abort;
  so.eig_thresh = 10.0f;  // :596
  so.min_rows = 10;       // :501
  so.too_few_continue = 1;
  so.nan_reset = 1;
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  const bool inline_search = env_once().odom_inline;  // A/B switch
  // Like the scan-to-map loop: the first batch of iterations is sized from the previous sweep's count
  // (+1 spare), then the host looks at the state and enqueues five more at a time -- launches after the
  // loop has ended exit at once but still cost a few microseconds each (25 x 2 of them per sweep).
  // loop_iter advances by one per solve launch until the loop is done, so the host knows which
  // launches refresh the correspondences (every fifth, :357,:423): nearest neighbour per lane, then
  // the ring-window searches one wavefront per query, then the residual pass on the cached indices
  int launched = 0;
  int batch = ctx->od_iter_hint < 1 ? 1 : ctx->od_iter_hint;
  for (;;) {
    if (batch > max_it - launched) batch = max_it - launched;
    for (int b = 0; b < batch; ++b) {
      const int it = launched + b;
      if (inline_search) {
        oa.mode = 0;
      } else {
        if (it % 5 == 0) {
          oa.mode = 1;
          HIP_TRY(launch_odom_sweep(oa, ctx->stream));
          HIP_TRY(launch_odom_window(oa, ctx->stream));
        }
        oa.mode = 2;
      }
      HIP_TRY(launch_odom_sweep(oa, ctx->stream));
      HIP_TRY(launch_solve(so, ctx->stream));
    }
    launched += batch;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->h_state->done || launched >= max_it) break;
    batch = 5;
  }
  ctx->od_iter_hint = ctx->h_state->loop_iter + 1;
  const GNState &g = *ctx->h_state;
  for (int i = 0; i < 6; ++i) pose[i] = g.pose[i];
  st.iterations = g.iter;
  st.sweeps = g.sweeps;
  st.n_rows = g.n_rows;
  st.n_line = g.n_line;
  st.n_plane = g.n_plane;
  st.degenerate = g.degenerate;
  st.converged = g.converged;
  st.delta_r = g.delta_r;
  st.delta_t = g.delta_t;
  st.point_residuals = (int64_t)g.sweeps * (int64_t)nq;
  HIP_TRY(hipEventElapsedTime(&st.gpu_ms_total, ctx->ev0, ctx->ev1));
  st.status = g.converged ? LSLAM_OK : LSLAM_NOT_CONVERGED;
  ctx->have_scan = false;
  return st.status;
}
extern "C" {

// LaserOdometry::transformToEnd (odometry/LaserOdometry.cpp:156-168) on a host cloud, in place.
int lslam_transform_to_end(lslam_ctx *ctx, void *cloud, size_t n, size_t stride_bytes, const float pose[6]) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (!pose || (n && !cloud) || stride_bytes < 16 || (stride_bytes & 3) || n > 0x3FFFFFFFu) {
    set_err("bad transform_to_end arguments (points need {x,y,z} and the intensity)");
    return LSLAM_ERR_INVALID;
  }
  if (n == 0) return LSLAM_OK;
  const size_t ioff = stride_bytes == 16 ? 12 : 16;  // PointXYZI: intensity at byte 16
  std::vector<float4> h(n);
  char *p = static_cast<char *>(cloud);
  for (size_t i = 0; i < n; ++i) {
    float v[3], w;
    std::memcpy(v, p + i * stride_bytes, 12);
    std::memcpy(&w, p + i * stride_bytes + ioff, 4);
    h[i] = make_float4(v[0], v[1], v[2], w);
  }
  HIP_TRY(ctx->t_q.reserve(n));
  HIP_TRY(ctx->t_small.reserve(64));
  HIP_TRY(hipMemcpyAsync(ctx->t_q.p, h.data(), n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->t_small.p, pose, 6 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(launch_odom_to_end(ctx->t_q.p, (int)n, ctx->t_small.p, ctx->stream));
  HIP_TRY(hipMemcpyAsync(h.data(), ctx->t_q.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n; ++i) std::memcpy(p + i * stride_bytes, &h[i], 12);
  return LSLAM_OK;
}

// util/transform_utils.h:502-507 transformAssociate: Wnew = (Wold * Lold^-1) * Lnew
// (Eigen::Isometry3f::inverse() = [R^T | -R^T t]; fp32 products in row.column order)
void lslam_transform_associate(const float Lold[16], const float Lnew[16], const float Wold[16],
                               float Wnew[16]) {
  auto mul = [](const float A[16], const float B[16], float C[16]) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        C[r * 4 + c] = (A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c]) + A[r * 4 + 2] * B[2 * 4 + c];
      C[r * 4 + 3] = ((A[r * 4 + 0] * B[3] + A[r * 4 + 1] * B[7]) + A[r * 4 + 2] * B[11]) + A[r * 4 + 3];
    }
    C[12] = 0.f; C[13] = 0.f; C[14] = 0.f; C[15] = 1.f;
  };
  float Linv[16], L2W[16];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Linv[r * 4 + c] = Lold[c * 4 + r];
    Linv[r * 4 + 3] = -((Lold[0 * 4 + r] * Lold[3] + Lold[1 * 4 + r] * Lold[7]) + Lold[2 * 4 + r] * Lold[11]);
  }
  Linv[12] = 0.f; Linv[13] = 0.f; Linv[14] = 0.f; Linv[15] = 1.f;
  mul(Wold, Linv, L2W);
  mul(L2W, Lnew, Wnew);
}

// util/transform_utils.h:313-323 + :54-60
void lslam_isometry_to_pose(const float T[16], float pose[6]) {
  pose[0] = std::atan2(T[2 * 4 + 1], T[2 * 4 + 2]);
  pose[1] = std::asin(-T[2 * 4 + 0]);
  pose[2] = std::atan2(T[1 * 4 + 0], T[0 * 4 + 0]);
  pose[3] = T[0 * 4 + 3];
  pose[4] = T[1 * 4 + 3];
  pose[5] = T[2 * 4 + 3];
}

// util/transform_utils.h:308-311 + :288-299
void lslam_pose_to_isometry(const float pose[6], float T[16]) {
  float R[9], t[3], sc[6];
  pose_to_Rt_sc(pose, R, t, sc, HostSinCos());
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[r * 4 + c] = R[r * 3 + c];
    T[r * 4 + 3] = t[r];
  }
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

int lslam_knn5(lslam_ctx *ctx, int which_map, const void *queries, size_t nq, size_t stride_bytes,
               int32_t *idx_out, float *d2_out) {
  return lslam_knn5_ex(ctx, which_map, queries, nq, stride_bytes, LSLAM_SEARCH_LANE, idx_out, d2_out, nullptr);
}

int lslam_knn5_ex(lslam_ctx *ctx, int which_map, const void *queries, size_t nq, size_t stride_bytes,
                  int32_t search_mode, int32_t *idx_out, float *d2_out, int32_t *n_ties) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (!ctx->have_map || ctx->cube_mode) { set_err("no whole-map tree set"); return LSLAM_ERR_NO_MAP; }
  if ((which_map != 0 && which_map != 1) || stride_bytes < 12 || (stride_bytes & 3) ||
      (nq && (!queries || !idx_out || !d2_out)) || nq > 0x0FFFFFFFu) {
    set_err("bad knn5 arguments");
    return LSLAM_ERR_INVALID;
  }
  if (nq == 0) return LSLAM_OK;
  std::vector<float4> q;
  pack_cloud(queries, nq, stride_bytes, q);
  HIP_TRY(ctx->t_q.reserve(nq));
  HIP_TRY(ctx->t_idx.reserve(nq * 5));
  HIP_TRY(ctx->t_d2.reserve(nq * 5));
  HIP_TRY(hipMemcpyAsync(ctx->t_q.p, q.data(), nq * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  const TreeView &T = which_map ? ctx->ts.view : ctx->tc.view;
  uint32_t *ovf = nullptr;
  if (n_ties) *n_ties = 0;
  if (search_mode == LSLAM_SEARCH_PACKET) {
    rc = ensure_packet_nodes(ctx);
    if (rc) return rc;
    if (!T.pn) { set_err("this map has no packet-search nodes"); return LSLAM_ERR_INVALID; }
    const size_t nthr = ((nq + 255) / 256) * 256;
    HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words(nthr)));
    HIP_TRY(ctx->t_small.reserve(64));
    int32_t *d_tie = reinterpret_cast<int32_t *>(ctx->t_small.p);
    HIP_TRY(hipMemsetAsync(d_tie, 0, sizeof(int32_t), ctx->stream));
    HIP_TRY(launch_knn5_packet(T, ctx->t_q.p, (int)nq, ctx->t_idx.p, ctx->t_d2.p, ctx->stack_ovf.p, d_tie, ctx->stream));
    if (n_ties) HIP_TRY(hipMemcpyAsync(n_ties, d_tie, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  } else if ((search_mode & 0xFF) == LSLAM_SEARCH_GRID) {
    // the grid probe; n_ties receives the queries it could not prove (searched in the tree by the same kernel)
    rc = ensure_grid(ctx, ctx->env_grid_cell > 0.0f ? ctx->env_grid_cell : 0.0f);
    if (rc) return rc;
    const CellGrid &G = which_map ? ctx->ks.view : ctx->kc.view;
    if (!G.cell_start) { set_err("this map has no cell grid (status %d)", ctx->grid_status); return LSLAM_ERR_INVALID; }
    const size_t nthr = ((nq + 255) / 256) * 256;
    HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words(nthr)));
    HIP_TRY(ctx->t_small.reserve(64));
    int32_t *d_cnt = reinterpret_cast<int32_t *>(ctx->t_small.p);
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), ctx->stream));
    HIP_TRY(launch_knn5_grid(G, T, ctx->t_q.p, (int)nq, ctx->t_idx.p, ctx->t_d2.p, ctx->stack_ovf.p, d_cnt, ctx->stream));
    if (n_ties) HIP_TRY(hipMemcpyAsync(n_ties, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  } else {
    rc = ensure_stack_ovf(ctx, ((nq + 127) / 128) * 128, &ovf);
    if (rc) return rc;
    HIP_TRY(launch_knn5(T, ctx->t_q.p, (int)nq, ctx->t_idx.p, ctx->t_d2.p, ovf, ctx->stream));
  }
  HIP_TRY(hipMemcpyAsync(idx_out, ctx->t_idx.p, nq * 5 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d2_out, ctx->t_d2.p, nq * 5 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

// Parity tap of the search a map WITHOUT kd-trees is matched through (include/lslam_c.h): the wide probe on the resident cell
// grids, every cell within the acceptance gate, one wavefront per query.  Builds no tree.
int lslam_debug_knn5_wide(lslam_ctx *ctx, int which_map, const void *queries, size_t nq, size_t stride_bytes, int32_t nf_margin,
                          int32_t *idx_out, float *d2_out, uint8_t *undecided_out) {
  int rc = check_ctx(ctx, true);
  if (rc) return rc;
  if (!ctx->have_map || ctx->cube_mode) { set_err("no whole-map search structure set"); return LSLAM_ERR_NO_MAP; }
  if ((which_map != 0 && which_map != 1) || stride_bytes < 12 || (stride_bytes & 3) ||
      (nq && (!queries || !idx_out || !d2_out || !undecided_out)) || nq > 0x0FFFFFFFu) {
    set_err("bad knn5 arguments");
    return LSLAM_ERR_INVALID;
  }
  if (nq == 0) return LSLAM_OK;
  if (!ctx->trees_pending) {  // a map with trees: its grids are made on first use
    rc = ensure_grid(ctx, ctx->env_grid_cell > 0.0f ? ctx->env_grid_cell : 0.0f);
    if (rc) return rc;
  }
  const CellGrid &G = which_map ? ctx->ks.view : ctx->kc.view;
  if (!G.cell_start) { set_err("this map has no cell grid (status %d)", ctx->grid_status); return LSLAM_ERR_INVALID; }
  std::vector<float4> q;
  pack_cloud(queries, nq, stride_bytes, q);
  HIP_TRY(ctx->t_q.reserve(nq));
  HIP_TRY(ctx->t_idx.reserve(nq * 5));
  HIP_TRY(ctx->t_d2.reserve(nq * 5));
  HIP_TRY(ctx->t_flags.reserve(nq + 1));
  HIP_TRY(hipMemcpyAsync(ctx->t_q.p, q.data(), nq * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(launch_knn5_wide(G, ctx->t_q.p, (int)nq, nf_margin ? GRID_NF_PRUNE_SLACK_WIDE : 0.0f, ctx->t_idx.p, ctx->t_d2.p, ctx->t_flags.p, ctx->stream));
  HIP_TRY(hipMemcpyAsync(idx_out, ctx->t_idx.p, nq * 5 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d2_out, ctx->t_d2.p, nq * 5 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(undecided_out, ctx->t_flags.p, nq, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));  // (q is a local)
  return LSLAM_OK;
}

// Parity tap of lslam_sort.hip (include/lslam_c.h)
int lslam_debug_sort_pairs(lslam_ctx *ctx, const uint64_t *keys, const uint32_t *values, size_t n, uint64_t *keys_out,
                           uint32_t *values_out) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (n > SMALL_SORT_MAX || (n && (!keys || !values || !keys_out || !values_out))) {
    set_err("bad sort arguments (n <= %zu)", (size_t)SMALL_SORT_MAX);
    return LSLAM_ERR_INVALID;
  }
  if (n == 0) return LSLAM_OK;
  // [keys in | keys out | values in | values out | scratch] in the taps' index buffer
  const size_t words = 4 * n + 2 * n + (small_sort_tmp_bytes(n) + 3) / 4 + 16;
  HIP_TRY(ctx->t_idx.reserve(words));
  uint64_t *k0 = reinterpret_cast<uint64_t *>(ctx->t_idx.p), *k1 = k0 + n;
  uint32_t *v0 = reinterpret_cast<uint32_t *>(k1 + n), *v1 = v0 + n;
  void *tmp = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(v1 + n) + 15) & ~(uintptr_t)15);
  HIP_TRY(hipMemcpyAsync(k0, keys, n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(v0, values, n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(small_sort_pairs(ctx->stream, k0, k1, v0, v1, n, tmp));
  HIP_TRY(hipMemcpyAsync(keys_out, k1, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(values_out, v1, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

#ifdef LSLAM_PACKET_STATS
// profiling build only: queries (map frame, caller's order = packet order) -> per-wave counters [nwaves][8]
int lslam_debug_packet_stats(lslam_ctx *ctx, int which_map, const float *q_xyzw, size_t nq, uint32_t *out) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  const TreeView &T = which_map ? ctx->ts.view : ctx->tc.view;
  HIP_TRY(ctx->t_q.reserve(nq));
  HIP_TRY(ctx->t_idx.reserve(((nq + 63) / 64) * 8 + 64));
  HIP_TRY(hipMemcpyAsync(ctx->t_q.p, q_xyzw, nq * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(launch_packet_stats(T, ctx->t_q.p, (int)nq, reinterpret_cast<unsigned *>(ctx->t_idx.p), ctx->stream));
  HIP_TRY(hipMemcpyAsync(out, ctx->t_idx.p, ((nq + 63) / 64) * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}
#endif

int lslam_sweep(lslam_ctx *ctx, const float pose[6], int32_t jtj_mode, int32_t *idx_out,
                float *d2_out, float *coeff_out, uint8_t *flags_out, float *sums_out) {
  return lslam_sweep_ex(ctx, pose, jtj_mode, LSLAM_SEARCH_LANE, idx_out, d2_out, coeff_out, flags_out, sums_out);
}

int lslam_residuals(lslam_ctx *ctx, const float pose[6], float *coeff_out, uint8_t *valid_out, float *JtJ27_out) {
  float sums[32];
  const int rc = lslam_sweep_ex(ctx, pose, 1, LSLAM_SEARCH_LANE, nullptr, nullptr, coeff_out, valid_out, JtJ27_out ? sums : nullptr);
  if (rc == LSLAM_OK && JtJ27_out) std::memcpy(JtJ27_out, sums, 27 * sizeof(float));
  return rc;
}

int lslam_scanmatch_batch(lslam_ctx *ctx, int32_t n_problems, float *poses, const lslam_opts *opts, lslam_stats *stats) {
  return lslam_scanmatch_run_batch(ctx, n_problems, poses, opts, stats);
}

int lslam_sweep_ex(lslam_ctx *ctx, const float pose[6], int32_t jtj_mode, int32_t search_mode, int32_t *idx_out,
                   float *d2_out, float *coeff_out, uint8_t *flags_out, float *sums_out) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (!ctx->have_map) { set_err("no map set"); return LSLAM_ERR_NO_MAP; }
  if (!ctx->have_scan) { set_err("no scan set"); return LSLAM_ERR_NO_SCAN; }
  if (!pose) { set_err("null pose"); return LSLAM_ERR_INVALID; }
  if (ctx->n_prob != 1) { set_err("lslam_sweep is a single-scan tap"); return LSLAM_ERR_INVALID; }
  const size_t N = ctx->n_points;
  init_state(*ctx->h_state, pose);
  HIP_TRY(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(GNState), hipMemcpyHostToDevice, ctx->stream));
  SweepArgs sa;
  fill_sweep_args(ctx, sa);
  rc = ensure_stack_ovf(ctx, (size_t)sa.nb_total * SWEEP_BLOCK, &sa.stack_ovf);
  if (rc) return rc;
  sa.stack_mode = resolve_stack_mode(ctx, search_mode);
  const bool carried = (search_mode & LSLAM_SWEEP_CARRIED) != 0, first = (search_mode & LSLAM_SWEEP_FIRST) != 0;
  search_mode &= 0xFF;
  if (first && (carried || search_mode != LSLAM_SEARCH_GRID)) {
    set_err("LSLAM_SWEEP_FIRST needs LSLAM_SEARCH_GRID and excludes LSLAM_SWEEP_CARRIED");
    return LSLAM_ERR_INVALID;
  }
  if (carried && (search_mode != LSLAM_SEARCH_GRID || !ctx->prev_q.p || !ctx->grid_state_valid)) {
    set_err("LSLAM_SWEEP_CARRIED needs LSLAM_SEARCH_GRID and the state a grid-sweep lslam_scanmatch_run* left on this scan");
    return LSLAM_ERR_INVALID;
  }
  if (sa.stack_mode == SWEEP_STACK_SHALLOW && !ctx->cube_mode) {  // the batch kernel's stack shape on this (unbounded) tap
    HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words((size_t)std::max(sa.nb_total, 1) * SWEEP_BLOCK, std::max(ctx->tc.depth, ctx->ts.depth))));
    sa.stack_ovf = ctx->stack_ovf.p;
  }
  if (search_mode == LSLAM_SEARCH_PACKET) {
    rc = ensure_packet_nodes(ctx);
    if (rc) return rc;
    if (ctx->cube_mode || !ctx->tc.view.pn || !ctx->ts.view.pn) { set_err("this map has no packet-search nodes"); return LSLAM_ERR_INVALID; }
    sa.tc = ctx->tc.view;
    sa.ts = ctx->ts.view;
    HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words((size_t)std::max(sa.nb_total, 1) * SWEEP_BLOCK)));
    sa.stack_ovf = ctx->stack_ovf.p;
    sa.packet = 1;
  }
  if (search_mode == LSLAM_SEARCH_GRID) {  // the grid sweep on this (unbounded) tap: probe + proof, tree search for the rest
    rc = ensure_grid(ctx, ctx->env_grid_cell > 0.0f ? ctx->env_grid_cell : 0.0f);
    if (rc) return rc;
    if (ctx->cube_mode || !ctx->kc.view.cell_start || !ctx->ks.view.cell_start) { set_err("this map has no cell grid (status %d)", ctx->grid_status); return LSLAM_ERR_INVALID; }
    sa.kc = ctx->kc.view;
    sa.ks = ctx->ks.view;
    sa.grid = 1;
    sa.grid_hint = ctx->prev_lb.p;
    HIP_TRY(ctx->stack_ovf.reserve(stack_ovf_words((size_t)std::max(sa.nb_total, 1) * SWEEP_BLOCK, std::max(ctx->tc.depth, ctx->ts.depth))));
    sa.stack_ovf = ctx->stack_ovf.p;
    sa.need_list = ctx->need_list.p;
    sa.need_cnt = ctx->need_cnt.p;
    sa.groups = ctx->groups.p;
    sa.n_groups = (int32_t)ctx->h_groups.size();
    sa.group_block_base = 0;
    HIP_TRY(hipMemsetAsync(ctx->cert_count.p, 0, 6 * sizeof(int32_t), ctx->stream));
    if (carried || first) {  // a sweep of the production loop, kernel for kernel (include/lslam_c.h LSLAM_SWEEP_CARRIED / _FIRST)
      HIP_TRY(ctx->prev_q.reserve(std::max<size_t>(N, 1)));
      sa.bounded = 1;
      sa.prev_valid = carried ? 1 : 0;
      sa.prev_q = ctx->prev_q.p;
      sa.grid_clip_margin = GRID_CLIP_MARGIN_MIN;
      ctx->grid_state_valid = true;  // what this sweep leaves is the state the next sweep of the loop would find
    }
  }
  const bool taps = idx_out || d2_out || coeff_out || flags_out;
  if (taps) {
    HIP_TRY(ctx->t_idx.reserve(N * 5 + 1));
    HIP_TRY(ctx->t_d2.reserve(N * 5 + 1));
    HIP_TRY(ctx->t_coeff.reserve(N + 1));
    HIP_TRY(ctx->t_flags.reserve(N + 1));
    sa.idx_out = ctx->t_idx.p;
    sa.d2_out = ctx->t_d2.p;
    sa.coeff_out = ctx->t_coeff.p;
    sa.flags_out = ctx->t_flags.p;
  }
  HIP_TRY(sweep_launch(ctx, sa, jtj_mode));
  SolveArgs so{};
  so.states = ctx->d_state;
  so.partials = ctx->partials.p;
  so.probs = ctx->probs.p;
  so.n_prob = 1;
  so.reduce_only = 1;
  HIP_TRY(launch_solve(so, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState), hipMemcpyDeviceToHost, ctx->stream));
  if (N) {
    if (idx_out) HIP_TRY(hipMemcpyAsync(idx_out, ctx->t_idx.p, N * 5 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (d2_out) HIP_TRY(hipMemcpyAsync(d2_out, ctx->t_d2.p, N * 5 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (coeff_out) HIP_TRY(hipMemcpyAsync(coeff_out, ctx->t_coeff.p, N * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (flags_out) HIP_TRY(hipMemcpyAsync(flags_out, ctx->t_flags.p, N, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (sums_out) {
    const GNState &g = *ctx->h_state;
    for (int i = 0; i < 27; ++i) sums_out[i] = (float)g.sums[i];
    sums_out[27] = (float)g.sums[COL_ROWS];
    sums_out[28] = (float)(g.sums[COL_LINE] + g.sums[COL_PLANE]);
    sums_out[29] = (float)g.sums[COL_SCORE];
  }
  return LSLAM_OK;
}

int lslam_gn_step(lslam_ctx *ctx, const float AtA[36], const float Atb[6], int32_t iter,
                  float pose[6], float matP[36], int32_t *degenerate, float delta_r_abort,
                  float delta_t_abort, float x_out[6], float *delta_r, float *delta_t,
                  int32_t *converged) {
  int rc = check_ctx(ctx);
  if (rc) return rc;
  if (!AtA || !Atb || !pose || !matP || !degenerate) { set_err("null argument"); return LSLAM_ERR_INVALID; }
  init_state(*ctx->h_state, pose);
  ctx->h_state->iter = iter;
  ctx->h_state->loop_iter = iter;
  ctx->h_state->degenerate = *degenerate;
  std::memcpy(ctx->h_state->matP, matP, sizeof(float) * 36);
  HIP_TRY(ctx->t_small.reserve(64));
  float tmp[42];
  std::memcpy(tmp, AtA, sizeof(float) * 36);
  std::memcpy(tmp + 36, Atb, sizeof(float) * 6);
  HIP_TRY(hipMemcpyAsync(ctx->t_small.p, tmp, sizeof(tmp), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->d_state, ctx->h_state, sizeof(GNState), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));  // tmp is a local
  HIP_TRY(launch_gn_step_tap(ctx->d_state, ctx->t_small.p, ctx->t_small.p + 36, delta_r_abort,
                             delta_t_abort, 100.0f, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->h_state, ctx->d_state, sizeof(GNState), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const GNState &g = *ctx->h_state;
  for (int i = 0; i < 6; ++i) pose[i] = g.pose[i];
  std::memcpy(matP, g.matP, sizeof(float) * 36);
  *degenerate = g.degenerate;
  if (x_out) for (int i = 0; i < 6; ++i) x_out[i] = g.x[i];
  if (delta_r) *delta_r = g.delta_r;
  if (delta_t) *delta_t = g.delta_t;
  if (converged) *converged = g.converged;
  return LSLAM_OK;
}

}  // extern "C"
