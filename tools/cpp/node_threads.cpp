// The per-sweep chain as the reference runs it -- scan registration, odometry and mapping are three nodelets with their own
// threads (nodelets.xml; LaserOdometry.cpp spin(), LaserMapping.cpp:27-37) joined by topics -- written the way a maintainer would
// write it against this library: three std::threads, a context each (one call in flight per context), bounded queues between
// them, the C ABI and the header-only mirrors of include/.  Registration -> odometry carries feature sets in HBM (a pool of
// them goes round), odometry -> mapping the two last clouds.  No interpreter, no GIL: bench.py's Python threads measure
// Python's lock as much as the library (sweep_pipeline.*.node_threads_python).
//   input   a file written by bench.py: uint32 rings, float lower_deg, upper_deg, uint32 sweeps, then per sweep uint32 n and
//           n x {x, y, z, *} floats (a raw driver cloud in arrival order)
//   output  one line: "NODE_THREADS ms_per_sweep <f> sweeps_timed <n> travelled_m <f> odometry_busy_ms <f> mapping_busy_ms <f>
//           registration_busy_ms <f>"
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "lslam_pipeline.hpp"

template <typename T>
class BoundedQueue {
public:
  explicit BoundedQueue(size_t cap) : _cap(cap) {}
  void put(T v) {
    std::unique_lock<std::mutex> lk(_mu);
    _not_full.wait(lk, [this] { return _q.size() < _cap; });
    _q.push_back(std::move(v));
    _not_empty.notify_one();
  }
  T get() {
    std::unique_lock<std::mutex> lk(_mu);
    _not_empty.wait(lk, [this] { return !_q.empty(); });
    T v = std::move(_q.front());
    _q.pop_front();
    _not_full.notify_one();
    return v;
  }

private:
  size_t _cap;
  std::deque<T> _q;
  std::mutex _mu;
  std::condition_variable _not_full, _not_empty;
};

struct OdomOut {
  bool end = false;
  std::vector<float> corner, surf;
  float Tsum[16];
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  uint32_t rings = 0, sweeps = 0;
  float lo = 0, hi = 0;
  if (std::fread(&rings, 4, 1, f) != 1 || std::fread(&lo, 4, 1, f) != 1 || std::fread(&hi, 4, 1, f) != 1 || std::fread(&sweeps, 4, 1, f) != 1) return 2;
  std::vector<std::vector<float>> raws(sweeps);
  for (auto &r : raws) {
    uint32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 2;
    r.resize(4 * (size_t)n);
    if (n && std::fread(r.data(), 16, n, f) != n) return 2;
  }
  std::fclose(f);
  const uint32_t warm = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 8;
  lslam_ctx *ctx_r = nullptr, *ctx_o = nullptr, *ctx_m = nullptr;
  if (lslam_ctx_create(0, &ctx_r) != LSLAM_OK || lslam_ctx_create(0, &ctx_o) != LSLAM_OK || lslam_ctx_create(0, &ctx_m) != LSLAM_OK) {
    std::fprintf(stderr, "backend unavailable: %s\n", lslam_last_error());
    return 1;
  }
  constexpr int POOL = 5;  // one being filled, two queued, one being consumed, one spare
  lslam_fset *pool_sets[POOL];
  BoundedQueue<lslam_fset *> pool(POOL), q1(2);
  for (int k = 0; k < POOL; ++k) {
    if (lslam_fset_create(ctx_r, &pool_sets[k]) != LSLAM_OK) return 1;
    pool.put(pool_sets[k]);
  }
  BoundedQueue<OdomOut> q2(2);
  lidar_slam::LaserOdometry *odo_p = new lidar_slam::LaserOdometry(ctx_o);
  lidar_slam::LaserMapping *mapping_p = new lidar_slam::LaserMapping(ctx_m, 21, 21, 11);
  lidar_slam::LaserOdometry &odo = *odo_p;
  lidar_slam::LaserMapping &mapping = *mapping_p;
  double t0 = 0, t1 = 0, busy_r = 0, busy_o = 0, busy_m = 0;
  std::atomic<bool> failed(false);
  float last_pose[16] = {0};

  if (argc > 3 && std::strcmp(argv[3], "seq") == 0) {  // diagnostics: the same calls from ONE thread, one sweep after the other
    std::vector<float> reg;
    std::vector<int32_t> ranges(2 * rings);
    t0 = now_s();
    for (uint32_t k = 0; k < sweeps; ++k) {
      if (k == warm) t0 = now_s();
      const size_t n = raws[k].size() / 4;
      reg.resize(4 * n + 4);
      size_t m = 0, counts[4];
      if (lslam_multiscan_register(ctx_r, raws[k].data(), n, 16, lo, hi, (int32_t)rings, 0.1f, reg.data(), n, &m, ranges.data()) < 0 ||
          lslam_extract_features_dev(ctx_r, reg.data(), m, 16, 12, ranges.data(), rings, nullptr, pool_sets[0], counts) < 0) {
        std::fprintf(stderr, "sweep %u: registration failed: %s\n", k, lslam_last_error());
        return 1;
      }
      if (odo.processFeatureSet(pool_sets[0])) {
        if (!mapping.process(odo.lastCornerCloud(), odo.lastSurfaceCloud(), odo.Tsum())) {
          std::fprintf(stderr, "sweep %u: mapping failed: %s\n", k, mapping.lastError().c_str());
          return 1;
        }
      } else if (!odo.lastError().empty()) {
        std::fprintf(stderr, "sweep %u: odometry failed: %s\n", k, odo.lastError().c_str());
        return 1;
      }
    }
    const float *P = mapping.lidarMapped();
    std::printf("SEQUENTIAL ms_per_sweep %.6f sweeps_timed %u travelled_m %.6f\n", 1e3 * (now_s() - t0) / (sweeps - warm), sweeps - warm,
                std::sqrt((double)P[3] * P[3] + (double)P[7] * P[7] + (double)P[11] * P[11]));
    return 0;
  }
  std::thread registration([&] {  // MultiScanRegistration nodelet: raw sweep -> feature clouds (in HBM)
    std::vector<float> reg;
    std::vector<int32_t> ranges(2 * rings);
    for (uint32_t k = 0; k < sweeps && !failed; ++k) {
      const size_t n = raws[k].size() / 4;
      reg.resize(4 * n + 4);
      size_t m = 0;
      double t = now_s();
      if (lslam_multiscan_register(ctx_r, raws[k].data(), n, 16, lo, hi, (int32_t)rings, 0.1f, reg.data(), n, &m, ranges.data()) < 0) {
        std::fprintf(stderr, "registration failed: %s\n", lslam_last_error());
        failed = true;
        break;
      }
      double dt = now_s() - t;
      lslam_fset *fs = pool.get();
      t = now_s();
      size_t counts[4];
      if (lslam_extract_features_dev(ctx_r, reg.data(), m, 16, 12, ranges.data(), rings, nullptr, fs, counts) < 0) {
        std::fprintf(stderr, "extraction failed: %s\n", lslam_last_error());
        failed = true;
        break;
      }
      if (k >= warm) busy_r += dt + (now_s() - t);
      q1.put(fs);
    }
    q1.put(nullptr);
  });
  std::thread odometry([&] {  // LaserOdometry nodelet
    uint32_t k = 0;
    for (;;) {
      lslam_fset *fs = q1.get();
      if (!fs) break;
      const double t = now_s();
      const bool moved = odo.processFeatureSet(fs);
      if (k >= warm) busy_o += now_s() - t;
      ++k;
      pool.put(fs);
      if (!moved) {
        if (!odo.lastError().empty()) {
          std::fprintf(stderr, "odometry failed: %s\n", odo.lastError().c_str());
          failed = true;
        }
        continue;
      }
      OdomOut o;
      o.corner = odo.lastCornerCloud();
      o.surf = odo.lastSurfaceCloud();
      std::memcpy(o.Tsum, odo.Tsum(), sizeof(o.Tsum));
      q2.put(std::move(o));
    }
    OdomOut end;
    end.end = true;
    q2.put(std::move(end));
  });
  std::thread mapper([&] {  // LaserMapping nodelet
    uint32_t k = 1;  // (the first sweep produces no odometry)
    for (;;) {
      OdomOut o = q2.get();
      if (o.end) break;
      // the clock starts where the LAST node takes up its warm-th sweep: the nodes before it run ahead by what the queues hold,
      // and the slow first sweeps (every buffer being sized) must be behind all three
      if (k == warm) t0 = now_s();
      const double t = now_s();
      if (!mapping.process(o.corner, o.surf, o.Tsum)) {
        std::fprintf(stderr, "mapping failed: %s\n", mapping.lastError().c_str());
        failed = true;
      }
      if (k >= warm) busy_m += now_s() - t;
      ++k;
      std::memcpy(last_pose, mapping.lidarMapped(), sizeof(last_pose));
    }
    t1 = now_s();
  });
  registration.join();
  odometry.join();
  mapper.join();
  delete odo_p;
  delete mapping_p;
  for (int k = 0; k < POOL; ++k) lslam_fset_destroy(pool_sets[k]);
  lslam_ctx_destroy(ctx_r);
  lslam_ctx_destroy(ctx_o);
  lslam_ctx_destroy(ctx_m);
  if (failed) return 1;
  const uint32_t n = sweeps - warm;
  const double travelled = std::sqrt((double)last_pose[3] * last_pose[3] + (double)last_pose[7] * last_pose[7] + (double)last_pose[11] * last_pose[11]);
  std::printf("NODE_THREADS ms_per_sweep %.6f sweeps_timed %u travelled_m %.6f odometry_busy_ms %.6f mapping_busy_ms %.6f registration_busy_ms %.6f\n",
              1e3 * (t1 - t0) / n, n, travelled, 1e3 * busy_o / n, 1e3 * busy_m / n, 1e3 * busy_r / n);
  return 0;
}
