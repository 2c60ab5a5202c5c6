// C++ drop-in check of include/lslam_pipeline.hpp: the LaserOdometry and LaserMapping mirrors driven from C++
// through the C ABI.  Reads the four feature clouds of N consecutive sweeps from a file written by the test
// (uint32 count + count x {x,y,z,intensity} floats, four clouds per sweep), feeds them through
// LaserOdometry::process -> LaserMapping::process and prints one "POSE" line per mapped sweep, which the test
// compares with the Python mirrors of the same state machines (same ABI calls: same bits).
#include <cstdint>
#include <cstdio>
#include <vector>

#include "lslam_pipeline.hpp"
#include "lslam_scan_match.hpp"

struct Pt { float x, y, z, intensity; };
struct Cloud { std::vector<Pt> points; };

static bool read_cloud(FILE *f, Cloud &c) {
  uint32_t n = 0;
  if (std::fread(&n, 4, 1, f) != 1) return false;
  c.points.resize(n);
  return n == 0 || std::fread(c.points.data(), sizeof(Pt), n, f) == n;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  lidar_slam::ScanMatch sm(10);  // owns the context; never throws
  if (!sm.ok()) {
    std::fprintf(stderr, "backend unavailable: %s\n", sm.initError().c_str());
    return 1;
  }
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  lidar_slam::LaserOdometry odo(sm.context());
  lidar_slam::LaserMapping mapping(sm.context(), 21, 21, 11);
  Cloud sharp, less_sharp, flat, less_flat;
  int sweep = 0;
  while (read_cloud(f, sharp) && read_cloud(f, less_sharp) && read_cloud(f, flat) && read_cloud(f, less_flat)) {
    const bool moved = odo.process(sharp, less_sharp, flat, less_flat);
    if (moved) {
      if (!mapping.process(odo.lastCornerCloud(), odo.lastSurfaceCloud(), odo.Tsum())) {
        std::fprintf(stderr, "mapping failed: %s\n", mapping.lastError().c_str());
        return 1;
      }
      const float *T = mapping.lidarMapped(), *S = odo.Tsum();
      std::printf("POSE %d", sweep);
      for (int i = 0; i < 12; ++i) std::printf(" %a", (double)T[i]);
      for (int i = 0; i < 12; ++i) std::printf(" %a", (double)S[i]);
      std::printf(" %d %d\n", odo.lastStats().iterations, mapping.lastStats().iterations);
    } else if (!odo.lastError().empty()) {
      std::fprintf(stderr, "odometry failed: %s\n", odo.lastError().c_str());
      return 1;
    }
    ++sweep;
  }
  std::fclose(f);
  std::printf("OK sweeps %d\n", sweep);
  return 0;
}
