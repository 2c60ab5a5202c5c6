// How long the host takes to notice that a stream has drained: hipStreamSynchronize against a spin on hipStreamQuery, after a
// kernel of ~20 us and after a 4-byte device-to-host copy behind it (the pattern of every "read a count back" in the library).
//   hipcc --offload-arch=gfx950 -O2 -o build/ubench_sync tools/ubench_sync.hip && build/ubench_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void spin_kernel(int *out, long cycles) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) out[0] = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int *d; hipMalloc(&d, 4);
  int *h; hipHostMalloc(&h, 4);
  hipStream_t s; hipStreamCreate(&s);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<double> t;
    for (int r = 0; r < 300; ++r) {
      const double a = now_us();
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, d, 2000L);  // 20 us at 100 MHz
      hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, s);
      if (mode == 0) hipStreamSynchronize(s);
      else while (hipStreamQuery(s) == hipErrorNotReady) {}
      t.push_back(now_us() - a);
    }
    std::sort(t.begin(), t.end());
    printf("%s: launch + 20 us kernel + 4-byte D2H + wait: median %.1f us, min %.1f us\n", mode == 0 ? "hipStreamSynchronize" : "spin on hipStreamQuery", t[150], t[0]);
  }
  return 0;
}
