// Dependent-launch latency on MI355X: N tiny kernels back to back in a stream against the same N captured in a hipGraph.
//   hipcc --offload-arch=gfx950 -O2 -o build/ubench_launch tools/ubench_launch.hip && build/ubench_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(double *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0000001 + 1e-9;
}
int main() {
  const int n = 30000, N = 400;
  double *d;
  CK(hipMalloc(&d, n * sizeof(double)));
  CK(hipMemset(d, 0, n * sizeof(double)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto run_stream = [&]() { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3((n + 255) / 256), dim3(256), 0, s, d, n); };
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    run_stream();
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("stream: %d dependent launches %.1f us  (%.2f us each)\n", N, us, us / N);
  }
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  run_stream();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("graph:  %d dependent launches %.1f us  (%.2f us each)\n", N, us, us / N);
  }
  return 0;
}
