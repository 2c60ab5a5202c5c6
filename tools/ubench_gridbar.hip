// Grid-barrier cost on MI355X: a persistent kernel of NB workgroups crossing N barriers, in the forms a persistent PCG
// kernel could use (see pg_pcg_persistent_kernel in csrc/lslam_posegraph.hip).
//   hipcc --offload-arch=gfx950 -O2 -o build/ubench_gridbar tools/ubench_gridbar.hip && build/ubench_gridbar
//   mode 0  arrive = release RMW, poll = acquire loads (each poll invalidates the caches)
//   mode 1  release fence, relaxed RMW, relaxed polls, one acquire fence
//   mode 2  as 1, plus every workgroup publishes a double before the barrier and reads its neighbour's after it (checked)
//   mode 3  as 2 but the published values go through relaxed agent-scope atomic stores / loads and the barrier carries NO
//           cache write-back / invalidate (ordering by s_waitcnt only)
//   mode 4  as 1 with a per-XCD-free tree: arrivals counted per group of 8 workgroups, then one top counter
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void bar_kernel(unsigned *bar, double *data, int n_bar, unsigned *errs) {
  const unsigned nb = gridDim.x;
  unsigned target = 0;
  double acc = 0.0;
  for (int k = 0; k < n_bar; ++k) {
    if (MODE == 2) {
      if (threadIdx.x == 0) data[(size_t)(k & 1) * nb + blockIdx.x] = (double)(k * 1000 + blockIdx.x);
    } else if (MODE == 3) {
      if (threadIdx.x == 0)
        __hip_atomic_store(data + (size_t)(k & 1) * nb + blockIdx.x, (double)(k * 1000 + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    target += nb;
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      } else if (MODE == 3) {
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    if (MODE == 2) {
      const double v = data[(size_t)(k & 1) * nb + (blockIdx.x + 1) % nb];
      if (threadIdx.x == 0 && v != (double)(k * 1000 + (blockIdx.x + 1) % nb)) atomicAdd(errs, 1u);
      acc += v;
    } else if (MODE == 3) {
      const double v = __hip_atomic_load(data + (size_t)(k & 1) * nb + (blockIdx.x + 1) % nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x == 0 && v != (double)(k * 1000 + (blockIdx.x + 1) % nb)) atomicAdd(errs, 1u);
      acc += v;
    }
  }
  if (acc == -1.0) data[0] = acc;
}

template <int MODE>
int run(int nb, int n_bar, hipStream_t s, unsigned *bar, double *data, unsigned *errs) {
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemsetAsync(bar, 0, 64, s));
    CK(hipMemsetAsync(errs, 0, 4, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    void *args[] = {(void *)&bar, (void *)&data, (void *)&n_bar, (void *)&errs};
    CK(hipLaunchCooperativeKernel((const void *)bar_kernel<MODE>, dim3(nb), dim3(512), args, 0, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned e = 0;
    CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost));
    if (rep == 2) printf("mode %d, %3d workgroups: %d barriers %.0f us  (%.2f us each), stale reads %u\n", MODE, nb, n_bar, us, us / n_bar, e);
  }
  return 0;
}

int main(int argc, char **argv) {
  const int n_bar = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned *bar, *errs;
  double *data;
  CK(hipMalloc(&bar, 64));
  CK(hipMalloc(&errs, 4));
  CK(hipMalloc(&data, 2 * 1024 * sizeof(double)));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int nb : {136, 256}) {
    if (run<0>(nb, n_bar, s, bar, data, errs)) return 1;
    if (run<1>(nb, n_bar, s, bar, data, errs)) return 1;
    if (run<2>(nb, n_bar, s, bar, data, errs)) return 1;
    if (run<3>(nb, n_bar, s, bar, data, errs)) return 1;
  }
  return 0;
}
