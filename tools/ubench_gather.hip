// ubench_gather.hip -- what the vector memory pipe of one gfx950 CU does with the access
// patterns of a kd-tree walk (profiling tool, not part of the product library).
//
//   throughput: every lane of every wave issues U independent 16-byte (or 4-byte) loads per
//   iteration from a table of `bytes` bytes; pattern = how the 64 lanes of one instruction
//   spread over cache lines (div: 64 random 16-B pieces; g4 / g8 / g16: groups of 4 / 8 / 16
//   lanes read consecutive pieces of one random 64 / 128 / 256-byte run; same: one address).
//   Reported: lane-loads per clock per CU and instruction issue interval.
//
//   latency: one wave chases a dependent chain through a table of `bytes` bytes.
//
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// GROUP lanes read consecutive 16-byte pieces; pieces = table size in float4
template <int GROUP, int U>
__global__ __launch_bounds__(256) void gather16(const float4 *tab, uint32_t pieces, int iters, float *out) {
  const uint32_t lane = threadIdx.x & 63;
  // GROUP == 64: the whole wave reads one contiguous 1 KB run (fully coalesced), a different one per load
  const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
  const uint32_t sub = lane % GROUP;
  float acc = 0.f;
  uint32_t s = mix(gid * 2654435761u + 1u);
  for (int it = 0; it < iters; ++it) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      const uint32_t base = ((s >> 9) & (pieces / GROUP - 1)) * GROUP;
      v[u] = tab[base + sub];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void gather4(const float *tab, uint32_t words, int iters, float *out) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  uint32_t s = mix(gid * 2654435761u + 1u);
  for (int it = 0; it < iters; ++it) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      v[u] = tab[(s >> 9) & (words - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc == 123.456f) out[0] = acc;
}

// LDS: every lane reads 16 bytes at a random 16-byte slot of a 32 KB LDS array
template <int U>
__global__ __launch_bounds__(256) void lds16(int iters, float *out) {
  __shared__ float4 buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = make_float4(i, 1, 2, 3);
  __syncthreads();
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  uint32_t s = mix(gid * 2654435761u + 1u);
  for (int it = 0; it < iters; ++it) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      v[u] = buf[(s >> 9) & 2047];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;
}

__global__ void chase(const uint32_t *next, int steps, uint32_t *out, long long *cycles) {
  uint32_t p = threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; ++i) p = next[p];
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <typename F>
float time_ms(F f, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a));
    f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  int cus = 256, clk_khz = 2400000;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  printf("CUs %d clock %.0f MHz\n", cus, clk_khz / 1e3);
  const double ghz = clk_khz / 1e6;
  float *out;
  CK(hipMalloc(&out, 1 << 20));
  const size_t sizes[] = {16u << 10, 1u << 20, 32u << 20, 128u << 20};
  const char *names[] = {"16KB(L1)", "1MB(L2)", "32MB(L2agg/MALL)", "128MB(MALL)"};
  const int blocks = cus * 8, iters = 64;
  constexpr int U = 8;
  for (int si = 0; si < 4; ++si) {
    float4 *tab;
    CK(hipMalloc(&tab, sizes[si]));
    CK(hipMemset(tab, 0, sizes[si]));
    const uint32_t pieces = (uint32_t)(sizes[si] / 16);
    auto report = [&](const char *pat, float ms, double lane_loads_per_wave_instr) {
      const double instr = (double)blocks * 4 * iters * U;                 // wave-level load instructions
      const double cyc_per_cu = ms * 1e-3 * ghz * 1e9;                      // cycles each CU ran
      const double instr_per_cu = instr / cus;
      printf("  %-18s %-6s %8.3f ms  %7.1f clk/instr/CU  %6.2f lane-loads/clk/CU  %7.1f GB/s/CU\n", names[si], pat, ms,
             cyc_per_cu / instr_per_cu, instr_per_cu * lane_loads_per_wave_instr / cyc_per_cu,
             instr_per_cu * lane_loads_per_wave_instr * 16 / (ms * 1e-3) / 1e9);
    };
    report("div", time_ms([&] { hipLaunchKernelGGL((gather16<1, U>), dim3(blocks), dim3(256), 0, 0, tab, pieces, iters, out); }), 64);
    report("g2", time_ms([&] { hipLaunchKernelGGL((gather16<2, U>), dim3(blocks), dim3(256), 0, 0, tab, pieces, iters, out); }), 64);
    report("g4", time_ms([&] { hipLaunchKernelGGL((gather16<4, U>), dim3(blocks), dim3(256), 0, 0, tab, pieces, iters, out); }), 64);
    report("g8", time_ms([&] { hipLaunchKernelGGL((gather16<8, U>), dim3(blocks), dim3(256), 0, 0, tab, pieces, iters, out); }), 64);
    report("g16", time_ms([&] { hipLaunchKernelGGL((gather16<16, U>), dim3(blocks), dim3(256), 0, 0, tab, pieces, iters, out); }), 64);
    report("coal", time_ms([&] { hipLaunchKernelGGL((gather16<64, U>), dim3(blocks), dim3(256), 0, 0, tab, pieces, iters, out); }), 64);
    {
      const float ms = time_ms([&] { hipLaunchKernelGGL((gather4<U>), dim3(blocks), dim3(256), 0, 0, (const float *)tab, pieces * 4, iters, out); });
      const double instr_per_cu = (double)blocks * 4 * iters * U / cus, cyc = ms * 1e-3 * ghz * 1e9;
      printf("  %-18s %-6s %8.3f ms  %7.1f clk/instr/CU  %6.2f lane-loads/clk/CU (4-byte loads)\n", names[si], "div4", ms, cyc / instr_per_cu,
             instr_per_cu * 64 / cyc);
    }
    // dependent chain latency: random cycle through the table (one wave; lane 0's time)
    {
      const uint32_t n = (uint32_t)(sizes[si] / 64);  // one hop per 64-byte piece
      std::vector<uint32_t> perm(n), next((size_t)n * 16, 0);
      for (uint32_t i = 0; i < n; ++i) perm[i] = i;
      uint32_t s = 12345;
      for (uint32_t i = n - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; std::swap(perm[i], perm[mix(s) % (i + 1)]); }
      for (uint32_t i = 0; i < n; ++i) next[(size_t)perm[i] * 16] = perm[(i + 1) % n] * 16;
      uint32_t *d_next, *d_out;
      long long *d_cyc, h_cyc = 0;
      CK(hipMalloc(&d_next, next.size() * 4));
      CK(hipMalloc(&d_out, 256));
      CK(hipMalloc(&d_cyc, 8));
      CK(hipMemcpy(d_next, next.data(), next.size() * 4, hipMemcpyHostToDevice));
      const int steps = 4096;
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d_next, steps, d_out, d_cyc);
      CK(hipMemcpy(&h_cyc, d_cyc, 8, hipMemcpyDeviceToHost));
      printf("  %-18s chase  %.0f cycles/hop (shader clock counter; %.0f ns at 100 MHz ticks if that is what it counts)\n", names[si],
             (double)h_cyc / steps, (double)h_cyc / steps * 10.0);
      CK(hipFree(d_next)); CK(hipFree(d_out)); CK(hipFree(d_cyc));
    }
    CK(hipFree(tab));
  }
  {
    const float ms = time_ms([&] { hipLaunchKernelGGL((lds16<U>), dim3(blocks), dim3(256), 0, 0, iters, out); });
    const double instr_per_cu = (double)blocks * 4 * iters * U / cus, cyc = ms * 1e-3 * ghz * 1e9;
    printf("  LDS 32KB ds_read_b128 random: %8.3f ms  %7.1f clk/instr/CU  %6.2f lane-loads/clk/CU\n", ms, cyc / instr_per_cu, instr_per_cu * 64 / cyc);
  }
  return 0;
}
