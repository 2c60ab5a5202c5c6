// What a vector-ALU wave-instruction costs on MI355X (gfx950): N independent instructions per wave, timed with s_memtime
// (shader cycles), at 1, 2 and 4 waves per SIMD, with all 64 lanes and with the lower 32 only.  bench.py prices the sweep
// kernel's SQ_INSTS_VALU with these figures (roofline.valu_issue_frac) instead of assuming four cycles per instruction.
//   hipcc --offload-arch=gfx950 -O2 -o build/ubench_valu tools/ubench_valu.hip && build/ubench_valu > profiles/r06_ubench_valu.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int REPS = 4096;  // x 8 instructions per repetition (~100 us per launch: the workgroups' start-up stagger does not count)
enum { OP_FMA_F32 = 0, OP_MED3_U32, OP_CNDMASK, OP_ADD_U32, OP_FMA_F64, OP_MUL_F32, OP_AND_OR, OP_FMAC_F32, OP_CNDMASK_SGPR, N_OPS };
static const char *OP_NAME[N_OPS] = {"v_fma_f32", "v_med3_u32", "v_cndmask_b32 (vcc)", "v_add_u32", "v_fma_f64", "v_mul_f32", "v_and_or_b32", "v_fmac_f32", "v_cndmask_b32 (sgpr pair)"};

#define R8(S) S S S S S S S S
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, float *sink, int lanes) {
  const int lane = threadIdx.x & 63;
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  double d0 = threadIdx.x, d1 = 1., d2 = 2., d3 = 3., d4 = 4., d5 = 5., d6 = 6., d7 = 7.;
  const float m = 1.0000001f, c = 1.0e-9f;
  const double md = 1.0000001, cd = 1.0e-9;
  unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  if (lane < lanes) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int r = 0; r < REPS; ++r) {
      if (OP == OP_FMA_F32)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      else if (OP == OP_MUL_F32)
        asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                     "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
      else if (OP == OP_MED3_U32)
        asm volatile("v_med3_u32 %0, %0, %8, %9\n\tv_med3_u32 %1, %1, %8, %9\n\tv_med3_u32 %2, %2, %8, %9\n\tv_med3_u32 %3, %3, %8, %9\n\t"
                     "v_med3_u32 %4, %4, %8, %9\n\tv_med3_u32 %5, %5, %8, %9\n\tv_med3_u32 %6, %6, %8, %9\n\tv_med3_u32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      else if (OP == OP_AND_OR)
        asm volatile("v_and_or_b32 %0, %0, %8, %9\n\tv_and_or_b32 %1, %1, %8, %9\n\tv_and_or_b32 %2, %2, %8, %9\n\tv_and_or_b32 %3, %3, %8, %9\n\t"
                     "v_and_or_b32 %4, %4, %8, %9\n\tv_and_or_b32 %5, %5, %8, %9\n\tv_and_or_b32 %6, %6, %8, %9\n\tv_and_or_b32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      else if (OP == OP_CNDMASK)
        asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                     "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
      else if (OP == OP_FMAC_F32)
        asm volatile("v_fmac_f32 %0, %8, %9\n\tv_fmac_f32 %1, %8, %9\n\tv_fmac_f32 %2, %8, %9\n\tv_fmac_f32 %3, %8, %9\n\t"
                     "v_fmac_f32 %4, %8, %9\n\tv_fmac_f32 %5, %8, %9\n\tv_fmac_f32 %6, %8, %9\n\tv_fmac_f32 %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      else if (OP == OP_CNDMASK_SGPR)
        asm volatile("v_cndmask_b32 %0, %0, %8, %9\n\tv_cndmask_b32 %1, %1, %8, %9\n\tv_cndmask_b32 %2, %2, %8, %9\n\tv_cndmask_b32 %3, %3, %8, %9\n\t"
                     "v_cndmask_b32 %4, %4, %8, %9\n\tv_cndmask_b32 %5, %5, %8, %9\n\tv_cndmask_b32 %6, %6, %8, %9\n\tv_cndmask_b32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(0x5555555555555555ull));
      else if (OP == OP_ADD_U32)
        asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                     "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
      else if (OP == OP_FMA_F64)
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(md), "v"(cd));
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  }
  if (lane == 0) {
    out[(size_t)blockIdx.x * 32 + 2 * (threadIdx.x >> 6)] = t1 - t0;
    out[(size_t)blockIdx.x * 32 + 2 * (threadIdx.x >> 6) + 1] = r1 - r0;  // 100 MHz
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) == 12345.678f) sink[0] = a0;  // keep the chains
}

template <int OP>
static int run(int cus, unsigned long long *d_out, float *d_sink, bool &first) {
  std::vector<unsigned long long> h;
  for (int lanes : {64, 32})
    for (int wps : {1, 2, 4}) {
      const int blocks = cus;  // ONE workgroup per CU of 4 wps wavefronts: wps per SIMD, resident together by construction
      CK(hipMemset(d_out, 0, (size_t)blocks * 32 * 8));
      hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256 * wps), 0, 0, d_out, d_sink, lanes);  // warm-up (clocks, code)
      hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256 * wps), 0, 0, d_out, d_sink, lanes);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> raw((size_t)blocks * 32);
      CK(hipMemcpy(raw.data(), d_out, raw.size() * 8, hipMemcpyDeviceToHost));
      h.clear();
      std::vector<unsigned long long> hr;
      for (size_t i = 0; i < raw.size(); i += 2) {
        if (raw[i] == 0) continue;  // (slots of wavefronts this configuration does not have)
        h.push_back(raw[i]);
        hr.push_back(raw[i + 1]);
      }
      std::sort(h.begin(), h.end());
      std::sort(hr.begin(), hr.end());
      const double med = (double)h[h.size() / 2], med_ns = 10.0 * (double)hr[hr.size() / 2];
      const double per_wave = med / (REPS * 8.0);          // s_memtime ticks between two instructions of ONE wave
      const double per_simd = per_wave / wps;              // ticks of SIMD time per wave-instruction, wps waves sharing it
      printf("%s  {\"op\": \"%s\", \"lanes\": %d, \"waves_per_simd\": %d, \"ticks_per_instruction_of_a_wave\": %.3f, \"simd_ticks_per_wave_instruction\": %.3f, "
             "\"simd_ns_per_wave_instruction\": %.4f, \"memtime_ticks_per_us\": %.1f}",
             first ? "" : ",\n", OP_NAME[OP], lanes, wps, per_wave, per_simd, med_ns / (REPS * 8.0) / wps, 1e3 * med / med_ns);
      first = false;
    }
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long *d_out;
  float *d_sink;
  CK(hipMalloc((void **)&d_out, (size_t)cus * 32 * 8));
  CK(hipMalloc((void **)&d_sink, 64));
  printf("{\"device\": \"%s\", \"cus\": %d, \"method\": \"%d independent instructions per wave (8 chains), s_memtime and s_memrealtime (100 MHz) around them, medians over the waves; "
         "one workgroup of 4 x waves_per_simd wavefronts per CU\", \"results\": [\n", prop.name, cus, REPS * 8);
  bool first = true;
  if (run<OP_FMA_F32>(cus, d_out, d_sink, first)) return 1;
  if (run<OP_MUL_F32>(cus, d_out, d_sink, first)) return 1;
  if (run<OP_MED3_U32>(cus, d_out, d_sink, first)) return 1;
  if (run<OP_AND_OR>(cus, d_out, d_sink, first)) return 1;
  if (run<OP_CNDMASK>(cus, d_out, d_sink, first)) return 1;
  if (run<OP_ADD_U32>(cus, d_out, d_sink, first)) return 1;
  if (run<OP_FMA_F64>(cus, d_out, d_sink, first)) return 1;
  if (run<OP_FMAC_F32>(cus, d_out, d_sink, first)) return 1;
  if (run<OP_CNDMASK_SGPR>(cus, d_out, d_sink, first)) return 1;
  printf("\n]}\n");
  return 0;
}
