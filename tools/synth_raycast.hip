// synth_raycast.hip -- the synthetic LiDAR of the-cooper-mapper_amd/synth.py (make_scan) as a HIP kernel.
//
// BENCH / TEST INFRASTRUCTURE, not part of the product library: bench.py needs ten thousand
// ray-cast frames for the "10k-frame voxel map" of BASELINE configs[1]/[2] and a few
// hundred distinct 64-ring query scans; numpy takes ~0.5 s per 64-ring scan.  Same world model
// (ground plane + axis-aligned solids), same ring tables, same corner labelling rule as synth.py;
// the range noise comes from a counter-based hash instead of numpy's generator.
//
// build: make -C tools   (hipcc --offload-arch=gfx950 -shared -fPIC)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace {

struct Solid {  // x0, x1, y0, y1, height, kind (0 box, 1 pole, 2 wall)
  float x0, x1, y0, y1, h;
  int32_t kind;
};

struct CastArgs {
  const Solid *solids;
  int32_t n_solids;
  double R[9], t[3];  // sensor -> world
  int32_t rings, az_steps;
  double el_lo, el_hi;  // radians
  float noise_sigma, corner_band, max_range;
  uint32_t seed;
  float4 *pts;      // [rings * az_steps] sensor frame {x, y, z, ring + relTime}
  uint8_t *label;   // 0 no return, 1 surface, 2 corner
};

__device__ inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void raycast_kernel(CastArgs a) {
  extern __shared__ Solid sh[];
  for (int i = threadIdx.x; i < a.n_solids; i += blockDim.x) sh[i] = a.solids[i];
  __syncthreads();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.rings * a.az_steps;
  if (idx >= n) return;
  const int ring = idx / a.az_steps, az_i = idx % a.az_steps;
  const double el = a.rings > 1 ? a.el_lo + (a.el_hi - a.el_lo) * (double)ring / (double)(a.rings - 1) : a.el_lo;
  const double az = 2.0 * M_PI * (double)az_i / (double)a.az_steps;
  const double ds[3] = {cos(el) * cos(az), cos(el) * sin(az), sin(el)};
  double d[3], inv[3];
  for (int r = 0; r < 3; ++r) d[r] = a.R[r * 3] * ds[0] + a.R[r * 3 + 1] * ds[1] + a.R[r * 3 + 2] * ds[2];
  for (int r = 0; r < 3; ++r) inv[r] = fabs(d[r]) > 1e-12 ? 1.0 / d[r] : INFINITY;
  double t_best = d[2] < 0.0 ? -a.t[2] / d[2] : INFINITY;
  int sid = isfinite(t_best) ? -1 : -2, axis = 2;
  for (int k = 0; k < a.n_solids; ++k) {
    const Solid s = sh[k];
    const double lo[3] = {s.x0, s.y0, 0.0}, hi[3] = {s.x1, s.y1, s.h};
    double tn = -INFINITY, tf = INFINITY;
    int ax = 0;
    for (int r = 0; r < 3; ++r) {
      const double t1 = (lo[r] - a.t[r]) * inv[r], t2 = (hi[r] - a.t[r]) * inv[r];
      const double mn = fmin(t1, t2), mx = fmax(t1, t2);
      if (mn > tn) { tn = mn; ax = r; }
      tf = fmin(tf, mx);
    }
    if (tn <= tf && tn > 1e-6 && tn < t_best) { t_best = tn; sid = k; axis = ax; }
  }
  const bool valid = isfinite(t_best) && t_best < (double)a.max_range;
  uint8_t label = 0;
  float4 out = make_float4(0.f, 0.f, 0.f, (float)((double)ring + (double)az_i / (double)a.az_steps * 0.1));
  if (valid) {
    // Box-Muller on two hashed uniforms
    const uint32_t h1 = mix32(a.seed * 0x9E3779B9u + (uint32_t)idx * 2u + 1u), h2 = mix32(h1 ^ 0x85EBCA6Bu ^ ((uint32_t)idx << 1));
    const double u1 = ((double)h1 + 1.0) / 4294967297.0, u2 = (double)h2 / 4294967296.0;
    const double g = sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
    const double r = t_best + (double)a.noise_sigma * g;
    out.x = (float)(ds[0] * r); out.y = (float)(ds[1] * r); out.z = (float)(ds[2] * r);
    label = 1;
    if (sid >= 0) {
      const Solid s = sh[sid];
      const double hx = a.t[0] + d[0] * t_best, hy = a.t[1] + d[1] * t_best;
      const double dx = fmin(fabs(hx - s.x0), fabs(hx - s.x1)), dy = fmin(fabs(hy - s.y0), fabs(hy - s.y1));
      const bool near_edge = s.kind == 0 && axis != 2 && (axis == 0 ? dy : dx) < (double)a.corner_band;
      if (s.kind == 1 || near_edge) label = 2;
    }
  }
  a.pts[idx] = out;
  a.label[idx] = label;
}

struct Pool {
  Solid *solids = nullptr; size_t cap_s = 0;
  float4 *pts = nullptr; uint8_t *label = nullptr; size_t cap_p = 0;
  hipStream_t stream = nullptr;
};
Pool g_pool;

}  // namespace

extern "C" {

// solids: [n][6] floats (x0, x1, y0, y1, h, kind); pose = {rx, ry, rz, tx, ty, tz} (R = Rz Ry Rx);
// out_pts [rings*az_steps][4], out_label [rings*az_steps] on the host.  Returns 0 or a hipError_t.
int synth_raycast(int device, const float *solids, int n_solids, const double pose[6], int rings, int az_steps,
                  double el_lo_deg, double el_hi_deg, float noise_sigma, uint32_t seed, float corner_band, float max_range,
                  float *out_pts, uint8_t *out_label) {
  hipError_t e;
#define T(x) if ((e = (x)) != hipSuccess) return (int)e
  T(hipSetDevice(device));
  Pool &p = g_pool;
  if (!p.stream) T(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
  const size_t n = (size_t)rings * az_steps;
  if ((size_t)n_solids > p.cap_s) {
    if (p.solids) (void)hipFree(p.solids);
    T(hipMalloc((void **)&p.solids, sizeof(Solid) * (size_t)(n_solids + 64)));
    p.cap_s = (size_t)n_solids + 64;
  }
  if (n > p.cap_p) {
    if (p.pts) (void)hipFree(p.pts);
    if (p.label) (void)hipFree(p.label);
    T(hipMalloc((void **)&p.pts, sizeof(float4) * n));
    T(hipMalloc((void **)&p.label, n));
    p.cap_p = n;
  }
  if (sizeof(Solid) * (size_t)n_solids > 150 * 1024) return -1;  // cull the solid list on the host first
  static Solid stage[8192];
  if (n_solids > 8192) return -1;
  for (int i = 0; i < n_solids; ++i) {
    const float *s = solids + 6 * (size_t)i;
    stage[i] = Solid{s[0], s[1], s[2], s[3], s[4], (int32_t)s[5]};
  }
  T(hipMemcpyAsync(p.solids, stage, sizeof(Solid) * (size_t)n_solids, hipMemcpyHostToDevice, p.stream));
  CastArgs a{};
  a.solids = p.solids;
  a.n_solids = n_solids;
  const double cx = cos(pose[0]), sx = sin(pose[0]), cy = cos(pose[1]), sy = sin(pose[1]), cz = cos(pose[2]), sz = sin(pose[2]);
  // Rz * Ry * Rx
  a.R[0] = cz * cy; a.R[1] = cz * sy * sx - sz * cx; a.R[2] = cz * sy * cx + sz * sx;
  a.R[3] = sz * cy; a.R[4] = sz * sy * sx + cz * cx; a.R[5] = sz * sy * cx - cz * sx;
  a.R[6] = -sy;     a.R[7] = cy * sx;                a.R[8] = cy * cx;
  for (int i = 0; i < 3; ++i) a.t[i] = pose[3 + i];
  a.rings = rings;
  a.az_steps = az_steps;
  a.el_lo = el_lo_deg * M_PI / 180.0;
  a.el_hi = el_hi_deg * M_PI / 180.0;
  a.noise_sigma = noise_sigma;
  a.corner_band = corner_band;
  a.max_range = max_range;
  a.seed = seed;
  a.pts = p.pts;
  a.label = p.label;
  const size_t lds = sizeof(Solid) * (size_t)n_solids;
  if (lds > 64 * 1024)
    T(hipFuncSetAttribute(reinterpret_cast<const void *>(raycast_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(raycast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, p.stream, a);
  T(hipGetLastError());
  T(hipMemcpyAsync(out_pts, p.pts, sizeof(float4) * n, hipMemcpyDeviceToHost, p.stream));
  T(hipMemcpyAsync(out_label, p.label, n, hipMemcpyDeviceToHost, p.stream));
  T(hipStreamSynchronize(p.stream));
#undef T
  return 0;
}

}  // extern "C"
