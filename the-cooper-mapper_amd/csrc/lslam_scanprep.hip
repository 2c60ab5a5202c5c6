// lslam_scanprep.hip -- device side of lslam_scan_set_batch: Morton ordering of the resident scans.
//
// Scan points are processed one per lane and a wavefront is as slow as its most expensive kd-tree
// traversal, so every cloud (per scan: corner, surf) is ordered along a Morton curve of 0.25 m cells
// (see lslam_api.hip).  The order is the one the host implementation defines -- ascending
// (30-bit Morton key, original index) inside each cloud -- produced here by ONE stable radix sort
// (rocPRIM) of the whole batch on the key (cloud id | Morton key); the original index travels in .w.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "lslam_internal.hpp"

namespace lslam {

namespace {

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__device__ __forceinline__ uint32_t quant(float v) {
  float c = __fadd_rn(__fmul_rn(v, 4.0f), 512.0f);  // 0.25 m cells, +-128 m
  c = c < 0.0f ? 0.0f : (c > 1023.0f ? 1023.0f : c);
  return (uint32_t)c;
}

__global__ void sp_key_kernel(const float4 *pts, int n, const int32_t *seg_off, int nseg, uint64_t *keys,
                              uint32_t *idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = nseg - 1;  // last segment with seg_off <= i
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg_off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  const float4 p = pts[i];
  const uint32_t m = spread10(quant(p.x)) | (spread10(quant(p.y)) << 1) | (spread10(quant(p.z)) << 2);
  keys[i] = ((uint64_t)lo << 30) | m;
  idx[i] = (uint32_t)i;
}

__global__ void sp_gather_kernel(const float4 *pts, const uint32_t *idx, const uint64_t *keys, const int32_t *seg_off,
                                 int n, float4 *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t src = idx[i];
  float4 p = pts[src];
  const int seg = (int)(keys[i] >> 30);
  p.w = __builtin_bit_cast(float, src - (uint32_t)seg_off[seg]);  // index inside its own cloud
  out[i] = p;
}

}  // namespace

struct ScanPrep {
  void *keys0 = nullptr, *keys1 = nullptr, *idx0 = nullptr, *idx1 = nullptr, *tmp = nullptr, *raw = nullptr, *seg = nullptr;
  size_t cap = 0, tmp_cap = 0, seg_cap = 0;
};

ScanPrep *scanprep_create() { return new ScanPrep(); }
void scanprep_destroy(ScanPrep *sp) {
  if (!sp) return;
  for (void *p : {sp->keys0, sp->keys1, sp->idx0, sp->idx1, sp->tmp, sp->raw, sp->seg})
    if (p) (void)hipFree(p);
  delete sp;
}

// h_pts: n packed points in caller order (cloud after cloud); h_seg_off: nseg + 1 offsets.
// Writes the ordered points to d_out (n float4) on `s`, asynchronously: h_pts must stay untouched until `s` has been waited for.
hipError_t scanprep_order(ScanPrep *sp, hipStream_t s, const float4 *h_pts, size_t n, const int32_t *h_seg_off,
                          int nseg, float4 *d_out) {
  if (n == 0) return hipSuccess;
  hipError_t e;
  if (n > sp->cap) {
    for (void **p : {&sp->keys0, &sp->keys1, &sp->idx0, &sp->idx1, &sp->raw}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    sp->cap = 0;
    const size_t want = n + n / 4 + 1024;
    if ((e = hipMalloc(&sp->keys0, want * 8)) != hipSuccess) return e;
    if ((e = hipMalloc(&sp->keys1, want * 8)) != hipSuccess) return e;
    if ((e = hipMalloc(&sp->idx0, want * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&sp->idx1, want * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&sp->raw, want * sizeof(float4))) != hipSuccess) return e;
    sp->cap = want;
  }
  if ((size_t)nseg + 1 > sp->seg_cap) {
    if (sp->seg) (void)hipFree(sp->seg);
    sp->seg = nullptr;
    sp->seg_cap = 0;
    if ((e = hipMalloc(&sp->seg, ((size_t)nseg + 64) * 4)) != hipSuccess) return e;
    sp->seg_cap = (size_t)nseg + 64;
  }
  if ((e = hipMemcpyAsync(sp->raw, h_pts, n * sizeof(float4), hipMemcpyHostToDevice, s)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(sp->seg, h_seg_off, ((size_t)nseg + 1) * 4, hipMemcpyHostToDevice, s)) != hipSuccess) return e;
  int seg_bits = 1;
  while ((1 << seg_bits) < nseg) ++seg_bits;
  const unsigned end_bit = 30u + (unsigned)seg_bits;
  uint64_t *k0 = (uint64_t *)sp->keys0, *k1 = (uint64_t *)sp->keys1;
  uint32_t *i0 = (uint32_t *)sp->idx0, *i1 = (uint32_t *)sp->idx1;
  // the library's radix sort; LSLAM_SMALL_SORT=1 (A/B): lslam_sort.hip for a frame's scan (<= SMALL_SORT_MAX points)
  const bool small = n <= SMALL_SORT_MAX && env_once().small_sort;
  size_t tmp_bytes = 0;
  if (small) tmp_bytes = small_sort_tmp_bytes(n);
  else if ((e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, i0, i1, n, 0u, end_bit, s)) != hipSuccess) return e;
  if (tmp_bytes > sp->tmp_cap) {
    if (sp->tmp) (void)hipFree(sp->tmp);
    sp->tmp = nullptr;
    sp->tmp_cap = 0;
    if ((e = hipMalloc(&sp->tmp, tmp_bytes + tmp_bytes / 4)) != hipSuccess) return e;
    sp->tmp_cap = tmp_bytes + tmp_bytes / 4;
  }
  const dim3 blk(256), grd((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(sp_key_kernel, grd, blk, 0, s, (const float4 *)sp->raw, (int)n, (const int32_t *)sp->seg, nseg, k0, i0);
  if (small) e = small_sort_pairs(s, k0, k1, i0, i1, n, sp->tmp);
  else e = rocprim::radix_sort_pairs(sp->tmp, tmp_bytes, k0, k1, i0, i1, n, 0u, end_bit, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sp_gather_kernel, grd, blk, 0, s, (const float4 *)sp->raw, i1, k1, (const int32_t *)sp->seg, (int)n,
                     d_out);
  // No wait: h_seg_off is pageable (consumed when hipMemcpyAsync returned), h_pts is the context's pinned staging area,
  // which the caller does not touch again before it has waited on `s` (lslam_ctx::stage_busy).
  return hipGetLastError();
}

}  // namespace lslam
