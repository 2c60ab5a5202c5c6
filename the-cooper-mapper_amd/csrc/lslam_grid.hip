// lslam_grid.hip -- builds the cell grid of lslam_grid.hpp over a map whose kd-tree exists (the grid search's fallback and
// the source of the bounding box): points sorted by cell and the cell -> first point table.  Everything on the device, stream-ordered; the one value the host needs (the bounding box) it already has from the
// tree build (TreeView::bb_lo / bb_hi = nanoflann's root_bbox, nanoflann.hpp:1406-1427).
//
// The key sort is rocPRIM's radix sort, held to its onesweep form (its default below 1 Mi keys is a merge sort of ~20 launches for
// the 587 k points of a surf surround: 140 us against 30); the cell -> first point table is written straight from the sorted keys
// (grid_cellstart_kernel: no count table, no memset, no device-wide scan -- for a deferred-trees map this runs once per frame).
#include "lslam_grid.hpp"
#include "lslam_internal.hpp"

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace lslam {

namespace {

LSLAM_DEV int cell_of(const CellGrid &G, const float4 &p, bool &ok) {
  const float ux = __fmul_rn(__fsub_rn(p.x, G.org[0]), G.inv_c);
  const float uy = __fmul_rn(__fsub_rn(p.y, G.org[1]), G.inv_c);
  const float uz = __fmul_rn(__fsub_rn(p.z, G.org[2]), G.inv_c);
  ok = ux >= 0.0f && ux < (float)G.nx && uy >= 0.0f && uy < (float)G.ny && uz >= 0.0f && uz < (float)G.nz;
  return ok ? (int)ux + G.nx * ((int)uy + G.ny * (int)uz) : 0;
}

__global__ __launch_bounds__(256) void grid_key_kernel(CellGrid G, const float4 *tree_pts, uint32_t *key, uint32_t *val, int32_t *err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= G.n_pts) return;
  bool ok;
  const int c = cell_of(G, tree_pts[i], ok);
  if (!ok) atomicAdd(err, 1);  // a point outside its own bounding box: not a number
  key[i] = (uint32_t)c;
  val[i] = (uint32_t)i;
}

// cell_start[c] = number of points in cells < c, for c in [0, ncell]: each workgroup owns CS_CELLS consecutive cells, finds its
// slice of the SORTED keys by two binary searches, counts it into LDS and scans there.  HBM traffic: the table written once
// (4 B per cell), the keys read once.
constexpr int CS_CELLS = 4096;
__global__ __launch_bounds__(256) void grid_cellstart_kernel(const uint32_t *keys, int n, uint32_t ncell_plus1, uint32_t *cell_start) {
  __shared__ uint32_t cnt[CS_CELLS];
  __shared__ uint32_t wave_sum[4];
  __shared__ int slice[2];
  const int tid = threadIdx.x;
  const uint32_t base = blockIdx.x * (uint32_t)CS_CELLS;
#pragma unroll
  for (int k = 0; k < CS_CELLS / 256 / 4; ++k) reinterpret_cast<uint4 *>(cnt)[tid + 256 * k] = make_uint4(0u, 0u, 0u, 0u);
  if (tid < 2) {  // first key >= base (tid 0), first key >= base + CS_CELLS (tid 1)
    const uint64_t target = (uint64_t)base + (uint64_t)tid * CS_CELLS;
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((uint64_t)keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    slice[tid] = lo;
  }
  __syncthreads();
  const int j0 = slice[0], j1 = slice[1];
  for (int j = j0 + tid; j < j1; j += 256) atomicAdd(&cnt[keys[j] - base], 1u);
  __syncthreads();
  // exclusive scan of the 4096 counters: 16 per thread, wavefront scan of the thread sums, four wavefront sums through LDS
  uint32_t v[16];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint4 q = reinterpret_cast<const uint4 *>(cnt)[4 * tid + k];
    v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
  }
  uint32_t run = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t c = v[k];
    v[k] = run;
    run += c;
  }
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t incl = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t before = (uint32_t)j0 + incl - run;
  for (int w = 0; w < wave; ++w) before += wave_sum[w];
  const uint32_t c0 = base + 16u * (uint32_t)tid;
  if (c0 + 16u <= ncell_plus1) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      reinterpret_cast<uint4 *>(cell_start + c0)[k] = make_uint4(before + v[4 * k], before + v[4 * k + 1], before + v[4 * k + 2], before + v[4 * k + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (c0 + (uint32_t)k < ncell_plus1) cell_start[c0 + k] = before + v[k];
  }
}

__global__ __launch_bounds__(256) void grid_place_kernel(int n, const float4 *tree_pts, const uint32_t *val_sorted, float4 *gpts) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t t = val_sorted[j];
  gpts[j] = tree_pts[t];  // .w carries the original index already
}

__device__ __forceinline__ uint32_t ordered_u32(float f) {  // monotone map float -> uint32 (for atomicMin / atomicMax)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(256) void grid_bbox_kernel(const float4 *pts, int n, uint32_t *box /* [6]: min xyz, max xyz (ordered) */) {
  __shared__ uint32_t sm[6];
  if (threadIdx.x < 3) sm[threadIdx.x] = 0xFFFFFFFFu;
  else if (threadIdx.x < 6) sm[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    const uint32_t v[3] = {ordered_u32(p.x), ordered_u32(p.y), ordered_u32(p.z)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = min(lo[a], v[a]);
      hi[a] = max(hi[a], v[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    atomicMin(&sm[a], lo[a]);
    atomicMax(&sm[3 + a], hi[a]);
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&box[threadIdx.x], sm[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&box[threadIdx.x], sm[threadIdx.x]);
}

__global__ __launch_bounds__(256) void grid_unsort_kernel(int n, const float4 *gpts, float4 *out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float4 p = gpts[j];
  out[__float_as_int(p.w)] = p;
}

}  // namespace

// The cell-sorted points back in their original order ({x, y, z, bitcast(index)}): what a kd-tree build deferred at map-set
// time starts from.
hipError_t grid_unsort(const CellGrid &G, float4 *out, hipStream_t s) {
  if (G.n_pts <= 0) return hipSuccess;
  hipLaunchKernelGGL(grid_unsort_kernel, dim3((G.n_pts + 255) / 256), dim3(256), 0, s, G.n_pts, G.pts, out);
  return hipGetLastError();
}

// Bounding boxes of two device clouds (nanoflann's root_bbox: componentwise min / max), ONE host round trip for both.
hipError_t grid_bbox2(const float4 *const pts[2], const int n[2], uint32_t *d_box12, float lo[2][3], float hi[2][3], hipStream_t s) {
  uint32_t init[12];
  for (int k = 0; k < 2; ++k)
    for (int a = 0; a < 6; ++a) init[6 * k + a] = a < 3 ? 0xFFFFFFFFu : 0u;
  hipError_t e = hipMemcpyAsync(d_box12, init, sizeof(init), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  for (int k = 0; k < 2; ++k)
    if (n[k] > 0) hipLaunchKernelGGL(grid_bbox_kernel, dim3(std::min((n[k] + 255) / 256, 512)), dim3(256), 0, s, pts[k], n[k], d_box12 + 6 * k);
  uint32_t h[12];
  e = hipMemcpyAsync(h, d_box12, sizeof(h), hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(s);  // (also covers `init`: a local)
  if (e != hipSuccess) return e;
  for (int k = 0; k < 2; ++k)
    for (int a = 0; a < 6; ++a) {
      const uint32_t u = (h[6 * k + a] & 0x80000000u) ? (h[6 * k + a] & 0x7FFFFFFFu) : ~h[6 * k + a];
      float f;
      std::memcpy(&f, &u, 4);
      (a < 3 ? lo[k][a] : hi[k][a - 3]) = f;
    }
  return hipSuccess;
}

// Host side of a grid (one per map type): owns the device arrays.
hipError_t GridDev::build(const TreeView &T, float cell, hipStream_t s, int *status) {
  return build(T.pts, T.n_pts, T.bb_lo, T.bb_hi, cell, s, status, true);
}

// src: n points {x, y, z, bitcast(original index)} in any order; lo / hi: their bounding box
// wait: read the "point outside the table" counter back (one host round trip).  A caller whose box was reduced from these very
// points a moment ago has nothing to learn from it (a non-finite coordinate shows in the box) and passes false: stream-ordered, no wait.
hipError_t GridDev::build(const float4 *src, int n_src, const float lo[3], const float hi[3], float cell, hipStream_t s, int *status, bool wait) {
  *status = 0;
  view = CellGrid{};
  if (n_src <= 0 || !(cell > 0.05f) || !(cell < 1.45f)) return hipSuccess;  // rg <= 1.5 c must stay inside the sqrt(5) m gate
  const int margin = grid_margin_cells(cell);
  CellGrid G{};
  G.c = cell;
  G.inv_c = 1.0f / cell;
  G.n_pts = n_src;
  int dims[3];
  for (int a = 0; a < 3; ++a) {
    if (!std::isfinite(lo[a]) || !std::isfinite(hi[a])) { *status = 1; return hipSuccess; }
    G.org[a] = lo[a] - (float)margin * cell;
    dims[a] = (int)std::floor((hi[a] - G.org[a]) * G.inv_c) + 1 + margin;
    if (dims[a] > GRID_MAX_DIM) { *status = 2; return hipSuccess; }  // the map is larger than the grid's rounding slack allows
  }
  G.nx = dims[0]; G.ny = dims[1]; G.nz = dims[2];
  const size_t ncell = (size_t)G.nx * G.ny * G.nz;
  if (ncell > ((size_t)1 << 30)) { *status = 2; return hipSuccess; }
  const int n = n_src;
  hipError_t e;
#define G_TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
  G_TRY(reserve(pts, cap_pts, (size_t)n + 16));
  G_TRY(reserve(cell_start, cap_cell, ncell + 1));
  G_TRY(reserve(key0, cap_k0, (size_t)n));
  G_TRY(reserve(key1, cap_k1, (size_t)n));
  G_TRY(reserve(val0, cap_v0, (size_t)n));
  G_TRY(reserve(val1, cap_v1, (size_t)n));
  G_TRY(reserve(err, cap_err, 1));
  G_TRY(hipMemsetAsync(err, 0, sizeof(int32_t), s));
  const dim3 blk(256), grd((n + 255) / 256);
  hipLaunchKernelGGL(grid_key_kernel, grd, blk, 0, s, G, src, key0, val0, err);
  unsigned end_bit = 1;
  while (((size_t)1 << end_bit) < ncell) ++end_bit;
  using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 16384>;
  size_t tmp_sort = 0;
  G_TRY(rocprim::radix_sort_pairs<SortCfg>(nullptr, tmp_sort, key0, key1, val0, val1, (size_t)n, 0u, end_bit, s));
  G_TRY(reserve(tmp, cap_tmp, tmp_sort));
  G_TRY(rocprim::radix_sort_pairs<SortCfg>((void *)tmp, tmp_sort, key0, key1, val0, val1, (size_t)n, 0u, end_bit, s));
  hipLaunchKernelGGL(grid_cellstart_kernel, dim3((unsigned)((ncell + 1 + CS_CELLS - 1) / CS_CELLS)), blk, 0, s, (const uint32_t *)key1, n,
                     (uint32_t)(ncell + 1), cell_start);
  hipLaunchKernelGGL(grid_place_kernel, grd, blk, 0, s, n, src, val1, pts);
  // the candidate loop loads pts[cur] for lanes that have run out of candidates at index 0: nothing to pad; a leaf-style
  // over-read does not exist here
  int32_t h_err = 0;
  if (wait) {
    G_TRY(hipMemcpyAsync(&h_err, err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    G_TRY(hipStreamSynchronize(s));
  }
#undef G_TRY
  if (h_err) { *status = 1; return hipSuccess; }
  G.cell_start = cell_start;
  G.pts = pts;
  view = G;
  n_cells = ncell;
  return hipSuccess;
}

void GridDev::release() {
  for (void *q : {(void *)pts, (void *)cell_start, (void *)key0, (void *)key1, (void *)val0, (void *)val1,
                  (void *)err, (void *)tmp})
    if (q) (void)hipFree(q);
  pts = nullptr; cell_start = nullptr; key0 = key1 = val0 = val1 = nullptr; err = nullptr; tmp = nullptr;
  cap_pts = cap_cell = cap_k0 = cap_k1 = cap_v0 = cap_v1 = cap_err = cap_tmp = 0;
  view = CellGrid{};
  n_cells = 0;
}

}  // namespace lslam
