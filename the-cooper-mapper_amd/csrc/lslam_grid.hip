// lslam_grid.hip -- builds the cell grid of lslam_grid.hpp over a map whose kd-tree exists (the grid search's fallback and
// the source of the bounding box): points sorted by cell and the cell -> first point table.  Everything on the device, stream-ordered; the one value the host needs (the bounding box) it already has from the
// tree build (TreeView::bb_lo / bb_hi = nanoflann's root_bbox, nanoflann.hpp:1406-1427).
//
// The key sort and the scan of the cell counts are rocPRIM's (map-set path, outside every timed region).
#include "lslam_grid.hpp"
#include "lslam_internal.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>

namespace lslam {

namespace {

LSLAM_DEV int cell_of(const CellGrid &G, const float4 &p, bool &ok) {
  const float ux = __fmul_rn(__fsub_rn(p.x, G.org[0]), G.inv_c);
  const float uy = __fmul_rn(__fsub_rn(p.y, G.org[1]), G.inv_c);
  const float uz = __fmul_rn(__fsub_rn(p.z, G.org[2]), G.inv_c);
  ok = ux >= 0.0f && ux < (float)G.nx && uy >= 0.0f && uy < (float)G.ny && uz >= 0.0f && uz < (float)G.nz;
  return ok ? (int)ux + G.nx * ((int)uy + G.ny * (int)uz) : 0;
}

__global__ __launch_bounds__(256) void grid_key_kernel(CellGrid G, const float4 *tree_pts, uint32_t *key, uint32_t *val,
                                                       uint32_t *count, int32_t *err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= G.n_pts) return;
  bool ok;
  const int c = cell_of(G, tree_pts[i], ok);
  if (!ok) atomicAdd(err, 1);  // a point outside its own bounding box: not a number
  key[i] = (uint32_t)c;
  val[i] = (uint32_t)i;
  atomicAdd(count + c, 1u);
}

__global__ __launch_bounds__(256) void grid_place_kernel(int n, const float4 *tree_pts, const uint32_t *val_sorted, float4 *gpts) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t t = val_sorted[j];
  gpts[j] = tree_pts[t];  // .w carries the original index already
}

}  // namespace

// Host side of a grid (one per map type): owns the device arrays.
hipError_t GridDev::build(const TreeView &T, float cell, hipStream_t s, int *status) {
  *status = 0;
  view = CellGrid{};
  if (T.n_pts <= 0 || !(cell > 0.05f) || !(cell < 1.45f)) return hipSuccess;  // rg <= 1.5 c must stay inside the sqrt(5) m gate
  const int margin = grid_margin_cells(cell);
  CellGrid G{};
  G.c = cell;
  G.inv_c = 1.0f / cell;
  G.n_pts = T.n_pts;
  int dims[3];
  for (int a = 0; a < 3; ++a) {
    if (!std::isfinite(T.bb_lo[a]) || !std::isfinite(T.bb_hi[a])) { *status = 1; return hipSuccess; }
    G.org[a] = T.bb_lo[a] - (float)margin * cell;
    dims[a] = (int)std::floor((T.bb_hi[a] - G.org[a]) * G.inv_c) + 1 + margin;
    if (dims[a] > GRID_MAX_DIM) { *status = 2; return hipSuccess; }  // the map is larger than the grid's rounding slack allows
  }
  G.nx = dims[0]; G.ny = dims[1]; G.nz = dims[2];
  const size_t ncell = (size_t)G.nx * G.ny * G.nz;
  if (ncell > ((size_t)1 << 30)) { *status = 2; return hipSuccess; }
  const int n = T.n_pts;
  hipError_t e;
#define G_TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
  G_TRY(reserve(pts, cap_pts, (size_t)n + 16));
  G_TRY(reserve(cell_start, cap_cell, ncell + 1));
  G_TRY(reserve(count, cap_count, ncell + 1));
  G_TRY(reserve(key0, cap_k0, (size_t)n));
  G_TRY(reserve(key1, cap_k1, (size_t)n));
  G_TRY(reserve(val0, cap_v0, (size_t)n));
  G_TRY(reserve(val1, cap_v1, (size_t)n));
  G_TRY(reserve(err, cap_err, 1));
  G_TRY(hipMemsetAsync(count, 0, (ncell + 1) * sizeof(uint32_t), s));
  G_TRY(hipMemsetAsync(err, 0, sizeof(int32_t), s));
  const dim3 blk(256), grd((n + 255) / 256);
  hipLaunchKernelGGL(grid_key_kernel, grd, blk, 0, s, G, T.pts, key0, val0, count, err);
  unsigned end_bit = 1;
  while (((size_t)1 << end_bit) < ncell) ++end_bit;
  size_t tmp_sort = 0, tmp_scan = 0;
  G_TRY(rocprim::radix_sort_pairs(nullptr, tmp_sort, key0, key1, val0, val1, (size_t)n, 0u, end_bit, s));
  G_TRY(rocprim::exclusive_scan(nullptr, tmp_scan, count, cell_start, 0u, ncell + 1, rocprim::plus<uint32_t>(), s));
  G_TRY(reserve(tmp, cap_tmp, std::max(tmp_sort, tmp_scan)));
  G_TRY(rocprim::radix_sort_pairs((void *)tmp, tmp_sort, key0, key1, val0, val1, (size_t)n, 0u, end_bit, s));
  G_TRY(rocprim::exclusive_scan((void *)tmp, tmp_scan, count, cell_start, 0u, ncell + 1, rocprim::plus<uint32_t>(), s));
  hipLaunchKernelGGL(grid_place_kernel, grd, blk, 0, s, n, T.pts, val1, pts);
  // the candidate loop loads pts[cur] for lanes that have run out of candidates at index 0: nothing to pad; a leaf-style
  // over-read does not exist here
  int32_t h_err = 0;
  G_TRY(hipMemcpyAsync(&h_err, err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  G_TRY(hipStreamSynchronize(s));
#undef G_TRY
  if (h_err) { *status = 1; return hipSuccess; }
  G.cell_start = cell_start;
  G.pts = pts;
  view = G;
  n_cells = ncell;
  return hipSuccess;
}

void GridDev::release() {
  for (void *q : {(void *)pts, (void *)cell_start, (void *)count, (void *)key0, (void *)key1, (void *)val0, (void *)val1,
                  (void *)err, (void *)tmp})
    if (q) (void)hipFree(q);
  pts = nullptr; cell_start = nullptr; count = nullptr; key0 = key1 = val0 = val1 = nullptr; err = nullptr; tmp = nullptr;
  cap_pts = cap_cell = cap_count = cap_k0 = cap_k1 = cap_v0 = cap_v1 = cap_err = cap_tmp = 0;
  view = CellGrid{};
  n_cells = 0;
}

}  // namespace lslam
