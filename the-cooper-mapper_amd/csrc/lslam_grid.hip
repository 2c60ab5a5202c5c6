// lslam_grid.hip -- builds the cell grid of lslam_grid.hpp: points grouped by cell and the cell -> first point table.
// Everything on the device, stream-ordered; the one value the host needs (the bounding box) it either has from the tree build
// (TreeView::bb_lo / bb_hi = nanoflann's root_bbox, nanoflann.hpp:1406-1427) or reduces in one round trip (grid_bbox2).
//
// A counting sort without a sort: count per cell (atomics into a dense table that every build leaves zeroed for the next one),
// one scan kernel (grid_scan_kernel: coarse counts per 4096 cells give each workgroup its offset, no device-wide scan
// primitive), placement through the counts as cursors, and a rank pass that puts the points of a cell in the order of their
// original index -- the grid is the same bits whatever order the atomics ran in.  For a deferred-trees map this runs once
// per frame: four launches per feature type (rocPRIM's radix sort + scan took 25: 140 + 65 us for 587 k points).
#include "lslam_grid.hpp"
#include "lslam_internal.hpp"


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

constexpr int CS_CELLS = 4096;   // cells per workgroup of the scan
constexpr int RANK_MAX = 1024;   // cells with more points keep the order the atomics gave them (such a cell is never proven from)

__global__ __launch_bounds__(256) void grid_count_kernel(CellGrid G, const float4 *src, uint32_t *count, uint32_t *coarse, int32_t *err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= G.n_pts) return;
  bool ok;
  const int c = cell_of(G, src[i], ok);
  if (!ok) atomicAdd(err, 1);  // a point outside its own bounding box: not a number
  {
    // a map comes in voxel order: neighbouring lanes often share a cell -- one atomic per run of equal cells in the wavefront
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(c, 1, 64);
    const unsigned long long alive = __ballot(1);
    const bool head = lane == 0 || c != prev || !((alive >> (lane - 1)) & 1ull);
    const unsigned long long heads = __ballot(head);
    if (head) {
      const unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1));
      const int upto = after ? lane + 1 + __builtin_ctzll(after) : 64;  // the next head, or the end of the wavefront
      const int run = __popcll(alive & (upto == 64 ? ~0ull : ((1ull << upto) - 1ull)) & ~((1ull << lane) - 1ull));
      atomicAdd(count + c, (uint32_t)run);
    }
  }
  // the coarse count: neighbouring points share it more often than not -- one atomic per distinct value in the wavefront
  const int cb = c / CS_CELLS;
  unsigned long long todo = __ballot(1);
  while (todo) {
    const int first = __builtin_amdgcn_readlane(cb, __builtin_ctzll(todo));
    const unsigned long long same = __ballot(cb == first) & todo;
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(same)) atomicAdd(coarse + first, (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

// cell_start[c] = number of points in cells < c, for c in [0, ncell]: each workgroup owns CS_CELLS consecutive cells; what
// lies before them is the sum of the coarse counts of the workgroups before it.  HBM traffic: the counts read once, the table
// written once.
__global__ __launch_bounds__(256) void grid_scan_kernel(const uint32_t *count, const uint32_t *coarse, uint32_t ncell_plus1,
                                                        uint32_t n_pts, uint32_t *cell_start) {
  __shared__ uint32_t wave_sum[4], wave_before[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t base = blockIdx.x * (uint32_t)CS_CELLS;
  uint32_t pre = 0;
  for (uint32_t b = tid; b < blockIdx.x; b += 256) pre += coarse[b];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) pre += __shfl_xor(pre, d, 64);
  if (lane == 0) wave_before[wave] = pre;
  const uint32_t c0 = base + 16u * (uint32_t)tid;
  uint32_t v[16];
  // most 4 096-cell blocks of a map's box are empty (the coarse count says so): their counts are not read at all
  const bool empty = blockIdx.x < (ncell_plus1 - 1u + CS_CELLS - 1u) / CS_CELLS ? coarse[blockIdx.x] == 0u : true;
  if (empty) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 0u;
  } else if (c0 + 16u <= ncell_plus1 - 1u) {  // (the table has ncell_plus1 - 1 counts)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint4 q = reinterpret_cast<const uint4 *>(count + c0)[k];
      v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (c0 + (uint32_t)k < ncell_plus1 - 1u) ? count[c0 + k] : 0u;
  }
  uint32_t run = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t c = v[k];
    v[k] = run;
    run += c;
  }
  uint32_t incl = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t before = wave_before[0] + wave_before[1] + wave_before[2] + wave_before[3] + incl - run;
  for (int w = 0; w < wave; ++w) before += wave_sum[w];
  (void)n_pts;
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

// every point to a slot of its cell's run; the count is the cursor and ends at zero -- the table is clean for the next build
__global__ __launch_bounds__(256) void grid_scatter_kernel(CellGrid G, const float4 *src, const uint32_t *cell_start, uint32_t *count,
                                                           float4 *tmp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= G.n_pts) return;
  const float4 p = src[i];
  bool ok;
  const int c = cell_of(G, p, ok);
  const uint32_t left = atomicSub(count + c, 1u);
  tmp[cell_start[c] + left - 1u] = p;
}

// ... and inside its run to the place its original index gives it (.w): the grid does not depend on the order of the atomics
__global__ __launch_bounds__(256) void grid_rank_kernel(CellGrid G, const float4 *tmp, const uint32_t *cell_start, float4 *pts) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= G.n_pts) return;
  const float4 p = tmp[j];
  bool ok;
  const int c = cell_of(G, p, ok);
  const uint32_t s = cell_start[c], e = cell_start[c + 1];
  uint32_t at = (uint32_t)j;
  if (e - s <= (uint32_t)RANK_MAX) {
    const uint32_t me = __float_as_uint(p.w);
    uint32_t rank = 0;
    for (uint32_t i = s; i < e; ++i) rank += __float_as_uint(tmp[i].w) < me ? 1u : 0u;
    at = s + rank;
  }
  pts[at] = p;
}

__device__ __forceinline__ uint32_t ordered_u32(float f) {  // monotone map float -> uint32 (for atomicMin / atomicMax)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(256) void grid_bbox_kernel(const float4 *pts, int n, uint32_t *box /* [6]: min xyz, max xyz (ordered) */) {
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
  // wavefront reduction in registers, then six atomics per wavefront (256 lanes on six LDS words serialise)
#pragma unroll
  for (int d = 32; d > 0; d >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], d, 64));
      hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], d, 64));
    }
  __shared__ uint32_t part[4][6];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      part[wave][a] = lo[a];
      part[wave][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {  // six atomics per workgroup: same-address atomics from a whole grid serialise at the memory side
    const int a = threadIdx.x;
    uint32_t v = part[0][a];
    for (int w = 1; w < 4; ++w) v = a < 3 ? min(v, part[w][a]) : max(v, part[w][a]);
    if (a < 3) atomicMin(&box[a], v); else atomicMax(&box[a], v);
  }
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
    if (n[k] > 0) hipLaunchKernelGGL(grid_bbox_kernel, dim3(std::min((n[k] + 4095) / 4096, 256)), dim3(256), 0, s, pts[k], n[k], d_box12 + 6 * k);
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
  // The two dense tables (count, cell_start: 8 bytes per cell) are memset / rewritten in full by every build: a map that is wide
  // but sparse must not cost gigabytes of both.  GRID_MAX_CELLS cells (512 MB per type; the bench surround has 16 M), or 4 096
  // cells per point, whichever is smaller -- beyond it there is no grid (status 2) and the kd-tree search is used, as for a
  // map larger than GRID_MAX_DIM cells on an axis.
  if (ncell > GRID_MAX_CELLS || ncell > (size_t)4096 * (size_t)std::max(n_src, 4096)) { *status = 2; return hipSuccess; }
  const int n = n_src;
  hipError_t e;
#define G_TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
  // An allocation that fails is "no grid" (status 3), not a failed call: the tree search needs none of these arrays.
#define G_ALLOC(x) do { e = (x); if (e != hipSuccess) { (void)hipGetLastError(); release(); *status = 3; return hipSuccess; } } while (0)
  G_ALLOC(reserve(pts, cap_pts, (size_t)n + 16));
  G_ALLOC(reserve(tmp_pts, cap_tmp, (size_t)n + 16));
  G_ALLOC(reserve(cell_start, cap_cell, ncell + 1));
  const size_t n_coarse = (ncell + CS_CELLS - 1) / CS_CELLS;
  // the coarse counts and, right behind them (at a 16-byte boundary), the "point outside the table" counter: zeroed by ONE fill
  const size_t n_coarse_pad = (n_coarse + 3) & ~(size_t)3;
  G_ALLOC(reserve(coarse, cap_coarse, n_coarse_pad + 4));
  err = reinterpret_cast<int32_t *>(coarse + n_coarse_pad);
  {
    // the count table is left zeroed by every build (the scatter counts it down); only a new allocation is cleared
    const size_t before = cap_count;
    G_ALLOC(reserve(count, cap_count, ncell + 16));
    if (cap_count != before || !count_clean) G_TRY(hipMemsetAsync(count, 0, cap_count * sizeof(uint32_t), s));
    count_clean = false;
  }
  G_TRY(hipMemsetAsync(coarse, 0, (n_coarse_pad + 4) * sizeof(uint32_t), s));
  const dim3 blk(256), grd((n + 255) / 256);
  hipLaunchKernelGGL(grid_count_kernel, grd, blk, 0, s, G, src, count, coarse, err);
  hipLaunchKernelGGL(grid_scan_kernel, dim3((unsigned)((ncell + 1 + CS_CELLS - 1) / CS_CELLS)), blk, 0, s, (const uint32_t *)count,
                     (const uint32_t *)coarse, (uint32_t)(ncell + 1), (uint32_t)n, cell_start);
  hipLaunchKernelGGL(grid_scatter_kernel, grd, blk, 0, s, G, src, (const uint32_t *)cell_start, count, tmp_pts);
  G.cell_start = cell_start;  // (the rank kernel reads the table through the view)
  hipLaunchKernelGGL(grid_rank_kernel, grd, blk, 0, s, G, (const float4 *)tmp_pts, (const uint32_t *)cell_start, pts);
  G_TRY(hipGetLastError());
  count_clean = true;  // once the stream gets there
  // the candidate loop loads pts[cur] for lanes that have run out of candidates at index 0: nothing to pad; a leaf-style
  // over-read does not exist here
  int32_t h_err = 0;
  if (wait) {
    G_TRY(hipMemcpyAsync(&h_err, err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    G_TRY(hipStreamSynchronize(s));
  }
#undef G_TRY
#undef G_ALLOC
  if (h_err) { *status = 1; return hipSuccess; }
  G.cell_start = cell_start;
  G.pts = pts;
  view = G;
  n_cells = ncell;
  return hipSuccess;
}

void GridDev::release() {
  for (void *q : {(void *)pts, (void *)tmp_pts, (void *)cell_start, (void *)count, (void *)coarse})  // (err lives behind coarse)
    if (q) (void)hipFree(q);
  pts = tmp_pts = nullptr; cell_start = count = coarse = nullptr; err = nullptr;
  cap_pts = cap_tmp = cap_cell = cap_count = cap_coarse = cap_err = 0;
  count_clean = false;
  view = CellGrid{};
  n_cells = 0;
}

}  // namespace lslam
