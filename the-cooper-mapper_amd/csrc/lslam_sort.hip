// lslam_sort.hip -- a hand-written sort for the per-frame map maintenance, BUILT, EXACT AND SLOWER THAN THE
// LIBRARY'S: an A/B switch (LSLAM_SMALL_SORT=1) and a parity tap (lslam_debug_sort_pairs), not the default.
//
// (64-bit key, 32-bit value) pairs, ascending by key, equal keys in ascending order of their values -- with values = input
// positions, which is what every caller passes, a STABLE sort by key (the order pcl::VoxelGrid's index sort and the Morton
// ordering of lslam_scanprep.hip are restated with).
//
// The lead (round-4 review, item 6): a frame sorts a few thousand to a few ten thousand keys four times (the VoxelGrid of the
// scan, the new points of addFeatureCloud per feature type, the Morton order of the scan), and below 2^20 keys
// rocprim::radix_sort_pairs is a merge sort of one block-sort launch plus one launch per doubling -- eight launches of ~6 us
// for 42 k keys, three for 5 k.  Here:
//   sort_tile_kernel    one workgroup of 512 threads sorts a tile of 4 096 pairs: eight per thread in registers (a 19-exchange
//                       network), then nine merge rounds through LDS by merge path -- every thread finds where its eight
//                       outputs start in the two runs (a binary search on the cross diagonal) and merges them serially; the
//                       first six rounds stay inside a wavefront and need no workgroup barrier
//   sort_rank_kernel    more than one tile (up to thirty-two): every pair's final place is its place in its own tile plus, for
//                       each other tile, the number of pairs there that precede it -- binary searches, sixteen in flight
//                       together, in L2-resident arrays, one scatter.  No further passes.
// Two launches for anything up to 131 072 pairs, one up to 4 096 -- twenty launches fewer per frame -- and MEASURED SLOWER
// (same box, interleaved, 300 frames each, twice): a mapping frame 1.34 ms against 1.27 - 1.29 with the library's sort.  A
// tile is one workgroup's chain of LDS latencies (33 us per tile: nine rounds of a <= 12-step search and eight dependent
// reads, two wavefronts per SIMD to hide them), where the library's block sort spreads 42 k keys over twenty workgroups
// (16 us) and its merge launches cost ~6 us each; tiles of 8 192 by 1 024 threads were LDS-bandwidth bound (1.41 ms).  The
// launches were not the cost.
#include <hip/hip_runtime.h>

#include "lslam_internal.hpp"

namespace lslam {

namespace {

constexpr int ST_THREADS = 512, ST_ITEMS = 8, ST_TILE = ST_THREADS * ST_ITEMS;
static_assert(ST_TILE == SMALL_SORT_TILE, "lslam_internal.hpp");
// LDS layout: position p lives at p + p / 8 -- a thread's eight consecutive pairs start 9 words after its neighbour's, so the
// blocked reads and writes of a wavefront (stride 72 bytes, not 64) spread over all banks
__device__ __forceinline__ int st_at(int p) { return p + (p >> 3); }
constexpr int ST_LDS = ST_TILE + ST_TILE / 8;

__device__ __forceinline__ bool kv_less(uint64_t ak, uint32_t av, uint64_t bk, uint32_t bv) {
  return ak < bk || (ak == bk && av < bv);
}
__device__ __forceinline__ void kv_cx(uint64_t &ak, uint32_t &av, uint64_t &bk, uint32_t &bv) {
  const bool sw = kv_less(bk, bv, ak, av);
  const uint64_t tk = ak;
  const uint32_t tv = av;
  ak = sw ? bk : ak; av = sw ? bv : av;
  bk = sw ? tk : bk; bv = sw ? tv : bv;
}

__global__ __launch_bounds__(ST_THREADS) void sort_tile_kernel(const uint64_t *k_in, const uint32_t *v_in, int n, uint64_t *k_out,
                                                               uint32_t *v_out) {
  __shared__ uint64_t lk[ST_LDS];
  __shared__ uint32_t lv[ST_LDS];
  const int tid = threadIdx.x, base = blockIdx.x * ST_TILE, cnt = min(ST_TILE, n - base);
  // coalesced in, blocked out of LDS: thread t owns positions 8 t .. 8 t + 7; the padding (all key bits set, values from 2^31
  // on: the callers' values are positions below that) sorts behind every real pair
  for (int i = tid; i < ST_TILE; i += ST_THREADS) {
    lk[st_at(i)] = i < cnt ? k_in[base + i] : ~0ull;
    lv[st_at(i)] = i < cnt ? v_in[base + i] : (0x80000000u | (uint32_t)i);  // (distinct: a place is a count of strictly smaller pairs)
  }
  __syncthreads();
  uint64_t k[ST_ITEMS];
  uint32_t v[ST_ITEMS];
#pragma unroll
  for (int u = 0; u < ST_ITEMS; ++u) { k[u] = lk[tid * (ST_ITEMS + 1) + u]; v[u] = lv[tid * (ST_ITEMS + 1) + u]; }
  // Batcher's odd-even merge sort of eight: 19 exchanges
#define CX(a, b) kv_cx(k[a], v[a], k[b], v[b])
  CX(0, 1); CX(2, 3); CX(4, 5); CX(6, 7);
  CX(0, 2); CX(1, 3); CX(4, 6); CX(5, 7);
  CX(1, 2); CX(5, 6);
  CX(0, 4); CX(1, 5); CX(2, 6); CX(3, 7);
  CX(2, 4); CX(3, 5);
  CX(1, 2); CX(3, 4); CX(5, 6);
#undef CX
  // Runs of w threads (8 w pairs) are merged two by two until one run covers the tile's pairs, by merge path: a thread finds
  // how many of the outputs before its eight come from the first run (a binary search on the cross diagonal) and merges its
  // eight serially.  (Tried instead: every pair's place = its place in its own run + the pairs of the sibling run before it,
  // eight searches per thread in flight together, scattered into a second buffer -- 81 us per tile against 33: the scattered
  // LDS traffic of eight times as many searches costs more than the dependent reads it avoids.)
  int used = 1;  // threads that hold real pairs, rounded up to a power of two
  while (used * ST_ITEMS < cnt) used <<= 1;
  for (int w = 1; w < used; w <<= 1) {
    if (w <= 32) {  // both runs belong to this wavefront: its LDS operations execute in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < ST_ITEMS; ++u) { lk[tid * (ST_ITEMS + 1) + u] = k[u]; lv[tid * (ST_ITEMS + 1) + u] = v[u]; }
    if (w <= 32) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
      __syncthreads();
    }
    const int L = w * ST_ITEMS;
    const int ps = (tid & ~(2 * w - 1)) * ST_ITEMS;  // first position of the pair of runs
    const int A0 = ps, B0 = ps + L;                   // run A = positions [A0, A0 + L), run B = [B0, B0 + L)
    const int diag = (tid & (2 * w - 1)) * ST_ITEMS;  // outputs of the merge before this thread's
    int lo = max(0, diag - L), hi = min(diag, L);     // how many of them come from A
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const int pa = st_at(A0 + mid), pb = st_at(B0 + diag - 1 - mid);
      if (kv_less(lk[pa], lv[pa], lk[pb], lv[pb])) lo = mid + 1; else hi = mid;
    }
    int a = lo, b = diag - lo;
    uint64_t ak = a < L ? lk[st_at(A0 + a)] : ~0ull, bk = b < L ? lk[st_at(B0 + b)] : ~0ull;
    uint32_t av = a < L ? lv[st_at(A0 + a)] : 0xFFFFFFFFu, bv = b < L ? lv[st_at(B0 + b)] : 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < ST_ITEMS; ++u) {
      const bool take_a = b >= L || (a < L && kv_less(ak, av, bk, bv));
      k[u] = take_a ? ak : bk;
      v[u] = take_a ? av : bv;
      if (take_a) {
        ++a;
        ak = a < L ? lk[st_at(A0 + a)] : ~0ull;
        av = a < L ? lv[st_at(A0 + a)] : 0xFFFFFFFFu;
      } else {
        ++b;
        bk = b < L ? lk[st_at(B0 + b)] : ~0ull;
        bv = b < L ? lv[st_at(B0 + b)] : 0xFFFFFFFFu;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < ST_ITEMS; ++u) {
    const int i = tid * ST_ITEMS + u;
    if (i < cnt) { k_out[base + i] = k[u]; v_out[base + i] = v[u]; }
  }
}

// k / v: tiles of ST_TILE pairs, each sorted.  Pair i goes to (its place in its tile) + sum over the other tiles of the pairs
// that precede it there.
__global__ __launch_bounds__(256) void sort_rank_kernel(const uint64_t *k, const uint32_t *v, int n, uint64_t *k_out, uint32_t *v_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int n_tiles = (n + ST_TILE - 1) / ST_TILE, mine = i / ST_TILE;
  const uint64_t key = k[i];
  const uint32_t val = v[i];
  int rank = i - mine * ST_TILE;
  constexpr int CH = 16;  // searches in flight together
  for (int t0 = 0; t0 < n_tiles; t0 += CH) {
    int lo[CH], hi[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int t = t0 + c;
      lo[c] = t * ST_TILE;
      hi[c] = (t < n_tiles && t != mine) ? min(n, (t + 1) * ST_TILE) : lo[c];
    }
    for (int step = 0; step < 13; ++step) {  // 2^12 = ST_TILE: thirteen halvings empty every range
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (lo[c] < hi[c]) {
          const int mid = (lo[c] + hi[c]) >> 1;
          const uint64_t mk = k[mid];
          const bool before = mk < key || (mk == key && v[mid] < val);
          lo[c] = before ? mid + 1 : lo[c];
          hi[c] = before ? hi[c] : mid;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) rank += lo[c] - (t0 + c) * ST_TILE;
  }
  k_out[rank] = key;
  v_out[rank] = val;
}

}  // namespace

size_t small_sort_tmp_bytes(size_t n) { return n > (size_t)ST_TILE ? n * 12 + 64 : 0; }

hipError_t small_sort_pairs(hipStream_t s, const uint64_t *k_in, uint64_t *k_out, const uint32_t *v_in, uint32_t *v_out, size_t n,
                            void *tmp) {
  if (n == 0) return hipSuccess;
  if (n > SMALL_SORT_MAX) return hipErrorInvalidValue;
  const unsigned tiles = (unsigned)((n + ST_TILE - 1) / ST_TILE);
  if (tiles == 1) {
    hipLaunchKernelGGL(sort_tile_kernel, dim3(1), dim3(ST_THREADS), 0, s, k_in, v_in, (int)n, k_out, v_out);
    return hipGetLastError();
  }
  uint64_t *tk = reinterpret_cast<uint64_t *>(tmp);
  uint32_t *tv = reinterpret_cast<uint32_t *>(tk + ((n + 7) & ~(size_t)7));
  hipLaunchKernelGGL(sort_tile_kernel, dim3(tiles), dim3(ST_THREADS), 0, s, k_in, v_in, (int)n, tk, tv);
  hipLaunchKernelGGL(sort_rank_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint64_t *)tk, (const uint32_t *)tv, (int)n,
                     k_out, v_out);
  return hipGetLastError();
}

}  // namespace lslam
