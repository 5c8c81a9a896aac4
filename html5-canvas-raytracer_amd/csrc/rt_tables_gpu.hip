// rt_tables_gpu.hip — the product kernel's launch table, built ON THE GPU (what rt_api.hip: dispatch_order runs; rt_tables.cpp's
// build_launch_table is the same table built on the host: the CPU tests' oracle and the -m gpu test that compares the two).
//
// The reference recomputes everything on every redraw() (main.js:180-201) and its camera is a parameter (lookAt, main.js:92-100).
// The table - per block of 32 x 8 pixels: is it sky, which spheres can its primary rays meet, which spheres can shadow its hits,
// what will it cost (rt_block.h) - depends on the camera, the frame size and the tile set; on the host it takes 5-30 ms for a
// 3840x2160 frame against a 0.07 ms trace.  Here it is three small launches on the render stream, no host step in between:
//
//   rt_table_rows   one workgroup per ROW BLOCK (a row of blocks): every work-item states its block (rt_block_statement, rt_block_cost:
//                   the host's own source), then the row decides in LDS which blocks start an entry - a block that shows a sphere is an
//                   entry of its own, consecutive sky blocks share one (a run never crosses a multiple of 32 blocks, so "do I start a
//                   run" is a question about the left neighbour) - how many entries of the same cost lie to the left of each (its
//                   rank among equals in the row), and the row's histogram of entry costs;
//   rt_table_scan   ONE workgroup: per cost, the exclusive prefix of the rows' histograms down the rows (a wave scans a column in
//                   chunks of 64 rows), the totals, and their exclusive prefix from the dearest cost down: where each cost class
//                   starts in the dispatch order.  Counting sort, stable: equal costs keep the grid's order, as on the host.  It
//                   also writes the table's header {entries, ceil(entries / 8)} and publishes the number of entries to the host;
//   rt_table_emit   one work-item per block: an entry's workgroup index is class start + entries of its class in the rows above +
//                   its rank in the row; its 16 bytes go to slot (b % 8) * ceil(blocks / 8) + b / 8 (one contiguous part per XCD).
//
// The trace kernel is launched right behind them with one workgroup per BLOCK (the number of entries is only known on the device):
// it reads the number of entries from the table's header and workgroups beyond the last entry leave at once; once the host has
// seen the published count it launches exactly that many.  Compiled without FMA contraction, like rt_tables.cpp: both builds state the same words.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_block.h"
#include "rt_tables_gpu.h"

namespace {

constexpr uint32_t WG = 256;

__global__ void __launch_bounds__(WG) rt_table_rows(const rt_table_dev T) {
  extern __shared__ uint32_t lds[];
  const rt_table_params &P = *T.params;
  const uint32_t y = blockIdx.x, tiles_x = P.tiles_x, bins = P.cost_bins;
  uint32_t *l_touched = lds;                 // [tiles_x] 1: shows a sphere (or no sky marking at all)
  uint32_t *l_key = lds + tiles_x;           // [tiles_x] entry starts: 1 + cost bin; 0: not an entry start
  uint32_t *l_hist = lds + 2u * tiles_x;     // [bins]
  const bool sky = (P.flags & RT_TABLE_SKY) && (P.flags & RT_TABLE_GEOMETRY);
  const bool rank = (P.flags & RT_TABLE_RANK) != 0u;
  for (uint32_t c = threadIdx.x; c < bins; c += WG) l_hist[c] = 0u;
  for (uint32_t x = threadIdx.x; x < tiles_x; x += WG) {
    uint32_t touched, cands, smask;
    rt_block_statement(P, T.balls, x, y, &touched, &cands, &smask);
    const uint32_t cost = rank ? rt_block_cost(P, T.rects, x, y) : 1u;
    const size_t at = (size_t)y * tiles_x + x;
    T.blk[3u * at] = cost; T.blk[3u * at + 1u] = smask; T.blk[3u * at + 2u] = cands;
    l_touched[x] = (!sky || touched) ? 1u : 0u;
    // every slot of the table starts as zero (rt_table_emit fills the entries in; a trace workgroup that reads a zero slot has no rows
    // and leaves): block `at` clears slot `at`, the first blocks also the up to 7 slots behind the last block
    const uint32_t n = tiles_x * P.ny, n8 = (n + 7u) / 8u;
    ((uint4 *)T.entries)[(size_t)(at & 7u) * n8 + (at >> 3)] = uint4{0u, 0u, 0u, 0u};
    for (size_t b = (size_t)at + n; b < (size_t)8u * n8; b += n) ((uint4 *)T.entries)[(b & 7u) * n8 + (b >> 3)] = uint4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  for (uint32_t x = threadIdx.x; x < tiles_x; x += WG) {
    uint32_t key = 0u, run = 0u;
    if (l_touched[x]) key = 1u + (bins - T.blk[3u * ((size_t)y * tiles_x + x)]);          // dearest first: bin 0 = the largest cost
    else if (x % RT_SKY_RUN_MAX == 0u || l_touched[x - 1u]) {                            // a sky run starts here
      run = 1u;
      while (x + run < tiles_x && (x + run) % RT_SKY_RUN_MAX != 0u && !l_touched[x + run]) run++;
      key = 1u + (bins - 1u);                                                              // (sky: base cost, last in the ranked order)
    }
    l_key[x] = key;
    if (key) atomicAdd(&l_hist[key - 1u], 1u);
    T.item[(size_t)y * tiles_x + x] = key ? (key | (run << 16)) : 0u;                      // key <= 1024, run <= 32; the rank in the row is added below
  }
  __syncthreads();
  for (uint32_t x = threadIdx.x; x < tiles_x; x += WG) {
    const uint32_t key = l_key[x];
    if (!key) continue;
    uint32_t before = 0u;                                                                  // entries of the same cost to the left: the grid's order among equals
    for (uint32_t i = 0; i < x; i++) before += (l_key[i] == key) ? 1u : 0u;
    T.rank_in_row[(size_t)y * tiles_x + x] = before;
  }
  for (uint32_t c = threadIdx.x; c < bins; c += WG) T.row_hist[(size_t)y * bins + c] = l_hist[c];
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (uint32_t d = 1; d < 64u; d <<= 1) {
    const uint32_t o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

__global__ void __launch_bounds__(1024) rt_table_scan(const rt_table_dev T) {
  __shared__ uint32_t tot[RT_COST_MAX + 1u];
  const rt_table_params &P = *T.params;
  const uint32_t bins = P.cost_bins, ny = P.ny, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  // per cost class: exclusive prefix of the rows' counts down the rows, in place; the class total
  for (uint32_t c = wave; c < bins; c += n_waves) {
    uint32_t carry = 0u;
    for (uint32_t y0 = 0; y0 < ny; y0 += 64u) {
      const uint32_t y = y0 + lane;
      const uint32_t v = y < ny ? T.row_hist[(size_t)y * bins + c] : 0u;
      const uint32_t inc = wave_inclusive_scan(v);
      if (y < ny) T.row_hist[(size_t)y * bins + c] = carry + inc - v;
      carry += __shfl(inc, 63);
    }
    if (lane == 0u) tot[c] = carry;
  }
  __syncthreads();
  // where each class starts (dearest first), and the number of entries
  if (wave == 0u) {
    uint32_t carry = 0u;
    for (uint32_t c0 = 0; c0 < bins; c0 += 64u) {
      const uint32_t c = c0 + lane;
      const uint32_t v = c < bins ? tot[c] : 0u;
      const uint32_t inc = wave_inclusive_scan(v);
      if (c < bins) T.bin_start[c] = carry + inc - v;
      carry += __shfl(inc, 63);
    }
    if (lane == 0u) {
      T.header[0] = carry; T.header[1] = (carry + 7u) / 8u; T.header[2] = 0u; T.header[3] = 0u;
      if (T.known) __hip_atomic_store(T.known, ((unsigned long long)T.known_tag << 32) | (carry + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void __launch_bounds__(WG) rt_table_emit(const rt_table_dev T) {
  const rt_table_params &P = *T.params;
  const uint32_t n = P.tiles_x * P.ny, at = blockIdx.x * WG + threadIdx.x;
  if (at >= n) return;
  const uint32_t it = T.item[at];
  if (!it) return;
  const uint32_t y = at / P.tiles_x, x = at - y * P.tiles_x;
  const uint32_t bin = (it & 0xffffu) - 1u, run = it >> 16;
  const uint32_t b = T.bin_start[bin] + T.row_hist[(size_t)y * P.cost_bins + bin] + T.rank_in_row[at];     // the workgroup that renders this entry
  const uint32_t n8 = (n + 7u) / 8u;                  // the table's stride: from the number of BLOCKS (known to the host before the build)
  uint32_t w0, w1;
  rt_block_place(P, x, y, &w0, &w1);
  const bool words23 = (P.flags & RT_TABLE_GEOMETRY) != 0u;
  uint4 e;
  e.x = w0;
  e.y = w1 | (run ? (0x80000000u | ((run - 1u) << 24)) : 0u);
  e.z = (words23 && (P.flags & RT_TABLE_MASKS)) ? T.blk[3u * (size_t)at + 1u] : 0xffffffffu;
  e.w = (words23 && (P.flags & RT_TABLE_CANDS)) ? T.blk[3u * (size_t)at + 2u] : 0u;
  ((uint4 *)T.entries)[(size_t)(b & 7u) * n8 + (b >> 3)] = e;
}

}  // namespace

// A few KB from PINNED host memory (a staging slot) into device memory, by one workgroup on `stream`: what follows a camera move (the
// scene's camera block, a launch table's parameters).  A copy engine would do it too, but its hand-overs to and from the compute
// queue cost more than the copy.  `bytes` is rounded up to 16 (both buffers are 256-byte aligned and padded).
__global__ void __launch_bounds__(256) rt_small_copy(uint4 *dst, const uint4 *src, uint32_t n16) {
  for (uint32_t i = threadIdx.x; i < n16; i += 256u) dst[i] = src[i];
}
extern "C" int rt_launch_small_copy(void *dst, const void *pinned_src, size_t bytes, hipStream_t stream) {
  hipLaunchKernelGGL(rt_small_copy, dim3(1), dim3(256), 0, stream, (uint4 *)dst, (const uint4 *)pinned_src, (uint32_t)((bytes + 15u) / 16u));
  return (int)hipGetLastError();
}

// Enqueue the three launches on `stream`.  Returns a hipError_t as int.
extern "C" int rt_launch_table_build(const rt_table_dev *T, uint32_t tiles_x, uint32_t ny, uint32_t cost_bins, hipStream_t stream) {
  const uint32_t n = tiles_x * ny;
  hipLaunchKernelGGL(rt_table_rows, dim3(ny), dim3(WG), (2u * tiles_x + cost_bins) * sizeof(uint32_t), stream, *T);
  hipLaunchKernelGGL(rt_table_scan, dim3(1), dim3(1024), 0, stream, *T);
  hipLaunchKernelGGL(rt_table_emit, dim3((n + WG - 1u) / WG), dim3(WG), 0, stream, *T);
  return (int)hipGetLastError();
}
