// rt_tables_gpu.hip — the product kernel's launch table, built ON THE GPU (what rt_api.hip: dispatch_order runs; rt_tables.cpp's
// build_launch_table is the same table built on the host: the CPU tests' oracle and the -m gpu test that compares the two).
//
// The reference recomputes everything on every redraw() (main.js:180-201) and its camera is a parameter (lookAt, main.js:92-100).
// The table - per block of 32 x 8 pixels: is it sky, which spheres can its primary rays meet, which spheres can shadow its hits,
// what will it cost (rt_block.h) - depends on the camera, the frame size and the tile set; on the host it takes 5-30 ms for a
// 3840x2160 frame against a 0.07 ms trace.  Here it is three small launches on the render stream, no host step in between:
//
//   rt_table_rows   one workgroup per ROW BLOCK (a row of blocks): eight work-items state a block (the pieces of rt_block_statement,
//                   rt_block_cost: the host's own source), then the row decides in LDS which blocks start an entry - a block that shows a sphere is an
//                   entry of its own, consecutive sky blocks share one (a run never crosses a multiple of 32 blocks, so "do I start a
//                   run" is a question about the left neighbour) - how many entries of the same cost lie to the left of each (its
//                   rank among equals in the row), and the row's histogram of entry costs;
//   rt_table_scan   one wave per cost: the exclusive prefix of the rows' histograms down the rows, the total; the wave that finishes
//                   last (a ticket) scans the totals from the dearest cost down: where each cost class starts in the dispatch
//                   order.  Counting sort, stable: equal costs keep the grid's order, as on the host.  It
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

constexpr uint32_t WG = 256;          // rt_table_emit
// OR over the SUB adjacent lanes that share a block (a group never straddles a wave; groups are active or idle as a whole)
template <uint32_t SUB>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
#pragma unroll
  for (uint32_t d = 1; d < SUB; d <<= 1) v |= (uint32_t)__shfl_xor((int)v, (int)d);
  return v;
}
template <uint32_t SUB>
__device__ __forceinline__ uint32_t group_add(uint32_t v) {
#pragma unroll
  for (uint32_t d = 1; d < SUB; d <<= 1) v += (uint32_t)__shfl_xor((int)v, (int)d);
  return v;
}
template <uint32_t SUB>
__device__ __forceinline__ uint64_t group_or64(uint64_t v) { return (uint64_t)group_or<SUB>((uint32_t)v) | ((uint64_t)group_or<SUB>((uint32_t)(v >> 32)) << 32); }

// The parameters, the cone-test spheres and the cost rectangles are staged in LDS first - one coalesced load per work-item - and
// read from there: every workgroup is the first on its CU to touch them (a copy kernel wrote them a moment ago), and as scalar
// loads through the cold constant cache the ~30 dependent lines cost 20 of the kernel's 27 us (profiles/r03_ab_log.md).
// `stage_bytes` = what of (params | balls | rects), contiguous from T.params on, goes to LDS.  STAGED = false: too many spheres for
// it - they are read where they are.  (Two instantiations, so that in the staged one every read is known to be an LDS read: through
// a pointer that may be either, each field is a flat load and its own wait - 20 us of dependent latencies per block.)
//
// SUB work-items per block.  With one, a 3840x2160 frame's 32 400 blocks are 506 waves - one per SIMD on half of the chip - and
// each of them walks, one after the other, every sphere that ANY of its 64 blocks names a candidate (the root interval, the patch,
// per light a loop over the occluders: ~2800 instructions per wave at ~20 cycles each, a lone wave waiting for its own LDS reads
// and square roots).  Here work-item `sub` of a block's group takes spheres sub, sub + SUB, ...: the cone (computed by all of
// them, the same bits), its spheres' cone tests, and for those that are candidates their shadow masks; candidate set, masks and
// "no statement" are OR-ed across the group (rt_block.h's pieces: the same operations as the host's rt_block_statement, in another
// order only where the order cannot matter).  Measured (profiles/r03_ab_log.md section 3; 3840x2160): 8 spheres 24.0 us with one
// work-item per block, 19.6 with four (512-thread workgroups), 27.4 with eight; 64 spheres 817 us with one, 273 with eight.
template <bool STAGED, uint32_t SUB, uint32_t WG_ROWS>
__global__ void __launch_bounds__(WG_ROWS) rt_table_rows(const rt_table_dev T, uint32_t stage_bytes) {
  extern __shared__ uint4 lds_raw[];
  if (STAGED) {
    const uint4 *src = (const uint4 *)T.params;
    for (uint32_t i = threadIdx.x; i < stage_bytes / 16u; i += WG_ROWS) lds_raw[i] = src[i];
    __syncthreads();
  }
  const uint8_t *base = STAGED ? (const uint8_t *)lds_raw : (const uint8_t *)T.params;
  const rt_table_params &P = *(const rt_table_params *)base;
  const rt_ball *balls = (const rt_ball *)(base + ((const uint8_t *)T.balls - (const uint8_t *)T.params));
  const rt_cost_rect *rects = (const rt_cost_rect *)(base + ((const uint8_t *)T.rects - (const uint8_t *)T.params));
  uint32_t *lds = (uint32_t *)((uint8_t *)lds_raw + stage_bytes);
  const uint32_t y = blockIdx.x, tiles_x = P.tiles_x, bins = P.cost_bins;
  uint32_t *l_touched = lds;                 // [tiles_x] 1: shows a sphere (or no sky marking at all)
  uint32_t *l_key = lds + tiles_x;           // [tiles_x] entry starts: 1 + cost bin; 0: not an entry start
  uint32_t *l_hist = lds + 2u * tiles_x;     // [bins]
  const bool sky = (P.flags & RT_TABLE_SKY) && (P.flags & RT_TABLE_GEOMETRY);
  const bool rank = (P.flags & RT_TABLE_RANK) != 0u;
  for (uint32_t c = threadIdx.x; c < bins; c += WG_ROWS) l_hist[c] = 0u;
  const uint32_t sub = threadIdx.x % SUB;
  for (uint32_t x = threadIdx.x / SUB; x < tiles_x; x += WG_ROWS / SUB) {
    uint32_t touched = 1u, cands = 0u, smask = 0xffffffffu, extra = 0u;
    if (P.flags & RT_TABLE_GEOMETRY) {
      const rt_cone K = rt_block_cone(P, x, y);
      uint64_t cand[4] = {0ull, 0ull, 0ull, 0ull};
      uint32_t everywhere = 0u;
      if (!K.doubt)
        for (uint32_t j = sub; j < P.n_balls; j += SUB) {
          const uint32_t t = rt_ball_touch(K, balls[j]);
          if (t == 2u) everywhere = 1u;
          else if (t) cand[j >> 6] |= 1ull << (j & 63u);
        }
      everywhere = group_or<SUB>(everywhere);
      uint32_t n_cand = 0u;
#pragma unroll
      for (uint32_t wd = 0; wd < 4u; wd++) {
        if (wd * 64u < P.n_balls) cand[wd] = group_or64<SUB>(cand[wd]);      // (uniform: every work-item of the grid sees the same n_balls)
        n_cand += (uint32_t)__popcll(cand[wd]);
      }
      const bool doubt = K.doubt || everywhere;
      touched = (K.hit || everywhere || n_cand) ? 1u : 0u;
      if (!doubt && n_cand) {
        if (sub == 0u) cands = rt_cand_word(P, balls, cand, n_cand);
        if (P.flags & RT_TABLE_BOUNCE) {                // (ranking only) what the block's mirrors show of the scene's dearest spheres
          for (uint32_t ci = sub; ci < P.n_balls; ci += SUB) if ((cand[ci >> 6] >> (ci & 63u)) & 1ull) extra += rt_bounce_cost(P, K, balls, ci);
          extra = group_add<SUB>(extra);
        }
        if (P.flags & RT_TABLE_MASKS) {
          uint32_t mk[2] = {0u, 0u}, none = 0u;
          for (uint32_t ci = sub; ci < P.n_balls; ci += SUB)
            if ((cand[ci >> 6] >> (ci & 63u)) & 1ull) {
              if (!rt_cand_masks(P, K, balls, ci, mk)) none = 1u;
              if (mk[0] == 0xffffu && mk[1] == 0xffffu) break;       // (as on the host: the word is 0xffffffff whatever follows)
            }
          none = group_or<SUB>(none);
          const uint32_t m0 = group_or<SUB>(mk[0]), m1 = group_or<SUB>(mk[1]);
          if (!none) smask = m0 | (m1 << 16);
        }
      }
    }
    if (sub != 0u) continue;
    uint32_t cost = rank ? rt_block_cost(P, rects, x, y) + extra : 1u;
    cost = cost < RT_COST_MAX ? cost : RT_COST_MAX;
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
  for (uint32_t x = threadIdx.x; x < tiles_x; x += WG_ROWS) {
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
  for (uint32_t x = threadIdx.x; x < tiles_x; x += WG_ROWS) {
    const uint32_t key = l_key[x];
    if (!key) continue;
    uint32_t before = 0u;                                                                  // entries of the same cost to the left: the grid's order among equals
    for (uint32_t i = 0; i < x; i++) before += (l_key[i] == key) ? 1u : 0u;
    T.rank_in_row[(size_t)y * tiles_x + x] = before;
  }
  for (uint32_t c = threadIdx.x; c < bins; c += WG_ROWS) T.row_hist[(size_t)y * bins + c] = l_hist[c];
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

// One wave per cost class: the exclusive prefix of the rows' counts down the rows (in place) and the class total; the wave that
// finishes LAST (a ticket) turns the totals into class starts, writes the header and publishes the count.
__global__ void __launch_bounds__(64) rt_table_scan(const rt_table_dev T) {
  const rt_table_params &P = *T.params;
  const uint32_t bins = P.cost_bins, ny = P.ny, lane = threadIdx.x, c = blockIdx.x;
  uint32_t carry = 0u;
  for (uint32_t y0 = 0; y0 < ny; y0 += 1024u) {
    uint32_t v[16];
#pragma unroll
    for (uint32_t i = 0; i < 16u; i++) { const uint32_t y = y0 + i * 64u + lane; v[i] = y < ny ? T.row_hist[(size_t)y * bins + c] : 0u; }   // sixteen loads in flight
#pragma unroll
    for (uint32_t i = 0; i < 16u; i++) {
      const uint32_t y = y0 + i * 64u + lane;
      const uint32_t inc = wave_inclusive_scan(v[i]);
      if (y < ny) T.row_hist[(size_t)y * bins + c] = carry + inc - v[i];
      carry += __shfl(inc, 63);
    }
  }
  uint32_t last = 0u;
  if (lane == 0u) {
    __hip_atomic_store(&T.bin_start[c], carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the class total, for now
    __threadfence();
    last = (atomicAdd(T.ticket, 1u) == bins - 1u) ? 1u : 0u;
  }
  if (!__shfl(last, 0)) return;
  __threadfence();
  // where each class starts (dearest first), and the number of entries
  uint32_t total = 0u;
  for (uint32_t c0 = 0; c0 < bins; c0 += 64u) {
    const uint32_t k = c0 + lane;
    const uint32_t v = k < bins ? __hip_atomic_load(&T.bin_start[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const uint32_t inc = wave_inclusive_scan(v);
    if (k < bins) T.bin_start[k] = total + inc - v;
    total += __shfl(inc, 63);
  }
  if (lane == 0u) {
    *T.ticket = 0u;                                       // for the next build
    // header: {entries, ceil(entries / 8), entries in front of the sky runs' class (a ranked table: its non-sky blocks; else all), 0}
    T.header[0] = total; T.header[1] = (total + 7u) / 8u; T.header[2] = (P.flags & RT_TABLE_RANK) ? T.bin_start[bins - 1u] : total; T.header[3] = 0u;
    if (T.known) __hip_atomic_store(T.known, ((unsigned long long)T.known_tag << 32) | (total + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
  // a frame put together from several GPUs' tiles: the senders' tables hold no sky entries, the owner's fill table nothing else
  // (ranks and counts are those of the whole table; a slot without an entry stays zero and its workgroup leaves at once)
  if (run ? (P.flags & RT_TABLE_NO_SKY) : (P.flags & RT_TABLE_SKY_ONLY)) return;
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
// scene's camera block, a launch table's parameters: up to two pieces in one launch).  A copy engine would do it too, but its hand-overs to and from the compute
// queue cost more than the copy.  `bytes` is rounded up to 16 (both buffers are 256-byte aligned and padded).
__global__ void __launch_bounds__(256) rt_small_copy(uint4 *dst0, const uint4 *src0, uint32_t n0, uint4 *dst1, const uint4 *src1, uint32_t n1) {
  for (uint32_t i = threadIdx.x; i < n0 + n1; i += 256u) { if (i < n0) dst0[i] = src0[i]; else dst1[i - n0] = src1[i - n0]; }
}
extern "C" int rt_launch_small_copy(void *dst0, const void *pinned_src0, size_t bytes0, void *dst1, const void *pinned_src1, size_t bytes1, hipStream_t stream) {
  hipLaunchKernelGGL(rt_small_copy, dim3(1), dim3(256), 0, stream, (uint4 *)dst0, (const uint4 *)pinned_src0, (uint32_t)((bytes0 + 15u) / 16u),
                     (uint4 *)dst1, (const uint4 *)pinned_src1, (uint32_t)((bytes1 + 15u) / 16u));
  return (int)hipGetLastError();
}

// Enqueue the three launches on `stream`.  Returns a hipError_t as int.
extern "C" int rt_launch_table_build(const rt_table_dev *T, uint32_t tiles_x, uint32_t ny, uint32_t cost_bins, uint32_t dyn_bytes, int wide, hipStream_t stream) {
  // wide & 2: the build runs BESIDE a trace kernel that keeps every CU full (rt_scene_set_camera's side stream): four-wave workgroups,
  // which find room whenever ONE trace workgroup retires - an eight- or sixteen-wave workgroup needs two or four of them to retire on
  // the same CU at once and waited for the trace to drain (77 us instead of 20: profiles/r04_ab_log.md)
  const bool beside = (wide & 2) != 0;
  wide &= 1;
  const uint32_t n = tiles_x * ny;
  const uint32_t stage_bytes = dyn_bytes <= 40u * 1024u ? ((dyn_bytes + 15u) & ~15u) : 0u;      // (params | balls | rects) of up to ~170 spheres fit LDS beside the row's arrays
  const uint32_t row_bytes = (2u * tiles_x + cost_bins) * sizeof(uint32_t);
  // few spheres (no more than 16 in the loops: per-sphere shadow sets): four work-items per block; many: eight, in 16-wave workgroups
  if (!stage_bytes) hipLaunchKernelGGL((rt_table_rows<false, 8u, 1024u>), dim3(ny), dim3(1024), row_bytes, stream, *T, 0u);
  else if (wide && beside) hipLaunchKernelGGL((rt_table_rows<true, 8u, 256u>), dim3(ny), dim3(256), stage_bytes + row_bytes, stream, *T, stage_bytes);
  else if (wide) hipLaunchKernelGGL((rt_table_rows<true, 8u, 1024u>), dim3(ny), dim3(1024), stage_bytes + row_bytes, stream, *T, stage_bytes);
  else if (beside) hipLaunchKernelGGL((rt_table_rows<true, 4u, 256u>), dim3(ny), dim3(256), stage_bytes + row_bytes, stream, *T, stage_bytes);
  else hipLaunchKernelGGL((rt_table_rows<true, 4u, 512u>), dim3(ny), dim3(512), stage_bytes + row_bytes, stream, *T, stage_bytes);
  hipLaunchKernelGGL(rt_table_scan, dim3(cost_bins), dim3(64), 0, stream, *T);
  hipLaunchKernelGGL(rt_table_emit, dim3((n + WG - 1u) / WG), dim3(WG), 0, stream, *T);
  return (int)hipGetLastError();
}
