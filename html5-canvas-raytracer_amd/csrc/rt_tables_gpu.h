// rt_tables_gpu.h — the device-side build of the launch table (rt_tables_gpu.hip), as rt_api.hip drives it.
#ifndef RT_TABLES_GPU_H
#define RT_TABLES_GPU_H

#include <stdint.h>

#include "rt_block.h"

// Device memory of one launch table (one per (frame size, tile set) of a resident scene; rebuilt in place when the camera moves).
struct rt_table_dev {
  const rt_table_params *params;   // the parameters (rt_tables.cpp: make_table_params), ...
  const rt_ball *balls;            // ... the spheres as the cone test sees them ...
  const rt_cost_rect *rects;       // ... and their cost rectangles: one small block, copied per build
  uint32_t *blk;                   // per block: cost, shadow masks, candidates
  uint32_t *item;                  // per block: 0, or (1 + cost bin) | run << 16 where an entry starts
  uint32_t *rank_in_row;           // per block: entries of the same cost to its left in the row
  uint32_t *row_hist;              // [row blocks][cost bins]: entries per cost, then their exclusive prefix down the rows
  uint32_t *bin_start;             // [cost bins]: first workgroup of a cost class
  uint32_t *ticket;                // one word, zero between builds: which wave of rt_table_scan finishes last
  uint32_t *header;                // the table: 4 words {entries, ceil(entries / 8), 0, 0} ...
  uint32_t *entries;               // ... and 8 * ceil(blocks / 8) slots of 16 bytes behind them
  unsigned long long *known;       // pinned host word that receives known_tag << 32 | entries + 1, or NULL
  uint32_t known_tag;              // (the camera generation the table is built for)
};

#ifdef __HIPCC__
extern "C" int rt_launch_small_copy(void *dst0, const void *pinned_src0, size_t bytes0, void *dst1, const void *pinned_src1, size_t bytes1, hipStream_t stream);
extern "C" int rt_launch_table_build(const rt_table_dev *T, uint32_t tiles_x, uint32_t ny, uint32_t cost_bins, uint32_t dyn_bytes, int wide, hipStream_t stream);   // dyn_bytes: of (params | balls | rects), contiguous from T->params; wide: RT_TABLE_WIDE (many spheres)
#endif

#endif
