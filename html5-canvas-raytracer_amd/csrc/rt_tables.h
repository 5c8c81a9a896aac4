// rt_tables.h — host-built tables of a scene (pure host logic, rt_tables.cpp); shared by rt_api.hip only.
#ifndef RT_TABLES_H
#define RT_TABLES_H

#include <stdint.h>

#include <vector>

#include "rt_block.h"
#include "rt_device.h"

namespace rt_tables {

// {x_lo, x_hi, y_lo, y_hi}: bounds of X/D and Y/D over the pixels whose line of sight meets the sphere (infinite where unbounded)
rt_geom cull_rect(const rt_scene_header *hd, const rt_sphere &o);

// the sphere that strictly contains every other sphere, every light and the camera (a skybox), or ~0u
uint32_t enclosing_sphere(const rt_scene_header *hd, const rt_sphere *ob, const double lights[][3]);

// per-light grids of sphere bit sets for the shadow scans of many-sphere scenes (layout: rt_tables.cpp)
std::vector<uint64_t> build_shadow_grid(const rt_sphere *objs, uint32_t n_loop, uint32_t n_lights, const double lights[][3]);

// per (sphere, cube-map direction cell) bit sets of the spheres a ray leaving that sphere in that direction can meet
std::vector<uint64_t> build_bounce_table(const rt_sphere *objs, uint32_t n_objects, uint32_t n_loop);

// per sphere: its cull rectangle (scene order) and the cost weight of a tile that shows it (launch-table ranking)
void scene_tile_weights(const rt_scene_header *hd, const rt_sphere *ob, std::vector<rt_geom> *cull, std::vector<uint32_t> *weight);

// the product kernel's launch table for `tiles` of a w x h frame (RT_ENTRY_WORDS words per workgroup, XCD-contiguous layout); empty if the
// launch is too large for the table.  mark_sky: set bit 31 of the second word of every workgroup none of whose primary rays can meet a
// sphere other than `sky_sphere` (scene order; ~0u: none) - the kernel stores the background constant there and skips everything else;
// consecutive marked blocks of a row block share ONE entry (run length - 1 in bits 24..30); `ob` = the scene's sphere records
// (host copy, scene order); *n_entries = the number of entries = workgroups of the launch;
// name_candidates: word 3 = the (at most two) loop spheres the block's primary rays can meet (count << 16 | second << 8 | first),
// or 0.  shadow_masks (at most two lights): word 2 of an entry = per light the 16-bit set of loop-order spheres that can
// shadow a PRIMARY hit of the block, 0xffffffff = scan everything
std::vector<uint32_t> build_launch_table(const rt_scene_header *hd, const rt_sphere *ob, const std::vector<rt_geom> &cull, const std::vector<uint32_t> &weight,
                                         uint32_t w, uint32_t h, uint32_t ss,
                                         const rt_tiles *tiles, uint32_t tiles_x, uint32_t rb_per_tile, double proj_w, double proj_h, double proj_d, int ranked,
                                         bool mark_sky, uint32_t sky_sphere, bool shadow_masks, bool name_candidates, const double lights[][3], uint32_t *n_entries, uint32_t sky_part = 0u);
// the parameters, cone-test spheres and cost rectangles of a launch table (rt_block.h), shared by the host build and the GPU build;
// returns non-zero when the launch is beyond the table
int make_table_params(const rt_scene_header *hd, const rt_sphere *ob, const std::vector<rt_geom> &cull, const std::vector<uint32_t> &weight,
                      uint32_t w, uint32_t h, uint32_t ss, const rt_tiles *tiles, uint32_t tiles_x, uint32_t rb_per_tile, double proj_w, double proj_h, double proj_d,
                      int ranked, bool mark_sky, uint32_t sky_sphere, bool shadow_masks, bool name_candidates, const double lights[][3],
                      rt_table_params *P, std::vector<rt_ball> *balls, std::vector<rt_cost_rect> *rects);
constexpr uint32_t RT_ENTRY_WORDS = 4u;      // {tile_x | rows_valid << 11 | first frame row << 15, band row | run << 24 | sky << 31, shadow masks, primary candidates}

}  // namespace rt_tables

#endif
