// Kernel-side view of a resident scene and of one launch.  Shared by rt_api.cpp (host) and
// rt_kernel.hip (device).  Not part of the ABI (include/rt_hip.h is).
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <stdint.h>
#include "../../include/rt_hip.h"

// Workgroup geometry: 256 work-items = 4 waves of 64.  A wave owns an 8x8 pixel block (compact,
// so its 64 rays stay coherent); the 4 waves sit side by side, so a workgroup owns a 32x8 tile
// whose every row is one whole 128-byte line of the RGBA8 framebuffer.
#ifndef RT_WG_THREADS
#define RT_WG_THREADS 256
#endif
#define RT_TILE_W (RT_WG_THREADS / 8)   /* waves side by side, 8 pixels each */
/* Which product launches run ONE-WAVE workgroups (four per launch-table entry; rt_kernel.hip: W1): the reflection-only variants -
 * the many-sphere ones stage nothing, the few-sphere ones two 16-byte loads per work-item; 10 doubles of fold state per lane -
 * unless the launch stores through the peer-store path without supersampling, which puts whole 128-byte lines together across a
 * workgroup's four waves (callers pass `scatter` = "that path": a 2x2 launch stores per wave there too).
 * (The general kernel's 13-double fold state and LDS image would leave a CU 16 one-wave workgroups: it keeps four waves.)
 * Host (LDS size, scratch figure) and launcher ask the same question. */
static inline bool rt_one_wave_workgroups(bool strict, bool count, bool refract, bool scatter) {
  return !strict && !count && !refract && !scatter;
}
#define RT_TILE_H 8

// Shadow grid (product kernel, scenes with more than RT_SGRID_MIN_LOOP spheres in the loops): cells per axis of a
// light's projective view of the scene; see build_shadow_grid in rt_tables.cpp for the buffer layout.
#define RT_MAX_SCATTER 16u          /* frames of one rt_render_scatter_device call */
#ifndef RT_SGRID
#define RT_SGRID 32u
#endif
#define RT_SGRID_MIN_LOOP 12u
// Bounce table (same scenes): directions are binned on a cube map of RT_BGRID x RT_BGRID cells per face; see
// build_bounce_table in rt_tables.cpp.
#ifndef RT_BGRID
#define RT_BGRID 8u
#endif
#define RT_BTABLE_MIN_LOOP 24u     /* measured: +10 % on the 64-sphere scene, -2 % on the reference's 14-sphere scene */
#define RT_BCELLS (6u * RT_BGRID * RT_BGRID)

// Sphere geometry as the uniform object loops read it: 32 bytes, scalar-loaded (s_load_dwordx8).
struct rt_geom { double ox, oy, oz, r2; };

// A sphere's material as the workgroup's LDS copy holds it: what the kernel reads per hit, packed by the host from rt_sphere
// (20 doubles instead of 24: 64 spheres + the fold state then fit 32 KB, i.e. five workgroups per CU instead of four).
struct rt_mtl {
  double origin[3];
  double inv_r;                      // 1 / radius: the product kernel's normal is (h - origin) * inv_r
  double albedo[5];
  double specular_exponent;
  double refract_index;
  int32_t sampler_kind, texture;
  double c[8];                       // colour: c[0..2] = mtl.color; checker: c[0..5] = its two colours, c[6..7] = its frequencies;
};                                   // stars: c[6..7] = threshold and gain (rt_sphere.checker_freq)

// Everything one launch needs, passed by value in the kernarg segment (scalar-loaded into
// SGPRs: all of it is wave-uniform).
struct rt_launch {
  // resident scene (HBM)
  const rt_sphere *objects;          // n_objects records of 192 B (materials; staged into LDS per workgroup)
  const rt_geom *geom;               // n_objects compact geometry records for the scalar-loaded loops
  const rt_geom *geom_cam;           // anchored at the camera: {o - cam, |o - cam|^2 - r2} per sphere
  const void *lds_image;             // [materials (n_objects x rt_mtl) | 16 texture descriptors | cull rectangles if cull_in_lds]: the workgroup's LDS image
  const rt_geom *cull;               // per sphere {x_lo, x_hi, y_lo, y_hi}: bounds of X/D, Y/D of the pixels whose line meets it
  const void *shadow_grid;           // per-light shadow grids (rt_tables.cpp: build_shadow_grid), or NULL when the scene is small
  const void *bounce_table;          // per (sphere a ray starts on, direction cell): bit set of the spheres it can meet, or NULL
  const rt_geom *geom_light;         // anchored at light k: [k*n_objects + j] = {o_j - light_k, |o_j - light_k|^2 - r2_j}
  const rt_texture_desc *textures;   // texels_offset is relative to `texel_base`
  const uint8_t *texel_base;
  uint32_t *out;                     // RGBA8 packed little-endian (R in the low byte)
  unsigned long long *counters;      // rays, shadow rays, sphere tests (COUNT variant only)
  // camera + projection (main.js:85-105), projection constants computed on the host in binary64
  double cam_origin[3], cam_axis_x[3], cam_axis_y[3], cam_axis_z[3];
  double cam_axis_sum[3];            // axisX[k] + axisY[k] + axisZ[k] (product kernel's ray generation)
  double ray_bias[3];                // product kernel: {0.5 - proj_w, proj_h - 0.5, axis_sum.z * proj_d}
  double proj_w, proj_h, proj_d;     // of the SAMPLE grid (2w x 2h when supersampling)
  double epsilon, light_intensity, miss_color[3];
  double lights[RT_MAX_LIGHTS][3];
  uint32_t n_objects, n_lights, segs;
  uint32_t w, h;                     // output frame size in pixels
  uint32_t tile_rows, tile_first, tile_stride, n_tiles;   // rt_tiles
  uint32_t tiles_x;                  // workgroup tiles per row of the frame
  uint32_t rb_per_tile, rb_shift;    // workgroup row blocks per tile; log2 of it when a power of two, else ~0u
  uint32_t n_frames;                 // frames of the batch (grid z); frame f is written at out + f*frame_stride
  uint64_t frame_stride;             // in 32-bit words (= pixels for RGBA8 output)
  uint32_t n_loop;                   // spheres the per-ray loops walk (n_objects, or n_objects-1 when `enclosing` is set)
  uint32_t enclosing;                // device index (== n_loop, the table's last entry) of a sphere that strictly contains
                                     // every other sphere, every light and the camera; ~0u if none or not used
  uint32_t sky_fast;                 // ... and its colour is a constant: a wave whose primary rays all miss the loop spheres stores sky_rgb
  double sky_rgb[3];
  uint32_t enclosing_flat;           // that sphere neither lights nor spawns rays and its colour ignores the hit point: its test is skipped
  uint32_t cull_in_lds;              // 1: few spheres (no shadow grid, no bounce table): the cull rectangles are part of the LDS image, the product launch takes the few-sphere kernel; 0: the many-sphere kernel, every lane fetches its sphere's rectangle from `cull` (HBM / L2)
  uint32_t rgb24;                    // RT_FLAG_RGB24: rows are w*3 bytes (R,G,B), no alpha byte; w % 4 == 0
  uint32_t compact;                  // RT_FLAG_COMPACT (with rgb24): block b of the launch is stored whole at out + b * block bytes (a compact band for a collective)
  uint32_t scatter;                  // rt_render_scatter_device: frame f goes to out_frames[f] (possibly another GPU's memory,
                                     // peer-mapped), its rows in FRAME order; `out` and frame_stride are unused
  uint32_t *out_frames[RT_MAX_SCATTER];
  // Launch table of the product kernel (rt_api.hip: dispatch_order): workgroup b of the flat grid renders the tile described by
  // entry b = {tile_x | rows_valid << 11 | first frame row << 15, first row in the output band}; the table build (rt_tables_gpu.hip) lists the tiles
  // dearest first, so that a launch ends on cheap tiles.  The strict kernel runs on the plain 2-D grid and ignores it.
  const uint32_t *order;             // the entries; the four words in front of them are the table's header {entries, ceil(entries / 8), 0, 0}
  uint32_t order_n8;                 // ceil(BLOCKS / 8): entry of workgroup b sits at (b % 8) * order_n8 + b / 8 (one contiguous part per XCD)
                                     // (the slots behind the last entry are zero: a workgroup that reads one has no rows and leaves - the host launches one
                                     // workgroup per BLOCK while it does not know the number of entries of a table built on the GPU a moment ago)
  uint32_t grid_x, grid_y;           // host side only: the product launch's flat grid (one workgroup per table entry; 0 = the plain 2-D grid)
  // Samples on exact coincidences.  A sample whose outcome in the reference hinges on the last bit of the reference's own
  // arithmetic is traced a second time with the reference's own operation sequence by the strict build's list-driven kernel
  // (rt_kernel.hip: rt_retrace; rt_api.hip launches it after the product launch): the samples the product kernel MARKED - a sampler
  // coordinate within rounding of a texel / checker boundary (main.js:129-130, 344-347) - and the centre row / column of an odd
  // sample grid (a primary ray with an exactly-zero component, main.js:186).
  uint32_t *marks;                   // words 0, 1: the mark counters of alternate launches; words 4...: entries of 8 bytes (sample x | y << 20 | frame << 40)
  uint32_t marks_slot;               // the counter THIS launch (and its rt_retrace) uses; rt_retrace clears the other
  uint32_t marks_cap;                // entries the list holds; beyond it only the count grows and rt_retrace traces every sample
  double flag_tol;                   // tolerance of the boundary test: RT_FLAG_T1 x the largest sampler frequency of the scene
  uint32_t mark_flags;               // RT_MARK_* (below)
  uint32_t mark_pad;
  // rt_retrace only
  unsigned long long *marks_known;   // pinned host word that receives known_tag << 32 | count + 1, or NULL
  uint32_t known_tag;                // (the camera generation the frame is rendered with)
  uint32_t centre_row, centre_col;   // frame row / column of the odd sample grid's centre within this call's tiles, or ~0u
  uint32_t retrace_all;              // test build (RT_EXACT_ALL): every sample of the call
  // ... of a compact launch: the table's arrays, from which a sample's block finds its place in the band (rt_tables_gpu.hip: rt_table_emit's arithmetic)
  const uint32_t *tb_item, *tb_rank_in_row, *tb_row_hist, *tb_bin_start;
  uint32_t tb_bins;
  uint32_t four_waves;               // host side only: this product launch takes the four-wave form of its kernel although it could run one-wave
                                     // workgroups (rt_one_wave_workgroups): the first frame from a new camera, beside which the NEXT camera's launch
                                     // table may be built (rt_scene_set_camera) - one-wave workgroups take every slot the moment it frees, and the
                                     // build's four-wave workgroups would wait for the trace to drain
#ifdef RT_WAVE_LOG
  // measurement builds only (profiles/ab_build.sh ... "-DRT_WAVE_LOG" hybrid; profiles/wave_timeline.py): per wave of the product launch
  // four words {s_memrealtime at entry, at exit, HW_ID | XCC_ID << 32, workgroup}; NULL = off
  unsigned long long *wave_log;
#endif
#ifdef RT_TESTING
  // test build only (librt_hip_test.so): per-node records of ONE sample's ray tree, for parity debugging
  double *probe;                     // RT_PROBE_NODES records of RT_PROBE_WORDS doubles, or NULL
  uint32_t probe_x, probe_y;         // the sample, in sample-grid coordinates
#endif
};

// Boundary test of the samplers (product kernels): a sampler coordinate x = u * frequency is "on a boundary" when it lies within
// RT_FLAG_T1 * (the scene's largest frequency) of an integer: 1e-9 for the reference's checker (5000 per unit u, main.js:129), three
// to four orders of magnitude above what the product kernel's u, v differ from the reference's by at a primary or a shallow bounce
// hit (u to ~1e-16, a coordinate to ~1e-12), and rare enough (4e-9 per sampled hit) that a 3840x2160 frame has a marked sample once
// in ~50 frames.  Marked samples are traced again by rt_retrace.  RT_MARKS_CAP: entries of a launch's mark list.
#define RT_FLAG_T1 2e-13
#define RT_MARKS_CAP 8192u
#define RT_MARK_ALL 4u               /* test build (RT_MARK_ALL): every hit with a texture / checker sampler is marked */
#define RT_MARK_ZERO 16u             /* test build (RT_TEST_MARK_STRIPES): a coordinate with frequency 0 is marked like any exact integer (round 3's behaviour: overflows the list) */
#define RT_MARK_WEIGHT 32u           /* every albedo and colour of the scene lies in [0, 1]: a sample whose accumulated weight is below a byte's worth is not marked */
#define RT_MARK_NEVER 8u             /* test build (RT_NO_FIXUP): nothing is marked - the product kernel's own pixels */

#define RT_PROBE_WORDS 24u
#define RT_PROBE_NODES 64u

#endif
