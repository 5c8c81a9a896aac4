/*
 * rt_hip.h — C ABI of the MI355X ray-sphere trace/shade path.
 *
 * This is the drop-in boundary for ONE path of termuxinator/html5-canvas-raytracer:
 * the per-pixel loop of main.js (primary-ray generation :184-193, intersectWorld
 * :220-337, intersectSphere :420-451, samplers :126-133/:343-351/:404, RGBA8 store
 * :195-198).  The reference has no FFI of its own; the entry points
 * below are what a Node N-API / Python ctypes binding for that path binds to.  Each
 * comment names the reference construct the entry point or field replaces.
 *
 * Plain C types only.  No torch, no C++ across the ABI.  All functions return 0 on
 * success and a negative rt_status on failure; rt_last_error() describes the failure
 * of the calling thread's last call.
 *
 * Threads: rt_init, rt_shutdown and rt_render serialise on an internal lock.  The device entry points
 * (rt_scene_upload, rt_render_tiles_device, rt_render_batch_device, rt_deinterleave_*) may be called from several
 * threads at once; work on one HIP stream is ordered by the stream.  Launches with RT_FLAG_COUNT share one counter
 * buffer per device: one at a time per device.
 *
 * The scene crosses the boundary as ONE contiguous, pointer-free blob
 * (rt_scene_header followed by the tables it gives offsets to), so any host language
 * can build it with typed arrays and the library can upload it with one copy.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2u
#define RT_SCENE_MAGIC 0x31535452u /* "RTS1" little endian */
/* `const build = '741'` (main.js:3): the reference build whose per-pixel path this library reproduces, and this library's own
 * revision of it; rt_build_id() returns "<reference build>.<revision>". */
#define RT_REFERENCE_BUILD "741"
#define RT_LIBRARY_REVISION "r4"

#define RT_MAX_OBJECTS  256u
#define RT_MAX_LIGHTS   16u
#define RT_MAX_TEXTURES 16u
#define RT_MAX_SEGS     16u

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,     /* malformed scene blob / bad argument */
  RT_ERR_UNSUPPORTED = -2, /* e.g. an unknown sampler kind */
  RT_ERR_DEVICE = -3,      /* HIP / RCCL failure, or no GPU */
  RT_ERR_NOMEM = -4,
  RT_ERR_STATE = -5        /* rt_init not called, bad handle */
} rt_status;

/* mtl.sampler closures of the reference, enumerated */
enum {
  RT_SAMPLER_COLOR = 0,   /* main.js:404       constant mtl.color                 */
  RT_SAMPLER_TEXTURE = 1, /* main.js:143-145   sampleTexture(tex, hit.u, hit.v)   */
  RT_SAMPLER_CHECKER = 2, /* main.js:126-133   sphere checker on its own u,v      */
  RT_SAMPLER_STARS = 3    /* main.js:135-139   night stars, with Math.random() replaced by a counter-based hash of
                           *                    (sample index in the frame, position in the ray tree): deterministic, the same
                           *                    on every implementation here, NOT comparable with the (random) reference.
                           *                    checker_freq[0] = threshold (0.001), checker_freq[1] = scale (1000). */
};

/* One sphere + its material: createSphere (main.js:408-418) + createMaterial (:397-406).
 * 24 doubles = 192 bytes. */
typedef struct rt_sphere {
  double origin[3];           /* obj.origin            main.js:411 */
  double r2;                  /* obj.r2                main.js:413 */
  double color[3];            /* mtl.color             main.js:399 */
  double specular_exponent;   /* mtl.specular_exponent main.js:401 */
  double albedo[5];           /* ambient,diffuse,specular,reflect,refract  main.js:400 */
  double refract_index;       /* mtl.refract_index     main.js:402 */
  double checker_freq[2];     /* the literals 5000, 2500 of main.js:129-130 */
  double checker_color[2][3]; /* the table of main.js:131 */
  int32_t sampler_kind;       /* RT_SAMPLER_* */
  int32_t texture;            /* index into the texture table for RT_SAMPLER_TEXTURE, else -1 */
  double reserved;
} rt_sphere;

/* createTexture (main.js:339-341); texels are RGBA8 rows, top row first (getImageData layout, :388-391) */
typedef struct rt_texture_desc {
  uint32_t width;
  uint32_t height;
  uint64_t texels_offset; /* bytes from blob start to width*height*4 bytes */
} rt_texture_desc;

typedef struct rt_scene_header {
  uint32_t magic;          /* RT_SCENE_MAGIC */
  uint32_t abi_version;    /* RT_ABI_VERSION */
  uint64_t total_bytes;    /* size of the whole blob */
  double cam_origin[3];    /* origin  main.js:85,93 */
  double cam_axis_x[3];    /* axisX   main.js:86,97 */
  double cam_axis_y[3];    /* axisY   main.js:87,98 */
  double cam_axis_z[3];    /* axisZ   main.js:88,99 */
  double fov_deg;          /* 60      main.js:102 */
  double light_intensity;  /* 50      main.js:284 (shared across lights, quirk q2) */
  double epsilon;          /* 0.001   main.js:430-436,445 */
  double miss_color[3];    /* [1,0,0] main.js:231 */
  uint32_t segs;           /* 8       main.js:194 */
  uint32_t supersample;    /* 1, or k in {2,3,4} = render kw x kh by the reference's rule and box-average every k x k block of RGBA8
                            * samples with (sum + k*k/2) / (k*k), integer division: k = 2 is (a+b+c+d+2)>>2 (cfg5).  Not in the
                            * reference: defined so that the oracle stays "main.js + an integer post-step". */
  uint32_t n_objects;      /* objs.length, already in the reference's sorted order (main.js:159-163) */
  uint32_t n_lights;       /* lights.length main.js:283 */
  uint32_t n_textures;
  uint32_t reserved0;
  uint64_t objects_offset;  /* rt_sphere[n_objects] */
  uint64_t lights_offset;   /* double[3*n_lights] */
  uint64_t textures_offset; /* rt_texture_desc[n_textures] */
} rt_scene_header;

/* Which rows of the w x h frame a call renders, as row tiles dealt round-robin:
 * tile t (t = tile_first + i*tile_stride, i in [0,n_tiles)) covers frame rows
 * [t*tile_rows, min(h,(t+1)*tile_rows)) and is stored at out + i*tile_rows*w*4.
 * The whole frame is {tile_rows=h, tile_first=0, tile_stride=1, n_tiles=1}.
 * Replaces the scanline scheduler spanish(y) (main.js:183-201). */
typedef struct rt_tiles {
  uint32_t tile_rows;
  uint32_t tile_first;
  uint32_t tile_stride;
  uint32_t n_tiles;
} rt_tiles;

/* Per-render counters and timings; counters are filled only when RT_FLAG_COUNT is set
 * (they come from an instrumented kernel variant, never from the timed one). */
typedef struct rt_stats {
  double kernel_ms;        /* hipEvent time of the trace kernel(s) on the render stream */
  double total_ms;         /* host wall time of the call */
  uint64_t pixels;         /* output pixels written */
  uint64_t rays;           /* intersectWorld invocations with segs>0 (main.js:220-221) */
  uint64_t shadow_rays;    /* lights tested for occlusion (main.js:293-304) */
  uint64_t sphere_tests;   /* intersectSphere calls (main.js:228,296) */
  uint64_t exact_samples;  /* samples the product kernel traced a second time with the reference's own operation sequence:
                            * a sampler coordinate on a texel / checker boundary (main.js:129-130, 344-347) up to rounding, or the
                            * centre row / column of an odd sample grid (main.js:186: x - w/2 + 0.5 == 0).  Every stats call. */
} rt_stats;

enum {
  RT_FLAG_NONE = 0,
  RT_FLAG_COUNT = 1,        /* run the counting variant and fill rays/shadow_rays/sphere_tests */
  RT_FLAG_STRICT_FP = 2,    /* no FMA contraction: operation-for-operation with the JS expression trees */
  RT_FLAG_RGB24 = 4,        /* device entry points only: store 3 bytes per pixel (R,G,B, rows packed, w*3 bytes each)
                             * instead of RGBA8.  The alpha byte is the constant 255 in the reference (main.js:198),
                             * so a band that is about to cross an xGMI link does not carry it; the receiving side
                             * restores it with rt_deinterleave_rgb24_device.  Needs w % 4 == 0. */
  RT_FLAG_NO_SKY = 8,       /* device entry points: blocks of 32 x 8 pixels in which nothing but a constant background can show (the
                             * reference's skybox with a plain colour, or the miss colour of main.js:231 - half of the headline frame) are
                             * NOT stored.  For a frame assembled from several GPUs' tiles in ONE buffer (rt_render_scatter_device into the
                             * owner's frame over xGMI): the senders leave the sky out ... */
  RT_FLAG_SKY_ONLY = 16,    /* ... and the frame's owner stores exactly those blocks, of whatever tiles the call names (the whole frame),
                             * from its own table: the two kinds of call together store every pixel once, and about half of the
                             * headline's pixels never cross a link.  Scenes without a constant background: RT_FLAG_NO_SKY leaves nothing
                             * out and RT_FLAG_SKY_ONLY stores nothing.  Not with RT_FLAG_COUNT. */
  RT_FLAG_COMPACT = 32      /* rt_render_batch_device with RT_FLAG_RGB24 | RT_FLAG_NO_SKY: a COMPACT band for a collective - the blocks that
                             * are stored at all (everything but the sky) back to back, block b of the launch (32 pixels x 8 rows, x 2
                             * rows with supersample 2; RGB24, row by row: 768 / 192 bytes) at d_out + b * block_bytes, dearest block
                             * first.  rt_compact_count says how many there are; the receiver, which holds the same scene with the same
                             * camera, puts them back with rt_compact_expand_device and fills the sky itself (RT_FLAG_SKY_ONLY).  Not
                             * for scenes the strict kernel renders (RT_ERR_UNSUPPORTED: send plain bands), not with RT_FLAG_COUNT. */
};

typedef struct rt_scene_dev rt_scene_dev; /* opaque: a scene resident in one GPU's HBM */

/* Library lifetime.  rt_init(max_devices): use up to max_devices GPUs (0 = all visible). */
int rt_init(int max_devices);
void rt_shutdown(void);
int rt_device_count(void);            /* GPUs in use after rt_init, or a negative rt_status */
const char *rt_last_error(void);
uint32_t rt_abi_version(void);
const char *rt_build_id(void);        /* RT_REFERENCE_BUILD "." RT_LIBRARY_REVISION, e.g. "741.r4" (main.js:3) */

/* Validate a scene blob without touching a GPU (host logic; usable in CPU-only tests). */
int rt_scene_validate(const void *scene_blob, size_t blob_bytes);

/* Host-logic probe (no GPU): the conservative screen rectangle the product kernel uses to cull spheres for
 * primary rays, per sphere in scene order: {x_lo, x_hi, y_lo, y_hi} bounding X/D and Y/D of every pixel whose
 * line meets the sphere (X = x - w/2 + 0.5, Y = h/2 - y - 0.5, D = (w/2)/tan(fov/2); +-inf = unbounded). */
int rt_scene_cull_rects(const void *scene_blob, size_t blob_bytes, double *out_4n);

/* Host-logic probe (no GPU): the candidate set the product kernel's bounce table gives a reflected / refracted ray
 * that starts on sphere `from` (scene order) with direction `dir`: bit j of out_words (ceil(n_objects/64) words) is
 * set if sphere j is tested.  Conservative by construction: every sphere such a ray can meet is in the set. */
int rt_scene_bounce_candidates(const void *scene_blob, size_t blob_bytes, uint32_t from, const double dir[3], uint64_t *out_words);

/* Host-logic probe (no GPU): the launch table of the product kernel for `tiles` of the w x h frame (scene supersample 1 or 2), as
 * the HOST builds it; the library builds the same table on the GPU (same per-block source), per camera, frame size and tile set.
 * The kernel runs on a flat grid; workgroup b renders the 32-pixel-wide, 8-row (2 with supersample 2) block described by its
 * 16-byte entry {tile_x | rows_valid << 11 | first frame row << 15, first row in this call's output band | (run - 1) << 24 |
 * sky << 31, shadow masks, primary candidates}, stored in slot (b % 8) * ceil(blocks / 8) + b / 8.  `ranked` is a bit set:
 *   1  the blocks are listed dearest first (a cost estimate from the spheres' screen rectangles), which is the order the
 *      hardware then hands them out in;
 *   2  blocks none of whose primary rays can meet a sphere are marked (sky = 1) and consecutive ones of a row block share one
 *      entry (a run of blocks that does not cross a multiple of 32 blocks), as the launch of a scene with a constant background
 *      does: their workgroup stores the background;
 *   4  word 2 = per light (16 bits each) the loop-order spheres that can shadow a primary hit of the block at all (scenes of at
 *      most 16 loop spheres and 2 lights; 0xffffffff = no statement), word 3 = the at most two spheres the block can show.
 * out_entries: 32 * ceil(blocks / 8) words, or NULL to ask for *n_workgroups (the number of entries) and *n_blocks only. */
int rt_scene_launch_table(const void *scene_blob, size_t blob_bytes, uint32_t w, uint32_t h, const rt_tiles *tiles, int ranked,
                          uint32_t *out_entries, uint32_t *n_workgroups, uint32_t *n_blocks);

/* Upload a scene to `device` (index into the GPUs in use) and keep it resident. */
int rt_scene_upload(int device, const void *scene_blob, size_t blob_bytes, rt_scene_dev **out);
void rt_scene_free(rt_scene_dev *scene);

/* Move the camera of a resident scene (the reference's lookAt(), main.js:92-100: origin and the three axes; the reference
 * recomputes everything on every redraw, main.js:180-201).  Asynchronous: what depends on the camera - one small block of the
 * resident scene and, per frame size in use, the launch table - exists twice (even / odd camera generations); the move copies the
 * new block and rebuilds those tables ON THE GPU on a stream of the library's own, beside the previous camera's frames that are
 * still rendering, and the next render of the scene waits for them by event (no host wait).  A plain loop
 * `rt_scene_set_camera; rt_render_tiles_device; ...` on ONE stream therefore overlaps a frame's table build with its predecessor's
 * trace.  (The one host wait in that loop is bounded: the render that follows a move so closely that the rebuilt table's entry
 * count has not reached the host yet waits for it for at most 0.1 ms + 1 us per 256 blocks - the caller is ahead of the GPU
 * then - and launches for every block instead of every entry if it does not come.)  `hip_stream` is accepted for source compatibility and not used.  Renders of one scene belong on one stream (several
 * work: a move then drains the device first).  Camera moves and renders of ONE scene handle must not be issued concurrently from
 * different threads (renders among themselves may).  A camera that crosses the scene's enclosing sphere (a skybox) is
 * RT_ERR_UNSUPPORTED: upload the scene again. */
int rt_scene_set_camera(rt_scene_dev *scene, const double origin[3], const double axis_x[3], const double axis_y[3], const double axis_z[3],
                        void *hip_stream);

/* Render tiles of the w x h frame into DEVICE memory `d_out_rgba` (at least
 * n_tiles*tile_rows*w*4 bytes) on `hip_stream` (a hipStream_t; NULL = the library's own
 * stream for that device).  Asynchronous unless `stats` is non-NULL (then it waits and
 * times).  This is the per-pixel loop main.js:185-199 for those rows. */
int rt_render_tiles_device(rt_scene_dev *scene, uint32_t w, uint32_t h, const rt_tiles *tiles,
                           void *d_out_rgba, void *hip_stream, uint32_t flags, rt_stats *stats);

/* The same for a BATCH of n_frames frames in one launch (grid z = frame): frame f's tiles go to
 * d_out_rgba + f*frame_stride_bytes.  All frames use `scene` (synthetic batches; a real animation uploads one
 * scene per frame and calls rt_render_tiles_device per frame).  Used by the multi-GPU plan, where a step
 * renders this rank's row tiles of N frames and one all-to-all reassembles frame f on rank f. */
int rt_render_batch_device(rt_scene_dev *scene, uint32_t w, uint32_t h, const rt_tiles *tiles, uint32_t n_frames,
                           void *d_out_rgba, uint64_t frame_stride_bytes, void *hip_stream, uint32_t flags, rt_stats *stats);

/* The same batch with one destination PER FRAME: frame f's tiles are written into d_frames[f], a whole w x h RGBA8
 * frame buffer, at their rows of the FRAME (not contiguously as a band).  d_frames is a HOST array of n_frames
 * (<= 16) device pointers; they may point into other GPUs' memory (peer-mapped, e.g. opened with rt_ipc_open): on an
 * xGMI node every rank then stores its tiles of frame f straight into the memory of the rank that owns frame f, and
 * no exchange or de-interleave pass is left - only a barrier.  RGBA8 only (no RT_FLAG_RGB24). */
int rt_render_scatter_device(rt_scene_dev *scene, uint32_t w, uint32_t h, const rt_tiles *tiles, uint32_t n_frames,
                             void *const *d_frames, void *hip_stream, uint32_t flags, rt_stats *stats);

/* Sharing a device allocation between the processes of one node (one process per GPU): rt_ipc_export fills a 64-byte
 * handle for memory obtained from rt_alloc_device (the pointer must be the allocation's base); rt_ipc_open maps it in
 * another process for `device` (with peer access over xGMI when it lives on another GPU) and rt_ipc_close unmaps it. */
#define RT_IPC_HANDLE_BYTES 64u
int rt_ipc_export(int device, const void *d_ptr, void *handle_out);
int rt_ipc_open(int device, const void *handle, void **d_ptr_out);
int rt_ipc_close(int device, void *d_ptr);

/* render(width,height,scene): whole frame into HOST memory (any host pointer; memory from rt_alloc_pinned is what the copy engine
 * and the GPU's own stores reach directly).  One GPU: frames of 8 MiB and more are rendered as 4 row bands whose copy-out overlaps
 * the next band's render; into smaller pinned frames the trace kernel stores directly, over PCIe.  Either way the call takes about
 * max(kernel, frame bytes / PCIe rate).  With more than one GPU in use the frame is sharded by interleaved row tiles and put
 * together on GPU 0 (peer stores, or one RCCL gather) before the copy-out.  Replaces redraw()/spanish() + ImageData
 * (main.js:83,180-201). */
int rt_render(const void *scene_blob, size_t blob_bytes, uint32_t w, uint32_t h,
              uint8_t *out_rgba, uint32_t flags, rt_stats *stats);

/* How the one-GPU rt_render / rt_render_progressive hand the frame over (process-wide; for measurements - the defaults (1, 4) follow
 * profiles/r03_ab_log.md section 4).  direct_stores: 0 never, 1 the kernel stores straight into a pinned caller buffer for frames
 * below 8 MiB, 2 for every pinned caller buffer.  copy_bands (1..64): the bands of the copy-out plan for frames of 8 MiB and more. */
int rt_render_options(int direct_stores, uint32_t copy_bands);

/* The same, delivered progressively: the frame is rendered as n_bands (1..64) row bands and on_band(user, first_row,
 * n_rows) is called - on the calling thread, in row order - as soon as a band's rows are in out_rgba, while later bands
 * are still rendering or crossing PCIe.  This is the reference's row-by-row display (spanish(y) per macrotask,
 * main.js:183-201) with bands for rows.  With several GPUs in use the frame arrives whole (one call).  on_band runs
 * inside the library's render lock: it must not call rt_render / rt_render_progressive itself. */
typedef void (*rt_band_callback)(void *user, uint32_t first_row, uint32_t n_rows);
int rt_render_progressive(const void *scene_blob, size_t blob_bytes, uint32_t w, uint32_t h, uint8_t *out_rgba,
                          uint32_t n_bands, rt_band_callback on_band, void *user, uint32_t flags, rt_stats *stats);

/* The reference's end-of-frame report (main.js:204-205: `'build #' + build + ' (' + elapsed + 'ms)'`, drawn over the canvas with
 * fillText): the same string for a finished render, with elapsed = stats->total_ms rounded to whole milliseconds as Date.now()
 * differences are.  Writes at most cap bytes including the terminator; returns the length the full string needs (snprintf rule),
 * or a negative rt_status. */
int rt_elapsed_report(const rt_stats *stats, char *out, size_t cap);

/* Pinned host framebuffers (the ImageData buffer of main.js:83 becomes one of these). */
void *rt_alloc_pinned(size_t bytes);
void rt_free_pinned(void *p);

/* Device scratch helpers for hosts without their own allocator (the Python/torch host
 * passes torch storage instead and never calls these). */
void *rt_alloc_device(int device, size_t bytes);
void rt_free_device(int device, void *p);
int rt_copy_to_host(int device, void *dst_host, const void *src_device, size_t bytes);
int rt_memset_device(int device, void *dst_device, int byte_value, size_t bytes);   /* synchronous */

/* De-interleave a gathered frame: src holds, for rank g in [0,n_ranks), that rank's tiles
 * (g, g+n_ranks, ...) contiguously with `rank_stride_bytes` between ranks; dst receives the
 * frame in row order.  One HBM->HBM pass on `hip_stream`. */
int rt_deinterleave_device(int device, const void *d_src, void *d_dst, uint32_t w, uint32_t h,
                           uint32_t tile_rows, uint32_t n_ranks, uint64_t rank_stride_bytes,
                           void *hip_stream);

/* The same for bands rendered with RT_FLAG_RGB24: src rows are w*3 bytes, dst is the RGBA8 frame
 * (ImageData.data layout) with the alpha byte set to 255.  Needs w % 4 == 0. */
int rt_deinterleave_rgb24_device(int device, const void *d_src, void *d_dst, uint32_t w, uint32_t h,
                                 uint32_t tile_rows, uint32_t n_ranks, uint64_t rank_stride_bytes,
                                 void *hip_stream);

/* Compact bands (RT_FLAG_COMPACT).  rt_compact_count: for `tiles` of the w x h frame of a resident scene and its current camera, the
 * number of blocks a compact launch stores and the bytes of one block; waits until the launch table behind the answer has been built
 * (one small synchronous read).  The collective then moves n_blocks * block_bytes bytes instead of the band. */
int rt_compact_count(rt_scene_dev *scene, uint32_t w, uint32_t h, const rt_tiles *tiles, void *hip_stream, uint32_t *n_blocks, uint32_t *block_bytes);

/* Put the blocks of a compact band back: `d_compact` holds what a launch with RT_FLAG_COMPACT over `tiles` stored - on this GPU or
 * on another rank that holds the same scene with the same camera -; every pixel of every block goes to its place in the RGBA8 frame
 * `d_frame` (w x h, alpha 255).  The sky blocks are not part of the band: the frame's owner stores them with RT_FLAG_SKY_ONLY. */
int rt_compact_expand_device(rt_scene_dev *scene, uint32_t w, uint32_t h, const rt_tiles *tiles, const void *d_compact, void *d_frame, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
