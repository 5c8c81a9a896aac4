// rt_api.hip — the C ABI of include/rt_hip.h: scene validation/upload, the host-built tables (cull rectangles, shadow
// grids, bounce table, launch table), launches, pinned framebuffers, and the single-process multi-GPU frame (interleaved
// row tiles stored straight into GPU 0's frame over xGMI; fallback: RGB24 bands + one RCCL gather + de-interleave).
//
// Host-side counterpart of the reference's driver code: main() sets up what a frame needs
// (main.js:77-105), redraw()/spanish() walks the rows (:180-201).  Here a frame is one kernel
// launch per GPU.  There is no CPU rendering path in this library: without a GPU every render
// entry point fails with RT_ERR_DEVICE.

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>
#include <mutex>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_tables.h"
#include "rt_tables_gpu.h"

extern "C" int rt_launch_trace_fast(const rt_launch *, int, int, int, unsigned, hipStream_t);
extern "C" int rt_launch_trace_strict(const rt_launch *, int, int, int, unsigned, hipStream_t);
extern "C" int rt_launch_retrace(const rt_launch *, int, int, unsigned, hipStream_t);
extern "C" int rt_scratch_trace_fast(int, int, int, int, int, size_t *);
extern "C" int rt_scratch_trace_strict(int, int, int, int, int, size_t *);
extern "C" int rt_scratch_retrace(int, int, size_t *);

using namespace rt_tables;   // the host-built tables (pure host logic, rt_tables.cpp)

// ------------------------------------------------------------------------------------ state
namespace {

thread_local char g_err[512] = "";

// A/B and test switches exist only in the TEST build of this library (csrc/Makefile: librt_hip_test.so, -DRT_TESTING,
// selected by the tests with RT_HIP_LIB).  The product library reads no environment variable on the render path.
#ifdef RT_TESTING
#define RT_TEST_ENV(name) getenv(name)
#else
#define RT_TEST_ENV(name) ((const char *)nullptr)
#endif

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));     \
  } while (0)

struct device_state {
  int hip_id = -1;
  hipStream_t stream = nullptr;          // created on first use
  hipStream_t copy_stream = nullptr;     // rt_render: PCIe copy-out overlapped with rendering
  unsigned long long *d_counters = nullptr;
  void *d_frame = nullptr;               // rt_render scratch: this device's tiles (or the whole frame)
  size_t frame_bytes = 0;
  void *d_gather = nullptr;              // device 0 only: gather target (fallback plan of rt_render on several GPUs)
  size_t gather_bytes = 0;
  int peer_to_root = 0;                  // rt_render on several GPUs: 1 = this device may store into device 0's memory, -1 = it may not, 0 = not asked yet
  // rt_render: the scene of the previous call stays resident; a call with the same blob (byte for byte) reuses it
  // (upload + table builds cost 0.1 ms for 8 spheres and 1.8 ms for 64, against a 0.7 ms frame)
  struct rt_scene_dev *cached_scene = nullptr;
  std::vector<uint8_t> cached_blob;
  // scratch_guard: wave slots of the device (CUs x waves per CU) and, per stream, the largest per-lane scratch figure whose reservation
  // has been held against the free device memory
  size_t wave_slots = 0;
  struct scratch_seen { hipStream_t stream; size_t per_lane; };
  std::vector<scratch_seen> scratch_checked;
};

struct lib_state {
  bool inited = false;
  std::vector<device_state> dev;
  std::mutex mu;
  std::mutex dev_mu;                     // lazy per-device stream creation (ensure_device may run inside rt_render, which holds `mu`)
  // RCCL, resolved lazily with dlopen so that single-GPU users never load it
  void *rccl = nullptr;
  void *comms[16] = {nullptr};
  bool comms_ready = false;
  // RT_EMULATE_DEVICES=N (test aid for 1-GPU boxes): rt_init reports N devices that all map to HIP device 0, and
  // rt_render's gather becomes device-to-device copies instead of ncclGather (RCCL refuses two ranks on one GPU).
  // Everything else of the multi-GPU frame - tile plan, per-device scenes and streams, RGB24 bands, de-interleave -
  // runs as on a real node.
  bool emulated = false;
  bool all_visible = false;
} G;

int ensure_device(int d) {
  if (!G.inited) return fail(RT_ERR_STATE, "rt_init has not been called");
  if (d < 0 || d >= (int)G.dev.size()) return fail(RT_ERR_INVALID, "device %d out of range (0..%d)", d, (int)G.dev.size() - 1);
  device_state &s = G.dev[d];
  HIP_TRY(hipSetDevice(s.hip_id));
  std::lock_guard<std::mutex> lk(G.dev_mu);
  if (!s.stream) {
    hipStream_t st = nullptr;
    HIP_TRY(hipMalloc(&s.d_counters, 3 * sizeof(unsigned long long)));
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    s.stream = st;                         // published last: a non-NULL stream means the device state is complete
  }
  return RT_OK;
}

// Kernels with a private segment (the general kernel's park stack: 3472 B per lane; the strict kernels' recursion stack) make the
// runtime reserve scratch memory: bytes per lane x 64 lanes x the wave slots the dispatch can occupy (at most CUs x waves per CU =
// 8192 on MI355X: 1.8 GB for the general kernel).  When that reservation cannot be met the HIP runtime does not return an error from
// the launch: the queue's error callback ABORTS the process (profiles/r04_scratch_refusal.log).  So every launch path of a kernel that
// needs scratch holds the figure against the free device memory first - once per (stream, larger figure): later launches reuse the
// queue's scratch - and the library returns RT_ERR_NOMEM with the numbers instead of reaching that abort.
int scratch_guard(device_state &D, hipStream_t stream, size_t per_lane, uint64_t waves_in_grid, const char *what) {
#ifdef RT_TESTING
  bool pretend = false;
  if (const char *q = getenv("RT_TEST_SCRATCH_PER_LANE")) { per_lane = (size_t)strtoull(q, nullptr, 10); pretend = true; }    // as if the kernel asked for this much
#else
  const bool pretend = false;
#endif
  if (per_lane == 0) return RT_OK;
  std::lock_guard<std::mutex> lk(G.dev_mu);
  if (!pretend) for (const device_state::scratch_seen &c : D.scratch_checked) if (c.stream == stream && c.per_lane >= per_lane) return RT_OK;
  if (!D.wave_slots) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, D.hip_id));
    D.wave_slots = (size_t)p.multiProcessorCount * (size_t)(p.maxThreadsPerMultiProcessor / 64);
    if (!D.wave_slots) D.wave_slots = 8192;
  }
  const uint64_t slots = waves_in_grid < D.wave_slots ? waves_in_grid : D.wave_slots;
  const uint64_t reservation = (uint64_t)per_lane * 64u * slots;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const uint64_t margin = (uint64_t)64 << 20;
  if (reservation + margin > free_b)
    return fail(RT_ERR_NOMEM, "%s needs %zu bytes of scratch per lane: the runtime reserves that for %llu wave slots = %llu bytes, and %zu of %zu bytes of device memory are free",
                what, per_lane, (unsigned long long)slots, (unsigned long long)reservation, free_b, total_b);
  if (!pretend) {
    bool found = false;
    for (device_state::scratch_seen &c : D.scratch_checked) if (c.stream == stream) { c.per_lane = per_lane; found = true; }
    if (!found) { if (D.scratch_checked.size() >= 64u) D.scratch_checked.clear(); D.scratch_checked.push_back(device_state::scratch_seen{stream, per_lane}); }
  }
  return RT_OK;
}

// per-lane scratch of a kernel instantiation, from its code object (asked once per instantiation)
int kernel_scratch(bool strict, bool retrace, int refract, int count, int ss2, int grid_variant, size_t *out, bool one_wave = false) {
  static std::mutex mu;
  static size_t cache[3][2][2][2][4];
  static bool have[3][2][2][2][4];
  const int k = retrace ? 2 : (strict ? 1 : 0), c = retrace ? 0 : (count ? 1 : 0), g = (retrace || strict) ? 0 : ((grid_variant ? 1 : 0) + (one_wave ? 2 : 0));
  std::lock_guard<std::mutex> lk(mu);
  if (!have[k][refract ? 1 : 0][c][ss2 ? 1 : 0][g]) {
    size_t b = 0;
    const int e = retrace ? rt_scratch_retrace(refract, ss2, &b) : (strict ? rt_scratch_trace_strict(refract, count, ss2, 0, 0, &b) : rt_scratch_trace_fast(refract, count, ss2, grid_variant, one_wave ? 1 : 0, &b));
    if (e != 0) return fail(RT_ERR_DEVICE, "hipFuncGetAttributes: %s", hipGetErrorString((hipError_t)e));
    cache[k][refract ? 1 : 0][c][ss2 ? 1 : 0][g] = b; have[k][refract ? 1 : 0][c][ss2 ? 1 : 0][g] = true;
  }
  *out = cache[k][refract ? 1 : 0][c][ss2 ? 1 : 0][g];
  return RT_OK;
}

}  // namespace

struct rt_scene_dev {
  int device;
  // Everything the scene keeps in HBM is ONE allocation (`arena`), filled by one copy at upload; the pointers below point into it.
  // Its last part is the CAMERA BLOCK - what depends on the camera: per ordering the camera-anchored geometry and the cull
  // rectangles, and (few spheres) the LDS images, whose tails are the cull rectangles - rewritten with one small asynchronous copy
  // when the camera moves (rt_scene_set_camera).
  uint8_t *arena = nullptr;
  size_t arena_bytes = 0;
  void *d_blob;                  // the uploaded scene blob
  rt_texture_desc *d_texdesc;    // RT_MAX_TEXTURES descriptors (zero padded)
  rt_geom *d_geom;               // camera-independent geometry tables, per ordering [plain N | anchored at light k: NL x N]
  rt_sphere *d_objects_b;        // object records with the enclosing sphere moved last (ordering B); NULL if none
  uint64_t *d_shadow_grid;       // light grids for the product kernel's loop order, or NULL (few spheres)
  uint64_t *d_bounce_table;      // bounce table for the same order, or NULL (few spheres, or depth < 2)
  uint8_t *d_lds_image;          // many spheres: per ordering [materials (rt_mtl) | 16 texture descriptors], the LDS image (few spheres: it holds the cull rectangles and lives in the camera block)
  // The camera block: per ordering [anchored at the camera N | cull rectangles N], then (few spheres) the LDS images.  TWO of them:
  // camera generation g lives in block g & 1, so that the block of the NEXT camera can be written - on the scene's own side stream,
  // by rt_scene_set_camera - while launches with the current one are still running.
  uint8_t *d_cam_buf[2];
  size_t cam_lds_offset;         // of the LDS images inside a camera block
  size_t cam_bytes;              // (with the padding the many-sphere staging may read over)
  size_t cam_bytes_used;         // what a camera move has to copy
  bool has_b;                    // two orderings (an enclosing sphere)
  bool cull_in_lds;
  size_t lds_image_bytes;        // of one ordering
  uint64_t cam_gen = 1;          // bumped when the camera moves: launch tables and mark counts of an older camera are stale
  uint32_t renders_with_camera = 0;   // product launches since the camera last moved (or the upload)
  // ... and per (frame size, tile set, sky part): many-sphere scenes get their shadow masks with the SECOND frame of a kind from a camera
  struct camera_use { uint32_t w, h, ss, tile_rows, tile_first, tile_stride, n_tiles, part; uint64_t cam_gen; uint32_t uses; };
  std::vector<camera_use> camera_uses;
  hipStream_t last_stream = nullptr;     // the stream of the scene's last launch; several: launches of this scene are in flight on more than one
  bool any_launch = false, several_streams = false, launched_since_move = false;
  // The camera pipeline (rt_scene_set_camera).  `side`: a stream of the scene's own, on which a move's camera block is copied and the
  // launch tables of the frame sizes in use are rebuilt - beside the previous camera's launches, which run on the caller's stream.
  //   old_done[x]   recorded on the caller's stream at the move to generation g (x = (g - 1) & 1): every launch with generations < g
  //                 precedes it.  The move to g + 1 writes block / tables (g + 1) & 1 = x only behind it.
  //   prep_done[b]  recorded on `side` behind the copy and the builds of generation g (b = g & 1): the first launch of generation g
  //                 on a stream waits for it (prep_waited: which streams already do).
  hipStream_t side = nullptr;
  hipEvent_t old_done[2] = {nullptr, nullptr}, prep_done[2] = {nullptr, nullptr};
  bool old_done_valid[2] = {false, false}, prep_valid[2] = {false, false};
  struct waited_on { hipStream_t stream; uint64_t gen; };
  std::vector<waited_on> prep_waited;
  // pinned staging for the small copies that follow a camera move (the camera block; a launch table's parameters): a ring of slots,
  // each guarded by an event recorded behind the copy that read it
  struct stage_slot { uint8_t *h = nullptr; hipEvent_t done = nullptr; bool used = false; };
  stage_slot stages[16];                 // (16: the host may run eight frames ahead of the GPU in an animation; one pinned allocation behind them)
  uint8_t *stage_pool = nullptr;
  size_t stage_bytes = 0;
  uint32_t stage_next = 0;
  std::vector<uint8_t> host_blob;        // the scene as uploaded (patched: 1/r per sphere), for rebuilding the camera block
  std::vector<rt_sphere> host_objects_b; // ordering B of its sphere records
  rt_scene_header hd;            // host copy
  bool refract;                  // any albedo[4] > 0  -> general (binary-tree) kernel variant
  unsigned lds_bytes;
  double lights[RT_MAX_LIGHTS][3];   // host copy: lights travel in the kernarg segment
  uint32_t enclosing;            // sphere that strictly contains everything else (a skybox), or ~0u
  bool enclosing_flat;           // ... and it has no lighting, no children and a sampler that ignores the hit point (colour / stars)
  bool sky_const;                // ... a plain colour: the pixel of a ray that meets nothing else is the constant sky_rgb
  double sky_rgb[3];
  // cost-ordered dispatch (dispatch_order below): per sphere its screen rectangle (X/D, Y/D bounds, scene order) and a weight,
  // and the order tables built so far, one per (frame size, tile set), kept on the device
  std::vector<rt_sphere> host_objects;   // the scene's sphere records (scene order), for the launch table's sky marking
  std::vector<rt_geom> host_cull;
  std::vector<uint32_t> tile_weight;
  // One launch table per (frame size, tile set, flags), built on the GPU (rt_tables_gpu.hip) on the stream of the launch that needs
  // it first and again when the camera has moved since (cam_gen).  `Tb` = its device memory; `d_blockb` = ONE allocation
  // behind all of T's arrays; `n_blocks` workgroups are launched until the host has seen the number of entries the build published
  // (`known`: generation << 32 | entries + 1, a pinned host word), from then on exactly that many.
  struct order_entry {
    uint32_t w, h, ss, tile_rows, tile_first, tile_stride, n_tiles; int ranked; bool sky, masks, cands; uint32_t part;     // ranked: 0 grid order, 1 ranked when large enough, 2 always (a compact band's launch)
    uint64_t cam_gen; uint32_t n_blocks; volatile unsigned long long *known; hipStream_t built_on; hipEvent_t built;
    bool shared;                   // launched with on a stream other than the one it was built on
    // two tables, like the camera blocks: generation g's is Tb[g & 1] (the next camera's is built while this one's is still read)
    rt_table_dev Tb[2]; uint8_t *d_blockb[2]; size_t hist_wordsb[2];
    uint32_t cost_bins;            // of the current build (rt_retrace of a compact launch)
    uint64_t used_gen;             // the last camera generation a launch used it with: a move rebuilds the tables in use ahead of the next render
  };
  std::vector<order_entry> orders;
  uint32_t order_evict = 0;
  rt_texture_desc descs[RT_MAX_TEXTURES];
  // Marked samples (rt_device.h, rt_kernel.hip: rt_retrace).  One state per (launch table, stream): the device list the product
  // launch appends to and rt_retrace reads, which of its two counters the next launch uses, and a pinned host word in which
  // rt_retrace publishes how many samples a frame of this scene, camera, size and tile set marks - the same every time, so once it
  // says "none" (and the sample grid has no odd centre) the second launch is skipped.  Launches that share a state share a
  // stream, i.e. they are ordered; mark_mu makes a launch pair one step for the threads of this process.
  struct mark_state { uint32_t order_index; hipStream_t stream; uint32_t *d_marks; volatile unsigned long long *h_known; uint32_t slot; };   // *h_known: camera generation << 32 | marks + 1
  std::vector<mark_state> mark_states;
  unsigned long long *h_known_pool = nullptr;      // 2 x RT_KNOWN_WORDS pinned words: the mark states', then the launch tables'
  std::mutex launch_mu;          // a product launch - its table (found or built), the trace launch, rt_retrace - is one step for the threads of this process
  bool needs_strict;             // the scene sits on an exact coincidence (below): every launch uses the strict kernel
  bool needs_strict_scene;       // ... whatever the camera (a light on a surface, a sphere without a radius, exotic checker frequencies)
  bool unit_weights;             // every albedo and colour in [0, 1] (RT_MARK_WEIGHT)
  double flag_tol;               // RT_FLAG_T1 x the largest sampler frequency of the scene (texture width / height, checker frequencies): rt_device.h
};

// ------------------------------------------------------------------------------------ lifetime
extern "C" uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
extern "C" const char *rt_build_id(void) { return RT_REFERENCE_BUILD "." RT_LIBRARY_REVISION; }

// main.js:204-205: 'build #' + build + ' (' + elapsed + 'ms)', elapsed a whole number of milliseconds (Date.now() difference)
extern "C" int rt_elapsed_report(const rt_stats *stats, char *out, size_t cap) {
  if (!stats || (!out && cap)) return fail(RT_ERR_INVALID, "rt_elapsed_report: NULL argument");
  const double ms = (stats->total_ms >= 0.0 && stats->total_ms < 1e15) ? stats->total_ms : 0.0;
  return snprintf(out, cap, "build #%s (%lldms)", rt_build_id(), (long long)llround(ms));
}
extern "C" const char *rt_last_error(void) { return g_err; }

extern "C" int rt_init(int max_devices) {
  std::lock_guard<std::mutex> lk(G.mu);
  if (G.inited) {
    // a second rt_init must not silently change how rt_render shards a frame: asking for a different number of GPUs than
    // the library already uses is an error (0 = "whatever is in use" is fine); rt_shutdown first to change it
    if (max_devices > 0 && max_devices != (int)G.dev.size() && !(max_devices > (int)G.dev.size() && G.all_visible))
      return fail(RT_ERR_STATE, "rt_init(%d): the library is already initialised with %d device(s); call rt_shutdown first", max_devices, (int)G.dev.size());
    return RT_OK;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(RT_ERR_DEVICE, "no HIP device visible (%s); this library has no CPU path", e == hipSuccess ? "count 0" : hipGetErrorString(e));
  const char *emu = RT_TEST_ENV("RT_EMULATE_DEVICES");
  G.emulated = emu && atoi(emu) > 1;
  if (G.emulated) n = atoi(emu);
  G.all_visible = !(max_devices > 0 && n > max_devices);   // every visible GPU is in use: a later request for "up to more" changes nothing
  if (max_devices > 0 && n > max_devices) n = max_devices;
  if (n > 16) n = 16;
  G.dev.resize(n);
  for (int i = 0; i < n; i++) G.dev[i].hip_id = G.emulated ? 0 : i;
  G.inited = true;
  return RT_OK;
}

extern "C" int rt_device_count(void) {
  if (!G.inited) return fail(RT_ERR_STATE, "rt_init has not been called");
  return (int)G.dev.size();
}

// ------------------------------------------------------------------------------------ validation (host logic only)
extern "C" int rt_scene_validate(const void *blob, size_t bytes) {
  if (!blob || bytes < sizeof(rt_scene_header)) return fail(RT_ERR_INVALID, "scene blob shorter than its header");
  if (((uintptr_t)blob & 7u) != 0) return fail(RT_ERR_INVALID, "scene blob must be 8-byte aligned");
  const rt_scene_header *hd = (const rt_scene_header *)blob;
  if (hd->magic != RT_SCENE_MAGIC) return fail(RT_ERR_INVALID, "bad scene magic 0x%08x", hd->magic);
  if (hd->abi_version != RT_ABI_VERSION) return fail(RT_ERR_INVALID, "scene ABI version %u, library speaks %u", hd->abi_version, RT_ABI_VERSION);
  if (hd->total_bytes != bytes) return fail(RT_ERR_INVALID, "total_bytes %llu != blob size %zu", (unsigned long long)hd->total_bytes, bytes);
  if (hd->n_objects < 1 || hd->n_objects > RT_MAX_OBJECTS) return fail(RT_ERR_INVALID, "n_objects %u not in 1..%u", hd->n_objects, RT_MAX_OBJECTS);
  if (hd->n_lights > RT_MAX_LIGHTS) return fail(RT_ERR_INVALID, "n_lights %u > %u", hd->n_lights, RT_MAX_LIGHTS);
  if (hd->n_textures > RT_MAX_TEXTURES) return fail(RT_ERR_INVALID, "n_textures %u > %u", hd->n_textures, RT_MAX_TEXTURES);
  if (hd->segs > RT_MAX_SEGS) return fail(RT_ERR_INVALID, "segs %u > %u", hd->segs, RT_MAX_SEGS);
  if (hd->supersample < 1 || hd->supersample > 4) return fail(RT_ERR_INVALID, "supersample must be 1, 2, 3 or 4");
  if (!(hd->fov_deg > 0.0 && hd->fov_deg < 180.0)) return fail(RT_ERR_INVALID, "fov_deg must be in (0,180)");
  auto in_range = [&](uint64_t off, uint64_t len) { return (off & 7u) == 0 && off >= sizeof(rt_scene_header) && off <= bytes && len <= bytes - off; };
  if (!in_range(hd->objects_offset, (uint64_t)hd->n_objects * sizeof(rt_sphere))) return fail(RT_ERR_INVALID, "object table out of bounds");
  if (!in_range(hd->lights_offset, (uint64_t)hd->n_lights * 24u)) return fail(RT_ERR_INVALID, "light table out of bounds");
  if (!in_range(hd->textures_offset, (uint64_t)hd->n_textures * sizeof(rt_texture_desc))) return fail(RT_ERR_INVALID, "texture table out of bounds");
  const uint8_t *base = (const uint8_t *)blob;
  const rt_texture_desc *td = (const rt_texture_desc *)(base + hd->textures_offset);
  for (uint32_t t = 0; t < hd->n_textures; t++) {
    if (td[t].width == 0 || td[t].height == 0 || td[t].width > 16384 || td[t].height > 16384) return fail(RT_ERR_INVALID, "texture %u: bad size", t);
    if ((td[t].texels_offset & 3u) != 0 || td[t].texels_offset > bytes || (uint64_t)td[t].width * td[t].height * 4u > bytes - td[t].texels_offset)
      return fail(RT_ERR_INVALID, "texture %u: texels out of bounds", t);
  }
  const rt_sphere *ob = (const rt_sphere *)(base + hd->objects_offset);
  for (uint32_t i = 0; i < hd->n_objects; i++) {
    const int k = ob[i].sampler_kind;
    if (k != RT_SAMPLER_COLOR && k != RT_SAMPLER_TEXTURE && k != RT_SAMPLER_CHECKER && k != RT_SAMPLER_STARS)
      return fail(RT_ERR_UNSUPPORTED, "object %u: sampler kind %d is not supported (0 colour, 1 texture, 2 checker, 3 hashed stars)", i, k);
    if (k == RT_SAMPLER_TEXTURE && (ob[i].texture < 0 || (uint32_t)ob[i].texture >= hd->n_textures))
      return fail(RT_ERR_INVALID, "object %u: texture index %d out of range", i, ob[i].texture);
  }
  return RT_OK;
}

// Host-logic probe for tests: the bounce table's answer for one ray, with the kernel's own direction -> cell mapping
// (an exact division where the kernel uses a 2^-24 reciprocal: the cells overlap by 1e-6 rad for that).
extern "C" int rt_scene_bounce_candidates(const void *blob, size_t bytes, uint32_t from, const double dir[3], uint64_t *out_words) {
  int rc = rt_scene_validate(blob, bytes);
  if (rc) return rc;
  const rt_scene_header *hd = (const rt_scene_header *)blob;
  if (!dir || !out_words || from >= hd->n_objects) return fail(RT_ERR_INVALID, "bad bounce probe arguments");
  const rt_sphere *ob = (const rt_sphere *)((const uint8_t *)blob + hd->objects_offset);
  const uint32_t n = hd->n_objects, words = (n + 63u) / 64u;
  const std::vector<uint64_t> tab = build_bounce_table(ob, n, n);
  const double ax = fabs(dir[0]), ay = fabs(dir[1]), az = fabs(dir[2]);
  const bool bx = (ax >= ay) && (ax >= az), by = !bx && (ay >= az);
  const double dm = bx ? dir[0] : (by ? dir[1] : dir[2]);
  const double du = bx ? dir[1] : dir[0], dv = (bx || by) ? dir[2] : dir[1];
  const double sc = (0.5 * RT_BGRID) / fabs(dm);
  const double fu = fmin(fmax(du * sc + 0.5 * RT_BGRID, 0.0), (double)(RT_BGRID - 1u)), fv = fmin(fmax(dv * sc + 0.5 * RT_BGRID, 0.0), (double)(RT_BGRID - 1u));
  const uint32_t face = (bx ? 0u : (by ? 2u : 4u)) + ((dm < 0.0) ? 1u : 0u);
  const uint32_t cell = face * (RT_BGRID * RT_BGRID) + (uint32_t)fv * RT_BGRID + (uint32_t)fu;
  memcpy(out_words, tab.data() + ((size_t)from * RT_BCELLS + cell) * words, words * sizeof(uint64_t));
  return RT_OK;
}

// Host-logic probe for tests: the cull rectangle of every sphere, scene order, 4 doubles each.
extern "C" int rt_scene_cull_rects(const void *blob, size_t bytes, double *out) {
  int rc = rt_scene_validate(blob, bytes);
  if (rc) return rc;
  if (!out) return fail(RT_ERR_INVALID, "out is NULL");
  const rt_scene_header *hd = (const rt_scene_header *)blob;
  const rt_sphere *ob = (const rt_sphere *)((const uint8_t *)blob + hd->objects_offset);
  for (uint32_t i = 0; i < hd->n_objects; i++) {
    const rt_geom r = cull_rect(hd, ob[i]);
    out[4 * i] = r.ox; out[4 * i + 1] = r.oy; out[4 * i + 2] = r.oz; out[4 * i + 3] = r.r2;
  }
  return RT_OK;
}

// ------------------------------------------------------------------------------------ upload
namespace {
constexpr size_t RT_KNOWN_WORDS = 256;

void free_order_entry(rt_scene_dev::order_entry &e) {
  for (int b = 0; b < 2; b++) { if (e.d_blockb[b]) (void)hipFree(e.d_blockb[b]); e.d_blockb[b] = nullptr; }
  if (e.built) (void)hipEventDestroy(e.built);
  e.built = nullptr;
}
inline uint8_t *cam_block(const rt_scene_dev *s) { return s->d_cam_buf[s->cam_gen & 1u]; }
inline uint8_t *lds_image_of(const rt_scene_dev *s) { return s->cull_in_lds ? cam_block(s) + s->cam_lds_offset : s->d_lds_image; }

// [materials (rt_mtl) | 16 texture descriptors | cull rectangles (few spheres)] of ordering `ord`: the workgroup's LDS image
void fill_lds_image(const rt_scene_dev *s, uint8_t *dst, int ord) {
  const uint32_t NO = s->hd.n_objects;
  const rt_sphere *src = ord ? s->host_objects_b.data() : (const rt_sphere *)(s->host_blob.data() + s->hd.objects_offset);
  rt_mtl *mt = (rt_mtl *)dst;
  for (uint32_t i = 0; i < NO; i++) {
    const rt_sphere &o = src[i];
    rt_mtl &m = mt[i];
    memset(&m, 0, sizeof m);
    memcpy(m.origin, o.origin, sizeof m.origin);
    m.inv_r = o.reserved;                           // 1/r, patched at upload
    memcpy(m.albedo, o.albedo, sizeof m.albedo);
    m.specular_exponent = o.specular_exponent; m.refract_index = o.refract_index;
    m.sampler_kind = o.sampler_kind; m.texture = o.texture;
    if (o.sampler_kind == RT_SAMPLER_CHECKER) memcpy(m.c, o.checker_color, 6 * sizeof(double));
    else memcpy(m.c, o.color, 3 * sizeof(double));
    m.c[6] = o.checker_freq[0]; m.c[7] = o.checker_freq[1];
  }
  memcpy(dst + (size_t)NO * sizeof(rt_mtl), s->descs, sizeof s->descs);
  if (s->cull_in_lds) {
    rt_geom *cr = (rt_geom *)(dst + (size_t)NO * sizeof(rt_mtl) + sizeof s->descs);
    for (uint32_t i = 0; i < NO; i++) cr[i] = cull_rect(&s->hd, src[i]);
  }
}

// The camera block (rt_scene_dev): per ordering [anchored at the camera {o - cam, |o - cam|^2 - r2} N | primary-ray cull
// rectangles N], then the LDS images when they hold the rectangles.  `dst`: cam_bytes of host memory.
void fill_camera_block(const rt_scene_dev *s, uint8_t *dst) {
  const uint32_t NO = s->hd.n_objects;
  const int n_ord = s->has_b ? 2 : 1;
  for (int ord = 0; ord < n_ord; ord++) {
    const rt_sphere *src = ord ? s->host_objects_b.data() : (const rt_sphere *)(s->host_blob.data() + s->hd.objects_offset);
    rt_geom *g = (rt_geom *)dst + (size_t)ord * 2u * NO;
    for (uint32_t i = 0; i < NO; i++) {
      const double lx = src[i].origin[0] - s->hd.cam_origin[0], ly = src[i].origin[1] - s->hd.cam_origin[1], lz = src[i].origin[2] - s->hd.cam_origin[2];
      g[i] = rt_geom{lx, ly, lz, (lx * lx + ly * ly + lz * lz) - src[i].r2};
      g[NO + i] = cull_rect(&s->hd, src[i]);
    }
  }
  if (s->cull_in_lds) {
    uint8_t *img = dst + s->cam_lds_offset;
    for (int ord = 0; ord < n_ord; ord++) fill_lds_image(s, img + ord * s->lds_image_bytes, ord);
  }
}

// what of a resident scene depends on the camera and is decided on the host: is it a strict-kernel scene, which sphere encloses
// everything, the background constant, the cull rectangles and cost weights of the launch tables
void camera_decisions(rt_scene_dev *s) {
  const rt_scene_header *hd = &s->hd;
  const rt_sphere *ob = (const rt_sphere *)(s->host_blob.data() + hd->objects_offset);
  s->needs_strict = s->needs_strict_scene;
  for (int c = 0; c < 3; c++) if (hd->cam_axis_x[c] + hd->cam_axis_y[c] + hd->cam_axis_z[c] == 0.0) s->needs_strict = true;
  // cost-ordered dispatch: what a tile that shows sphere j is expected to cost, in rough units of one shaded hit - a guess
  // that only has to RANK tiles: lit hits 2, one more per bounce a reflective or refractive hit can spawn, and the binary tree
  // of a sphere that does both (main.js:268-278) its node count; pure-ambient spheres (the reference's skybox) nothing
  scene_tile_weights(hd, ob, &s->host_cull, &s->tile_weight);
}

// a staging slot of `bytes` (<= stage_bytes), free to be written: its previous copy has been read
uint8_t *acquire_stage(rt_scene_dev *s, rt_scene_dev::stage_slot **slot) {
  rt_scene_dev::stage_slot &g = s->stages[s->stage_next++ & 15u];
  if (g.used) (void)hipEventSynchronize(g.done);
  g.used = true;
  *slot = &g;
  return g.h;
}
}  // namespace

extern "C" int rt_scene_upload(int device, const void *blob, size_t bytes, rt_scene_dev **out) {
  if (!out) return fail(RT_ERR_INVALID, "out handle is NULL");
  *out = nullptr;
  int rc = rt_scene_validate(blob, bytes);
  if (rc) return rc;
  if ((rc = ensure_device(device))) return rc;
  const rt_scene_header *hd = (const rt_scene_header *)blob;
  rt_scene_dev *s = new rt_scene_dev();
  s->device = device; s->hd = *hd; s->d_blob = nullptr; s->d_texdesc = nullptr; s->d_geom = nullptr; s->d_objects_b = nullptr; s->d_lds_image = nullptr; s->d_shadow_grid = nullptr; s->d_bounce_table = nullptr;
  s->d_cam_buf[0] = s->d_cam_buf[1] = nullptr;
  const uint8_t *base = (const uint8_t *)blob;
  const rt_sphere *ob = (const rt_sphere *)(base + hd->objects_offset);
  s->refract = false;
  for (uint32_t i = 0; i < hd->n_objects; i++) if (ob[i].albedo[4] > 0.0) s->refract = true;
  // Scenes whose picture hinges on exact coincidences are rendered by the strict kernel throughout (the product kernel's
  // short cuts - 1/r from the host, anchored discriminants, shadow rays walked from the light - assume a generic scene):
  //   * a light exactly ON a sphere's surface (the reference's own `t < light_len`, main.js:297, then compares two numbers
  //     that are equal up to rounding: a coin flip that only the reference's own arithmetic reproduces);
  //   * a sphere with r2 <= 0 or not finite (no 1/r);
  //   * a sphere-checker whose frequencies are negative, NaN or >= 2^31 (below);
  //   * a camera whose axis sums (main.js:187-191, quirk q1) have an exactly zero component: EVERY primary ray then lies in
  //     a coordinate plane through the camera (camera_decisions; rt_retrace traces the centre row / column of an odd sample grid
  //     for the same reason).
  s->needs_strict_scene = false;
  for (uint32_t i = 0; i < hd->n_objects; i++) {
    if (!(ob[i].r2 > 0.0) || !std::isfinite(ob[i].r2)) s->needs_strict_scene = true;
    const double *lt = (const double *)(base + hd->lights_offset);
    for (uint32_t k = 0; k < hd->n_lights; k++) {
      const double x = lt[3 * k] - ob[i].origin[0], y = lt[3 * k + 1] - ob[i].origin[1], z = lt[3 * k + 2] - ob[i].origin[2];
      if (fabs((x * x + y * y + z * z) - ob[i].r2) <= 1e-9 * fmax(ob[i].r2, 1.0)) s->needs_strict_scene = true;
    }
  }
  // the boundary test of the product kernel's samplers (rt_device.h: RT_FLAG_T1): a coordinate is u * frequency
  {
    double fmaxq = 1.0;
    const rt_texture_desc *td = (const rt_texture_desc *)(base + hd->textures_offset);
    for (uint32_t i = 0; i < hd->n_objects; i++) {
      if (ob[i].sampler_kind != RT_SAMPLER_TEXTURE && ob[i].sampler_kind != RT_SAMPLER_CHECKER) continue;
      if (ob[i].sampler_kind == RT_SAMPLER_TEXTURE) fmaxq = fmax(fmaxq, (double)(td[ob[i].texture].width > td[ob[i].texture].height ? td[ob[i].texture].width : td[ob[i].texture].height));
      if (ob[i].sampler_kind == RT_SAMPLER_CHECKER) {
        const double f0 = ob[i].checker_freq[0], f1 = ob[i].checker_freq[1];
        if (fabs(f0) > fmaxq) fmaxq = fabs(f0);                       // (NaN frequencies: every sample of such a sphere is NaN, and marked)
        if (fabs(f1) > fmaxq) fmaxq = fabs(f1);
        // the product kernel takes ToInt32(u * f) & 1 (main.js:129-130) from a fixed-point sum that holds it for products in [0, 2^31):
        // other frequencies (negative, huge, NaN) make the scene a strict-kernel scene
        if (!(f0 >= 0.0 && f0 < 2147483648.0 && f1 >= 0.0 && f1 < 2147483648.0)) s->needs_strict_scene = true;
      }
    }
    // The hot path's prefilter passes coordinates within 2^-20 of an integer to the precise test against flag_tol = RT_FLAG_T1 x this
    // frequency, scaled by the hit's magnification bound (rt_kernel.hip): the band has to leave that scaling room.  Up to 2^17 per unit u
    // it is 36 x the flat tolerance; the adversarial soak's second pixel (a checker at 1e6 per unit, two bounces: an error of 1.75e-6
    // squares, beyond the band) is what set the limit.  Scenes with finer samplers take the strict kernel.
    if (fmaxq > 131072.0) s->needs_strict_scene = true;
    s->flag_tol = RT_FLAG_T1 * fmaxq;
    // (boundary marks) every albedo and colour within [0, 1]: then a node's colour moves the pixel by at most its accumulated weight
    s->unit_weights = true;
    for (uint32_t i = 0; i < hd->n_objects; i++) {
      for (int c = 0; c < 5; c++) if (!(ob[i].albedo[c] >= 0.0 && ob[i].albedo[c] <= 1.0)) s->unit_weights = false;
      for (int c = 0; c < 3; c++) if (!(ob[i].color[c] >= 0.0 && ob[i].color[c] <= 1.0)) s->unit_weights = false;
      if (ob[i].sampler_kind == RT_SAMPLER_CHECKER) for (int c = 0; c < 6; c++) if (!(ob[i].checker_color[c / 3][c % 3] >= 0.0 && ob[i].checker_color[c / 3][c % 3] <= 1.0)) s->unit_weights = false;
    }
  }
  memset(s->lights, 0, sizeof s->lights);
  if (hd->n_lights) memcpy(s->lights, base + hd->lights_offset, hd->n_lights * 24u);
  rt_texture_desc (&descs)[RT_MAX_TEXTURES] = s->descs;
  memset(descs, 0, sizeof descs);
  if (hd->n_textures) memcpy(descs, base + hd->textures_offset, hd->n_textures * sizeof(rt_texture_desc));
  s->enclosing = enclosing_sphere(hd, ob, s->lights);     // (rt_tables.cpp)
  s->enclosing_flat = false;
  if (s->enclosing != ~0u) {
    const rt_sphere &sk = ob[s->enclosing];
    s->enclosing_flat = !(sk.albedo[1] > 0.0) && !(sk.albedo[2] > 0.0) && !(sk.albedo[3] > 0.0) && !(sk.albedo[4] > 0.0) &&
                        (sk.sampler_kind == RT_SAMPLER_COLOR || sk.sampler_kind == RT_SAMPLER_STARS);
  }
  s->sky_const = s->enclosing_flat && ob[s->enclosing].sampler_kind == RT_SAMPLER_COLOR && hd->segs > 0;
  for (int c = 0; c < 3; c++) {
    s->sky_rgb[c] = 0.0;
    if (s->sky_const) {
      // main.js:322-336 for a hit without light and without children: diffuse = specular = 0, reflect = refract = [0,0,0]
      const volatile double col = ob[s->enclosing].color[c], a0 = ob[s->enclosing].albedo[0], zero = 0.0;
      const volatile double amb = col * a0, d0 = col * zero, s0 = col * zero;
      const volatile double shade = d0 + s0;
      const double m1 = (shade > 1.0) ? 1.0 : shade;                    // Math.min(1, shade); NaN stays NaN
      s->sky_rgb[c] = (m1 < amb) ? (double)amb : m1;                    // Math.max(amb, .) as the kernel's maxa() evaluates it
    }
  }
  s->host_objects.assign(ob, ob + hd->n_objects);
  // device copy of the blob: the `reserved` slot of each sphere record carries 1/r for the product kernel
  s->host_blob.assign((const uint8_t *)blob, (const uint8_t *)blob + bytes);
  {
    rt_sphere *pob = (rt_sphere *)(s->host_blob.data() + hd->objects_offset);
    for (uint32_t i = 0; i < hd->n_objects; i++) pob[i].reserved = 1.0 / sqrt(pob[i].r2);
  }
  camera_decisions(s);
  // Two orderings of the spheres.  A = the scene's own order (strict kernels, counting variant).  B = the enclosing sphere moved to
  // the end, so that the product kernel's loops run over [0, N-1) and never test it.
  const uint32_t NO = hd->n_objects, NL = hd->n_lights;
  s->has_b = s->enclosing != ~0u;
  const bool has_b = s->has_b;
  const int n_ord = has_b ? 2 : 1;
  const rt_sphere *pob_a = (const rt_sphere *)(s->host_blob.data() + hd->objects_offset);
  s->host_objects_b.clear();
  if (has_b) {
    for (uint32_t i = 0; i < NO; i++) if (i != s->enclosing) s->host_objects_b.push_back(pob_a[i]);
    s->host_objects_b.push_back(pob_a[s->enclosing]);
  }
  const uint32_t n_loop_b = has_b ? NO - 1 : NO;       // spheres in the product kernel's loops
  static const uint32_t sgrid_min = RT_TEST_ENV("RT_SGRID_MIN") ? (uint32_t)atoi(RT_TEST_ENV("RT_SGRID_MIN")) : RT_SGRID_MIN_LOOP;     // A/B switches (test build)
  static const uint32_t btable_min = RT_TEST_ENV("RT_BTABLE_MIN") ? (uint32_t)atoi(RT_TEST_ENV("RT_BTABLE_MIN")) : RT_BTABLE_MIN_LOOP;
  const bool want_shadow_grid = n_loop_b > sgrid_min && NL > 0;
  const bool want_bounce_table = n_loop_b > btable_min && hd->segs > 1;      // rays bounce at all only from depth 2 on
  // few spheres: the cull rectangles ride in the LDS image; scenes that get a shadow grid or a bounce table run the many-sphere
  // kernel variant, which fetches them per lane (rt_kernel.hip: 64 spheres + the fold state then fit 32 KB of LDS, five workgroups
  // per CU instead of four)
  s->cull_in_lds = !(want_shadow_grid || want_bounce_table);
  s->lds_image_bytes = (size_t)NO * (sizeof(rt_mtl) + (s->cull_in_lds ? sizeof(rt_geom) : 0u)) + sizeof descs;
  s->lds_bytes = (unsigned)s->lds_image_bytes;
  std::vector<uint64_t> sg, bt;
  if (want_shadow_grid) sg = build_shadow_grid(has_b ? s->host_objects_b.data() : pob_a, n_loop_b, NL, s->lights);
  if (want_bounce_table) bt = build_bounce_table(has_b ? s->host_objects_b.data() : pob_a, NO, n_loop_b);
  // ---- the arena's layout (every part 256-byte aligned) ----
  auto up = [](size_t x) { return (x + 255u) & ~(size_t)255u; };
  size_t at = 0;
  const size_t off_blob = at; at = up(at + bytes);
  const size_t off_tex = at; at = up(at + sizeof descs);
  const size_t geom_per_order = (size_t)NO * (1 + NL);                 // [plain N | anchored at light k: NL x N]
  const size_t off_geom = at; at = up(at + (geom_per_order * n_ord + 1) * sizeof(rt_geom));   // + one record of padding: the kernel's scans fetch a light's first two records at once, also when it has one
  const size_t off_objs_b = at; at = up(at + (has_b ? NO * sizeof(rt_sphere) : 0));
  const size_t off_img = at; at = up(at + (s->cull_in_lds ? 0 : s->lds_image_bytes * n_ord + 4096u));   // the many-sphere kernel reads whole 4 KB pieces (rt_kernel.hip staging)
  const size_t off_sg = at; at = up(at + sg.size() * sizeof(uint64_t));
  const size_t off_bt = at; at = up(at + bt.size() * sizeof(uint64_t));
  s->cam_lds_offset = up((size_t)n_ord * 2u * NO * sizeof(rt_geom));
  s->cam_bytes_used = s->cam_lds_offset + (s->cull_in_lds ? s->lds_image_bytes * n_ord : 0);
  s->cam_bytes = s->cam_bytes_used + (s->cull_in_lds ? 4096u : 0);
  const size_t off_cam0 = at; at = up(at + s->cam_bytes);
  const size_t off_cam1 = at; at = up(at + s->cam_bytes);
  s->arena_bytes = at;
  std::vector<uint8_t> host(at, 0);
  memcpy(host.data() + off_blob, s->host_blob.data(), bytes);
  memcpy(host.data() + off_tex, descs, sizeof descs);
  auto anchored = [&](const rt_sphere &o, const double a[3]) {
    const double lx = o.origin[0] - a[0], ly = o.origin[1] - a[1], lz = o.origin[2] - a[2];
    return rt_geom{lx, ly, lz, (lx * lx + ly * ly + lz * lz) - o.r2};
  };
  for (int ord = 0; ord < n_ord; ord++) {
    const rt_sphere *src = ord ? s->host_objects_b.data() : pob_a;
    rt_geom *dst = (rt_geom *)(host.data() + off_geom) + ord * geom_per_order;
    for (uint32_t i = 0; i < NO; i++) {
      dst[i] = rt_geom{src[i].origin[0], src[i].origin[1], src[i].origin[2], src[i].r2};
      for (uint32_t k = 0; k < NL; k++) dst[(size_t)NO * (1 + k) + i] = anchored(src[i], s->lights[k]);
    }
  }
  ((rt_geom *)(host.data() + off_geom))[geom_per_order * n_ord] = rt_geom{0.0, 0.0, 0.0, -1.0};
  if (has_b) memcpy(host.data() + off_objs_b, s->host_objects_b.data(), NO * sizeof(rt_sphere));
  if (!sg.empty()) memcpy(host.data() + off_sg, sg.data(), sg.size() * sizeof(uint64_t));
  if (!bt.empty()) memcpy(host.data() + off_bt, bt.data(), bt.size() * sizeof(uint64_t));
  // the LDS images' camera-independent part: [materials | texture descriptors] (fill_camera_block below writes them again, with
  // their cull rectangles, when they live in the camera block)
  if (!s->cull_in_lds) for (int ord = 0; ord < n_ord; ord++) fill_lds_image(s, host.data() + off_img + ord * s->lds_image_bytes, ord);
  fill_camera_block(s, host.data() + off_cam0);
  memcpy(host.data() + off_cam1, host.data() + off_cam0, s->cam_bytes_used);
  // ---- one allocation, one copy ----
  hipError_t e = hipMalloc((void **)&s->arena, s->arena_bytes);
  if (e == hipSuccess) e = hipMemcpy(s->arena, host.data(), s->arena_bytes, hipMemcpyHostToDevice);
  // pinned staging for what follows a camera move
  {
    const size_t table_dyn = 512u + (size_t)NO * (sizeof(rt_ball) + sizeof(rt_cost_rect));      // a launch table's parameters, cone-test spheres and cost rectangles
    s->stage_bytes = up(s->cam_bytes > table_dyn ? s->cam_bytes : table_dyn);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->stage_pool, s->stage_bytes * 16u, hipHostMallocDefault);
    for (size_t i = 0; i < 16u; i++) {
      s->stages[i].h = s->stage_pool ? s->stage_pool + i * s->stage_bytes : nullptr;
      if (e == hipSuccess) e = hipEventCreateWithFlags(&s->stages[i].done, hipEventDisableTiming);
    }
  }
  if (e != hipSuccess) {
    const std::string why = hipGetErrorString(e);
    rt_scene_free(s);
    return fail(RT_ERR_DEVICE, "scene upload: %s", why.c_str());
  }
  s->d_blob = s->arena + off_blob;
  s->d_texdesc = (rt_texture_desc *)(s->arena + off_tex);
  s->d_geom = (rt_geom *)(s->arena + off_geom);
  s->d_objects_b = has_b ? (rt_sphere *)(s->arena + off_objs_b) : nullptr;
  s->d_shadow_grid = sg.empty() ? nullptr : (uint64_t *)(s->arena + off_sg);
  s->d_bounce_table = bt.empty() ? nullptr : (uint64_t *)(s->arena + off_bt);
  s->d_cam_buf[0] = s->arena + off_cam0; s->d_cam_buf[1] = s->arena + off_cam1;
  s->d_lds_image = s->cull_in_lds ? nullptr : s->arena + off_img;
  *out = s;
  return RT_OK;
}

extern "C" void rt_scene_free(rt_scene_dev *s) {
  if (!s) return;
  if (G.inited && s->device < (int)G.dev.size()) (void)hipSetDevice(G.dev[s->device].hip_id);
  (void)hipDeviceSynchronize();                    // nothing of this scene is in flight any more
  if (s->arena) (void)hipFree(s->arena);
  for (rt_scene_dev::stage_slot &g : s->stages) if (g.done) (void)hipEventDestroy(g.done);
  if (s->stage_pool) (void)hipHostFree(s->stage_pool);
  for (int b = 0; b < 2; b++) { if (s->old_done[b]) (void)hipEventDestroy(s->old_done[b]); if (s->prep_done[b]) (void)hipEventDestroy(s->prep_done[b]); }
  if (s->side) (void)hipStreamDestroy(s->side);
  for (rt_scene_dev::order_entry &e : s->orders) free_order_entry(e);
  for (const rt_scene_dev::mark_state &m : s->mark_states) (void)hipFree(m.d_marks);
  if (s->h_known_pool) (void)hipHostFree(s->h_known_pool);
  delete s;
}

// The camera of a resident scene moves (lookAt, main.js:92-100; the reference recomputes everything per redraw, main.js:180-201).
// What depends on it - the camera-anchored geometry, the cull rectangles, the LDS images that hold them (ONE block of the scene's
// arena) and the launch tables of the frame sizes in use - exists twice, for even and odd camera generations.  The move stages the
// new block (pinned host memory) and, on the scene's OWN side stream, copies it and rebuilds the tables the previous camera's frames
// used: beside those frames' launches, which are still running on the caller's stream, and ordered against them by two events
// (rt_scene_dev: old_done, prep_done).  A plain `set_camera; render; set_camera; render ...` loop on one stream thereby overlaps frame
// k + 1's table build with frame k's trace - what round 3 needed two scene handles on two streams for.  Nothing waits on the host
// unless launches of this scene are in flight on several caller streams (then the device is drained first).
namespace {
bool build_table(rt_scene_dev *s, int found, const rt_tiles *tiles, uint32_t tiles_x, uint32_t rb_per_tile, double proj_w, double proj_h, double proj_d, hipStream_t stream,
                 rt_scene_dev::stage_slot *cam);
}  // namespace

extern "C" int rt_scene_set_camera(rt_scene_dev *s, const double origin[3], const double axis_x[3], const double axis_y[3], const double axis_z[3], void *hip_stream) {
  if (!s || !origin || !axis_x || !axis_y || !axis_z) return fail(RT_ERR_INVALID, "rt_scene_set_camera: NULL argument");
  int rc = ensure_device(s->device);
  if (rc) return rc;
  (void)hip_stream;                                  // (kept in the signature: the copy and the rebuilds run on the scene's own side stream)
  std::lock_guard<std::mutex> lk(s->launch_mu);
  rt_scene_header nh = s->hd;
  memcpy(nh.cam_origin, origin, 24); memcpy(nh.cam_axis_x, axis_x, 24); memcpy(nh.cam_axis_y, axis_y, 24); memcpy(nh.cam_axis_z, axis_z, 24);
  if (memcmp(&nh, &s->hd, sizeof nh) == 0) return RT_OK;
  // the two orderings of the scene's tables are built around the sphere that encloses everything INCLUDING the camera
  if (enclosing_sphere(&nh, s->host_objects.data(), s->lights) != s->enclosing)
    return fail(RT_ERR_UNSUPPORTED, "rt_scene_set_camera: the camera crossed the enclosing sphere (the scene's tables are laid out around it): upload the scene again");
  if (!s->side) {
    // HIGH priority: its few hundred waves are launched INTO a chip the previous frame's trace keeps full; at normal priority the
    // table build's workgroups waited for slots and took 77 us instead of 20 (profiles/r04_ab_log.md)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    HIP_TRY(hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, prio_hi));
    for (int b = 0; b < 2; b++) { HIP_TRY(hipEventCreateWithFlags(&s->old_done[b], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&s->prep_done[b], hipEventDisableTiming)); }
  }
  // launches of this scene in flight on SEVERAL caller streams: no single event covers them (rare: drain the device)
  if (s->any_launch && s->several_streams) { HIP_TRY(hipDeviceSynchronize()); s->any_launch = false; s->several_streams = false; s->launched_since_move = false; s->old_done_valid[0] = s->old_done_valid[1] = false; }
  const uint64_t old_gen = s->cam_gen;
  s->hd = nh;
  memcpy(s->host_blob.data(), &nh, sizeof nh);
  camera_decisions(s);
  const uint64_t G = ++s->cam_gen;
  const uint32_t b = (uint32_t)(G & 1u);
  s->renders_with_camera = 0;
  // every launch so far (generations < G) precedes this event on the caller's stream; the move to G + 1 will write block / tables
  // (G + 1) & 1 - the ones generation G - 1 used - only behind it.  (No launch since the last move: the older record still covers them.)
  if (s->launched_since_move && s->any_launch) { HIP_TRY(hipEventRecord(s->old_done[(G - 1u) & 1u], s->last_stream)); s->old_done_valid[(G - 1u) & 1u] = true; }
  s->launched_since_move = false;
  // block and tables b were last read by generation G - 2
  if (s->old_done_valid[b]) HIP_TRY(hipStreamWaitEvent(s->side, s->old_done[b], 0));
  rt_scene_dev::stage_slot *slot = nullptr;
  uint8_t *st = acquire_stage(s, &slot);
  fill_camera_block(s, st);
  // the tables the previous camera's frames used are rebuilt now, on the side stream, beside those frames' launches: the next render
  // of such a frame finds its table (up to four; others are built by the launch that needs them, on its stream).  Many-sphere scenes:
  // the first frame from a camera takes the table without shadow masks (rt_render_batch: masks_pay).
  const uint32_t n_loop = s->hd.n_objects - (s->enclosing != ~0u ? 1u : 0u);
  int built = 0;
  bool cam_sent = false;
  for (size_t i = 0; i < s->orders.size() && built < 4; i++) {
    rt_scene_dev::order_entry &e = s->orders[i];
    if (e.used_gen != old_gen || !e.built || (n_loop > 16u && e.masks)) continue;
    const rt_tiles tiles = {e.tile_rows, e.tile_first, e.tile_stride, e.n_tiles};
    const bool ss2 = e.ss == 2u;
    const uint32_t rows_per_wg = ss2 ? 2u : RT_TILE_H;
    const uint32_t tiles_x = (e.w + RT_TILE_W - 1) / RT_TILE_W, rb_per_tile = (e.tile_rows + rows_per_wg - 1) / rows_per_wg;
    const double sw = ss2 ? 2.0 * e.w : (double)e.w, sh = ss2 ? 2.0 * e.h : (double)e.h;        // (the expressions of render_batch_impl: the same bits)
    const double projA = s->hd.fov_deg * M_PI / 180.0, pw = sw / 2.0, ph = sh / 2.0, pd = pw / tan(projA / 2.0);
    if (!build_table(s, (int)i, &tiles, tiles_x, rb_per_tile, pw, ph, pd, s->side, cam_sent ? nullptr : slot)) return RT_ERR_DEVICE;
    cam_sent = true;
    built++;
  }
  if (!cam_sent) {
    hipError_t e = (hipError_t)rt_launch_small_copy(cam_block(s), st, s->cam_bytes_used, nullptr, nullptr, 0u, s->side);
    if (e == hipSuccess) e = hipEventRecord(slot->done, s->side);
    if (e != hipSuccess) return fail(RT_ERR_DEVICE, "camera block: %s", hipGetErrorString(e));
  }
  HIP_TRY(hipEventRecord(s->prep_done[b], s->side));
  s->prep_valid[b] = true;
  s->prep_waited.clear();
  return RT_OK;
}

// ------------------------------------------------------------------------------------ launch
extern "C" int rt_render_tiles_device(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, void *d_out, void *hip_stream,
                                      uint32_t flags, rt_stats *stats) {
  return rt_render_batch_device(s, w, h, tiles, 1u, d_out, 0u, hip_stream, flags, stats);
}

namespace {
int render_batch_impl(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, uint32_t n_frames, void *d_out, uint64_t frame_stride_bytes,
                      void *const *d_frames, void *hip_stream, uint32_t flags, rt_stats *stats, uint32_t ss_override = 0u);
#ifdef RT_TESTING
thread_local struct { double *d_buf; uint32_t x, y; } g_probe = {nullptr, 0u, 0u};
#endif
}  // namespace

#ifdef RT_TESTING
// Test build only: the ray tree of ONE sample (sample-grid coordinates sx, sy) as RT_PROBE_NODES records of RT_PROBE_WORDS
// doubles {path, hcode, t, hit point, normal, direction, sampled colour, diffuse, specular, segs left, light intensity after
// the scans, ray origin, children mask, valid}; the row that holds the sample is rendered into scratch memory.
extern "C" int rt_test_probe(rt_scene_dev *s, uint32_t w, uint32_t h, uint32_t sx, uint32_t sy, uint32_t flags, double *out_records) {
  if (!s || !out_records) return fail(RT_ERR_INVALID, "rt_test_probe: NULL argument");
  int rc = ensure_device(s->device);
  if (rc) return rc;
  const size_t bytes = (size_t)RT_PROBE_NODES * RT_PROBE_WORDS * sizeof(double);
  double *d_probe = nullptr;
  void *d_row = nullptr;
  HIP_TRY(hipMalloc((void **)&d_probe, bytes));
  hipError_t e = hipMemset(d_probe, 0, bytes);
  if (e == hipSuccess) e = hipMalloc(&d_row, (size_t)w * 4u);
  if (e != hipSuccess) { (void)hipFree(d_probe); return fail(RT_ERR_DEVICE, "rt_test_probe: %s", hipGetErrorString(e)); }
  const uint32_t ss = s->hd.supersample;
  rt_tiles t = {1u, sy / ss, 1u, 1u};
  rt_stats st;
  g_probe.d_buf = d_probe; g_probe.x = sx; g_probe.y = sy;
  rc = rt_render_tiles_device(s, w, h, &t, d_row, nullptr, flags & ~(uint32_t)RT_FLAG_RGB24, &st);
  g_probe.d_buf = nullptr;
  if (!rc) { e = hipMemcpy(out_records, d_probe, bytes, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, "rt_test_probe: %s", hipGetErrorString(e)); }
  (void)hipFree(d_probe); (void)hipFree(d_row);
  return rc;
}
#endif

extern "C" int rt_render_batch_device(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, uint32_t n_frames, void *d_out,
                                      uint64_t frame_stride_bytes, void *hip_stream, uint32_t flags, rt_stats *stats) {
  if (!d_out) return fail(RT_ERR_INVALID, "NULL scene, tiles or output");
  return render_batch_impl(s, w, h, tiles, n_frames, d_out, frame_stride_bytes, nullptr, hip_stream, flags, stats);
}

extern "C" int rt_render_scatter_device(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, uint32_t n_frames, void *const *d_frames,
                                        void *hip_stream, uint32_t flags, rt_stats *stats) {
  if (!d_frames) return fail(RT_ERR_INVALID, "NULL frame pointer array");
  if (n_frames == 0 || n_frames > RT_MAX_SCATTER) return fail(RT_ERR_INVALID, "scatter: n_frames %u not in 1..%u", n_frames, RT_MAX_SCATTER);
  if (flags & RT_FLAG_RGB24) return fail(RT_ERR_INVALID, "scatter writes whole RGBA8 frames: RT_FLAG_RGB24 does not apply");
  for (uint32_t f = 0; f < n_frames; f++) if (!d_frames[f] || ((uintptr_t)d_frames[f] & 3u)) return fail(RT_ERR_INVALID, "scatter: frame pointer %u is NULL or unaligned", f);
  return render_batch_impl(s, w, h, tiles, n_frames, nullptr, 0u, d_frames, hip_stream, flags, stats);
}

namespace {
// a pinned host word of the scene's pool (generation << 32 | value + 1, written by a kernel): [0, RT_KNOWN_WORDS) the mark states',
// [RT_KNOWN_WORDS, 2 RT_KNOWN_WORDS) the launch tables'
volatile unsigned long long *known_word(rt_scene_dev *s, size_t index) {
  if (!s->h_known_pool) {
    if (hipHostMalloc((void **)&s->h_known_pool, 2 * RT_KNOWN_WORDS * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); s->h_known_pool = nullptr; return nullptr; }
    memset(s->h_known_pool, 0, 2 * RT_KNOWN_WORDS * sizeof(unsigned long long));
  }
  return index < 2 * RT_KNOWN_WORDS ? s->h_known_pool + index : nullptr;
}
// what a kernel published for camera generation `gen`: value + 1, or 0 (nothing yet, or an older camera's)
uint32_t known_value(const volatile unsigned long long *p, uint64_t gen) {
  if (!p) return 0u;
  const unsigned long long v = *p;
  return (uint32_t)(v >> 32) == (uint32_t)gen ? (uint32_t)v : 0u;
}

// The camera has moved since `stream` last launched the scene: its work comes behind the copy of the camera's block and the tables
// rebuilt for it on the scene's side stream (rt_scene_set_camera).  One event wait per (stream, camera); launch_mu held.
int behind_the_camera(rt_scene_dev *s, hipStream_t stream) {
  const uint32_t cb = (uint32_t)(s->cam_gen & 1u);
  if (!s->prep_valid[cb]) return RT_OK;
  for (const rt_scene_dev::waited_on &q : s->prep_waited) if (q.stream == stream && q.gen == s->cam_gen) return RT_OK;
  HIP_TRY(hipStreamWaitEvent(stream, s->prep_done[cb], 0));
  if (s->prep_waited.size() >= 16u) s->prep_waited.clear();
  s->prep_waited.push_back(rt_scene_dev::waited_on{stream, s->cam_gen});
  return RT_OK;
}

// Build the launch table of entry `found` for the scene's CURRENT camera (generation g, into the entry's table g & 1) on `stream`:
// one small copy of its parameters - which also carries the staged camera block `cam` of a move, if given - and three small launches
// (rt_tables_gpu.hip); nothing waits for them.  Called with the scene's launch_mu held.  false: rt_last_error says why.
bool build_table(rt_scene_dev *s, int found, const rt_tiles *tiles, uint32_t tiles_x, uint32_t rb_per_tile, double proj_w, double proj_h, double proj_d, hipStream_t stream,
                 rt_scene_dev::stage_slot *cam) {
  rt_scene_dev::order_entry &e = s->orders[found];
  const uint32_t tb = (uint32_t)(s->cam_gen & 1u);
  rt_table_params P;
  std::vector<rt_ball> balls;
  std::vector<rt_cost_rect> rects;
  if (make_table_params(&s->hd, s->host_objects.data(), s->host_cull, s->tile_weight, e.w, e.h, e.ss, tiles, tiles_x, rb_per_tile, proj_w, proj_h, proj_d, e.ranked, e.sky,
                        s->enclosing, e.masks, e.cands, s->lights, &P, &balls, &rects)) {
    fail(RT_ERR_INVALID, "a launch of %llu workgroups is beyond the launch table", (unsigned long long)tiles_x * tiles->n_tiles * rb_per_tile);
    return false;
  }
  P.flags |= e.part == 1u ? RT_TABLE_NO_SKY : (e.part == 2u ? RT_TABLE_SKY_ONLY : 0u);
  const uint32_t n = P.tiles_x * P.ny;
  const size_t hist_words = (size_t)P.ny * P.cost_bins;
  rt_table_dev &T = e.Tb[tb];
  // the table's device memory: one allocation behind all its arrays; the per-row histograms grow with the camera's cost range
  if (!e.d_blockb[tb] || e.hist_wordsb[tb] < hist_words) {
    if (e.d_blockb[tb]) { (void)hipDeviceSynchronize(); (void)hipFree(e.d_blockb[tb]); e.d_blockb[tb] = nullptr; }
    auto up = [](size_t x) { return (x + 255u) & ~(size_t)255u; };
    const size_t cap_hist = hist_words > (size_t)P.ny * 128u ? hist_words : (size_t)P.ny * 128u;
    const size_t dyn_bytes = up(sizeof(rt_table_params)) + up((size_t)RT_MAX_OBJECTS * sizeof(rt_ball)) + up((size_t)RT_MAX_OBJECTS * sizeof(rt_cost_rect)) + 256u;
    size_t at = 0;
    const size_t o_dyn = at; at += dyn_bytes;
    const size_t o_blk = at; at = up(at + (size_t)n * 12u);
    const size_t o_item = at; at = up(at + (size_t)n * 4u);
    const size_t o_rank = at; at = up(at + (size_t)n * 4u);
    const size_t o_hist = at; at = up(at + cap_hist * 4u);
    const size_t o_bins = at; at = up(at + (size_t)(RT_COST_MAX + 1u) * 4u);
    const size_t o_head = at; at = up(at + 16u + ((size_t)(n + 7u) / 8u) * 8u * 16u);
    uint8_t *blk = nullptr;
    hipError_t er = hipMalloc((void **)&blk, at);
    if (er == hipSuccess) er = hipMemsetAsync(blk + o_dyn + dyn_bytes - 256u, 0, 256u, stream);     // the scan's ticket
    if (er == hipSuccess && !e.built) er = hipEventCreateWithFlags(&e.built, hipEventDisableTiming);
    if (er != hipSuccess) { if (blk) (void)hipFree(blk); fail(RT_ERR_DEVICE, "launch table (%zu bytes): %s", at, hipGetErrorString(er)); return false; }
    e.d_blockb[tb] = blk;
    e.hist_wordsb[tb] = cap_hist;
    T.params = (const rt_table_params *)(blk + o_dyn);
    T.ticket = (uint32_t *)(blk + o_dyn + dyn_bytes - 256u);
    T.blk = (uint32_t *)(blk + o_blk); T.item = (uint32_t *)(blk + o_item); T.rank_in_row = (uint32_t *)(blk + o_rank);
    T.row_hist = (uint32_t *)(blk + o_hist); T.bin_start = (uint32_t *)(blk + o_bins);
    T.header = (uint32_t *)(blk + o_head); T.entries = T.header + 4;
    T.known = (unsigned long long *)e.known;
  } else if (e.shared && stream != s->side) {
    // rebuilt lazily on a caller's stream while launches on ANOTHER caller's stream may still read this table's older contents: only
    // when nothing is in flight (rare; a move's own rebuilds, on the side stream, come behind old_done instead)
    (void)hipDeviceSynchronize();
  }
  e.n_blocks = n;
  e.cost_bins = P.cost_bins;
  e.cam_gen = s->cam_gen;
  T.known_tag = (uint32_t)s->cam_gen;
  e.built_on = stream; e.shared = false;
  // parameters, cone-test spheres and cost rectangles, packed: one staging slot, ONE small copy kernel - which also carries the
  // scene's camera block of a move (an SDMA copy in front of the build would cost two engine hand-overs, more than the copy)
  rt_scene_dev::stage_slot *slot = nullptr;
  uint8_t *st = acquire_stage(s, &slot);
  const size_t o_balls = (sizeof(rt_table_params) + 15u) & ~(size_t)15u, o_rects = o_balls + balls.size() * sizeof(rt_ball);
  const size_t copy_bytes = o_rects + rects.size() * sizeof(rt_cost_rect);
  T.balls = (const rt_ball *)((const uint8_t *)T.params + o_balls);
  T.rects = (const rt_cost_rect *)((const uint8_t *)T.params + o_rects);
  memcpy(st, &P, sizeof P);
  if (!balls.empty()) memcpy(st + o_balls, balls.data(), balls.size() * sizeof(rt_ball));
  if (!rects.empty()) memcpy(st + o_rects, rects.data(), rects.size() * sizeof(rt_cost_rect));
  hipError_t er = (hipError_t)rt_launch_small_copy((void *)T.params, st, copy_bytes, cam ? cam_block(s) : nullptr, cam ? cam->h : nullptr, cam ? s->cam_bytes_used : 0u, stream);
  if (er == hipSuccess) er = hipEventRecord(slot->done, stream);
  if (er == hipSuccess && cam) er = hipEventRecord(cam->done, stream);
  if (er == hipSuccess) er = (hipError_t)rt_launch_table_build(&T, P.tiles_x, P.ny, P.cost_bins, (uint32_t)copy_bytes, ((P.flags & RT_TABLE_WIDE) ? 1 : 0) | (stream == s->side ? 2 : 0), stream);
  if (er == hipSuccess) er = hipEventRecord(e.built, stream);
  if (er != hipSuccess) { e.cam_gen = 0; fail(RT_ERR_DEVICE, "launch table build: %s", hipGetErrorString(er)); return false; }
  return true;
}

// The launch table of this (frame size, tile set, flags) for the scene's CURRENT camera: found - built by rt_scene_set_camera on the
// scene's side stream, or by an earlier launch - or built now on `stream`.  Called with the scene's launch_mu held.  Returns the
// entry's index, or -1 (rt_last_error says why).
int dispatch_order(rt_scene_dev *s, uint32_t w, uint32_t h, uint32_t ss, const rt_tiles *tiles, uint32_t tiles_x, uint32_t rb_per_tile,
                   double proj_w, double proj_h, double proj_d, int ranked, bool mark_sky, bool shadow_masks, bool name_candidates, uint32_t sky_part, hipStream_t stream) {
  // sky_part: 0 every entry; 1 (RT_FLAG_NO_SKY) a table without the sky runs; 2 (RT_FLAG_SKY_ONLY) a table of nothing else - tables of
  // their own, so that the trace kernel knows nothing of it (a test of the launch record in its prologue cost the headline 1.5 %)
  int found = -1;
  for (size_t i = 0; i < s->orders.size(); i++) {
    const rt_scene_dev::order_entry &e = s->orders[i];
    if (e.w == w && e.h == h && e.ss == ss && e.tile_rows == tiles->tile_rows && e.tile_first == tiles->tile_first && e.tile_stride == tiles->tile_stride &&
        e.n_tiles == tiles->n_tiles && e.ranked == ranked && e.sky == mark_sky && e.masks == shadow_masks && e.cands == name_candidates && e.part == sky_part) { found = (int)i; break; }
  }
  if (found >= 0 && s->orders[found].cam_gen == s->cam_gen) {
    rt_scene_dev::order_entry &e = s->orders[found];
    // built on another caller's stream: this stream's launches come behind the build (the side stream's builds: behind prep_done,
    // which every stream waits for before its first launch with a camera)
    if (e.built_on != stream && e.built_on != s->side) { if (hipStreamWaitEvent(stream, e.built, 0) != hipSuccess) { fail(RT_ERR_DEVICE, "launch table: hipStreamWaitEvent"); return -1; } e.shared = true; }
    e.used_gen = s->cam_gen;
    return found;
  }
  if (found < 0) {
    if ((uint64_t)tiles_x * tiles->n_tiles * rb_per_tile >= (1ull << 31) || tiles_x > 2048u) {
      fail(RT_ERR_INVALID, "a launch of %llu workgroups is beyond the launch table", (unsigned long long)tiles_x * tiles->n_tiles * rb_per_tile);
      return -1;
    }
    // a scene that has been rendered with 64 different (frame size, tile set) pairs gives up its oldest table (nothing of it may be
    // in flight: the device is drained first; rare)
    if (s->orders.size() >= 64u) {
      (void)hipDeviceSynchronize();
      free_order_entry(s->orders[s->order_evict % 64u]);
      found = (int)(s->order_evict++ % 64u);
      for (rt_scene_dev::mark_state &m : s->mark_states) if (m.order_index == (uint32_t)found && m.h_known) *m.h_known = 0ull;     // its mark counts were another table's
    } else {
      s->orders.push_back(rt_scene_dev::order_entry());
      found = (int)s->orders.size() - 1;
    }
    rt_scene_dev::order_entry &e = s->orders[found];
    memset(&e, 0, sizeof e);
    e.w = w; e.h = h; e.ss = ss; e.tile_rows = tiles->tile_rows; e.tile_first = tiles->tile_first; e.tile_stride = tiles->tile_stride; e.n_tiles = tiles->n_tiles;
    e.ranked = ranked; e.sky = mark_sky; e.masks = shadow_masks; e.cands = name_candidates; e.part = sky_part;
    e.known = known_word(s, RT_KNOWN_WORDS + (size_t)found);
    if (e.known) *e.known = 0ull;                        // (a table evicted from this slot may have published its count for the same camera)
  }
  if (!build_table(s, found, tiles, tiles_x, rb_per_tile, proj_w, proj_h, proj_d, stream, nullptr)) return -1;
  s->orders[found].used_gen = s->cam_gen;
  return found;
}

}  // namespace

#ifdef RT_TESTING
// Test build only: the launch table as the library builds it ON THE GPU for `tiles` of the w x h frame of a resident scene (its
// current camera), read back: same arguments and layout as the host-logic probe rt_scene_launch_table below, whose table (the host
// build of the same rt_block.h) it must equal word for word.
extern "C" int rt_test_launch_table(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, int ranked, uint32_t *out_entries, uint32_t *n_workgroups, uint32_t *n_blocks) {
  if (!s || !tiles || !n_workgroups) return fail(RT_ERR_INVALID, "rt_test_launch_table: NULL argument");
  int rc = ensure_device(s->device);
  if (rc) return rc;
  hipStream_t stream = G.dev[s->device].stream;
  const uint32_t ss = s->hd.supersample, rows_per_wg = ss == 2u ? 2u : RT_TILE_H;
  if (ss > 2u) return fail(RT_ERR_INVALID, "supersample 3 and 4 launch on the sample grid");
  const uint32_t tiles_x = (w + RT_TILE_W - 1) / RT_TILE_W, rb_per_tile = (tiles->tile_rows + rows_per_wg - 1) / rows_per_wg;
  const double pw = (double)w * ss / 2.0, ph = (double)h * ss / 2.0, pd = pw / tan(s->hd.fov_deg * M_PI / 180.0 / 2.0);
  std::lock_guard<std::mutex> lk(s->launch_mu);
  if ((rc = behind_the_camera(s, stream))) return rc;
  const int oi = dispatch_order(s, w, h, ss, tiles, tiles_x, rb_per_tile, pw, ph, pd, (ranked & 1) != 0, (ranked & 2) != 0, (ranked & 4) != 0, (ranked & 4) != 0,
                                (ranked & 8) ? 1u : ((ranked & 16) ? 2u : 0u), stream);
  if (oi < 0) return RT_ERR_DEVICE;
  HIP_TRY(hipStreamSynchronize(stream));
  const rt_scene_dev::order_entry &e = s->orders[oi];
  uint32_t header[4];
  const rt_table_dev &T = e.Tb[s->cam_gen & 1u];
  HIP_TRY(hipMemcpy(header, T.header, sizeof header, hipMemcpyDeviceToHost));
  if (known_value(e.known, s->cam_gen) != header[0] + 1u) return fail(RT_ERR_STATE, "the build published %u entries to the host, its header says %u", known_value(e.known, s->cam_gen), header[0] + 1u);
  *n_workgroups = header[0];
  if (n_blocks) *n_blocks = e.n_blocks;
  if (out_entries) HIP_TRY(hipMemcpy(out_entries, T.entries, (size_t)((e.n_blocks + 7u) / 8u) * 8u * 16u, hipMemcpyDeviceToHost));
  return RT_OK;
}
#endif

// Host-logic probe (no GPU): the product kernel's launch table for `tiles` of the w x h frame as the HOST builds it (rt_tables.cpp;
// the library builds the same table on the GPU, rt_tables_gpu.hip): out_entries receives 4 words per slot, 8 * ceil(blocks / 8) slots
// with workgroup b's entry in slot (b % 8) * ceil(blocks / 8) + b / 8; *n_workgroups = the number of entries, *n_blocks = the number
// of blocks.  Pass out_entries = NULL to ask for the two numbers only.
extern "C" int rt_scene_launch_table(const void *blob, size_t bytes, uint32_t w, uint32_t h, const rt_tiles *tiles, int ranked,
                                     uint32_t *out_entries, uint32_t *n_workgroups, uint32_t *n_blocks) {
  int rc = rt_scene_validate(blob, bytes);
  if (rc) return rc;
  if (!tiles || !n_workgroups || w == 0 || h == 0 || w > 65536 || h > 65536 || tiles->tile_rows == 0 || tiles->tile_stride == 0 || tiles->n_tiles == 0)
    return fail(RT_ERR_INVALID, "bad launch-table probe arguments");
  const rt_scene_header *hd = (const rt_scene_header *)blob;
  if (hd->supersample > 2) return fail(RT_ERR_INVALID, "supersample 3 and 4 launch on the sample grid: probe that size with a supersample-1 scene");
  const rt_sphere *ob = (const rt_sphere *)((const uint8_t *)blob + hd->objects_offset);
  std::vector<rt_geom> cull;
  std::vector<uint32_t> weight;
  scene_tile_weights(hd, ob, &cull, &weight);
  const uint32_t ss = hd->supersample, rows_per_wg = ss == 2u ? 2u : RT_TILE_H;
  const uint32_t tiles_x = (w + RT_TILE_W - 1) / RT_TILE_W, rb_per_tile = (tiles->tile_rows + rows_per_wg - 1) / rows_per_wg;
  if ((uint64_t)tiles->n_tiles * rb_per_tile > 65535u) return fail(RT_ERR_INVALID, "too many row blocks");
  const double pw = (double)w * ss / 2.0, ph = (double)h * ss / 2.0, pd = pw / tan(hd->fov_deg * M_PI / 180.0 / 2.0);
  // (bit 1 of `ranked`: also mark the workgroups no sphere but the enclosing one can show in, as a launch of a constant-background scene does)
  double lights[RT_MAX_LIGHTS][3];
  memset(lights, 0, sizeof lights);
  if (hd->n_lights) memcpy(lights, (const uint8_t *)blob + hd->lights_offset, hd->n_lights * 24u);
  const uint32_t sky_sphere = enclosing_sphere(hd, ob, lights);
  uint32_t n_entries = 0;
  const std::vector<uint32_t> table = build_launch_table(hd, ob, cull, weight, w, h, ss, tiles, tiles_x, rb_per_tile, pw, ph, pd, (ranked & 1) != 0, (ranked & 2) != 0, sky_sphere,
                                                         (ranked & 4) != 0, (ranked & 4) != 0, lights, &n_entries, (ranked & 8) ? 1u : ((ranked & 16) ? 2u : 0u));
  if (table.empty()) return fail(RT_ERR_INVALID, "a launch of this size is beyond the launch table");
  *n_workgroups = n_entries;
  if (n_blocks) *n_blocks = tiles_x * tiles->n_tiles * rb_per_tile;
  if (out_entries) memcpy(out_entries, table.data(), table.size() * sizeof(uint32_t));
  return RT_OK;
}

namespace {
// k x k box filter of the two-pass supersampling (k = 3, 4): `src` holds the rendered SAMPLES of this call's tiles as a band
// (rows of k*w RGBA8 samples, k sample rows per output row, tiles contiguous), the output pixel is (sum + k*k/2) / (k*k) per
// channel, alpha 255, stored where the trace kernel would have stored it: in the band (`out`, frame f at f*frame_stride) or,
// scatter mode, at its row of the whole frame out_frames[f].  One work-item per output pixel; rows walked by grid y.
struct rt_box_launch {
  const uint32_t *src; uint64_t src_frame_stride;      // in samples (words)
  uint32_t *out; uint64_t frame_stride; uint32_t *out_frames[RT_MAX_SCATTER]; uint32_t scatter;
  uint32_t w, h, band_rows, tile_rows, tile_first, tile_stride;
};
template <uint32_t K>
__global__ void __launch_bounds__(256) rt_box_filter_kernel(const rt_box_launch B) {
  const uint32_t x = blockIdx.x * 256u + threadIdx.x, f = blockIdx.z;
  if (x >= B.w) return;
  const uint32_t *__restrict__ src = B.src + (size_t)f * B.src_frame_stride;
  for (uint32_t lrow = blockIdx.y; lrow < B.band_rows; lrow += gridDim.y) {
    const uint32_t tile_i = lrow / B.tile_rows, trow = lrow - tile_i * B.tile_rows;
    const uint32_t frow = (B.tile_first + tile_i * B.tile_stride) * B.tile_rows + trow;
    if (frow >= B.h) continue;
    uint32_t r = 0, g = 0, b = 0;
#pragma unroll
    for (uint32_t j = 0; j < K; j++) {
      const uint32_t *__restrict__ p = src + ((size_t)lrow * K + j) * ((size_t)B.w * K) + (size_t)x * K;
#pragma unroll
      for (uint32_t i = 0; i < K; i++) { const uint32_t v = p[i]; r += v & 255u; g += (v >> 8) & 255u; b += (v >> 16) & 255u; }
    }
    const uint32_t px = ((r + K * K / 2u) / (K * K)) | (((g + K * K / 2u) / (K * K)) << 8) | (((b + K * K / 2u) / (K * K)) << 16) | 0xff000000u;
    if (B.scatter) B.out_frames[f][(size_t)frow * B.w + x] = px;
    else B.out[(size_t)f * B.frame_stride + (size_t)lrow * B.w + x] = px;
  }
}

// supersample 3 and 4: the k*w x k*h sample frame of this call's tiles is rendered by the ordinary launch
// (supersample 1 on the sample grid: same kernels, same centre-row/column rule, same tiles with k times the rows) into
// scratch memory, in pieces of at most ~512 MiB, and box-filtered into the caller's output.
int render_supersampled(rt_scene_dev *s, uint32_t k, uint32_t w, uint32_t h, const rt_tiles *tiles, uint32_t n_frames, void *d_out, uint64_t frame_stride_bytes,
                        void *const *d_frames, hipStream_t stream, uint32_t flags, rt_stats *stats) {
  if (flags & RT_FLAG_RGB24) return fail(RT_ERR_INVALID, "RT_FLAG_RGB24 needs supersample 1 or 2 (the %ux%u box filter stores RGBA8)", k, k);
  if ((uint64_t)w * k > 65536u || (uint64_t)h * k > 65536u) return fail(RT_ERR_INVALID, "supersample %u: the %llu x %llu sample grid exceeds 65536", k, (unsigned long long)w * k, (unsigned long long)h * k);
  const auto t_begin = std::chrono::steady_clock::now();
  const size_t row_bytes = (size_t)w * k * 4u * k;                      // the k sample rows of one output row
  const size_t budget = (size_t)512u << 20;
  // pieces: whole tiles while they fit, else (one tile per call, starting on a multiple of the piece height) row pieces of a tile
  uint32_t tiles_per_piece = (uint32_t)(budget / (row_bytes * tiles->tile_rows * (size_t)n_frames));
  uint32_t piece_rows = tiles->tile_rows;
  if (tiles_per_piece == 0) {
    tiles_per_piece = 1;
    piece_rows = (uint32_t)(budget / (row_bytes * n_frames)) / RT_TILE_H * RT_TILE_H;
    if (piece_rows == 0) piece_rows = RT_TILE_H;
    if (piece_rows >= tiles->tile_rows) piece_rows = tiles->tile_rows;
    else if (tiles->n_tiles != 1 || ((uint64_t)tiles->tile_first * tiles->tile_rows) % piece_rows != 0)
      return fail(RT_ERR_NOMEM, "supersample %u: a tile of %u rows needs more than 512 MiB of sample scratch; render smaller tiles", k, tiles->tile_rows);
  }
  rt_stats agg;
  memset(&agg, 0, sizeof agg);
  for (uint32_t t0 = 0; t0 < tiles->n_tiles; t0 += tiles_per_piece) {
    const uint32_t nt = (tiles->n_tiles - t0 < tiles_per_piece) ? tiles->n_tiles - t0 : tiles_per_piece;
    for (uint32_t r0 = 0; r0 < tiles->tile_rows; r0 += piece_rows) {
      // this piece as a tile set of the OUTPUT frame ...
      rt_tiles po;
      if (piece_rows == tiles->tile_rows) po = rt_tiles{tiles->tile_rows, tiles->tile_first + t0 * tiles->tile_stride, tiles->tile_stride, nt};
      else po = rt_tiles{piece_rows, (uint32_t)(((uint64_t)tiles->tile_first * tiles->tile_rows + r0) / piece_rows), 1u, 1u};
      if ((uint64_t)po.tile_first * po.tile_rows >= h) continue;
      // ... and of the sample frame
      const rt_tiles ps = {po.tile_rows * k, po.tile_first, po.tile_stride, po.n_tiles};
      const uint32_t band_rows = po.n_tiles * po.tile_rows;
      const size_t frame_words = (size_t)band_rows * k * w * k;
      void *scratch = nullptr;
      hipError_t e = hipMalloc(&scratch, frame_words * 4u * n_frames);
      if (e != hipSuccess) return fail(RT_ERR_NOMEM, "supersample scratch (%zu bytes): %s", frame_words * 4u * n_frames, hipGetErrorString(e));
#ifdef RT_TESTING
      (void)hipMemsetAsync(scratch, 0xA5, frame_words * 4u * n_frames, stream);      // test build: a sample nobody writes shows up as 0xA5, not as stale data
#endif
      rt_stats st;
      int rc = render_batch_impl(s, w * k, h * k, &ps, n_frames, scratch, frame_words * 4u, nullptr, stream, flags, stats ? &st : nullptr, 1u);
      if (!rc) {
        rt_box_launch B;
        memset(&B, 0, sizeof B);
        B.src = (const uint32_t *)scratch; B.src_frame_stride = frame_words;
        B.w = w; B.h = h; B.band_rows = band_rows; B.tile_rows = po.tile_rows; B.tile_first = po.tile_first; B.tile_stride = po.tile_stride;
        const size_t out_row0 = (size_t)t0 * tiles->tile_rows + r0;            // this piece's first row in the caller's band
        B.out = d_out ? (uint32_t *)d_out + out_row0 * w : nullptr; B.frame_stride = frame_stride_bytes / 4u;
        B.scatter = d_frames ? 1u : 0u;
        if (d_frames) for (uint32_t f = 0; f < n_frames; f++) B.out_frames[f] = (uint32_t *)d_frames[f];
        const dim3 grid((w + 255u) / 256u, band_rows < 65535u ? band_rows : 65535u, n_frames), block(256);
        if (k == 3u) hipLaunchKernelGGL(rt_box_filter_kernel<3u>, grid, block, 0, stream, B);
        else hipLaunchKernelGGL(rt_box_filter_kernel<4u>, grid, block, 0, stream, B);
        e = hipGetLastError();
        if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, "box filter launch: %s", hipGetErrorString(e));
      }
      (void)hipStreamSynchronize(stream);            // (a 9x / 16x render: the allocation and this wait are noise beside it)
      e = hipFree(scratch);
      if (rc) return rc;
      if (e != hipSuccess) return fail(RT_ERR_DEVICE, "supersample scratch release: %s", hipGetErrorString(e));
      if (stats) { agg.kernel_ms += st.kernel_ms; agg.rays += st.rays; agg.shadow_rays += st.shadow_rays; agg.sphere_tests += st.sphere_tests; }
    }
  }
  if (stats) {
    HIP_TRY(hipStreamSynchronize(stream));
    uint64_t px = 0;
    for (uint32_t i = 0; i < tiles->n_tiles; i++) {
      const uint64_t r0 = (uint64_t)(tiles->tile_first + (uint64_t)i * tiles->tile_stride) * tiles->tile_rows;
      if (r0 < h) px += ((r0 + tiles->tile_rows <= h) ? tiles->tile_rows : (h - r0)) * (uint64_t)w;
    }
    agg.pixels = px * n_frames;
    agg.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *stats = agg;
  }
  return RT_OK;
}

int render_batch_impl(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, uint32_t n_frames, void *d_out, uint64_t frame_stride_bytes,
                      void *const *d_frames, void *hip_stream, uint32_t flags, rt_stats *stats, uint32_t ss_override) {
  if (!s || !tiles) return fail(RT_ERR_INVALID, "NULL scene, tiles or output");
  if (n_frames == 0 || n_frames > 65535u) return fail(RT_ERR_INVALID, "n_frames %u not in 1..65535", n_frames);
  if ((frame_stride_bytes & 3u) != 0) return fail(RT_ERR_INVALID, "frame stride must be a multiple of 4 bytes");
  if (w == 0 || h == 0 || w > 65536 || h > 65536) return fail(RT_ERR_INVALID, "frame size %ux%u not in 1..65536", w, h);
  if (tiles->tile_rows == 0 || tiles->tile_stride == 0 || tiles->n_tiles == 0) return fail(RT_ERR_INVALID, "empty tile set");
  if ((uint64_t)tiles->n_tiles * tiles->tile_rows > (1ull << 24)) return fail(RT_ERR_INVALID, "too many rows in one call");
  if ((uint64_t)tiles->n_tiles * tiles->tile_rows * w >= (1ull << 32)) return fail(RT_ERR_INVALID, "a call may cover at most 2^32 - 1 pixels per frame");
  if ((flags & RT_FLAG_RGB24) && (w & 3u)) return fail(RT_ERR_INVALID, "RT_FLAG_RGB24 needs a frame width that is a multiple of 4 (got %u)", w);
  if ((flags & (RT_FLAG_NO_SKY | RT_FLAG_SKY_ONLY)) == (RT_FLAG_NO_SKY | RT_FLAG_SKY_ONLY) || ((flags & (RT_FLAG_NO_SKY | RT_FLAG_SKY_ONLY)) && (flags & RT_FLAG_COUNT)))
    return fail(RT_ERR_INVALID, "RT_FLAG_NO_SKY and RT_FLAG_SKY_ONLY exclude each other and RT_FLAG_COUNT");
  if ((flags & RT_FLAG_COMPACT) && ((flags & (RT_FLAG_RGB24 | RT_FLAG_NO_SKY | RT_FLAG_COUNT | RT_FLAG_STRICT_FP)) != (RT_FLAG_RGB24 | RT_FLAG_NO_SKY) || d_frames))
    return fail(RT_ERR_INVALID, "RT_FLAG_COMPACT goes with RT_FLAG_RGB24 | RT_FLAG_NO_SKY into a band (no counting, no strict kernel, no scatter)");
  int rc = ensure_device(s->device);
  if (rc) return rc;
  device_state &D = G.dev[s->device];
  hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : D.stream;
  const auto t_begin = std::chrono::steady_clock::now();
  {
    // which streams the scene's launches run on (rt_scene_set_camera, dispatch_order), and: behind the last write of the camera block
    std::lock_guard<std::mutex> lk(s->launch_mu);
    if ((rc = behind_the_camera(s, stream))) return rc;
    if (s->any_launch && s->last_stream != stream) s->several_streams = true;
    s->last_stream = stream; s->any_launch = true; s->launched_since_move = true;
  }

  const rt_scene_header &hd = s->hd;
  const uint32_t ss = ss_override ? ss_override : hd.supersample;
  if (ss > 2u) {
    // (3x3 / 4x4 supersampling filters whole blocks of samples: a NO_SKY call stores every pixel, a SKY_ONLY call none)
    if (flags & RT_FLAG_SKY_ONLY) { if (stats) memset(stats, 0, sizeof *stats); return RT_OK; }
    return render_supersampled(s, ss, w, h, tiles, n_frames, d_out, frame_stride_bytes, d_frames, stream, flags & ~(uint32_t)RT_FLAG_NO_SKY, stats);
  }
  const bool ss2 = ss == 2u;
  const bool count = (flags & RT_FLAG_COUNT) != 0;
  // Which kernel.  The product (FMA) kernel unless the caller asks for the strict one - or the scene itself sits on an exact
  // coincidence whose outcome in the reference is decided by the last bit of its own arithmetic (s->needs_strict, see
  // rt_scene_upload): only the operation-for-operation kernel reproduces those.
  const bool no_fixup = RT_TEST_ENV("RT_NO_FIXUP") != nullptr;                 // test build: the product kernel's own pixels everywhere (read per call)
  const bool strict_main = (flags & RT_FLAG_STRICT_FP) != 0 || (s->needs_strict && !no_fixup);
  const bool compact = (flags & RT_FLAG_COMPACT) != 0;
  if (compact && (strict_main || ss > 2u)) return fail(RT_ERR_UNSUPPORTED, "RT_FLAG_COMPACT: this scene is rendered by the strict kernel (or supersampled 3x3 / 4x4), which knows no launch table: send plain bands");
  const uint8_t *db = (const uint8_t *)s->d_blob;
  static const bool no_grid = RT_TEST_ENV("RT_NO_SHADOW_GRID") != nullptr;     // A/B switches (test build only)
  static const bool no_bounce = RT_TEST_ENV("RT_NO_BOUNCE_TABLE") != nullptr;
  // RT_LDS_PAD (bytes): occupancy experiments only — extra dynamic LDS per workgroup caps the workgroups per CU
  static const unsigned lds_pad = RT_TEST_ENV("RT_LDS_PAD") ? (unsigned)atoi(RT_TEST_ENV("RT_LDS_PAD")) : 0u;
  rt_launch L;
  memset(&L, 0, sizeof L);
  // the part of the launch record that depends on the kernel: ordering B (enclosing sphere last, outside the loops) and the
  // shadow grids / bounce table for the product kernel; the strict kernel and the counting variant walk the scene in its own
  // order so that they stay literal / count what the reference counts
  auto bind_kernel = [&](rt_launch &K, bool strict) {
    const bool plain = strict || count;
    const bool order_b = s->d_objects_b && !plain;
    const rt_geom *gt = s->d_geom + (order_b ? (size_t)hd.n_objects * (1 + hd.n_lights) : 0);      // [plain N | anchored at light k: NL x N]
    const rt_geom *gc = (const rt_geom *)cam_block(s) + (order_b ? 2 * (size_t)hd.n_objects : 0);     // this camera's block: [anchored at the camera N | cull rectangles N]
    K.objects = order_b ? s->d_objects_b : (const rt_sphere *)(db + hd.objects_offset);
    K.geom = gt;
    K.geom_cam = gc;
    K.cull = gc + hd.n_objects;
    K.geom_light = gt + hd.n_objects;
    K.lds_image = lds_image_of(s) + (order_b ? s->lds_image_bytes : 0);
    K.shadow_grid = (!plain && !no_grid) ? s->d_shadow_grid : nullptr;
    K.bounce_table = (!plain && !no_bounce) ? s->d_bounce_table : nullptr;
    K.n_loop = order_b ? hd.n_objects - 1 : hd.n_objects;
    K.enclosing = order_b ? hd.n_objects - 1 : ~0u;
    K.enclosing_flat = (order_b && s->enclosing_flat) ? 1u : 0u;
    K.cull_in_lds = s->cull_in_lds ? 1u : 0u;
    K.sky_fast = (K.enclosing_flat && s->sky_const) ? 1u : 0u;
    for (int c = 0; c < 3; c++) K.sky_rgb[c] = s->sky_rgb[c];
    memcpy(K.miss_color, hd.miss_color, sizeof K.miss_color);
    if (!plain && s->enclosing == ~0u && hd.segs > 0) {
      // no enclosing sphere at all: a primary ray that meets nothing is the miss colour (main.js:231), a constant as well
      K.sky_fast = 1u;
      memcpy(K.sky_rgb, hd.miss_color, sizeof K.sky_rgb);
    } else if (K.sky_fast) {
      // A flat sky of constant colour needs no hit record at all: "met nothing in the loops" IS "met the sky", whose pixel term
      // is the constant the host evaluated - so for the product kernel that constant takes the place of the miss colour
      // (main.js:231 is unreachable in such a scene: the sky encloses every ray) and the sphere leaves the kernel's view.
      // Lanes that end on the sky then take the two-instruction miss branch, at every level of the ray tree.
      K.enclosing = ~0u;
      memcpy(K.miss_color, s->sky_rgb, sizeof K.miss_color);
    }
  };
  bool four_waves = false;          // (rt_launch::four_waves: set below, once the launch knows whether it is a camera's first frame)
  auto lds_for = [&](bool strict) {
    // (the reflection-only many-sphere variants keep only the fold state in LDS: rt_kernel.hip, IMAGE_IN_LDS - and run one-wave
    // workgroups, rt_device.h, unless they store through the peer-store path)
    if (!strict && !count && !s->cull_in_lds && !s->refract)
      return lds_pad + 10u * (rt_one_wave_workgroups(false, count != 0, s->refract, (d_frames != nullptr && !ss2) || four_waves) ? 64u : RT_WG_THREADS) * 8u;
    if (!strict && rt_one_wave_workgroups(false, count != 0, s->refract, (d_frames != nullptr && !ss2) || four_waves)) return s->lds_bytes + lds_pad + 10u * 64u * 8u;
    return s->lds_bytes + lds_pad + (!strict ? (s->refract ? 13u : 10u) * RT_WG_THREADS * 8u     // + the product kernels' fold state
                                             : RT_WG_THREADS * 8u);                              //   (strict: one slot, the scatter store's tile)
  };
  bind_kernel(L, strict_main);
  // the boundary test of the product kernel's samplers (rt_device.h)
  L.flag_tol = s->flag_tol;
  bool test_marks = false;               // test build: a switch that changes what is marked or re-traced - nothing is cached then
#ifdef RT_TESTING
  if (const char *fs = getenv("RT_FLAG_SCALE")) { L.flag_tol *= atof(fs); test_marks = true; }        // a wider boundary band, to exercise the second launch
  if (getenv("RT_MARK_ALL") || getenv("RT_EXACT_ALL") || no_fixup) test_marks = true;
#endif
  L.mark_flags = (RT_TEST_ENV("RT_MARK_ALL") ? RT_MARK_ALL : 0u) | (no_fixup ? RT_MARK_NEVER : 0u) | (RT_TEST_ENV("RT_TEST_MARK_STRIPES") ? RT_MARK_ZERO : 0u) | (s->unit_weights ? RT_MARK_WEIGHT : 0u);
  L.marks_cap = RT_MARKS_CAP;
  L.textures = s->d_texdesc;
  L.texel_base = db;
  L.out = (uint32_t *)d_out;
  L.counters = D.d_counters;
  memcpy(L.cam_origin, hd.cam_origin, 12 * sizeof(double));   // origin, axisX, axisY, axisZ are contiguous
  // projection constants (main.js:102-105) of the sample grid, in binary64 on the host
  const double sw = ss2 ? 2.0 * w : (double)w, sh = ss2 ? 2.0 * h : (double)h;
  const double projA = hd.fov_deg * M_PI / 180.0;
  L.proj_w = sw / 2.0; L.proj_h = sh / 2.0; L.proj_d = L.proj_w / tan(projA / 2.0);
  L.epsilon = hd.epsilon; L.light_intensity = hd.light_intensity;
  L.n_objects = hd.n_objects; L.n_lights = hd.n_lights; L.segs = hd.segs;
  L.w = w; L.h = h;
  L.tile_rows = tiles->tile_rows; L.tile_first = tiles->tile_first; L.tile_stride = tiles->tile_stride; L.n_tiles = tiles->n_tiles;
  L.tiles_x = (w + RT_TILE_W - 1) / RT_TILE_W;
  memcpy(L.lights, s->lights, sizeof L.lights);

  const uint32_t rows_per_wg = ss2 ? 2u : RT_TILE_H;
  L.rb_per_tile = (tiles->tile_rows + rows_per_wg - 1) / rows_per_wg;
  L.rb_shift = ~0u;
  for (uint32_t b = 0; b < 31; b++) if (L.rb_per_tile == (1u << b)) L.rb_shift = b;
  if ((uint64_t)tiles->n_tiles * L.rb_per_tile > 65535u) return fail(RT_ERR_INVALID, "%u tiles x %u row blocks exceed the grid's y limit (65535)", tiles->n_tiles, L.rb_per_tile);
  L.n_frames = n_frames;
  L.frame_stride = frame_stride_bytes / 4u;
  L.rgb24 = (flags & RT_FLAG_RGB24) ? 1u : 0u;
  L.compact = compact ? 1u : 0u;
  L.scatter = d_frames ? 1u : 0u;
  if (d_frames) for (uint32_t f = 0; f < n_frames; f++) L.out_frames[f] = (uint32_t *)d_frames[f];
  for (int c = 0; c < 3; c++) L.cam_axis_sum[c] = hd.cam_axis_x[c] + hd.cam_axis_y[c] + hd.cam_axis_z[c];
  L.ray_bias[0] = 0.5 - L.proj_w; L.ray_bias[1] = L.proj_h - 0.5; L.ray_bias[2] = L.cam_axis_sum[2] * L.proj_d;
  if (count) HIP_TRY(hipMemsetAsync(D.d_counters, 0, 3 * sizeof(unsigned long long), stream));
#ifdef RT_TESTING
  L.probe = g_probe.d_buf; L.probe_x = g_probe.x; L.probe_y = g_probe.y;
#endif

  // timing events of a stats call; released on every way out of this function
  struct event_pair {
    hipEvent_t a = nullptr, b = nullptr;
    ~event_pair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  hipEvent_t &ev0 = ev.a, &ev1 = ev.b;
  if (stats) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
  static const bool no_order = RT_TEST_ENV("RT_NO_DISPATCH_ORDER") != nullptr;    // A/B switch (test build): the grid's own order
  struct marks_guard {                                   // a per-call mark list is released on every way out, after the stream has drained
    void *p = nullptr; hipStream_t st = nullptr;
    ~marks_guard() { if (p) { (void)hipStreamSynchronize(st); (void)hipFree(p); } }
  } temp_marks;
  int err = 0;
  uint32_t marks_read_slot = 0;
  const uint32_t *marks_read = nullptr;                 // stats: where this launch's mark count can be read afterwards
  uint64_t centre_items = 0;
  bool retraced_all = false, overflowed_strict = false;
  if (strict_main && (flags & RT_FLAG_SKY_ONLY)) {
    // (the strict kernels know no sky blocks: the RT_FLAG_NO_SKY calls of such a launch store every pixel, this one none)
  } else if (strict_main) {
    std::lock_guard<std::mutex> lk(s->launch_mu);
    {
      size_t per_lane = 0;
      if ((rc = kernel_scratch(true, false, s->refract, count, ss2, 0, &per_lane))) return rc;
      if ((rc = scratch_guard(D, stream, per_lane, (uint64_t)L.tiles_x * L.n_tiles * L.rb_per_tile * n_frames * (RT_WG_THREADS / 64u), "the strict trace kernel"))) return rc;
    }
    err = rt_launch_trace_strict(&L, s->refract, count, ss2, lds_for(true), stream);
  } else {
    // ---- the product launch: its table (found, or built on the GPU for this camera), the trace, and - unless this frame is KNOWN
    //      to have nothing for it - the list-driven strict launch behind it; one step for the threads of this process ----
    std::lock_guard<std::mutex> lk(s->launch_mu);
    // workgroups no sphere can show in are marked in the table and store the background constant without tracing (rt_block.h);
    // the counting variant traces them like any other (its counters are what the caller wants)
    static const bool no_sky_tiles = RT_TEST_ENV("RT_NO_SKY_TILES") != nullptr;   // A/B switch (test build)
    const bool mark_sky = L.sky_fast != 0u && !count && !no_sky_tiles;
    // per block and light, the spheres that can shadow a primary hit of the block at all, and the block's primary candidates;
    // needs every lit primary hit to lie on a loop sphere, i.e. no enclosing sphere or a flat one
    static const bool no_shadow_masks = RT_TEST_ENV("RT_NO_SHADOW_MASKS") != nullptr;   // A/B switch (test build)
    // Many spheres (more than 16 in the loops: a light's set is "empty or not"): the masks cost the table build ten times what
    // they save ONE frame (64 spheres at 3840x2160: 0.33 ms of a 0.36 ms build against 0.013 ms of a 0.11 ms trace; few spheres:
    // 0.011 against 0.020: profiles/r03_ab_log.md section 3) - the first frame from a camera is rendered from a table without
    // them, a camera that stays gets the full table with its second frame.  (The picture is the same either way: masks only prune.)
    // ("second frame" is counted per (frame size, tile set, sky part): the bands of one rt_render frame and the owner's sky fill
    // are several launches of ONE frame, and all of them are first launches from a new camera)
    const uint32_t sky_part = (flags & RT_FLAG_NO_SKY) ? 1u : ((flags & RT_FLAG_SKY_ONLY) ? 2u : 0u);
    uint32_t uses_before = 0;
    if (!count) {
      rt_scene_dev::camera_use *cu = nullptr;
      for (rt_scene_dev::camera_use &c : s->camera_uses)
        if (c.w == w && c.h == h && c.ss == (ss2 ? 2u : 1u) && c.tile_rows == tiles->tile_rows && c.tile_first == tiles->tile_first && c.tile_stride == tiles->tile_stride &&
            c.n_tiles == tiles->n_tiles && c.part == sky_part) { cu = &c; break; }
      if (!cu) {
        if (s->camera_uses.size() >= 64u) s->camera_uses.erase(s->camera_uses.begin());
        s->camera_uses.push_back(rt_scene_dev::camera_use{w, h, ss2 ? 2u : 1u, tiles->tile_rows, tiles->tile_first, tiles->tile_stride, tiles->n_tiles, sky_part, 0u, 0u});
        cu = &s->camera_uses.back();
      }
      if (cu->cam_gen != s->cam_gen) { cu->cam_gen = s->cam_gen; cu->uses = 0u; }
      uses_before = cu->uses++;
      s->renders_with_camera++;
    }
    const bool masks_pay = L.n_loop <= 16u || uses_before >= 1u;
    // the first frame from a camera that has moved: a caller that moves the camera every frame has the next camera's table built
    // beside this launch (rt_scene_set_camera), and that build needs the trace's workgroups to be as wide as its own (rt_launch::four_waves)
    four_waves = !count && uses_before == 0u && s->cam_gen != 0u;
    L.four_waves = four_waves ? 1u : 0u;
    // (a table of nothing but sky runs is read by workgroups that store a constant: neither masks nor candidates)
    const bool shadow_masks = !count && !no_shadow_masks && masks_pay && sky_part != 2u && (s->enclosing == ~0u || s->enclosing_flat);
    const bool name_candidates = !count && !no_shadow_masks && sky_part != 2u;
    const int oi = dispatch_order(s, w, h, ss2 ? 2u : 1u, tiles, L.tiles_x, L.rb_per_tile, L.proj_w, L.proj_h, L.proj_d, compact ? 2 : ((!count && !no_order) ? 1 : 0), mark_sky,
                                  shadow_masks, name_candidates, sky_part, stream);
    if (oi < 0) return RT_ERR_DEVICE;
    rt_scene_dev::order_entry &oe = s->orders[oi];
    L.order = oe.Tb[s->cam_gen & 1u].entries;
    // one workgroup per table entry (runs of sky blocks share one).  How many there are is known on the device; until the build's
    // count has reached the host, one workgroup per BLOCK is launched: those behind the last entry read a zero slot and leave
    uint32_t n_known = known_value(oe.known, s->cam_gen);
    // A table that rt_scene_set_camera is rebuilding on the side stream - beside the previous frame's trace - publishes its count
    // while that trace is still running.  A caller that issues frames back to back arrives here earlier than that: it is given a
    // short, BOUNDED wait for the word (it is ahead of the GPU anyway, and stays one frame ahead: the trace in flight has tens of
    // microseconds left when the word arrives); one workgroup per block costs a 4K frame 80 us instead of 68.  A caller that comes
    // later (a frame per display refresh) finds the word there; a word that does not come in time: one workgroup per block.
    if (!n_known && oe.built_on == s->side && oe.cam_gen == s->cam_gen) {
      // (the bound grows with the table: a 4K frame's build takes ~50 us beside a trace, an 8K frame's four times that)
      static const long wait_env = RT_TEST_ENV("RT_COUNT_WAIT_US") ? atol(RT_TEST_ENV("RT_COUNT_WAIT_US")) : -1;       // A/B switch (test build)
      const long wait_us = wait_env >= 0 ? wait_env : 100 + (long)(oe.n_blocks / 256u);
      const auto t0 = std::chrono::steady_clock::now();
      while (!n_known && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < wait_us) {
        __builtin_ia32_pause();
        n_known = known_value(oe.known, s->cam_gen);
      }
    }
    L.order_n8 = (oe.n_blocks + 7u) / 8u;
    L.grid_x = n_known ? n_known - 1u : oe.n_blocks;
    L.grid_y = 1u;
    rt_scene_dev::mark_state *ms = nullptr;
    for (rt_scene_dev::mark_state &m : s->mark_states) if (m.order_index == (uint32_t)oi && m.stream == stream) ms = &m;
    rt_scene_dev::mark_state temp_state = {~0u, stream, nullptr, nullptr, 0u};
    if (!ms) {
      // first launch of this (table, stream) pair: a list of its own (beyond RT_KNOWN_WORDS pairs: one per call, nothing cached)
      const size_t bytes = 16u + (size_t)RT_MARKS_CAP * 8u;
      uint32_t *d = nullptr;
      hipError_t e = hipMalloc((void **)&d, bytes);
      if (e == hipSuccess) e = hipMemsetAsync(d, 0, 16u, stream);
      if (e != hipSuccess) { if (d) (void)hipFree(d); return fail(RT_ERR_DEVICE, "mark list: %s", hipGetErrorString(e)); }
      if (s->mark_states.size() >= RT_KNOWN_WORDS) { temp_marks.p = d; temp_marks.st = stream; temp_state.d_marks = d; ms = &temp_state; }
      else {
        s->mark_states.push_back(rt_scene_dev::mark_state{(uint32_t)oi, stream, d, known_word(s, s->mark_states.size()), 0u});
        ms = &s->mark_states.back();
      }
    }
    L.marks = ms->d_marks; L.marks_slot = ms->slot;
    const uint32_t known = test_marks ? 0u : known_value(ms->h_known, s->cam_gen);        // 0: not known (yet); else the frame's mark count + 1
    // A frame KNOWN to mark more samples than the list holds (a legal scene can: every hit of a sphere whose sampler coordinate is
    // an exact integer everywhere) would be traced twice in full, product kernel then rt_retrace over every sample: the strict
    // kernel renders it once instead, the same bytes (the count stays known: nothing republishes it for this camera).
    const bool overflow_known = known != 0u && known - 1u > RT_MARKS_CAP && !count && !compact;
    if (overflow_known) {
      if (!(flags & RT_FLAG_SKY_ONLY)) {              // (as every strict launch: a NO_SKY call stores every pixel, a SKY_ONLY call none)
        rt_launch S = L;
        S.order = nullptr; S.grid_x = S.grid_y = 0u;
        bind_kernel(S, true);
        size_t per_lane = 0;
        if ((rc = kernel_scratch(true, false, s->refract, 0, ss2, 0, &per_lane))) return rc;
        if ((rc = scratch_guard(D, stream, per_lane, (uint64_t)L.tiles_x * L.n_tiles * L.rb_per_tile * n_frames * (RT_WG_THREADS / 64u), "the strict trace kernel"))) return rc;
        err = rt_launch_trace_strict(&S, s->refract, 0, ss2, lds_for(true), stream);
        overflowed_strict = true;
      }
    } else {
    {
      size_t per_lane = 0;
      if ((rc = kernel_scratch(false, false, s->refract, count, ss2, !count && !L.cull_in_lds, &per_lane, rt_one_wave_workgroups(false, count != 0, s->refract, (L.scatter != 0u && !ss2) || four_waves)))) return rc;
      if ((rc = scratch_guard(D, stream, per_lane, (uint64_t)L.grid_x * n_frames * (RT_WG_THREADS / 64u), "the trace kernel"))) return rc;
    }
#ifdef RT_WAVE_LOG
    // measurement build: RT_WAVE_LOG_FILE=<path> - every wave's entry / exit time and place of THIS launch, written after it has finished
    unsigned long long *d_wave_log = nullptr;
    const size_t wave_log_words = (size_t)((L.grid_x + 7u) / 8u * 8u) * n_frames * (RT_WG_THREADS / 64u) * 4u;
    if (getenv("RT_WAVE_LOG_FILE")) {
      if (hipMalloc((void **)&d_wave_log, wave_log_words * 8u) == hipSuccess) (void)hipMemsetAsync(d_wave_log, 0, wave_log_words * 8u, stream);
      L.wave_log = d_wave_log;
    }
#endif
    err = rt_launch_trace_fast(&L, s->refract, count, ss2, lds_for(false), stream);
#ifdef RT_WAVE_LOG
    if (d_wave_log) {
      std::vector<unsigned long long> hostlog(wave_log_words);
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpy(hostlog.data(), d_wave_log, wave_log_words * 8u, hipMemcpyDeviceToHost);
      (void)hipFree(d_wave_log);
      if (FILE *fp = fopen(getenv("RT_WAVE_LOG_FILE"), "wb")) { fwrite(hostlog.data(), 8u, wave_log_words, fp); fclose(fp); }
      L.wave_log = nullptr;
    }
#endif
    // Centre row / centre column of a sample grid with an ODD number of rows / columns (supersample 2 makes it even).  The primary
    // rays there have a direction component that is EXACTLY zero (main.js:186: x - w/2 + 0.5 == 0), so they - and every ray they
    // spawn that stays in that plane - live in a coordinate plane through the camera, and a sphere centred on that plane (the
    // reference's own scene has several) is met with a normal component of exactly 0: u or v lands exactly ON a texel / checker
    // boundary (main.js:127-130, 344-347), and on which side the reference falls is decided by whether ITS OWN rounding noise
    // (e.g. main.js:257-259 at refract_index 1, where q is 0 or 1e-16 depending on the last bit of cosi) pushed the ray off the
    // plane.  No arithmetic but the reference's own reproduces such coin flips: rt_retrace traces those samples too.
    rt_launch F = L;
    F.order = nullptr; F.grid_x = F.grid_y = 0u;
    F.centre_row = F.centre_col = ~0u;
    if (!ss2 && (h & 1u)) {
      const uint32_t crow = (h - 1u) / 2u, tc = crow / tiles->tile_rows;
      if (tc >= tiles->tile_first && (tc - tiles->tile_first) % tiles->tile_stride == 0 && (tc - tiles->tile_first) / tiles->tile_stride < tiles->n_tiles) { F.centre_row = crow; centre_items += (uint64_t)w * n_frames; }
    }
    if (!ss2 && (w & 1u)) { F.centre_col = (w - 1u) / 2u; centre_items += (uint64_t)tiles->n_tiles * tiles->tile_rows * n_frames; }
    const bool retrace_all = RT_TEST_ENV("RT_EXACT_ALL") != nullptr && !no_fixup;
    const bool need = !no_fixup && !(flags & RT_FLAG_SKY_ONLY) && (known != 1u || centre_items != 0 || retrace_all);     // (a sky-only launch traces nothing; the centre lines belong to the calls that trace)
    if (err == 0 && need) {
      bind_kernel(F, true);                             // the scene in its own order, every sphere in the loops, the reference's own miss colour
      if (compact) {                                    // where a sample's block sits in the compact band: from the table's own arrays
        const rt_table_dev &T = oe.Tb[s->cam_gen & 1u];
        F.tb_item = T.item; F.tb_rank_in_row = T.rank_in_row; F.tb_row_hist = T.row_hist; F.tb_bin_start = T.bin_start; F.tb_bins = oe.cost_bins;
      }
      F.marks_known = test_marks ? nullptr : (unsigned long long *)ms->h_known;
      F.known_tag = (uint32_t)s->cam_gen;
      F.retrace_all = retrace_all ? 1u : 0u;
      retraced_all = retrace_all;
      // The grid.  Count known: its items and the centre lines.  Not known yet (the first frame from a camera): the list may hold up
      // to RT_MARKS_CAP items or have overflowed - 256 workgroups (idle ones leave at once) walk an overflowed 3840x2160 frame at
      // ~130 samples per lane, once; from the next frame on the count is known (and an overflow takes the strict kernel above).
      uint64_t n_wg = (((known ? known - 1u : 0u) + centre_items) * (ss2 ? 4u : 1u) + RT_WG_THREADS - 1) / RT_WG_THREADS + 2u;      // (supersample 2: a lane per sample)
      if (!known && n_wg < 256u) n_wg = 256u;
      if (retrace_all) n_wg = ((uint64_t)tiles->n_tiles * tiles->tile_rows * w * n_frames + RT_WG_THREADS - 1) / RT_WG_THREADS;
      if (n_wg > 8192u) n_wg = 8192u;
      {
        size_t per_lane = 0;
        if ((rc = kernel_scratch(true, true, s->refract, 0, ss2, 0, &per_lane))) return rc;
        if ((rc = scratch_guard(D, stream, per_lane, n_wg * (RT_WG_THREADS / 64u), "the list-driven strict launch (rt_retrace)"))) return rc;
      }
      err = rt_launch_retrace(&F, s->refract, ss2, (unsigned)n_wg, stream);
      marks_read = ms->d_marks; marks_read_slot = ms->slot;
      ms->slot ^= 1u;                                   // rt_retrace cleared the other counter: the next launch's
    }
    }
  }
  if (err != 0) return fail(RT_ERR_DEVICE, "kernel launch: %s", hipGetErrorString((hipError_t)err));
  if (stats) {
    HIP_TRY(hipEventRecord(ev1, stream));
    HIP_TRY(hipEventSynchronize(ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    memset(stats, 0, sizeof *stats);
    stats->kernel_ms = ms;
    uint64_t px = 0;
    for (uint32_t i = 0; i < tiles->n_tiles; i++) {
      const uint64_t r0 = (uint64_t)(tiles->tile_first + (uint64_t)i * tiles->tile_stride) * tiles->tile_rows;
      if (r0 < h) px += ((r0 + tiles->tile_rows <= h) ? tiles->tile_rows : (h - r0)) * (uint64_t)w;
    }
    stats->pixels = px * n_frames;
    if (count) {
      unsigned long long c[3];
      HIP_TRY(hipMemcpy(c, D.d_counters, sizeof c, hipMemcpyDeviceToHost));
      stats->rays = c[0]; stats->shadow_rays = c[1]; stats->sphere_tests = c[2];
    }
    // samples the second launch traced again: the marked ones (read back from the list's counter) and the odd grid's centre lines
    if (overflowed_strict) stats->exact_samples = stats->pixels;      // the strict kernel rendered the call
    if (marks_read) {
      uint32_t n_marked = 0;
      HIP_TRY(hipMemcpy(&n_marked, marks_read + marks_read_slot, sizeof n_marked, hipMemcpyDeviceToHost));
      stats->exact_samples = (n_marked > RT_MARKS_CAP || retraced_all) ? stats->pixels : n_marked + centre_items;   // (list overflow / test build: every pixel of the call)
    }
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return RT_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------ compact bands (RT_FLAG_COMPACT)
namespace {
// the launch table a compact launch over `tiles` uses (found, or built now on `stream`), under launch_mu; -1: rt_last_error
int compact_table(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, hipStream_t stream, uint32_t *rows_per_wg_out) {
  if (!s || !tiles) { fail(RT_ERR_INVALID, "NULL scene or tiles"); return -1; }
  if (w == 0 || h == 0 || w > 65536 || h > 65536 || (w & 3u) || tiles->tile_rows == 0 || tiles->tile_stride == 0 || tiles->n_tiles == 0) { fail(RT_ERR_INVALID, "compact band: bad frame size or tile set (w must be a multiple of 4)"); return -1; }
  const uint32_t ss = s->hd.supersample;
  if (ss > 2u || s->needs_strict) { fail(RT_ERR_UNSUPPORTED, "compact bands: this scene is rendered by the strict kernel (or supersampled 3x3 / 4x4): send plain bands"); return -1; }
  const bool ss2 = ss == 2u;
  const uint32_t rows_per_wg = ss2 ? 2u : RT_TILE_H;
  const uint32_t tiles_x = (w + RT_TILE_W - 1) / RT_TILE_W, rb_per_tile = (tiles->tile_rows + rows_per_wg - 1) / rows_per_wg;
  const double sw = ss2 ? 2.0 * w : (double)w, sh = ss2 ? 2.0 * h : (double)h;                    // (render_batch_impl's expressions: the same bits)
  const double projA = s->hd.fov_deg * M_PI / 180.0, pw = sw / 2.0, ph = sh / 2.0, pd = pw / tan(projA / 2.0);
  if (behind_the_camera(s, stream)) return -1;
  // the product launch's own choices (render_batch_impl): sky marks for a constant background; candidates; masks do not matter for the
  // ORDER of the blocks - a table with and one without them list the same blocks at the same places
  const bool sky_fast = (s->enclosing_flat && s->sky_const) || (s->enclosing == ~0u && s->hd.segs > 0);
  const uint32_t n_loop = s->hd.n_objects - (s->enclosing != ~0u ? 1u : 0u);
  const bool masks = (n_loop <= 16u) && (s->enclosing == ~0u || s->enclosing_flat);
  *rows_per_wg_out = rows_per_wg;
  return dispatch_order(s, w, h, ss, tiles, tiles_x, rb_per_tile, pw, ph, pd, 2, sky_fast, masks, true, 1u, stream);
}

struct rt_expand_launch { const uint32_t *entries; uint32_t n8, n_blocks, w, rows_per_wg; const uint8_t *src; uint32_t *dst; };
// one workgroup of 256 per entry: the block's 32 x RH pixels (RGB24, row by row) to their place in the RGBA8 frame
__global__ void __launch_bounds__(256) rt_compact_expand_kernel(const rt_expand_launch E) {
  const uint32_t b = blockIdx.x;
  const uint4 e = ((const uint4 *)E.entries)[(size_t)(b & 7u) * E.n8 + (b >> 3)];
  const uint32_t tile_x = e.x & 2047u, rows_valid = (e.x >> 11) & 15u, frow0 = e.x >> 15;
  if (rows_valid == 0u || (e.y >> 31)) return;                     // (no entry, or a sky run: not part of a compact band)
  const uint32_t r = threadIdx.x >> 5, i = threadIdx.x & 31u, px = tile_x * RT_TILE_W + i;
  if (r >= rows_valid || r >= E.rows_per_wg || px >= E.w) return;
  const uint8_t *p = E.src + (size_t)b * (RT_TILE_W * 3u * E.rows_per_wg) + ((size_t)r * RT_TILE_W + i) * 3u;
  E.dst[(size_t)(frow0 + r) * E.w + px] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xff000000u;
}
}  // namespace

extern "C" int rt_compact_count(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, void *hip_stream, uint32_t *n_blocks, uint32_t *block_bytes) {
  if (!n_blocks || !block_bytes) return fail(RT_ERR_INVALID, "rt_compact_count: NULL argument");
  int rc = s ? ensure_device(s->device) : RT_ERR_INVALID;
  if (rc) return rc == RT_ERR_INVALID ? fail(RT_ERR_INVALID, "NULL scene") : rc;
  hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : G.dev[s->device].stream;
  uint32_t rows_per_wg = 0, header[4];
  const uint32_t *d_header = nullptr;
  {
    std::lock_guard<std::mutex> lk(s->launch_mu);
    const int oi = compact_table(s, w, h, tiles, stream, &rows_per_wg);
    if (oi < 0) return (strstr(rt_last_error(), "strict kernel") != nullptr) ? RT_ERR_UNSUPPORTED : RT_ERR_DEVICE;
    d_header = s->orders[oi].Tb[s->cam_gen & 1u].header;
  }
  HIP_TRY(hipStreamSynchronize(stream));
  HIP_TRY(hipMemcpy(header, d_header, sizeof header, hipMemcpyDeviceToHost));
  *n_blocks = header[2];                                             // the entries in front of the sky runs' class: a ranked table's non-sky blocks
  *block_bytes = RT_TILE_W * 3u * rows_per_wg;
  return RT_OK;
}

extern "C" int rt_compact_expand_device(rt_scene_dev *s, uint32_t w, uint32_t h, const rt_tiles *tiles, const void *d_compact, void *d_frame, void *hip_stream) {
  if (!d_compact || !d_frame || ((uintptr_t)d_frame & 3u)) return fail(RT_ERR_INVALID, "rt_compact_expand_device: NULL or unaligned buffer");
  int rc = s ? ensure_device(s->device) : RT_ERR_INVALID;
  if (rc) return rc == RT_ERR_INVALID ? fail(RT_ERR_INVALID, "NULL scene") : rc;
  hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : G.dev[s->device].stream;
  rt_expand_launch E;
  uint32_t grid = 0;
  {
    std::lock_guard<std::mutex> lk(s->launch_mu);
    uint32_t rows_per_wg = 0;
    const int oi = compact_table(s, w, h, tiles, stream, &rows_per_wg);
    if (oi < 0) return (strstr(rt_last_error(), "strict kernel") != nullptr) ? RT_ERR_UNSUPPORTED : RT_ERR_DEVICE;
    const rt_scene_dev::order_entry &oe = s->orders[oi];
    E.entries = oe.Tb[s->cam_gen & 1u].entries; E.n8 = (oe.n_blocks + 7u) / 8u; E.n_blocks = oe.n_blocks; E.w = w; E.rows_per_wg = rows_per_wg;
    E.src = (const uint8_t *)d_compact; E.dst = (uint32_t *)d_frame;
    const uint32_t n_known = known_value(oe.known, s->cam_gen);
    grid = n_known ? n_known - 1u : oe.n_blocks;                     // (workgroups behind the last entry read a zero slot and leave)
    if (s->any_launch && s->last_stream != stream) s->several_streams = true;
    s->last_stream = stream; s->any_launch = true; s->launched_since_move = true;
  }
  if (grid) hipLaunchKernelGGL(rt_compact_expand_kernel, dim3(grid), dim3(256), 0, stream, E);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

// ------------------------------------------------------------------------------------ sharing memory between the ranks of a node
extern "C" int rt_ipc_export(int device, const void *d_ptr, void *handle_out) {
  static_assert(sizeof(hipIpcMemHandle_t) == RT_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is expected to be 64 bytes");
  if (!d_ptr || !handle_out) return fail(RT_ERR_INVALID, "rt_ipc_export: NULL argument");
  int rc = ensure_device(device);
  if (rc) return rc;
  hipIpcMemHandle_t hnd;
  HIP_TRY(hipIpcGetMemHandle(&hnd, const_cast<void *>(d_ptr)));
  memcpy(handle_out, &hnd, sizeof hnd);
  return RT_OK;
}

extern "C" int rt_ipc_open(int device, const void *handle, void **d_ptr_out) {
  if (!handle || !d_ptr_out) return fail(RT_ERR_INVALID, "rt_ipc_open: NULL argument");
  int rc = ensure_device(device);
  if (rc) return rc;
  hipIpcMemHandle_t hnd;
  memcpy(&hnd, handle, sizeof hnd);
  void *p = nullptr;
  HIP_TRY(hipIpcOpenMemHandle(&p, hnd, hipIpcMemLazyEnablePeerAccess));
  *d_ptr_out = p;
  return RT_OK;
}

extern "C" int rt_ipc_close(int device, void *d_ptr) {
  if (!d_ptr) return RT_OK;
  int rc = ensure_device(device);
  if (rc) return rc;
  HIP_TRY(hipIpcCloseMemHandle(d_ptr));
  return RT_OK;
}

// ------------------------------------------------------------------------------------ memory helpers
// Pinned framebuffers are recycled: hipHostMalloc of a 33 MB frame costs ~15 ms, ten times the render.
// A freed buffer goes to a small pool (same-size reuse); at most 4 buffers / 1 GiB are kept.
namespace {
struct pinned_buf { void *p; size_t bytes; };
std::mutex g_pin_mu;
std::vector<pinned_buf> g_pin_live, g_pin_free;
}  // namespace

extern "C" void *rt_alloc_pinned(size_t bytes) {
  if (!bytes) bytes = 1;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pin_free.size(); i++)
      if (g_pin_free[i].bytes == bytes) {
        pinned_buf b = g_pin_free[i];
        g_pin_free.erase(g_pin_free.begin() + i);
        g_pin_live.push_back(b);
        return b.p;
      }
  }
  void *p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { fail(RT_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pin_live.push_back({p, bytes});
  return p;
}

extern "C" void rt_free_pinned(void *p) {
  if (!p) return;
  pinned_buf b = {p, 0};
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pin_live.size(); i++)
      if (g_pin_live[i].p == p) { b = g_pin_live[i]; g_pin_live.erase(g_pin_live.begin() + i); break; }
    size_t pooled = 0;
    for (const pinned_buf &f : g_pin_free) pooled += f.bytes;
    if (b.bytes && g_pin_free.size() < 4 && pooled + b.bytes <= ((size_t)1 << 30)) { g_pin_free.push_back(b); return; }
  }
  (void)hipHostFree(p);
}

extern "C" void *rt_alloc_device(int device, size_t bytes) {
  if (ensure_device(device)) return nullptr;
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) { fail(RT_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
  return p;
}
extern "C" void rt_free_device(int device, void *p) {
  if (!p || ensure_device(device)) return;
  (void)hipFree(p);
}
extern "C" int rt_copy_to_host(int device, void *dst, const void *src, size_t bytes) {
  int rc = ensure_device(device);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(G.dev[device].stream));
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return RT_OK;
}

extern "C" int rt_memset_device(int device, void *dst, int value, size_t bytes) {
  if (!dst) return fail(RT_ERR_INVALID, "rt_memset_device: NULL destination");
  int rc = ensure_device(device);
  if (rc) return rc;
  HIP_TRY(hipMemset(dst, value, bytes));
  HIP_TRY(hipDeviceSynchronize());
  return RT_OK;
}

// ------------------------------------------------------------------------------------ de-interleave
// src: for rank g, its tiles (g, g+R, g+2R, ...) stored contiguously, ranks `rank_stride` bytes apart.
// dst: the frame in row order.  One workgroup row per frame row (grid y), so the tile/rank arithmetic is
// wave-uniform scalar work done once; a work-item moves 16 bytes (T = uint4) or, for ragged widths, 4 (T = uint32_t).
template <typename T>
__global__ void __launch_bounds__(256) rt_deinterleave_kernel(const T *__restrict__ src, T *__restrict__ dst, uint32_t row_elems, uint32_t tile_rows,
                                                              uint32_t n_ranks, uint64_t rank_stride_elems) {
  const uint32_t row = blockIdx.y;
  const uint32_t tile = row / tile_rows, r = row - tile * tile_rows;
  const uint32_t rank = tile % n_ranks, local_tile = tile / n_ranks;
  const T *__restrict__ s = src + rank * rank_stride_elems + ((uint64_t)local_tile * tile_rows + r) * row_elems;
  T *__restrict__ d = dst + (uint64_t)row * row_elems;
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < row_elems; x += gridDim.x * blockDim.x) d[x] = s[x];
}

extern "C" int rt_deinterleave_device(int device, const void *d_src, void *d_dst, uint32_t w, uint32_t h, uint32_t tile_rows, uint32_t n_ranks,
                                      uint64_t rank_stride_bytes, void *hip_stream) {
  if (!d_src || !d_dst || !w || !h || !tile_rows || !n_ranks || (rank_stride_bytes & 3u) || h > 65535u * 16u) return fail(RT_ERR_INVALID, "bad de-interleave arguments");
  int rc = ensure_device(device);
  if (rc) return rc;
  hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : G.dev[device].stream;
  if (h > 65535u) return fail(RT_ERR_INVALID, "de-interleave: more than 65535 rows");
  const bool wide = (w % 4u == 0) && (rank_stride_bytes % 16u == 0) && (((uintptr_t)d_src | (uintptr_t)d_dst) % 16u == 0);
  const uint32_t row_elems = wide ? w / 4u : w;
  const dim3 grid((row_elems + 255u) / 256u, h), block(256);
  if (wide)
    hipLaunchKernelGGL(rt_deinterleave_kernel<uint4>, grid, block, 0, stream, (const uint4 *)d_src, (uint4 *)d_dst, row_elems, tile_rows, n_ranks,
                       rank_stride_bytes / 16u);
  else
    hipLaunchKernelGGL(rt_deinterleave_kernel<uint32_t>, grid, block, 0, stream, (const uint32_t *)d_src, (uint32_t *)d_dst, row_elems, tile_rows,
                       n_ranks, rank_stride_bytes / 4u);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

// RGB24 bands -> RGBA8 frame.  A work-item turns 3 source words (4 pixels x 3 bytes) into one uint4 (4 pixels x RGBA);
// w % 4 == 0, so rows of both sides start word-aligned.
__global__ void __launch_bounds__(256) rt_deinterleave_rgb24_kernel(const uint32_t *__restrict__ src, uint4 *__restrict__ dst, uint32_t row_quads,
                                                                    uint32_t tile_rows, uint32_t n_ranks, uint64_t rank_stride_words) {
  const uint32_t row = blockIdx.y;
  const uint32_t tile = row / tile_rows, r = row - tile * tile_rows;
  const uint32_t rank = tile % n_ranks, local_tile = tile / n_ranks;
  const uint32_t *__restrict__ s = src + rank * rank_stride_words + ((uint64_t)local_tile * tile_rows + r) * row_quads * 3u;
  uint4 *__restrict__ d = dst + (uint64_t)row * row_quads;
  for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < row_quads; x += gridDim.x * blockDim.x) {
    const uint32_t a = s[3u * x], b = s[3u * x + 1u], c = s[3u * x + 2u];
    uint4 o;
    o.x = a | 0xff000000u;
    o.y = (a >> 24) | (b << 8) | 0xff000000u;
    o.z = (b >> 16) | (c << 16) | 0xff000000u;
    o.w = (c >> 8) | 0xff000000u;
    d[x] = o;
  }
}

extern "C" int rt_deinterleave_rgb24_device(int device, const void *d_src, void *d_dst, uint32_t w, uint32_t h, uint32_t tile_rows, uint32_t n_ranks,
                                            uint64_t rank_stride_bytes, void *hip_stream) {
  if (!d_src || !d_dst || !w || !h || !tile_rows || !n_ranks || (rank_stride_bytes & 3u) || (w & 3u) || (((uintptr_t)d_src) & 3u) || (((uintptr_t)d_dst) & 15u))
    return fail(RT_ERR_INVALID, "bad RGB24 de-interleave arguments (w must be a multiple of 4, dst 16-byte aligned)");
  if (h > 65535u) return fail(RT_ERR_INVALID, "de-interleave: more than 65535 rows");
  int rc = ensure_device(device);
  if (rc) return rc;
  hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : G.dev[device].stream;
  const uint32_t row_quads = w / 4u;
  const dim3 grid((row_quads + 255u) / 256u, h), block(256);
  hipLaunchKernelGGL(rt_deinterleave_rgb24_kernel, grid, block, 0, stream, (const uint32_t *)d_src, (uint4 *)d_dst, row_quads, tile_rows, n_ranks,
                     rank_stride_bytes / 4u);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

// ------------------------------------------------------------------------------------ RCCL (lazy)
namespace {
typedef int (*nccl_comm_init_all_t)(void **comms, int ndev, const int *devlist);
typedef int (*nccl_gather_t)(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t stream);
typedef int (*nccl_group_t)(void);
typedef int (*nccl_comm_destroy_t)(void *comm);
typedef const char *(*nccl_errstr_t)(int);
struct { nccl_comm_init_all_t init_all; nccl_gather_t gather; nccl_group_t group_start, group_end; nccl_comm_destroy_t destroy; nccl_errstr_t errstr; } NCCL;
const int NCCL_UINT8 = 1;   // ncclUint8 (rccl.h ncclDataType_t)

int ensure_rccl(int ndev) {
  if (G.comms_ready) return RT_OK;
  if (!G.rccl) {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) if ((G.rccl = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!G.rccl) return fail(RT_ERR_DEVICE, "cannot load RCCL: %s", dlerror());
    NCCL.init_all = (nccl_comm_init_all_t)dlsym(G.rccl, "ncclCommInitAll");
    NCCL.gather = (nccl_gather_t)dlsym(G.rccl, "ncclGather");
    NCCL.group_start = (nccl_group_t)dlsym(G.rccl, "ncclGroupStart");
    NCCL.group_end = (nccl_group_t)dlsym(G.rccl, "ncclGroupEnd");
    NCCL.destroy = (nccl_comm_destroy_t)dlsym(G.rccl, "ncclCommDestroy");
    NCCL.errstr = (nccl_errstr_t)dlsym(G.rccl, "ncclGetErrorString");
    if (!NCCL.init_all || !NCCL.gather || !NCCL.group_start || !NCCL.group_end || !NCCL.destroy || !NCCL.errstr)
      return fail(RT_ERR_DEVICE, "RCCL is missing ncclCommInitAll/ncclGather/ncclGroup*");
  }
  int ids[16];
  for (int i = 0; i < ndev; i++) ids[i] = G.dev[i].hip_id;
  const int r = NCCL.init_all(G.comms, ndev, ids);
  if (r != 0) return fail(RT_ERR_DEVICE, "ncclCommInitAll: %s", NCCL.errstr(r));
  G.comms_ready = true;
  return RT_OK;
}

// rt_render's scene for `device`: the resident one if the blob is the same bytes, else a fresh upload that replaces it.
int scene_for(int device, const void *blob, size_t bytes, rt_scene_dev **out) {
  device_state &D = G.dev[device];
  if (D.cached_scene && D.cached_blob.size() == bytes && memcmp(D.cached_blob.data(), blob, bytes) == 0) { *out = D.cached_scene; return RT_OK; }
  // the same scene from another camera (an animation: lookAt per frame, main.js:92-100): the resident scene moves its camera
  if (D.cached_scene && D.cached_blob.size() == bytes) {
    const size_t c0 = offsetof(rt_scene_header, cam_origin), c1 = c0 + 12 * sizeof(double);
    const uint8_t *a = D.cached_blob.data(), *b = (const uint8_t *)blob;
    if (memcmp(a, b, c0) == 0 && memcmp(a + c1, b + c1, bytes - c1) == 0) {
      const rt_scene_header *nh = (const rt_scene_header *)blob;
      if (rt_scene_set_camera(D.cached_scene, nh->cam_origin, nh->cam_axis_x, nh->cam_axis_y, nh->cam_axis_z, nullptr) == RT_OK) {
        memcpy(D.cached_blob.data() + c0, b + c0, c1 - c0);
        *out = D.cached_scene;
        return RT_OK;
      }
    }
  }
  if (D.cached_scene) { rt_scene_free(D.cached_scene); D.cached_scene = nullptr; D.cached_blob.clear(); }
  rt_scene_dev *s = nullptr;
  const int rc = rt_scene_upload(device, blob, bytes, &s);
  if (rc) return rc;
  D.cached_scene = s;
  D.cached_blob.assign((const uint8_t *)blob, (const uint8_t *)blob + bytes);
  *out = s;
  return RT_OK;
}

// A device allocation must live on the device it was made for: every hipMalloc of the multi-GPU path is checked against
// hipPointerGetAttributes (a wrong current device would otherwise only show as a fault, or as silent xGMI traffic, on a real
// multi-GPU node - nothing a one-GPU box can catch).
int check_on_device(const void *p, const device_state &D, const char *what) {
  hipPointerAttribute_t attr;
  HIP_TRY(hipPointerGetAttributes(&attr, p));
  if (attr.device != D.hip_id) return fail(RT_ERR_DEVICE, "%s was allocated on HIP device %d, expected %d", what, attr.device, D.hip_id);
  return RT_OK;
}

// rt_render's per-device scratch frame, allocated with THAT device current (ensure_device does the hipSetDevice)
int ensure_frame(int device, size_t bytes) {
  int rc = ensure_device(device);
  if (rc) return rc;
  device_state &D = G.dev[device];
  if (D.frame_bytes >= bytes) return RT_OK;
  if (D.d_frame) (void)hipFree(D.d_frame);
  D.d_frame = nullptr; D.frame_bytes = 0;
  HIP_TRY(hipMalloc(&D.d_frame, bytes));
  D.frame_bytes = bytes;
  return check_on_device(D.d_frame, D, "rt_render's frame buffer");
}
}  // namespace

// ------------------------------------------------------------------------------------ render(width,height,scene)
namespace {
int render_to_host(const void *blob, size_t bytes, uint32_t w, uint32_t h, uint8_t *out_rgba, uint32_t flags, rt_stats *stats,
                   uint32_t want_bands, rt_band_callback on_band, void *user);
int g_last_plan = 0;      // how the last rt_render put its frame together: 0 one GPU (banded copy-out), 1 peer stores, 2 ncclGather (or its emulation), 3 one GPU storing into the pinned frame
int g_direct_stores = 1;  // one GPU: store straight into a pinned (mapped) caller buffer: 0 never, 1 frames below 8 MiB, 2 always (rt_render_options)
int g_copy_bands = 4;     // one GPU, copy-out plan: bands whose copy-out overlaps the next band's render (rt_render_options)
}  // namespace

#ifdef RT_TESTING
extern "C" int rt_test_last_plan(void) { return g_last_plan; }
#endif

extern "C" int rt_render_options(int direct_stores, uint32_t copy_bands) {
  if (copy_bands == 0 || copy_bands > 64u) return fail(RT_ERR_INVALID, "copy_bands %u not in 1..64", copy_bands);
  if (direct_stores < 0 || direct_stores > 2) return fail(RT_ERR_INVALID, "direct_stores %d not in 0..2", direct_stores);
  std::lock_guard<std::mutex> lk(G.mu);
  g_direct_stores = direct_stores;
  g_copy_bands = (int)copy_bands;
  return RT_OK;
}

extern "C" int rt_render(const void *blob, size_t bytes, uint32_t w, uint32_t h, uint8_t *out_rgba, uint32_t flags, rt_stats *stats) {
  return render_to_host(blob, bytes, w, h, out_rgba, flags, stats, 0u, nullptr, nullptr);
}

extern "C" int rt_render_progressive(const void *blob, size_t bytes, uint32_t w, uint32_t h, uint8_t *out_rgba, uint32_t n_bands,
                                     rt_band_callback on_band, void *user, uint32_t flags, rt_stats *stats) {
  if (n_bands == 0 || n_bands > 64u) return fail(RT_ERR_INVALID, "n_bands %u not in 1..64", n_bands);
  if (!on_band) return fail(RT_ERR_INVALID, "on_band is NULL");
  return render_to_host(blob, bytes, w, h, out_rgba, flags, stats, n_bands, on_band, user);
}

namespace {
int render_to_host(const void *blob, size_t bytes, uint32_t w, uint32_t h, uint8_t *out_rgba, uint32_t flags, rt_stats *stats,
                   uint32_t want_bands, rt_band_callback on_band, void *user) {
  if (!out_rgba) return fail(RT_ERR_INVALID, "out_rgba is NULL");
  if (flags & RT_FLAG_RGB24) return fail(RT_ERR_INVALID, "RT_FLAG_RGB24 applies to the device entry points only; rt_render returns ImageData.data (RGBA8)");
  if (!G.inited) return fail(RT_ERR_STATE, "rt_init has not been called");
  std::lock_guard<std::mutex> lk(G.mu);
  const auto t_begin = std::chrono::steady_clock::now();
  const int ndev = (int)G.dev.size();
  const size_t frame_bytes = (size_t)w * h * 4u;
  int rc;
  rt_stats agg;
  memset(&agg, 0, sizeof agg);

  // test build: RT_FORCE_GATHER=1 takes the ncclGather plan - also with ONE device, which runs the real RCCL symbols
  // (ncclCommInitAll, ncclGroupStart/End, ncclGather with one rank) on a one-GPU box
  const bool force_gather = RT_TEST_ENV("RT_FORCE_GATHER") != nullptr;
  g_last_plan = 0;
  if ((ndev == 1 && !force_gather) || h < (uint32_t)ndev * RT_TILE_H) {
    // ---- one GPU.  Large frames are rendered as a few row bands so that the PCIe copy-out of band i (copy
    //      stream) runs while band i+1 renders (render stream): the frame costs ~max(render, copy), not the sum ----
    rt_scene_dev *s = nullptr;
    if ((rc = scene_for(0, blob, bytes, &s))) return rc;
    device_state &D = G.dev[0];
    rc = ensure_device(0);                                          // (makes device 0 current: a previous multi-GPU call may have left another one)
    const bool count = (flags & RT_FLAG_COUNT) != 0;
    // Where the frame goes.  A buffer from rt_alloc_pinned (what the N-API layer hands in: the ImageData.data of main.js:83,
    // 195-200) is mapped into the GPU's address space: the kernel can store its pixels STRAIGHT into it over PCIe - 128-byte lines,
    // posted writes - with no staging frame in HBM, no copy engine and no band bookkeeping; the call then takes
    // ~max(kernel, frame bytes / PCIe).  Measured (r03_ab_log.md section 4) that is what the banded copy-out below takes as well -
    // the link, ~50-55 GB/s here, is the bound either way - and the copy engine is 2-7 % ahead for frames of 8 MiB and more, the
    // direct stores 3 % for smaller ones: the default follows the measurement.  Pageable memory always takes the copy-out.
    void *d_direct = nullptr;
    if (!rc && (g_direct_stores == 2 || (g_direct_stores == 1 && frame_bytes < (8u << 20)))) {
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, out_rgba) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) d_direct = attr.devicePointer;
      else (void)hipGetLastError();
    }
    if (d_direct && !rc) {
      // (rt_render_progressive: one launch per band, announced when its event has passed)
      const uint32_t nb = count ? 1u : (want_bands ? want_bands : 1u);
      const uint32_t rows_per = ((h + nb - 1) / nb + RT_TILE_H - 1) / RT_TILE_H * RT_TILE_H;
      rt_stats st;
      memset(&st, 0, sizeof st);
      if (nb == 1) {
        rt_tiles whole = {h, 0, 1, 1};
        rc = rt_render_tiles_device(s, w, h, &whole, d_direct, nullptr, flags, &st);       // (waits: stats)
        if (!rc && on_band) on_band(user, 0u, h);
      } else {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        std::vector<hipEvent_t> done(nb, nullptr);
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipEventRecord(e0, D.stream);
        for (uint32_t b = 0; b < nb && !rc && e == hipSuccess && b * rows_per < h; b++) {
          rt_tiles band = {rows_per, b, 1, 1};
          rc = rt_render_tiles_device(s, w, h, &band, (uint8_t *)d_direct + (size_t)b * rows_per * w * 4u, nullptr, flags, nullptr);
          if (rc) break;
          e = hipEventCreateWithFlags(&done[b], hipEventDisableTiming);
          if (e == hipSuccess) e = hipEventRecord(done[b], D.stream);
        }
        if (e == hipSuccess && !rc) e = hipEventRecord(e1, D.stream);
        for (uint32_t b = 0; b < nb && e == hipSuccess && !rc && done[b]; b++) {
          e = hipEventSynchronize(done[b]);
          const uint32_t r0 = b * rows_per;
          if (e == hipSuccess) on_band(user, r0, (r0 + rows_per <= h) ? rows_per : h - r0);
        }
        { const hipError_t e2 = hipStreamSynchronize(D.stream); if (e == hipSuccess) e = e2; }      // (nothing may still be storing into the caller's buffer)
        if (e == hipSuccess && !rc) { float ms = 0.f; e = hipEventElapsedTime(&ms, e0, e1); st.kernel_ms = ms; }
        if (e != hipSuccess && !rc) rc = fail(RT_ERR_DEVICE, "banded render into the pinned frame: %s", hipGetErrorString(e));
        st.pixels = (uint64_t)w * h;
        for (hipEvent_t ev : done) if (ev) (void)hipEventDestroy(ev);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
      }
      if (rc) return rc;
      agg = st;
      g_last_plan = 3;
    } else {
    if (!rc) rc = ensure_frame(0, frame_bytes);
    if (!rc && !D.copy_stream) {
      hipError_t e = hipStreamCreateWithFlags(&D.copy_stream, hipStreamNonBlocking);
      if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, "copy stream: %s", hipGetErrorString(e));
    }
    // The bands (first row, rows).  Counters come from one instrumented launch; a caller that asked for bands (rt_render_progressive)
    // gets that many.  Frames of 8 MiB and more: 4 bands - measured against 1, 2, 8, 16 equal bands, growing bands and the direct
    // stores above in profiles/r03_ab_log.md section 4: every plan ends within a few percent of frame bytes / PCIe rate.
    std::vector<std::pair<uint32_t, uint32_t>> bands;
    {
      const uint32_t n = count ? 1u : (want_bands ? want_bands : (frame_bytes < (8u << 20) ? 1u : (uint32_t)g_copy_bands));
      const uint32_t rows = ((h + n - 1) / n + RT_TILE_H - 1) / RT_TILE_H * RT_TILE_H;
      for (uint32_t r0 = 0; r0 < h; r0 += rows) bands.push_back({r0, rows});
    }
    const uint32_t n_bands = (uint32_t)bands.size();
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> band_done(n_bands, nullptr), copy_done(n_bands, nullptr);
    if (!rc) {
      rt_stats st;
      memset(&st, 0, sizeof st);
      if (n_bands == 1) {
        rt_tiles whole = {h, 0, 1, 1};
        rc = rt_render_tiles_device(s, w, h, &whole, D.d_frame, nullptr, flags, &st);
        if (!rc) {
          hipError_t e = hipMemcpyAsync(out_rgba, D.d_frame, frame_bytes, hipMemcpyDeviceToHost, D.stream);
          if (e == hipSuccess) e = hipStreamSynchronize(D.stream);
          if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, "copy-out: %s", hipGetErrorString(e));
          if (!rc && on_band) on_band(user, 0u, h);
        }
      } else {
        hipError_t e = hipEventCreate(&ev0);
        if (e == hipSuccess) e = hipEventCreate(&ev1);
        if (e == hipSuccess) e = hipEventRecord(ev0, D.stream);
        for (uint32_t b = 0; b < n_bands && !rc && e == hipSuccess; b++) {
          const uint32_t r0 = bands[b].first, band_rows = bands[b].second;
          const uint32_t rows = (r0 + band_rows <= h) ? band_rows : h - r0;
          rt_tiles band = {band_rows, b, 1, 1};
          uint8_t *d_band = (uint8_t *)D.d_frame + (size_t)r0 * w * 4u;
          rc = rt_render_tiles_device(s, w, h, &band, d_band, nullptr, flags, nullptr);
          if (rc) break;
          e = hipEventCreateWithFlags(&band_done[b], hipEventDisableTiming);
          if (e == hipSuccess) e = hipEventRecord(band_done[b], D.stream);
          if (e == hipSuccess) e = hipStreamWaitEvent(D.copy_stream, band_done[b], 0);
          if (e == hipSuccess) e = hipMemcpyAsync(out_rgba + (size_t)r0 * w * 4u, d_band, (size_t)rows * w * 4u, hipMemcpyDeviceToHost, D.copy_stream);
          if (e == hipSuccess && on_band) e = hipEventCreateWithFlags(&copy_done[b], hipEventDisableTiming);
          if (e == hipSuccess && on_band) e = hipEventRecord(copy_done[b], D.copy_stream);
        }
        if (e == hipSuccess && !rc) e = hipEventRecord(ev1, D.stream);
        // progressive delivery: every band is announced as soon as its rows are in the caller's buffer, while the
        // later bands are still rendering or on the PCIe link (the reference shows its frame row by row, main.js:201)
        for (uint32_t b = 0; on_band && b < n_bands && e == hipSuccess && !rc; b++) {
          if (!copy_done[b]) break;
          e = hipEventSynchronize(copy_done[b]);
          const uint32_t r0 = bands[b].first, band_rows = bands[b].second;
          if (e == hipSuccess) on_band(user, r0, (r0 + band_rows <= h) ? band_rows : h - r0);
        }
        // on EVERY way out the copies already queued into the caller's buffer are finished first: the caller may hand that
        // (pinned) buffer back to the pool as soon as this returns
        {
          const hipError_t e1 = hipStreamSynchronize(D.stream), e2 = hipStreamSynchronize(D.copy_stream);
          if (e == hipSuccess) e = (e1 != hipSuccess) ? e1 : e2;
        }
        if (e == hipSuccess && !rc) { float ms = 0.f; e = hipEventElapsedTime(&ms, ev0, ev1); st.kernel_ms = ms; }
        if (e != hipSuccess && !rc) rc = fail(RT_ERR_DEVICE, "banded render/copy-out: %s", hipGetErrorString(e));
        st.pixels = (uint64_t)w * h;
        for (hipEvent_t ev : band_done) if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : copy_done) if (ev) (void)hipEventDestroy(ev);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
      }
      agg = st;
    }
    if (rc) return rc;
    }
  } else {
    // ---- G GPUs of one node (one process): interleaved row tiles (sky rows are cheap, floor rows are not), reassembled
    //      on GPU 0.  Primary plan: PEER STORES - every GPU's kernel writes its tiles straight into GPU 0's frame buffer over
    //      xGMI (hipDeviceEnablePeerAccess; rows at their place in the frame, whole 128-byte lines: the scatter store), so
    //      there is no gather buffer, no collective and no de-interleave pass.  Fallback (no peer access between some pair,
    //      or the test build's RT_FORCE_GATHER): RGB24 bands, ONE ncclGather to GPU 0, one de-interleave pass. ----
    const uint32_t tile_rows = (h >= (uint32_t)ndev * 64u) ? 16u : RT_TILE_H;
    const uint32_t n_tiles_total = (h + tile_rows - 1) / tile_rows;
    const uint32_t tiles_per_rank = (n_tiles_total + ndev - 1) / ndev;
    bool peer_plan = !force_gather;
    for (int g = 1; g < ndev && peer_plan && !G.emulated; g++) {
      device_state &D = G.dev[g];
      if (D.peer_to_root == 0) {
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, D.hip_id, G.dev[0].hip_id);
        if (e == hipSuccess && can) {
          e = hipSetDevice(D.hip_id);
          if (e == hipSuccess) e = hipDeviceEnablePeerAccess(G.dev[0].hip_id, 0);
          if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
        }
        D.peer_to_root = (e == hipSuccess && can) ? 1 : -1;
      }
      if (D.peer_to_root < 0) peer_plan = false;
    }
    std::vector<rt_scene_dev *> scenes(ndev, nullptr);
    std::vector<hipEvent_t> ev0(ndev, nullptr), ev1(ndev, nullptr);
    const uint32_t kflags = flags & ~(uint32_t)RT_FLAG_COUNT;
    rc = RT_OK;
    if (peer_plan) {
      for (int g = 0; g < ndev && !rc; g++) rc = scene_for(g, blob, bytes, &scenes[g]);     // (the scenes stay cached on their devices)
      if (!rc) rc = ensure_frame(0, frame_bytes);
      void *root_frame[1] = {G.dev[0].d_frame};
      for (int g = 0; g < ndev && !rc; g++) {
        rt_tiles t = {tile_rows, (uint32_t)g, (uint32_t)ndev, tiles_per_rank};
        if ((rc = ensure_device(g))) break;
        hipError_t e = hipEventCreate(&ev0[g]);
        if (e == hipSuccess) e = hipEventCreate(&ev1[g]);
        if (e == hipSuccess) e = hipEventRecord(ev0[g], G.dev[g].stream);
        if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, "timing events on device %d: %s", g, hipGetErrorString(e)); break; }
        // (the sky blocks of the whole frame are GPU 0's own work, below: the other GPUs do not send theirs over the links)
        rc = rt_render_scatter_device(scenes[g], w, h, &t, 1u, root_frame, nullptr, kflags | (g ? RT_FLAG_NO_SKY : 0u), nullptr);
        if (!rc && g == 0) {
          rt_tiles whole = {h, 0u, 1u, 1u};
          rc = rt_render_scatter_device(scenes[0], w, h, &whole, 1u, root_frame, nullptr, kflags | RT_FLAG_SKY_ONLY, nullptr);
        }
        if (!rc && (e = hipEventRecord(ev1[g], G.dev[g].stream)) != hipSuccess) rc = fail(RT_ERR_DEVICE, "timing events on device %d: %s", g, hipGetErrorString(e));
      }
      // every GPU's stores have landed in GPU 0's frame once its stream is drained; then the copy-out
      for (int g = 0; g < ndev; g++) {
        if (!G.dev[g].stream) continue;
        hipError_t e = hipSetDevice(G.dev[g].hip_id);
        if (e == hipSuccess) e = hipStreamSynchronize(G.dev[g].stream);
        if (e != hipSuccess && !rc) rc = fail(RT_ERR_DEVICE, "peer-store plan, device %d: %s", g, hipGetErrorString(e));
      }
      if (!rc) {
        device_state &R = G.dev[0];
        hipError_t e = hipSetDevice(R.hip_id);
        if (e == hipSuccess) e = hipMemcpyAsync(out_rgba, R.d_frame, frame_bytes, hipMemcpyDeviceToHost, R.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(R.stream);
        if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, "copy-out: %s", hipGetErrorString(e));
      }
    } else {
      if (!G.emulated && (rc = ensure_rccl(ndev))) return rc;      // nothing allocated yet
      // bands cross xGMI as RGB24 when the width allows it (the alpha byte is the constant 255, main.js:198; the
      // de-interleave restores it); tile_rows >= 8, so a band is a multiple of 96 bytes and d_final stays 16-byte aligned
      // (the 3x3 / 4x4 box filter of the two-pass supersampling stores RGBA8: those scenes gather RGBA8 bands)
      for (int g = 0; g < ndev && !rc; g++) rc = scene_for(g, blob, bytes, &scenes[g]);
      const bool rgb24 = (w & 3u) == 0 && !rc && scenes[0]->hd.supersample <= 2u;
      const size_t band_bytes = (size_t)tiles_per_rank * tile_rows * w * (rgb24 ? 3u : 4u);
      for (int g = 0; g < ndev && !rc; g++) rc = ensure_frame(g, band_bytes);
      if (!rc && !(rc = ensure_device(0))) {
        device_state &R = G.dev[0];
        if (R.gather_bytes < band_bytes * ndev + frame_bytes) {
          if (R.d_gather) (void)hipFree(R.d_gather);
          R.d_gather = nullptr; R.gather_bytes = 0;
          hipError_t e = hipMalloc(&R.d_gather, band_bytes * ndev + frame_bytes);
          if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, "gather buffer: %s", hipGetErrorString(e));
          else { R.gather_bytes = band_bytes * ndev + frame_bytes; rc = check_on_device(R.d_gather, R, "rt_render's gather buffer"); }
        }
      }
      for (int g = 0; g < ndev && !rc; g++) {
        rt_tiles t = {tile_rows, (uint32_t)g, (uint32_t)ndev, tiles_per_rank};
        if ((rc = ensure_device(g))) break;
        hipError_t e = hipEventCreate(&ev0[g]);
        if (e == hipSuccess) e = hipEventCreate(&ev1[g]);
        if (e == hipSuccess) e = hipEventRecord(ev0[g], G.dev[g].stream);
        if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, "timing events on device %d: %s", g, hipGetErrorString(e)); break; }
        rc = rt_render_tiles_device(scenes[g], w, h, &t, G.dev[g].d_frame, nullptr, kflags | (rgb24 ? RT_FLAG_RGB24 : 0u), nullptr);
        if (!rc && (e = hipEventRecord(ev1[g], G.dev[g].stream)) != hipSuccess) rc = fail(RT_ERR_DEVICE, "timing events on device %d: %s", g, hipGetErrorString(e));
      }
      if (!rc && G.emulated) {                      // one physical GPU: the gather is a set of device-to-device copies
        for (int g = 0; g < ndev; g++) {
          hipError_t e = hipMemcpyAsync((uint8_t *)G.dev[0].d_gather + band_bytes * g, G.dev[g].d_frame, band_bytes, hipMemcpyDeviceToDevice, G.dev[g].stream);
          if (e == hipSuccess) e = hipStreamSynchronize(G.dev[g].stream);
          if (e != hipSuccess && !rc) rc = fail(RT_ERR_DEVICE, "emulated gather: %s", hipGetErrorString(e));
        }
      } else if (!rc) {
        // one ncclGather per device inside one group; a group that was opened is always closed, and every return code counts
        int r = NCCL.group_start();
        if (r != 0) rc = fail(RT_ERR_DEVICE, "ncclGroupStart: %s", NCCL.errstr(r));
        else {
          for (int g = 0; g < ndev && !rc; g++) {
            hipError_t e = hipSetDevice(G.dev[g].hip_id);
            if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, "hipSetDevice(%d): %s", G.dev[g].hip_id, hipGetErrorString(e)); break; }
            r = NCCL.gather(G.dev[g].d_frame, g == 0 ? G.dev[0].d_gather : nullptr, band_bytes, NCCL_UINT8, 0, G.comms[g], G.dev[g].stream);
            if (r != 0) rc = fail(RT_ERR_DEVICE, "ncclGather on device %d: %s", g, NCCL.errstr(r));
          }
          r = NCCL.group_end();
          if (r != 0 && !rc) rc = fail(RT_ERR_DEVICE, "ncclGroupEnd: %s", NCCL.errstr(r));
        }
      }
      if (!rc) {
        device_state &R = G.dev[0];
        uint8_t *d_final = (uint8_t *)R.d_gather + band_bytes * ndev;
        rc = (rgb24 ? rt_deinterleave_rgb24_device : rt_deinterleave_device)(0, R.d_gather, d_final, w, h, tile_rows, (uint32_t)ndev, band_bytes, nullptr);
        if (!rc) {
          hipError_t e = hipSetDevice(R.hip_id);
          if (e == hipSuccess) e = hipMemcpyAsync(out_rgba, d_final, frame_bytes, hipMemcpyDeviceToHost, R.stream);
          if (e == hipSuccess) e = hipStreamSynchronize(R.stream);
          if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, "copy-out: %s", hipGetErrorString(e));
        }
      }
    }
    for (int g = 0; g < ndev; g++) {
      (void)hipSetDevice(G.dev[g].hip_id);
      if (G.dev[g].stream) (void)hipStreamSynchronize(G.dev[g].stream);
      if (ev0[g] && ev1[g]) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev0[g], ev1[g]) == hipSuccess && ms > agg.kernel_ms) agg.kernel_ms = ms;   // slowest GPU
      }
      if (ev0[g]) (void)hipEventDestroy(ev0[g]);
      if (ev1[g]) (void)hipEventDestroy(ev1[g]);
    }
    (void)hipSetDevice(G.dev[0].hip_id);
    if (rc) return rc;
    agg.pixels = (uint64_t)w * h;
    g_last_plan = peer_plan ? 1 : 2;
    if (on_band) on_band(user, 0u, h);           // several GPUs: the frame arrives whole
  }
  agg.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (stats) *stats = agg;
  return RT_OK;
}
}  // namespace

extern "C" void rt_shutdown(void) {
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const pinned_buf &f : g_pin_free) (void)hipHostFree(f.p);
    g_pin_free.clear();
  }
  std::lock_guard<std::mutex> lk(G.mu);
  if (!G.inited) return;
  if (G.comms_ready) { for (size_t g = 0; g < G.dev.size(); g++) if (G.comms[g]) NCCL.destroy(G.comms[g]); G.comms_ready = false; }
  for (device_state &D : G.dev) {
    if (!D.stream) continue;
    (void)hipSetDevice(D.hip_id);
    (void)hipStreamSynchronize(D.stream);
    if (D.cached_scene) rt_scene_free(D.cached_scene);
    if (D.d_frame) (void)hipFree(D.d_frame);
    if (D.d_gather) (void)hipFree(D.d_gather);
    if (D.d_counters) (void)hipFree(D.d_counters);
    (void)hipStreamDestroy(D.stream);
    if (D.copy_stream) (void)hipStreamDestroy(D.copy_stream);
    D = device_state();
  }
  G.dev.clear();
  G.inited = false;
}
