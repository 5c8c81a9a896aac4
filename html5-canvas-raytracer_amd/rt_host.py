"""Python host side of the MI355X ray-sphere path: scene loading/flattening and the ctypes
binding of the C ABI in include/rt_hip.h.

This mirrors what the Node host (js/index.js + napi/rt_napi.cc) does, for the Python callers
the build contract requires (pytest, bench.py, __graft_entry__).  It contains NO rendering
code and no CPU fallback: every render goes through librt_hip.so -> the HIP kernel, and
`load_library()` raises if that library is missing.

Scene schema = the reference's locals/literals (main.js:85-163, :194, :283-284), see
js/scene.js; blob layout = include/rt_hip.h (rt_scene_header + tables).
"""
import base64
import ctypes as C
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SCENES_DIR = os.path.join(HERE, "scenes")
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(HERE, "csrc", "librt_hip.so")   # RT_HIP_LIB: A/B builds
# The TEST build of the same sources (-DRT_TESTING): the only library that reads the A/B and test environment switches
# (RT_EMULATE_DEVICES, RT_NO_BOUNCE_TABLE, RT_NO_FIXUP, ...) and exports rt_test_probe.  Tests select it with RT_HIP_LIB.
TEST_LIB_PATH = os.path.join(HERE, "csrc", "librt_hip_test.so")

RT_SCENE_MAGIC = 0x31535452
RT_ABI_VERSION = 2
HEADER_BYTES, SPHERE_BYTES, TEXDESC_BYTES = 208, 192, 16
SAMPLER_COLOR, SAMPLER_TEXTURE, SAMPLER_CHECKER, SAMPLER_STARS = 0, 1, 2, 3

RT_FLAG_COUNT = 1
RT_FLAG_STRICT_FP = 2
RT_FLAG_RGB24 = 4        # device entry points: 3 bytes per pixel, the constant alpha stays home (include/rt_hip.h)
RT_FLAG_NO_SKY = 8       # blocks that can only show the constant background are not stored (the frame's owner stores them: RT_FLAG_SKY_ONLY)
RT_FLAG_SKY_ONLY = 16
RT_FLAG_COMPACT = 32     # with RT_FLAG_RGB24 | RT_FLAG_NO_SKY: a compact band (the stored blocks back to back, for a collective)


# --------------------------------------------------------------------------- scenes
def load_scene(name_or_path):
    """Load a scene JSON written by js/flatten.js sceneToJSON (textures resolved to bytes)."""
    path = name_or_path
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(SCENES_DIR, name_or_path + ".json")
    with open(path) as f:
        scene = json.load(f)
    d = os.path.dirname(os.path.abspath(path))
    for t in scene["textures"]:
        if "base64" in t:
            t["texels"] = base64.b64decode(t.pop("base64"))
        else:
            with open(os.path.join(d, t["file"]), "rb") as f:
                t["texels"] = f.read()
        if len(t["texels"]) != t["width"] * t["height"] * 4:
            raise ValueError("texture is not width*height*4 bytes of RGBA8")
    return scene


def validate_scene(scene):
    """Host-side checks, same rules as js/scene.js validateScene."""
    cam = scene["camera"]
    for k in ("origin", "axisX", "axisY", "axisZ"):
        if len(cam[k]) != 3:
            raise ValueError("scene.camera.%s must be a 3-vector" % k)
    if not (isinstance(scene["segs"], int) and 0 <= scene["segs"] <= 16):
        raise ValueError("scene.segs must be an integer in [0,16]")
    if scene.get("supersample", 1) not in (1, 2, 3, 4):
        raise ValueError("scene.supersample must be 1, 2, 3 or 4")
    if not (1 <= len(scene["objects"]) <= 256):
        raise ValueError("scene.objects must hold 1..256 spheres")
    if len(scene["lights"]) > 16:
        raise ValueError("scene.lights must hold 0..16 lights")
    for i, o in enumerate(scene["objects"]):
        s = o["mtl"]["sampler"]
        if s["kind"] not in (SAMPLER_COLOR, SAMPLER_TEXTURE, SAMPLER_CHECKER, SAMPLER_STARS):
            raise ValueError("object %d: unsupported sampler kind %r (0 colour, 1 texture, 2 checker, 3 hashed stars)" % (i, s["kind"]))
        if s["kind"] == SAMPLER_TEXTURE and not (0 <= s["texture"] < len(scene["textures"])):
            raise ValueError("object %d: texture index out of range" % i)


def flatten_scene(scene):
    """scene dict -> pointer-free blob (bytes), byte-identical to js/flatten.js flattenScene."""
    validate_scene(scene)
    objs, lights, texs = scene["objects"], scene["lights"], scene["textures"]
    objects_off = HEADER_BYTES
    lights_off = objects_off + len(objs) * SPHERE_BYTES
    tex_off = lights_off + len(lights) * 24
    cursor = tex_off + len(texs) * TEXDESC_BYTES
    texel_off = []
    for t in texs:
        texel_off.append(cursor)
        cursor += (len(t["texels"]) + 7) & ~7
    total = cursor
    cam = scene["camera"]
    out = bytearray(total)
    hdr = struct.pack(
        "<IIQ12d3d3dIIIIIIQQQ", RT_SCENE_MAGIC, RT_ABI_VERSION, total,
        *cam["origin"], *cam["axisX"], *cam["axisY"], *cam["axisZ"],
        float(scene.get("fovDeg", 60)), float(scene.get("light_intensity", 50)), float(scene.get("epsilon", 0.001)),
        *[float(x) for x in scene.get("miss_color", [1, 0, 0])],
        scene["segs"], scene.get("supersample", 1), len(objs), len(lights), len(texs), 0,
        objects_off, lights_off, tex_off)
    assert len(hdr) == HEADER_BYTES
    out[0:HEADER_BYTES] = hdr
    o = objects_off
    for ob in objs:
        m = ob["mtl"]
        s = m["sampler"]
        if s["kind"] == SAMPLER_CHECKER:
            ck = [s["freqU"], s["freqV"], *s["colors"][0], *s["colors"][1]]
        elif s["kind"] == SAMPLER_STARS:
            ck = [s["threshold"], s["scale"], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
        else:
            ck = [0.0] * 8
        rec = struct.pack("<22d2id", *ob["origin"], ob["r2"], *m["color"], m["specular_exponent"],
                          *m["albedo"], m["refract_index"], *[float(x) for x in ck],
                          s["kind"], s["texture"] if s["kind"] == SAMPLER_TEXTURE else -1, 0.0)
        assert len(rec) == SPHERE_BYTES
        out[o:o + SPHERE_BYTES] = rec
        o += SPHERE_BYTES
    for l in lights:
        out[o:o + 24] = struct.pack("<3d", *l)
        o += 24
    for t, off in zip(texs, texel_off):
        out[o:o + TEXDESC_BYTES] = struct.pack("<IIQ", t["width"], t["height"], off)
        o += TEXDESC_BYTES
    for t, off in zip(texs, texel_off):
        out[off:off + len(t["texels"])] = t["texels"]
    return bytes(out)


# --------------------------------------------------------------------------- C ABI binding
class RtTiles(C.Structure):
    _fields_ = [("tile_rows", C.c_uint32), ("tile_first", C.c_uint32), ("tile_stride", C.c_uint32), ("n_tiles", C.c_uint32)]


class RtStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("pixels", C.c_uint64), ("rays", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("sphere_tests", C.c_uint64), ("exact_samples", C.c_uint64)]


# every symbol include/rt_hip.h declares: (restype, argtypes)
ABI = {
    "rt_init": (C.c_int, [C.c_int]),
    "rt_shutdown": (None, []),
    "rt_device_count": (C.c_int, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (C.c_uint32, []),
    "rt_build_id": (C.c_char_p, []),
    "rt_elapsed_report": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "rt_scene_validate": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rt_scene_cull_rects": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_double)]),
    "rt_scene_bounce_candidates": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "rt_scene_launch_table": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(RtTiles), C.c_int, C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_scene_upload": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rt_scene_free": (None, [C.c_void_p]),
    "rt_scene_set_camera": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "rt_render_tiles_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(RtTiles), C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.POINTER(RtStats)]),
    "rt_render_batch_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(RtTiles), C.c_uint32, C.c_void_p, C.c_uint64,
                                         C.c_void_p, C.c_uint32, C.POINTER(RtStats)]),
    "rt_render": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(RtStats)]),
    "rt_render_progressive": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.POINTER(RtStats)]),
    "rt_render_options": (C.c_int, [C.c_int, C.c_uint32]),
    "rt_alloc_pinned": (C.c_void_p, [C.c_size_t]),
    "rt_free_pinned": (None, [C.c_void_p]),
    "rt_alloc_device": (C.c_void_p, [C.c_int, C.c_size_t]),
    "rt_free_device": (None, [C.c_int, C.c_void_p]),
    "rt_copy_to_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rt_deinterleave_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint64, C.c_void_p]),
    "rt_memset_device": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_size_t]),
    "rt_render_scatter_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(RtTiles), C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p,
                                           C.c_uint32, C.POINTER(RtStats)]),
    "rt_ipc_export": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "rt_ipc_open": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rt_ipc_close": (C.c_int, [C.c_int, C.c_void_p]),
    "rt_compact_count": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_compact_expand_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_deinterleave_rgb24_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_uint64, C.c_void_p]),
}


class RtError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen librt_hip.so and type every entry point.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RtError("HIP library %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'`; "
                      "there is no CPU fallback for the render path" % p)
    lib = C.CDLL(p)
    older = os.environ.get("RT_HIP_LIB_OLDER") == "1"    # A/B runs against a library built from an older revision (profiles/ab_run.sh)
    for name, (res, args) in ABI.items():
        if older and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)        # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.rt_last_error()
        raise RtError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


class Renderer:
    """A scene resident on one GPU.  render_tiles() writes into caller-provided DEVICE memory
    (a torch uint8 CUDA tensor's data_ptr, or rt_alloc_device memory)."""

    def __init__(self, scene, device=0, lib=None):
        self.lib = lib or load_library()
        self.blob = scene if isinstance(scene, (bytes, bytearray)) else flatten_scene(scene)
        self.device = device
        if self.lib.rt_device_count() < 0:          # not initialised yet: use every visible GPU (an earlier rt_init's choice stands)
            _check(self.lib, self.lib.rt_init(0), "rt_init")
        h = C.c_void_p()
        buf = C.create_string_buffer(self.blob, len(self.blob))
        _check(self.lib, self.lib.rt_scene_upload(device, buf, len(self.blob), C.byref(h)), "rt_scene_upload")
        self.handle = h

    def set_camera(self, camera, stream=None):
        """Move the camera of the resident scene (lookAt, main.js:92-100): `camera` = {"origin", "axisX", "axisY", "axisZ"}.  One small
        asynchronous copy; the next render rebuilds what depends on it on the GPU."""
        v = [(C.c_double * 3)(*camera[k]) for k in ("origin", "axisX", "axisY", "axisZ")]
        _check(self.lib, self.lib.rt_scene_set_camera(self.handle, v[0], v[1], v[2], v[3], C.c_void_p(stream or 0)), "rt_scene_set_camera")

    def render_tiles(self, w, h, d_out, tiles=None, stream=None, flags=0, want_stats=False):
        t = tiles if isinstance(tiles, RtTiles) else RtTiles(*(tiles or (h, 0, 1, 1)))
        st = RtStats() if want_stats else None
        rc = self.lib.rt_render_tiles_device(self.handle, w, h, C.byref(t), C.c_void_p(d_out), C.c_void_p(stream or 0),
                                             flags, C.byref(st) if st is not None else None)
        _check(self.lib, rc, "rt_render_tiles_device")
        return st

    def render_batch(self, w, h, d_out, tiles, n_frames, frame_stride_bytes, stream=None, flags=0, want_stats=False):
        """n_frames frames' worth of `tiles` in ONE launch; frame f lands at d_out + f*frame_stride_bytes."""
        t = tiles if isinstance(tiles, RtTiles) else RtTiles(*tiles)
        st = RtStats() if want_stats else None
        rc = self.lib.rt_render_batch_device(self.handle, w, h, C.byref(t), n_frames, C.c_void_p(d_out), frame_stride_bytes,
                                             C.c_void_p(stream or 0), flags, C.byref(st) if st is not None else None)
        _check(self.lib, rc, "rt_render_batch_device")
        return st

    def render_scatter(self, w, h, frame_ptrs, tiles, stream=None, flags=0, want_stats=False):
        """`tiles` of len(frame_ptrs) frames in ONE launch; frame f's rows go to their place in the whole RGBA8 frame at
        frame_ptrs[f] (which may be another GPU's memory, peer-mapped with rt_ipc_open)."""
        t = tiles if isinstance(tiles, RtTiles) else RtTiles(*tiles)
        st = RtStats() if want_stats else None
        arr = (C.c_void_p * len(frame_ptrs))(*frame_ptrs)
        rc = self.lib.rt_render_scatter_device(self.handle, w, h, C.byref(t), len(frame_ptrs), arr, C.c_void_p(stream or 0), flags,
                                               C.byref(st) if st is not None else None)
        _check(self.lib, rc, "rt_render_scatter_device")
        return st

    def compact_count(self, w, h, tiles, stream=None):
        """(blocks, bytes per block) of a compact band (RT_FLAG_COMPACT) over `tiles` for the scene's current camera."""
        t = tiles if isinstance(tiles, RtTiles) else RtTiles(*tiles)
        n, bb = C.c_uint32(), C.c_uint32()
        _check(self.lib, self.lib.rt_compact_count(self.handle, w, h, C.byref(t), C.c_void_p(stream or 0), C.byref(n), C.byref(bb)), "rt_compact_count")
        return n.value, bb.value

    def compact_expand(self, w, h, tiles, d_compact, d_frame, stream=None):
        """Put the blocks of a compact band over `tiles` (rendered here or on a rank that holds the same scene and camera) back into the RGBA8 frame."""
        t = tiles if isinstance(tiles, RtTiles) else RtTiles(*tiles)
        _check(self.lib, self.lib.rt_compact_expand_device(self.handle, w, h, C.byref(t), C.c_void_p(d_compact), C.c_void_p(d_frame), C.c_void_p(stream or 0)), "rt_compact_expand_device")

    def close(self):
        if self.handle:
            self.lib.rt_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_id(lib=None):
    """`const build = '741'` (main.js:3) + the library's revision, e.g. '741.r4'."""
    return (lib or load_library()).rt_build_id().decode()


def elapsed_report(stats, lib=None):
    """The reference's end-of-frame string for a finished render (main.js:204-205): 'build #<id> (<elapsed>ms)'."""
    lib = lib or load_library()
    buf = C.create_string_buffer(96)
    n = lib.rt_elapsed_report(C.byref(stats), buf, len(buf))
    if n < 0:
        raise RtError("rt_elapsed_report failed: " + lib.rt_last_error().decode())
    return buf.value.decode()


def render(width, height, scene, flags=0, lib=None, max_devices=1):
    """render(width,height,scene) -> (bytes RGBA8, RtStats): the whole-frame entry point, host buffer out.
    max_devices > 1 (or 0 = all) lets rt_render shard the frame over the node's GPUs (RCCL gather)."""
    lib = lib or load_library()
    blob = scene if isinstance(scene, (bytes, bytearray)) else flatten_scene(scene)
    _check(lib, lib.rt_init(max_devices), "rt_init")
    n = width * height * 4
    p = lib.rt_alloc_pinned(n)
    if not p:
        raise RtError("rt_alloc_pinned(%d) failed: %s" % (n, lib.rt_last_error().decode()))
    try:
        st = RtStats()
        buf = C.create_string_buffer(blob, len(blob))
        _check(lib, lib.rt_render(buf, len(blob), width, height, C.c_void_p(p), flags, C.byref(st)), "rt_render")
        return C.string_at(p, n), st
    finally:
        lib.rt_free_pinned(p)
