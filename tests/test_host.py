"""CPU tests of the host logic and of the C-ABI library surface (no compute calls without a GPU)."""
import ctypes as C
import hashlib
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import oracle_util as ou
import rt_host

ROOT = ou.ROOT
needs_node = pytest.mark.skipif(ou.node_path() is None, reason="node not installed")
SCENE_NAMES = ["cfg1", "cfg2", "h8", "h8_d8", "default14", "default14_stars", "lcg64", "lcg64_ss1"]


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    body = header[header.index("/* Library lifetime."):]      # prototypes follow the type declarations
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", body))
    assert declared == set(rt_host.ABI), declared ^ set(rt_host.ABI)
    lib = rt_host.load_library()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rt_abi_version() == rt_host.RT_ABI_VERSION


def test_struct_sizes_match_header(built):
    assert C.sizeof(rt_host.RtTiles) == 16 and C.sizeof(rt_host.RtStats) == 56
    blob = rt_host.flatten_scene(rt_host.load_scene("h8"))
    magic, ver, total = struct.unpack_from("<IIQ", blob, 0)
    assert (magic, ver, total) == (rt_host.RT_SCENE_MAGIC, rt_host.RT_ABI_VERSION, len(blob))
    assert len(blob) == 208 + 8 * 192 + 2 * 24 + 2 * 16 + 2 * 131072


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_validate_accepts_good_scenes(name, built):
    lib = rt_host.load_library()
    blob = rt_host.flatten_scene(rt_host.load_scene(name))
    buf = C.create_string_buffer(blob, len(blob))
    assert lib.rt_scene_validate(buf, len(blob)) == 0, lib.rt_last_error()


def test_validate_rejects_bad_blobs(built):
    lib = rt_host.load_library()
    good = bytearray(rt_host.flatten_scene(rt_host.load_scene("cfg2")))

    def check(mutate, code, needle):
        b = bytearray(good)
        mutate(b)
        buf = C.create_string_buffer(bytes(b), len(b))
        assert lib.rt_scene_validate(buf, len(b)) == code
        assert needle in lib.rt_last_error().decode()

    check(lambda b: struct.pack_into("<I", b, 0, 0xdeadbeef), -1, "magic")
    check(lambda b: struct.pack_into("<I", b, 4, 7), -1, "ABI version")
    check(lambda b: struct.pack_into("<Q", b, 8, len(b) + 8), -1, "total_bytes")
    check(lambda b: struct.pack_into("<I", b, 160, 99), -1, "segs")
    check(lambda b: struct.pack_into("<I", b, 168, 0), -1, "n_objects")
    check(lambda b: struct.pack_into("<Q", b, 184, len(b) - 8), -1, "object table")
    # an unknown sampler kind is an explicit error, not silence
    check(lambda b: struct.pack_into("<i", b, 208 + 176, 7), -2, "sampler kind 7")
    check(lambda b: struct.pack_into("<i", b, 208 + 0 * 192 + 180, 9), -1, "texture index")
    buf = C.create_string_buffer(bytes(good[:100]), 100)
    assert lib.rt_scene_validate(buf, 100) == -1


def test_no_gpu_means_loud_failure_not_fallback(built):
    """Without a GPU the render entry points must fail with RT_ERR_DEVICE / RT_ERR_STATE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt_host.RtError, match="no HIP device|rt_init"):
        rt_host.render(16, 16, rt_host.load_scene("cfg1"))


def test_python_host_rejects_unsupported_sampler():
    s = rt_host.load_scene("cfg1")
    s["objects"][0]["mtl"]["sampler"] = {"kind": 9}
    with pytest.raises(ValueError, match="unsupported sampler kind"):
        rt_host.flatten_scene(s)


@needs_node
@pytest.mark.parametrize("name", SCENE_NAMES)
def test_js_and_python_flatteners_agree(name, tmp_path):
    out = tmp_path / "scene.blob"
    r = ou.node_cli("flatten", ou.scene_json(name), "--out", out)
    blob = rt_host.flatten_scene(rt_host.load_scene(name))
    assert r["bytes"] == len(blob)
    assert hashlib.sha256(blob).hexdigest() == r["sha256"]


JS_PIPELINE = r"""
const S = require('%(pkg)s/js/scene.js'), SC = require('%(pkg)s/js/scenes.js'), P = require('%(pkg)s/js/png.js');
const zlib = require('zlib');
const cam = S.lookAt([0,1.5,10],[0,1.5,0],[0,1,0]);
const tex = {earth: S.textureFromRGBA(2,1,new Uint8Array(8)), mars: S.textureFromRGBA(2,1,new Uint8Array(8))};
const sc = SC.default14(tex);
const order = sc.objects.map(o => o.origin.join(','));
const ck = S.checkerTexture(S.createTexture(), 16, 8, [0,0,0], [1,1,1]);
// hand-made 2x2 RGBA PNG with filter types 0 (None) and 1 (Sub)
function crc(buf){let c,t=[];for(let n=0;n<256;n++){c=n;for(let k=0;k<8;k++)c=c&1?0xedb88320^(c>>>1):c>>>1;t[n]=c>>>0;}c=0xffffffff;for(const b of buf)c=t[(c^b)&255]^(c>>>8);return (c^0xffffffff)>>>0;}
function chunk(type,data){const b=Buffer.concat([Buffer.from(type),data]);const o=Buffer.alloc(12+data.length);o.writeUInt32BE(data.length,0);b.copy(o,4);o.writeUInt32BE(crc(b),8+data.length);return o;}
const ihdr=Buffer.alloc(13);ihdr.writeUInt32BE(2,0);ihdr.writeUInt32BE(2,4);ihdr[8]=8;ihdr[9]=6;
const raw=Buffer.from([0, 10,20,30,255, 40,50,60,128,  1, 1,2,3,4, 1,1,1,1]);
const png=Buffer.concat([Buffer.from([0x89,0x50,0x4e,0x47,0x0d,0x0a,0x1a,0x0a]),chunk('IHDR',ihdr),chunk('IDAT',zlib.deflateSync(raw)),chunk('IEND',Buffer.alloc(0))]);
const img=P.decodePNG(png);
let threw=false; try { const m=S.createMaterial([1,1,1],[1,0,0,0,0],0,1); m.sampler=function(){}; S.createScene({objects:[S.createSphere([0,0,0],1,m)]}); } catch(e){threw=/closures/.test(e.message);}
console.log(JSON.stringify({cam, order, ck: Array.from(ck.texels.slice(0,12)), img: Array.from(img.data), threw}));
"""


@needs_node
def test_js_host_scene_pipeline(tmp_path):
    """constructors, sort order (q4), lookAt, checker texture, PNG decode."""
    script = tmp_path / "t.js"
    script.write_text(JS_PIPELINE % {"pkg": os.path.join(ROOT, "html5-canvas-raytracer_amd")})
    out = json.loads(subprocess.check_output([ou.node_path(), str(script)], text=True))
    # main.js:92-100 with the default arguments gives the mirrored frame of quirk q1
    assert out["cam"]["axisX"] == [-1, 0, 0] and out["cam"]["axisY"] == [0, 1, 0] and out["cam"]["axisZ"] == [0, 0, -1]
    # chrome(1,.25,3), matte, mirror, chrome(1.5,2.5,0), metal, mars, glass, bubble, blue, red, green, earth, home, skybox
    assert out["order"] == ["1,0.25,3", "0,0.25,3", "0,2.5,-2", "1.5,2.5,0", "-1.5,2.5,0", "-50,20,-100", "-2.5,0.5,3", "2.5,0.5,3",
                            "0,1,-2", "-1.5,1,0", "1.5,1,0", "50,20,-100", "0,-500,0", "0,0,0"]
    assert out["ck"] == [0, 0, 0, 255, 255, 255, 255, 255, 0, 0, 0, 255]
    assert out["img"] == [10, 20, 30, 255, 40, 50, 60, 128, 1, 2, 3, 4, 2, 3, 4, 5]
    assert out["threw"]


@pytest.mark.skipif(not ou.have_reference(), reason="/root/reference not present")
@pytest.mark.reference
def test_png_decoder_matches_pil_on_reference_textures():
    from PIL import Image
    import numpy as np
    for n in ("earth", "mars"):
        pil = np.asarray(Image.open(os.path.join(ou.REFERENCE_DIR, n + ".png")).convert("RGBA")).tobytes()
        ours = open(os.path.join(ou.SCENES, n + "_256x128.rgba"), "rb").read()
        assert pil == ours


def test_cull_rectangles_are_conservative(built):
    """Host logic of the product kernel's primary-ray cull (rt_scene_cull_rects): for random spheres and the
    reference camera, every pixel whose primary ray hits the sphere (and every pixel whose LINE meets it) lies
    inside the rectangle; spheres around the camera are unbounded."""
    import math
    import random
    import numpy as np
    lib = rt_host.load_library()
    rng = random.Random(7)
    scene = rt_host.load_scene("h8")
    base = scene["objects"][0]
    objs = []
    for _ in range(60):
        r = rng.choice([0.05, 0.3, 1.0, 4.0, 30.0, 500.0])
        o = dict(base)
        o["origin"] = [rng.uniform(-60, 60), rng.uniform(-40, 40), rng.uniform(-120, 30)]
        o["r2"] = r * r
        objs.append(o)
    scene["objects"] = objs
    blob = rt_host.flatten_scene(scene)
    buf = C.create_string_buffer(blob, len(blob))
    out = (C.c_double * (4 * len(objs)))()
    assert lib.rt_scene_cull_rects(buf, len(blob), out) == 0, lib.rt_last_error()
    rects = np.array(out).reshape(-1, 4)
    w, h = 160, 90
    D = (w / 2) / math.tan(math.radians(scene["fovDeg"]) / 2)
    cam = np.array(scene["camera"]["origin"])
    s = np.array(scene["camera"]["axisX"]) + np.array(scene["camera"]["axisY"]) + np.array(scene["camera"]["axisZ"])
    X = (np.arange(w) - w / 2 + 0.5)[None, :].repeat(h, 0)
    Y = (h / 2 - np.arange(h) - 0.5)[:, None].repeat(w, 1)
    ray = np.stack([s[0] * X, s[1] * Y, np.full_like(X, s[2] * D)], -1)
    ray /= np.linalg.norm(ray, axis=-1, keepdims=True)
    bounded = 0
    for o, (x0, x1, y0, y1) in zip(objs, rects):
        c = np.array(o["origin"]) - cam
        tca = ray @ c
        d2 = c @ c - tca * tca
        line_meets = d2 <= o["r2"]
        inside = (X / D >= x0) & (X / D <= x1) & (Y / D >= y0) & (Y / D <= y1)
        assert not (line_meets & ~inside).any(), (o["origin"], o["r2"], (x0, x1, y0, y1))
        if c @ c <= o["r2"]:
            assert math.isinf(x0) and math.isinf(x1) and math.isinf(y0) and math.isinf(y1)
        bounded += int(math.isfinite(x0) and math.isfinite(y0))
        if math.isfinite(x0) and x0 > X[0, 0] / D and x1 < X[0, -1] / D and line_meets.any():   # tight when the image is inside the frame
            cols = np.where(line_meets.any(0))[0]
            assert (X[0, cols[0]] / D - x0) * D < 1.5 and (x1 - X[0, cols[-1]] / D) * D < 1.5
    assert bounded > 20


def test_validator_survives_corrupted_blobs(built):
    """Truncated and randomly corrupted blobs are rejected (or accepted) without crashing; whatever the validator
    accepts, the host-side scene analysis (cull rectangles) must digest.  The same loop was run once against an
    AddressSanitizer build of the library (hipcc -fsanitize=address -fno-gpu-sanitize): clean."""
    import random
    lib = rt_host.load_library()
    rng = random.Random(5)
    for name in ("cfg2", "default14", "lcg64"):
        blob = rt_host.flatten_scene(rt_host.load_scene(name))
        n = struct.unpack_from("<I", blob, 168)[0]
        out = (C.c_double * (4 * n))()
        for cut in (0, 8, 100, 207, 208, 300, len(blob) - 1):
            b = C.create_string_buffer(blob[:cut], max(cut, 1))
            assert lib.rt_scene_validate(b, cut) != 0
        accepted = 0
        for _ in range(200):
            b = bytearray(blob)
            for _ in range(rng.randrange(1, 6)):
                b[rng.randrange(0, 208 + 192 * n + 64)] = rng.randrange(256)
            buf = C.create_string_buffer(bytes(b), len(b))
            if lib.rt_scene_validate(buf, len(b)) == 0:
                accepted += 1
                assert lib.rt_scene_cull_rects(buf, len(b), out) == 0
        assert accepted > 0


def test_bounce_table_is_conservative(built):
    """Host logic of the product kernel's bounce table (rt_scene_bounce_candidates): rays that start ON a sphere (on its
    outside or its inside surface) in random directions - every sphere such a ray actually meets in front of its origin
    must be in the candidate set of (sphere, direction cell).  Includes overlapping spheres, a sphere inside another,
    a huge ground and an enclosing sky."""
    import random
    import numpy as np
    lib = rt_host.load_library()
    rng = random.Random(11)
    scene = rt_host.load_scene("h8")
    base = scene["objects"][0]
    objs = []
    for k in range(48):
        r = rng.choice([0.05, 0.3, 0.3, 1.0, 1.0, 4.0])
        o = dict(base)
        o["origin"] = [rng.uniform(-6, 6), rng.uniform(-1, 6), rng.uniform(-6, 6)]
        o["r2"] = r * r
        o["mtl"] = dict(base["mtl"], albedo=[0.0, 0.5, 0.5, 0.5 if k % 2 else 0.0, 0.0 if k % 2 else 0.8])   # every sphere spawns rays
        objs.append(o)
    for origin, r in (([0.0, -500.0, 0.0], 500.0), ([0.0, 0.0, 0.0], 5000.0)):
        o = dict(base)
        o["origin"], o["r2"] = origin, r * r
        o["mtl"] = dict(base["mtl"], albedo=[0.0, 0.5, 0.5, 0.3, 0.0])
        objs.append(o)
    scene["objects"] = objs
    blob = rt_host.flatten_scene(scene)
    buf = C.create_string_buffer(blob, len(blob))
    n = len(objs)
    cen = np.array([o["origin"] for o in objs])
    r2 = np.array([o["r2"] for o in objs])
    words = (n + 63) // 64
    out = (C.c_uint64 * words)()
    nrng = np.random.default_rng(5)
    checked = hits = listed = 0
    for trial in range(1500):
        i = rng.randrange(n)
        nrm = nrng.normal(size=3)
        nrm /= np.linalg.norm(nrm)
        p = cen[i] + nrm * np.sqrt(r2[i])
        d = nrng.normal(size=3)
        d /= np.linalg.norm(d)
        if trial % 7 == 0:                       # axis-aligned and face-diagonal directions sit on cell edges
            d = np.array(rng.choice([[1, 0, 0], [0, -1, 0], [0, 0, 1], [1, 1, 0], [-1, 0, 1], [1, -1, 1]]), dtype=float)
            d /= np.linalg.norm(d)
        dd = (C.c_double * 3)(*d)
        assert lib.rt_scene_bounce_candidates(buf, len(blob), i, dd, out) == 0, lib.rt_last_error()
        mask = [(out[j >> 6] >> (j & 63)) & 1 for j in range(n)]
        L = cen - p
        tca = L @ d
        disc = r2 - ((L * L).sum(axis=1) - tca * tca)
        thc = np.sqrt(np.maximum(disc, 0.0))
        met = (disc >= 0) & (tca + thc >= 0.001)          # some root at or beyond the reference's epsilon
        for j in np.nonzero(met)[0]:
            assert mask[j], (trial, i, j, d.tolist())
        checked += 1
        hits += int(met.sum())
        listed += sum(mask)
    assert checked == 1500 and hits > 3000
    assert listed < 0.75 * checked * n                    # and it does prune (the two giant spheres are always listed)


def test_build_stamp_and_elapsed_report(built):
    """`const build = '741'` (main.js:3) and the end-of-frame string 'build #' + build + ' (' + elapsed + 'ms)'
    (main.js:204-205), elapsed a whole number of milliseconds as a Date.now() difference is."""
    import re
    lib = rt_host.load_library()
    assert re.fullmatch(r"741\.r\d+", rt_host.build_id(lib))
    st = rt_host.RtStats()
    for ms, shown in [(0.0, 0), (0.49, 0), (12.5, 13), (19490.2, 19490), (-3.0, 0)]:
        st.total_ms = ms
        assert rt_host.elapsed_report(st, lib) == "build #%s (%dms)" % (rt_host.build_id(lib), shown)
    import ctypes as C
    assert lib.rt_elapsed_report(None, C.create_string_buffer(8), 8) == -1
    small = C.create_string_buffer(8)
    st.total_ms = 7.0
    assert lib.rt_elapsed_report(C.byref(st), small, 8) == len("build #%s (7ms)" % rt_host.build_id(lib)) and small.value == b"build #"    # snprintf rule


def _launch_table(lib, blob, w, h, tiles, ranked):
    import ctypes as C
    buf = C.create_string_buffer(blob, len(blob))
    t = rt_host.RtTiles(*tiles)
    n, nb = C.c_uint32(), C.c_uint32()
    assert lib.rt_scene_launch_table(buf, len(blob), w, h, C.byref(t), int(ranked), None, C.byref(n), C.byref(nb)) == 0, lib.rt_last_error()
    n8 = (nb.value + 7) // 8                                 # the table's stride comes from the number of BLOCKS
    out = (C.c_uint32 * (32 * n8))()                         # 4 words per slot
    assert lib.rt_scene_launch_table(buf, len(blob), w, h, C.byref(t), int(ranked), out, C.byref(n), C.byref(nb)) == 0
    entries = []
    for b in range(n.value):
        at = (b % 8) * n8 + b // 8                           # one contiguous part of the table per XCD
        e0, e1, e2, e3 = out[4 * at], out[4 * at + 1], out[4 * at + 2], out[4 * at + 3]
        entries.append((e0 & 2047, (e0 >> 11) & 15, e0 >> 15, e1 & 0xffffff) + (((e1 >> 31), ((e1 >> 24) & 127) + 1) if (int(ranked) & 2) else ()) +
                       ((e2, e3) if (int(ranked) & 4) else ()))
    return entries                                            # tile_x, rows_valid, first frame row, first band row [, sky flag, run of blocks] [, shadow masks]


@pytest.mark.parametrize("scene,w,h,tiles,min_share", [
    ("h8", 3840, 2160, (2160, 0, 1, 1), 0.40),     # the headline: 45 % of the workgroups are sky
    ("h8", 1001, 333, (16, 1, 3, 7), 0.2),        # ragged size, interleaved tiles
    ("default14", 640, 360, (360, 0, 1, 1), 0.1),
    ("lcg64", 512, 256, (256, 0, 1, 1), 0.2),      # supersample 2: 64 x 4 samples per workgroup
    ("cfg1", 256, 256, (256, 0, 1, 1), 0.2)])      # no enclosing sphere: the miss colour is the background
def test_launch_table_sky_marks_are_conservative(built, scene, w, h, tiles, min_share):
    """Workgroups marked as sky in the launch table (bit 31 of the second word; the kernel stores the background constant there
    and traces nothing): every marked workgroup is checked sample by sample here - corners, edges and a random interior
    subset, all of them for small frames - with the reference's own ray (main.js:186-193) and the exact line-sphere
    discriminant in binary64: no ray may come within a relative 1e-9 of meeting any sphere but the enclosing one.  And a
    sensible share of the frame is marked."""
    import math
    import random
    lib = rt_host.load_library()
    sc = rt_host.load_scene(scene)
    blob = rt_host.flatten_scene(sc)
    ss = sc.get("supersample", 1)
    rows_per_wg = 2 if ss == 2 else 8
    entries = _launch_table(lib, blob, w, h, tiles, 3)
    objs = sc["objects"]
    cam = sc["camera"]
    o = cam["origin"]
    asum = [cam["axisX"][k] + cam["axisY"][k] + cam["axisZ"][k] for k in range(3)]

    def dist(a, b):
        return math.sqrt(sum((a[i] - b[i]) ** 2 for i in range(3)))
    enclosing = None
    for e, q in enumerate(objs):
        lim = math.sqrt(q["r2"]) * (1 - 1e-6)
        if len(objs) > 1 and dist(o, q["origin"]) < lim and all(dist(l, q["origin"]) < lim for l in sc["lights"]) and \
           all(dist(p["origin"], q["origin"]) + math.sqrt(p["r2"]) < lim for j, p in enumerate(objs) if j != e):
            enclosing = e
            break
    W, H = w * ss, h * ss
    pw, ph = W / 2.0, H / 2.0
    pd = pw / math.tan(sc.get("fovDeg", 60) * math.pi / 180 / 2)
    wg_w, wg_h = 32 * ss, rows_per_wg * ss
    balls = [([q["origin"][k] - o[k] for k in range(3)], q["r2"]) for j, q in enumerate(objs) if j != enclosing]
    rng = random.Random(7)
    marked = live = 0
    exhaustive = w * h <= 256 * 256
    # the entries cover every block of the tile set exactly once: an ordinary entry one block, a sky entry a run of them
    tile_rows, first, stride, n_tiles = tiles
    rb = (tile_rows + rows_per_wg - 1) // rows_per_wg
    covered = {}
    for tile_x, valid, frow0, lrow, sky, run in entries:
        assert run == 1 or sky
        for t in range(run):
            key = (tile_x + t, lrow)
            assert key not in covered and tile_x + t < (w + 31) // 32
            covered[key] = 1
    assert len(covered) == ((w + 31) // 32) * n_tiles * rb
    blocks = [(tile_x + t, valid, frow0, sky) for tile_x, valid, frow0, _lrow, sky, run in entries for t in range(run)]
    for tile_x, valid, frow0, sky in blocks:
        if not valid:
            continue
        live += 1
        if not sky:
            continue
        marked += 1
        if not exhaustive and rng.random() > 0.03:           # large frames: a 3 % sample of the marked blocks
            continue
        xs = list(range(wg_w)) if exhaustive else sorted({0, wg_w - 1, wg_w // 2} | {rng.randrange(wg_w) for _ in range(5)})
        ys = list(range(wg_h)) if exhaustive else sorted({0, wg_h - 1} | {rng.randrange(wg_h) for _ in range(2)})
        for iy in ys:
            for ix in xs:
                sx, sy = tile_x * wg_w + ix, frow0 * ss + iy
                d = [asum[0] * (sx - pw + 0.5), asum[1] * (ph - sy - 0.5), asum[2] * pd]
                dd = sum(c * c for c in d)
                for C3, r2 in balls:
                    tca = sum(d[k] * C3[k] for k in range(3))
                    cc = sum(c * c for c in C3)
                    # the ray meets the ball iff tca > 0 (or the origin is inside) and tca^2 >= dd * (cc - r2)
                    assert cc > r2 and (tca <= 0 or tca * tca < dd * (cc - r2) * (1 - 1e-9)), (scene, tile_x, frow0, ix, iy)
    assert marked >= min_share * live, (scene, marked, live)


@pytest.mark.parametrize("scene,w,h,sample,min_empty", [
    ("h8", 3840, 2160, 0.02, 0.3),                 # the headline: most floor blocks have no possible occluder
    ("h8", 640, 360, 0.3, 0.05),
    ("cfg2", 480, 270, 0.5, 0.0),
    ("default14", 640, 360, 0.3, 0.0),             # 13 loop spheres: host logic only (its kernel variant uses the shadow grids)
    ("lcg64_ss1", 640, 360, 0.3, 0.0),             # 63 loop spheres: sets stored as empty / not empty
    ("cfg1", 128, 128, 1.0, 0.0),
    ("nolights:h8", 640, 360, 0.3, 0.0)])           # NO light: candidates are named, shadow sets are not
def test_launch_table_shadow_masks_are_conservative(built, scene, w, h, sample, min_empty):
    """Shadow masks of the launch table (word 2 of an entry: per light, the loop-order spheres that can shadow a PRIMARY hit of
    the block at all; the kernel skips the scan of a light whose set is empty): for sampled blocks every sample's primary hit
    is computed here with the exact discriminant, its shadow ray to each light is intersected with every other sphere, and
    every sphere that blocks (main.js:293-304: a root in (epsilon, light distance)) must be in the block's set; and the sphere
    that is hit must be among the block's primary candidates (word 3) when the entry names them."""
    import math
    import random
    lib = rt_host.load_library()
    if scene.startswith("nolights:"):
        sc = rt_host.load_scene(scene[9:])
        sc["lights"] = []
    else:
        sc = rt_host.load_scene(scene)
    blob = rt_host.flatten_scene(sc)
    ss = sc.get("supersample", 1)
    assert ss == 1
    entries = _launch_table(lib, blob, w, h, (h, 0, 1, 1), 4)
    objs = sc["objects"]
    cam = sc["camera"]
    o = cam["origin"]
    asum = [cam["axisX"][k] + cam["axisY"][k] + cam["axisZ"][k] for k in range(3)]
    eps = sc.get("epsilon", 0.001)

    def dist(a, b):
        return math.sqrt(sum((a[i] - b[i]) ** 2 for i in range(3)))
    enclosing = None
    for e, q in enumerate(objs):
        lim = math.sqrt(q["r2"]) * (1 - 1e-6)
        if len(objs) > 1 and dist(o, q["origin"]) < lim and all(dist(l, q["origin"]) < lim for l in sc["lights"]) and \
           all(dist(p["origin"], q["origin"]) + math.sqrt(p["r2"]) < lim for j, p in enumerate(objs) if j != e):
            enclosing = e
            break
    loop_of = {j: (j - 1 if (enclosing is not None and j > enclosing) else j) for j in range(len(objs)) if j != enclosing}
    pw, ph = w / 2.0, h / 2.0
    pd = pw / math.tan(sc.get("fovDeg", 60) * math.pi / 180 / 2)

    def hit_t(org, d, q):                                     # main.js:420-439 for a unit direction
        L3 = [q["origin"][k] - org[k] for k in range(3)]
        tca = sum(d[k] * L3[k] for k in range(3))
        d2 = sum(c * c for c in L3) - tca * tca
        if d2 > q["r2"]:
            return math.inf
        thc = math.sqrt(q["r2"] - d2)
        t0, t1 = tca - thc, tca + thc
        t = t1 if t0 < eps else t0
        return math.inf if t < eps else t
    rng = random.Random(11)
    stated = empty = named_blocks = 0
    for tile_x, valid, frow0, _lrow, smask, cands in entries:
        if not valid or (smask == 0xffffffff and cands == 0):
            continue
        stated += smask != 0xffffffff
        empty += (smask == 0)
        named_blocks += cands != 0
        if rng.random() > sample:
            continue
        for iy in sorted({0, valid - 1, rng.randrange(valid)}):
            for ix in sorted({0, 31, rng.randrange(32), rng.randrange(32)}):
                sx, sy = tile_x * 32 + ix, frow0 + iy
                if sx >= w:
                    continue
                d = [asum[0] * (sx - pw + 0.5), asum[1] * (ph - sy - 0.5), asum[2] * pd]
                n = math.sqrt(sum(c * c for c in d))
                d = [c / n for c in d]
                best, bi = math.inf, None
                for j, q in enumerate(objs):
                    t = hit_t(o, d, q)
                    if t < best:
                        best, bi = t, j
                if bi is None or bi == enclosing:
                    continue                                  # sky (flat: no lighting) or a miss
                # word 3: the (at most two) spheres the block's primary rays can meet at all - the kernel then skips its cull
                named = [cands & 255] + ([(cands >> 8) & 255] if (cands >> 16) > 1 else [])     # count << 16 | second << 8 | first
                assert cands == 0 or loop_of[bi] in named, (scene, tile_x, frow0, ix, iy, bi, hex(cands))
                hp = [o[k] + d[k] * best for k in range(3)]
                for k, lt in enumerate(sc["lights"] if smask != 0xffffffff else []):
                    sv = [lt[c] - hp[c] for c in range(3)]
                    llen = math.sqrt(sum(c * c for c in sv))
                    sv = [c / llen for c in sv]
                    for j, q in enumerate(objs):
                        if j == bi or j == enclosing:
                            continue
                        if hit_t(hp, sv, q) < llen:
                            bit = (1 << loop_of[j]) if len(loop_of) <= 16 else 1      # more than 16 loop spheres: empty (0) or not (0xffff)
                            assert (smask >> (16 * k)) & bit, (scene, tile_x, frow0, ix, iy, k, j, hex(smask))
    assert named_blocks > 0 and (stated > 0 or not sc["lights"]) and empty >= min_empty * stated, (scene, stated, empty, named_blocks)


def _soak_like_scene(seed):
    import soak_gpu_parity as soak
    sc, _w, _h = soak.draw_scene(seed, False, False)
    sc["supersample"] = 1
    return sc


def _horizon_scene():
    """A block that sees a sphere's SILHOUETTE has hit distances up to the tangent length sqrt(|C|^2 - r^2); rt_tables.cpp bounded it
    with an inflated radius once (before e31efa6), which is too SHORT by |C| r 1e-7 / tangent - 6e-4 on the horizon of the
    reference's ground sphere, far above the 1e-6 margins.  A ground with its horizon in the frame and small occluders whose
    shadows graze the far field."""
    sc = rt_host.load_scene("h8")
    keep = [o for o in sc["objects"] if o["r2"] >= 250000.0]          # ground (checker) and sky
    for k in range(6):
        keep.append({"origin": [-20.0 + 8.0 * k, 0.6 + 0.3 * k, -30.0 - 2.0 * k], "r2": 0.36 + 0.05 * k,
                     "mtl": {"color": [1, 0, 0], "albedo": [0, 0.8, 0.3, 0.0, 0.0], "specular_exponent": 50, "refract_index": 1.0, "sampler": {"kind": 0}}})
    sc["objects"] = sorted(keep, key=lambda o: 4 * 3.141592653589793 * o["r2"] / max(sum((o["origin"][k] - sc["camera"]["origin"][k]) ** 2 for k in range(3)) ** 0.5, 1e-300))
    sc["lights"] = [[60.0, 3.0, -60.0], [-80.0, 2.0, -20.0]]          # low lights: long shadows towards the horizon
    return sc


@pytest.mark.parametrize("scene,w,h", [
    ("h8", 256, 144), ("h8", 250, 256), ("cfg2", 256, 144), ("default14", 256, 144), ("lcg64_ss1", 256, 144), ("cfg1", 128, 128),
    ("horizon", 256, 144), ("horizon", 256, 48),
    ("soak:11", 160, 90), ("soak:12", 131, 77), ("soak:13", 96, 64), ("soak:14", 256, 144), ("soak:15", 100, 90), ("soak:16", 160, 90),
    ("soak:17", 64, 48), ("soak:18", 256, 144), ("soak:19", 131, 77), ("soak:20", 160, 90), ("soak:21", 96, 64), ("soak:22", 256, 144)])
def test_launch_table_statements_hold_for_every_sample_of_every_block(built, scene, w, h):
    """EXHAUSTIVE form of the two tests above (numpy, every sample of every block of a small frame): what the launch table states
    about a block - sky (nothing but the background shows), its primary candidates (word 3), per light the spheres that can
    shadow a primary hit of it (word 2) - is checked against every one of the block's samples: the reference's own ray
    (main.js:186-193), the reference's root selection (main.js:420-439) for the primary hit and for every shadow ray
    (main.js:293-304).  Scenes: the BASELINE scenes, a horizon scene for the tangent bound (see _horizon_scene), and scenes and
    cameras of the soak's generator (cameras inside spheres, wide and narrow fields of view, 1..90 spheres)."""
    import math
    lib = rt_host.load_library()
    sc = _horizon_scene() if scene == "horizon" else (_soak_like_scene(int(scene[5:])) if scene.startswith("soak:") else rt_host.load_scene(scene))
    sc["supersample"] = 1
    blob = rt_host.flatten_scene(sc)
    entries = _launch_table(lib, blob, w, h, (h, 0, 1, 1), 2 | 4)
    objs, cam = sc["objects"], sc["camera"]
    N = len(objs)
    o = np.array(cam["origin"], dtype=np.float64)
    asum = np.array([cam["axisX"][k] + cam["axisY"][k] + cam["axisZ"][k] for k in range(3)])
    eps = sc.get("epsilon", 0.001)
    C3 = np.array([q["origin"] for q in objs], dtype=np.float64)
    R2 = np.array([q["r2"] for q in objs], dtype=np.float64)
    lights = [np.array(l, dtype=np.float64) for l in sc["lights"]]

    def dist(a, b):
        return math.sqrt(sum((a[i] - b[i]) ** 2 for i in range(3)))
    enclosing = None
    for e, q in enumerate(objs):
        lim = math.sqrt(q["r2"]) * (1 - 1e-6)
        if N > 1 and dist(o, q["origin"]) < lim and all(dist(l, q["origin"]) < lim for l in sc["lights"]) and \
           all(dist(p["origin"], q["origin"]) + math.sqrt(p["r2"]) < lim for j, p in enumerate(objs) if j != e):
            enclosing = e
            break
    loop_of = np.array([(j - 1 if (enclosing is not None and j > enclosing) else j) for j in range(N)])
    n_loop = N - (enclosing is not None)
    pw, ph = w / 2.0, h / 2.0
    pd = pw / math.tan(sc.get("fovDeg", 60) * math.pi / 180 / 2)

    def hits(org, d):                                          # (P,3) origins / unit directions -> (P,N) distances, main.js:420-439
        L3 = C3[None, :, :] - org[:, None, :]
        tca = (d[:, None, :] * L3).sum(axis=2)
        d2 = (L3 * L3).sum(axis=2) - tca * tca
        with np.errstate(invalid="ignore"):
            thc = np.sqrt(R2[None, :] - d2)
        t0, t1 = tca - thc, tca + thc
        t = np.where(t0 < eps, t1, t0)
        return np.where((d2 > R2[None, :]) | (t < eps) | np.isnan(t), np.inf, t)
    ys, xs = np.mgrid[0:h, 0:w]
    D = np.stack([asum[0] * (xs - pw + 0.5), asum[1] * (ph - ys - 0.5), np.full(xs.shape, asum[2] * pd)], axis=2).reshape(-1, 3)
    D = D / np.sqrt((D * D).sum(axis=1))[:, None]
    O = np.broadcast_to(o, D.shape)
    T = hits(O, D)
    best = T.argmin(axis=1)                                    # strict <, first wins (main.js:229)
    tb = T[np.arange(len(best)), best]
    miss = ~np.isfinite(tb)
    HP = O + D * np.where(miss, 0.0, tb)[:, None]
    blockers = []                                              # per light: (P,N) does sphere j stand between the hit point and the light
    for lt in lights:
        sv = lt[None, :] - HP
        llen = np.sqrt((sv * sv).sum(axis=1))
        with np.errstate(invalid="ignore", divide="ignore"):
            sv = sv / llen[:, None]
        B = hits(HP, sv) < llen[:, None]
        B[np.arange(len(best)), best] = False
        if enclosing is not None:
            B[:, enclosing] = False
        blockers.append(B)
    best = best.reshape(h, w); miss = miss.reshape(h, w)
    n_sky = n_named = n_stated = 0
    for tile_x, valid, frow0, _lrow, sky, run, smask, cands in entries:
        if not valid:
            continue
        x0, x1, y0, y1 = tile_x * 32, min(w, (tile_x + (run if sky else 1)) * 32), frow0, frow0 + valid
        blk_best, blk_miss = best[y0:y1, x0:x1], miss[y0:y1, x0:x1]
        if sky:                                               # nothing but the enclosing sphere (or nothing at all) shows
            n_sky += 1
            assert (blk_miss | (blk_best == (enclosing if enclosing is not None else -1))).all(), (scene, "sky", tile_x, frow0, run)
            continue
        shown = set(np.unique(blk_best[~blk_miss]).tolist()) - {enclosing}
        if cands:
            n_named += 1
            named = [cands & 255] + ([(cands >> 8) & 255] if (cands >> 16) > 1 else [])
            assert {int(loop_of[j]) for j in shown} <= set(named), (scene, "candidates", tile_x, frow0, sorted(shown), hex(cands))
        if smask != 0xffffffff:
            n_stated += 1
            lit = (~blk_miss) & (blk_best != (enclosing if enclosing is not None else -1))
            for k in range(len(lights)):
                Bk = blockers[k].reshape(h, w, N)[y0:y1, x0:x1][lit]
                need = np.nonzero(Bk.any(axis=0))[0]
                for j in need:
                    bit = (1 << int(loop_of[j])) if n_loop <= 16 else 1
                    assert (smask >> (16 * k)) & bit, (scene, "shadow mask", tile_x, frow0, k, int(j), hex(smask))
    assert n_sky + n_named + n_stated > 0 or scene.startswith("soak:"), scene


@pytest.mark.parametrize("scene,w,h,tiles", [
    ("h8", 3840, 2160, (2160, 0, 1, 1)),          # the headline launch: 32 400 workgroups, ranked
    ("h8", 3840, 2160, (16, 3, 8, 17)),           # rank 3 of 8: interleaved 16-row tiles, the last one past the frame's end
    ("default14", 203, 97, (97, 0, 1, 1)),        # ragged, below the ranking threshold
    ("lcg64", 4096, 1031, (24, 1, 2, 22)),        # supersample 2: 2 rows per workgroup, general tile split
    ("cfg1", 1, 1, (1, 0, 1, 1))])
def test_launch_table_lists_every_block_exactly_once(built, scene, w, h, tiles):
    """Host logic of the product kernel's launch table (csrc/rt_tables.cpp build_launch_table, through the rt_scene_launch_table
    probe): whatever the order, the entries are exactly the blocks of the tile set - every (tile column, first frame row) once,
    with the band row the plain grid would have used and the number of rows inside the tile and the frame - and the ranked table
    is a permutation of the unranked one that never lists a dearer block after a cheaper one."""
    import math
    lib = rt_host.load_library()
    sc = rt_host.load_scene(scene)
    blob = rt_host.flatten_scene(sc)
    ss = sc.get("supersample", 1)
    rows_per_wg = 2 if ss == 2 else 8
    tile_rows, first, stride, n_tiles = tiles
    tiles_x = (w + 31) // 32
    rb = (tile_rows + rows_per_wg - 1) // rows_per_wg
    expect = []
    for i in range(n_tiles):
        for r in range(rb):
            trow0 = r * rows_per_wg
            frow0 = (first + i * stride) * tile_rows + trow0
            valid = max(0, min(rows_per_wg, tile_rows - trow0, h - frow0)) if frow0 < h else 0
            for x in range(tiles_x):
                expect.append((x, valid, frow0 if frow0 < h else 0, i * tile_rows + trow0))
    plain = _launch_table(lib, blob, w, h, tiles, False)
    assert plain == expect                                   # unranked = the plain grid's own order
    ranked = _launch_table(lib, blob, w, h, tiles, True)
    assert sorted(ranked) == sorted(expect)
    if len(expect) < 4096:
        assert ranked == expect                              # small launches are not ranked
        return
    # cost of a block, restated from the screen rectangles (rt_scene_cull_rects) and the weights of build_launch_table
    import ctypes as C
    rects = (C.c_double * (4 * len(sc["objects"])))()
    buf = C.create_string_buffer(blob, len(blob))
    assert lib.rt_scene_cull_rects(buf, len(blob), rects) == 0
    segs = sc["segs"]
    W, H = w * ss, h * ss
    pw, ph = W / 2.0, H / 2.0
    pd = pw / math.tan(sc.get("fovDeg", 60) * math.pi / 180 / 2)
    wg_w, wg_h = 32 * ss, rows_per_wg * ss

    def weight(o):
        a = o["mtl"]["albedo"]
        depth = min(segs - 1, 4) if segs > 1 else 0
        wgt = (2 if (a[1] > 0 or a[2] > 0) else 0) + (3 * depth if (a[3] > 0 or a[4] > 0) else 0)
        if a[3] > 0 and a[4] > 0 and segs > 1:
            wgt += 8 * (1 << min(segs - 1, 5))
        return wgt

    def cost(x, frow0):
        c = 1
        row0 = frow0 * ss
        for j, o in enumerate(sc["objects"]):
            wgt = weight(o)
            if not wgt:
                continue
            x_lo, x_hi, y_lo, y_hi = rects[4 * j:4 * j + 4]
            sx0, sx1 = x_lo * pd + pw - 0.5, x_hi * pd + pw - 0.5
            sy0, sy1 = ph - 0.5 - y_hi * pd, ph - 0.5 - y_lo * pd
            if not (sx1 >= 0) or not (sx0 <= W) or not (sy1 >= 0) or not (sy0 <= H):
                continue
            tx0, tx1 = int(max(sx0, 0.0) / wg_w), int(min(min(sx1, W - 1.0) / wg_w, tiles_x - 1))
            ys0, ys1 = max(sy0, 0.0), min(sy1, H - 1.0)
            if row0 + wg_h <= ys0 or row0 > ys1 or not (tx0 <= x <= tx1):
                continue
            c += wgt
        return c

    costs = [cost(x, fr) for x, _, fr, _ in ranked]
    assert all(a >= b for a, b in zip(costs, costs[1:]))     # dearest first
    assert costs[0] > costs[-1]                              # ... and the frame does end on cheaper blocks (sky) than it starts with


def test_dispatch_ranking_sees_what_a_blocks_mirrors_show(built):
    """Round 4 (profiles/r04_ab_log.md section 2): in the reference's own scene the launch's last 70 us were waves of blocks that do not
    SHOW the sphere with binary ray trees but MIRROR it - the small chrome sphere next to it (around tile_x 42, rows 1640-1664 of the
    3840x2160 frame, then ranked ~7 000th of 16 000 entries), the flanks of the ornaments.  The ranking now adds, per reflecting
    candidate of a block, half the weight of every two-child sphere its mirrored cone can meet (rt_block.h: rt_bounce_cost): those
    blocks are dispatched right behind the two-child sphere's own (the first ~1 000 entries), far ahead of the ordinary mirrors."""
    lib = rt_host.load_library()
    blob = rt_host.flatten_scene(rt_host.load_scene("default14"))
    buf = C.create_string_buffer(blob, len(blob))
    w, h = 3840, 2160
    t = rt_host.RtTiles(h, 0, 1, 1)
    n, nb = C.c_uint32(), C.c_uint32()
    assert lib.rt_scene_launch_table(buf, len(blob), w, h, C.byref(t), 7, None, C.byref(n), C.byref(nb)) == 0
    n8 = (nb.value + 7) // 8
    out = (C.c_uint32 * (32 * n8))()
    assert lib.rt_scene_launch_table(buf, len(blob), w, h, C.byref(t), 7, out, C.byref(n), C.byref(nb)) == 0
    ent = np.frombuffer(out, dtype=np.uint32).reshape(-1, 4)
    b = np.arange(n.value)
    e0 = ent[(b % 8) * n8 + b // 8, 0]
    rank = {(int(x), int(y)): int(i) for i, x, y in zip(b, e0 & 2047, e0 >> 15)}
    two_child = [rank[(tx, fr)] for tx in range(19, 27) for fr in range(1456, 1656, 8) if (tx, fr) in rank]      # the sphere's own middle (world x = +2.5: left of centre, the picture is mirrored)
    chrome = [rank[(42, fr)] for fr in (1640, 1648, 1656, 1664)]                                               # the small chrome sphere beside it
    assert max(two_child) < 1500, (min(two_child), max(two_child))                                            # (~940 blocks show that sphere; a few hundred that also show a mirror of it come first)
    assert max(chrome) < 2500, chrome                                                                         # (round 3: 6 977 .. 7 052)
    floor = [rank[(tx, 2000)] for tx in range(40, 80)]                                                        # plain floor: far behind both
    assert min(floor) > max(chrome)
