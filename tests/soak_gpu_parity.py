#!/usr/bin/env python3
"""Randomised parity soak (run by hand on the GPU box; NOT collected by pytest):

    python tests/soak_gpu_parity.py [--seeds 300] [--first 1000] [--out gpurun_out/soak.json]

Each seed draws a scene far outside the fixed test cases - 1..90 spheres with or without the ground/sky pair, any of
the four samplers, transparent occluders, 0..4 lights anywhere (also inside spheres), any camera (also inside a sphere),
fov 20..150, depth 0..8, supersample 1 or 2, ragged frame sizes - renders it with BOTH kernels through the C ABI and
compares every channel with the oracle's C restatement (the checker; the product path never sees it).

Reported per kernel: the worst difference, the fraction of channels off by exactly 1 (rounding ties moved by an ulp:
the tolerance of SURVEY 8(c)) and the number of PIXELS off by more than 1.  A pixel off by more than 1 is a decision
flipped by a last-ulp difference at a discontinuity (a checker boundary, a silhouette, a shadow edge, a ToInt32 parity):
the restatement and the kernel then shade different surfaces, and no tolerance in LSB describes that.  The soak counts
them so that DESIGN.md can state how rare they are; the fixed tests (tests/test_gpu_parity.py) have none.
"""
import argparse
import json
import math
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as ou  # noqa: E402
import rt_host  # noqa: E402


def look_at(org, tgt, up):
    org, tgt, up = (np.array(v, dtype=np.float64) for v in (org, tgt, up))
    z = tgt - org
    x = np.cross(up, z)
    y = np.cross(z, x)
    unit = lambda v: v * (1.0 / np.sqrt((v * v).sum())) if (v * v).sum() != 0 else v   # noqa: E731
    return {"origin": org.tolist(), "axisX": unit(x).tolist(), "axisY": unit(y).tolist(), "axisZ": unit(z).tolist()}


def draw_scene(seed, degenerate=False, many=False):
    rng = random.Random(seed)
    base = rt_host.load_scene("default14_stars")          # carries the two PNG textures and the checker texture
    ground_sky = [o for o in base["objects"] if o["r2"] >= 250000.0]
    objs = []
    if rng.random() < 0.8:
        objs += [dict(o) for o in ground_sky]
        if rng.random() < 0.5:                              # plain black sky instead of stars
            sky = [o for o in objs if o["r2"] > 1e6][0]
            sky["mtl"] = dict(sky["mtl"], sampler={"kind": 0})
    refract = rng.random() < 0.4
    n = rng.choice([25, 33, 48, 64, 65, 90, 128, 200]) if many else rng.choice([1, 2, 3, 5, 8, 12, 13, 14, 20, 33, 64, 65, 90])
    spread = 2.0 + 0.12 * n
    while len(objs) < n:
        r = rng.choice([rng.uniform(0.05, 0.4), rng.uniform(0.3, 1.5), rng.uniform(1.0, 4.0)])
        kind = rng.choice([0, 0, 0, 1, 2, 2])
        samp = {"kind": kind}
        if kind == 1:
            samp["texture"] = rng.randrange(3)
        if kind == 2:
            samp.update(freqU=rng.choice([2.0, 8.0, 40.0, 5000.0, 7.5]), freqV=rng.choice([1.0, 4.0, 20.0, 2500.0, 3.25]),
                        colors=[[rng.random() for _ in range(3)], [rng.random() for _ in range(3)]])
        a4 = rng.choice([0.0, 0.0, 0.5, 0.8, 1.0]) if refract else 0.0
        objs.append({"origin": [rng.uniform(-spread, spread), rng.uniform(-0.5, 4.0), rng.uniform(-spread - 2, spread)], "r2": r * r,
                     "mtl": {"color": [rng.random() for _ in range(3)],
                             "albedo": [rng.choice([0.0, 0.1, 1.0]), rng.uniform(0.0, 1.0), rng.choice([0.0, rng.uniform(0.0, 1.0)]),
                                        rng.choice([0.0, 0.3, 0.6, 1.0]), a4],
                             "specular_exponent": rng.choice([0.0, 1.0, 5.0, 10.0, 50.0, 500.0, 12.5]),
                             "refract_index": rng.choice([1.0, 1.3, 1.5, 0.8]), "sampler": samp}})
    # The reference builds its rays per COMPONENT (main.js:187-191, quirk q1), which is a pinhole only for a camera whose
    # axes are the world's: most draws translate the reference camera (so the picture shows the scene), a quarter are
    # arbitrary (whatever they show, kernel and restatement must agree).
    u = rng.random()
    if u < 0.35:
        cam = look_at([0.0, 1.5, 10.0], [0.0, 1.5, 0.0], [0.0, 1.0, 0.0])
    elif u < 0.75:
        org = [rng.uniform(-4, 4), rng.uniform(0.2, 5.0), rng.uniform(2.0, spread + 8.0)]
        if rng.random() < 0.2 and objs:                     # camera inside a sphere
            o = rng.choice(objs)
            org = [o["origin"][0] + 0.1 * math.sqrt(o["r2"]), o["origin"][1], o["origin"][2]]
        cam = look_at(org, [org[0], org[1], org[2] - 10.0], [0.0, 1.0, 0.0])
    else:
        org = [rng.uniform(-8, 8), rng.uniform(0.1, 9.0), rng.uniform(-8, 12)]
        cam = look_at(org, [rng.uniform(-1, 1), rng.uniform(0, 2), rng.uniform(-1, 1)], rng.choice([[0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [0.3, 1.0, 0.1]]))
    objs.sort(key=lambda o: 4 * math.pi * o["r2"] / max(math.dist(o["origin"], cam["origin"]), 1e-300))   # main.js:159-163
    lights = [[rng.uniform(-8, 8), rng.uniform(0.3, 12.0), rng.uniform(-8, 8)] for _ in range(rng.choice([0, 1, 2, 2, 3, 4]))]
    small = [o for o in objs if o["r2"] < 250000.0]
    if lights and rng.random() < 0.2 and small:             # a light inside a sphere (at its centre)
        lights[0] = list(rng.choice(small)["origin"])
    # (Not drawn: a light EXACTLY on a sphere's surface, e.g. at [0,0,0], which lies on the ground sphere.  Whether the
    # ground then shadows a point is `t < light_len` between two numbers that are equal up to rounding - a coin flip in
    # the reference itself, which the strict kernel reproduces and the product kernel, walking the ray from the light,
    # does not: --degenerate-lights puts that case back and shows tens of flipped pixels per such scene.)
    if degenerate and lights and objs and rng.random() < 0.2:
        lights[0] = list(rng.choice(objs)["origin"])
    s = dict(base)
    s.update(objects=objs, camera=cam, lights=lights, segs=rng.choice([0, 1, 2, 3, 3, 3, 5, 5, 8]), supersample=rng.choice([1, 1, 1, 2]),
             fovDeg=rng.choice([60.0, 60.0, 20.0, 35.0, 90.0, 150.0]), light_intensity=rng.choice([50.0, 30.0, 5.0]))
    w = rng.choice([32, 64, 96, 100, 131, 160, 33])
    h = rng.choice([24, 48, 64, 64, 77, 90, 90, 9])
    if many:                                                # the shadow grids and the bounce table at work: depth >= 2, larger frames
        s["segs"] = rng.choice([2, 3, 5, 8])
        w, h = rng.choice([(160, 90), (192, 128), (256, 144), (131, 77)])
    return s, w, h


def draw_adversarial(seed):
    """Scenes built to stress the product kernel's boundary marks: a high-frequency sampler - the ground's checker at
    5000 .. 1 000 000 per unit u, textured spheres - seen THROUGH two to five bounces off a tight cluster of small, nearly perfect
    mirrors (radius 0.04 - 0.3: a mirror of radius r at distance t magnifies a direction error by ~2t/r per bounce) and small glass
    spheres, at 3840x2160-class pixel cones (a 960-pixel-wide window of such a frame: the same projection distance, a quarter of the
    oracle's work).  Returns (scene, w, h, tiles): two 8-row tiles through the cluster's image."""
    rng = random.Random(seed ^ 0x2545F491)
    base = rt_host.load_scene("default14_stars")
    objs = []
    for o in base["objects"]:
        if o["r2"] >= 250000.0:
            o = dict(o)
            if o["r2"] > 1e6:
                o["mtl"] = dict(o["mtl"], sampler={"kind": 0})                       # black sky
            else:
                f = rng.choice([5000.0, 5000.0, 20000.0, 100000.0, 1000000.0])
                o["mtl"] = dict(o["mtl"], sampler=dict(o["mtl"]["sampler"], freqU=f, freqV=f / 2))
            objs.append(o)
    cx, cz = rng.uniform(-2.0, 2.0), rng.uniform(0.0, 4.0)
    cy = rng.uniform(0.3, 1.2)
    n_mirror, n_glass, n_tex = rng.randrange(3, 8), rng.randrange(0, 3), rng.randrange(0, 3)
    for i in range(n_mirror + n_glass + n_tex):
        if i < n_mirror:
            r = rng.choice([rng.uniform(0.04, 0.12), rng.uniform(0.1, 0.3)])
            mtl = {"color": [rng.uniform(0.5, 1.0) for _ in range(3)], "albedo": [0.0, rng.choice([0.0, 0.1, 0.3]), rng.choice([0.0, 0.2]), rng.choice([0.8, 0.95, 1.0]), 0.0],
                   "specular_exponent": 50.0, "refract_index": 1.0, "sampler": {"kind": 0}}
        elif i < n_mirror + n_glass:
            r = rng.uniform(0.08, 0.3)
            mtl = {"color": [1.0, 1.0, 1.0], "albedo": [0.0, 0.1, 0.2, rng.choice([0.0, 0.2]), rng.choice([0.8, 0.9])], "specular_exponent": 50.0,
                   "refract_index": rng.choice([1.0, 1.3, 1.5]), "sampler": {"kind": 0}}
        else:
            r = rng.uniform(0.2, 0.6)
            mtl = {"color": [1.0, 1.0, 1.0], "albedo": [rng.choice([0.0, 0.3]), 0.8, 0.1, 0.0, 0.0], "specular_exponent": 10.0, "refract_index": 1.0,
                   "sampler": {"kind": 1, "texture": rng.randrange(3)}}
        spread = 0.35 if i < n_mirror + n_glass else 0.9
        objs.append({"origin": [cx + rng.uniform(-spread, spread), max(r + 0.01, cy + rng.uniform(-spread, spread)), cz + rng.uniform(-spread, spread)], "r2": r * r, "mtl": mtl})
    w, h = 960, rng.choice([2160, 2160, 4320])
    proj_d = (1920.0 if h == 2160 else 3840.0) / math.tan(math.radians(30.0))             # of the 3840- / 7680-wide frame at the reference's 60 degrees
    fov = 2.0 * math.degrees(math.atan((w / 2.0) / proj_d))
    cam_org = [cx + rng.uniform(-0.3, 0.3), rng.uniform(0.8, 2.5), cz + rng.uniform(5.0, 9.0)]
    cam = look_at(cam_org, [cam_org[0], cam_org[1], cam_org[2] - 10.0], [0.0, 1.0, 0.0])
    objs.sort(key=lambda o: 4 * math.pi * o["r2"] / max(math.dist(o["origin"], cam["origin"]), 1e-300))
    s = dict(base)
    s.update(objects=objs, camera=cam, lights=[[cx + rng.uniform(-6, 6), rng.uniform(4.0, 12.0), cz + rng.uniform(-4, 6)] for _ in range(rng.choice([1, 2, 2]))],
             segs=rng.choice([3, 4, 5, 6]), supersample=1, fovDeg=fov, light_intensity=50.0)
    # the cluster's image row: y = h/2 - (cy - cam_y) / depth * proj_d
    row = h / 2.0 - (cy - cam_org[1]) / (cam_org[2] - cz) * proj_d
    n_t = h // 8
    t0 = min(max(int(row // 8) + rng.randrange(-3, 4), 0), n_t - 2)
    stride = rng.randrange(1, min(6, n_t - t0))
    return s, w, h, rt_host.RtTiles(8, t0, stride, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=300)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--many-spheres", action="store_true", help="25..200 spheres, depth >= 2, larger frames: the shadow-grid / bounce-table variants")
    ap.add_argument("--degenerate-lights", action="store_true", help="also place lights at the centres of the ground/sky spheres ([0,0,0] is ON the ground sphere)")
    ap.add_argument("--windowed", action="store_true", help="narrow cones: each scene as two 8-row tiles of a 3840x2160 or 7680x4320 frame (the launch table's "
                    "sky blocks, shadow masks and candidates at the headline's block size of ~0.5 degrees)")
    ap.add_argument("--sky-parts", action="store_true", help="also put every frame together the multi-GPU way: 2-4 senders' interleaved 8-row tiles without the "
                    "sky blocks (RT_FLAG_NO_SKY) plus the owner's fill (RT_FLAG_SKY_ONLY), in one buffer; must be the plain frame's bytes")
    ap.add_argument("--camera-moves", action="store_true", help="after its frame every scene gets two random camera moves (rt_scene_set_camera: the launch "
                    "table rebuilt on the GPU, mark counts forgotten); each moved frame must be the bytes a fresh upload of the moved scene renders")
    ap.add_argument("--adversarial", action="store_true", help="scenes built against the boundary marks' tolerance: high-frequency samplers seen through 2-5 bounces off "
                    "tight clusters of small mirrors and glass spheres, at 3840x2160-class pixel cones (draw_adversarial)")
    args = ap.parse_args()
    lib = rt_host.load_library()
    assert lib.rt_init(1) == 0, lib.rt_last_error()
    import ctypes as C
    t0 = time.time()
    tot = {k: {"channels": 0, "off_by_one": 0, "flipped_pixels": 0, "worst": 0, "scenes_with_flips": [], "exact_samples": 0} for k in ("fma", "strict")}
    pixels = 0
    # a soak that is cut short (timeout's SIGTERM, a GPU box's limit) still reports what it covered
    import signal
    stop = {"now": False}
    signal.signal(signal.SIGTERM, lambda *_: stop.__setitem__("now", True))
    done = 0

    def report(interrupted):
        out = {"seeds": [args.first, args.first + done - 1], "pixels_per_kernel": pixels, "seconds": round(time.time() - t0, 1)}
        if interrupted:
            out["interrupted_after_scenes"] = done
        for k, T in tot.items():
            out[k] = {"off_by_one_channel_fraction": T["off_by_one"] / max(T["channels"], 1), "flipped_pixels": T["flipped_pixels"],
                      "flipped_pixel_fraction": T["flipped_pixels"] / max(pixels, 1), "worst_channel_difference": T["worst"],
                      "scenes_with_flips": T["scenes_with_flips"][:40], "n_scenes_with_flips": len(T["scenes_with_flips"]),
                      "exact_samples": T["exact_samples"], "frames_that_differ_between_the_table_without_and_with_shadow_masks": T.get("table_mismatches", 0),
                      "frames_put_together_from_sky_parts_that_differ": T.get("sky_part_mismatches", 0),
                      "camera_moves": T.get("camera_moves", 0), "camera_moves_refused": T.get("camera_moves_refused", 0),
                      "moved_frames_that_differ_from_a_fresh_upload": T.get("camera_move_mismatches", 0)}
        text = json.dumps(out, indent=1)
        if args.out:
            open(args.out, "w").write(text)
        return text

    for seed in range(args.first, args.first + args.seeds):
        if stop["now"]:
            break
        if args.adversarial:
            scene, w, h, tiles = draw_adversarial(seed)
            rows = [8 * t + k for t in (tiles.tile_first, tiles.tile_first + tiles.tile_stride) for k in range(8)]
        else:
            scene, w, h = draw_scene(seed, args.degenerate_lights, args.many_spheres)
            tiles = rt_host.RtTiles(h, 0, 1, 1)
            rows = None
        if args.windowed and not args.adversarial:
            wrng = random.Random(seed ^ 0x5bd1e995)
            w, h = wrng.choice([(3840, 2160), (3840, 2160), (7680, 4320)])
            scene["supersample"] = 1
            n_t = h // 8
            tile0 = wrng.randrange(0, n_t - 1)
            stride = wrng.randrange(1, n_t - tile0)
            tiles = rt_host.RtTiles(8, tile0, stride, 2)
            rows = [8 * t + k for t in (tile0, tile0 + stride) for k in range(8)]
        blob = rt_host.flatten_scene(scene)
        n_px = (len(rows) if rows else h) * w
        want = np.frombuffer(ou.c_oracle_rows(blob, w, h, rows) if rows else ou.c_oracle_render(blob, w, h), dtype=np.uint8).reshape(n_px, 4).astype(np.int16)
        pixels += n_px
        r = rt_host.Renderer(blob, 0, lib)
        d = lib.rt_alloc_device(0, n_px * 4)
        try:
            for name, flags in (("fma", 0), ("strict", rt_host.RT_FLAG_STRICT_FP)):
                st = r.render_tiles(w, h, d, tiles, flags=flags, want_stats=True)
                tot[name]["exact_samples"] += st.exact_samples     # samples the second, list-driven strict launch traced again
                host = C.create_string_buffer(n_px * 4)
                assert lib.rt_copy_to_host(0, host, d, n_px * 4) == 0
                if args.sky_parts and not rows:
                    import shard
                    plain = host.raw
                    G = 2 + seed % 3
                    plan = shard.TilePlan(w, h, 8, G)
                    assert lib.rt_memset_device(0, d, 0, n_px * 4) == 0
                    for g in range(G):
                        r.render_scatter(w, h, [d], rt_host.RtTiles(*plan.rt_tiles(g)), flags=flags | rt_host.RT_FLAG_NO_SKY, want_stats=True)
                    r.render_scatter(w, h, [d], rt_host.RtTiles(h, 0, 1, 1), flags=flags | rt_host.RT_FLAG_SKY_ONLY, want_stats=True)
                    assert lib.rt_copy_to_host(0, host, d, n_px * 4) == 0
                    if host.raw != plain:
                        tot[name]["sky_part_mismatches"] = tot[name].get("sky_part_mismatches", 0) + 1
                if name == "fma" and len(scene["objects"]) > 16:
                    # many spheres: the first frame from a camera comes from a launch table without shadow masks, the second from the full
                    # table (rt_api.hip: renders_with_camera) - the second is the one compared below, and the two must be the same bytes
                    first = host.raw
                    r.render_tiles(w, h, d, tiles, flags=flags, want_stats=True)
                    assert lib.rt_copy_to_host(0, host, d, n_px * 4) == 0
                    if host.raw != first:
                        tot[name]["table_mismatches"] = tot[name].get("table_mismatches", 0) + 1
                got = np.frombuffer(host.raw, dtype=np.uint8).reshape(n_px, 4).astype(np.int16)
                diff = np.abs(got - want)
                T = tot[name]
                T["channels"] += diff.size
                T["off_by_one"] += int((diff == 1).sum())
                flips = int((diff.max(axis=1) > 1).sum())
                T["flipped_pixels"] += flips
                T["worst"] = max(T["worst"], int(diff.max()))
                if flips:
                    T["scenes_with_flips"].append({"seed": seed, "w": w, "h": h, "pixels": flips, "spheres": len(scene["objects"]),
                                                   "segs": scene["segs"], "ss": scene["supersample"]})
                # (last: the moves leave the resident scene with another camera)
                if args.camera_moves and not rows:
                    crng = random.Random(seed * 31 + (7 if flags else 3))
                    for _ in range(2):
                        cam = look_at([crng.uniform(-6, 6), crng.uniform(0.2, 5), crng.uniform(-6, 9)], [crng.uniform(-2, 2), crng.uniform(0, 2.5), crng.uniform(-2, 2)], [0.0, 1.0, 0.0])
                        moved = dict(scene, camera=cam)
                        try:
                            r.set_camera(cam)
                        except rt_host.RtError:
                            tot[name]["camera_moves_refused"] = tot[name].get("camera_moves_refused", 0) + 1      # (across the enclosing sphere: upload again)
                            break
                        r.render_tiles(w, h, d, tiles, flags=flags, want_stats=True)
                        moved_host = C.create_string_buffer(n_px * 4)
                        assert lib.rt_copy_to_host(0, moved_host, d, n_px * 4) == 0
                        r2 = rt_host.Renderer(rt_host.flatten_scene(moved), 0, lib)
                        try:
                            r2.render_tiles(w, h, d, tiles, flags=flags, want_stats=True)
                            fresh_host = C.create_string_buffer(n_px * 4)
                            assert lib.rt_copy_to_host(0, fresh_host, d, n_px * 4) == 0
                        finally:
                            r2.close()
                        tot[name]["camera_moves"] = tot[name].get("camera_moves", 0) + 1
                        if moved_host.raw != fresh_host.raw:
                            tot[name]["camera_move_mismatches"] = tot[name].get("camera_move_mismatches", 0) + 1
                    try:
                        r.set_camera(scene["camera"])               # back where the scene was drawn: the other kernel's frame is next
                    except rt_host.RtError:
                        r.close()
                        r = rt_host.Renderer(blob, 0, lib)
        finally:
            lib.rt_free_device(0, d)
            r.close()
        done += 1
        if done % 50 == 0:
            print("seed %d: %d pixels, %.0f s; flipped fma=%d strict=%d" % (seed, pixels, time.time() - t0, tot["fma"]["flipped_pixels"],
                                                                           tot["strict"]["flipped_pixels"]), flush=True)
        if done % 5000 == 0:
            report(True)                                    # checkpoint: a killed run leaves its last 5000-scene mark behind
    print(report(done < args.seeds))


if __name__ == "__main__":
    main()
