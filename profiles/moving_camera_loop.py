#!/usr/bin/env python3
"""The moving-camera loop of bench.py on its own (for rocprofv3 --kernel-trace --stats): per step rt_scene_set_camera + render.
   python3 profiles/moving_camera_loop.py [scene] [w] [h] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd")); sys.path.insert(0, ROOT)
import rt_host
import bench
scene_name = sys.argv[1] if len(sys.argv) > 1 else "h8"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 256
scene = rt_host.load_scene(scene_name)
lib = rt_host.load_library()
assert lib.rt_init(1) == 0
r = rt_host.Renderer(scene, 0, lib)
d = lib.rt_alloc_device(0, w * h * 4)
cams = [bench.moving_camera(scene, k, 64) for k in range(64)]
whole = rt_host.RtTiles(h, 0, 1, 1)
for k in range(8):
    r.set_camera(cams[k]); r.render_tiles(w, h, d, whole)
r.render_tiles(w, h, d, whole, want_stats=True)
t0 = time.perf_counter()
for k in range(steps):
    r.set_camera(cams[k % 64]); r.render_tiles(w, h, d, whole)
st = r.render_tiles(w, h, d, whole, want_stats=True)
dt = time.perf_counter() - t0
print("%s %dx%d: %.4f ms per step with a camera move before every frame (%d steps), %.1f Mpixel/s" % (scene_name, w, h, dt / steps * 1e3, steps, w * h * steps / dt / 1e6))
lib.rt_free_device(0, d); r.close()
