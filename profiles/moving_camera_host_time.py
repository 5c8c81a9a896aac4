#!/usr/bin/env python3
"""Where a camera-move step's time goes on the HOST: wall time per step against the time spent inside rt_scene_set_camera and
inside the render call (both asynchronous), and the static frame as the control.   python3 profiles/moving_camera_host_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd")); sys.path.insert(0, ROOT)
import rt_host
import bench
scene = rt_host.load_scene("h8")
w, h, steps = 3840, 2160, 512
lib = rt_host.load_library()
assert lib.rt_init(1) == 0
r = rt_host.Renderer(scene, 0, lib)
d = lib.rt_alloc_device(0, w * h * 4)
cams = [bench.moving_camera(scene, k, 64) for k in range(64)]
whole = rt_host.RtTiles(h, 0, 1, 1)
for k in range(8):
    r.set_camera(cams[k]); r.render_tiles(w, h, d, whole)
r.render_tiles(w, h, d, whole, want_stats=True)
pc = time.perf_counter
for rep in range(5):
    t_cam = t_ren = 0.0
    t0 = pc()
    for k in range(steps):
        a = pc(); r.set_camera(cams[k % 64]); b = pc(); r.render_tiles(w, h, d, whole); c = pc()
        t_cam += b - a; t_ren += c - b
    t_issue = pc() - t0
    r.render_tiles(w, h, d, whole, want_stats=True)
    dt = pc() - t0
    print("moving: %.4f ms per step wall (issue loop alone %.4f); inside set_camera %.4f, inside render %.4f" % (dt / steps * 1e3, t_issue / steps * 1e3, t_cam / steps * 1e3, t_ren / steps * 1e3), flush=True)
for rep in range(3):
    t0 = pc()
    for k in range(steps):
        r.render_tiles(w, h, d, whole)
    t_issue = pc() - t0
    r.render_tiles(w, h, d, whole, want_stats=True)
    dt = pc() - t0
    print("static: %.4f ms per step wall (issue loop alone %.4f)" % (dt / steps * 1e3, t_issue / steps * 1e3), flush=True)
lib.rt_free_device(0, d); r.close()
