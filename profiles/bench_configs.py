#!/usr/bin/env python3
"""Times every BASELINE.json config on ONE MI355X (kernel only, frame resident in HBM) and checks sampled rows
of each frame against the oracle's C restatement (checker only).  Not the headline bench (that is bench.py);
this produces the per-config table in DESIGN.md and, with --out, the tracked JSON it is quoted from.
usage: python profiles/bench_configs.py [--big] [--out profiles/r02_configs.json]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as ou
import rt_host

lib = rt_host.load_library()
assert lib.rt_init(1) == 0
CONFIGS = [("cfg1 256x256 2 spheres 1 light depth 1", "cfg1", 256, 256), ("cfg2 1920x1080 earth+mars depth 2", "cfg2", 1920, 1080),
           ("cfg3 3840x2160 H8 depth 3 (headline)", "h8", 3840, 2160), ("cfg4 7680x4320 H8 depth 3 (1 GPU)", "h8", 7680, 4320),
           ("reference scene 3840x2160: 14 spheres, refraction, depth 8", "default14", 3840, 2160),
           ("cfg5 scene 4096x4096, 2x2 supersample, 64 spheres, depth 5", "lcg64", 4096, 4096),
           ("cfg5 scene 2048x2048, 3x3 supersample (8(f)-4)", "lcg64_ss3", 2048, 2048), ("cfg5 scene 2048x2048, 4x4 supersample (8(f)-4)", "lcg64_ss4", 2048, 2048)]
if "--big" in sys.argv:
    CONFIGS.append(("cfg5 16384x16384, 2x2 supersample, 64 spheres, depth 5 (1 GPU)", "lcg64", 16384, 16384))
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
results = []
for label, name, w, h in CONFIGS:
    scene = rt_host.load_scene(name)
    blob = rt_host.flatten_scene(scene)
    r = rt_host.Renderer(blob, 0, lib)
    n = w * h * 4
    d = lib.rt_alloc_device(0, n)
    t = rt_host.RtTiles(h, 0, 1, 1)
    r.render_tiles(w, h, d, t, want_stats=True)                       # warm-up
    ms = sorted(r.render_tiles(w, h, d, t, want_stats=True).kernel_ms for _ in range(7 if w * h < 1 << 27 else 3))
    st = r.render_tiles(w, h, d, t, flags=rt_host.RT_FLAG_COUNT, want_stats=True)
    rows = sorted(set(int((k + 0.5) * h / 6) for k in range(6)))
    host = C.create_string_buffer(w * 4)
    worst = 0
    t0 = time.time()
    for y in rows:
        assert lib.rt_copy_to_host(0, host, d + y * w * 4, w * 4) == 0
        worst = max(worst, ou.max_lsb(host.raw, ou.c_oracle_render(blob, w, h, y, y + 1))[0])
        if time.time() - t0 > 120:
            break
    ss = scene.get("supersample", 1)
    med = ms[len(ms) // 2]
    results.append({"config": label, "scene": name, "w": w, "h": h, "supersample": ss, "kernel_ms": round(med, 4), "kernel_ms_all": [round(x, 4) for x in ms],
                    "mpixel_per_s": round(w * h / med / 1e3, 1), "msample_per_s": round(w * h * ss * ss / med / 1e3, 1),
                    "rays_per_pixel": round(st.rays / st.pixels, 3), "sphere_tests_per_pixel": round(st.sphere_tests / st.pixels, 2),
                    "max_lsb_vs_c_oracle_rows": worst, "rows_checked": len(rows)})
    print(json.dumps(results[-1]), flush=True)
    lib.rt_free_device(0, d)
    r.close()
if out_path:
    import subprocess
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip() or None
    json.dump({"what": "every BASELINE config on ONE MI355X: kernel only (hipEvent around single launches, median), frame resident in HBM; rows checked "
                       "against oracle/rt_oracle.c", "build": rt_host.build_id(lib), "git_head_at_build_container": head, "configs": results},
              open(out_path, "w"), indent=1)
