"""rt_render's hand-over of the frame to the host, A/B on one GPU: the kernel storing straight into the pinned
frame against the banded copy-out with 1, 4, 8, 16 bands, and a pageable destination.  total_ms = what the caller waits for."""
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
import rt_host  # noqa: E402

lib = rt_host.load_library()
assert lib.rt_init(1) == 0, lib.rt_last_error()
rows = []
for scene, w, h in (("h8", 3840, 2160), ("default14", 3840, 2160), ("lcg64_ss1", 3840, 2160), ("h8", 7680, 4320), ("h8", 1920, 1080)):
    blob = rt_host.flatten_scene(rt_host.load_scene(scene))
    buf = C.create_string_buffer(blob, len(blob))
    n = w * h * 4
    pinned = lib.rt_alloc_pinned(n)
    pageable = C.create_string_buffer(n)
    ref = None
    for label, direct, bands, dst in (("direct stores into the pinned frame", 2, 4, pinned), ("copy-out, 1 band", 0, 1, pinned), ("copy-out, 2 bands", 0, 2, pinned), ("copy-out, 4 bands", 0, 4, pinned),
                                      ("copy-out, 8 bands", 0, 8, pinned), ("copy-out, 16 bands", 0, 16, pinned), 
                                      ("pageable destination, 4 bands", 1, 4, C.addressof(pageable))):
        assert lib.rt_render_options(direct, bands) == 0
        tot, ker, wall = [], [], []
        for i in range(24):
            st = rt_host.RtStats()
            t0 = time.perf_counter()
            rc = lib.rt_render(buf, len(blob), w, h, C.c_void_p(dst), 0, C.byref(st))
            t1 = time.perf_counter()
            assert rc == 0, lib.rt_last_error()
            if i >= 4:
                tot.append(st.total_ms); ker.append(st.kernel_ms); wall.append((t1 - t0) * 1e3)
        got = C.string_at(dst, n)
        if ref is None:
            ref = got
        rows.append({"scene": scene, "size": "%dx%d" % (w, h), "plan": label, "total_ms": round(statistics.median(tot), 4), "wall_ms": round(statistics.median(wall), 4),
                     "kernel_ms": round(statistics.median(ker), 4), "host_GBs": round(n / statistics.median(tot) / 1e6, 2), "same_bytes_as_direct": got == ref})
        print(json.dumps(rows[-1]), flush=True)
    lib.rt_free_pinned(pinned)
lib.rt_render_options(1, 4)
