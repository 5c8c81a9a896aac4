#!/usr/bin/env python3
"""Per-wave timeline of ONE product launch (measurement build: profiles/ab_build.sh wavelog "-DRT_WAVE_LOG" hybrid).

    RT_HIP_LIB=build/ab/librt_hip_wavelog.so python3 profiles/wave_timeline.py [scene] [w] [h] [--deep-us 40]

Every wave stamps s_memrealtime (100 MHz) at entry and exit and where it ran (HW_ID, XCC_ID).  The script renders the frame a few times,
logs one launch, and prints what a whole-kernel duration cannot show: how long the deep waves (the ones that walk the 31-node trees of the
sphere that both reflects and refracts) take while they share their SIMD with the rest of the frame, how they are spread over the CUs,
when the last cheap wave ended, and how many waves were resident over time."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
import rt_host  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
scene = args[0] if len(args) > 0 else "default14"
w = int(args[1]) if len(args) > 1 else 3840
h = int(args[2]) if len(args) > 2 else 2160
deep_us = float(sys.argv[sys.argv.index("--deep-us") + 1]) if "--deep-us" in sys.argv else 40.0

lib = rt_host.load_library()
assert lib.rt_init(1) == 0
r = rt_host.Renderer(rt_host.load_scene(scene), 0, lib)
d = lib.rt_alloc_device(0, w * h * 4)
whole = rt_host.RtTiles(h, 0, 1, 1)
for _ in range(20):
    st = r.render_tiles(w, h, d, whole, want_stats=True)
path = "/tmp/rt_wave_log_%d.bin" % os.getpid()
os.environ["RT_WAVE_LOG_FILE"] = path
r.render_tiles(w, h, d, whole, want_stats=True)
del os.environ["RT_WAVE_LOG_FILE"]
st2 = r.render_tiles(w, h, d, whole, want_stats=True)
log = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
os.unlink(path)
ran = log[:, 1] != 0
log = log[ran]
t0 = log[:, 0].astype(np.int64)
t1 = log[:, 1].astype(np.int64)
base = t0.min()
start = (t0 - base) / 100.0          # microseconds
end = (t1 - base) / 100.0
dur = end - start
hw = log[:, 2]
simd = (hw >> np.uint64(4)) & np.uint64(3)
cu = (hw >> np.uint64(8)) & np.uint64(15)
sh = (hw >> np.uint64(12)) & np.uint64(1)
se = (hw >> np.uint64(13)) & np.uint64(7)
xcc = (hw >> np.uint64(32)) & np.uint64(15)
cu_key = (((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu).astype(np.int64)
simd_key = cu_key * 4 + simd.astype(np.int64)
wg = log[:, 3].astype(np.int64)

print("scene %s %dx%d: kernel %.1f us by HIP events (logged launch excluded: %.1f us), %d waves ran, span of the stamps %.1f us"
      % (scene, w, h, st.kernel_ms * 1e3, st2.kernel_ms * 1e3, len(log), end.max()))
print("distinct CUs %d, distinct SIMDs %d" % (len(np.unique(cu_key)), len(np.unique(simd_key))))
q = np.percentile(dur, [0, 10, 50, 90, 99, 100])
print("wave duration us: min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % tuple(q))
deep = dur >= deep_us
print("waves of %.0f us and more: %d (%.1f %% of the waves, %.1f %% of the wave-time)" % (deep_us, deep.sum(), 100.0 * deep.mean(), 100.0 * dur[deep].sum() / dur.sum()))
if deep.any():
    dq = np.percentile(dur[deep], [0, 10, 50, 90, 100])
    print("  their duration us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % tuple(dq))
    sq = np.percentile(start[deep], [0, 50, 90, 100])
    print("  their start us: min %.1f median %.1f p90 %.1f max %.1f;  their end us: median %.1f p90 %.1f max %.1f"
          % (sq[0], sq[1], sq[2], sq[3], np.percentile(end[deep], 50), np.percentile(end[deep], 90), end[deep].max()))
    per_simd = np.bincount(simd_key[deep] - simd_key.min())
    per_simd = per_simd[per_simd > 0] if False else np.bincount(np.unique(simd_key[deep], return_inverse=True)[1])
    all_simds = len(np.unique(simd_key))
    hist = np.bincount(per_simd, minlength=1)
    print("  deep waves per SIMD (over the %d SIMDs that ran anything; %d of them ran none): %s"
          % (all_simds, all_simds - len(per_simd), {int(k): int(v) for k, v in enumerate(hist) if v}))
    # is a deep wave slower where more deep waves share its SIMD?
    inv = np.unique(simd_key[deep], return_inverse=True)[1]
    share = per_simd[inv]
    for k in sorted(set(share.tolist())):
        m = share == k
        print("    on SIMDs with %d deep waves: median duration %.1f us, median end %.1f us (%d waves)" % (k, np.median(dur[deep][m]), np.median(end[deep][m]), m.sum()))
    print("  the last wave that is NOT deep ended at %.1f us; the launch ended at %.1f us" % (end[~deep].max() if (~deep).any() else 0.0, end.max()))
# resident waves over time
edges = np.arange(0.0, end.max() + 10.0, 10.0)
res_all = [(int(((start <= t) & (end > t)).sum()), int(((start <= t) & (end > t) & deep).sum())) for t in edges]
print("resident waves every 10 us (all / deep): " + " ".join("%d/%d" % x for x in res_all))
# order of dispatch: does the table's order (dearest first) hold in time?
order = np.argsort(wg, kind="stable")
first = order[: len(order) // 20]
print("the first 5 %% of the launch table's workgroups: start median %.1f us, duration median %.1f us" % (np.median(start[first]), np.median(dur[first])))
# where in the picture the long waves are, and which of them started late: the launch table (host build = the GPU's, word for word)
# gives every workgroup its block
if "--map" in sys.argv:
    import ctypes as C
    blob = rt_host.flatten_scene(rt_host.load_scene(scene))
    buf = C.create_string_buffer(blob, len(blob))
    t = rt_host.RtTiles(h, 0, 1, 1)
    n, nb = C.c_uint32(), C.c_uint32()
    flags = int(sys.argv[sys.argv.index("--table-flags") + 1]) if "--table-flags" in sys.argv else 7
    assert lib.rt_scene_launch_table(buf, len(blob), w, h, C.byref(t), flags, None, C.byref(n), C.byref(nb)) == 0
    n8 = (nb.value + 7) // 8
    ent = (C.c_uint32 * (32 * n8))()
    assert lib.rt_scene_launch_table(buf, len(blob), w, h, C.byref(t), flags, ent, C.byref(n), C.byref(nb)) == 0
    ent = np.frombuffer(ent, dtype=np.uint32).reshape(-1, 4)
    slot = (wg % 8) * n8 + wg // 8
    e0 = ent[slot, 0]
    tx, fr = (e0 & 2047).astype(np.int64), (e0 >> 15).astype(np.int64)
    ss = rt_host.load_scene(scene).get("supersample", 1)
    cw, ch = 96, 72                                   # pixels per map cell
    gx, gy = (w + cw - 1) // cw, (h + ch - 1) // ch
    mx = np.zeros((gy, gx)); late = np.zeros((gy, gx))
    cx, cy = np.minimum(tx * 32 // cw, gx - 1), np.minimum(fr // ch, gy - 1)
    np.maximum.at(mx, (cy, cx), dur)
    np.maximum.at(late, (cy, cx), np.where(deep, start, 0.0))
    print("longest wave per %dx%d-pixel cell, in units of 20 us (. = below 20 us):" % (cw, ch))
    for row in mx:
        print("".join("." if v < 20 else "%x" % min(15, int(v // 20)) for v in row))
    print("latest START of a deep wave per cell, in units of 20 us (. = no deep wave, 0 = started within 20 us):")
    for y in range(gy):
        print("".join("." if mx[y, x] < deep_us else "%x" % min(15, int(late[y, x] // 20)) for x in range(gx)))
    worst = np.argsort(-end)[:12]
    print("the 12 waves that ended last: " + "; ".join("tile_x %d row %d start %.0f dur %.0f rank %d" % (tx[i], fr[i], start[i], dur[i], wg[i]) for i in worst))
r.close()
lib.rt_free_device(0, d)
