#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's configs.

    python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg4|cfg5]
    (N>1: started bare it launches `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` itself, as a
     child process, and relays rank 0's JSON line; started under torch.distributed.run it is one of the ranks)

--config cfg3 (default, the headline): BASELINE configs[2] — 3840x2160, the 8-sphere "H8" scene, 2 lights, depth 3.
    N = 1: a step is ONE kernel launch writing one frame (the hot path, main.js:184-199 + :216-451, through the C ABI's
    rt_render_tiles_device).  N > 1: a step is ONE frame, row-tiled over the N ranks and whole on rank 0 after one collective /
    barrier - north_star's form, strong scaling: `value`, with `efficiency_vs_n1` and `predicted` (what the links allow, stated
    per N).  The batch form (N frames per step, frame f whole on rank f, every directed link in use; weak scaling) is measured
    in the same run and rides in the line as `batch_mode`.
--config cfg4: BASELINE configs[3] — ONE 7680x4320 H8 frame per step, row-tiled over the N ranks and whole on rank 0 after
    one collective / barrier (strong scaling: the frame is fixed, per-GPU work shrinks with N).
--config cfg5: BASELINE configs[4] — ONE 16384x16384 frame, 2x2 supersample, 64 spheres, depth 5, the same way.

One process per GPU, torch.distributed backend "nccl" = RCCL over xGMI.  Every frame is sharded by interleaved 16-row tiles
across the N ranks.  Two plans put a frame together on the rank that owns it:

  * exchange plan (rt_render_batch_device + ONE collective + de-interleave): the bands cross the links as RGB24 - the alpha
    byte is the constant 255 (main.js:198) and is restored by the de-interleave.  Batch mode (cfg3): one all_to_all_single
    (band of frame f to rank f: all N(N-1) directed links at once; a gather to one root would be bound by that root's inbound
    links).  Single-frame mode (cfg4/cfg5): one gather to rank 0.  The bands of 4 consecutive steps
    (RT_BENCH_EXCHANGE_EVERY) share one collective, because issuing a c10d collective costs the host about as much as a
    step's GPU work; a group's exchange overlaps the renders of the next group (two slots, side stream).
  * peer stores (rt_render_scatter_device): every rank's kernel stores its tiles of a frame straight into the owning rank's
    frame buffer - peer-mapped once through IPC handles - rows in frame order, RGBA8, whole 128-byte lines.  No data
    collective, no send/recv buffers, no de-interleave: one all_reduce of one int per group of steps is the barrier.
    Set-up and a one-step PRE-FLIGHT (every owner checks its frame against the reference's rows) run first; any failure on
    any rank makes all ranks fall back to the exchange plan, and the JSON says so.  Senders leave the constant-background
    blocks out (RT_FLAG_NO_SKY) and each owner stores them itself (RT_FLAG_SKY_ONLY): half of the headline's pixels never
    cross a link.  Default at every N>1: BOTH plans are set up and timed for a few groups of steps and the faster one (slowest
    rank decides) runs the measurement (`config.plan_calibration_ms_per_step`).  RT_BENCH_P2P=1/0 forces a plan.

The scene is resident in HBM before the timed region and the frames stay in HBM (PCIe copy-out rate: DESIGN.md §6, never here).

Before the W warm-up steps the GPU is brought to steady clocks: trains of launches until two consecutive trains agree within
1 % (at least 50 ms), so a short timed region (the driver's --steps 20) measures the steady-state kernel.

Rank 0 prints ONE JSON line.  `roofline` prices the trace kernel against the HBM-store roofline the metric names (4 algorithmic
bytes per pixel); `roofline.traffic` and `fp64_valu.measured` come from rocprofv3 PMC passes over a short child run of this
same command (N=1; --no-pmc or a missing rocprofv3 fall back to the committed profile and say so).  The path is FP64-VALU bound,
so `fp64_valu` prices it against that bound: `measured` from the SQ_INSTS_VALU_*_F64 counters, `model` from the reference's
algorithmic operation count.  `cpu_baseline` is the oracle's JS restatement (bit-identical to main.js, tests/test_oracle.py)
on one host thread over a bounded sample of the same frame (N=1 only).
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TF = 78.6     # MI355X vector FP64 (FMA) peak = half the 157.3 TF FP32 vector rate
TILE_ROWS = 16
CONFIGS = {   # name: (scene, width, height, one frame per step shared by all ranks?)
    "cfg3": ("h8", 3840, 2160, False),
    "cfg4": ("h8", 7680, 4320, True),
    "cfg5": ("lcg64", 16384, 16384, True),
}


def cpu_baseline(scene_name, w, h):
    """Oracle leg (checker code, timed beside the GPU; never on the product path)."""
    import oracle_util as ou
    px_target = 4 * 3840 * 2160            # ~10 s of single-thread CPU work at ~4 Mpixel/s on the headline scene
    rows = min(h, max(8, px_target // w))
    reps = max(1, px_target // (rows * w))
    if ou.node_path():
        r = ou.node_cli("time", ou.scene_json(scene_name), w, h, rows, reps, timeout=900)
        what = ("%d whole %dx%d frames" % (reps, w, h)) if rows == h else ("%d evenly spaced rows of the %dx%d frame, %d times" % (rows, w, h, reps))
        return {"value": round(r["mpixel_per_s"], 4), "unit": "Mpixel/s", "cores": 1, "kind": "port",
                "sample": "%s (%d pixels, %.1f s), oracle/restate.js (bit-identical to main.js) under node %s, 1 thread, "
                          "after an untimed JIT warm-up pass over a quarter of the rows" % (what, r["pixels"], r["ms"] / 1e3, r["node"]),
                "mray_per_s": round(r["rays"] / r["ms"] / 1e3, 4), "host_cpus": os.cpu_count(),
                "reference_ratio": "the reference's own main.js cannot travel to this box; in the build container it needs 10.57 s for the 3840x2160 H8 frame "
                                   "against 2.73 s for restate.js (same SHA-256): the allocation-free restatement is 3.9x FASTER than main.js, "
                                   "so this baseline flatters the CPU by that factor"}
    import rt_host
    blob = rt_host.flatten_scene(rt_host.load_scene(scene_name))
    rows = min(h, 256)
    ys = [min(h - 1, int((k + 0.5) * h / rows)) for k in range(rows)]
    t0 = time.perf_counter()
    ou.c_oracle_rows(blob, w, h, ys)
    dt = time.perf_counter() - t0
    return {"value": round(rows * w / dt / 1e6, 4), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": "%d of %d rows evenly spaced, oracle/rt_oracle.c (gcc -O2, no FMA), 1 thread (node not installed)" % (rows, h),
            "host_cpus": os.cpu_count()}


def pmc_passes(argv, kernel_substr="rt_trace", only_traffic=False):
    """rocprofv3 counter passes over a short CHILD run of this same command (N=1): HBM bytes per launch (WRITE_SIZE, FETCH_SIZE in
    separate passes - they do not fit one - with the guide's gfx950 correction: FETCH_SIZE counts half the bytes) and the FP64
    VALU instruction counters.  Returns (dict, note); every failure is reported in the note and leaves the values None."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    import csv
    import glob
    out = {}
    passes = [("WRITE_SIZE",), ("FETCH_SIZE",), ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU", "SQ_INSTS_SALU"),
              ("SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY")]     # issue cycles (optional: a failure of this pass voids nothing else)
    optional = {"SQ_BUSY_CYCLES"}
    if only_traffic:
        passes = passes[:2]
    env = dict(os.environ, TMPDIR="/tmp", RT_BENCH_CHILD="1", RT_BENCH_NO_SETTLE="1")     # counters do not depend on clocks: no need to settle them
    for counters in passes:
        d = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), *argv,
                   "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-pmc"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            if r.returncode != 0:
                if counters[0] in optional:
                    out["issue_note"] = "rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(counters), r.returncode, (r.stderr or r.stdout)[-200:])
                    continue
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counters[0], r.returncode, (r.stderr or r.stdout)[-300:])
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                if counters[0] in optional:
                    out["issue_note"] = "rocprofv3 --pmc %s wrote no counter_collection.csv" % counters[0]
                    continue
                return None, "rocprofv3 --pmc %s wrote no counter_collection.csv" % counters[0]
            sums, launches = {}, {}
            for fpath in files:
                with open(fpath) as fh:
                    for row in csv.DictReader(fh):
                        kn = row.get("Kernel_Name", "")
                        m = re.search(r"rt_trace<(\w+), (\w+)", kn) or re.search(r"rt_traceILb([01])ELb([01])E", kn)
                        if kernel_substr not in kn or not m or m.group(2) in ("true", "1"):     # not the trace kernel, or its counting variant
                            continue
                        name = row["Counter_Name"]
                        sums[name] = sums.get(name, 0.0) + float(row["Counter_Value"])
                        launches.setdefault(name, set()).add(row.get("Dispatch_Id"))
            for name in counters:
                if name not in sums:
                    if counters[0] in optional:
                        out["issue_note"] = "counter %s missing from rocprofv3's output" % name
                        continue
                    return None, "counter %s missing from rocprofv3's output" % name
                out[name] = sums[name] / max(1, len(launches[name]))
        except Exception as e:      # noqa: BLE001
            return None, "rocprofv3 pass %s: %r" % (counters[0], e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out, None


def sky_block_fraction(lib, blob, w, h, tiles):
    """Share of the 32 x 8 blocks of `tiles` that can only show the constant background (what RT_FLAG_NO_SKY leaves out), from the
    host build of the launch table (a no-GPU probe of the library; reporting only)."""
    import ctypes as C
    import rt_host
    buf = C.create_string_buffer(blob, len(blob))
    t = rt_host.RtTiles(*tiles)
    n, nb = C.c_uint32(), C.c_uint32()
    if lib.rt_scene_launch_table(buf, len(blob), w, h, C.byref(t), 2, None, C.byref(n), C.byref(nb)) != 0 or nb.value == 0:
        return 0.0
    n8 = (nb.value + 7) // 8
    out = (C.c_uint32 * (32 * n8))()
    if lib.rt_scene_launch_table(buf, len(blob), w, h, C.byref(t), 2, out, C.byref(n), C.byref(nb)) != 0:
        return 0.0
    sky = 0
    for b in range(n.value):
        e1 = out[4 * ((b % 8) * n8 + b // 8) + 1]
        if e1 >> 31:
            sky += ((e1 >> 24) & 127) + 1
    return sky / nb.value


def look_at(org, tgt, up=(0.0, 1.0, 0.0)):
    """lookAt (main.js:92-100): the three camera axes from origin, target and up."""
    import numpy as np
    org, tgt, up = (np.array(v, dtype=np.float64) for v in (org, tgt, up))
    z = tgt - org
    x = np.cross(up, z)
    y = np.cross(z, x)
    unit = lambda v: v * (1.0 / np.sqrt((v * v).sum()))   # noqa: E731
    return {"origin": org.tolist(), "axisX": unit(x).tolist(), "axisY": unit(y).tolist(), "axisZ": unit(z).tolist()}


def moving_camera(scene, k, n):
    """Camera k of n on a slow orbit around the scene's own camera position (never axis-aligned: the reference's component-indexed
    target formula, main.js:187-191, degenerates there)."""
    import math
    o = scene["camera"]["origin"]
    r = math.hypot(o[0], o[2]) or 10.0
    a0 = math.atan2(o[0], o[2])
    a = a0 + 0.35 * math.sin(2.0 * math.pi * (k + 0.37) / n) + 0.02
    return look_at([r * math.sin(a), o[1] + 0.3 * math.cos(2.0 * math.pi * k / n) + 0.05, r * math.cos(a)], [0.1, 1.5, 0.0])


def reference_scene_leg(args, w, h, lib, dev_index, stream, torch, np):
    """The scene the reference's own main() draws (main.js:107-163: 14 spheres, glass and a sphere that both reflects and refracts, depth 8,
    main.js:194; Math.random pinned, so the sky is black) at the headline's frame size: the general kernel (rt_trace<REFRACT=1>) on one
    GPU, frames in HBM, launches back to back - kernel time by HIP events on the launch stream, HBM traffic from the same PMC passes as
    the headline's (a child run with --scene default14), parity against rows the reference itself rendered."""
    import rt_host
    import oracle_util as ou
    name = "default14"
    scene = rt_host.load_scene(name)
    out = {"scene": "%s: %d spheres, %d lights, depth %d (the reference's own scene, main.js:107-163, :194)" % (name, len(scene["objects"]), len(scene["lights"]), scene["segs"]),
           "w": w, "h": h}
    r = rt_host.Renderer(scene, dev_index, lib)
    frame = torch.empty((h, w, 4), dtype=torch.uint8, device=torch.device("cuda", dev_index))
    whole = rt_host.RtTiles(h, 0, 1, 1)
    for _ in range(30):
        r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream)
    torch.cuda.synchronize()
    n = 200
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record()
    for _ in range(n):
        r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream)
    ev_b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev_a.elapsed_time(ev_b) / n
    out.update(kernel_ms=round(kernel_ms, 4), ms_per_step=round(wall / n * 1e3, 4), value=round(w * h * n / wall / 1e6, 2), unit="Mpixel/s", steps=n)
    entry = next((f for f in ou.manifest()["frames"] if f["scene"] == name and (f["w"], f["h"]) == (w, h) and f["rows"]), None)
    host = frame.cpu().numpy()
    if entry:
        out["max_lsb_vs_reference_rows"] = int(ou.max_lsb(np.ascontiguousarray(host[entry["rows"]]).reshape(-1), ou.golden_frame(entry))[0])
        out["parity_checked_against"] = "tests/golden/%s (rendered by the reference itself)" % entry["file"]
    else:
        rows = sorted(set(int((k + 0.5) * h / 5) for k in range(5)))
        want = np.frombuffer(ou.c_oracle_rows(rt_host.flatten_scene(scene), w, h, rows), dtype=np.uint8)
        out["max_lsb_vs_reference_rows"] = int(ou.max_lsb(np.ascontiguousarray(host[rows]).reshape(-1), want)[0])
        out["parity_checked_against"] = "oracle/rt_oracle.c rows %s" % rows
    out["parity_ok"] = out["max_lsb_vs_reference_rows"] <= 1
    r.close()
    algo = 4.0 * w * h
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
            "algorithmic_bytes_per_launch": algo, "kernel": "rt_trace<REFRACT=1>"}
    if not args.no_pmc and os.environ.get("RT_BENCH_CHILD") != "1":
        pmc, note = pmc_passes(["--scene", name, "--width", str(w), "--height", str(h)], only_traffic=True)
        if pmc:
            roof["traffic"] = pmc["WRITE_SIZE"] * 1024.0 + 2.0 * pmc["FETCH_SIZE"] * 1024.0
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / algo, 3)
            roof["traffic_source"] = "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes) over a 12-step child run with --scene %s; FETCH_SIZE doubled (gfx950)" % name
        else:
            roof["traffic_source"] = "not collected: %s" % note
    out["roofline"] = roof
    return out


def cold_and_moving(args, scene, scene_name, w, h, lib, dev_index, stream, torch, np, steps):
    """What a frame costs when the scene or the camera is NOT the previous frame's (the reference recomputes everything on every
    redraw, main.js:180-201, and its camera is a parameter, main.js:92-100): (a) a cold frame - upload of the scene (one allocation,
    one copy; host-built geometry tables), the launch table's build on the GPU, the first frame - and (b) a train of frames in
    which the camera moves before EVERY frame (rt_scene_set_camera + render, nothing waits in between), with one sampled frame
    checked against the oracle's rows for that camera.  The code objects are loaded by then (the caller has rendered already)."""
    import rt_host
    import oracle_util as ou
    frame = torch.empty((h, w, 4), dtype=torch.uint8, device=torch.device("cuda", dev_index))
    whole = rt_host.RtTiles(h, 0, 1, 1)
    out = {}
    # (a) cold frame, three times (a new Renderer each): medians
    ups, firsts, builds = [], [], []
    blob = rt_host.flatten_scene(scene)
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = rt_host.Renderer(blob, dev_index, lib)
        t1 = time.perf_counter()
        r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ups.append((t1 - t0) * 1e3); firsts.append((t2 - t1) * 1e3)
        # the table build alone: GPU time of (move the camera + render) minus a render with the table in place, HIP events
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream)
        torch.cuda.synchronize()
        e[0].record(); r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream); e[1].record()
        r.set_camera(moving_camera(scene, 1, 64), stream=stream)
        e[2].record(); r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream); e[3].record()
        torch.cuda.synchronize()
        builds.append(e[2].elapsed_time(e[3]) - e[0].elapsed_time(e[1]))
        r.close()
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    out["cold_frame"] = {"upload_ms": round(med(ups), 4), "table_build_ms": round(med(builds), 4), "first_frame_ms": round(med(firsts), 4),
                         "note": "medians of 3; upload = rt_scene_upload (host-built geometry tables, ONE allocation, ONE copy); first frame = launch table built on the GPU "
                                 "(3 launches) + trace + the list-driven strict launch, until the frame is in HBM; table_build = GPU time of a frame after a camera move minus "
                                 "a frame with its table in place (HIP events)"}
    # (a') throughput with TWO static frames in flight: consecutive frames on alternate HIP streams and frame buffers (one resident
    # scene; a frame's tail - its deepest waves - runs beside the next frame's start).  Not the headline: kernels overlap, so a
    # kernel's own duration is no longer the step time.
    try:
        r = rt_host.Renderer(blob, dev_index, lib)
        s2 = torch.cuda.Stream(device=torch.device("cuda", dev_index)).cuda_stream
        frame_b = torch.empty_like(frame)
        pairs = [(stream, frame), (s2, frame_b)]
        for k2 in range(16):
            r.render_tiles(w, h, pairs[k2 & 1][1].data_ptr(), whole, stream=pairs[k2 & 1][0])
        torch.cuda.synchronize()
        n2 = max(64, min(steps, 600))
        t0 = time.perf_counter()
        for k2 in range(n2):
            r.render_tiles(w, h, pairs[k2 & 1][1].data_ptr(), whole, stream=pairs[k2 & 1][0])
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        same = bool(torch.equal(frame, frame_b))
        r.close()
        out["two_frames_in_flight"] = {"value": round(w * h * n2 / dt2 / 1e6, 2), "unit": "Mpixel/s", "ms_per_step": round(dt2 / n2 * 1e3, 4), "steps": n2,
                                       "both_frames_identical": same,
                                       "note": "the static frame on alternate HIP streams and frame buffers, nothing waits in between: throughput of a host that keeps two "
                                               "frames in flight (kernels overlap: not the per-kernel rate the roofline is priced on)"}
    except Exception as e:      # noqa: BLE001
        out["two_frames_in_flight"] = {"note": "failed: %r" % (e,)}
    # (b) the camera moves before every frame
    r = rt_host.Renderer(blob, dev_index, lib)
    n_cam = 64
    cams = [moving_camera(scene, k, n_cam) for k in range(n_cam)]
    for k in range(8):
        r.set_camera(cams[k], stream=stream); r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        r.set_camera(cams[k % n_cam], stream=stream)
        r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # one sampled camera against the oracle's rows
    k = 17 % n_cam
    r.set_camera(cams[k], stream=stream); r.render_tiles(w, h, frame.data_ptr(), whole, stream=stream)
    torch.cuda.synchronize()
    rows = sorted(set(int((j + 0.5) * h / 6) for j in range(6)))
    moved = dict(scene, camera=cams[k])
    want = np.frombuffer(ou.c_oracle_rows(rt_host.flatten_scene(moved), w, h, rows), dtype=np.uint8)
    worst = int(ou.max_lsb(np.ascontiguousarray(frame.cpu().numpy()[rows]).reshape(-1), want)[0])
    r.close()
    # (c) the same with TWO frames in flight: two resident copies of the scene and two frame buffers, consecutive frames on alternate
    # HIP streams - frame k+1's table build runs beside frame k's trace (what an animation host that double-buffers its frames
    # does; the library itself is unchanged: one stream per scene handle).  Both copies' last frames are checked against the oracle.
    two = None
    try:
        rr = [rt_host.Renderer(blob, dev_index, lib) for _ in range(2)]
        ss = [stream, torch.cuda.Stream(device=torch.device("cuda", dev_index)).cuda_stream]
        ff = [frame, torch.empty_like(frame)]
        for k2 in range(16):
            rr[k2 & 1].set_camera(cams[k2 % n_cam], stream=ss[k2 & 1]); rr[k2 & 1].render_tiles(w, h, ff[k2 & 1].data_ptr(), whole, stream=ss[k2 & 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k2 in range(steps):
            rr[k2 & 1].set_camera(cams[k2 % n_cam], stream=ss[k2 & 1])
            rr[k2 & 1].render_tiles(w, h, ff[k2 & 1].data_ptr(), whole, stream=ss[k2 & 1])
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        worst2 = 0
        for i in range(2):
            rr[i].set_camera(cams[k], stream=ss[i]); rr[i].render_tiles(w, h, ff[i].data_ptr(), whole, stream=ss[i])
            torch.cuda.synchronize()
            worst2 = max(worst2, int(ou.max_lsb(np.ascontiguousarray(ff[i].cpu().numpy()[rows]).reshape(-1), want)[0]))
        for x in rr:
            x.close()
        two = {"value": round(w * h * steps / dt2 / 1e6, 2), "unit": "Mpixel/s", "ms_per_step": round(dt2 / steps * 1e3, 4), "max_lsb_vs_oracle_rows": worst2,
               "note": "two frames in flight: two resident copies of the scene, alternate HIP streams and frame buffers; a frame's table build runs beside its predecessor's trace"}
    except Exception as e:      # noqa: BLE001
        two = {"note": "failed: %r" % (e,)}
    out["new_camera_every_step"] = {"value": round(w * h * steps / dt / 1e6, 2), "unit": "Mpixel/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
                                    "two_frames_in_flight": two,
                                    "max_lsb_vs_oracle_rows": worst, "parity_ok": worst <= 1,
                                    "note": "per step: rt_scene_set_camera (lookAt on a slow orbit, %d cameras; one small copy) + render: launch table rebuilt on the GPU, trace, "
                                            "list-driven strict launch; frames stay in HBM; nothing waits between steps; camera %d checked against oracle/rt_oracle.c rows %s" % (n_cam, k, rows)}
    return out


def launch_ranks(n, argv):
    """`bench.py --gpus N` started bare: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <argv>` as a child
    (one rank per GPU; rendezvous on 127.0.0.1, a free port), hand its stdout - rank 0's one JSON line - on, return its exit code.
    Nothing in this process has initialised the GPU by now (importing this file and parsing arguments does not), so the ranks start clean."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)          # stderr goes straight through
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in lines[-1:]:
        sys.stdout.write(l + "\n")
    sys.stdout.flush()
    if r.returncode == 0 and not lines:
        print("bench.py: the ranks ended without a JSON line", file=sys.stderr)
        return 1
    return r.returncode


XGMI_GBS_PER_DIRECTION = 64.0       # what a kernel's stores / an RCCL copy sustain over ONE xGMI link in one direction (153 GB/s per link both ways, nominal)
LAUNCH_FLOOR_MS = 0.006             # a launch's fixed cost on the stream (cfg1's 256x256 frame: 0.007 ms)


def predicted_single_frame(n, w, h, kernel_ms_n1, bytes_per_pixel, sky_fraction):
    """What ONE frame row-tiled over n GPUs and put together on rank 0 should cost per step, stated BEFORE the first run on real links, so
    that the run is judged against an expectation: every rank renders 1/n of the frame (its kernel time shrinks to 1/n of the one-GPU
    kernel plus a launch's fixed cost), rank g > 0 ships its share over ITS OWN link to rank 0 (xGMI is point to point), steps are
    pipelined, so a step costs the larger of the two.  All of rank 0's inbound bytes land in one GPU's links: the form is link-bound
    by construction at every n for frames a GPU renders faster than a link carries them."""
    render = kernel_ms_n1 / n + LAUNCH_FLOOR_MS
    link = (w * h * bytes_per_pixel * (1.0 - sky_fraction) / n) / (XGMI_GBS_PER_DIRECTION * 1e9) * 1e3
    step = max(render, link)
    return {"ms_per_step": round(step, 4), "render_ms": round(render, 4), "link_ms": round(link, 4),
            "mpixel_per_s": round(w * h / step / 1e3, 1), "efficiency_vs_n1": round((kernel_ms_n1 / step) / n, 3),
            "bound": "link into rank 0" if link >= render else "render"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)     # 0.25 s timed at N=1: start-up and the final sync (~0.8 ms) no longer show
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3", help="which BASELINE config (default: the headline, configs[2])")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--strict-fp", action="store_true", help="time the no-FMA kernel variant instead")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (traffic / measured FP64 come from profiles/ then)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-frame and moving-camera measurements (N=1)")
    args = ap.parse_args()
    cfg_scene, cfg_w, cfg_h, single = CONFIGS[args.config]
    scene_name = args.scene or cfg_scene
    w, h = args.width or cfg_w, args.height or cfg_h
    if args.config != "cfg3" and "--steps" not in sys.argv:
        args.steps = 400 if args.config == "cfg4" else 20

    # `python3 bench.py --gpus N` with N > 1 and no rank environment: start the N ranks ourselves.  A CHILD process (never an exec),
    # started before anything here has touched the GPU; its one JSON line is relayed, its exit code is ours.
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a "Hostname / Librccl path" banner
    # on stdout when a communicator is created), so file descriptor 1 is pointed at stderr for the whole run and the JSON
    # line alone goes to the real stdout, kept in `json_fd`.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # the pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails
    # (hipIpcGetMemHandle: invalid argument).  Already exported on the boxes; kept here so a bare launch works too.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import rt_host
    import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    if world > 1 and os.environ.get("RT_BENCH_REHEARSE") != "1" and torch.cuda.device_count() < world:      # (counting devices initialises nothing)
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s); RT_BENCH_REHEARSE=1 rehearses the N>1 control flow on one GPU (never a measurement)"
                         % (world, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the render path has no CPU fallback")
    # RT_BENCH_REHEARSE=1: rehearsal of the N>1 control flow on a ONE-GPU box — every rank shares GPU 0 and
    # the exchange goes over gloo through host memory.  Never a measurement (the JSON says so).
    rehearse = world > 1 and os.environ.get("RT_BENCH_REHEARSE") == "1"
    # RT_BENCH_FORCE_EXCHANGE=1 (N=1 only): run the N>1 code path with ONE rank - RGB24 tiles, a real 1-rank RCCL
    # collective (a self copy), the side stream, the de-interleave.  It measures what the plan itself costs (host
    # issue time, copy kernels and de-interleave competing with the render) on a one-GPU box; the JSON says so.
    force = world == 1 and os.environ.get("RT_BENCH_FORCE_EXCHANGE") == "1"
    multi = world > 1 or force
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ctl_dev = "cpu" if rehearse else dev          # where the small control tensors of the collectives live
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def run(single):
        """One measurement: batch mode (single = False: N frames per step, frame f whole on rank f) or single-frame mode (True: ONE frame
        per step, row-tiled over the ranks, whole on rank 0).  Returns (the JSON object on rank 0, parity_ok, max_lsb)."""
        result = None
        scene = rt_host.load_scene(scene_name)
        ss = scene.get("supersample", 1)
        lib = rt_host.load_library()
        renderer = rt_host.Renderer(scene, dev_index, lib)          # scene resident in HBM from here on
        flags = rt_host.RT_FLAG_STRICT_FP if args.strict_fp else 0

        # Which ranks end up owning whole frames: every rank (batch mode: frame f of a step on rank f) or rank 0 alone
        # (single-frame mode: the one frame of a step).
        owners = [0] if (single and multi) else list(range(world))
        i_own = rank in owners
        frames_per_step = len(owners)

        # Which plan reassembles the frames (module docstring).  The exchange plan is always set up unless peer stores are forced.
        p2p_env = os.environ.get("RT_BENCH_P2P")
        if p2p_env == "auto":                                                         # set both up and calibrate, at any N
            p2p_env = None
            p2p = multi
        else:
            # (default since round 3: both plans at every N > 1, the faster one measured - the peer stores carry half the bytes now)
            p2p = multi and (p2p_env == "1" or p2p_env is None)
        a2a_channels = 3 if (multi and w % 4 == 0 and os.environ.get("RT_BENCH_RGBA_EXCHANGE") != "1") else 4
        plan = shard.TilePlan(w, h, TILE_ROWS, world, a2a_channels)
        batch_flags = flags | (rt_host.RT_FLAG_RGB24 if a2a_channels == 3 else 0)
        # Single-frame mode, exchange plan: COMPACT bands (RT_FLAG_COMPACT) - only the blocks a rank stores at all, back to back, cross the
        # links; rank 0 puts them back (rt_compact_expand_device, its own tables of every rank's tile set) and fills the sky itself.  The
        # collective needs one message size: the largest rank's, known before the first step (rt_compact_count).  Scenes the strict kernel
        # renders, 3x3 / 4x4 supersampling, RT_BENCH_NO_COMPACT=1: plain RGB24 bands as before.
        compact = None
        if multi and single and a2a_channels == 3 and not args.strict_fp:
            mine = None
            if os.environ.get("RT_BENCH_NO_COMPACT") != "1":
                try:
                    mine = renderer.compact_count(w, h, rt_host.RtTiles(*plan.rt_tiles(rank)))
                except rt_host.RtError:
                    mine = None
            counts = [None] * world
            dist.all_gather_object(counts, mine)
            if all(c is not None for c in counts):
                bb = counts[0][1]
                compact = {"blocks": [c[0] for c in counts], "block_bytes": bb, "msg": ((max(c[0] for c in counts) * bb + 255) // 256) * 256}
        # a dedicated (non-null) HIP stream, made torch's current stream: the kernel launches, the
        # torch.cuda.Events that time them and c10d's stream dependencies all refer to this one stream
        # N>1: the render stream gets HIGH priority, so the exchange's copy kernels and the de-interleave (normal priority,
        # memory-bound) fill in around the render instead of competing with it for wave slots (RT_BENCH_NO_PRIORITY=1: A/B)
        hi_prio = multi and os.environ.get("RT_BENCH_NO_PRIORITY") != "1"
        tstream = torch.cuda.Stream(device=dev, priority=-1) if hi_prio else torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
        assert stream != 0
        # N>1: ONE exchange serves `every` consecutive steps (RT_BENCH_EXCHANGE_EVERY, default 4; 1 for the 1 GiB frames of cfg5)
        every = max(1, int(os.environ.get("RT_BENCH_EXCHANGE_EVERY", "1" if w * h >= (1 << 27) else "4"))) if multi else 1
        # the frames this rank reassembles (N=1: the frame); ranks that own nothing keep a token allocation
        frames = torch.empty((every, h, w, 4) if i_own else (1, 1, 1, 4), dtype=torch.uint8, device=dev)
        frame = frames[0]
        # Opt-in (RT_BENCH_TWO_STREAMS=1, N=1): consecutive frames alternate between two HIP streams and two frame
        # buffers, so the tail of frame k overlaps the head of frame k+1 (+3 % measured).  Off by default so that
        # every launch of the timed region runs alone and rocprof's per-kernel average equals `kernel_ms`.
        two_streams = not multi and os.environ.get("RT_BENCH_TWO_STREAMS") == "1"
        frame_b = torch.empty((h, w, 4), dtype=torch.uint8, device=dev) if two_streams else None
        tstream_b = torch.cuda.Stream(device=dev) if two_streams else None
        whole = rt_host.RtTiles(h, 0, 1, 1)
        my_tiles = rt_host.RtTiles(*plan.rt_tiles(rank))
        frame_bytes = w * h * 4
        import oracle_util as ou

        def reference_rows():
            """(rows, expected bytes) to check a reassembled frame against: the rows the reference itself rendered (tests/golden) when
            this frame size has them, else a few rows from the oracle's C restatement (checker only, untimed)."""
            for f in ou.manifest()["frames"]:
                if f["scene"] == scene_name and (f["w"], f["h"]) == (w, h) and f["rows"]:
                    return f["rows"], ou.golden_frame(f), "tests/golden/%s (rendered by the reference itself)" % f["file"]
            rows = sorted(set(int((k + 0.5) * h / 5) for k in range(5)))
            blob = rt_host.flatten_scene(scene)
            return rows, np.frombuffer(ou.c_oracle_rows(blob, w, h, rows), dtype=np.uint8), "oracle/rt_oracle.c rows %s" % rows

        check_rows, check_bytes, check_source = reference_rows()

        def worst_lsb(host_frame):
            return int(ou.max_lsb(np.ascontiguousarray(host_frame[check_rows]).reshape(-1), check_bytes)[0])

        p2p_note = None
        my_buf, peer_buf = None, {}
        # peer stores: the senders leave out the blocks in which only the constant background can show (RT_FLAG_NO_SKY), the owner of
        # a frame stores them itself (RT_FLAG_SKY_ONLY): half of the headline's pixels never cross a link (RT_BENCH_SEND_SKY=1: A/B)
        sky_out = 0 if (os.environ.get("RT_BENCH_SEND_SKY") == "1" or ss > 2) else rt_host.RT_FLAG_NO_SKY
        if p2p:
            # Set-up and PRE-FLIGHT of the peer-store plan; anything that goes wrong on any rank (no IPC, no peer access, a frame
            # that does not match the reference's rows) makes every rank fall back to the exchange plan, and the JSON says so.
            import ctypes as C
            problem, mine = None, None
            try:
                if i_own:
                    my_buf = lib.rt_alloc_device(dev_index, 2 * every * frame_bytes)      # [slot][step of the group] whole frames
                    if not my_buf:
                        raise RuntimeError("rt_alloc_device: " + lib.rt_last_error().decode())
                    hnd = C.create_string_buffer(64)
                    if lib.rt_ipc_export(dev_index, my_buf, hnd) != 0:
                        raise RuntimeError("rt_ipc_export: " + lib.rt_last_error().decode())
                    mine = hnd.raw
                if os.environ.get("RT_BENCH_P2P_INJECT_FAILURE") == str(rank):     # test hook for the fallback
                    raise RuntimeError("injected failure on rank %d" % rank)
            except Exception as e:      # noqa: BLE001
                problem, mine = repr(e), None
            handles = [None] * world
            dist.all_gather_object(handles, (problem, mine))
            if problem is None:
                bad = [g for g in range(world) if handles[g][0] is not None or (g in owners and handles[g][1] is None)]
                if bad:
                    problem = "rank %d: %s" % (bad[0], handles[bad[0]][0] or "could not export its buffer")
            if problem is None:
                try:
                    for g in owners:
                        if g == rank:
                            peer_buf[g] = my_buf
                        else:
                            q = C.c_void_p()
                            hb = C.create_string_buffer(handles[g][1], 64)
                            if lib.rt_ipc_open(dev_index, hb, C.byref(q)) != 0:
                                raise RuntimeError("rt_ipc_open(rank %d): %s" % (g, lib.rt_last_error().decode()))
                            peer_buf[g] = q.value
                except Exception as e:      # noqa: BLE001
                    problem = repr(e)
            oks = [None] * world
            dist.all_gather_object(oks, problem)
            if all(x is None for x in oks):
                # pre-flight: one step through the peer stores, then every owner checks the frame it owns
                if i_own:
                    lib.rt_memset_device(dev_index, my_buf, 0, frame_bytes)
                dist.barrier()
                renderer.render_scatter(w, h, [peer_buf[g] for g in owners], my_tiles, flags=flags | sky_out, want_stats=True)     # returns when the launch is done
                if i_own and sky_out:      # the sky blocks of the whole frame: the owner's own work, from its own launch table
                    renderer.render_scatter(w, h, [my_buf], whole, flags=flags | rt_host.RT_FLAG_SKY_ONLY, want_stats=True)
                dist.barrier()
                worst = 0
                if i_own:
                    host = np.empty((h, w, 4), dtype=np.uint8)
                    lib.rt_copy_to_host(dev_index, host.ctypes.data, my_buf, frame_bytes)
                    worst = worst_lsb(host)
                    if not (host[..., 3] == 255).all():
                        worst = max(worst, 255)                                # a row nobody wrote
                    del host
                worsts = [None] * world
                dist.all_gather_object(worsts, worst)
                if max(worsts) > 1:
                    oks = ["pre-flight frame differs from the reference's rows by %d LSB" % max(worsts)]
            if not all(x is None for x in oks):
                p2p_note = "peer-store plan not usable (%s): exchange plan used instead" % next(x for x in oks if x is not None)
                if rank == 0:
                    print("bench.py: " + p2p_note, file=sys.stderr, flush=True)
                for g, q in peer_buf.items():
                    if g != rank:
                        lib.rt_ipc_close(dev_index, q)
                peer_buf = {}
                dist.barrier()
                if my_buf:
                    lib.rt_free_device(dev_index, my_buf)
                    my_buf = None
                p2p = False
            else:
                token = torch.zeros(1, dtype=torch.int32, device=ctl_dev)
        # `mode["p2p"]`: the plan the step machinery below uses right now (the calibration switches it back and forth)
        mode = {"p2p": p2p}
        have_exchange = multi and not (p2p and p2p_env == "1")
        if have_exchange:
            if not single:
                # [destination rank][step of the group][band]: what all_to_all_single sends to rank g is send[g], contiguous
                send = [torch.empty((world, every, plan.band_rows, w, a2a_channels), dtype=torch.uint8, device=dev) for _ in range(2)]
                recv = [torch.empty((world, every, plan.band_rows, w, a2a_channels), dtype=torch.uint8, device=dev) for _ in range(2)]
                host_recv = torch.empty((world, every, plan.band_rows, w, a2a_channels), dtype=torch.uint8) if rehearse else None
            elif compact:
                # [step of the group][the rank's blocks, padded to the largest rank's] on every rank; rank 0 gathers [source rank][step][blocks]
                send = [torch.zeros((every, compact["msg"]), dtype=torch.uint8, device=dev) for _ in range(2)]
                recv = [torch.empty((world, every, compact["msg"]), dtype=torch.uint8, device=dev) if i_own else None for _ in range(2)]
                host_recv = torch.empty((world, every, compact["msg"]), dtype=torch.uint8) if (rehearse and i_own) else None
            else:
                # [step of the group][band] on every rank; rank 0 gathers [source rank][step][band]
                send = [torch.empty((every, plan.band_rows, w, a2a_channels), dtype=torch.uint8, device=dev) for _ in range(2)]
                recv = [torch.empty((world, every, plan.band_rows, w, a2a_channels), dtype=torch.uint8, device=dev) if i_own else None for _ in range(2)]
                host_recv = torch.empty((world, every, plan.band_rows, w, a2a_channels), dtype=torch.uint8) if (rehearse and i_own) else None
        pending = []      # (work, slot, steps in it, plan) of exchanges in flight; at most 2
        group = {"slot": 0, "fill": 0, "last_slot": 0}                 # the exchange buffer being filled, and how many steps are in it
        # the wait for an exchange and the de-interleaves that follow run on a SIDE stream, so the render stream
        # never stalls behind communication; an event per slot tells the render stream when a slot may be reused
        side = torch.cuda.Stream(device=dev) if multi else None
        slot_free = [torch.cuda.Event() for _ in range(2)] if multi else None

        def render_step(slot, j=0):
            if not multi:
                renderer.render_tiles(w, h, frame.data_ptr(), whole, stream=stream, flags=flags)
            elif mode["p2p"]:   # frame of owner o in this step -> o's buffer, slot `slot`, position j; rows in frame order
                off = (slot * every + j) * frame_bytes
                renderer.render_scatter(w, h, [peer_buf[g] + off for g in owners], my_tiles, stream=stream, flags=flags | sky_out)
                if i_own and sky_out:      # the sky blocks of the whole frame: the owner's own work (nothing of them crosses a link)
                    renderer.render_scatter(w, h, [my_buf + off], whole, stream=stream, flags=flags | rt_host.RT_FLAG_SKY_ONLY)
            elif not single:    # this rank's tiles of the `world` frames of this step, one launch: frame f -> send[slot][f, j]
                renderer.render_batch(w, h, send[slot][0, j].data_ptr(), my_tiles, world, every * plan.band_bytes, stream=stream, flags=batch_flags)
            elif compact:       # this rank's blocks of the step's one frame, the sky left out, back to back -> send[slot][j]
                renderer.render_batch(w, h, send[slot][j].data_ptr(), my_tiles, 1, 0, stream=stream, flags=batch_flags | rt_host.RT_FLAG_NO_SKY | rt_host.RT_FLAG_COMPACT)
            else:               # this rank's tiles of the step's one frame -> send[slot][j]
                renderer.render_batch(w, h, send[slot][j].data_ptr(), my_tiles, 1, 0, stream=stream, flags=batch_flags)

        def finish(item):
            work, slot, count, was_p2p = item
            with torch.cuda.stream(side):
                work.wait()                                          # the side stream waits for the exchange (p2p: the barrier)
                if was_p2p or not i_own:
                    count = 0                                        # the frames are already whole, in place (or live elsewhere)
                elif rehearse:
                    recv[slot].copy_(host_recv)
                for j in range(count):                               # one whole frame per step of the group ends up on an owner
                    if compact and single:
                        # the sky of the whole frame from this rank's own table, then every rank's blocks to their places
                        renderer.render_scatter(w, h, [frames[j].data_ptr()], whole, stream=side.cuda_stream, flags=flags | rt_host.RT_FLAG_SKY_ONLY)
                        for g in range(world):
                            renderer.compact_expand(w, h, rt_host.RtTiles(*plan.rt_tiles(g)), recv[slot][g, j].data_ptr(), frames[j].data_ptr(), stream=side.cuda_stream)
                    else:
                        shard.deinterleave(plan, recv[slot][:, j], frames[j], lib=lib, device_index=dev_index, stream=side.cuda_stream)
                slot_free[slot].record(side)
            tstream.wait_event(slot_free[slot])                      # ordering only: that work is two groups old by the time it matters

        def launch_exchange():
            slot = group["slot"]
            if mode["p2p"]:
                if rehearse:
                    torch.cuda.synchronize()                          # gloo knows nothing of the GPU: finish the stores first
                work = dist.all_reduce(token, async_op=True)          # after every rank's stores of this group (stream order)
            elif not single:
                work = shard.exchange_bands(send[slot].cpu(), host_recv, async_op=True) if rehearse else shard.exchange_bands(send[slot], recv[slot], async_op=True)
            elif rehearse:
                work = shard.gather_bands(send[slot].cpu(), host_recv, dst=0, async_op=True)
            else:
                work = shard.gather_bands(send[slot], recv[slot], dst=0, async_op=True)
            pending.append((work, slot, group["fill"], mode["p2p"]))   # overlaps with the renders of the next group
            group["last_slot"] = slot
            group["slot"], group["fill"] = slot ^ 1, 0

        def step(k):
            if not multi:
                if two_streams and (k & 1):
                    renderer.render_tiles(w, h, frame_b.data_ptr(), whole, stream=tstream_b.cuda_stream, flags=flags)
                else:
                    render_step(0)
                return
            if group["fill"] == 0 and len(pending) == 2:             # about to refill a slot: the exchange that last used it
                finish(pending.pop(0))
            render_step(group["slot"], group["fill"])
            group["fill"] += 1
            if group["fill"] == every:
                launch_exchange()

        def drain():
            if multi and group["fill"] > 0:                      # a partial group still travels (whole buffer; only its steps count)
                launch_exchange()
            while pending:
                finish(pending.pop(0))

        def fence():
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
                torch.cuda.synchronize()

        # initialisation, not a warm-up step: the first launch loads the code object, the first exchange builds the
        # RCCL communicator (seconds) — both must be out of the way even when the caller asks for --warmup 0
        step(0)
        drain()
        fence()

        # What ONE of these GPUs does with the same frame on its own (rank 0, the others wait): the reference of efficiency_vs_n1
        n1_mpix = None
        if world > 1:
            if rank == 0:
                tmp = frame if (i_own and frames.shape[1] == h) else torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
                k1 = max(10, min(400, int(0.05 / max(1e-6, w * h / 1e11))))
                for _ in range(5):
                    renderer.render_tiles(w, h, tmp.data_ptr(), whole, stream=stream, flags=flags)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k1):
                    renderer.render_tiles(w, h, tmp.data_ptr(), whole, stream=stream, flags=flags)
                torch.cuda.synchronize()
                n1_mpix = w * h * k1 / (time.perf_counter() - t1) / 1e6
            fence()

        # Both plans are set up (nothing forced): time a few groups of steps with each and keep the faster one.  The
        # slower rank decides (MAX over ranks), so every rank makes the same choice.
        calibration = None
        if multi and p2p and p2p_env is None:
            def timed(n):
                fence()
                t0 = time.perf_counter()
                for k in range(n):
                    step(k)
                drain()
                torch.cuda.synchronize()
                tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ctl_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt.item())
            calibration = {}
            n_cal = max(every, min(8 * every, int(0.25 / max(1e-4, w * h / 6e10 / max(1, world if single else 1)))))   # ~0.25 s per plan at most
            for name, flag in (("exchange", False), ("peer_stores", True)):
                mode["p2p"] = flag
                timed(every)                                   # this plan's own first-use costs
                calibration[name] = round(timed(n_cal) / n_cal * 1e3, 4)      # ms per step
            mode["p2p"] = calibration["peer_stores"] <= calibration["exchange"]
            fence()

        # ---- steady clocks first: trains of launches until two consecutive trains agree within 1 % and at least 50 ms have passed
        #      (a GPU that has just been idle runs its first milliseconds below its sustained clock: a 5-step warm-up followed
        #      by 20 timed steps would measure that transient, not the kernel) ----
        settle = {"trains": 0, "ms": 0.0, "last_two_ms_per_step": None}
        if os.environ.get("RT_BENCH_NO_SETTLE") != "1":
            n_train = max(every, min(64, int(0.01 / max(1e-5, w * h / 6e10)) or 1))       # ~10 ms of launches per train
            n_train = (n_train + every - 1) // every * every
            prev, t_begin = None, time.perf_counter()
            for _ in range(40):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(n_train):
                    step(k)
                drain()
                torch.cuda.synchronize()
                cur = (time.perf_counter() - t0) / n_train
                settle["trains"] += 1
                stop = prev is not None and abs(cur - prev) <= 0.01 * prev and (time.perf_counter() - t_begin) >= 0.05
                settle["last_two_ms_per_step"] = [round(1e3 * x, 4) for x in (prev if prev is not None else cur, cur)]
                prev = cur
                if multi:                                            # every rank must leave the loop together
                    tt = torch.tensor([0 if stop else 1], dtype=torch.int32, device=ctl_dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    stop = int(tt.item()) == 0
                if stop:
                    break
            settle["ms"] = round(1e3 * (time.perf_counter() - t_begin), 1)
            fence()

        for k in range(args.warmup):
            step(k)
        drain()
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        drain()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        fence()
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

        channels = 4 if (not multi or mode["p2p"]) else a2a_channels          # bytes per pixel the chosen plan's launches store
        # ---- dominant kernel: average launch duration, HIP events on the launch stream.  At N=1 the timed region IS a train of
        #      these launches on this stream, so two events around a train of them give the average over every launch (what
        #      rocprofv3's per-kernel average of the same command shows); at N>1 the region also waits for slots, so the kernel is
        #      timed on a train of its own ----
        fence()
        n_train = max(20, min(args.steps, 2000)) if not multi else min(max(20, args.steps), 200)
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(min(10, n_train)):
            render_step(0)
        ev_a.record()
        for _ in range(n_train):
            render_step(0)
        ev_b.record()
        torch.cuda.synchronize()
        kernel_ms = ev_a.elapsed_time(ev_b) / n_train
        launch_pixels = w * h if not multi else frames_per_step * plan.pixels_of(rank)

        # one more (untimed) group of steps; EVERY owner checks the frames it reassembled against the rows the reference itself
        # rendered (tests/golden, fixtures) or, for a frame size without fixtures, a few rows of the C restatement
        fence()
        frames.zero_()
        if p2p and i_own and lib.rt_memset_device(dev_index, my_buf, 0, 2 * every * frame_bytes) != 0:
            raise SystemExit("bench.py: rt_memset_device: " + lib.rt_last_error().decode())
        fence()                                                        # nobody stores into a buffer that is still being cleared
        for k in range(every):                                         # one whole group, so every frame slot is rewritten
            step(k)
        drain()
        fence()

        def reassembled(j):
            if not mode["p2p"]:
                return frames[j].cpu().numpy()
            host = np.empty((h, w, 4), dtype=np.uint8)
            if lib.rt_copy_to_host(dev_index, host.ctypes.data, my_buf + (group["last_slot"] * every + j) * frame_bytes, frame_bytes) != 0:
                raise SystemExit("bench.py: rt_copy_to_host: " + lib.rt_last_error().decode())
            return host

        max_lsb = 0
        if i_own:
            for j in range(every):
                fr = reassembled(j)
                max_lsb = max(max_lsb, worst_lsb(fr))
                if not (fr[..., 3] == 255).all():
                    max_lsb = 255                                          # a row nobody wrote
                del fr
        if world > 1:
            t = torch.tensor([int(max_lsb)], dtype=torch.int64, device=ctl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            max_lsb = int(t.item())
        parity_ok = max_lsb <= 1          # tolerance: 1 LSB per channel
        if rank == 0:
            # work counters from the instrumented variant (untimed)
            cnt_frame = frame if i_own and frames.shape[1] == h else torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
            st = renderer.render_tiles(w, h, cnt_frame.data_ptr(), whole, stream=stream, flags=flags | rt_host.RT_FLAG_COUNT, want_stats=True)
            rays_pp, shadow_pp, tests_pp = st.rays / st.pixels, st.shadow_rays / st.pixels, st.sphere_tests / st.pixels
            total_pixels = w * h * frames_per_step * args.steps
            value = total_pixels / elapsed / 1e6
            flops_pp = 15.0 * tests_pp + 120.0 * rays_pp + 60.0 * shadow_pp        # SURVEY §8(d) algorithmic FP64 flop model
            algo_bytes = float(channels) * launch_pixels        # what one launch stores: RGBA8, or RGB24 bands at N>1
            if compact and single and not mode["p2p"]:
                algo_bytes = float(compact["blocks"][rank] * compact["block_bytes"])      # ... or only this rank's blocks (a compact band)
            achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
            # HBM traffic and executed FP64 instructions per launch: PMC passes over a short child run of this same command
            traffic, traffic_src, fp64_measured = None, None, None
            pmc, pmc_note = (None, "not collected (--no-pmc)") if (args.no_pmc or multi or os.environ.get("RT_BENCH_CHILD") == "1") else pmc_passes(
                ["--config", args.config] + (["--scene", scene_name] if args.scene else [])
                + (["--width", str(w)] if args.width else []) + (["--height", str(h)] if args.height else []) + (["--strict-fp"] if args.strict_fp else []))
            if pmc:
                traffic = pmc["WRITE_SIZE"] * 1024.0 + 2.0 * pmc["FETCH_SIZE"] * 1024.0          # KiB -> bytes; FETCH_SIZE counts half (MI355X_MICROARCH.md, HBM)
                traffic_src = "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes) over a 12-step child run of this command; FETCH_SIZE doubled (gfx950)"
                fl = (2.0 * pmc["SQ_INSTS_VALU_FMA_F64"] + pmc["SQ_INSTS_VALU_ADD_F64"] + pmc["SQ_INSTS_VALU_MUL_F64"]) * 64.0
                fp64_measured = {"valu_insts_per_launch": pmc["SQ_INSTS_VALU"], "salu_insts_per_launch": pmc["SQ_INSTS_SALU"],
                                 "fma_f64": pmc["SQ_INSTS_VALU_FMA_F64"], "add_f64": pmc["SQ_INSTS_VALU_ADD_F64"], "mul_f64": pmc["SQ_INSTS_VALU_MUL_F64"],
                                 "trans_f64": pmc["SQ_INSTS_VALU_TRANS_F64"], "flop_per_launch_upper_bound": fl,
                                 "achieved": round(fl / (kernel_ms * 1e-3) / 1e12, 3), "frac": round(fl / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, 4),
                                 "note": "wave-instructions x 64 lanes (an upper bound: assumes a full exec mask), FMA = 2 flop; this run's counters, this run's kernel_ms"}
                if "SQ_BUSY_CYCLES" in pmc and pmc["SQ_BUSY_CYCLES"] > 0:
                    # issue fractions: cycles in which a vector / scalar / any instruction was issued, per busy SQ cycle (SQ_ACTIVE_INST_* are
                    # summed over an SQ's SIMDs, so VALU is also given per SIMD) and per resident wave-cycle
                    fp64_measured["issue"] = {"valu_per_busy_cycle": round(pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_BUSY_CYCLES"], 4),
                                              "valu_per_busy_cycle_per_simd": round(pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_BUSY_CYCLES"] / 4.0, 4),
                                              # what bounds the kernel: a 64-wide wave's VALU instruction occupies its 16-lane SIMD for 4 cycles.
                                              # (a) against the SQ's own busy cycles (SQ_BUSY_CYCLES sums 32 instances - one per shader engine of each
                                              # XCD, 32 SIMDs each - SQ_ACTIVE_INST_VALU counts issued instructions); (b) against this run's kernel_ms
                                              # at the 2.4 GHz peak clock (a lower bound: under FP64 load the part clocks lower)
                                              "valu_busy_frac_of_sq_busy_cycles": round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (pmc["SQ_BUSY_CYCLES"] * 32.0), 4),
                                              "valu_busy_frac_at_peak_clock": round(pmc["SQ_INSTS_VALU"] * 4.0 / (1024.0 * kernel_ms * 1e-3 * 2.4e9), 4),
                                              "scalar_per_busy_cycle": round(pmc["SQ_ACTIVE_INST_SCA"] / pmc["SQ_BUSY_CYCLES"], 4),
                                              "any_per_wave_cycle": round(pmc["SQ_ACTIVE_INST_ANY"] / pmc["SQ_WAVE_CYCLES"], 4) if pmc.get("SQ_WAVE_CYCLES") else None,
                                              "counters": {k: pmc[k] for k in ("SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY") if k in pmc}}
                elif "issue_note" in pmc:
                    fp64_measured["issue"] = {"note": pmc["issue_note"]}
            else:
                tpath = os.path.join(ROOT, "profiles", "traffic.json")
                key = "%s_%dx%d" % (scene_name, w, h)
                if os.path.exists(tpath) and world == 1:
                    tj = json.load(open(tpath))
                    if key in tj:
                        traffic = tj[key]["hbm_bytes_per_launch"]
                        traffic_src = "REPLAYED from profiles/traffic.json (%s): %s" % (tj[key].get("source", "committed profile"), pmc_note)
                        if "fp64" in tj[key]:
                            fl = tj[key]["fp64"]["flop_per_launch_upper_bound"]
                            fp64_measured = dict(tj[key]["fp64"], achieved=round(fl / (kernel_ms * 1e-3) / 1e12, 3),
                                                 frac=round(fl / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, 4),
                                                 note="counters REPLAYED from profiles/traffic.json, this run's kernel_ms")
            if not multi:
                how = "one launch per frame" + ("; consecutive frames alternate between two HIP streams and two frame buffers" if two_streams else "")
            elif mode["p2p"]:
                how = ("a step = %s: interleaved %d-row tiles over %d ranks, one launch per rank whose stores go straight into the "
                       "frame buffer of the rank that owns each frame (peer-mapped over xGMI, RGBA8, rows in place): no data collective, no "
                       "de-interleave; one all_reduce per %d steps is the barrier"
                       % ("ONE frame, whole on rank 0" if single else "a batch of %d frames" % world, TILE_ROWS, world, every))
            else:
                how = ("a step = %s: interleaved %d-row tiles over %d ranks, one launch per rank, ONE %s (RCCL over xGMI) "
                       "reassembles %s (bands travel as %s), de-interleave to RGBA8 in HBM; the bands of %d consecutive steps share one "
                       "collective, which overlaps the renders of the next %d steps"
                       % ("ONE frame" if single else "a batch of %d frames" % world, TILE_ROWS, world, "gather to rank 0" if single else "all-to-all",
                          "the frame on rank 0" if single else "frame f on rank f", "RGB24, alpha restored on arrival" if channels == 3 else "RGBA8", every, every))
            # the store roofline as this box delivers it (quote a measured fill next to the nominal peak): a 2 GiB
            # torch fill, best of 5, on the launch stream
            fill_gbs = None
            if not multi:
                try:
                    big = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
                    best = None
                    for _ in range(6):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        big.fill_(7)
                        e1.record()
                        torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1)
                        best = ms if best is None else min(best, ms)
                    fill_gbs = round((2 << 30) / (best * 1e-3) / 1e9, 1)
                    del big
                except Exception:      # noqa: BLE001  (a small box: the nominal peak stands alone)
                    fill_gbs = None
            model_tf = flops_pp * launch_pixels / (kernel_ms * 1e-3) / 1e12
            out = {
                "metric": "Mpixel/s", "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if (single and multi) else "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "%s: %s scene (%d spheres, %d lights, depth %d, supersample %d) at %dx%d; SCENE, CAMERA AND LAUNCH TABLE RESIDENT (a static frame rendered again and again: "
                                       "cold_frame / new_camera_every_step are the other cases); %s" % (
                    args.config if (scene_name, w, h) == CONFIGS[args.config][:3] else "custom", scene_name, len(scene["objects"]), len(scene["lights"]),
                    scene["segs"], ss, w, h, how),
                    "kernel": "strict (no FMA)" if args.strict_fp else "fma", "frames_per_step": frames_per_step,
                    "pixels_per_gpu_per_step": (w * h * frames_per_step) // world if multi else w * h,
                    "steady_state_warmup": settle,
                    **({"REHEARSAL": "all ranks on one GPU, gloo through host memory - not a measurement"} if rehearse else {}),
                    **({"FORCED_EXCHANGE": "the N>1 plan run by ONE rank (1-rank RCCL collective = self copy): the plan's own overhead, not the headline"} if force else {})},
                "mray_per_s": round(value * rays_pp, 2), "mshadow_per_s": round(value * shadow_pp, 2),
                "rays_per_pixel": round(rays_pp, 4), "shadow_rays_per_pixel": round(shadow_pp, 4), "sphere_tests_per_pixel": round(tests_pp, 3),
                "max_lsb_vs_reference_rows": max_lsb, "parity_ok": parity_ok, "parity_checked_against": check_source,
                "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                             "traffic": traffic, "traffic_source": traffic_src, "kernel": "rt_trace", "kernel_ms": round(kernel_ms, 4),
                             "algorithmic_bytes_per_launch": algo_bytes,
                             "measured_fill_GBs": fill_gbs, "frac_of_measured_fill": (round(achieved / fill_gbs, 6) if fill_gbs else None),
                             "binding_bound": "fp64 vector issue under divergence (fp64_valu.measured), not HBM: the store roofline is the one the metric names",
                             "note": "%d B per output pixel (one %s store); the path is FP64-VALU bound, see fp64_valu" % (channels, "RGBA8" if channels == 4 else "RGB24")},
                "fp64_valu": {"peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "measured": fp64_measured,
                              "reference_work_rate": {"flop_per_pixel": round(flops_pp, 1), "tflop_per_s_of_reference_work": round(model_tf, 3),
                                                      "note": "NOT a utilisation: the REFERENCE's algorithmic operation count (SURVEY 8(d): 15/test + 120/ray + 60/shadow ray) per pixel x "
                                                              "this kernel's pixel rate; the product kernel does not execute those operations (anchored tests are 4, not 10; sky workgroups, "
                                                              "most floor blocks' shadow scans and the culled tests are never executed) - `measured` is what it executes"},
                              "kernel_mpixel_per_s": round(launch_pixels / (kernel_ms * 1e-3) / 1e6, 1)},
            }
            if n1_mpix:
                out["n1_reference"] = {"mpixel_per_s": round(n1_mpix, 2), "note": "rank 0 alone, the same %dx%d frame, launches back to back, frame in HBM, measured in this run" % (w, h)}
                out["efficiency_vs_n1"] = round(value / (world * n1_mpix), 4)
            if p2p_note:
                out["config"]["plan_note"] = p2p_note
            if calibration:
                out["config"]["plan_calibration_ms_per_step"] = calibration
            if multi and mode["p2p"]:
                sender = rank if any(g != rank for g in owners) else min(world - 1, 1)       # (single-frame mode: rank 0 owns, rank 1 sends)
                remote_px = sum(1 for g in owners if g != sender) * plan.pixels_of(sender)
                sky_frac = sky_block_fraction(lib, rt_host.flatten_scene(scene), w, h, plan.rt_tiles(sender)) if sky_out else 0.0
                out["exchange"] = {"plan": "peer stores (rt_render_scatter_device through IPC-mapped frame buffers); the senders leave the constant-background blocks out "
                                           "(RT_FLAG_NO_SKY), each owner stores them itself (RT_FLAG_SKY_ONLY)" if sky_out else
                                           "peer stores (rt_render_scatter_device through IPC-mapped frame buffers), sky blocks included (RT_BENCH_SEND_SKY=1)",
                                   "collective": "all_reduce of one int per group (barrier)",
                                   "bytes_stored_remotely_per_rank_per_step": int(remote_px * 4 * (1.0 - sky_frac)),
                                   "bytes_stored_remotely_per_rank_per_step_with_the_sky": remote_px * 4, "sky_fraction_of_this_ranks_blocks": round(sky_frac, 4),
                                   "the_rank_these_are_for": sender, "steps_per_barrier": every}
            elif multi:
                out["exchange"] = {"plan": "exchange", "collective": "gather to rank 0" if single else "all_to_all_single",
                                   "bytes_sent_per_rank_per_step": plan.band_bytes if single else (world - 1) * plan.band_bytes,
                                   "bytes_per_directed_link_per_step": plan.band_bytes, "bytes_per_pixel_on_the_link": channels,
                                   "steps_per_collective": every}
                if compact and single:
                    sent = sum(compact["blocks"]) * compact["block_bytes"]
                    out["exchange"].update({"bands": "compact (RT_FLAG_COMPACT): the blocks a rank stores at all, back to back; rank 0 puts them back and fills the sky itself",
                                            "bytes_sent_per_rank_per_step": compact["msg"], "bytes_per_directed_link_per_step": compact["msg"],
                                            "bytes_of_blocks_per_rank": [b * compact["block_bytes"] for b in compact["blocks"]],
                                            "bytes_sent_per_rank_per_step_with_the_sky": plan.band_bytes,
                                            "sky_fraction_left_out": round(1.0 - sent / float(world * plan.band_bytes), 4)})
            if n1_mpix and single:
                # the expectation this form is judged against (stated per N, so that the first run on real links has something to be held to)
                k1 = w * h / n1_mpix / 1e3                                  # one GPU's ms per frame, this run
                bpp = channels
                skyf = out["exchange"].get("sky_fraction_of_this_ranks_blocks", 0.0) if (mode["p2p"] and sky_out) else out["exchange"].get("sky_fraction_left_out", 0.0)
                out["predicted"] = {"model": "step = max(one GPU's frame time / N + %.3f ms launch floor, this rank's share of the frame's bytes / one xGMI link at %.0f GB/s per direction); "
                                             "steps pipelined; rank 0's inbound links carry (N-1)/N of every frame, each sender on its own link" % (LAUNCH_FLOOR_MS, XGMI_GBS_PER_DIRECTION),
                                    "bytes_per_pixel_on_the_link": bpp, "sky_fraction_left_out": round(skyf, 4), "one_gpu_ms_per_frame": round(k1, 4),
                                    "per_n": {str(g): predicted_single_frame(g, w, h, k1, bpp, skyf) for g in (2, 4, 8)}}
                if str(world) in out["predicted"]["per_n"] and not rehearse:
                    out["predicted"]["measured_over_predicted_at_this_n"] = round(out["ms_per_step"] / out["predicted"]["per_n"][str(world)]["ms_per_step"], 3)
            if world == 1 and not args.no_cold and not args.strict_fp and os.environ.get("RT_BENCH_CHILD") != "1" and w * h <= 7680 * 4320:
                try:
                    out.update(cold_and_moving(args, scene, scene_name, w, h, lib, dev_index, stream, torch, np, max(256, min(args.steps, 512))))   # (a pipeline: 256 steps and more, so that its fill and drain do not show)
                except Exception as e:   # noqa: BLE001  (the headline must still be reported)
                    out["cold_frame"] = {"note": "failed: %r" % (e,)}
            if world == 1 and not args.no_cold and not args.strict_fp and args.config == "cfg3" and not args.scene and os.environ.get("RT_BENCH_CHILD") != "1":
                try:
                    out["reference_scene"] = reference_scene_leg(args, w, h, lib, dev_index, stream, torch, np)
                except Exception as e:   # noqa: BLE001  (the headline must still be reported)
                    out["reference_scene"] = {"note": "failed: %r" % (e,)}
            if world == 1 and not args.no_cpu_baseline:
                try:
                    out["cpu_baseline"] = cpu_baseline(scene_name, w, h)
                except Exception as e:   # the GPU number must still be reported
                    out["cpu_baseline"] = {"value": None, "unit": "Mpixel/s", "cores": 1, "kind": "port", "sample": "failed: %s" % e}
            result = out

        renderer.close()
        if p2p:
            fence()
            for g, q in peer_buf.items():
                if g != rank:
                    lib.rt_ipc_close(dev_index, q)
            fence()
            if my_buf:
                lib.rt_free_device(dev_index, my_buf)
        return result, parity_ok, max_lsb


    # cfg3 on several GPUs.  The headline is north_star's own form: ONE 3840x2160 frame per step, row-tiled over the ranks, whole on rank 0
    # after one collective / barrier (strong scaling: the frame is fixed).  The batch form - N frames per step, frame f whole on rank f,
    # every directed link in use - is measured first and rides in the same JSON line as `batch_mode` (weak scaling; never the headline).
    batch = None
    if world > 1 and args.config == "cfg3" and not single and os.environ.get("RT_BENCH_NO_BATCH") != "1":
        batch = run(False)
    out, parity_ok, max_lsb = run(True if (world > 1 and args.config == "cfg3") else single)
    if batch is not None:
        out_b, ok_b, lsb_b = batch
        parity_ok, max_lsb = parity_ok and ok_b, max(max_lsb, lsb_b)
        if rank == 0:
            out["batch_mode"] = {k: out_b[k] for k in ("value", "unit", "ms_per_step", "scaling", "max_lsb_vs_reference_rows", "parity_ok", "n1_reference", "efficiency_vs_n1", "exchange") if k in out_b}
            out["batch_mode"]["workload"] = out_b["config"]["workload"]
            out["batch_mode"]["frames_per_step"] = out_b["config"]["frames_per_step"]
            for k in ("plan_note", "plan_calibration_ms_per_step"):
                if k in out_b["config"]:
                    out["batch_mode"][k] = out_b["config"][k]
    if rank == 0:
        ref_leg = out.get("reference_scene") or {}
        if ref_leg.get("parity_ok") is False:
            parity_ok, max_lsb = False, max(max_lsb, ref_leg["max_lsb_vs_reference_rows"])
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:      # a fast frame that differs from the reference's is not a result
        raise SystemExit("bench.py: reassembled frame differs from the reference's rows by %d LSB (tolerance 1)" % max_lsb)


if __name__ == "__main__":
    main()
