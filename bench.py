#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one frame: every rank renders its interleaved row tiles
of the frame with the HIP kernel (through the C ABI, rt_render_tiles_device), the tiles are gathered
to rank 0 over RCCL (torch.distributed, backend "nccl") and de-interleaved into the final RGBA8
frame in rank 0's HBM.  At N=1 a step is exactly one kernel launch writing the frame.

Workload: BASELINE configs[2] — 3840x2160, the 8-sphere "H8" scene, 2 lights, depth 3 — at N=1.
For N>1 the frame is scaled at constant aspect so that every GPU keeps one 4K frame's worth of
pixels (N=4 is exactly configs[3], 7680x4320): weak scaling.  The scene is resident in HBM before
the timed region; the frame stays in HBM (the PCIe copy-out rate is quoted in DESIGN.md, never here).

Rank 0 prints ONE JSON line.  `roofline` prices the trace kernel against the HBM-store roofline the
metric names (4 algorithmic bytes per pixel) — the path is FP64-VALU bound, so that fraction is
small by construction; the `fp64_valu` object prices it against the binding bound.
`cpu_baseline` is the oracle's JS restatement (bit-identical to main.js, see tests/test_oracle.py)
on one host thread, on a bounded sample of rows of the same frame.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TF = 78.6     # MI355X vector FP64 (FMA) peak = half the 157.3 TF FP32 vector rate
TILE_ROWS = 16


def frame_size_for(n):
    """One 4K frame's worth of pixels per GPU at constant 16:9 aspect; whole 32x8 workgroup tiles."""
    if n == 1:
        return 3840, 2160
    s = math.sqrt(n)
    return int(round(3840 * s / 32)) * 32, int(round(2160 * s / TILE_ROWS)) * TILE_ROWS


def cpu_baseline(scene_name, w, h):
    """Oracle leg (checker code, timed beside the GPU; never on the product path)."""
    import oracle_util as ou
    rows = min(h, 1080)
    cores = 1
    if ou.node_path():
        r = ou.node_cli("time", ou.scene_json(scene_name), w, h, rows, timeout=900)
        return {"value": round(r["mpixel_per_s"], 4), "unit": "Mpixel/s", "cores": cores, "kind": "port",
                "sample": "%d of %d rows evenly spaced (%d pixels), oracle/restate.js (bit-identical to main.js) under node %s, 1 thread, "
                          "after an untimed JIT warm-up pass" % (rows, h, r["pixels"], r["node"]),
                "mray_per_s": round(r["rays"] / r["ms"] / 1e3, 4), "host_cpus": os.cpu_count()}
    import rt_host
    blob = rt_host.flatten_scene(rt_host.load_scene(scene_name))
    rows = min(h, 256)
    ys = [min(h - 1, int((k + 0.5) * h / rows)) for k in range(rows)]
    t0 = time.perf_counter()
    ou.c_oracle_rows(blob, w, h, ys)
    dt = time.perf_counter() - t0
    return {"value": round(rows * w / dt / 1e6, 4), "unit": "Mpixel/s", "cores": cores, "kind": "port",
            "sample": "%d of %d rows evenly spaced, oracle/rt_oracle.c (gcc -O2, no FMA), 1 thread (node not installed)" % (rows, h),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="h8")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--strict-fp", action="store_true", help="time the no-FMA kernel variant instead")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import rt_host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the render path has no CPU fallback")
    # RT_BENCH_REHEARSE=1: rehearsal of the N>1 control flow on a ONE-GPU box — every rank shares GPU 0 and
    # the gather goes over gloo through host memory.  Never a measurement (the JSON says so).
    rehearse = world > 1 and os.environ.get("RT_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w, h = frame_size_for(world)
    if args.width and args.height:
        w, h = args.width, args.height
    scene = rt_host.load_scene(args.scene)
    ss = scene.get("supersample", 1)
    lib = rt_host.load_library()
    renderer = rt_host.Renderer(scene, dev_index, lib)          # scene resident in HBM from here on
    flags = rt_host.RT_FLAG_STRICT_FP if args.strict_fp else 0

    import shard
    plan = shard.TilePlan(w, h, TILE_ROWS, world)
    per_rank, band_rows = plan.tiles_per_rank, plan.band_rows
    # a dedicated (non-null) HIP stream, made torch's current stream: the kernel launches, the
    # torch.cuda.Events that time them and c10d's stream dependencies all refer to this one stream
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    if world == 1:
        frame = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
        bands = None
    else:
        bands = [torch.empty((band_rows, w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
        frame = torch.empty((h, w, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
        gathered = [torch.empty((world, band_rows, w, 4), dtype=torch.uint8, device=dev) for _ in range(2)] if rank == 0 else None
    my_tiles = rt_host.RtTiles(*plan.rt_tiles(rank))
    host_gathered = torch.empty((world, band_rows, w, 4), dtype=torch.uint8) if (rehearse and rank == 0) else None
    whole = rt_host.RtTiles(h, 0, 1, 1)

    pending = []      # (work, slot) of gathers in flight; at most 2

    def finish(slot_work):
        work, slot = slot_work
        work.wait()                                              # current stream waits for the gather
        if rank == 0:
            if rehearse:
                gathered[slot].copy_(host_gathered)
            shard.deinterleave(plan, gathered[slot], frame, lib=lib, device_index=dev_index, stream=stream)

    def step(k):
        if world == 1:
            renderer.render_tiles(w, h, frame.data_ptr(), whole, stream=stream, flags=flags)
            return
        slot = k & 1
        if len(pending) == 2:                                    # the gather that last used this slot
            finish(pending.pop(0))
        renderer.render_tiles(w, h, bands[slot].data_ptr(), my_tiles, stream=stream, flags=flags)
        if rehearse:
            work = shard.gather_bands(bands[slot].cpu(), host_gathered if rank == 0 else None, dst=0, async_op=True)
        else:
            work = shard.gather_bands(bands[slot], gathered[slot] if rank == 0 else None, dst=0, async_op=True)
        pending.append((work, slot))                             # overlaps with the next step's render

    def drain():
        while pending:
            finish(pending.pop(0))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

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
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- dominant kernel: average launch duration, HIP events on the launch stream ----
    tiles = whole if world == 1 else my_tiles
    target = frame if world == 1 else bands[0]
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 20))]
    for a, b in evs:
        a.record()
        renderer.render_tiles(w, h, target.data_ptr(), tiles, stream=stream, flags=flags)
        b.record()
    torch.cuda.synchronize()
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    launch_pixels = plan.pixels_of(rank) if world > 1 else w * h

    # one more (untimed) frame on every rank, checked on rank 0 against the reference's rows
    step(0)
    drain()
    torch.cuda.synchronize()
    if rank == 0:
        max_lsb = None
        import oracle_util as ou
        for f in ou.manifest()["frames"]:
            if f["scene"] == args.scene and (f["w"], f["h"]) == (w, h) and f["rows"]:
                got = frame[f["rows"]].cpu().numpy().reshape(-1)
                max_lsb = ou.max_lsb(got, ou.golden_frame(f))[0]
        # work counters from the instrumented variant (untimed)
        st = renderer.render_tiles(w, h, target.data_ptr(), tiles, stream=stream, flags=flags | rt_host.RT_FLAG_COUNT, want_stats=True)
        rays_pp, shadow_pp, tests_pp = st.rays / st.pixels, st.shadow_rays / st.pixels, st.sphere_tests / st.pixels
        pixels = w * h
        total_pixels = pixels * args.steps
        value = total_pixels / elapsed / 1e6
        flops_pp = 15.0 * tests_pp + 120.0 * rays_pp + 60.0 * shadow_pp        # SURVEY §8(d) algorithmic FP64 flop model
        algo_bytes = 4.0 * launch_pixels
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            key = "%s_%dx%d" % (args.scene, w, h)
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
        out = {
            "metric": "Mpixel/s", "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s scene (%d spheres, %d lights, depth %d, supersample %d) at %dx%d; %s" % (
                args.scene, len(scene["objects"]), len(scene["lights"]), scene["segs"], ss, w, h,
                "one launch per frame" if world == 1 else "interleaved %d-row tiles over %d ranks + RCCL gather to rank 0 + de-interleave" % (TILE_ROWS, world)),
                "kernel": "strict (no FMA)" if args.strict_fp else "fma", "pixels_per_gpu": pixels // world,
                **({"REHEARSAL": "all ranks on one GPU, gloo through host memory - not a measurement"} if rehearse else {})},
            "mray_per_s": round(value * rays_pp, 2), "mshadow_per_s": round(value * shadow_pp, 2),
            "rays_per_pixel": round(rays_pp, 4), "shadow_rays_per_pixel": round(shadow_pp, 4), "sphere_tests_per_pixel": round(tests_pp, 3),
            "max_lsb_vs_reference_rows": max_lsb,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "kernel": "rt_trace", "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "4 B per output pixel (one RGBA8 store); the path is FP64-VALU bound, see fp64_valu"},
            "fp64_valu": {"flop_per_pixel_model": round(flops_pp, 1), "achieved": round(flops_pp * launch_pixels / (kernel_ms * 1e-3) / 1e12, 3),
                          "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": round(flops_pp * launch_pixels / (kernel_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, 4),
                          "kernel_mpixel_per_s": round(launch_pixels / (kernel_ms * 1e-3) / 1e6, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.scene, w, h)
            except Exception as e:   # the GPU number must still be reported
                out["cpu_baseline"] = {"value": None, "unit": "Mpixel/s", "cores": 1, "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out), flush=True)

    renderer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
