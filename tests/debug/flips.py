#!/usr/bin/env python3
"""Parity debugging aid (run by hand on the GPU box; NOT collected by pytest):

    RT_HIP_LIB=html5-canvas-raytracer_amd/csrc/librt_hip_test.so python tests/debug/flips.py 1153727 1189883 d1101 m5000 ...

For each soak seed (prefix d = drawn with --degenerate-lights, m = --many-spheres, a = --adversarial) it renders the scene with both kernels
and the C restatement, lists the pixels that differ by more than 1 LSB, and for each of them (up to --max) prints the
ray tree of that sample as the kernel walked it (rt_test_probe, test build only) next to the restatement's
(oracle_probe_sample), node by node, marking the first node where they part.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as ou  # noqa: E402
import rt_host  # noqa: E402
import soak_gpu_parity as soak  # noqa: E402

WORDS, NODES = 24, 64
NAMES = ["path", "hcode", "t", "hx", "hy", "hz", "nx", "ny", "nz", "dx", "dy", "dz", "c0", "c1", "c2", "diffuse", "specular", "segs", "li",
         "px", "py", "pz", "kids", "valid"]


def probe_gpu(lib, r, w, h, sx, sy, flags):
    buf = (C.c_double * (WORDS * NODES))()
    lib.rt_test_probe.restype = C.c_int
    lib.rt_test_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    rc = lib.rt_test_probe(r.handle, w, h, sx, sy, flags, buf)
    assert rc == 0, lib.rt_last_error()
    a = np.array(buf[:], dtype=np.float64).reshape(NODES, WORDS)
    return {int(x[0]): x for x in a if x[23] == 1.0}


def probe_oracle(blob, w, h, sx, sy):
    lib = ou.c_oracle()
    buf = (C.c_double * (WORDS * NODES))()
    b = C.create_string_buffer(blob, len(blob))
    lib.oracle_probe_sample.restype = C.c_int
    lib.oracle_probe_sample.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    assert lib.oracle_probe_sample(b, len(blob), w, h, sx, sy, buf) == 0
    a = np.array(buf[:], dtype=np.float64).reshape(NODES, WORDS)
    return {int(x[0]): x for x in a if x[23] == 1.0}


def show(tag, rec):
    return tag + " " + " ".join("%s=%r" % (k, float(v)) for k, v in zip(NAMES[:23], rec[:23]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("seeds", nargs="+")
    ap.add_argument("--max", type=int, default=3)
    args = ap.parse_args()
    lib = rt_host.load_library()
    assert lib.rt_init(1) == 0, lib.rt_last_error()
    has_probe = hasattr(lib, "rt_test_probe")
    for spec in args.seeds:
        deg, many, adv = spec.startswith("d"), spec.startswith("m"), spec.startswith("a")
        seed = int(spec.lstrip("dma"))
        scene, w, h = soak.draw_adversarial(seed)[:3] if adv else soak.draw_scene(seed, deg, many)
        ss = scene.get("supersample", 1)
        blob = rt_host.flatten_scene(scene)
        want = np.frombuffer(ou.c_oracle_render(blob, w, h), dtype=np.uint8).reshape(h, w, 4).astype(np.int16)
        r = rt_host.Renderer(blob, 0, lib)
        d = lib.rt_alloc_device(0, w * h * 4)
        got = {}
        for name, flags in (("fma", 0), ("strict", rt_host.RT_FLAG_STRICT_FP)):
            r.render_tiles(w, h, d, rt_host.RtTiles(h, 0, 1, 1), flags=flags, want_stats=True)
            host = C.create_string_buffer(w * h * 4)
            assert lib.rt_copy_to_host(0, host, d, w * h * 4) == 0
            got[name] = np.frombuffer(host.raw, dtype=np.uint8).reshape(h, w, 4).astype(np.int16)
        for name in ("fma", "strict"):
            diff = np.abs(got[name] - want).max(axis=2)
            ys, xs = np.nonzero(diff > 1)
            print("seed %s %dx%d ss%d spheres %d segs %d lights %d: %s kernel has %d pixels beyond 1 LSB (worst %d)" % (
                spec, w, h, ss, len(scene["objects"]), scene["segs"], len(scene["lights"]), name, len(ys), int(diff.max())), flush=True)
            if name != "fma" or not has_probe:
                continue
            for x, y in list(zip(xs.tolist(), ys.tolist()))[:args.max]:
                print("  pixel (%d,%d): oracle %s fma %s strict %s" % (x, y, want[y, x, :3].tolist(), got["fma"][y, x, :3].tolist(), got["strict"][y, x, :3].tolist()))
                for sub in range(ss * ss):
                    sx, sy = x * ss + sub % ss, y * ss + sub // ss
                    g = probe_gpu(lib, r, w, h, sx, sy, 0)
                    o = probe_oracle(blob, w, h, sx, sy)
                    for path in sorted(set(g) | set(o)):
                        a, b = g.get(path), o.get(path)
                        if a is None or b is None:
                            print("    sample (%d,%d) path %d only in %s" % (sx, sy, path, "oracle" if a is None else "kernel"))
                            print("      " + show("oracle" if a is None else "kernel", b if a is None else a))
                            continue
                        same_hit = bool(np.allclose(a[2:6], b[2:6], rtol=1e-7, atol=1e-7)) and (int(a[1]) & 1) == (int(b[1]) & 1) and (a[1] < 0) == (b[1] < 0)
                        col_same = np.allclose(a[12:15], b[12:15], atol=1e-9, equal_nan=True)
                        lit_same = abs(a[15] - b[15]) < 1e-9 and abs(a[16] - b[16]) < 1e-9
                        if not (same_hit and col_same and lit_same):
                            print("    sample (%d,%d) path %d DIFFERS (hit %s colour %s lighting %s)" % (sx, sy, path, same_hit, col_same, lit_same))
                            print("      " + show("kernel", a))
                            print("      " + show("oracle", b))
                            hi = int(b[1]) >> 1
                            if hi >= 0:
                                ob = scene["objects"][hi]
                                print("      oracle's sphere %d: origin %r r2 %r sampler %r albedo %r" % (hi, ob["origin"], ob["r2"], ob["mtl"]["sampler"], ob["mtl"]["albedo"]))
        lib.rt_free_device(0, d)
        r.close()


if __name__ == "__main__":
    main()
