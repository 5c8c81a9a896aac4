"""TEST INFRASTRUCTURE: access to the oracle (oracle/) and the golden frames (tests/golden/).
Only tests, smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import json
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE = os.path.join(ROOT, "oracle")
C_ORACLE_PATH = os.path.join(ORACLE, "librt_oracle.so")
SCENES = os.path.join(ROOT, "html5-canvas-raytracer_amd", "scenes")
REFERENCE_DIR = "/root/reference"

_c = None


def node_path():
    return shutil.which("node")


def have_reference():
    return os.path.exists(os.path.join(REFERENCE_DIR, "main.js")) and node_path() is not None


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def golden_frame(entry):
    return np.fromfile(os.path.join(GOLDEN, entry["file"]), dtype=np.uint8)


def c_oracle():
    global _c
    if _c is None:
        if not os.path.exists(C_ORACLE_PATH):
            subprocess.check_call(["make"], cwd=ORACLE)
        _c = C.CDLL(C_ORACLE_PATH)
        _c.oracle_render_rows.restype = C.c_int
        _c.oracle_render_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.POINTER(C.c_uint64)]
    return _c


def c_oracle_render(blob, w, h, row0=0, row1=None, counters=None):
    """Plain-C restatement (oracle/rt_oracle.c): rows [row0,row1) of the w x h frame -> bytes."""
    lib = c_oracle()
    row1 = h if row1 is None else row1
    out = C.create_string_buffer((row1 - row0) * w * 4)
    buf = C.create_string_buffer(blob, len(blob))
    cnt = (C.c_uint64 * 3)()
    rc = lib.oracle_render_rows(buf, len(blob), w, h, row0, row1, out, cnt)
    if rc != 0:
        raise RuntimeError("oracle_render_rows failed")
    if counters is not None:
        counters[:] = list(cnt)
    return out.raw


def c_oracle_rows(blob, w, h, rows):
    return b"".join(c_oracle_render(blob, w, h, y, y + 1) for y in rows)


def node_cli(*args, timeout=600):
    """Run oracle/cli.js; returns the parsed JSON line."""
    r = subprocess.run([node_path(), os.path.join(ORACLE, "cli.js"), *[str(a) for a in args]], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("oracle/cli.js %s failed: %s %s" % (args, r.stdout, r.stderr))
    return json.loads(r.stdout.strip().splitlines()[-1])


def scene_json(name):
    return os.path.join(SCENES, name + ".json")


def max_lsb(a, b):
    a = np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a
    b = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else b
    assert a.size == b.size, (a.size, b.size)
    d = np.abs(a.reshape(-1).astype(np.int16) - b.reshape(-1).astype(np.int16))
    return int(d.max()), float((d > 0).mean())


def stars_statistics_check(frame_stars, frame_black, w, h, frame_h=None):
    """Hold a frame of the default14_stars scene (hashed stars) to the statistics of the REFERENCE's own random stars
    (tests/golden/stars_statistics.json, made by oracle/make_stars_fixture.js from main() with its real Math.random, main.js:135-139).
    frame_black: the same scene with the constant-black sky (Math.random pinned).  The frames may be the first h rows of a frame of
    frame_h rows.  Returns a dict of what was measured."""
    with open(os.path.join(GOLDEN, "stars_statistics.json")) as f:
        fx = json.load(f)
    a = np.frombuffer(frame_stars, dtype=np.uint8).reshape(h, w, 4)
    b = np.frombuffer(frame_black, dtype=np.uint8).reshape(h, w, 4)
    # the sky seen directly: the whole rows above the scene that are black with the pinned sampler (the fixture's rule)
    black_rows = (b[..., :3] == 0).all(axis=(1, 2))
    sky_rows = int(np.argmin(black_rows)) if not black_rows.all() else h
    assert sky_rows < h and abs(sky_rows / (frame_h or h) - fx["sky_rows"] / fx["h"]) < 0.01, (sky_rows, h)            # the same horizon as the reference's frame
    sky = a[:sky_rows].reshape(-1, 4)
    grey = (sky[:, 0] == sky[:, 1]) & (sky[:, 1] == sky[:, 2])
    assert grey.all()                                                                    # nothing but black and grey (c, c, c) up there
    lit = sky[:, 0] > 0
    n_sky, n_lit = len(sky), int(lit.sum())
    # density: the reference's runs (24 draws of Binomial(sky pixels, ~0.001)) give the rate and its spread; a frame of n_sky pixels
    # has to lie within 4 sigma of the reference's pooled rate
    p_ref = sum(fx["stars_per_run"]) / (fx["runs"] * fx["sky_pixels"])
    assert 0.0009 < p_ref < 0.0011, p_ref                                                # (the fixture itself: main.js:137's 0.001)
    mean, sigma = n_sky * p_ref, (n_sky * p_ref) ** 0.5
    assert abs(n_lit - mean) <= 4.0 * sigma + 1.0, (n_lit, mean, sigma)
    # grey levels: uniform on the byte range, as the reference's (16 bins; chi-square against the reference's pooled histogram)
    hist = np.bincount(sky[lit, 0] >> 4, minlength=16).astype(np.float64)
    ref = np.array(fx["grey_histogram"], dtype=np.float64)
    expect = ref / ref.sum() * hist.sum()
    chi2 = float((((hist - expect) ** 2) / np.maximum(expect, 1e-9)).sum()) if hist.sum() >= 80 else None   # (too few stars in a small frame: density only)
    if chi2 is not None:
        assert chi2 < 45.0, (chi2, hist.tolist())                                        # 15 degrees of freedom: P(chi2 > 45) < 1e-4
    # below the horizon nothing changes but what mirrors the sky: a few per mille of the picture, as in the reference's runs
    elsewhere = int((a[sky_rows:, :, :3] != b[sky_rows:, :, :3]).any(axis=2).sum())
    ref_else = max(fx["changed_elsewhere_per_run"]) / (fx["w"] * fx["h"] - fx["sky_pixels"])
    assert elsewhere <= 3.0 * ref_else * (w * h - n_sky) + 20, (elsewhere, ref_else)
    return {"sky_pixels": n_sky, "stars": n_lit, "expected": round(mean, 1), "sigma": round(sigma, 1), "chi2": chi2, "changed_elsewhere": elsewhere}
