"""CPU tests of the oracle itself: the restatements are pinned to the reference.

  * restate.js (JS)  == golden frames produced by /root/reference/main.js   BIT-EXACT (sha256)
  * rt_oracle.c (C)  vs the same golden frames                               <= 1 LSB / channel
  * with /root/reference present (build container): re-run the reference itself and compare again,
    so stale fixtures cannot hide a regression.
"""
import hashlib

import numpy as np
import pytest

import oracle_util as ou
import rt_host

M = ou.manifest()
FRAMES = {f["name"]: f for f in M["frames"]}
needs_node = pytest.mark.skipif(ou.node_path() is None, reason="node not installed")
needs_ref = pytest.mark.skipif(not ou.have_reference(), reason="/root/reference not present (GPU box)")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def test_golden_files_match_manifest():
    for f in M["frames"]:
        assert sha(ou.golden_frame(f)) == f["sha256"], f["name"]
    # the reference's own frame through main() and the same scene through our schema agree
    assert FRAMES["default14_main_160x90"]["sha256"] == FRAMES["default14_160x90"]["sha256"]
    # SURVEY §8(c) known answer
    assert FRAMES["default14_main_64x48"]["sha256"] == "8843c2630a9caed89e4cc2467b2fe4f880ae8768b38373c9c6aca2a37b260ecb"


@needs_node
@pytest.mark.parametrize("name", [n for n, f in FRAMES.items() if f["rows"] is None and f["via"] != "main()"])
def test_js_restatement_bit_exact_vs_golden(name):
    f = FRAMES[name]
    r = ou.node_cli("restate", ou.scene_json(f["scene"]), f["w"], f["h"])
    assert r["sha256"] == f["sha256"]


@needs_node
def test_js_restatement_bit_exact_vs_main_frame():
    for name in ("default14_main_64x48", "default14_main_160x90"):
        f = FRAMES[name]
        assert ou.node_cli("restate", ou.scene_json("default14"), f["w"], f["h"])["sha256"] == f["sha256"]


@needs_node
@pytest.mark.parametrize("name", ["h8_960x540", "h8_d8_960x540", "default14_main_256x256", "default14_main_640x360"])
def test_js_restatement_bit_exact_larger_hashes(name):
    h = {x["name"]: x for x in M["hashes"]}[name]
    assert ou.node_cli("restate", ou.scene_json(h["scene"]), h["w"], h["h"])["sha256"] == h["sha256"]


@needs_node
@pytest.mark.parametrize("name", [n for n, f in FRAMES.items() if f["rows"] is not None])
def test_js_restatement_bit_exact_row_bands(name, tmp_path):
    f = FRAMES[name]
    gold = ou.golden_frame(f).reshape(len(f["rows"]), f["w"] * 4)
    for k, y in enumerate(f["rows"]):
        out = tmp_path / "row.rgba"
        ou.node_cli("restate", ou.scene_json(f["scene"]), f["w"], f["h"], y, y + 1, "--out", out)
        assert np.array_equal(np.fromfile(out, dtype=np.uint8), gold[k]), (name, y)


@pytest.mark.parametrize("name", list(FRAMES))
def test_c_restatement_within_1_lsb_of_golden(name, built):
    f = FRAMES[name]
    blob = rt_host.flatten_scene(rt_host.load_scene(f["scene"]))
    got = ou.c_oracle_rows(blob, f["w"], f["h"], f["rows"]) if f["rows"] else ou.c_oracle_render(blob, f["w"], f["h"])
    worst, frac = ou.max_lsb(got, ou.golden_frame(f))
    assert worst <= 1, (name, worst)
    assert frac < 0.01            # rounding ties only (SURVEY: ~0.04 % of channels)


@needs_node
def test_work_counters_match_survey(built):
    # H8 depth 3 = 1.127 rays, 15.64 sphere tests, 0.973 shadow rays per pixel (at 4K);
    # JS and C restatements count identically.
    r = ou.node_cli("restate", ou.scene_json("h8"), 480, 270)
    cnt = [0, 0, 0]
    ou.c_oracle_render(rt_host.flatten_scene(rt_host.load_scene("h8")), 480, 270, counters=cnt)
    assert [r["rays"], r["shadowRays"], r["sphereTests"]] == cnt
    px = 480 * 270
    assert abs(r["rays"] / px - 1.127) < 0.01 and abs(r["sphereTests"] / px - 15.64) < 0.1


# ---------------------------------------------------------------- against the live reference (build container)
@needs_ref
@pytest.mark.reference
def test_reference_main_still_produces_the_golden_frame():
    f = FRAMES["default14_main_64x48"]
    assert ou.node_cli("main", f["w"], f["h"])["sha256"] == f["sha256"]


@needs_ref
@pytest.mark.reference
@pytest.mark.parametrize("scene,w,h", [("cfg1", 96, 96), ("cfg2", 160, 90), ("h8", 200, 112), ("h8_d8", 120, 68),
                                       ("default14", 128, 72), ("lcg64", 48, 48), ("lcg64_ss1", 64, 64)])
def test_restatement_bit_exact_vs_live_reference(scene, w, h):
    # sizes differ from the committed fixtures on purpose
    a = ou.node_cli("reference", ou.scene_json(scene), w, h)
    b = ou.node_cli("restate", ou.scene_json(scene), w, h)
    assert a["sha256"] == b["sha256"]


@needs_ref
@needs_node
@pytest.mark.reference
def test_restatements_vs_live_reference_on_random_scenes(built):
    """A slice of tests/soak_oracle_vs_reference.py (600 scenes by hand: profiles/r01_soak_oracle_vs_reference.json):
    random scenes far from the fixtures - sphere counts, samplers, transparent occluders, lights inside spheres, arbitrary
    cameras, fov, depth, supersampling - through the reference's own intersectWorld; the JS restatement must be
    bit-identical, the C restatement within 1 LSB with no flipped pixel."""
    import soak_oracle_vs_reference as sr
    res = sr.run(10, 1000, verbose=False)
    assert res["scenes"] == 10 and res["restate_mismatch"] == []
    assert res["c_flipped_pixels"] == 0


# ---------------------------------------------------------------- hashed stars sampler (kind 3)
@needs_node
def test_stars_sampler_js_and_c_restatements_agree(built, tmp_path):
    """main.js:135-139 draws its stars with Math.random(), so there is no reference frame to compare with: parity of
    this sampler is UNPINNED against the reference by construction.  What is pinned is that the counter-based hash
    that replaces Math.random() gives the same stars in every implementation here, and that they look right
    (about 0.1 % of the sky's samples, grey)."""
    w, h = 320, 180
    out = tmp_path / "stars.rgba"
    ou.node_cli("restate", ou.scene_json("default14_stars"), w, h, "--out", out)
    js = np.fromfile(out, dtype=np.uint8)
    blob = rt_host.flatten_scene(rt_host.load_scene("default14_stars"))
    c = np.frombuffer(ou.c_oracle_render(blob, w, h), dtype=np.uint8)
    assert ou.max_lsb(js, c)[0] <= 1
    black = np.frombuffer(ou.c_oracle_render(rt_host.flatten_scene(rt_host.load_scene("default14")), w, h), dtype=np.uint8)
    diff = (c.reshape(-1, 4)[:, :3] != black.reshape(-1, 4)[:, :3]).any(axis=1)
    stars = c.reshape(-1, 4)[diff]
    assert 5 <= len(stars) <= 120                                  # ~0.001 of the sky samples (direct and reflected)
    direct = stars[(stars[:, 0] == stars[:, 1]) & (stars[:, 1] == stars[:, 2])]
    assert len(direct) >= 3                                        # stars seen directly are grey (c, c, c)


def test_hashed_stars_have_the_statistics_of_the_references_random_stars(built):
    """VERDICT r03 (missing 4): the reference's stars are random, so the fixture is statistical - star density and grey-level histogram
    of 24 runs of the reference's own main() with its real Math.random (tests/golden/stars_statistics.json).  The counter-based hash
    that stands in for Math.random() in the restatements (and in the kernel: tests/test_gpu_parity.py) must fall inside it."""
    w, h = 1920, 1080
    rows = (0, 300)                                                     # the sky and the first rows below the horizon
    stars = ou.c_oracle_render(rt_host.flatten_scene(rt_host.load_scene("default14_stars")), w, h, *rows)
    black = ou.c_oracle_render(rt_host.flatten_scene(rt_host.load_scene("default14")), w, h, *rows)
    got = ou.stars_statistics_check(stars, black, w, rows[1], frame_h=h)
    assert got["stars"] > 200 and got["chi2"] is not None, got


def _check_fdlibm_vectors(path):
    import ctypes as C
    lib = ou.c_oracle()
    lib.oracle_fd_atan2.restype = C.c_double
    lib.oracle_fd_atan2.argtypes = [C.c_double, C.c_double]
    lib.oracle_fd_asin.restype = C.c_double
    lib.oracle_fd_asin.argtypes = [C.c_double]
    import struct
    n = bad = 0
    with open(path) as f:
        for line in f:
            t, a, b, r = line.split()
            x, y = struct.unpack(">d", bytes.fromhex(a))[0], struct.unpack(">d", bytes.fromhex(b))[0]
            got = lib.oracle_fd_atan2(x, y) if t == "2" else lib.oracle_fd_asin(x)
            want = struct.unpack(">d", bytes.fromhex(r))[0]
            same = struct.pack(">d", got) == bytes.fromhex(r) or (got != got and want != want)
            bad += not same
            n += 1
    return n, bad


def test_fdlibm_trig_matches_v8_fixture(built):
    """oracle/fdlibm_trig.h (the atan2 / asin of the C restatement, and of the strict kernel) against vectors written by V8 itself
    (node v12: oracle/fdlibm_vectors.js, committed as tests/golden/fdlibm_trig_vectors_v8.txt): bit for bit, NaNs as NaNs."""
    import os
    n, bad = _check_fdlibm_vectors(os.path.join(ou.ROOT, "tests", "golden", "fdlibm_trig_vectors_v8.txt"))
    assert n > 6000 and bad == 0, (n, bad)


def test_fdlibm_trig_matches_node(built, tmp_path):
    """The same against a fresh, larger set from the Node that is installed here (100 000 random, tiny, huge and special inputs)."""
    import os
    import subprocess
    node = ou.node_path()
    if not node:
        pytest.skip("node is not installed")
    out = str(tmp_path / "vec.txt")
    subprocess.run([node, os.path.join(ou.ROOT, "oracle", "fdlibm_vectors.js"), out, "40000"], check=True, stdout=subprocess.PIPE, timeout=300)
    n, bad = _check_fdlibm_vectors(out)
    assert n > 90000 and bad == 0, (n, bad)
