#!/usr/bin/env python3
"""Oracle soak against the LIVE reference (build container only; CPU; run by hand):

    python tests/soak_oracle_vs_reference.py [--seeds 200] [--first 1000]

The same random scenes as tests/soak_gpu_parity.py (stars samplers replaced by black: the reference's stars are
Math.random()), each rendered three ways:
  reference   /root/reference/main.js's own intersectWorld, through oracle/ref_harness.js
  restate     oracle/restate.js           must be BIT-IDENTICAL to the reference
  C           oracle/rt_oracle.c          <= 1 LSB per channel, except pixels where glibc and V8 disagree in the last ulp
                                          of atan2/asin/pow right at a discontinuity (counted, expected ~0)
"""
import argparse
import base64
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as ou  # noqa: E402
import rt_host  # noqa: E402
import soak_gpu_parity as soak  # noqa: E402


def scene_without_stars(seed):
    s, w, h = soak.draw_scene(seed)
    for o in s["objects"]:
        if o["mtl"]["sampler"]["kind"] == 3:
            o["mtl"] = dict(o["mtl"], color=[0.0, 0.0, 0.0], sampler={"kind": 0})
    return s, w, h


def to_json(scene):
    j = dict(scene)
    j["textures"] = [{"width": t["width"], "height": t["height"], "base64": base64.b64encode(t["texels"]).decode()} for t in scene["textures"]]
    return json.dumps(j)


def run(seeds, first, verbose=True):
    assert ou.have_reference(), "/root/reference is not here"
    res = {"scenes": 0, "pixels": 0, "restate_mismatch": [], "c_off_by_one_channels": 0, "c_flipped_pixels": 0, "c_scenes_with_flips": []}
    with tempfile.TemporaryDirectory() as td:
        for seed in range(first, first + seeds):
            scene, w, h = scene_without_stars(seed)
            p = os.path.join(td, "s.json")
            open(p, "w").write(to_json(scene))
            ref_out, re_out = os.path.join(td, "ref.rgba"), os.path.join(td, "re.rgba")
            a = ou.node_cli("reference", p, w, h, "--out", ref_out)
            b = ou.node_cli("restate", p, w, h, "--out", re_out)
            if a["sha256"] != b["sha256"]:
                res["restate_mismatch"].append(seed)
            ref = np.fromfile(ref_out, dtype=np.uint8).reshape(h * w, 4).astype(np.int16)
            c = np.frombuffer(ou.c_oracle_render(rt_host.flatten_scene(scene), w, h), dtype=np.uint8).reshape(h * w, 4).astype(np.int16)
            d = np.abs(c - ref)
            flips = int((d.max(axis=1) > 1).sum())
            res["scenes"] += 1
            res["pixels"] += w * h
            res["c_off_by_one_channels"] += int((d == 1).sum())
            res["c_flipped_pixels"] += flips
            if flips:
                res["c_scenes_with_flips"].append({"seed": seed, "pixels": flips})
            if verbose and (seed - first + 1) % 25 == 0:
                print("seed %d: restate mismatches %d, C flipped pixels %d" % (seed, len(res["restate_mismatch"]), res["c_flipped_pixels"]), flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=1000)
    args = ap.parse_args()
    print(json.dumps(run(args.seeds, args.first), indent=1))
