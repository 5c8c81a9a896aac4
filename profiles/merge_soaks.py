#!/usr/bin/env python3
"""Merge the JSON reports of soak_gpu_parity.py runs that covered disjoint seed ranges (parallel processes on one GPU box):
   python profiles/merge_soaks.py out.json part1.json part2.json ..."""
import json
import sys

out, parts = sys.argv[1], [json.load(open(p)) for p in sys.argv[2:]]
m = {"parts": [p["seeds"] for p in parts], "pixels_per_kernel": sum(p["pixels_per_kernel"] for p in parts), "seconds_max": max(p["seconds"] for p in parts),
     "interrupted": [p.get("interrupted_after_scenes") for p in parts if "interrupted_after_scenes" in p]}
m["scenes"] = sum(p.get("interrupted_after_scenes", p["seeds"][1] - p["seeds"][0] + 1) for p in parts)
for k in ("fma", "strict"):
    ch = sum(p["pixels_per_kernel"] * 4 for p in parts)
    m[k] = {"flipped_pixels": sum(p[k]["flipped_pixels"] for p in parts), "worst_channel_difference": max(p[k]["worst_channel_difference"] for p in parts),
            "off_by_one_channel_fraction": sum(p[k]["off_by_one_channel_fraction"] * p["pixels_per_kernel"] * 4 for p in parts) / max(ch, 1),
            "scenes_with_flips": sum((p[k]["scenes_with_flips"] for p in parts), []), "exact_samples": sum(p[k].get("exact_samples", 0) for p in parts),
            "camera_moves": sum(p[k].get("camera_moves", 0) for p in parts), "camera_moves_refused": sum(p[k].get("camera_moves_refused", 0) for p in parts),
            "moved_frames_that_differ_from_a_fresh_upload": sum(p[k].get("moved_frames_that_differ_from_a_fresh_upload", 0) for p in parts),
            "frames_put_together_from_sky_parts_that_differ": sum(p[k].get("frames_put_together_from_sky_parts_that_differ", 0) for p in parts),
            "frames_that_differ_between_the_table_without_and_with_shadow_masks": sum(p[k].get("frames_that_differ_between_the_table_without_and_with_shadow_masks", 0) for p in parts)}
json.dump(m, open(out, "w"), indent=1)
print(json.dumps({k: (v if k not in ("fma", "strict") else {a: b for a, b in v.items() if a != "scenes_with_flips"} | {"n_scenes_with_flips": len(v["scenes_with_flips"])}) for k, v in m.items()}))
