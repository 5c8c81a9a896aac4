#!/usr/bin/env python3
"""Regenerates profiles/traffic.json (bench.py's fallback when rocprofv3 is missing or --no-pmc is given) from the committed
rocprofv3 summaries:  python profiles/make_traffic_json.py"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = {"h8_3840x2160": "r03_rocprof_summary.json", "default14_3840x2160": "r03_default14_rocprof_summary.json",
           "lcg64_3840x2160": "r03_lcg64_rocprof_summary.json"}
out = {}
for key, name in SOURCES.items():
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    c = d["dominant_kernel_counters_per_dispatch"]
    t = d["traffic"]
    fma, add, mul = c["SQ_INSTS_VALU_FMA_F64"], c["SQ_INSTS_VALU_ADD_F64"], c["SQ_INSTS_VALU_MUL_F64"]
    out[key] = {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "write_bytes": t["write_bytes"], "fetch_bytes_corrected": t["fetch_bytes_corrected"],
                "kernel": d["dominant_kernel"],
                "source": "profiles/%s (rocprofv3 --pmc, separate passes; FETCH_SIZE x2 on gfx950)" % name,
                "fp64": {"valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
                         "fma_f64": fma, "add_f64": add, "mul_f64": mul, "trans_f64": c.get("SQ_INSTS_VALU_TRANS_F64"),
                         "flop_per_launch_upper_bound": (2 * fma + add + mul) * 64}}
json.dump(out, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
print(json.dumps({k: (round(v["hbm_bytes_per_launch"] / 1e6, 2), v["fp64"]["valu_insts_per_launch"], v["fp64"]["salu_insts_per_launch"]) for k, v in out.items()}))
