#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of run_profile.sh into the small summaries committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, summ, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(summ, exist_ok=True)
lines = []


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


# ---- kernel trace: per-kernel durations
for f in find("trace/**/*kernel_stats.csv"):
    lines.append("== kernel stats (%s) ==" % os.path.relpath(f, out))
    lines += [l.rstrip() for l in open(f)]
durs = defaultdict(list)
for f in find("trace/**/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
# the same kernel launched with different grids is listed per grid (bench.py's default command renders the timed frames with one
# workgroup per launch-table ENTRY, the frames behind a camera move with one per BLOCK: the count is still on the device there)
by_grid = defaultdict(list)
for k, v in durs.items():
    for d, r in v:
        by_grid[(k, r.get("Grid_Size_X", r.get("Grid_Size")))].append(d)
lines.append("")
lines.append("== rt_trace launches by grid size ==")
for (k, g), d in sorted(by_grid.items(), key=lambda kv: -sum(kv[1])):
    if "rt_trace" in k:
        lines.append("%-80s grid=%s calls=%d avg=%.2f us min=%.1f us max=%.1f us" % (k[:80], g, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
lines.append("")
lines.append("== per-kernel durations from the kernel trace ==")
kern = {}
for k, v in sorted(durs.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    d = [x for x, _ in v]
    r = v[0][1]
    kern[k] = {"calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}
    lines.append("%-90s calls=%d avg=%.1f us min=%.1f us max=%.1f us  VGPR=%s SGPR=%s LDS=%s scratch=%s grid=%s wg=%s" % (
        k[:90], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
        r.get("Scratch_Size"), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size")))

# ---- counters: average per dispatch of the trace kernel
pmc = defaultdict(lambda: defaultdict(list))
for f in find("pmc_*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines.append("")
lines.append("== PMC counters, average per dispatch ==")
counters = {}
for k, cs in pmc.items():
    if "rt_trace" not in k:
        continue
    lines.append(k[:120])
    for c, v in sorted(cs.items()):
        counters[c] = sum(v) / len(v)
        lines.append("   %-28s %18.1f  (n=%d)" % (c, counters[c], len(v)))
# HBM traffic per launch of the dominant kernel, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
# come from separate passes, both are in KiB, and on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (x2).
traffic = None
# the dominant trace kernel of the profiled command: the rt_trace instantiation with the most dispatches (the counting
# variant <*, true, *, *> runs once, untimed)
main = sorted((k for k in pmc if "rt_trace<" in k and not k.split("rt_trace<")[1].split(",")[1].strip() == "true"),
              key=lambda k: -max(len(v) for v in pmc[k].values()))[:1]
if main and "WRITE_SIZE" in pmc[main[0]] and "FETCH_SIZE" in pmc[main[0]]:
    wr = sum(pmc[main[0]]["WRITE_SIZE"]) / len(pmc[main[0]]["WRITE_SIZE"])
    rd = sum(pmc[main[0]]["FETCH_SIZE"]) / len(pmc[main[0]]["FETCH_SIZE"])
    traffic = {"write_bytes": wr * 1024, "fetch_bytes_corrected": 2 * rd * 1024, "hbm_bytes_per_launch": (wr + 2 * rd) * 1024,
               "note": "WRITE_SIZE + 2*FETCH_SIZE (KiB, separate --pmc passes, gfx950 FETCH x2 correction)"}
    lines.append("")
    lines.append("== HBM traffic per launch (%s) ==" % main[0][:100])
    lines.append("   WRITE_SIZE %.1f KiB  FETCH_SIZE %.1f KiB (x2 on gfx950)  ->  %.2f MB per launch" % (wr, rd, (wr + 2 * rd) * 1024 / 1e6))
open(os.path.join(summ, "%s_rocprof_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
dom = {c: sum(v) / len(v) for c, v in pmc[main[0]].items()} if main else {}
json.dump({"kernels": kern, "dominant_kernel": main[0] if main else None, "dominant_kernel_counters_per_dispatch": dom,
           "rt_trace_counters_per_dispatch": counters, "traffic": traffic}, open(os.path.join(summ, "%s_rocprof_summary.json" % tag), "w"), indent=1)
print("\n".join(lines[-60:]))
