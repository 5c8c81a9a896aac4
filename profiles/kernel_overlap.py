#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: per kernel its calls and average duration, the wall time per step of the traced loop, and how
much of each kernel's time ran BESIDE an rt_trace launch (the camera pipeline's overlap).   python3 profiles/kernel_overlap.py <dir> [steps]"""
import csv, glob, sys, collections
import re
def short(name):
    m = re.search(r"(rt_\w+|__amd_\w+|\w+)(?=<|\(|$)", name.replace("void ", "").replace("(anonymous namespace)::", ""))
    return m.group(1) if m else name[:40]
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", "?")))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tail = rows[-steps * 8:] if len(rows) > steps * 8 else rows            # the loop's last kernels (steady state)
traces = [(a, b) for a, b, n, s in tail if n.startswith("rt_trace")]
span = (tail[-1][1] - tail[0][0]) / 1e3
print("kernels %d, span %.1f us, rt_trace launches %d -> %.1f us per frame" % (len(tail), span, len(traces), span / max(1, len(traces))))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
import bisect
starts = [a for a, b in traces]
for a, b, n, s in tail:
    ov = 0.0
    if not n.startswith("rt_trace"):
        i = bisect.bisect_right(starts, b)
        for ta, tb in traces[max(0, i - 3):i + 1]:
            ov += max(0, min(b, tb) - max(a, ta))
    g = agg[n + " [stream " + s + "]"]
    g[0] += 1; g[1] += (b - a) / 1e3; g[2] += ov / 1e3
for n, (c, t, ov) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-56s calls %5d  avg %7.2f us  beside rt_trace %5.1f %%" % (n, c, t / c, 100.0 * ov / t if t else 0.0))
# gaps on the main stream: idle time between the end of a frame's last kernel and the start of the next rt_trace
gaps = [traces[i + 1][0] - traces[i][1] for i in range(len(traces) - 1)]
gaps.sort()
if gaps: print("between one rt_trace's end and the next one's start: median %.1f us, p90 %.1f us" % (gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3))
