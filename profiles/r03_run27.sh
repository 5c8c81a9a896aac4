#!/bin/bash
# Round-3 GPU call 27: is rt_table_rows' time the cold instruction cache?  The same launch twice in a row, per-call durations from the kernel trace
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
RT_HIP_LIB_OLDER=1 RT_HIP_LIB=$R/build/ab/librt_hip_rowstwice.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_twice -- python3 $R/profiles/moving_camera_loop.py h8 3840 2160 64 > /tmp/prof_twice.log 2>&1
f=$(find /tmp/prof_twice -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/r03_rows_twice.log
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"][:40], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"])) for r in rows]
first, second = [], []
for i in range(1, len(seq)):
    if "rt_table_rows" in seq[i][0] and "rt_table_rows" in seq[i - 1][0]:
        first.append(seq[i - 1][1]); second.append(seq[i][1])
print("pairs", len(first), "first launch avg ns", sum(first) / max(len(first), 1), "second (warm) avg ns", sum(second) / max(len(second), 1))
# one step's sequence with gaps
k = [i for i, s in enumerate(seq) if "rt_small_copy" in s[0]][20]
for i in range(k, k + 8):
    print(seq[i][0], seq[i][1], "gap before next", seq[i + 1][2] - (seq[i][2] + seq[i][1]))
PY
