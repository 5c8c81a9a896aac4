#!/bin/bash
# Round-3 GPU call 11: per-kernel times of the moving-camera loop (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_moving -- python3 $GRAFT_REPO_ROOT/profiles/moving_camera_loop.py h8 3840 2160 256 > $GRAFT_REPO_ROOT/gpurun_out/r03_moving_camera_rocprof.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r03_moving_camera_rocprof.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_moving -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s calls %6s avg_us %9.2f total_ms %9.3f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03_moving_camera_kernel_stats.csv
