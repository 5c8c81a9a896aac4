#!/bin/bash
# Round-3 GPU call 14: where rt_table_rows spends its 27 us (ablations; the rendered frames are wrong in these builds)
mkdir -p gpurun_out; R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in product tnostmt tnocost tnorank tnone; do
  LIB=$R/build/ab/librt_hip_$v.so; [ $v = product ] && LIB=$R/html5-canvas-raytracer_amd/csrc/librt_hip.so
  RT_HIP_LIB=$LIB rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/profiles/moving_camera_loop.py h8 3840 2160 128 > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rt_table" in r["Name"] or "rt_trace" in r["Name"]: print(sys.argv[2], r["Name"][:60], "avg_us %.2f" % (float(r["AverageNs"]) / 1e3))
PY
done
