#!/bin/bash
# Round-3 GPU call 44: what of rt_table_rows' time is the shadow masks (test build: RT_NO_SHADOW_MASKS), what the sky marks (RT_NO_SKY_TILES)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
T=$R/html5-canvas-raytracer_amd/csrc/librt_hip_test.so
cd /tmp && export TMPDIR=/tmp
for sc in h8 lcg64_ss1; do
 for env in "X=1" "RT_NO_SHADOW_MASKS=1" "RT_NO_SHADOW_MASKS=1 RT_NO_SKY_TILES=1"; do
  rm -rf /tmp/prof_x
  env $env RT_HIP_LIB=$T timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python3 $R/profiles/moving_camera_loop.py $sc 3840 2160 48 > /tmp/prof_x.log 2>&1
  f=$(find /tmp/prof_x -name "*kernel_stats.csv" | head -1)
  echo "$sc [$env]: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'rt_table_rows' in r['Name'] or 'rt_trace' in r['Name']: print(r['Name'][28:60], round(float(r['AverageNs'])/1e3,1), end=' us; ')
")"
 done
done 2>&1 | tee $R/gpurun_out/r03_table_rows_ablation.log
