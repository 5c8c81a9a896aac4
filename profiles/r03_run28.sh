#!/bin/bash
# Round-3 GPU call 28: rt_table_rows: work-items per block x workgroup size
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in rows_s1_w128 rows_s1_w256 rows_s2_w128 rows_s2_w256 rows_s4_w256 rows_s4_w512 rows_s8_w512 rows_s8_w1024; do
  RT_HIP_LIB_OLDER=1 RT_HIP_LIB=$R/build/ab/librt_hip_$v.so timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/profiles/moving_camera_loop.py h8 3840 2160 96 > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "$v h8: $(grep rt_table_rows $f | cut -d, -f2-4) step: $(tail -1 /tmp/prof_$v.log | cut -c1-60)"
  RT_HIP_LIB_OLDER=1 RT_HIP_LIB=$R/build/ab/librt_hip_$v.so timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2_$v -- python3 $R/profiles/moving_camera_loop.py lcg64_ss1 3840 2160 48 > /tmp/prof2_$v.log 2>&1
  f=$(find /tmp/prof2_$v -name "*kernel_stats.csv" | head -1)
  echo "$v lcg64_ss1: $(grep rt_table_rows $f | cut -d, -f2-4)"
done 2>&1 | tee $R/gpurun_out/r03_ab_table_rows_lanes.log
