#!/bin/bash
# Round-3 GPU call 26: rt_table_rows with eight work-items per block: table equality (incl. 65 / 128 / 200 spheres), the moving camera (timing, kernel stats), the suite
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests26.log 2>&1; rc=$?; tail -5 gpurun_out/r03_gpu_tests26.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do timeout -k 10 120 python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null; done | tee gpurun_out/r03_moving_camera_rows8.log
timeout -k 10 120 python3 profiles/moving_camera_loop.py lcg64 3840 2160 256 2>/dev/null | tee -a gpurun_out/r03_moving_camera_rows8.log
timeout -k 10 120 python3 profiles/moving_camera_loop.py default14 3840 2160 256 2>/dev/null | tee -a gpurun_out/r03_moving_camera_rows8.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rows8 -- python3 $R/profiles/moving_camera_loop.py h8 3840 2160 256 > /tmp/prof_rows8.log 2>&1
f=$(find /tmp/prof_rows8 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r03_moving_camera_kernel_stats_rows8.csv; cut -c1-200 "$f" | head -12
