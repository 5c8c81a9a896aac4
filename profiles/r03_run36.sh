#!/bin/bash
# Round-3 GPU call 36: the block statements as a kernel of their own, blocks dealt round-robin to the workgroups: table equality, kernel stats
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "launch_table or moved_camera or owner_fills or retrace or boundary" > gpurun_out/r03_gpu_tests36.log 2>&1; rc=$?; tail -5 gpurun_out/r03_gpu_tests36.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for sc in h8 lcg64_ss1 default14; do
  timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$sc -- python3 $R/profiles/moving_camera_loop.py $sc 3840 2160 128 > /tmp/prof_$sc.log 2>&1
  f=$(find /tmp/prof_$sc -name "*kernel_stats.csv" | head -1)
  echo "== $sc: $(grep 'ms per step' /tmp/prof_$sc.log)"; cut -d, -f1-4 $f | cut -c1-150 | head -7
  [ $sc = h8 ] && cp $f $R/gpurun_out/r03_moving_camera_kernel_stats.csv
  [ $sc = lcg64_ss1 ] && cp $f $R/gpurun_out/r03_moving_camera_kernel_stats_lcg64_ss1.csv
done 2>&1 | tee $R/gpurun_out/r03_table_stmt_split.log
cd $R; for i in 1 2 3; do timeout -k 10 120 python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null; done | tee -a gpurun_out/r03_table_stmt_split.log
