#!/bin/bash
# Round-3 GPU call 45: many-sphere scenes: the first frame from a camera without shadow masks in its table - the suite, camera-move steps, static frames
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests45.log 2>&1; rc=$?; tail -4 gpurun_out/r03_gpu_tests45.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
for sc in lcg64_ss1 lcg64 h8 default14; do for i in 1 2; do timeout -k 10 120 taskset -c 0-15 python3 profiles/moving_camera_loop.py $sc 3840 2160 256 2>/dev/null; done; done | tee gpurun_out/r03_moving_camera_masks_deferred.log
export STEPS=300
for sc in lcg64_ss1 lcg64; do echo "== $sc"; BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 product; done 2>&1 | grep -v "^/opt\|Traceback\|  File\|    " | tee -a gpurun_out/r03_moving_camera_masks_deferred.log
