#!/bin/bash
# Round-3 GPU call 41: a cheaper boundary-mark test (offset 2^32 instead of 1.5 * 2^32: no sign fix-up; one shift-add per coordinate): the suite, headline and cfg2 against round 2's library and the no-mark build
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests41.log 2>&1; rc=$?; tail -4 gpurun_out/r03_gpu_tests41.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
export STEPS=600
BENCH_ARGS="" bash profiles/ab_run.sh r02 product nomark 2>&1 | grep -v "^/opt\|Traceback\|  File\|    " | tee gpurun_out/r03_ab_cheaper_mark_test.log
BENCH_ARGS="--scene cfg2 --width 1920 --height 1080" bash profiles/ab_run.sh r02 product nomark 2>&1 | grep -v "^/opt\|Traceback\|  File\|    " | tee -a gpurun_out/r03_ab_cheaper_mark_test.log
