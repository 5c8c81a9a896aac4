#!/bin/bash
# Round-3 GPU call 33: sky parts as launch tables of their own (no test in the kernel): the suite; the headline against round 2's library, without the zero-slot exit, without the mark test
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests33.log 2>&1; rc=$?; tail -5 gpurun_out/r03_gpu_tests33.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
export STEPS=600
BENCH_ARGS="" bash profiles/ab_run.sh r02 product nozero nomark nozeronomark 2>&1 | grep -v "^/opt\|Traceback\|  File\|    " | tee gpurun_out/r03_ab_head_vs_r02.log
