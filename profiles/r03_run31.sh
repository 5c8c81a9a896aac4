#!/bin/bash
# Round-3 GPU call 31: headline at HEAD against round 2's library and against HEAD without the sky-part test (same box, interleaved)
mkdir -p gpurun_out
export STEPS=600
BENCH_ARGS="" bash profiles/ab_run.sh r02 product noskypart 2>&1 | grep -v "^/opt\|Traceback\|  File\|    " | tee gpurun_out/r03_ab_head_vs_r02.log
