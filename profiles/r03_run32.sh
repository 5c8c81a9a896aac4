#!/bin/bash
# Round-3 GPU call 32: where the headline's 1.5 % beyond the sky-part test went: launch-record fields read at the point of use against held in registers
mkdir -p gpurun_out
export STEPS=600
BENCH_ARGS="" bash profiles/ab_run.sh r02 noskypart hotli hottex hotboth 2>&1 | grep -v "^/opt\|Traceback\|  File\|    " | tee gpurun_out/r03_ab_hot_fields.log
