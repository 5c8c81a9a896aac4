#!/bin/bash
mkdir -p gpurun_out
export STEPS=600
for sc in h8; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 product knownn8
done > gpurun_out/r03_ab_known_n8.log 2>&1
grep -v "^/opt\|Traceback\|  File\|    " gpurun_out/r03_ab_known_n8.log
