#!/bin/bash
# Round-3 GPU call 8: launch table built on the GPU (padding slots zeroed, known word cleared on eviction): new tests, whole GPU suite, A/B
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests8.log 2>&1; tail -12 gpurun_out/r03_gpu_tests8.log | cut -c1-300
export STEPS=600
for sc in h8 cfg2; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 direct product
done > gpurun_out/r03_ab_gpu_tables.log 2>&1
grep -v "^/opt\|Traceback\|  File\|    " gpurun_out/r03_ab_gpu_tables.log
