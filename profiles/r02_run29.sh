#!/bin/bash
# Round-2 GPU call 29: LDS staging of the many-sphere variants with all loads in flight (8 words per work-item) vs the serial loop
mkdir -p gpurun_out
export STEPS=400
for sc in lcg64 default14 lcg64_ss1; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base stage
done > gpurun_out/r02_ab_stage.log 2>&1
cat gpurun_out/r02_ab_stage.log
