#!/bin/bash
# Round-2 GPU call 23: base (pair loads, interleaved trig) vs hoist (a light's first two scan records fetched with its position)
mkdir -p gpurun_out
export STEPS=600
for sc in h8 default14; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base hoist
done > gpurun_out/r02_ab_hoist.log 2>&1
cat gpurun_out/r02_ab_hoist.log
