#!/bin/bash
# Round-2 GPU call 33: primary-ray cull's five compares combined as ballots on the scalar unit (cur) vs one boolean expression (base)
mkdir -p gpurun_out
export STEPS=400
for sc in lcg64 h8 default14; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base cur
done > gpurun_out/r02_ab_cull_ballots.log 2>&1
cat gpurun_out/r02_ab_cull_ballots.log
