#!/bin/bash
# Round-2 GPU call 39: issue priority for waves deep in the ray tree (s_setprio after 2 / 4 / 8 nodes; general kernel only)
mkdir -p gpurun_out
export STEPS=400
for sc in default14 h8; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base prio
done > gpurun_out/r02_ab_prio.log 2>&1
cat gpurun_out/r02_ab_prio.log
