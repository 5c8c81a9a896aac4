#!/bin/bash
# Round-2 GPU call 22: base (pair loads + early trig loads) vs uvtrig (atan2 and asin chains interleaved, 4 x16 loads)
mkdir -p gpurun_out
export STEPS=600
for sc in h8 default14; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base uvtrig
done > gpurun_out/r02_ab_uvtrig.log 2>&1
cat gpurun_out/r02_ab_uvtrig.log
