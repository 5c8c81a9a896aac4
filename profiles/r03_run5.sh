#!/bin/bash
# Round-3 GPU call 5: boundary marks with the list append in the cold block: split vs merged sampler trigonometry, coefficient prefetch distance
mkdir -p gpurun_out
export STEPS=600
for sc in h8 cfg2; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 product splitnotest merged ahead1 ahead3
done > gpurun_out/r03_ab_marks_split.log 2>&1
cat gpurun_out/r03_ab_marks_split.log
