#!/bin/bash
# Round-3 GPU call 4: what the boundary marks cost the headline, piece by piece (600-step trains, two rounds)
mkdir -p gpurun_out
export STEPS=600
for sc in h8 cfg2; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 product direct directnotest
done > gpurun_out/r03_ab_marks_pieces.log 2>&1
cat gpurun_out/r03_ab_marks_pieces.log
