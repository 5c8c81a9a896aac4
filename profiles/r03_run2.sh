#!/bin/bash
# Round-3 GPU call 2: the two tests that failed in call 1, on their own; what the in-kernel re-trace costs the headline, piece by piece
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "cfg1_256x256 or h8_d8_160x90" > gpurun_out/r03_gpu_tests2.log 2>&1; tail -40 gpurun_out/r03_gpu_tests2.log | cut -c1-400
export STEPS=600
for sc in h8; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 neither nomark nocall product
done > gpurun_out/r03_ab_exact_inline_pieces.log 2>&1
cat gpurun_out/r03_ab_exact_inline_pieces.log
