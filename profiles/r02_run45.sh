#!/bin/bash
# Round-2 GPU call 45: blocks whose entry names at most two primary candidates skip the cull (cur) vs HEAD (base); in the launch table (per block and light: can ANY sphere shadow a primary hit of the block?  if
# not the scan is skipped): GPU suite, cur vs base (= HEAD), 20 000-scene soak
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests45.log 2>&1; tail -4 gpurun_out/r02_gpu_tests45.log | cut -c1-300
export STEPS=600
for sc in h8 cfg2 h8_d8; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base cur
done > gpurun_out/r02_ab_host_candidates.log 2>&1
cat gpurun_out/r02_ab_host_candidates.log
timeout -k 10 250 python tests/soak_gpu_parity.py --seeds 20000 --first 12200000 --out gpurun_out/r02_soak_20000_host_candidates.json > gpurun_out/r02_soak_20000_host_candidates.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_20000_host_candidates.json
