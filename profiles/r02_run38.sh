#!/bin/bash
# Round-2 GPU call 38: sky workgroups marked in the launch table (geometric cone test on the host) store the background constant
# and skip staging, ray, cull and trace: GPU suite, then cur vs base (= HEAD before the change)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests38.log 2>&1; tail -4 gpurun_out/r02_gpu_tests38.log | cut -c1-300
export STEPS=600
for sc in h8 default14 lcg64 h8_sky_only cfg2; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base cur
done > gpurun_out/r02_ab_sky_tiles.log 2>&1
cat gpurun_out/r02_ab_sky_tiles.log
