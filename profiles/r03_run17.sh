#!/bin/bash
# Round-3 GPU call 17: is the general kernel's park stack (3472 B of scratch per lane) what holds its occupancy down?  park stack of 16 (product), 8, 4 entries on the reference's own scene
# (depth 8: at most 7 parked nodes; with 4 entries the frame is wrong - timing only); plus the whole GPU suite and the many-sphere numbers at HEAD (no spills any more)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests17.log 2>&1; tail -4 gpurun_out/r03_gpu_tests17.log | cut -c1-300
export STEPS=300
for sc in default14; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh product park8 park4
done > gpurun_out/r03_ab_park_depth.log 2>&1
grep -v "^/opt\|Traceback\|  File\|    " gpurun_out/r03_ab_park_depth.log
for sc in lcg64_ss1 lcg64; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 product
done > gpurun_out/r03_ab_many_spheres.log 2>&1
grep -v "^/opt\|Traceback\|  File\|    " gpurun_out/r03_ab_many_spheres.log
