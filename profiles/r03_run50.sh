#!/bin/bash
# Round-3 GPU call 50: cells per axis of the shadow grids (32) and of the bounce table's cube-map faces (8): a sweep on the many-sphere scenes and the reference's own
mkdir -p gpurun_out
export STEPS=200
for sc in lcg64 lcg64_ss1 default14; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh product sg16 sg48 sg64 bg4 bg12 bg16
done 2>&1 | grep -v "^/opt\|Traceback\|  File\|    " | tee gpurun_out/r03_ab_grid_cells.log
