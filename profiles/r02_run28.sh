#!/bin/bash
# Round-2 GPU call 28: workgroup size: 256 threads (32x8 tile, 4 waves, one barrier) vs 128 vs 64 (one wave per workgroup: no
# cross-wave barrier, a wave's slot is reusable the moment it ends); scatter (peer-store) path not valid in these builds
mkdir -p gpurun_out
export STEPS=600
for sc in h8 default14 lcg64; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base w128 w64
done > gpurun_out/r02_ab_wgsize.log 2>&1
cat gpurun_out/r02_ab_wgsize.log
