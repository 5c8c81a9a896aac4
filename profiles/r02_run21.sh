#!/bin/bash
# Round-2 GPU call 21: scalar-load latency experiments on product-style builds (no RT_TESTING: 96 VGPRs, 5 waves/SIMD):
#   base | pair16 (one s_load_dwordx16 per two spheres in the shadow / bounce scans) | trigpf (atan2/asin table in 3+2 loads, issued early) | both
mkdir -p gpurun_out
export STEPS=600
for sc in h8 default14; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base pair16 trigpf both
done > gpurun_out/r02_ab_smem.log 2>&1
cat gpurun_out/r02_ab_smem.log
