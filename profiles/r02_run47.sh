#!/bin/bash
# Round-2 GPU call 47: shadow masks (empty / not empty) and host-named candidates for the many-sphere kernels too (cur) vs HEAD (base)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests47.log 2>&1; tail -4 gpurun_out/r02_gpu_tests47.log | cut -c1-300
export STEPS=400
for sc in default14 lcg64 h8 lcg64_ss1; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base cur
done > gpurun_out/r02_ab_masks_many.log 2>&1
cat gpurun_out/r02_ab_masks_many.log
timeout -k 10 200 python tests/soak_gpu_parity.py --many-spheres --seeds 2500 --first 14000000 --out gpurun_out/r02_soak_masks_many.json > gpurun_out/r02_soak_masks_many.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_masks_many.json
timeout -k 10 150 python tests/soak_gpu_parity.py --seeds 10000 --first 14100000 --out gpurun_out/r02_soak_masks_many_ord.json > gpurun_out/r02_soak_masks_many_ord.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_masks_many_ord.json
