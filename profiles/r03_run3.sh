#!/bin/bash
# Round-3 GPU call 3: marked samples -> list -> rt_retrace (second, list-driven strict launch, skipped once a frame is known to have none):
# GPU suite, A/B against round 2's library, 20 000-scene soak
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s > gpurun_out/r03_gpu_tests3.log 2>&1; tail -25 gpurun_out/r03_gpu_tests3.log | cut -c1-300
export STEPS=600
for sc in h8 cfg2 h8_d8 default14 lcg64_ss1; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 product
done > gpurun_out/r03_ab_marks_retrace.log 2>&1
cat gpurun_out/r03_ab_marks_retrace.log
timeout -k 10 200 python tests/soak_gpu_parity.py --seeds 20000 --first 17000000 --out gpurun_out/r03_soak_20000_a.json > gpurun_out/r03_soak_20000_a.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted\|exact_samples" gpurun_out/r03_soak_20000_a.json
