#!/bin/bash
# Round-2 GPU call 30: LDS image = packed materials (160 B) + texture descriptors, cull rectangles fetched from HBM/L2 per lane,
# deep staging for many spheres  (cur)  vs  the previous commit (base); golden-frame tests first
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "golden or cfg5 or cfg4 or many or lcg or multi_device" > gpurun_out/r02_gpu_tests30.log 2>&1; tail -4 gpurun_out/r02_gpu_tests30.log | cut -c1-300
export STEPS=400
for sc in lcg64 h8 default14 lcg64_ss1; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh base cur
done > gpurun_out/r02_ab_ldsimage.log 2>&1
cat gpurun_out/r02_ab_ldsimage.log
