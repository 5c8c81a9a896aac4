#!/bin/bash
# Round-3 GPU call 7: launch table built on the GPU, one arena per scene, rt_scene_set_camera: new tests first, then the whole GPU suite, A/B vs round 2
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "built_on_the_gpu or moved_camera" > gpurun_out/r03_gpu_tests7a.log 2>&1; tail -15 gpurun_out/r03_gpu_tests7a.log | cut -c1-400
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests7.log 2>&1; tail -12 gpurun_out/r03_gpu_tests7.log | cut -c1-300
export STEPS=600
for sc in h8 default14 lcg64_ss1; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 product
done > gpurun_out/r03_ab_gpu_tables.log 2>&1
cat gpurun_out/r03_ab_gpu_tables.log
