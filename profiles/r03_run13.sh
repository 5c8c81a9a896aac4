#!/bin/bash
# Round-3 GPU call 13: table kernels with scalar loads; table entry before the staging loads (A/B, steady state and moving camera)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "built_on_the_gpu or moved_camera" > gpurun_out/r03_gpu_tests13.log 2>&1; tail -4 gpurun_out/r03_gpu_tests13.log | cut -c1-300
export STEPS=600
for sc in h8 cfg2; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh product entryfirst
done > gpurun_out/r03_ab_entry_first.log 2>&1
grep -v "^/opt\|Traceback\|  File\|    " gpurun_out/r03_ab_entry_first.log
for v in product entryfirst; do
  LIB=$PWD/build/ab/librt_hip_$v.so; [ $v = product ] && LIB=$PWD/html5-canvas-raytracer_amd/csrc/librt_hip.so
  for i in 1 2; do RT_HIP_LIB=$LIB python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null | sed "s/^/$v /"; done
done
bash profiles/r03_run11.sh 2>&1 | grep "calls\|ms per step"
