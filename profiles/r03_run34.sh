#!/bin/bash
# Round-3 GPU call 34: the exact cut of the general kernel (leaving ray first + "the continuing chain cannot change a byte"): parity of the variant, timing against HEAD
mkdir -p gpurun_out
RT_HIP_LIB_OLDER=1 RT_HIP_LIB=$PWD/build/ab/librt_hip_prune.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_frames or random or refract or soak_seeds or default14" > gpurun_out/r03_gpu_tests34.log 2>&1; tail -5 gpurun_out/r03_gpu_tests34.log | cut -c1-600
export STEPS=300
for sc in default14 "default14 --width 1920 --height 1080" "default14 --width 7680 --height 4320"; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh product prune
done 2>&1 | grep -v "^/opt\|Traceback\|  File\|    " | tee gpurun_out/r03_ab_prune.log
