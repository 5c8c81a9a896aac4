#!/bin/bash
# Round-2 GPU call 18: the whole -m gpu suite at HEAD, then the 200 000-scene soak at HEAD.
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests18.log 2>&1; rc=$?; tail -6 gpurun_out/r02_gpu_tests18.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 860 python tests/soak_gpu_parity.py --seeds 200000 --first 6000000 --out gpurun_out/r02_soak_200000_head.json > gpurun_out/r02_soak_200000_head.log 2>&1; rc=$?
tail -2 gpurun_out/r02_soak_200000_head.log; grep -h "flipped_pixels\|worst\|pixels_per_kernel\|seconds\|off_by_one" gpurun_out/r02_soak_200000_head.json
exit $rc
