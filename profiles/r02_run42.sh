#!/bin/bash
# Round-2 GPU call 42: 200 000-scene soak of the final kernel (sky workgroups included)
mkdir -p gpurun_out
timeout -k 10 1130 python tests/soak_gpu_parity.py --seeds 200000 --first 11000000 --out gpurun_out/r02_soak_200000_sky.json > gpurun_out/r02_soak_200000_sky.log 2>&1; rc=$?
tail -1 gpurun_out/r02_soak_200000_sky.log; grep -h "flipped_pixels\|worst\|pixels_per_kernel\|seconds\|interrupted" gpurun_out/r02_soak_200000_sky.json
exit $rc
