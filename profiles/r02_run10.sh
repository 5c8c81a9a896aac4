#!/bin/bash
# Round-2 GPU call 10: the 200 000-scene soak of the final kernel (1 G pixels per kernel against the C restatement).
mkdir -p gpurun_out
timeout -k 10 1150 python tests/soak_gpu_parity.py --seeds 200000 --first 3000000 --out gpurun_out/r02_soak_200000.json > gpurun_out/r02_soak_200000.log 2>&1; rc=$?
tail -3 gpurun_out/r02_soak_200000.log; grep -h "flipped_pixels\|worst\|pixels_per_kernel\|seconds" gpurun_out/r02_soak_200000.json
exit $rc
