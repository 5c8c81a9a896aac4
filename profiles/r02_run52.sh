#!/bin/bash
# Round-2 GPU call 52: 150 000-scene soak at HEAD (strict kernel with fdlibm's atan2 / asin): how often the product path meets a sampler
# boundary, and whether the strict kernel differs from the restatement anywhere at all
mkdir -p gpurun_out
timeout -k 10 700 python tests/soak_gpu_parity.py --seeds 150000 --first 16000000 --out gpurun_out/r02_soak_150000_head.json > gpurun_out/r02_soak_150000_head.log 2>&1; rc=$?
tail -1 gpurun_out/r02_soak_150000_head.log; grep -h "flipped_pixels\|worst\|pixels_per_kernel\|seconds\|interrupted\|off_by_one\|seed" gpurun_out/r02_soak_150000_head.json
exit $rc
