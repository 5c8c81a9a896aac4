#!/bin/bash
# Round-2 GPU call 46: soaks of the FINAL kernel (sky workgroups, shadow masks, host-named candidates): 120 000 ordinary scenes,
# 3000 many-sphere scenes, 3000 with degenerate lights
mkdir -p gpurun_out
timeout -k 10 560 python tests/soak_gpu_parity.py --seeds 120000 --first 13000000 --out gpurun_out/r02_soak_final.json > gpurun_out/r02_soak_final.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_final.json
timeout -k 10 250 python tests/soak_gpu_parity.py --many-spheres --seeds 3000 --first 13300000 --out gpurun_out/r02_soak_final_many.json > gpurun_out/r02_soak_final_many.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_final_many.json
timeout -k 10 200 python tests/soak_gpu_parity.py --degenerate-lights --seeds 3000 --first 13400000 --out gpurun_out/r02_soak_final_degenerate.json > gpurun_out/r02_soak_final_degenerate.log 2>&1
grep -h "flipped_pixels\|worst\|interrupted" gpurun_out/r02_soak_final_degenerate.json
