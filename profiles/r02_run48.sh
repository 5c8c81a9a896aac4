#!/bin/bash
# Round-2 GPU call 48: soaks at the final HEAD: 60 000 ordinary scenes, 3000 many-sphere, 2000 degenerate-light scenes
mkdir -p gpurun_out
timeout -k 10 330 python tests/soak_gpu_parity.py --seeds 60000 --first 15000000 --out gpurun_out/r02_soak_head.json > gpurun_out/r02_soak_head.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_head.json
timeout -k 10 230 python tests/soak_gpu_parity.py --many-spheres --seeds 3000 --first 15100000 --out gpurun_out/r02_soak_head_many.json > gpurun_out/r02_soak_head_many.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_head_many.json
timeout -k 10 100 python tests/soak_gpu_parity.py --degenerate-lights --seeds 2000 --first 15200000 --out gpurun_out/r02_soak_head_degenerate.json > gpurun_out/r02_soak_head_degenerate.log 2>&1
grep -h "flipped_pixels\|worst\|interrupted" gpurun_out/r02_soak_head_degenerate.json
