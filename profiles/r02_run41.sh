#!/bin/bash
# Round-2 GPU call 41: soaks of the kernel with sky workgroups: 40 000 ordinary scenes, 4000 many-sphere scenes, 3000 with degenerate lights
mkdir -p gpurun_out
timeout -k 10 400 python tests/soak_gpu_parity.py --seeds 40000 --first 10000000 --out gpurun_out/r02_soak_40000_sky.json > gpurun_out/r02_soak_40000_sky.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_40000_sky.json
timeout -k 10 330 python tests/soak_gpu_parity.py --many-spheres --seeds 5000 --first 10100000 --out gpurun_out/r02_soak_many_sky.json > gpurun_out/r02_soak_many_sky.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_many_sky.json
timeout -k 10 200 python tests/soak_gpu_parity.py --degenerate-lights --seeds 3000 --first 10200000 --out gpurun_out/r02_soak_degenerate_sky.json > gpurun_out/r02_soak_degenerate_sky.log 2>&1
grep -h "flipped_pixels\|worst\|interrupted" gpurun_out/r02_soak_degenerate_sky.json
