#!/bin/bash
# Round-2 GPU call 35: soaks of the FINAL kernel: 12 000 many-sphere scenes (shadow-grid / bounce-table kernels, the new LDS image
# and staging), 3000 with degenerate lights, 200 000 ordinary scenes; counters of the 64-sphere kernel
mkdir -p gpurun_out
timeout -k 10 400 python tests/soak_gpu_parity.py --many-spheres --seeds 12000 --first 8000000 --out gpurun_out/r02_soak_12000_many_spheres.json > gpurun_out/r02_soak_many.log 2>&1; tail -2 gpurun_out/r02_soak_many.log
grep -h "flipped_pixels\|worst\|pixels_per_kernel" gpurun_out/r02_soak_12000_many_spheres.json
timeout -k 10 200 python tests/soak_gpu_parity.py --degenerate-lights --seeds 3000 --first 8100000 --out gpurun_out/r02_soak_3000_degenerate_final.json > gpurun_out/r02_soak_degen.log 2>&1
grep -h "flipped_pixels\|worst" gpurun_out/r02_soak_3000_degenerate_final.json
timeout -k 10 300 bash profiles/run_profile.sh r02_lcg64 --scene lcg64 --steps 300 > gpurun_out/r02_profile_lcg64.log 2>&1; tail -2 gpurun_out/r02_profile_lcg64.log
timeout -k 10 1100 python tests/soak_gpu_parity.py --seeds 200000 --first 9000000 --out gpurun_out/r02_soak_200000_final.json > gpurun_out/r02_soak_200000_final.log 2>&1; rc=$?
tail -2 gpurun_out/r02_soak_200000_final.log; grep -h "flipped_pixels\|worst\|pixels_per_kernel\|seconds" gpurun_out/r02_soak_200000_final.json
exit $rc
