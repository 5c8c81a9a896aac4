#!/bin/bash
# Round-2 GPU call 51: strict kernel with fdlibm's atan2 / asin (the JS engines' own): GPU suite; the three maths-library seeds;
# a 20 000-scene soak (strict must stay bit-close to the restatement, which now uses the same functions)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests51.log 2>&1; tail -12 gpurun_out/r02_gpu_tests51.log | cut -c1-300
timeout -k 10 250 python tests/soak_gpu_parity.py --seeds 20000 --first 15000000 --out gpurun_out/r02_soak_fdlibm.json > gpurun_out/r02_soak_fdlibm.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted\|off_by_one" gpurun_out/r02_soak_fdlibm.json
