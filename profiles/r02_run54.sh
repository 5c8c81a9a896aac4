#!/bin/bash
# Round-2 GPU call 54: primary candidates named independently of the lights and of the sky's flatness: GPU suite, 12 000-scene soak, bench
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests54.log 2>&1; tail -3 gpurun_out/r02_gpu_tests54.log | cut -c1-200
timeout -k 10 200 python tests/soak_gpu_parity.py --seeds 12000 --first 17000000 --out gpurun_out/r02_soak_54.json > gpurun_out/r02_soak_54.log 2>&1
grep -h "flipped_pixels\|worst\|pixels_per_kernel\|interrupted" gpurun_out/r02_soak_54.json
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_54.json 2>/dev/null; cut -c1-200 gpurun_out/r02_bench_54.json
