#!/bin/bash
# Round-2 GPU call 2: dispatch-order A/B on the reference's scene, degenerate-light soak with and without the strict routing.
mkdir -p gpurun_out
T=html5-canvas-raytracer_amd/csrc/librt_hip_test.so
RT_NO_FIXUP=1 RT_HIP_LIB=$T timeout -k 10 300 python tests/soak_gpu_parity.py --degenerate-lights --seeds 3000 --first 1000 --out gpurun_out/r02_soak_degenerate_nofix.json > gpurun_out/r02_soak_degenerate_nofix.log 2>&1 || exit 1
timeout -k 10 300 python tests/soak_gpu_parity.py --degenerate-lights --seeds 3000 --first 1000 --out gpurun_out/r02_soak_degenerate.json > gpurun_out/r02_soak_degenerate.log 2>&1 || exit 1
timeout -k 10 400 python tests/soak_gpu_parity.py --seeds 20000 --first 2000000 --out gpurun_out/r02_soak_20000.json > gpurun_out/r02_soak_20000.log 2>&1 || exit 1
BENCH_ARGS="--scene default14" STEPS=300 bash profiles/ab_run.sh tbase revy revxy > gpurun_out/r02_ab_order_d14.log 2>&1
BENCH_ARGS="" STEPS=600 bash profiles/ab_run.sh tbase revy > gpurun_out/r02_ab_order_h8.log 2>&1
cat gpurun_out/r02_ab_order_d14.log gpurun_out/r02_ab_order_h8.log
grep -h "flipped_pixels\|worst" gpurun_out/r02_soak_degenerate_nofix.json gpurun_out/r02_soak_degenerate.json gpurun_out/r02_soak_20000.json
