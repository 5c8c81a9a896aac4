#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reference_frames" > gpurun_out/r02_gpu_tests6.log 2>&1; tail -30 gpurun_out/r02_gpu_tests6.log | cut -c1-600
for i in 1 2; do
for v in "lpt:build/ab/librt_hip_lpt.so" "diet:build/ab/librt_hip_diet.so" "product:html5-canvas-raytracer_amd/csrc/librt_hip.so"; do
  n=${v%%:*}; l=${v#*:}
  for sc in default14 h8; do
    RT_HIP_LIB=$PWD/$l python3 bench.py --scene $sc --steps 400 --warmup 10 --no-cpu-baseline --no-pmc 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('$n $sc', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['max_lsb_vs_reference_rows'])
except Exception as e: print('$n $sc FAILED', l[:200]); print(open('gpurun_out/ab_err.log').read()[-1500:])
"
  done
done; done > gpurun_out/r02_ab_diet.log 2>&1
cat gpurun_out/r02_ab_diet.log
