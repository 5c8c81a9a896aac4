#!/bin/bash
# Round-2 GPU call 26: (a) multi-device plans with supersampled scenes; (b) launch order: base-cost (sky) tiles spread among the
# weighted ones (RT_ORDER_MIX percent; product kernels under a test-build host layer)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "multi_device or supersampl or ss3 or ss4" > gpurun_out/r02_gpu_tests25.log 2>&1; tail -5 gpurun_out/r02_gpu_tests25.log | cut -c1-400
L=$PWD/build/ab/librt_hip_hybrid.so
for i in 1 2; do
for sc in h8 default14; do
for mix in 0 25 50 75 100; do
  RT_ORDER_MIX=$mix RT_HIP_LIB=$L python3 bench.py --scene $sc --steps 600 --warmup 10 --no-cpu-baseline --no-pmc 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('mix$mix $sc', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['max_lsb_vs_reference_rows'])
except Exception as e: print('mix$mix $sc FAILED', l[:200]); print(open('gpurun_out/ab_err.log').read()[-1500:])
"
done; done; done > gpurun_out/r02_ab_order_mix.log 2>&1
cat gpurun_out/r02_ab_order_mix.log
