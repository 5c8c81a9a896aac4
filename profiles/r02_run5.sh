#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tests/debug_ss.py > gpurun_out/r02_debug_ss.log 2>&1; cat gpurun_out/r02_debug_ss.log
T=html5-canvas-raytracer_amd/csrc/librt_hip_test.so
for i in 1 2; do
for v in "old:build/ab/librt_hip_tbase.so:" "unranked:$T:RT_NO_DISPATCH_ORDER=1" "ranked:$T:" "product:html5-canvas-raytracer_amd/csrc/librt_hip.so:"; do
  n=${v%%:*}; r=${v#*:}; l=${r%%:*}; e=${r#*:}
  for sc in default14 h8; do
    env $e RT_HIP_LIB=$PWD/$l python3 bench.py --scene $sc --steps 300 --warmup 10 --no-cpu-baseline --no-pmc 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('$n $sc', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['max_lsb_vs_reference_rows'])
except Exception as e: print('$n $sc FAILED', l[:200]); print(open('gpurun_out/ab_err.log').read()[-1500:])
"
  done
done; done > gpurun_out/r02_ab_lpt.log 2>&1
cat gpurun_out/r02_ab_lpt.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests5.log 2>&1; tail -30 gpurun_out/r02_gpu_tests5.log
