#!/bin/bash
# Round-2 GPU call 27: bounce table at 13 loop spheres (threshold 24 -> 12) on the reference's own scene; test build (env switch)
mkdir -p gpurun_out
L=$PWD/html5-canvas-raytracer_amd/csrc/librt_hip_test.so
for i in 1 2 3; do
for v in "bt24:RT_X=1" "bt12:RT_BTABLE_MIN=12" "bt8:RT_BTABLE_MIN=8"; do
  n=${v%%:*}; e=${v#*:}
  env $e RT_HIP_LIB=$L python3 bench.py --scene default14 --steps 400 --warmup 10 --no-cpu-baseline --no-pmc 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('$n default14', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['max_lsb_vs_reference_rows'])
except Exception as e: print('$n FAILED', l[:200]); print(open('gpurun_out/ab_err.log').read()[-1500:])
"
done; done > gpurun_out/r02_ab_btable13.log 2>&1
cat gpurun_out/r02_ab_btable13.log
