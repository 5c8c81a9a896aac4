#!/bin/bash
# Round-2 GPU call 19: bounce cones: A/B in ONE binary (test build, RT_NO_BOUNCE_CONES), suite, product bench lines.
mkdir -p gpurun_out
T=$PWD/html5-canvas-raytracer_amd/csrc/librt_hip_test.so
for i in 1 2 3; do
for v in "nocones:RT_NO_BOUNCE_CONES=1" "cones:RT_X=1"; do
  n=${v%%:*}; e=${v#*:}
  for sc in default14 h8 h8_d8; do
    env $e RT_HIP_LIB=$T python3 bench.py --scene $sc --steps 400 --warmup 10 --no-cpu-baseline --no-pmc 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('$n $sc', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['max_lsb_vs_reference_rows'])
except Exception as e: print('$n $sc FAILED', l[:200]); print(open('gpurun_out/ab_err.log').read()[-1500:])
"
  done
done; done > gpurun_out/r02_ab_cones.log 2>&1
cat gpurun_out/r02_ab_cones.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests19.log 2>&1; tail -8 gpurun_out/r02_gpu_tests19.log | cut -c1-400
for sc in h8 default14; do
timeout -k 10 200 python bench.py --scene $sc --no-cpu-baseline > gpurun_out/r02_bench_cones_$sc.json 2>gpurun_out/r02_bench_cones.err; python3 -c "
import json; d=json.load(open('gpurun_out/r02_bench_cones_$sc.json')); print('$sc', d['value'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['fp64_valu']['measured'])"
done
timeout -k 10 300 python tests/soak_gpu_parity.py --seeds 30000 --first 7000000 --out gpurun_out/r02_soak_30000_cones.json > gpurun_out/r02_soak_30000_cones.log 2>&1
grep -h "flipped_pixels\|worst\|off_by_one" gpurun_out/r02_soak_30000_cones.json
