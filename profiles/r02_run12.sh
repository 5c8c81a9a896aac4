#!/bin/bash
# Round-2 GPU call 12: table-loaded trig constants vs OCML (same tree otherwise), byte conversion fix; suite; soak.
mkdir -p gpurun_out
for i in 1 2 3; do
for v in "ocml:build/ab/librt_hip_ocml.so" "trig2:build/ab/librt_hip_trig2.so"; do
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
done; done > gpurun_out/r02_ab_trig2.log 2>&1
cat gpurun_out/r02_ab_trig2.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests12.log 2>&1; tail -8 gpurun_out/r02_gpu_tests12.log | cut -c1-400
timeout -k 10 300 python tests/soak_gpu_parity.py --seeds 30000 --first 4000000 --out gpurun_out/r02_soak_30000_trig2.json > gpurun_out/r02_soak_30000_trig2.log 2>&1
grep -h "flipped_pixels\|worst\|off_by_one" gpurun_out/r02_soak_30000_trig2.json
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_trig2.json 2>gpurun_out/r02_bench_trig2.err; python3 -c "
import json; d=json.load(open('gpurun_out/r02_bench_trig2.json')); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['fp64_valu']['measured'])"
