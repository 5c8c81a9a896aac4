#!/bin/bash
# Round-2 GPU call 17: the constant sky as the miss colour (bounced rays that end on the sky take the miss branch): A/B, suite, soak, bench.
mkdir -p gpurun_out
for i in 1 2 3; do
for v in "sky:build/ab/librt_hip_sky.so" "sky2:build/ab/librt_hip_sky2.so"; do
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
done; done > gpurun_out/r02_ab_sky2.log 2>&1
cat gpurun_out/r02_ab_sky2.log
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests17.log 2>&1; tail -8 gpurun_out/r02_gpu_tests17.log | cut -c1-400
for sc in h8 default14; do
timeout -k 10 200 python bench.py --scene $sc --no-cpu-baseline > gpurun_out/r02_bench_sky2_$sc.json 2>gpurun_out/r02_bench_sky2.err; python3 -c "
import json; d=json.load(open('gpurun_out/r02_bench_sky2_$sc.json')); print('$sc', d['value'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['fp64_valu']['measured'])"
done
timeout -k 10 300 python tests/soak_gpu_parity.py --seeds 30000 --first 5000000 --out gpurun_out/r02_soak_30000_sky2.json > gpurun_out/r02_soak_30000_sky2.log 2>&1
grep -h "flipped_pixels\|worst\|off_by_one" gpurun_out/r02_soak_30000_sky2.json
