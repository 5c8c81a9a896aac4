#!/bin/bash
# Round-2 GPU call 24: pair loads + interleaved trig + hoisted first scan pair: whole GPU suite, bench lines, 30 000-scene soak
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests24.log 2>&1; tail -6 gpurun_out/r02_gpu_tests24.log | cut -c1-300
for sc in h8 default14; do
timeout -k 10 200 python bench.py --scene $sc --no-cpu-baseline > gpurun_out/r02_bench24_$sc.json 2>gpurun_out/r02_bench24.err; python3 -c "
import json; d=json.load(open('gpurun_out/r02_bench24_$sc.json')); print('$sc', d['value'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['fp64_valu']['measured']['valu_insts_per_launch'], d['fp64_valu']['measured']['salu_insts_per_launch'])"
done
timeout -k 10 300 python tests/soak_gpu_parity.py --seeds 30000 --first 7200000 --out gpurun_out/r02_soak_30000_run24.json > gpurun_out/r02_soak_30000_run24.log 2>&1
grep -h "flipped_pixels\|worst\|off_by_one" gpurun_out/r02_soak_30000_run24.json
