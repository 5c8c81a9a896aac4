#!/bin/bash
# Round-3 GPU call 10: fixed-stride launch table (stride from the number of blocks), cold-frame / moving-camera bench: GPU suite, A/B vs round 2, the bench line
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests10.log 2>&1; tail -8 gpurun_out/r03_gpu_tests10.log | cut -c1-300
export STEPS=600
for sc in h8 cfg2 default14; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh r02 product
done > gpurun_out/r03_ab_gpu_tables2.log 2>&1
grep -v "^/opt\|Traceback\|  File\|    " gpurun_out/r03_ab_gpu_tables2.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_n1_driver_form.json 2> gpurun_out/r03_bench_n1_driver_form.err; tail -3 gpurun_out/r03_bench_n1_driver_form.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03_bench_n1_driver_form.json'))
print({k: d[k] for k in ('value','ms_per_step','parity_ok')}); print(d.get('cold_frame')); print(d.get('new_camera_every_step')); print(d['roofline']['frac'], d['roofline']['traffic']); print(d['fp64_valu']['measured'].get('issue') if d['fp64_valu']['measured'] else None); print(d.get('cpu_baseline',{}).get('value'))"
