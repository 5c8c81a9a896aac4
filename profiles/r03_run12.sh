#!/bin/bash
# Round-3 GPU call 12: faster table build (per-light constants hoisted, one wave per cost class + ticket, one merged copy kernel): tests, profile of the moving-camera loop, bench line
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests12.log 2>&1; tail -6 gpurun_out/r03_gpu_tests12.log | cut -c1-300
bash profiles/r03_run11.sh 2>&1 | grep "calls\|ms per step"
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > gpurun_out/r03_bench_quick.json 2> gpurun_out/r03_bench_quick.err; tail -2 gpurun_out/r03_bench_quick.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03_bench_quick.json'))
print({k: d[k] for k in ('value','ms_per_step','parity_ok')}); print(d.get('cold_frame')); print(d.get('new_camera_every_step'))"
# the stream-ordered allocator for the 3x3 / 4x4 supersampling scratch again: does the round-2 corruption still show in suite order?
RT_HIP_LIB=$PWD/build/ab/librt_hip_pooled.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/r03_gpu_tests12_pooled.log 2>&1; tail -5 gpurun_out/r03_gpu_tests12_pooled.log | cut -c1-300
