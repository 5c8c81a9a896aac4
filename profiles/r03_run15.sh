#!/bin/bash
# Round-3 GPU call 15: table rows kernel with known-LDS reads: table/camera tests, moving-camera profile, bench line
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "built_on_the_gpu or moved_camera" > gpurun_out/r03_gpu_tests15.log 2>&1; tail -4 gpurun_out/r03_gpu_tests15.log | cut -c1-300
bash profiles/r03_run11.sh 2>&1 | grep "calls\|ms per step"
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null; done
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > gpurun_out/r03_bench_quick.json 2> gpurun_out/r03_bench_quick.err; tail -2 gpurun_out/r03_bench_quick.err
python3 -c "
import json; d=json.load(open('gpurun_out/r03_bench_quick.json'))
print({k: d[k] for k in ('value','ms_per_step','parity_ok')}); print(d.get('cold_frame')); print(d.get('new_camera_every_step'))"
