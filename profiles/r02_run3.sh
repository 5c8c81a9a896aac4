#!/bin/bash
# Round-2 GPU call 3: whole -m gpu suite (new: multi-device plans, cfg4/cfg5 at full size, bench rehearsals), the driver's bench
# command, ablations of the general kernel on the reference's scene, and its rocprofv3 profile.
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests3.log 2>&1; rc=$?
tail -15 gpurun_out/r02_gpu_tests3.log
[ $rc -eq 124 ] && exit 124
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_driver_form.json 2>gpurun_out/r02_bench_driver_form.err; echo "bench rc=$?"
cat gpurun_out/r02_bench_driver_form.json
BENCH_ARGS="--scene default14" STEPS=200 bash profiles/ab_run.sh tbase nosampler nospec noshadow nolight depth1 noshade > gpurun_out/r02_ablate_d14.log 2>&1
cat gpurun_out/r02_ablate_d14.log
timeout -k 10 900 bash profiles/run_profile.sh r02_default14 --scene default14 --steps 600 > gpurun_out/r02_profile_d14.log 2>&1
tail -40 gpurun_out/r02_profile_d14.log
