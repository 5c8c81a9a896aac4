#!/bin/bash
# Round-2 GPU call 9: the evidence at HEAD - rocprofv3 summaries of the default bench command (headline) and of the reference's
# own scene, the per-config table, the bench line.
mkdir -p gpurun_out
timeout -k 10 500 bash profiles/run_profile.sh r02 > gpurun_out/r02_profile_h8.log 2>&1; tail -5 gpurun_out/r02_profile_h8.log
timeout -k 10 400 bash profiles/run_profile.sh r02_default14 --scene default14 --steps 600 > gpurun_out/r02_profile_d14.log 2>&1; tail -5 gpurun_out/r02_profile_d14.log
timeout -k 10 300 python profiles/bench_configs.py --big --out gpurun_out/r02_configs.json > gpurun_out/r02_configs.log 2>&1; cat gpurun_out/r02_configs.log | cut -c1-260
timeout -k 10 300 python bench.py > gpurun_out/r02_bench_n1.json 2>gpurun_out/r02_bench_n1.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r02_bench_n1.json
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_n1_driver_form.json 2>gpurun_out/r02_bench_n1_driver_form.err; cut -c1-300 gpurun_out/r02_bench_n1_driver_form.json
