#!/bin/bash
# Round-3 GPU call 24: final evidence of the round - profiles of the headline, of the reference's scene and of the 64-sphere scene, per-config table,
# bench lines (default form, the driver's form, cfg4, cfg5, the reference's scene)
mkdir -p gpurun_out
timeout -k 10 500 bash profiles/run_profile.sh r03 > gpurun_out/r03_profile_h8.log 2>&1; tail -4 gpurun_out/r03_profile_h8.log
timeout -k 10 400 bash profiles/run_profile.sh r03_default14 --scene default14 --steps 600 > gpurun_out/r03_profile_d14.log 2>&1; tail -4 gpurun_out/r03_profile_d14.log
timeout -k 10 300 bash profiles/run_profile.sh r03_lcg64 --scene lcg64 --steps 300 > gpurun_out/r03_profile_lcg64.log 2>&1; tail -2 gpurun_out/r03_profile_lcg64.log
rm -rf gpurun_out/prof_r03 gpurun_out/prof_r03_default14 gpurun_out/prof_r03_lcg64     # (raw traces: only the summaries under gpurun_out/profiles_* travel back)
timeout -k 10 300 python profiles/bench_configs.py --big --out gpurun_out/r03_configs.json > gpurun_out/r03_configs.log 2>&1; cut -c1-200 gpurun_out/r03_configs.log
