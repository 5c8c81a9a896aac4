#!/bin/bash
# Round-2 GPU call 31: counters of the 64-sphere kernel (3840x2160, 2x2 supersample) after the LDS-image change
mkdir -p gpurun_out
timeout -k 10 500 bash profiles/run_profile.sh r02_lcg64 --scene lcg64 --steps 300 > gpurun_out/r02_profile_lcg64.log 2>&1; tail -3 gpurun_out/r02_profile_lcg64.log
sed -n 1,40p gpurun_out/profiles_r02_lcg64/*summary.txt | cut -c1-180
