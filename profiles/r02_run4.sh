#!/bin/bash
# Round-2 GPU call 4: the whole -m gpu suite after 8(f)-4 (build stamp, 3x3 / 4x4 supersampling) and the multi-GPU rework.
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests4.log 2>&1; rc=$?
tail -40 gpurun_out/r02_gpu_tests4.log
exit $rc
