#!/bin/bash
# Round-2 GPU call 25: rt_render's multi-device plans with supersampled scenes (2x2 in the kernel, 3x3 two-pass)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "multi_device or supersampl or ss3 or ss4" > gpurun_out/r02_gpu_tests25.log 2>&1; tail -15 gpurun_out/r02_gpu_tests25.log | cut -c1-400
