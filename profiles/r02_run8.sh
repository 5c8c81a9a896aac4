#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests8.log 2>&1; tail -40 gpurun_out/r02_gpu_tests8.log | cut -c1-900
