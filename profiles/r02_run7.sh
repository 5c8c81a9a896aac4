#!/bin/bash
mkdir -p gpurun_out
RT_HIP_LIB=$PWD/html5-canvas-raytracer_amd/csrc/librt_hip_test.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reference_frames or supersample" > gpurun_out/r02_gpu_tests7.log 2>&1; tail -40 gpurun_out/r02_gpu_tests7.log | cut -c1-700
