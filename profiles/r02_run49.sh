#!/bin/bash
# Round-2 GPU call 49: the two pixels of the 60 000-scene soak in which BOTH kernels differ from the restatement: ray trees side by side
mkdir -p gpurun_out
RT_HIP_LIB=$PWD/html5-canvas-raytracer_amd/csrc/librt_hip_test.so timeout -k 10 300 python tests/debug_flips.py 15004219 15007010 > gpurun_out/r02_probe_soak_head_flips.log 2>&1
cut -c1-400 gpurun_out/r02_probe_soak_head_flips.log | head -80
