#!/bin/bash
# Round-2 GPU call 1: the whole -m gpu suite, the probe diagnosis of round 1's offending seeds (FMA kernel without the fix-up
# launches), and the reference scene's baseline with the shadow grid on (13 loop spheres > 12) and off.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests1.log 2>&1; rc=$?
tail -5 gpurun_out/r02_gpu_tests1.log
[ $rc -eq 124 ] && exit 124
T=html5-canvas-raytracer_amd/csrc/librt_hip_test.so
RT_NO_FIXUP=1 RT_HIP_LIB=$T timeout -k 10 300 python tests/debug_flips.py 1153727 1189883 d1101 d1616 d1734 d1911 --max 2 > gpurun_out/r02_debug_flips_nofix.log 2>&1 || exit 1
RT_HIP_LIB=$T timeout -k 10 300 python tests/debug_flips.py 1153727 1189883 d1101 d1616 d1734 d1911 --max 2 > gpurun_out/r02_debug_flips_fix.log 2>&1 || exit 1
for i in 1 2; do
  timeout -k 10 200 python bench.py --scene default14 --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r02_d14_base_$i.json 2>gpurun_out/r02_d14_base_$i.err || exit 1
  RT_HIP_LIB=$T RT_SGRID_MIN=13 timeout -k 10 200 python bench.py --scene default14 --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r02_d14_nogrid_$i.json 2>gpurun_out/r02_d14_nogrid_$i.err || exit 1
  RT_HIP_LIB=$T timeout -k 10 200 python bench.py --scene default14 --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r02_d14_testlib_$i.json 2>gpurun_out/r02_d14_testlib_$i.err || exit 1
done
timeout -k 10 200 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > gpurun_out/r02_h8_1.json 2>gpurun_out/r02_h8_1.err
grep -h -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' gpurun_out/r02_d14_*.json gpurun_out/r02_h8_1.json
