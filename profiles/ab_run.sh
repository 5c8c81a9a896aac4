#!/bin/bash
# On the GPU box: bench each named A/B build (build/ab/librt_hip_<name>.so), interleaved, 2 rounds.
#   bash profiles/ab_run.sh w2 w4 ...      prints: name Mpixel/s ms_per_step kernel_ms max_lsb
STEPS=${STEPS:-100}
for round in 1 2; do
for v in "$@"; do
  LIB=$PWD/build/ab/librt_hip_$v.so
  [ "$v" = "product" ] && LIB=$PWD/html5-canvas-raytracer_amd/csrc/librt_hip.so
  RT_HIP_LIB_OLDER=1 RT_HIP_LIB=$LIB python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-pmc --no-cold ${BENCH_ARGS:-} 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['max_lsb_vs_reference_rows'])
except Exception as e: print('$v FAILED', l[:200]); print(open('gpurun_out/ab_err.log').read()[-1500:])
"
done; done
