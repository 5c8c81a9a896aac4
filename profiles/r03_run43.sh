#!/bin/bash
# Round-3 GPU call 43: longer soaks at HEAD: 32 000 scenes as 8-row tiles of 3840x2160 / 7680x4320 frames, 24 000 many-sphere scenes, 40 000 with degenerate lights
mkdir -p gpurun_out
soak() {  # name, seeds per process, first seed, extra args
  local name=$1 n=$2 first=$3; shift 3
  for k in 0 1 2 3; do
    timeout -k 10 ${SOAK_LIMIT:-330} python tests/soak_gpu_parity.py --seeds $n --first $((first + k * n)) --out gpurun_out/r03_soak_${name}_p$k.json "$@" > gpurun_out/r03_soak_${name}_p$k.log 2>&1 &
  done
  wait
  python profiles/merge_soaks.py gpurun_out/r03_soak_${name}.json gpurun_out/r03_soak_${name}_p[0-3].json
}
SOAK_LIMIT=400 soak head_windowed_32k 8000 31000000 --windowed
SOAK_LIMIT=500 soak head_many_24k 6000 31100000 --many-spheres
SOAK_LIMIT=120 soak head_degenerate_40k 10000 31200000 --degenerate-lights
