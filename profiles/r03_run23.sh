#!/bin/bash
# Round-3 GPU call 23: soak at HEAD, part 2: windowed 3840x2160 / 7680x4320 tiles (8 000 scenes), many spheres (6 000), degenerate lights (4 000)
mkdir -p gpurun_out
soak() {  # name, seeds per process, first seed, extra args
  local name=$1 n=$2 first=$3; shift 3
  for k in 0 1 2 3; do
    timeout -k 10 ${SOAK_LIMIT:-330} python tests/soak_gpu_parity.py --seeds $n --first $((first + k * n)) --out gpurun_out/r03_soak_${name}_p$k.json "$@" > gpurun_out/r03_soak_${name}_p$k.log 2>&1 &
  done
  wait
  python profiles/merge_soaks.py gpurun_out/r03_soak_${name}.json gpurun_out/r03_soak_${name}_p[0-3].json
}
soak head_windowed 2000 22000000 --windowed
soak head_many 1500 22100000 --many-spheres
soak head_degenerate 1000 22200000 --degenerate-lights
