#!/bin/bash
# Round-3 GPU call 48: random camera moves of resident scenes against fresh uploads of the moved scenes: 20 000 general, 2 000 many-sphere scenes, two moves each, both kernels
mkdir -p gpurun_out
soak() {  # name, seeds per process, first seed, extra args
  local name=$1 n=$2 first=$3; shift 3
  for k in 0 1 2 3; do
    timeout -k 10 ${SOAK_LIMIT:-330} python tests/soak_gpu_parity.py --seeds $n --first $((first + k * n)) --out gpurun_out/r03_soak_${name}_p$k.json "$@" > gpurun_out/r03_soak_${name}_p$k.log 2>&1 &
  done
  wait
  python profiles/merge_soaks.py gpurun_out/r03_soak_${name}.json gpurun_out/r03_soak_${name}_p[0-3].json
}
SOAK_LIMIT=500 soak camera_moves_general 5000 34000000 --camera-moves
SOAK_LIMIT=400 soak camera_moves_many 500 34100000 --camera-moves --many-spheres
tail -3 gpurun_out/r03_soak_camera_moves_general_p0.log | cut -c1-300
