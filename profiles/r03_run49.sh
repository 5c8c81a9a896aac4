#!/bin/bash
# Round-3 GPU call 49: longer randomised runs of the round's new machinery: camera moves (80 000 scenes), sky parts (80 000), narrow-cone tiles (64 000)
mkdir -p gpurun_out
soak() {  # name, seeds per process, first seed, extra args
  local name=$1 n=$2 first=$3; shift 3
  for k in 0 1 2 3; do
    timeout -k 10 ${SOAK_LIMIT:-330} python tests/soak_gpu_parity.py --seeds $n --first $((first + k * n)) --out gpurun_out/r03_soak_${name}_p$k.json "$@" > gpurun_out/r03_soak_${name}_p$k.log 2>&1 &
  done
  wait
  python profiles/merge_soaks.py gpurun_out/r03_soak_${name}.json gpurun_out/r03_soak_${name}_p[0-3].json
}
SOAK_LIMIT=330 soak camera_moves_80k 20000 35000000 --camera-moves
SOAK_LIMIT=200 soak sky_parts_80k 20000 35100000 --sky-parts
SOAK_LIMIT=560 soak windowed_64k 16000 35200000 --windowed
