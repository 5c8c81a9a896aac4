#!/bin/bash
# Round-3 GPU call 46: soaks of the final HEAD (many-sphere scenes rendered twice: from the table without and with shadow masks): 12 000 many-sphere scenes, 120 000 general
mkdir -p gpurun_out
soak() {  # name, seeds per process, first seed, extra args
  local name=$1 n=$2 first=$3; shift 3
  for k in 0 1 2 3; do
    timeout -k 10 ${SOAK_LIMIT:-330} python tests/soak_gpu_parity.py --seeds $n --first $((first + k * n)) --out gpurun_out/r03_soak_${name}_p$k.json "$@" > gpurun_out/r03_soak_${name}_p$k.log 2>&1 &
  done
  wait
  python profiles/merge_soaks.py gpurun_out/r03_soak_${name}.json gpurun_out/r03_soak_${name}_p[0-3].json
}
SOAK_LIMIT=400 soak final_many 3000 32000000 --many-spheres
SOAK_LIMIT=500 soak final_general 30000 32100000
