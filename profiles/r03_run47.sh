#!/bin/bash
# Round-3 GPU call 47: frames put together the multi-GPU way (senders without the sky blocks + the owner's fill) on random scenes: 40 000 general, 4 000 many-sphere
mkdir -p gpurun_out
soak() {  # name, seeds per process, first seed, extra args
  local name=$1 n=$2 first=$3; shift 3
  for k in 0 1 2 3; do
    timeout -k 10 ${SOAK_LIMIT:-330} python tests/soak_gpu_parity.py --seeds $n --first $((first + k * n)) --out gpurun_out/r03_soak_${name}_p$k.json "$@" > gpurun_out/r03_soak_${name}_p$k.log 2>&1 &
  done
  wait
  python profiles/merge_soaks.py gpurun_out/r03_soak_${name}.json gpurun_out/r03_soak_${name}_p[0-3].json
}
SOAK_LIMIT=400 soak sky_parts_general 10000 33000000 --sky-parts
SOAK_LIMIT=400 soak sky_parts_many 1000 33100000 --sky-parts --many-spheres
tail -3 gpurun_out/r03_soak_sky_parts_general_p0.log
