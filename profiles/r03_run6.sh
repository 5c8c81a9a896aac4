#!/bin/bash
# Round-3 GPU call 6: boundary marks + rt_retrace at HEAD: GPU suite, then soaks in 4 parallel processes (general 160 000 scenes, windowed-4K/8K 8 000,
# many-sphere 6 000, degenerate lights 4 000)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests6.log 2>&1; tail -8 gpurun_out/r03_gpu_tests6.log | cut -c1-300
soak() {  # name, seeds per process, first seed, extra args
  local name=$1 n=$2 first=$3; shift 3
  for k in 0 1 2 3; do
    timeout -k 10 ${SOAK_LIMIT:-330} python tests/soak_gpu_parity.py --seeds $n --first $((first + k * n)) --out gpurun_out/r03_soak_${name}_p$k.json "$@" > gpurun_out/r03_soak_${name}_p$k.log 2>&1 &
  done
  wait
  python profiles/merge_soaks.py gpurun_out/r03_soak_${name}.json gpurun_out/r03_soak_${name}_p[0-3].json
}
SOAK_LIMIT=400 soak general 40000 18000000
soak windowed 2000 19000000 --windowed
soak many 1500 19100000 --many-spheres
soak degenerate 1000 19200000 --degenerate-lights
