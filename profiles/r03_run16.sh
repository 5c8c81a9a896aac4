#!/bin/bash
# Round-3 GPU call 16: the sphere / light tables staged in LDS and read as broadcast LDS reads (north_star's recipe) against scalar loads into SGPRs (the product)
mkdir -p gpurun_out
export STEPS=600
for sc in h8 cfg2 lcg64_ss1 default14; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh product ldsspheres
done > gpurun_out/r03_ab_lds_vs_sgpr_spheres.log 2>&1
grep -v "^/opt\|Traceback\|  File\|    " gpurun_out/r03_ab_lds_vs_sgpr_spheres.log
