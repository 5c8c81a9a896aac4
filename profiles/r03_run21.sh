#!/bin/bash
# Round-3 GPU call 21: general kernel - at a two-child node the child that leaves the sphere is traced first (park stack one record deep on the reference's own scene):
# timing against reflection-first (the same sources), HBM traffic of both (rocprofv3 --pmc, separate passes), the GPU suite
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests21.log 2>&1; rc=$?; tail -5 gpurun_out/r03_gpu_tests21.log | cut -c1-600
export STEPS=300
for sc in default14 "default14 --width 1920 --height 1080"; do
  echo "== $sc"
  BENCH_ARGS="--scene $sc" bash profiles/ab_run.sh reflfirst product
done > gpurun_out/r03_ab_park_order.log 2>&1
grep -v "^/opt\|Traceback\|  File\|    " gpurun_out/r03_ab_park_order.log
cd /tmp && export TMPDIR=/tmp
for v in reflfirst product; do
  LIB=$GRAFT_REPO_ROOT/build/ab/librt_hip_$v.so; [ "$v" = product ] && LIB=$GRAFT_REPO_ROOT/html5-canvas-raytracer_amd/csrc/librt_hip.so
  for c in WRITE_SIZE FETCH_SIZE; do
    RT_HIP_LIB_OLDER=1 RT_HIP_LIB=$LIB timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$v_$c -- python3 $GRAFT_REPO_ROOT/bench.py --scene default14 --steps 12 --warmup 2 --no-cpu-baseline --no-pmc --no-cold > /dev/null 2>&1
    python3 - <<PY
import csv,glob
tot=0;n=0
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$v_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("rt_trace") or "rt_trace" in r["Kernel_Name"]:
            tot+=float(r["Counter_Value"]); n+=1
print("$v $c per launch:", tot/max(n,1), "launches", n)
PY
  done
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r03_park_order_traffic.log
