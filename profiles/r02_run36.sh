#!/bin/bash
# Round-2 GPU call 36: GPU suite after the rt_tables split; where the general kernel's instructions and time go on the
# reference's own scene (ablation builds, timing-only images): counters per wave and kernel ms
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_gpu_tests36.log 2>&1; tail -3 gpurun_out/r02_gpu_tests36.log | cut -c1-300
export TMPDIR=/tmp
R=$PWD
cd /tmp
for v in base nosampler nospec noshadow nolight depth1 noshade; do
  rm -rf /tmp/pmc_$v
  RT_HIP_LIB=$R/build/ab/librt_hip_abl_$v.so RT_BENCH_NO_SETTLE=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d /tmp/pmc_$v -- python3 $R/bench.py --scene default14 --steps 6 --warmup 1 --no-cpu-baseline --no-pmc > /tmp/pmc_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_trace<true, false, false, true>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(v, {k: round(sum(x) / len(x) / 129600.0, 1) for k, x in sorted(acc.items())}, "per wave", flush=True)
PY
done 2>&1 | tee $R/gpurun_out/r02_valu_by_section_default14.log
cd $R
for v in base nosampler nospec noshadow nolight depth1 noshade; do
  RT_HIP_LIB=$R/build/ab/librt_hip_abl_$v.so python3 bench.py --scene default14 --steps 300 --warmup 10 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['roofline']['kernel_ms'])"
done 2>&1 | tee -a $R/gpurun_out/r02_valu_by_section_default14.log
