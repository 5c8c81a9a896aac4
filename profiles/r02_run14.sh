#!/bin/bash
# Round-2 GPU call 14: where the headline kernel's VALU instructions go: SQ_INSTS_VALU / SALU of the ablation builds (test
# builds, timing-only images) on the H8 frame.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
for v in base nosampler nospec noshadow nolight depth1 noshade; do
  rm -rf /tmp/pmc_$v
  RT_HIP_LIB=$R/build/ab/librt_hip_$v.so RT_BENCH_NO_SETTLE=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAVES --output-format csv -d /tmp/pmc_$v -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-pmc > /tmp/pmc_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_trace<false, false, false, false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(v, {k: round(sum(x) / len(x) / 129600.0, 1) for k, x in sorted(acc.items())}, "per wave", flush=True)
PY
done 2>&1 | tee $R/gpurun_out/r02_valu_by_section.log
