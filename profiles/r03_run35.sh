#!/bin/bash
# Round-3 GPU call 35: counters of rt_table_rows for the 64-sphere scene
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 100 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$n -- python3 $R/profiles/moving_camera_loop.py lcg64_ss1 3840 2160 16 > /tmp/pmc_$n.log 2>&1
  python3 - /tmp/pmc_$n <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_table_rows" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({k: round(v[0] / max(v[1], 1), 1) for k, v in acc.items()})
PY
done 2>&1 | tee $R/gpurun_out/r03_table_rows_counters_lcg64.log
