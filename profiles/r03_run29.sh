#!/bin/bash
# Round-3 GPU call 29: counters of rt_table_rows (one work-item per block against eight)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in rows_s1_w256 rows_s4_w512 rows_s8_w1024; do
 for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $pass | cut -d' ' -f1)
  RT_HIP_LIB_OLDER=1 RT_HIP_LIB=$R/build/ab/librt_hip_$v.so timeout -k 10 100 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_${v}_$n -- python3 $R/profiles/moving_camera_loop.py h8 3840 2160 24 > /tmp/pmc_${v}_$n.log 2>&1
  python3 - $v /tmp/pmc_${v}_$n <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_table_rows" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(sys.argv[1], {k: round(v[0] / max(v[1], 1), 1) for k, v in acc.items()})
PY
 done
done 2>&1 | tee $R/gpurun_out/r03_table_rows_counters.log
