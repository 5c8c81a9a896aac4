#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's headline workload on the GPU box.
#   usage (from the repo root, on the box):  bash profiles/run_profile.sh <tag> [extra bench.py arguments]
#   e.g.  bash profiles/run_profile.sh r02            (the default command = the headline)
#         bash profiles/run_profile.sh r02_default14 --scene default14 --steps 600
# Writes raw output under gpurun_out/prof_<tag>/ and the summaries to gpurun_out/profiles_<tag>/
# (copy those into profiles/ and commit).  Counters are collected in their own passes, never
# together with tracing (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r01}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
SUM=$REPO/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline --no-pmc --no-cold $*"   # the default command (2000 steps, 20 warm-up), minus the CPU leg, bench.py's own counter passes and the cold-frame / moving-camera legs (their rt_trace launches - one workgroup per block - would mix into the averages of the timed ones)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace rc=$?" >> "$OUT/trace.log"
for PASS in "WRITE_SIZE" "FETCH_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64"; do
  NAME=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc_$NAME" -- $BENCH > "$OUT/pmc_$NAME.log" 2>&1
  echo "pmc $NAME rc=$?" >> "$OUT/pmc_$NAME.log"
done
cd "$REPO"
python3 profiles/summarize_profile.py "$OUT" "$SUM" "$TAG"
