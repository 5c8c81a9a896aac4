#!/bin/bash
# The GPU-box side of every gpurun call of a round, as named steps (rounds 2 and 3 kept one numbered script per call: 104 of them).
#   gpurun -- 'bash profiles/gpu_steps.sh <tag> <step> [<step> ...]'        e.g.  bash profiles/gpu_steps.sh r04 tests bench rehearse
# Every step writes gpurun_out/<tag>_<step>*.{log,json}; steps are joined with && semantics: the first failure ends the call.
# Steps that take arguments read them from environment variables named in their comment.
set -u
TAG=${1:?tag}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/$TAG
say() { echo "== $TAG $*"; }

step_scratch_probe() {      # what the HIP runtime does when a kernel's scratch reservation cannot be met (once)
  ( ulimit -c 0
    for args in "8589934592 0" "400000000000 1" "2147483648 1"; do      # (leave 8 GB free: 4 KB per lane = 2 GB fits) (hold nothing: 32 KB per lane = 16 GB, memory is there) (leave 2 GB free: it is not)
      echo "---- scratch_refusal_probe $args"
      timeout -k 10 120 build/probe/scratch_refusal_probe $args; echo "exit status $? (134 = SIGABRT, 139 = SIGSEGV, 124 = timeout)"
    done ) > ${O}_scratch_refusal.log 2>&1
  cat ${O}_scratch_refusal.log | cut -c1-220
}
step_tests() {              # the whole -m gpu suite, one process
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > ${O}_gpu_tests.log 2>&1; rc=$?; tail -5 ${O}_gpu_tests.log | cut -c1-300; return $rc
}
step_tests_k() {            # K="expr": a slice of the suite
  timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "$K" > ${O}_gpu_tests_k.log 2>&1; rc=$?; tail -15 ${O}_gpu_tests_k.log | cut -c1-400; return $rc
}
step_bench() {              # the default line and the driver's form
  timeout -k 10 500 python bench.py > ${O}_bench_n1.json 2>${O}_bench_n1.err; rc=$?; cut -c1-400 ${O}_bench_n1.json; [ $rc = 0 ] || { tail -5 ${O}_bench_n1.err; return $rc; }
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > ${O}_bench_n1_driver_form.json 2>${O}_bench_n1_driver_form.err; rc=$?; cut -c1-300 ${O}_bench_n1_driver_form.json; return $rc
}
step_bench_more() {         # the other configs and the reference's own scene on one GPU
  timeout -k 10 200 python bench.py --config cfg4 --no-cpu-baseline > ${O}_bench_cfg4_n1.json 2>${O}_bench_cfg4_n1.err; cut -c1-300 ${O}_bench_cfg4_n1.json
  timeout -k 10 300 python bench.py --config cfg5 --no-cpu-baseline > ${O}_bench_cfg5_n1.json 2>${O}_bench_cfg5_n1.err; cut -c1-300 ${O}_bench_cfg5_n1.json
  timeout -k 10 200 python bench.py --scene default14 --no-cpu-baseline > ${O}_bench_default14_n1.json 2>${O}_bench_default14_n1.err; cut -c1-300 ${O}_bench_default14_n1.json
}
step_rehearse() {           # `bench.py --gpus 2` started bare (the driver's command), two ranks on the one GPU
  RT_BENCH_REHEARSE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 > ${O}_rehearsal_n2_bare.json 2>${O}_rehearsal_n2_bare.err; rc=$?
  cut -c1-600 ${O}_rehearsal_n2_bare.json; [ $rc = 0 ] || tail -20 ${O}_rehearsal_n2_bare.err; return $rc
}
step_profile() {            # PROFILE_TAG=<name> PROFILE_ARGS="<bench.py arguments>": rocprofv3 trace + counter passes (run_profile.sh)
  timeout -k 10 600 bash profiles/run_profile.sh ${PROFILE_TAG:-$TAG} ${PROFILE_ARGS:-} > ${O}_profile_${PROFILE_TAG:-$TAG}.log 2>&1; rc=$?
  tail -4 ${O}_profile_${PROFILE_TAG:-$TAG}.log; rm -rf gpurun_out/prof_${PROFILE_TAG:-$TAG}; return $rc
}
step_ab() {                 # AB="name1 name2 ..." [BENCH_ARGS=...] [STEPS=...]: variants built by profiles/ab_build.sh, interleaved
  bash profiles/ab_run.sh $AB > ${O}_ab_${AB_NAME:-variants}.log 2>&1; cat ${O}_ab_${AB_NAME:-variants}.log | cut -c1-200
}
step_configs() {            # every BASELINE config on one GPU, kernel only
  timeout -k 10 400 python profiles/bench_configs.py --big --out ${O}_configs.json > ${O}_configs.log 2>&1; rc=$?; cut -c1-200 ${O}_configs.log; return $rc
}
step_resources() { bash profiles/kernel_resources.sh > ${O}_kernel_resources.txt 2>&1; cat ${O}_kernel_resources.txt; }
step_soak() {               # SOAK_NAME=<name> SOAK_N=<seeds per process> SOAK_FIRST=<first seed> [SOAK_ARGS="--camera-moves ..."] [SOAK_LIMIT=seconds] [SOAK_PROCS=4]: processes on the one GPU
  local np=${SOAK_PROCS:-4}                      # (at most 6 processes may use the GPU together)
  for k in $(seq 0 $((np - 1))); do
    timeout -k 10 ${SOAK_LIMIT:-500} python tests/soak_gpu_parity.py --seeds $SOAK_N --first $((SOAK_FIRST + k * SOAK_N)) --out ${O}_soak_${SOAK_NAME}_p$k.json ${SOAK_ARGS:-} > ${O}_soak_${SOAK_NAME}_p$k.log 2>&1 &
  done
  wait
  python profiles/merge_soaks.py ${O}_soak_${SOAK_NAME}.json ${O}_soak_${SOAK_NAME}_p[0-9].json; rc=$?
  tail -2 ${O}_soak_${SOAK_NAME}_p0.log | cut -c1-300; rm -f ${O}_soak_${SOAK_NAME}_p[0-9].json
  python3 -c "import json,sys; d=json.load(open('${O}_soak_${SOAK_NAME}.json')); print({k: d[k] for k in d if not isinstance(d[k], (list, dict))})" | cut -c1-900
  return $rc
}
step_cmd() {                # CMD="...": anything else, output to gpurun_out/<tag>_<CMD_NAME>.log
  timeout -k 10 ${CMD_TIMEOUT:-600} bash -c "$CMD" > ${O}_${CMD_NAME:-cmd}.log 2>&1; rc=$?; tail -${CMD_TAIL:-30} ${O}_${CMD_NAME:-cmd}.log | cut -c1-400; return $rc
}

for s in "$@"; do
  say "$s"
  "step_$s" || { echo "== $TAG $s FAILED (rc $?): stopping"; exit 1; }
done
echo "== $TAG done"
