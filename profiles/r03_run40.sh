#!/bin/bash
# Round-3 GPU call 40: bench.py confined to one NUMA node: new_camera_every_step of six processes one after the other, and two unconfined
for i in 1 2 3 4 5 6; do python3 bench.py --no-pmc --no-cpu-baseline --steps 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('confined', d['value'], d['new_camera_every_step']['value'], d['new_camera_every_step']['ms_per_step'], d['config'].get('host_affinity'))"; done | tee gpurun_out/r03_bench_confined.log
for i in 1 2 3 4; do RT_BENCH_NO_AFFINITY=1 python3 bench.py --no-pmc --no-cpu-baseline --steps 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('free', d['value'], d['new_camera_every_step']['value'], d['new_camera_every_step']['ms_per_step'], d['config'].get('host_affinity'))"; done | tee -a gpurun_out/r03_bench_confined.log
