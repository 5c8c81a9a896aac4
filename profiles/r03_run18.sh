#!/bin/bash
# Round-3 GPU call 18: the whole GPU suite with the sky flags (owner fills the sky, senders leave it out) and bench.py's N>1 lines (batch + single frame, efficiency_vs_n1) rehearsed
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests18.log 2>&1; rc=$?; tail -6 gpurun_out/r03_gpu_tests18.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
RT_BENCH_REHEARSE=1 RT_BENCH_P2P=1 MASTER_ADDR=127.0.0.1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 4 > gpurun_out/r03_rehearsal_n2_p2p.json 2> gpurun_out/r03_rehearsal_n2_p2p.err
cat gpurun_out/r03_rehearsal_n2_p2p.json | cut -c1-3000
