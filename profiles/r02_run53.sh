#!/bin/bash
# Round-2 GPU call 53: last check at HEAD: smoke(), GPU suite, the default bench command and the driver's form
mkdir -p gpurun_out
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_gpu_tests53.log 2>&1; tail -3 gpurun_out/r02_gpu_tests53.log | cut -c1-200
timeout -k 10 300 python bench.py > gpurun_out/r02_bench_head.json 2>gpurun_out/r02_bench_head.err; echo "bench rc=$?"; cut -c1-260 gpurun_out/r02_bench_head.json
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_head_driver_form.json 2>gpurun_out/r02_bench_head_driver_form.err; cut -c1-260 gpurun_out/r02_bench_head_driver_form.json
