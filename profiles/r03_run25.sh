#!/bin/bash
# Round-3 GPU call 25: final bench lines of the round (default form, the driver's form, cfg4, cfg5, the reference's scene), Node end to end, kernel resources
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_n1.json 2>gpurun_out/r03_bench_n1.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r03_bench_n1.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_n1_driver_form.json 2>gpurun_out/r03_bench_n1_driver_form.err; cut -c1-300 gpurun_out/r03_bench_n1_driver_form.json
timeout -k 10 200 python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/r03_bench_cfg4_n1.json 2>gpurun_out/r03_bench_cfg4_n1.err; cut -c1-300 gpurun_out/r03_bench_cfg4_n1.json
timeout -k 10 300 python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/r03_bench_cfg5_n1.json 2>gpurun_out/r03_bench_cfg5_n1.err; cut -c1-300 gpurun_out/r03_bench_cfg5_n1.json
timeout -k 10 200 python bench.py --scene default14 --no-cpu-baseline > gpurun_out/r03_bench_default14_n1.json 2>gpurun_out/r03_bench_default14_n1.err; cut -c1-300 gpurun_out/r03_bench_default14_n1.json
for a in "h8 3840 2160" "default14 3840 2160" "h8 1920 1080" "h8 7680 4320"; do timeout -k 10 120 node --expose-gc profiles/node_render_loop.js $a 60; done > gpurun_out/r03_node_render_end_to_end.log 2>&1
cut -c1-400 gpurun_out/r03_node_render_end_to_end.log
bash profiles/kernel_resources.sh > gpurun_out/r03_kernel_resources.txt 2>&1; cat gpurun_out/r03_kernel_resources.txt
bash profiles/isa_histogram.sh 0 0 0 0 > gpurun_out/r03_isa_histogram_rt_trace_0000.txt 2>&1; tail -3 gpurun_out/r03_isa_histogram_rt_trace_0000.txt
