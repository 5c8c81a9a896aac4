#!/bin/bash
# Round-3 GPU call 20: hand-over plans at their final defaults (tests), Node end to end at the defaults
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "hands_the_frame_over or node or rt_render or api or bridge" > gpurun_out/r03_gpu_tests20.log 2>&1; rc=$?; tail -5 gpurun_out/r03_gpu_tests20.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
for a in "h8 3840 2160" "default14 3840 2160" "h8 1920 1080" "h8 7680 4320"; do timeout -k 10 120 node --expose-gc profiles/node_render_loop.js $a 60; done > gpurun_out/r03_node_render_end_to_end.log 2>&1
cut -c1-700 gpurun_out/r03_node_render_end_to_end.log
