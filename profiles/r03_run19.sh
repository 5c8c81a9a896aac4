#!/bin/bash
# Round-3 GPU call 19: rt_render's hand-over of the frame: direct stores into the pinned frame against the banded copy-out; Node end to end
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "hands_the_frame_over or node or rt_render or api" > gpurun_out/r03_gpu_tests19.log 2>&1; rc=$?; tail -5 gpurun_out/r03_gpu_tests19.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python profiles/render_handover_ab.py > gpurun_out/r03_render_handover_ab.log 2>&1; rc=$?; cat gpurun_out/r03_render_handover_ab.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
for a in "h8 3840 2160" "default14 3840 2160" "h8 1920 1080" "h8 7680 4320"; do timeout -k 10 120 node --expose-gc profiles/node_render_loop.js $a 60; done > gpurun_out/r03_node_render_end_to_end.log 2>&1
cat gpurun_out/r03_node_render_end_to_end.log
