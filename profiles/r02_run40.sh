#!/bin/bash
# Round-2 GPU call 40: is the reference's scene bound by its deepest waves?  kernel ms at 960x540 .. 7680x4320 (a throughput-bound
# kernel scales with the pixel count, a critical-path-bound one does not)
mkdir -p gpurun_out
for wh in "960 540" "1920 1080" "3840 2160" "7680 4320"; do
  set -- $wh
  python3 bench.py --scene default14 --width $1 --height $2 --steps 300 --warmup 10 --no-cpu-baseline --no-pmc 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('default14 $1x$2', d['value'], d['roofline']['kernel_ms'], d['max_lsb_vs_reference_rows'])"
done 2>&1 | tee gpurun_out/r02_default14_by_size.log
