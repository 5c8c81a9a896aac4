#!/bin/bash
# Round-2 GPU call 37: what a frame of nothing but sky costs (H8 with the camera lifted above the scene, 10 degrees of view)
mkdir -p gpurun_out
for i in 1 2; do
python3 bench.py --scene h8_sky_only --steps 1000 --warmup 10 --no-cpu-baseline --no-pmc 2>gpurun_out/ab_err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('sky-only 3840x2160', d['value'], d['roofline']['kernel_ms'], d['max_lsb_vs_reference_rows'], d.get('rays_per_pixel'))"
done
tail -3 gpurun_out/ab_err.log
