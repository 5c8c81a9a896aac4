#!/bin/bash
# Round-3 GPU call 38: does the camera-move loop's slow mode (0.27 ms per step in every other process) depend on where the host thread runs?
for f in /sys/class/drm/card*/device/local_cpulist; do echo "$f: $(cat $f)"; done 2>/dev/null | head -4
lscpu | grep -i "numa\|socket" | head -8
for c in 0-15 64-79 128-143 192-207; do
  for i in 1 2; do echo -n "cores $c: "; taskset -c $c python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null; done
done
for i in 1 2 3 4; do echo -n "free: "; python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null; done
