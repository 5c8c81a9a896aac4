#!/bin/bash
# Round-3 GPU call 39: the camera-move loop with a polled staging ring (unpinned host thread, eight processes one after the other)
for i in 1 2 3 4 5 6 7 8; do echo -n "free: "; python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null; done | tee gpurun_out/r03_moving_camera_polled_ring.log
