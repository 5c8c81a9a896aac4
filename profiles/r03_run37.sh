for i in 1 2 3 4; do python3 profiles/moving_camera_loop.py h8 64 64 4096 2>/dev/null; done
for i in 1 2 3 4; do python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null; done
python3 - <<'PY'
import os
print("cpus allowed:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
PY
for i in 1 2 3; do taskset -c 0-15 python3 profiles/moving_camera_loop.py h8 3840 2160 512 2>/dev/null; done
