"""Child process of test_gpu_parity.py::test_rt_render_multi_device_plan_on_emulated_devices.

Run with RT_EMULATE_DEVICES=N in the environment: rt_init then reports N devices that all map to the one physical
GPU, and rt_render takes its multi-GPU plan (interleaved row tiles per device, bands gathered to device 0 — by
device-to-device copies here, by ncclGather on a real node — and de-interleaved).  Prints one line per case:
    <scene> <w> <h> <devices> <sha256 of the RGBA8 frame>
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
import rt_host  # noqa: E402


def main():
    lib = rt_host.load_library()
    for spec in sys.argv[1:]:
        scene, w, h = spec.split(":")
        w, h = int(w), int(h)
        rgba, _ = rt_host.render(w, h, rt_host.load_scene(scene), lib=lib, max_devices=0)
        print(scene, w, h, lib.rt_device_count(), hashlib.sha256(rgba).hexdigest(), flush=True)
    lib.rt_shutdown()


if __name__ == "__main__":
    main()
