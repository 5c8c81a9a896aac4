"""Child process of test_gpu_parity.py::test_rt_render_multi_device_plan_on_emulated_devices.

Run with the TEST build of the library (RT_HIP_LIB) and RT_EMULATE_DEVICES=N in the environment: rt_init then reports N
devices that all map to the one physical GPU, and rt_render takes its multi-GPU plan: interleaved row tiles per device,
written straight into device 0's frame (peer-store plan), or - with RT_FORCE_GATHER=1 - RGB24 bands gathered to device 0
(device-to-device copies when emulated, ncclGather otherwise: with ONE real device that is a one-rank RCCL communicator,
the real symbols on a one-GPU box) and de-interleaved.  Every case is rendered TWICE, the second time twice as high, so
the per-device buffers grow between calls.  Prints one line per render:
    RESULT <scene> <w> <h> <devices> <plan: 0 one GPU with a copy-out, 1 peer stores, 2 gather, 3 one GPU storing into the pinned frame> <sha256 of the RGBA8 frame>
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
        for hh in (h, 2 * h):
            rgba, _ = rt_host.render(w, hh, rt_host.load_scene(scene), lib=lib, max_devices=0)
            print("RESULT", scene, w, hh, lib.rt_device_count(), lib.rt_test_last_plan(), hashlib.sha256(rgba).hexdigest(), flush=True)
    lib.rt_shutdown()


if __name__ == "__main__":
    main()
