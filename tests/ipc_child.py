"""Child process of test_gpu_parity.py::test_ipc_peer_process_renders_into_our_frame: opens the parent's frame buffer through
its IPC handle (rt_ipc_open) and scatters ITS tiles of the frame into it (rt_render_scatter_device), as a peer rank would
over xGMI.  argv: <handle hex> <scene> <w> <h> <tile_rows> <tile_first> <tile_stride> <n_tiles>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "html5-canvas-raytracer_amd"))
import rt_host  # noqa: E402


def main():
    handle = bytes.fromhex(sys.argv[1])
    scene, w, h = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    tiles = tuple(int(a) for a in sys.argv[5:9])
    lib = rt_host.load_library()
    assert lib.rt_init(1) == 0, lib.rt_last_error()
    p = C.c_void_p()
    hb = C.create_string_buffer(handle, len(handle))
    assert lib.rt_ipc_open(0, hb, C.byref(p)) == 0, lib.rt_last_error()
    r = rt_host.Renderer(rt_host.load_scene(scene), 0, lib)
    r.render_scatter(w, h, [p.value], rt_host.RtTiles(*tiles), want_stats=True)      # want_stats: returns when the launch has finished
    r.close()
    assert lib.rt_ipc_close(0, p) == 0, lib.rt_last_error()
    print("IPC_CHILD_DONE", flush=True)


if __name__ == "__main__":
    main()
