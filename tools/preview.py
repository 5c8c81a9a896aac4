#!/usr/bin/env python3
"""Render a named scene on the GPU through the C ABI and write it as a PNG (needs an MI355X and Pillow).

    python tools/preview.py default14_stars 960 540 out.png

docs/preview_default14_stars.png was made with exactly that command: the reference's own 14-sphere scene
(main.js:107-157) with the hashed stars sampler, at depth 8.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "html5-canvas-raytracer_amd"))
import rt_host  # noqa: E402


def main(argv):
    from PIL import Image
    name = argv[1] if len(argv) > 1 else "default14_stars"
    w = int(argv[2]) if len(argv) > 2 else 960
    h = int(argv[3]) if len(argv) > 3 else 540
    out = argv[4] if len(argv) > 4 else "%s_%dx%d.png" % (name, w, h)
    rgba, st = rt_host.render(w, h, rt_host.load_scene(name))
    Image.frombytes("RGBA", (w, h), bytes(rgba)).convert("RGB").save(out)
    print("%s: %dx%d, kernel %.3f ms -> %s" % (name, w, h, st.kernel_ms, out))


if __name__ == "__main__":
    main(sys.argv)
