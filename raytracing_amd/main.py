"""Counterpart of the reference's ``main.py``: render an example scene on the GPU and
write it as a PNG (reference main.py:9-33).

    python -m raytracing_amd.main [example_4] [--width 500 --height 500] [--out Images/result.png]
"""
import argparse
import os
import time

from .constants import HEIGHT, WIDTH, image_directory
from .imageio import save_png
from .raytrace import raytrace_rgb8
from .Scenes.scenes import build_scene


def example_1(width=WIDTH, height=HEIGHT):
    return raytrace_rgb8(build_scene("example_1", width, height))


def example_2(width=WIDTH, height=HEIGHT):
    return raytrace_rgb8(build_scene("example_2", width, height))


def example_3(width=WIDTH, height=HEIGHT):
    return raytrace_rgb8(build_scene("example_3", width, height))


def example_4(width=WIDTH, height=HEIGHT):
    return raytrace_rgb8(build_scene("example_4", width, height))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="example_4")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--out", default=os.path.join(image_directory, "result.png"))
    args = ap.parse_args(argv)
    start = time.time()
    image = raytrace_rgb8(build_scene(args.scene, args.width, args.height))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    save_png(args.out, image)
    print(f"The program took {time.time() - start} seconds to run.")


if __name__ == "__main__":
    main()
