"""Scene files written for the reference run unchanged.

The reference's scenes are plain Python modules that ``import objects``, ``import
materials``, ``from constants import *`` and ``import Scenes.Scene`` and leave the result in a
variable called ``scene`` (reference Scenes/example_1.py:1-19, README "Scenes" section).
:func:`load_scene_file` executes such a file with those module names bound to this package's
classes, so the scene it builds is made of parameter holders the GPU path flattens, at the
image size asked for.
"""
import contextlib
import runpy
import sys

from .. import constants, materials, objects
from . import Scene as _scene_module

_ALIASES = {"objects": objects, "materials": materials, "constants": constants,
            "Scenes": sys.modules[__package__], "Scenes.Scene": _scene_module}


@contextlib.contextmanager
def _reference_module_names(width, height):
    saved = {name: sys.modules.get(name) for name in _ALIASES}
    size = (constants.WIDTH, constants.HEIGHT, constants.SIZE)
    try:
        sys.modules.update(_ALIASES)
        if width is not None and height is not None:
            constants.WIDTH, constants.HEIGHT, constants.SIZE = int(width), int(height), int(width) * int(height)
        yield
    finally:
        constants.WIDTH, constants.HEIGHT, constants.SIZE = size
        for name, mod in saved.items():
            if mod is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = mod


def load_scene_file(path, width=None, height=None, variable="scene"):
    """Run the reference-style scene file ``path`` and return its ``scene`` object, built for a
    ``width`` x ``height`` image (default: the current ``constants.WIDTH`` / ``HEIGHT``)."""
    with _reference_module_names(width, height):
        namespace = runpy.run_path(str(path))
    if variable not in namespace:
        raise ValueError(f"{path} does not define '{variable}'")
    return namespace[variable]
