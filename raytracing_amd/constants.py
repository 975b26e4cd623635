"""Image settings, named colours and tracer limits.

Same names and values as the reference's module-level settings
(reference constants.py:2-19), so scene files written against the reference
keep working when they do ``from raytracing_amd.constants import *``.

Unlike the reference, the renderer does not bind these at import time: the
values are read when :func:`raytracing_amd.raytrace.raytrace` is called (see
:class:`RenderParams`), so a caller may change ``WIDTH``/``HEIGHT`` between
renders or pass an explicit ``RenderParams``.
"""
from dataclasses import dataclass, field

import numpy as np


def _rgb(r, g, b):
    return np.array([r, g, b], dtype=float)


# --- image size (reference constants.py:2-4) -------------------------------
WIDTH = 500
HEIGHT = 500
SIZE = HEIGHT * WIDTH
image_directory = "Images/"

# --- colours (reference constants.py:6-14) ----------------------------------
WHITE = _rgb(1.0, 1.0, 1.0)
BLACK = _rgb(0.0, 0.0, 0.0)
RED = _rgb(1.0, 0.0, 0.0)
GREEN = _rgb(0.0, 1.0, 0.0)
BLUE = _rgb(0.0, 0.0, 1.0)
YELLOW = _rgb(1.0, 1.0, 0.0)
SKY_BLUE = _rgb(0.251, 0.624, 0.769)
GREY = _rgb(0.5, 0.5, 0.5)
BACKGROUND_COLOR = SKY_BLUE

# --- tracer limits (reference constants.py:15-19) ---------------------------
MAX_REFLECTION_DEPTH = 3
MAX_REFRACTION_DEPTH = 5
EPSILON = 0.00001
AIR_REFRACTIVE_INDEX = 1
# A tree node whose whole batch holds fewer rays than this returns the
# background colour for all of them (reference raytrace.py:26-27).
SMALL_NUM = 3


@dataclass
class RenderParams:
    """Explicit render settings handed to the device path.

    The defaults are taken from this module's globals *at construction time*,
    which is what replaces the reference's import-time binding.
    """
    width: int = field(default_factory=lambda: int(WIDTH))
    height: int = field(default_factory=lambda: int(HEIGHT))
    max_reflection_depth: int = field(default_factory=lambda: int(MAX_REFLECTION_DEPTH))
    max_refraction_depth: int = field(default_factory=lambda: int(MAX_REFRACTION_DEPTH))
    epsilon: float = field(default_factory=lambda: float(EPSILON))
    air_refractive_index: float = field(default_factory=lambda: float(AIR_REFRACTIVE_INDEX))
    small_num: int = field(default_factory=lambda: int(SMALL_NUM))
    background_color: np.ndarray = field(
        default_factory=lambda: np.array(BACKGROUND_COLOR, dtype=float))

    @property
    def size(self):
        return self.width * self.height
