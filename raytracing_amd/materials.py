"""Surface material record (parameter holder only; all arithmetic on it runs on
the device after :func:`raytracing_amd.flatten.flatten_scene`).

Constructor signature, defaults and the derived absorption colour follow the
reference's ``Material`` (reference materials.py:4-29).
"""
import numpy as np

from . import constants as _c


def default_absorption_color(diffuse_color):
    """Absorption colour the reference derives when none is given
    (reference materials.py:21-27): the complement of the diffuse colour scaled
    so that its largest channel is 1, or white when the complement is black."""
    complement = 1 - np.asarray(diffuse_color, dtype=float)
    peak = np.max(complement)
    if peak == 0:
        return _c.WHITE.copy()
    return complement / peak


class Material:
    def __init__(self, ambient_color=None, diffuse_color=_c.YELLOW, specular_color=_c.WHITE,
                 diffuse_coefficient=0.8, specular_coefficient=0.3, shininess=100,
                 reflection_coefficient=0, transparency_coefficient=0, refractive_index=1,
                 smoothness=0, attenuation_coefficient=0.1, absorption_color=None):
        self.diffuse_color = diffuse_color
        self.ambient_color = diffuse_color if ambient_color is None else ambient_color
        self.specular_color = specular_color
        self.diffuse_coefficient = diffuse_coefficient
        self.specular_coefficient = specular_coefficient
        self.shininess = shininess
        self.reflection_coefficient = reflection_coefficient
        self.transparency_coefficient = transparency_coefficient
        self.refractive_index = refractive_index
        self.smoothness = smoothness
        self.attenuation_coefficient = attenuation_coefficient
        if absorption_color is None:
            absorption_color = default_absorption_color(diffuse_color)
        self.absorption_color = absorption_color

    def __repr__(self):
        keys = ("diffuse_color", "reflection_coefficient", "transparency_coefficient",
                "refractive_index", "smoothness")
        return "Material(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in keys) + ")"
