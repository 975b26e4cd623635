"""Example 1: white ground, blue and yellow opaque spheres, one point light
(same content as reference Scenes/example_1.py:5-19)."""
import numpy as np

from .. import materials, objects
from ..constants import BLUE, WHITE, YELLOW
from .Scene import Scene


def build(width=None, height=None):
    """Build the scene for a ``width`` x ``height`` image (defaults: the
    current ``constants.WIDTH``/``HEIGHT``)."""
    ground = objects.Sphere(x=0, y=0, z=-1000000, radius=1000000,
                            material=materials.Material(diffuse_color=WHITE,
                                                        reflection_coefficient=0, smoothness=1))
    blue = objects.Sphere(x=4, y=0, z=1, radius=1,
                          material=materials.Material(diffuse_color=BLUE,
                                                      reflection_coefficient=0.1, shininess=10))
    yellow = objects.Sphere(y=2, z=1.25, radius=0.5,
                            material=materials.Material(diffuse_color=YELLOW,
                                                        reflection_coefficient=0.5))
    lights = [objects.PointSource(x=4, y=0, z=5)]
    ambient = objects.AmbientLight(intensity=1 / 10, color=WHITE)
    camera = objects.Camera(x=0, y=1, z=4, pixels_x=width, pixels_y=height)
    return Scene([ground, blue, yellow], lights, camera, camera.screen, ambient)


scene = build()
