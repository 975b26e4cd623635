"""Example 4: mirror-like ground, a green glass sphere and a red sphere lit by
a disk light (soft shadows, 30 samples per hit) (same content as reference
Scenes/example_4.py:6-22)."""
import numpy as np

from .. import materials, objects
from ..constants import GREEN, RED, WHITE
from .Scene import Scene


def build(width=None, height=None):
    ground = objects.Sphere(z=-10000000, radius=10000000,
                            material=materials.Material(diffuse_color=WHITE,
                                                        reflection_coefficient=0, smoothness=1))
    green = objects.Sphere(x=0, y=0, z=1, radius=1,
                           material=materials.Material(diffuse_color=GREEN, reflection_coefficient=0,
                                                       specular_coefficient=1,
                                                       transparency_coefficient=0.9,
                                                       refractive_index=1.05, smoothness=1,
                                                       attenuation_coefficient=0.5))
    red = objects.Sphere(x=15, y=0, z=2, radius=4,
                         material=materials.Material(diffuse_color=RED, smoothness=0.6))
    lights = [objects.DiskSource(x=4, y=0, z=10, intensity=10 ** 2)]
    camera = objects.Camera(x=-4, y=1, z=1, viewing_direction=np.array([1.0, 0.0, -0.1]),
                            pixels_x=width, pixels_y=height)
    ambient = objects.AmbientLight(intensity=1 / 10, color=WHITE)
    return Scene([ground, green, red], lights, camera, camera.screen, ambient)


scene = build()
