"""Example 3: semi-glossy ground (radius 1e7, default x=4) and two fully
transparent spheres with refractive indices 1.03 and 1.5 (same content as
reference Scenes/example_3.py:6-23)."""
import numpy as np

from .. import materials, objects
from ..constants import WHITE
from .Scene import Scene


def build(width=None, height=None):
    ground = objects.Sphere(z=-10000000, radius=10000000,
                            material=materials.Material(diffuse_color=WHITE,
                                                        reflection_coefficient=0, smoothness=0.5))
    near = objects.Sphere(x=4, y=0, z=3, radius=3,
                          material=materials.Material(diffuse_color=WHITE, reflection_coefficient=0,
                                                      transparency_coefficient=1, refractive_index=1.03,
                                                      smoothness=1))
    far = objects.Sphere(x=30, y=0, z=5, radius=5,
                         material=materials.Material(diffuse_color=WHITE, reflection_coefficient=0,
                                                     transparency_coefficient=1, refractive_index=1.5,
                                                     smoothness=1))
    lights = [objects.PointSource(x=4, y=20, z=10, intensity=50 ** 2)]
    camera = objects.Camera(x=-4, y=1, z=1, viewing_direction=np.array([1.0, 0.0, -0.1]),
                            pixels_x=width, pixels_y=height)
    ambient = objects.AmbientLight(intensity=1 / 10, color=WHITE)
    return Scene([ground, near, far], lights, camera, camera.screen, ambient)


scene = build()
