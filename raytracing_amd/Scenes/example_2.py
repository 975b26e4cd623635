"""Example 2: slightly glossy ground, a glass-like sphere, the default yellow
sphere and a half-transparent red sphere under one strong point light (same
content as reference Scenes/example_2.py:5-19)."""
import numpy as np

from .. import materials, objects
from ..constants import RED, WHITE
from .Scene import Scene


def build(width=None, height=None):
    ground = objects.Sphere(x=0, y=0, z=-1000000, radius=1000000,
                            material=materials.Material(diffuse_color=WHITE, smoothness=0.4))
    glass = objects.Sphere(x=4, y=0, z=1, radius=1,
                           material=materials.Material(diffuse_color=WHITE, reflection_coefficient=0.1,
                                                       transparency_coefficient=0.9,
                                                       refractive_index=1.05))
    plain = objects.Sphere(x=4, y=2, z=1.25, radius=0.5)
    red = objects.Sphere(x=2.5, y=1.5, z=2, radius=0.5,
                         material=materials.Material(diffuse_color=RED, transparency_coefficient=0.5))
    lights = [objects.PointSource(x=4, y=0, z=10, intensity=10 ** 2)]
    camera = objects.Camera(x=-2, y=1, z=1, viewing_direction=np.array([1.0, 0.0, 0.0]),
                            pixels_x=width, pixels_y=height)
    ambient = objects.AmbientLight(intensity=1 / 10, color=WHITE)
    return Scene([ground, glass, plain, red], lights, camera, camera.screen, ambient)


scene = build()
