"""Synthetic benchmark scenes that are not part of the reference's example set.

* :func:`spotlight_scene` - ground + one reflective sphere under a
  ``DirectionalDiskSource`` (soft-shadow heavy; one scene per easing mode).
* :func:`many_spheres_scene` - one huge ground sphere plus ``n_spheres - 1``
  small random spheres, 70 % of them glass (intersection bound).

Both are built only from the reference's public classes' parameters, so the
same generator run against the reference's classes yields the same scene; the
golden images under ``tests/golden`` were produced that way.
"""
import numpy as np

from .. import materials, objects
from ..constants import BLUE, GREEN, GREY, RED, WHITE, YELLOW
from .Scene import Scene


def spotlight_scene(easing_mode="quadratic", n_points=64, width=None, height=None,
                    _objects=objects, _materials=materials, _scene_cls=Scene):
    """Ground (r=1e6) and a yellow half-mirror sphere lit by a 20 degree
    spotlight of radius 1 at (4, 0, 5); ``n_points`` disk samples per hit."""
    ground = _objects.Sphere(x=0, y=0, z=-1000000, radius=1000000,
                             material=_materials.Material(diffuse_color=WHITE,
                                                          reflection_coefficient=0, smoothness=1))
    ball = _objects.Sphere(y=2, z=1.25, radius=0.5,
                           material=_materials.Material(diffuse_color=YELLOW,
                                                        reflection_coefficient=0.5))
    light = _objects.DirectionalDiskSource(x=4, y=0, z=5, radius=1, intensity=15, angle=20,
                                           easing_mode=easing_mode)
    light.n_points = n_points
    cam_kwargs = {}
    if _objects is objects:
        cam_kwargs = dict(pixels_x=width, pixels_y=height)
    camera = _objects.Camera(x=0, y=1, z=4, **cam_kwargs)
    ambient = _objects.AmbientLight(intensity=1 / 10, color=WHITE)
    return _scene_cls([ground, ball], [light], camera, camera.screen, ambient)


def many_spheres_scene(n_spheres=4096, width=None, height=None,
                       _objects=objects, _materials=materials, _scene_cls=Scene):
    """Sphere 0 is the example_3 ground; spheres 1.. are drawn from
    ``np.random.RandomState(4096)`` attribute by attribute (centre x, y, z,
    radius, glass flag, transparency, refractive index, glass smoothness,
    attenuation, reflection, opaque smoothness), each as one array of length
    ``n_spheres - 1``."""
    m = n_spheres - 1
    rs = np.random.RandomState(4096)
    cx = rs.uniform(2, 60, m)
    cy = rs.uniform(-25, 25, m)
    cz = rs.uniform(0.3, 12, m)
    rad = rs.uniform(0.15, 0.6, m)
    glass = rs.random_sample(m) < 0.7
    transp = rs.uniform(.6, 1, m)
    ref_index = rs.uniform(1.03, 1.6, m)
    smooth_glass = rs.uniform(.5, 1, m)
    att = rs.uniform(.1, .5, m)
    refl = rs.uniform(0, .5, m)
    smooth_opaque = rs.uniform(0, 1, m)
    palette = [WHITE, RED, GREEN, BLUE, YELLOW, GREY]

    spheres = [_objects.Sphere(z=-10000000, radius=10000000,
                               material=_materials.Material(diffuse_color=WHITE,
                                                            reflection_coefficient=0,
                                                            smoothness=0.5))]
    for i in range(m):
        col = palette[(i + 1) % 6]
        if glass[i]:
            mat = _materials.Material(diffuse_color=col, transparency_coefficient=transp[i],
                                      refractive_index=ref_index[i], smoothness=smooth_glass[i],
                                      attenuation_coefficient=att[i])
        else:
            mat = _materials.Material(diffuse_color=col, reflection_coefficient=refl[i],
                                      smoothness=smooth_opaque[i])
        spheres.append(_objects.Sphere(x=cx[i], y=cy[i], z=cz[i], radius=rad[i], material=mat))

    lights = [_objects.PointSource(4, 20, 30, intensity=50 ** 2)]
    cam_kwargs = {}
    if _objects is objects:
        cam_kwargs = dict(pixels_x=width, pixels_y=height)
    camera = _objects.Camera(x=-4, y=1, z=1, viewing_direction=np.array([1.0, 0.0, -0.1]),
                             **cam_kwargs)
    ambient = _objects.AmbientLight(intensity=1 / 10, color=WHITE)
    return _scene_cls(spheres, lights, camera, camera.screen, ambient)
