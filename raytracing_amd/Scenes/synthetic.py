"""Synthetic benchmark scenes that are not part of the reference's example set.

* :func:`spotlight_scene` - ground + one reflective sphere under a
  ``DirectionalDiskSource`` (soft-shadow heavy; one scene per easing mode).
* :func:`many_spheres_scene` - one huge ground sphere plus ``n_spheres - 1``
  small random spheres, 70 % of them glass (intersection bound).

* :func:`tir_scene` - camera inside a glass sphere, a sphere of refractive index
  below 1 and two overlapping glass spheres: thousands of rays take the total
  internal reflection branch (reference raytrace.py:122-127).
* :func:`clip_scene` - a sphere of refractive index 0.8 with strong absorption over
  a mirror-like ground: transmitted rays that leave the surface almost tangentially
  miss the sphere they "entered", the reference's path length is then -1 and its
  attenuation factor exceeds 1 (reference raytrace.py:117-120), so the per-call clip
  (reference raytrace.py:170) changes colours below the top level.  Meant to be
  rendered with ``EPSILON = 1e-2`` (the band of such rays is 2 EPSILON / radius wide).

All are built only from the reference's public classes' parameters, so the
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


def _camera(_objects, width, height, **kwargs):
    if _objects is objects:
        kwargs.update(pixels_x=width, pixels_y=height)
    return _objects.Camera(**kwargs)


def tir_scene(width=None, height=None, _objects=objects, _materials=materials, _scene_cls=Scene):
    """Scene whose rays often meet a surface beyond the critical angle."""
    M = _materials.Material
    ground = _objects.Sphere(x=0, y=0, z=-1000000, radius=1000000,
                             material=M(diffuse_color=WHITE, reflection_coefficient=0, smoothness=0.3))
    # the camera at (0, 1, 4) sits 1.44 from the centre of this sphere of radius 2 and index 1.5:
    # sideways rays meet the wall at up to asin(0.72), beyond the critical angle asin(1 / 1.5)
    bubble = _objects.Sphere(x=1.2, y=1, z=3.2, radius=2.0,
                             material=M(diffuse_color=WHITE, transparency_coefficient=0.95, refractive_index=1.5,
                                        smoothness=0.3, attenuation_coefficient=0.05))
    # index below 1: rays from outside with cos(incidence) < 0.6 are totally reflected on entry
    thin = _objects.Sphere(x=5, y=-0.5, z=1, radius=1.0,
                           material=M(diffuse_color=RED, transparency_coefficient=0.9, refractive_index=0.8,
                                      smoothness=0.5))
    # overlapping glass: a ray inside one sphere meets the other's surface at any angle
    left = _objects.Sphere(x=5, y=2, z=1.0, radius=0.8,
                           material=M(diffuse_color=GREEN, transparency_coefficient=0.9, refractive_index=1.5,
                                      smoothness=0.6))
    right = _objects.Sphere(x=5.6, y=2.4, z=1.2, radius=0.8,
                            material=M(diffuse_color=BLUE, transparency_coefficient=0.8, refractive_index=1.3,
                                       smoothness=0.4, reflection_coefficient=0.1))
    lights = [_objects.PointSource(4, 0, 8, intensity=60)]
    camera = _camera(_objects, width, height, x=0, y=1, z=4)
    ambient = _objects.AmbientLight(intensity=1 / 10, color=WHITE)
    return _scene_cls([ground, bubble, thin, left, right], lights, camera, camera.screen, ambient)


def clip_scene(width=None, height=None, _objects=objects, _materials=materials, _scene_cls=Scene):
    """Scene in which the reference's attenuation factor exceeds 1 (see the module docstring)."""
    M = _materials.Material
    ground = _objects.Sphere(x=0, y=0, z=-1000000, radius=1000000,
                             material=M(diffuse_color=GREY, reflection_coefficient=0.6, smoothness=0.8))
    ball = _objects.Sphere(x=4, y=1, z=1.6, radius=1.5,
                           material=M(diffuse_color=WHITE, transparency_coefficient=0.9, refractive_index=0.8,
                                      smoothness=0.2, attenuation_coefficient=3.0))
    small = _objects.Sphere(x=3, y=-1.2, z=0.5, radius=0.5,
                            material=M(diffuse_color=YELLOW, transparency_coefficient=0.7, refractive_index=0.9,
                                       smoothness=0.1, attenuation_coefficient=2.0))
    lights = [_objects.PointSource(4, 0, 6, intensity=25)]
    camera = _camera(_objects, width, height, x=0, y=1, z=4)
    ambient = _objects.AmbientLight(intensity=1 / 10, color=WHITE)
    return _scene_cls([ground, ball, small], lights, camera, camera.screen, ambient)
