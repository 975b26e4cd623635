"""Scene building blocks: camera, screen, spheres and light sources.

These classes are *parameter holders* with the constructor signatures and
defaults of the reference (reference objects.py:7-171).  They carry no
intersection or lighting arithmetic: the renderer reads their attributes in
:func:`raytracing_amd.flatten.flatten_scene` and everything the reference
computes in ``Sphere.intersection``, ``compute_light_intensity`` and the easing
functions runs in the HIP kernels.  Objects built with the reference's own
classes expose the same attributes and are accepted as well (duck typing).
"""
import numpy as np

from . import constants as _c
from . import materials


class Object:
    """Anything with a position (reference objects.py:7-12)."""

    def __init__(self, x=0, y=0, z=0):
        self.x, self.y, self.z = x, y, z
        self.position = np.array([x, y, z], dtype=float)


class Screen(Object):
    """Image plane (reference objects.py:32-43).

    ``pixels_x``/``pixels_y`` default to the current
    ``constants.WIDTH``/``HEIGHT``; the screen is ``width`` units wide and
    ``width * pixels_y / pixels_x`` units high.  ``x_vector`` is
    ``normal_vector x y_vector`` and, as in the reference, is not
    re-normalised.
    """

    def __init__(self, x=1, y=0, z=0, normal_vector=None, y_vector=None, width=1,
                 pixels_x=None, pixels_y=None):
        super().__init__(x, y, z)
        px = _c.WIDTH if pixels_x is None else pixels_x
        py = _c.HEIGHT if pixels_y is None else pixels_y
        self.width = width
        self.height = width * py / px
        self.pixels_x = float(px)
        self.pixels_y = float(py)
        self.normal_vector = np.array([1, 0, 0]) if normal_vector is None else normal_vector
        self.y_vector = np.array([0, 0, 1]) if y_vector is None else y_vector
        self.x_vector = np.cross(self.normal_vector, self.y_vector)


class Camera(Object):
    """Pin-hole camera with its screen one unit ahead (reference objects.py:15-29).

    The caller's ``viewing_direction`` array is normalised in place, exactly
    as the reference does (reference objects.py:20).  The up vector starts as
    (0.1, 0, 0.97) and is made orthogonal to the viewing direction by two
    cross products whenever their dot product is non-zero.
    """

    def __init__(self, x=0, y=0, z=0, viewing_direction=None, pixels_x=None, pixels_y=None):
        super().__init__(x, y, z)
        if viewing_direction is None:
            viewing_direction = np.array([1 / 2 ** 0.5, 0, -1 / 2 ** 0.5])
        viewing_direction /= np.linalg.norm(viewing_direction)
        self.viewing_direction = viewing_direction
        up = np.array([0.1, 0, 0.97])
        up /= np.linalg.norm(up)
        if np.dot(viewing_direction, up) != 0:
            side = np.cross(viewing_direction, up)
            up = np.cross(side, viewing_direction)
            up = up / np.linalg.norm(up)
        self.y_vector = up
        sx, sy, sz = self.position + viewing_direction
        self.screen = Screen(sx, sy, sz, normal_vector=-viewing_direction, y_vector=self.y_vector,
                             pixels_x=pixels_x, pixels_y=pixels_y)


class Sphere(Object):
    """Sphere primitive (reference objects.py:54-58).  Note the reference's
    default centre is x=4, and the default material object is shared."""

    def __init__(self, x=4, y=0, z=0, radius=1.0, material=materials.Material(_c.YELLOW)):
        super().__init__(x, y, z)
        self.radius = radius
        self.material = material


class AmbientLight:
    """Constant ambient term (reference objects.py:84-87)."""

    def __init__(self, intensity=0, color=_c.WHITE):
        self.intensity = intensity
        self.color = color


class LightSource(Object):
    """Common light parameters (reference objects.py:69-77)."""

    def __init__(self, x=4.0, y=0.0, z=1000.0, intensity=1.05):
        super().__init__(x, y, z)
        self.intensity = intensity
        self.ambient_intensity = self.intensity / 50
        self.normal_vector = np.array([0.0, 0.0, -1.0])
        self.ambient_color = _c.WHITE.copy()
        self.diffuse_color = _c.WHITE.copy()
        self.specular_color = _c.WHITE.copy()


class PointSource(LightSource):
    """Point light with inverse-square fall-off (reference objects.py:90-92)."""

    def __init__(self, x=4.0, y=0.0, z=20.0, intensity=15.0):
        super().__init__(x, y, z, intensity=intensity)


class DiskSource(LightSource):
    """Disk-shaped area light sampled with ``n_points`` uniform disk samples
    per hit (reference objects.py:113-117)."""

    def __init__(self, x=4.0, y=0.0, z=20.0, radius=3.0, intensity=15.0, N=30):
        super().__init__(x, y, z, intensity=intensity)
        self.radius = radius
        self.n_points = N


class EasingModes:
    NONE = "none"
    LINEAR = "linear"
    QUADRATIC = "quadratic"
    CUBIC = "cubic"
    EXPONENTIAL = "exponential"


class DirectionalDiskSource(DiskSource):
    """Spotlight: a disk light restricted to a cone of half-angle ``angle``
    degrees with a 20 degree eased fall-off band (reference objects.py:165-174).
    ``n_points`` is fixed at 30 by the constructor; change it by attribute."""

    def __init__(self, x=4, y=0, z=20, radius=3, intensity=15, angle=30,
                 easing_mode=EasingModes.QUADRATIC):
        super().__init__(x, y, z, radius=radius, intensity=intensity)
        self.angle = angle * np.pi / 180
        self.fall_off_angle = 20 * np.pi / 180
        self.easing_mode = easing_mode
        self.n_points = 30
        if angle == 90:
            print("Using a directional disk source with an angle of 90 degrees is not recommended. "
                  "Use DiskSource instead.")
