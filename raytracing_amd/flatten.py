"""Scene flattening: duck-typed scene objects -> structure-of-arrays buffers.

Everything the device path needs is gathered here once per render, in
float64, from the attributes the reference's tracer reads:

* spheres: ``position``, ``radius`` and the twelve ``material`` parameters
  (reference objects.py:54-58, materials.py:4-29).  Colours that the reference
  multiplies on every use are pre-multiplied once: ``diffuse_color *
  diffuse_coefficient`` and ``specular_color * specular_coefficient``
  (reference raytrace.py:243-252).
* lights: kind, position, intensity, colours, and for disk lights the radius,
  sample count and the (not normalised) in-plane frame ``x_hat = normal x perp``,
  ``y_hat = normal x x_hat`` (reference objects.py:124-130), for spotlights the
  tangents of the cone and fall-off angles (reference objects.py:209-210) and
  the easing mode.
* ambient term ``intensity * color`` (reference raytrace.py:137-138), camera
  position and the screen frame (reference objects.py:45-51).
"""
from dataclasses import dataclass

import numpy as np

LIGHT_POINT, LIGHT_DISK, LIGHT_SPOT = 0, 1, 2
EASING_CODES = {"none": 0, "linear": 1, "quadratic": 2, "cubic": 3, "exponential": 4}

# columns of FlatScene.sphere_mat
MAT_AMBIENT, MAT_DIFFUSE, MAT_SPECULAR = 0, 3, 6
MAT_SHININESS, MAT_REFLECTION, MAT_TRANSPARENCY = 9, 10, 11
MAT_REFRACTIVE_INDEX, MAT_SMOOTHNESS, MAT_ATTENUATION, MAT_ABSORPTION = 12, 13, 14, 15
MAT_WIDTH = 18

MAX_LIGHTS = 8
MAX_SPHERES = 65535


@dataclass
class FlatLight:
    kind: int
    position: np.ndarray
    intensity: float
    diffuse_color: np.ndarray
    specular_color: np.ndarray
    radius: float = 0.0
    n_points: int = 1
    normal: np.ndarray = None
    x_hat: np.ndarray = None
    y_hat: np.ndarray = None
    tan_angle: float = 0.0
    tan_outer: float = 0.0
    easing: int = 0

    @property
    def draws_samples(self):
        return self.kind != LIGHT_POINT


@dataclass
class FlatScene:
    sphere_geom: np.ndarray      # (S, 4) centre xyz, radius
    sphere_mat: np.ndarray       # (S, 18) see MAT_* columns
    lights: list                 # [FlatLight]
    ambient: np.ndarray          # (3,) intensity * colour, zeros when no ambient light
    camera_position: np.ndarray  # (3,)
    screen_position: np.ndarray  # (3,)
    screen_x: np.ndarray         # (3,)
    screen_y: np.ndarray         # (3,)
    screen_width: float
    screen_height: float
    pixels_x: float
    pixels_y: float

    @property
    def n_spheres(self):
        return self.sphere_geom.shape[0]

    @property
    def has_sampled_lights(self):
        return any(l.draws_samples for l in self.lights)


def _vec3(v):
    a = np.array(v, dtype=float).reshape(-1)
    if a.shape != (3,):
        raise ValueError(f"expected a 3-vector, got shape {a.shape}")
    return a


def _disk_frame(normal):
    """In-plane frame of a disk light (reference objects.py:124-130)."""
    if normal[0] != 0 and normal[1] == 0 and normal[2] == 0:
        perp = np.array([0.0, 1.0, 0.0])
    else:
        perp = np.array([1.0, 0.0, 0.0])
    x_hat = np.cross(normal, perp)
    y_hat = np.cross(normal, x_hat)
    return x_hat, y_hat


def _flatten_light(light):
    common = dict(position=_vec3(light.position), intensity=float(light.intensity),
                  diffuse_color=_vec3(getattr(light, "diffuse_color", (1, 1, 1))),
                  specular_color=_vec3(getattr(light, "specular_color", (1, 1, 1))))
    is_disk = hasattr(light, "n_points") and hasattr(light, "radius")
    if not is_disk:
        return FlatLight(kind=LIGHT_POINT, **common)
    normal = _vec3(light.normal_vector)
    x_hat, y_hat = _disk_frame(normal)
    n_points = int(light.n_points)
    if n_points < 1:
        raise ValueError("a disk light needs at least one sample point")
    disk = dict(radius=float(light.radius), n_points=n_points, normal=normal,
                x_hat=x_hat, y_hat=y_hat)
    # A disk light evaluates its samples through a fresh PointSource(intensity / n_points), whose
    # colours are always white (reference objects.py:140-144): the disk light's own colour
    # attributes never reach the image.
    common["diffuse_color"] = np.ones(3)
    common["specular_color"] = np.ones(3)
    if hasattr(light, "angle") and hasattr(light, "easing_mode"):
        fall_off = float(getattr(light, "fall_off_angle", 20 * np.pi / 180))
        # An unknown easing string leaves the factor at one, like "none"
        # (reference objects.py:176-191).
        easing = EASING_CODES.get(light.easing_mode, 0)
        return FlatLight(kind=LIGHT_SPOT, tan_angle=float(np.tan(light.angle)),
                         tan_outer=float(np.tan(light.angle + fall_off)), easing=easing,
                         **disk, **common)
    return FlatLight(kind=LIGHT_DISK, **disk, **common)


def flatten_scene(scene):
    """Read a duck-typed scene (this package's classes or the reference's) into
    a :class:`FlatScene`."""
    spheres = list(scene.objects)
    if len(spheres) > MAX_SPHERES:
        raise ValueError(f"at most {MAX_SPHERES} spheres are supported")
    if len(scene.light_sources) > MAX_LIGHTS:
        raise ValueError(f"at most {MAX_LIGHTS} light sources are supported")
    geom = np.zeros((len(spheres), 4))
    mat = np.zeros((len(spheres), MAT_WIDTH))
    for i, s in enumerate(spheres):
        m = s.material
        geom[i, :3] = _vec3(s.position)
        geom[i, 3] = float(s.radius)
        mat[i, MAT_AMBIENT:MAT_AMBIENT + 3] = _vec3(m.ambient_color)
        mat[i, MAT_DIFFUSE:MAT_DIFFUSE + 3] = _vec3(m.diffuse_color) * m.diffuse_coefficient
        mat[i, MAT_SPECULAR:MAT_SPECULAR + 3] = _vec3(m.specular_color) * m.specular_coefficient
        mat[i, MAT_SHININESS] = m.shininess
        mat[i, MAT_REFLECTION] = m.reflection_coefficient
        mat[i, MAT_TRANSPARENCY] = m.transparency_coefficient
        mat[i, MAT_REFRACTIVE_INDEX] = m.refractive_index
        mat[i, MAT_SMOOTHNESS] = m.smoothness
        mat[i, MAT_ATTENUATION] = m.attenuation_coefficient
        mat[i, MAT_ABSORPTION:MAT_ABSORPTION + 3] = _vec3(m.absorption_color)

    ambient = np.zeros(3)
    if getattr(scene, "ambient_light", None) is not None:
        ambient = scene.ambient_light.intensity * _vec3(scene.ambient_light.color)

    screen = scene.screen
    return FlatScene(
        sphere_geom=geom, sphere_mat=mat,
        lights=[_flatten_light(l) for l in scene.light_sources],
        ambient=ambient,
        camera_position=_vec3(scene.camera.position),
        screen_position=_vec3(screen.position),
        screen_x=_vec3(screen.x_vector), screen_y=_vec3(screen.y_vector),
        screen_width=float(screen.width), screen_height=float(screen.height),
        pixels_x=float(screen.pixels_x), pixels_y=float(screen.pixels_y))
