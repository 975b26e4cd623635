"""CPU oracle: a NumPy float64 restatement of the reference's ray tracer.

TEST INFRASTRUCTURE ONLY.  Nothing in ``raytracing_amd`` imports this module;
it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / CPU baseline, never as
the rendering path.

Parity status: PINNED.  ``tests/golden/make_golden.py`` runs the reference
itself (imported from its own checkout) and stores its images, per-call traces
and unit vectors; ``tests/test_oracle_golden.py`` holds this module to those
files (images to 1e-9, traces and ray counts exactly).

The algorithm follows the reference call for call:

* ``pixel_colors``        - reference raytrace.py:6-16 and objects.py:45-51
* ``_node``               - reference raytrace.py:19-170 (one recursive call)
* ``_nearest``            - reference objects.py:288-299
* ``_roots``              - reference objects.py:238-271
* ``_sphere_roots``       - reference objects.py:60-66
* ``_shadow_multiplier``  - reference objects.py:274-285
* ``_point_intensity``    - reference objects.py:99-110
* ``_disk_light``         - reference objects.py:119-154
* ``_spot_light``         - reference objects.py:193-235, easing :302-338
* ``_surface``            - reference raytrace.py:215-234
* Fresnel / refraction / reflection - reference raytrace.py:173-212

but works on the flattened structure-of-arrays scene
(``raytracing_amd.flatten.FlatScene``): per-hit material values are gathered
by sphere index instead of looping over objects, and sphere loops are
evaluated in chunks of columns.  Random numbers come from the legacy global
``numpy.random`` stream in the same order and block sizes as the reference, so
``np.random.seed(k)`` before a render reproduces the reference's soft shadows.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from raytracing_amd import flatten as F
from raytracing_amd.constants import RenderParams


@dataclass
class NodeCall:
    """One recursive call, recorded in call (pre-)order."""
    a: int
    b: int
    rays: int
    hits: int = 0
    beam: tuple = ()
    path: str = ""


@dataclass
class OracleStats:
    calls: list = field(default_factory=list)
    shadow_rays: int = 0
    uniforms_drawn: int = 0

    def traced_rays(self, small_num=3):
        """Rays handed to the nearest-hit search: batches below ``small_num``
        return before it (reference raytrace.py:26-27)."""
        return sum(c.rays for c in self.calls if c.rays >= small_num)

    def hits(self):
        return sum(c.hits for c in self.calls)

    def visited(self, small_num=3):
        return sum(1 for c in self.calls if c.rays >= small_num)


def _dot(u, v):
    return np.sum(u * v, axis=-1)


def _reflect(d, n):
    return d - 2 * n * _dot(n, d)[:, None]


def _roots(B, C, furthest):
    """Roots of t^2 + B t + C = 0 chosen as the reference does: the smaller
    positive one ("closest") or the larger positive one ("furthest"), -1 when
    there is no positive real root."""
    out = -np.ones(B.shape)
    disc = B ** 2 - 4 * C
    real = 0 <= disc
    half_root = disc[real] ** 0.5 / 2
    mid = -B[real] / 2
    x1 = mid + half_root
    x2 = mid - half_root
    lo = np.minimum(x1, x2)
    hi = np.maximum(x1, x2)
    if furthest:
        val = np.where(hi > 0, hi, np.where(lo > 0, lo, -1.0))
    else:
        val = np.where(lo > 0, lo, np.where(hi > 0, hi, -1.0))
    out[real] = val
    return out


class Oracle:
    def __init__(self, flat, params=None, rng=None):
        self.s = flat
        self.p = params if params is not None else RenderParams()
        self.rng = np.random if rng is None else rng
        self.stats = OracleStats()
        self.centers = flat.sphere_geom[:, :3]
        self.radii = flat.sphere_geom[:, 3]
        self.mat = flat.sphere_mat
        self.bg = np.asarray(self.p.background_color, dtype=float)

    # ------------------------------------------------------------------ entry
    def render(self):
        """Whole image, ``(height, width, 3)`` (reference raytrace.py:273-279)."""
        H, W = self.p.height, self.p.width
        Y, X = np.indices((H, W), dtype=float)
        img = self.pixel_colors(X.reshape(-1, 1), Y.reshape(-1, 1))
        return img.reshape(H, W, 3)

    def pixel_colors(self, X, Y):
        s = self.s
        X = np.asarray(X, dtype=float).reshape(-1, 1)
        Y = np.asarray(Y, dtype=float).reshape(-1, 1)
        sx = X * float(s.screen_width) / s.pixels_x - float(s.screen_width) / 2
        sy = (s.pixels_y - Y) * float(s.screen_height) / s.pixels_y - float(s.screen_height) / 2
        pix = sx * s.screen_x + sy * s.screen_y + s.screen_position
        d = pix - s.camera_position
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        o = np.full(d.shape, s.camera_position)
        col = self._node(o, d, self.p.max_reflection_depth, self.p.max_refraction_depth, "")
        return np.clip(col, 0, 1)

    # ------------------------------------------------------------- geometry
    def _chunk(self, n_rows):
        S = self.centers.shape[0]
        return int(max(1, min(S, 4_000_000 // max(n_rows, 1))))

    def _root_matrix(self, o, d, lo, hi, furthest):
        """Roots of every ray against spheres lo..hi-1, shape (n, hi-lo)."""
        c = self.centers[lo:hi]
        r = self.radii[lo:hi]
        B = 2 * (_dot(d, o)[:, None] - d @ c.T)
        diff = c[None, :, :] - o[:, None, :]
        C = np.sum(diff * diff, axis=-1) - r ** 2
        return _roots(B, C, furthest)

    def _sphere_roots(self, sid, o, d, furthest):
        """Per-ray root against the ray's own sphere ``sid``.  The product of direction and
        centre is taken per sphere with ``np.dot`` on that sphere's rows, as the reference's
        ``Sphere.intersection`` does (reference objects.py:62): BLAS sums in another order than
        ``np.sum``, which shows in the last bits for the ground spheres."""
        c = self.centers[sid]
        d_c = np.empty(d.shape[0])
        for i in np.unique(sid):
            rows = sid == i
            d_c[rows] = np.dot(d[rows], self.centers[i])
        B = 2 * (_dot(d, o) - d_c)
        diff = c - o
        C = _dot(diff, diff) - self.radii[sid] ** 2
        return _roots(B, C, furthest)

    def _nearest(self, o, d):
        n = o.shape[0]
        S = self.centers.shape[0]
        best_t = np.full(n, np.inf)
        best_i = np.full(n, -1)
        step = self._chunk(n)
        for lo in range(0, S, step):
            hi = min(S, lo + step)
            T = self._root_matrix(o, d, lo, hi, False)
            T = np.where(T > 0, T, np.inf)
            # ties go to the later sphere (the reference overwrites on equality)
            rev = T[:, ::-1]
            j = (hi - lo - 1) - np.argmin(rev, axis=1)
            t = T[np.arange(n), j]
            better = np.isfinite(t) & (t <= best_t)
            best_t = np.where(better, t, best_t)
            best_i = np.where(better, lo + j, best_i)
        return best_i, best_t

    def _shadow_multiplier(self, o, d):
        m = o.shape[0]
        self.stats.shadow_rays += m
        S = self.centers.shape[0]
        mult = np.ones((m, 3))
        step = self._chunk(3 * m)
        for lo in range(0, S, step):
            hi = min(S, lo + step)
            near = self._root_matrix(o, d, lo, hi, False)
            far = self._root_matrix(o, d, lo, hi, True)
            ok = far > 0
            chord = np.where(ok, far - near, 0.0)
            tr = self.mat[lo:hi, F.MAT_TRANSPARENCY]
            coef = -self.mat[lo:hi, F.MAT_ATTENUATION, None] * self.mat[lo:hi, F.MAT_ABSORPTION:F.MAT_ABSORPTION + 3]
            factor = tr[None, :, None] * np.exp(coef[None, :, :] * chord[:, :, None])
            factor = np.where(ok[:, :, None], factor, 1.0)
            for j in range(hi - lo):
                mult *= factor[:, j, :]
        return mult

    # ---------------------------------------------------------------- lights
    def _point_intensity(self, light, intensity, P, L):
        """Unit light vectors and clipped RGB intensities for light vectors ``L``
        from points ``P`` (inverse-square fall-off times shadow multiplier)."""
        dist = np.linalg.norm(L, axis=-1, keepdims=True)
        Lu = L / dist
        mult = self._shadow_multiplier(P, Lu)
        i_d = light.diffuse_color * intensity / dist ** 2 * mult
        i_s = light.specular_color * intensity / dist ** 2 * mult
        return np.clip(i_d, 0, 1), np.clip(i_s, 0, 1), Lu

    def _draw(self, light_index, count):
        """The two uniform blocks of one disk-light evaluation: all angles first,
        then all radii (reference objects.py:132-133).  Subclasses may serve them
        from an uploaded buffer instead of the random stream."""
        return self.rng.random(count), self.rng.random(count)

    def _disk_light(self, light, P, light_index=0):
        h = P.shape[0]
        N = light.n_points
        tiled = np.tile(P, (N, 1))
        u_theta, u_rad = self._draw(light_index, N * h)
        theta = u_theta * 2 * math.pi
        rad = u_rad ** 0.5 * light.radius
        self.stats.uniforms_drawn += 2 * N * h
        local = rad[:, None] * (np.cos(theta)[:, None] * light.x_hat + np.sin(theta)[:, None] * light.y_hat)
        L = (light.position + local) - tiled
        i_d, i_s, Lu = self._point_intensity(light, light.intensity / N, tiled, L)
        i_d = np.sum(i_d.reshape(N, h, 3), axis=0) / N
        i_s = np.sum(i_s.reshape(N, h, 3), axis=0) / N
        return np.clip(i_d, 0, 1), np.clip(i_s, 0, 1), Lu.reshape(N, h, 3)

    @staticmethod
    def _ease(mode, x, a, d):
        out = np.ones(x.shape)
        ok = d != 0
        x, a, d = x[ok], a[ok], d[ok]
        if mode == 1:
            val = np.minimum(np.maximum((a - x) / d + 1, 0), 1)
        elif mode in (2, 3):
            before = x < a
            after = x > a + d
            mid = ~(before | after)
            val = np.zeros(x.shape)
            xm, am, dm = x[mid], a[mid], d[mid]
            if mode == 2:
                val[mid] = (xm - am + dm) * (1 / dm - (xm - am) / dm ** 2)
            else:
                q = (4 * am + 3 * dm) / dm ** 3
                p = (1 - q * dm ** 2) / (dm ** 2 * (dm + 2 * am))
                val[mid] = 1 - (xm - am) * (p * (xm ** 2 - (am + dm) ** 2) + q * (xm - (am + dm)) + 1 / dm)
            val[before] = 1
        elif mode == 4:
            val = 1 - 1 / (1 + np.exp(-10 / d * (x - a - d / 2)))
        else:
            return out
        out[ok] = val
        return out

    def beam_geometry(self, light, P):
        """(distance from the beam axis, allowed distance, fall-off width, in-beam mask)."""
        rel = P - light.position
        x = _dot(light.x_hat, rel)
        y = _dot(light.y_hat, rel)
        z = _dot(light.normal, rel)
        rho = (x ** 2 + y ** 2) ** 0.5
        allowed = light.radius + light.tan_angle * np.abs(z)
        edge = light.radius + light.tan_outer * np.abs(z)
        return rho, allowed, edge - allowed, rho <= edge

    def _spot_light(self, light, P, light_index=0):
        h = P.shape[0]
        N = light.n_points
        rho, allowed, width, inside = self.beam_geometry(light, P)
        i_d, i_s, Lu = self._disk_light(light, P[inside], light_index)
        factor = self._ease(light.easing, rho[inside], allowed[inside], width[inside])[:, None]
        tot_d = np.zeros((h, 3))
        tot_s = np.zeros((h, 3))
        tot_d[inside] = i_d * factor
        tot_s[inside] = i_s * factor
        vecs = np.zeros((N, h, 3))
        vecs[:, inside, :] = Lu
        return np.clip(tot_d, 0, 1), np.clip(tot_s, 0, 1), vecs, int(inside.sum())

    def _surface(self, sid, d, nrm, i_d, i_s, vecs):
        diffuse = self.mat[sid, F.MAT_DIFFUSE:F.MAT_DIFFUSE + 3]
        specular = self.mat[sid, F.MAT_SPECULAR:F.MAT_SPECULAR + 3]
        shin = self.mat[sid, F.MAT_SHININESS][:, None]
        surf = np.zeros(d.shape)
        for Lk in vecs:
            nl = _dot(nrm, Lk)[:, None]
            refl = -2 * nrm * nl + Lk
            rd = np.abs(_dot(refl, d))[:, None]
            k_d = np.clip(diffuse * nl, 0, 1)
            k_s = np.clip(specular * rd ** shin, 0, 1)
            surf += np.clip(k_d * i_d, 0, 1)
            surf += np.clip(k_s * i_s, 0, 1)
        return np.clip(surf, 0, 1)

    # ------------------------------------------------------------ per-hit pieces
    def _hit_record(self, o, d, t, sid):
        """Geometry of a batch of hits (reference raytrace.py:53-99): hit point,
        outward normal, inside flag, refraction normal, pulled-back point,
        transmitted mask and direction, Fresnel reflectance."""
        p = self.p
        h = o.shape[0]
        P = o + t * d
        nrm = P - self.centers[sid]
        nrm = nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)
        inside = _dot(d, nrm) > 0
        ior = self.mat[sid, F.MAT_REFRACTIVE_INDEX]
        air = float(p.air_refractive_index)
        n1 = np.where(inside, ior, air)
        n2 = np.where(inside, air, ior)
        N = np.where(inside[:, None], nrm, -nrm)
        Pb = P - p.epsilon * N

        mu = n1 / n2
        cos_i = _dot(N, d)
        k = 1 - mu ** 2 * (1 - cos_i ** 2)
        trans = k >= 0
        T = np.zeros((h, 3))
        T[trans] = (k[trans] ** 0.5)[:, None] * N[trans] + mu[trans, None] * (
            d[trans] - cos_i[trans, None] * N[trans])

        R0 = ((n1 - n2) / (n1 + n2)) ** 2
        cos_f = np.where(n2 >= n1, _dot(d, N), _dot(T, N))
        R = R0 + (1 - R0) * (1 - cos_f) ** 5
        R[~trans] = 1
        return P, nrm, inside, N, Pb, trans, T, R

    def _lighting(self, sid, d, nrm, Pb):
        """Clipped surface colour of a batch of hits under all lights plus ambient
        (reference raytrace.py:136-152); also the in-beam count per light."""
        h = d.shape[0]
        surf = np.zeros((h, 3))
        beams = []
        for li, light in enumerate(self.s.lights):
            if light.kind == F.LIGHT_POINT:
                i_d, i_s, Lu = self._point_intensity(light, light.intensity, Pb, light.position - Pb)
                vecs = Lu[None, :, :]
                beams.append(0)
            elif light.kind == F.LIGHT_DISK:
                i_d, i_s, vecs = self._disk_light(light, Pb, li)
                beams.append(0)
            else:
                i_d, i_s, vecs, nb = self._spot_light(light, Pb, li)
                beams.append(nb)
            surf += self._surface(sid, d, nrm, i_d, i_s, vecs)
        surf += np.clip(self.mat[sid, F.MAT_AMBIENT:F.MAT_AMBIENT + 3] * self.s.ambient, 0, 1)
        return np.clip(surf, 0, 1), beams

    def _combine(self, sid, inside, R, surf, refr, refl):
        """Weighted sum of surface, refraction and reflection colours of a batch of
        hits (reference raytrace.py:155-168, before the final clip)."""
        floor = np.where(inside, 0.0, self.mat[sid, F.MAT_REFLECTION])
        A = (floor + (self.mat[sid, F.MAT_SMOOTHNESS] - floor) * R)[:, None]
        tr = self.mat[sid, F.MAT_TRANSPARENCY][:, None]
        surf = surf * ((1 - tr) * (1 - A))
        refr = refr * (tr * (1 - A))
        refl = refl * A
        return surf + refr + refl

    def _absorption(self, sid, Pb, T):
        """RGB factor for light that crosses the sphere it just entered
        (reference raytrace.py:117-119)."""
        through = self._sphere_roots(sid, Pb, T, True)
        coef = -self.mat[sid, F.MAT_ATTENUATION, None] * self.mat[sid, F.MAT_ABSORPTION:F.MAT_ABSORPTION + 3]
        return np.exp(coef * through[:, None])

    # ------------------------------------------------------------ recursion
    def _node(self, o, d, a, b, path):
        p = self.p
        n = o.shape[0]
        call = NodeCall(a, b, n, path=path)
        self.stats.calls.append(call)
        if n < p.small_num:
            return np.zeros((n, 3)) + self.bg

        sid, t = self._nearest(o, d)
        hit = sid != -1
        h = int(hit.sum())
        if h == 0:
            return np.ones((n, 3)) * self.bg
        call.hits = h
        out = np.zeros((n, 3))
        out[~hit] = self.bg
        o, d, t, sid = o[hit], d[hit], t[hit][:, None], sid[hit]

        P, nrm, inside, N, Pb, trans, T, R = self._hit_record(o, d, t, sid)

        refr = np.zeros((h, 3))
        refl = np.zeros((h, 3))
        if b != 0:
            refr[trans] = self._node(Pb[trans] + 2 * p.epsilon * N[trans], T[trans], a, b - 1, path + "t")
            entering = trans & ~inside
            if entering.any():
                refr[entering] *= self._absorption(sid[entering], Pb[entering], T[entering])
        if b != 0 and a != 0:
            tir = ~trans
            if tir.any():
                refr[tir] = self._node(Pb[tir], _reflect(d[tir], -N[tir]), a - 1, b - 1, path + "i")
        if a != 0:
            refl = self._node(Pb, _reflect(d, -N), a - 1, b, path + "r")

        surf, beams = self._lighting(sid, d, nrm, Pb)
        call.beam = tuple(beams)
        out[hit] = self._combine(sid, inside, R, surf, refr, refl)
        return np.clip(out, 0, 1)


def render(scene, params=None, seed=None):
    """Convenience wrapper: flatten ``scene``, optionally seed the global
    NumPy stream, render, return ``(image, stats)``."""
    flat = scene if isinstance(scene, F.FlatScene) else F.flatten_scene(scene)
    if params is None:
        params = RenderParams(width=int(flat.pixels_x), height=int(flat.pixels_y))
    if seed is not None:
        np.random.seed(seed)
    orc = Oracle(flat, params)
    img = orc.render()
    return img, orc.stats
