"""Host-side construction of the many-sphere accelerators (uniform grid, point-light occluder maps),
checked without a GPU through the library's host-only exports: every sphere must be listed wherever a
ray that touches it can look for it (the structures only narrow the search down; the float64 test
of the reference, objects.py:60-66, decides)."""
import ctypes as C

import numpy as np

from raytracing_amd import _native


def _spheres(n, seed):
    rs = np.random.RandomState(seed)
    g = np.empty((n + 1, 4))
    g[0] = (4.0, 0.0, -1e7, 1e7)                       # a ground sphere: stays out of the grid
    g[1:, 0] = rs.uniform(2, 60, n)
    g[1:, 1] = rs.uniform(-25, 25, n)
    g[1:, 2] = rs.uniform(0.3, 12, n)
    g[1:, 3] = rs.uniform(0.15, 0.6, n)
    return g


def _grid(geom, density=0.0):
    lib = _native.load_library()
    n = geom.shape[0]
    dims = np.zeros(3, np.int32)
    oc = np.zeros(4)
    n_loose = C.c_int32(0)
    n_items = C.c_int64(0)
    loose = np.zeros(n, np.int32)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    assert lib.rt_build_grid(p(geom, C.c_double), n, density, p(dims, C.c_int32), p(oc, C.c_double), C.byref(n_loose),
                             p(loose, C.c_int32), C.byref(n_items), None, None) == 0
    cells = int(dims.prod())
    first = np.zeros(cells + 1, np.int32)
    items = np.zeros(max(1, n_items.value), np.int32)
    assert lib.rt_build_grid(p(geom, C.c_double), n, density, None, None, None, None, None, p(first, C.c_int32), p(items, C.c_int32)) == 0
    return dims, oc[:3], oc[3], loose[:n_loose.value], first, items[:n_items.value]


def test_grid_lists_every_sphere_in_every_cell_its_ball_reaches():
    geom = _spheres(700, 1)
    dims, origin, h, loose, first, items = _grid(geom)
    assert list(loose) == [0] and dims.min() >= 1 and h > 0
    assert first[-1] == items.size and np.all(np.diff(first) >= 0)
    lists = [set(items[first[c]:first[c + 1]].tolist()) for c in range(first.size - 1)]
    for c in range(first.size - 1):                    # ascending inside a cell
        assert np.all(np.diff(items[first[c]:first[c + 1]]) > 0)
    for i in range(1, geom.shape[0]):
        c, r = geom[i, :3] - origin, geom[i, 3]
        lo = np.clip(np.floor((c - r) / h).astype(int), 0, dims - 1)
        hi = np.clip(np.floor((c + r) / h).astype(int), 0, dims - 1)
        for z in range(lo[2], hi[2] + 1):
            for y in range(lo[1], hi[1] + 1):
                for x in range(lo[0], hi[0] + 1):
                    q = np.array([x, y, z])
                    d = np.maximum(np.maximum(q * h - c, c - (q + 1) * h), 0.0)
                    if np.dot(d, d) <= r * r:          # the ball reaches the cell's box
                        assert i in lists[(z * dims[1] + y) * dims[0] + x]


def test_grid_density_and_empty_scene():
    assert _grid(np.zeros((0, 4)))[0].prod() == 0
    coarse = _grid(_spheres(500, 2), 0.5)[0]
    fine = _grid(_spheres(500, 2), 8.0)[0]
    assert fine.prod() > coarse.prod()


def _light_map(geom, light, in_grid=None):
    lib = _native.load_library()
    n = geom.shape[0]
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    N = C.c_int32(0)
    n_always = C.c_int32(0)
    n_items = C.c_int64(0)
    always = np.zeros(n, np.int32)
    light = np.asarray(light, dtype=np.float64)
    flags = None if in_grid is None else p(np.ascontiguousarray(in_grid, dtype=np.uint8), C.c_uint8)
    assert lib.rt_build_light_map(p(geom, C.c_double), n, p(light, C.c_double), flags, C.byref(N), C.byref(n_always),
                                  p(always, C.c_int32), C.byref(n_items), None, None) == 0
    bins = 6 * N.value * N.value
    first = np.zeros(bins + 1, np.int32)
    items = np.zeros(max(1, n_items.value), np.int32)
    assert lib.rt_build_light_map(p(geom, C.c_double), n, p(light, C.c_double), flags, None, None, None, None,
                                  p(first, C.c_int32), p(items, C.c_int32)) == 0
    return N.value, always[:max(0, n_always.value)], n_always.value, first, items[:n_items.value]


def _bin(q, N):
    """Bin of direction q as the shadow test computes it (float32)."""
    q = q.astype(np.float32)
    a = int(np.argmax(np.abs(q) >= np.abs(q).max()))   # first of the largest components
    face = 2 * a + (1 if q[a] < 0 else 0)
    inv = np.float32(1.0) / np.abs(q[a])
    u, w = q[(a + 1) % 3] * inv, q[(a + 2) % 3] * inv
    iu = min(max(int(np.floor((u * np.float32(0.5) + np.float32(0.5)) * N)), 0), N - 1)
    iw = min(max(int(np.floor((w * np.float32(0.5) + np.float32(0.5)) * N)), 0), N - 1)
    return (face * N + iw) * N + iu


def test_light_map_lists_every_sphere_a_line_through_the_light_touches():
    geom = _spheres(900, 3)
    light = np.array([4.0, 20.0, 30.0])
    in_grid = np.ones(geom.shape[0], np.uint8)
    in_grid[0] = 0
    N, always, n_always, first, items = _light_map(geom, light, in_grid)
    assert n_always >= 1 and 0 in always               # the ground is tested for every ray
    rs = np.random.RandomState(4)
    checked = 0
    for _ in range(4000):
        # a line through the light and a point near some sphere: which spheres does it touch?
        i = rs.randint(1, geom.shape[0])
        target = geom[i, :3] + rs.normal(size=3) * geom[i, 3] * 0.7
        d = target - light
        d /= np.linalg.norm(d)
        oc = geom[:, :3] - light
        b = oc @ d
        dist2 = np.einsum("ij,ij->i", oc, oc) - b * b
        touched = np.nonzero(dist2 <= geom[:, 3] ** 2)[0]
        origin = light + d * rs.uniform(5.0, 60.0)     # where the shadow ray starts, on either side of the spheres
        q = origin - light
        listed = set(always.tolist())
        for side in (q, -q):
            k = _bin(side, N)
            listed |= set(items[first[k]:first[k + 1]].tolist())
        for s in touched:
            assert s in listed
            checked += 1
    assert checked > 3000


def test_light_inside_a_sphere_and_too_many_always():
    geom = _spheres(50, 5)
    geom[7] = (1.0, 2.0, 3.0, 5.0)                     # contains the light
    N, always, n_always, first, items = _light_map(geom, [1.5, 2.0, 3.5])
    assert 7 in always and 7 not in items
    # nothing has a screening entry: every sphere would be tested always, no map is made
    assert _light_map(geom, [1.5, 2.0, 3.5], np.zeros(geom.shape[0], np.uint8))[2] == -1
