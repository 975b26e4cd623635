"""ctypes binding of the C ABI in ``include/raytracing_amd.h``.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc,
gfx950).  There is no CPU fallback: if the library is missing, loading fails
with an error that says how to build it.
"""
import ctypes as C
import os

import numpy as np

from . import flatten as F

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB: another build of the same library (kernel experiments side by side on one GPU box)
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "_rt_hip.so")

RT_MAX_LIGHTS = 8
RT_MAT_WIDTH = 18
RT_OK, RT_ERR_INVALID, RT_ERR_HIP, RT_ERR_OVERFLOW, RT_ERR_STATE = 0, -1, -2, -3, -4
NOALLOC_MAX_SAMPLES = 1 << 20
KERNEL_KINDS = ("trace", "scan", "link", "surface", "combine", "init")


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"raytracing_amd native error {code}: {message}")
        self.code = code


class RtOverflow(RtError):
    """The ray arena was too small for the frame; retry with a larger one."""


class RtLight(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_points", C.c_int32), ("easing", C.c_int32), ("reserved", C.c_int32),
                ("position", C.c_double * 3), ("intensity", C.c_double),
                ("diffuse_color", C.c_double * 3), ("specular_color", C.c_double * 3),
                ("radius", C.c_double), ("normal", C.c_double * 3), ("x_hat", C.c_double * 3),
                ("y_hat", C.c_double * 3), ("tan_angle", C.c_double), ("tan_outer", C.c_double)]


class RtScene(C.Structure):
    _fields_ = [("n_spheres", C.c_int32), ("n_lights", C.c_int32),
                ("sphere_geom", C.POINTER(C.c_double)), ("sphere_mat", C.POINTER(C.c_double)),
                ("lights", C.POINTER(RtLight)),
                ("ambient", C.c_double * 3), ("camera_position", C.c_double * 3),
                ("screen_position", C.c_double * 3), ("screen_x", C.c_double * 3), ("screen_y", C.c_double * 3),
                ("screen_width", C.c_double), ("screen_height", C.c_double),
                ("pixels_x", C.c_double), ("pixels_y", C.c_double)]


class RtParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("max_reflection_depth", C.c_int32), ("max_refraction_depth", C.c_int32),
                ("small_num", C.c_int32), ("reserved", C.c_int32),
                ("epsilon", C.c_double), ("air_refractive_index", C.c_double),
                ("background", C.c_double * 3)]


class RtCounters(C.Structure):
    _fields_ = [("pixels", C.c_int64), ("rays", C.c_int64), ("hits", C.c_int64),
                ("shaded_hits", C.c_int64), ("shaded_samples", C.c_int64), ("fallback_samples", C.c_int64),
                ("stored_records", C.c_int64), ("live_hits", C.c_int64),
                ("arena_rays_used", C.c_int64), ("arena_capacity", C.c_int64),
                ("n_nodes", C.c_int32), ("n_levels", C.c_int32), ("overflow", C.c_int32), ("certify_mismatches", C.c_int32),
                ("generator_words", C.c_int64), ("generator_words_stored", C.c_int64)]


class RtProfile(C.Structure):
    _fields_ = [("ms", C.c_double * 6), ("launches", C.c_int64 * 6)]


_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_I64P = C.POINTER(C.c_int64)
_F64P = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol declared in include/raytracing_amd.h
SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_build_tree": (C.c_int, [C.c_int32, C.c_int32, _I32P, _I32P, _I32P, _I32P, _I32P, _I32P, _I32P, _I32P, _I32P]),
    "rt_create": (C.c_int, [C.c_int32, _P, C.POINTER(_P)]),
    "rt_destroy": (None, [_P]),
    "rt_set_scene": (C.c_int, [_P, C.POINTER(RtScene), C.POINTER(RtParams)]),
    "rt_begin_frame": (C.c_int, [_P, C.c_int32, C.c_int32, _F64P, _F64P, C.c_int64, C.c_int64]),
    "rt_trace_geometry": (C.c_int, [_P]),
    "rt_trace_level": (C.c_int, [_P, C.c_int32]),
    "rt_get_level_batches": (C.c_int, [_P, C.c_int32, _I64P]),
    "rt_set_global_batches": (C.c_int, [_P, C.c_int32, _I64P]),
    "rt_set_batch_buffer": (C.c_int, [_P, _P]),
    "rt_set_batch_buffer_shared": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "rt_set_batch_guess": (C.c_int, [_P, _P]),
    "rt_get_node_counts": (C.c_int, [_P, _I64P, _I64P, _I64P]),
    "rt_set_sample_layout": (C.c_int, [_P, _I64P, C.c_int64]),
    "rt_upload_samples": (C.c_int, [_P, C.c_int64, _F64P, C.c_int64]),
    "rt_upload_samples_noalloc": (C.c_int, [_P, C.c_int64, _F64P, C.c_int64]),
    "rt_build_hierarchy": (C.c_int, [_F64P, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint16),
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "rt_mt19937_jump_poly": (C.c_int, [C.c_int32, C.POINTER(C.c_uint32)]),
    "rt_build_grid": (C.c_int, [_F64P, C.c_int64, C.c_double, _I32P, _F64P, _I32P, _I32P, _I64P, _I32P, _I32P]),
    "rt_build_light_map": (C.c_int, [_F64P, C.c_int64, _F64P, C.POINTER(C.c_uint8), _I32P, _I32P, _I32P, _I64P, _I32P, _I32P]),
    "rt_trim": (C.c_int, [_P]),
    "rt_set_sample_layout_global": (C.c_int, [_P, _I64P, _I64P, _I64P, C.c_int64]),
    "rt_generate_samples": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "rt_set_precision": (C.c_int, [_P, C.c_int32]),
    "rt_alloc_host": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "rt_free_host": (None, [_P]),
    "rt_shade": (C.c_int, [_P]),
    "rt_set_output": (C.c_int, [_P, _P]),
    "rt_read_image": (C.c_int, [_P, _F64P]),
    "rt_quantise_rgb8": (C.c_int, [_P, _P]),
    "rt_read_image_rgb8": (C.c_int, [_P, C.POINTER(C.c_uint8)]),
    "rt_image_device_ptr": (C.c_int, [_P, C.POINTER(_P)]),
    "rt_synchronize": (C.c_int, [_P]),
    "rt_get_counters": (C.c_int, [_P, C.POINTER(RtCounters)]),
    "rt_profile_enable": (C.c_int, [_P, C.c_int32]),
    "rt_get_profile": (C.c_int, [_P, C.POINTER(RtProfile), C.c_int32]),
    "rt_render": (C.c_int, [_P, _F64P]),
    "rt_debug_eval": (C.c_int, [_P, C.c_int32, _F64P, _F64P, C.c_int64]),
}

_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels carry their own copy of the HIP runtime (torch/lib/libamdhip64.so, same
    SONAME as the system's).  One process must use one copy: with two, whichever initialises second
    finds no GPU, and streams or device pointers of one mean nothing to the other.  If torch is
    installed, its copy is loaded first (without importing torch), so this library and a later
    ``import torch`` resolve to the same runtime; without torch the system's runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load_library():
    """Load (once) and return the shared library, with prototypes attached."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` in the repository root "
            "(needs hipcc; no CPU fallback exists).")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc == RT_OK:
        return
    msg = load_library().rt_last_error().decode("utf-8", "replace")
    raise (RtOverflow if rc == RT_ERR_OVERFLOW else RtError)(rc, msg)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _vec3(dst, src):
    for i in range(3):
        dst[i] = float(src[i])


def device_count():
    """GPUs visible to the HIP runtime (0 when there is none)."""
    return int(load_library().rt_device_count())


def mt19937_jump_poly(k):
    """Coefficient words of x^(2^k) mod MT19937's characteristic polynomial (host computation)."""
    out = np.empty(624, dtype=np.uint32)
    _check(load_library().rt_mt19937_jump_poly(int(k), out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def build_hierarchy_native(sphere_geom):
    """The sphere grouping of many-sphere scenes as the library builds it (host code only, no GPU
    needed): ``(sphere_of_slot (blocks, 8, 8) uint16, cluster_bound (blocks, 8, 4) float32,
    block_bound (blocks, 4) float32, loose_spheres int32)``."""
    lib = load_library()
    g = np.ascontiguousarray(sphere_geom, dtype=np.float64).reshape(-1, 4)
    nb, nl = C.c_int32(), C.c_int32()
    gp = g.ctypes.data_as(_F64P)
    _check(lib.rt_build_hierarchy(gp, g.shape[0], C.byref(nb), C.byref(nl), None, None, None, None))
    slots = np.empty((nb.value, 8, 8), dtype=np.uint16)
    clusters = np.empty((nb.value, 8, 4), dtype=np.float32)
    blocks = np.empty((nb.value, 4), dtype=np.float32)
    loose = np.empty(nl.value, dtype=np.int32)
    _check(lib.rt_build_hierarchy(gp, g.shape[0], C.byref(nb), C.byref(nl), slots.ctypes.data_as(C.POINTER(C.c_uint16)),
                                  clusters.ctypes.data_as(C.POINTER(C.c_float)), blocks.ctypes.data_as(C.POINTER(C.c_float)),
                                  loose.ctypes.data_as(C.POINTER(C.c_int32))))
    return slots, clusters, blocks, loose


def build_tree_native(max_reflection_depth, max_refraction_depth):
    """Static ray tree as the library builds it (host code only, no GPU needed)."""
    lib = load_library()
    nk, nl = C.c_int32(), C.c_int32()
    none = C.cast(None, _I32P)
    _check(lib.rt_build_tree(max_reflection_depth, max_refraction_depth, C.byref(nk), C.byref(nl),
                             none, none, none, none, none, none, none))
    K, L = nk.value, nl.value
    arrs = {k: np.zeros(K, dtype=np.int32) for k in ("a", "b", "parent", "branch", "level")}
    arrs["child"] = np.zeros((K, 3), dtype=np.int32)
    arrs["level_start"] = np.zeros(L + 1, dtype=np.int32)
    _check(lib.rt_build_tree(max_reflection_depth, max_refraction_depth, C.byref(nk), C.byref(nl),
                             *[_ptr(arrs[k], C.c_int32) for k in
                               ("a", "b", "parent", "branch", "child", "level", "level_start")]))
    return arrs


class Context:
    """One GPU's renderer state (wraps ``RtContext``)."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        self._h = _P()
        _check(self.lib.rt_create(int(device), _P(stream) if stream else None, C.byref(self._h)))
        self.n_nodes = 0
        self.n_levels = 0
        self.level_start = None
        self.npix = 0
        self._keep = None

    def close(self):
        if self._h:
            self.lib.rt_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- scene ---------------------------------------------------------------------------
    def set_scene(self, flat, params, tree):
        """``flat``: FlatScene, ``params``: RenderParams, ``tree``: RayTree of the depth limits."""
        geom, mat = _f64(flat.sphere_geom), _f64(flat.sphere_mat)
        lights = (RtLight * max(len(flat.lights), 1))()
        for i, l in enumerate(flat.lights):
            L = lights[i]
            L.kind, L.n_points, L.easing = int(l.kind), int(l.n_points), int(l.easing)
            _vec3(L.position, l.position)
            L.intensity = float(l.intensity)
            _vec3(L.diffuse_color, l.diffuse_color)
            _vec3(L.specular_color, l.specular_color)
            L.radius = float(l.radius)
            if l.kind != F.LIGHT_POINT:
                _vec3(L.normal, l.normal); _vec3(L.x_hat, l.x_hat); _vec3(L.y_hat, l.y_hat)
            L.tan_angle, L.tan_outer = float(l.tan_angle), float(l.tan_outer)
        sc = RtScene()
        sc.n_spheres, sc.n_lights = flat.n_spheres, len(flat.lights)
        sc.sphere_geom, sc.sphere_mat = _ptr(geom, C.c_double), _ptr(mat, C.c_double)
        sc.lights = lights
        _vec3(sc.ambient, flat.ambient); _vec3(sc.camera_position, flat.camera_position)
        _vec3(sc.screen_position, flat.screen_position)
        _vec3(sc.screen_x, flat.screen_x); _vec3(sc.screen_y, flat.screen_y)
        sc.screen_width, sc.screen_height = flat.screen_width, flat.screen_height
        sc.pixels_x, sc.pixels_y = flat.pixels_x, flat.pixels_y
        p = RtParams()
        p.width, p.height = int(params.width), int(params.height)
        p.max_reflection_depth, p.max_refraction_depth = int(params.max_reflection_depth), int(params.max_refraction_depth)
        p.small_num = int(params.small_num)
        p.epsilon, p.air_refractive_index = float(params.epsilon), float(params.air_refractive_index)
        _vec3(p.background, params.background_color)
        _check(self.lib.rt_set_scene(self._h, C.byref(sc), C.byref(p)))
        self.n_nodes, self.n_levels = tree.n_nodes, tree.n_levels
        self.level_start = tree.level_start

    # -- frame ---------------------------------------------------------------------------
    def begin_rows(self, row0, n_rows, width, arena_rays=0):
        none = C.cast(None, _F64P)
        _check(self.lib.rt_begin_frame(self._h, int(row0), int(n_rows), none, none, 0, int(arena_rays)))
        self.npix = int(n_rows) * int(width)

    def begin_pixels(self, X, Y, arena_rays=0):
        X, Y = _f64(X).reshape(-1), _f64(Y).reshape(-1)
        if X.shape != Y.shape:
            raise ValueError("X and Y must have the same number of elements")
        _check(self.lib.rt_begin_frame(self._h, 0, 0, _ptr(X, C.c_double), _ptr(Y, C.c_double), X.size, int(arena_rays)))
        self.npix = X.size

    def trace_geometry(self):
        _check(self.lib.rt_trace_geometry(self._h))

    def trace_level(self, level):
        _check(self.lib.rt_trace_level(self._h, int(level)))

    def level_batches(self, level):
        n = int(self.level_start[level + 1] - self.level_start[level])
        out = np.zeros(n, dtype=np.int64)
        _check(self.lib.rt_get_level_batches(self._h, int(level), _ptr(out, C.c_int64)))
        return out

    def set_global_batches(self, level, n_global):
        a = np.ascontiguousarray(n_global, dtype=np.int64)
        _check(self.lib.rt_set_global_batches(self._h, int(level), _ptr(a, C.c_int64)))

    def set_batch_buffer(self, device_ptr, slot=0, n_slots=1):
        """Keep whole-image batch sizes in caller-owned device memory (int32[n_slots * n_nodes],
        level by level and frame by frame inside a level; this context is frame ``slot``);
        None: internal."""
        if device_ptr and n_slots > 1:
            _check(self.lib.rt_set_batch_buffer_shared(self._h, _P(device_ptr), int(slot), int(n_slots)))
        else:
            _check(self.lib.rt_set_batch_buffer(self._h, _P(device_ptr) if device_ptr else None))

    def set_batch_guess(self, device_ptr):
        """Sizes the small-batch rule goes by instead of the batch buffer's (same layout, read only;
        see rt_set_batch_guess); None: off."""
        _check(self.lib.rt_set_batch_guess(self._h, _P(device_ptr) if device_ptr else None))

    def node_counts(self):
        K = self.n_nodes
        batch, hits = np.zeros(K, dtype=np.int64), np.zeros(K, dtype=np.int64)
        beam = np.zeros((K, RT_MAX_LIGHTS), dtype=np.int64)
        _check(self.lib.rt_get_node_counts(self._h, _ptr(batch, C.c_int64), _ptr(hits, C.c_int64), _ptr(beam, C.c_int64)))
        return batch, hits, beam

    def set_sample_layout(self, offsets, total):
        off = np.full((self.n_nodes, RT_MAX_LIGHTS), -1, dtype=np.int64)
        off[:, :offsets.shape[1]] = offsets
        _check(self.lib.rt_set_sample_layout(self._h, _ptr(off, C.c_int64), int(total)))

    def upload_samples(self, dst_offset, u):
        u = _f64(u).reshape(-1)
        _check(self.lib.rt_upload_samples(self._h, int(dst_offset), _ptr(u, C.c_double), u.size))

    def upload_samples_noalloc(self, dst_offset, u):
        """upload_samples without any device allocation; at most NOALLOC_MAX_SAMPLES values per call."""
        u = _f64(u).reshape(-1)
        _check(self.lib.rt_upload_samples_noalloc(self._h, int(dst_offset), _ptr(u, C.c_double), u.size))

    def set_sample_layout_global(self, offsets, columns, first_column, total):
        """Where this context's columns sit in the whole image's stream (see the header)."""
        def full(a, fill):
            out = np.full((self.n_nodes, RT_MAX_LIGHTS), fill, dtype=np.int64)
            out[:, :a.shape[1]] = a
            return out
        o, c, f = full(np.asarray(offsets), -1), full(np.asarray(columns), 0), full(np.asarray(first_column), 0)
        _check(self.lib.rt_set_sample_layout_global(self._h, _ptr(o, C.c_int64), _ptr(c, C.c_int64), _ptr(f, C.c_int64), int(total)))

    def generate_samples(self, key, pos):
        """Generate the whole sample stream on the device from a legacy MT19937 state
        (``numpy.random.get_state()[1:3]``); returns the state after the draws."""
        key = np.ascontiguousarray(key, dtype=np.uint32)
        if key.shape != (624,):
            raise ValueError("MT19937 key must have 624 words")
        key_out = np.empty(624, dtype=np.uint32)
        pos_out = C.c_int32(0)
        _check(self.lib.rt_generate_samples(self._h, key.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos),
                                            key_out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos_out)))
        return key_out, int(pos_out.value)

    def set_precision(self, mode):
        """"f64", "mixed" or "mixed_checked" (see rt_set_precision in the header)."""
        _check(self.lib.rt_set_precision(self._h, {"f64": 0, "mixed": 1, "mixed_checked": 2}[mode]))

    def shade(self):
        _check(self.lib.rt_shade(self._h))

    def set_output(self, device_ptr):
        """Write the frame into caller-owned device memory (None: the context's own buffer)."""
        _check(self.lib.rt_set_output(self._h, _P(device_ptr) if device_ptr else None))

    def _host_array(self, shape, dtype):
        """NumPy array in page-locked memory of the library (freed when the array is collected):
        the device-to-host copy into it runs at the link's rate."""
        import weakref
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = _P()
        _check(self.lib.rt_alloc_host(n, C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        weakref.finalize(buf, self.lib.rt_free_host, _P(p.value))
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def alloc_host(self, shape, dtype=np.float64):
        """Page-locked NumPy array (rt_alloc_host; freed when the array is collected) to pass as ``out=``
        of read_image / read_image_rgb8: the copy into it is one transfer at the link's rate.  Meant to
        be allocated once and reused; pinning and unpinning memory per frame is slow."""
        return self._host_array(shape, dtype)

    def read_image(self, out=None):
        """The frame as an (npix, 3) float64 array: an ordinary NumPy array, or ``out`` (see alloc_host)."""
        if out is None:
            out = np.empty((self.npix, 3), dtype=np.float64)
        if out.dtype != np.float64 or out.size != self.npix * 3 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of npix * 3 values")
        _check(self.lib.rt_read_image(self._h, _ptr(out, C.c_double)))
        return out.reshape(self.npix, 3)

    def quantise_rgb8(self, device_ptr=None):
        """8-bit image into caller-owned device memory (None: the context's buffer); no synchronisation."""
        _check(self.lib.rt_quantise_rgb8(self._h, _P(device_ptr) if device_ptr else None))

    def read_image_rgb8(self, out=None):
        if out is None:
            out = np.empty((self.npix, 3), dtype=np.uint8)
        if out.dtype != np.uint8 or out.size != self.npix * 3 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of npix * 3 values")
        _check(self.lib.rt_read_image_rgb8(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.reshape(self.npix, 3)

    def trim(self):
        """Give the work buffers of the sample calls back (rt_trim)."""
        _check(self.lib.rt_trim(self._h))

    def image_device_ptr(self):
        p = _P()
        _check(self.lib.rt_image_device_ptr(self._h, C.byref(p)))
        return p.value

    def synchronize(self):
        _check(self.lib.rt_synchronize(self._h))

    def counters(self):
        c = RtCounters()
        _check(self.lib.rt_get_counters(self._h, C.byref(c)))
        return {name: getattr(c, name) for name, _ in RtCounters._fields_ if name != "reserved"}

    def profile_enable(self, on=True):
        _check(self.lib.rt_profile_enable(self._h, 1 if on else 0))

    def profile(self, reset=False):
        p = RtProfile()
        _check(self.lib.rt_get_profile(self._h, C.byref(p), 1 if reset else 0))
        return {k: {"ms": p.ms[i], "launches": p.launches[i]} for i, k in enumerate(KERNEL_KINDS)}

    DEBUG_FUNCTIONS = {"sphere_roots": (0, 7, 2), "nearest": (1, 6, 2), "shadow": (2, 6, 3), "optics": (3, 8, 8),
                       "ease": (4, 4, 1), "screen": (5, 2, 6)}

    def debug_eval(self, what, rows):
        """Evaluate one of the kernels' device functions for the given input rows (see
        rt_debug_eval in the header); returns the output rows."""
        code, in_w, out_w = self.DEBUG_FUNCTIONS[what]
        rows = _f64(rows).reshape(-1, in_w)
        out = np.empty((rows.shape[0], out_w), dtype=np.float64)
        _check(self.lib.rt_debug_eval(self._h, code, _ptr(rows, C.c_double), _ptr(out, C.c_double), rows.shape[0]))
        return out

    def render(self, height, width):
        out = np.empty((int(height) * int(width), 3), dtype=np.float64)
        _check(self.lib.rt_render(self._h, _ptr(out, C.c_double)))
        self.npix = out.shape[0]
        return out
