"""Drop-in entry points of the renderer: ``raytrace(scene)`` and
``get_pixel_color(X, Y, scene)`` with the reference's meaning (reference
raytrace.py:273-279 and :6-16), executed on an MI355X through the C ABI.

The host side only flattens the scene, sequences the device phases and draws
the disk-light random numbers; every per-ray computation runs in HIP kernels.
There is no CPU fallback.

Phases of one frame (:class:`Renderer`):

1. ``trace()``    geometry of all bounce levels (nearest hits, child rays,
                  per-node batch / hit / in-beam counts);
2. ``draw_samples()``  only for scenes with disk lights: the uniforms the
                  reference would draw from NumPy's global stream, in its order
                  (post-order over tree nodes, lights in scene order, all angles
                  then all radii of a block - reference objects.py:132-133).  With
                  NumPy's legacy stream (MT19937) the generator itself runs on the GPU
                  from the stream's current state and the state is advanced as the
                  host draws would have advanced it; any other ``rng`` object is
                  asked for the numbers (``rng.random(n)``) and they are uploaded;
3. ``shade()``    lighting and bottom-up colour combination;
4. ``image()``    read back ``(n, 3)`` float64 colours in [0, 1].
"""
import numpy as np

from . import _native
from .constants import RenderParams
from .flatten import FlatScene, flatten_scene
from .tree import build_tree, sample_offsets

DEFAULT_RAYS_PER_PIXEL = 24
SAMPLE_CHUNK = 1 << 24  # uniforms drawn and uploaded per step (128 MiB)


def legacy_mt19937_state(rng):
    """``rng.get_state()`` if ``rng`` is NumPy's legacy stream on MT19937 (the ``numpy.random``
    module or a ``RandomState``), else None."""
    get = getattr(rng, "get_state", None)
    if get is None:
        return None
    try:
        state = get()
    except TypeError:
        return None
    if isinstance(state, tuple) and len(state) >= 3 and state[0] == "MT19937":
        return state
    return None


class Renderer:
    """A scene bound to one GPU; renders whole frames, row blocks or pixel lists."""

    def __init__(self, scene, params=None, device=0, stream=None, rng=None, precision="mixed", sampler="auto"):
        self.flat = scene if isinstance(scene, FlatScene) else flatten_scene(scene)
        if params is None:
            params = RenderParams(width=int(self.flat.pixels_x), height=int(self.flat.pixels_y))
        self.params = params
        self.tree = build_tree(params.max_reflection_depth, params.max_refraction_depth)
        self.rng = np.random if rng is None else rng
        if sampler not in ("auto", "device", "host"):
            raise ValueError("sampler must be 'auto', 'device' or 'host'")
        self.sampler = sampler
        self.ctx = _native.Context(device, stream)
        self.ctx.set_scene(self.flat, params, self.tree)
        self.ctx.set_precision(precision)
        self._begin = None
        self.arena_rays = 0
        self.samples_total = 0
        self.node_batch = self.node_hits = self.node_beam = None

    def close(self):
        self.ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ frame set-up
    def begin_rows(self, row0=0, n_rows=None, arena_rays=0):
        n_rows = self.params.height - row0 if n_rows is None else n_rows
        self._begin = ("rows", row0, n_rows)
        self._arm(arena_rays or self.arena_rays or DEFAULT_RAYS_PER_PIXEL * n_rows * self.params.width)

    def begin_pixels(self, X, Y, arena_rays=0):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1)
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(-1)
        self._begin = ("pixels", X, Y)
        self._arm(arena_rays or self.arena_rays or DEFAULT_RAYS_PER_PIXEL * X.size)

    def _arm(self, arena_rays):
        self.arena_rays = int(arena_rays)
        if self._begin[0] == "rows":
            self.ctx.begin_rows(self._begin[1], self._begin[2], self.params.width, self.arena_rays)
        else:
            self.ctx.begin_pixels(self._begin[1], self._begin[2], self.arena_rays)

    # ------------------------------------------------------------------ phases
    def trace(self, check=True):
        """Geometry phase.  With ``check`` the node counts are fetched (one
        synchronisation) and an overflowing arena is doubled and the phase rerun."""
        while True:
            self.ctx.trace_geometry()
            if not check:
                return
            try:
                self.node_batch, self.node_hits, self.node_beam = self.ctx.node_counts()
                return
            except _native.RtOverflow:
                self._arm(self.arena_rays * 2)

    def draw_samples(self):
        """Draw the reference's disk-light uniforms from ``self.rng`` and upload them."""
        if not self.flat.has_sampled_lights:
            return 0
        if self.node_hits is None:
            raise RuntimeError("trace() must run (with check=True) before draw_samples()")
        offsets, _, total = sample_offsets(self.tree, self.flat.lights, self.node_hits, self.node_beam)
        self.ctx.set_sample_layout(offsets, total)
        self.samples_total = total
        state = legacy_mt19937_state(self.rng) if self.sampler != "host" else None
        if state is not None:
            # the generator behind the stream is MT19937 (numpy.random / RandomState): run it on
            # the GPU and leave the stream where the host draws would have left it
            key, pos = self.ctx.generate_samples(state[1], state[2])
            self.rng.set_state((state[0], key, pos) + tuple(state[3:]))
            return total
        if self.sampler == "device":
            raise ValueError("sampler='device' needs numpy.random or a numpy.random.RandomState on MT19937")
        for pos in range(0, total, SAMPLE_CHUNK):
            self.ctx.upload_samples(pos, self.rng.random(min(SAMPLE_CHUNK, total - pos)))
        return total

    def shade(self):
        self.ctx.shade()

    def image(self):
        return self.ctx.read_image()

    def image_rgb8(self):
        """The frame as ``plt.imsave`` would store it: ``(n, 3)`` uint8, quantised on the GPU."""
        return self.ctx.read_image_rgb8()

    def render_pixels(self):
        """All phases for the pixels chosen with begin_rows / begin_pixels."""
        if self.flat.has_sampled_lights:
            self.trace()            # the host lays the sample stream out by the hit counts
            self.draw_samples()
            self.shade()
            return self.image()
        # nothing on the host depends on the counts: the whole frame is queued and the host waits
        # once, for the image; an overflowing arena (every kernel after it does nothing) shows then
        while True:
            self.trace(check=False)
            self.shade()
            try:
                img = self.image()           # (reports the overflow too)
                self.node_batch, self.node_hits, self.node_beam = self.ctx.node_counts()
                return img
            except _native.RtOverflow:
                self._arm(self.arena_rays * 2)

    # ------------------------------------------------------------------ statistics
    def counters(self):
        return self.ctx.counters()


def raytrace(scene, *, params=None, device=0, devices=None, precision="mixed", sampler="auto"):
    """Render ``scene``; returns a ``(height, width, 3)`` float64 image in [0, 1].

    ``devices``: a list of GPU indices renders the frame split into row blocks on those GPUs, one
    process per GPU, started (and joined) by this call - see :mod:`raytracing_amd.launcher`; the
    image comes back to the caller and NumPy's global stream is left as after a single-GPU call.

    Same meaning as the reference's ``raytrace(scene)`` (reference
    raytrace.py:273-279).  The image size and depth limits come from ``params``
    (default: the scene's screen size with the current ``constants`` limits).
    Disk-light samples are taken from NumPy's global stream exactly as the
    reference takes them, so ``numpy.random.seed(k)`` before the call selects
    the same soft shadows.

    ``precision``: ``"mixed"`` (default) evaluates disk-light samples in float32
    with float64 re-evaluation of every shadow decision it cannot certify;
    ``"f64"`` evaluates them in float64 like the reference.  Mixed precision also keeps the
    colours that travel between the levels of the recursion as float32; every hit / miss
    decision, count and random-stream address is float64 in both modes."""
    if devices is not None and len(devices) > 1:
        from .launcher import raytrace_on_devices
        return raytrace_on_devices(scene, devices, params=params, precision=precision)
    if devices is not None and len(devices) == 1:
        device = devices[0]
    with Renderer(scene, params, device, precision=precision, sampler=sampler) as r:
        r.begin_rows(0, r.params.height)
        img = r.render_pixels()
        return img.reshape(r.params.height, r.params.width, 3)


def raytrace_rgb8(scene, *, params=None, device=0, precision="mixed"):
    """``raytrace(scene)`` followed by the quantisation of the reference's output step
    (reference main.py:32): ``(height, width, 3)`` uint8, exactly the bytes
    ``plt.imsave`` writes for the float image.  The conversion runs on the GPU, so only
    3 bytes per pixel cross PCIe."""
    with Renderer(scene, params, device, precision=precision) as r:
        r.begin_rows(0, r.params.height)
        r.trace()
        r.draw_samples()
        r.shade()
        return r.image_rgb8().reshape(r.params.height, r.params.width, 3)


def get_pixel_color(X, Y, scene, *, params=None, device=0, precision="mixed"):
    """Colours of the pixels with (possibly fractional) indices ``X``, ``Y``:
    ``(n, 3)`` float64, like the reference's ``get_pixel_color`` (reference
    raytrace.py:6-16).  As in the reference, the small-batch rule applies to the
    batches formed by the pixels of this call."""
    X = np.asarray(X, dtype=float)
    n = X.size
    if n == 0:
        return np.zeros((0, 3))
    with Renderer(scene, params, device, precision=precision) as r:
        r.begin_pixels(X, Y)
        return r.render_pixels()
