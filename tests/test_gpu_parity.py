"""Parity of the HIP path (through the C ABI) with the reference's golden
images and with the CPU oracle.  Needs an MI355X: run with -m gpu.

Tolerances: geometry is float64 in the reference's own rounding order, so ray /
hit / in-beam counts must match exactly and images of the float64 mode agree
with the reference to rounding noise (asserted: 1e-8 per channel).  The default
mixed mode evaluates disk-light samples in float32 with certified shadow
decisions and keeps the colours that travel between the levels of the recursion
(and the reflected fractions and in-sphere path lengths that weight them) as
float32; its images are held to 2e-5 per channel (the stated requirement is
1e-4)."""
import os

import numpy as np
import pytest

from oracle import numpy_oracle as O
from raytracing_amd import _native
from raytracing_amd.constants import RenderParams
from raytracing_amd.raytrace import Renderer, get_pixel_color, raytrace
from raytracing_amd.Scenes import synthetic
from raytracing_amd.Scenes.scenes import build_scene

pytestmark = pytest.mark.gpu

TOL = 1e-8
TOL_MIXED = 2e-5
PRECISIONS = [("f64", TOL), ("mixed", TOL_MIXED)]
EXAMPLES = ["example_1", "example_2", "example_3", "example_4"]


def _load(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path)


def _paths(tree):
    sym = "tir"
    out = [""] * tree.n_nodes
    for k in range(1, tree.n_nodes):
        out[k] = out[tree.parent[k]] + sym[tree.branch[k]]
    return out


def _render(scene, params=None, seed=0, precision="f64"):
    """Full frame on the GPU plus its node tables."""
    np.random.seed(seed)
    with Renderer(scene, params, precision=precision) as r:
        r.begin_rows(0, r.params.height)
        img = r.render_pixels().reshape(r.params.height, r.params.width, 3)
        return img, r.tree, r.node_batch, r.node_hits, r.node_beam, r.counters()


def _check_counts(tree, batch, hits, beam, g, key, small_num=3):
    by_path = {p: i for i, p in enumerate(g[key + "_path"])}
    rays, ghits, gbeam = g[key + "_rays"], g[key + "_hits"], g[key + "_beam"]
    for k, p in enumerate(_paths(tree)):
        i = by_path.get(p)
        want_n = int(rays[i]) if i is not None else 0
        want_h = int(ghits[i]) if i is not None else 0
        assert batch[k] == want_n, (p, batch[k], want_n)
        assert hits[k] == want_h, (p, hits[k], want_h)
        if i is not None and gbeam.shape[1] >= 1 and gbeam[i].any():
            assert beam[k, 0] == gbeam[i, 0], (p, beam[k, 0], gbeam[i, 0])


@pytest.mark.parametrize("precision,tol", PRECISIONS)
@pytest.mark.parametrize("name", EXAMPLES)
def test_examples_160x90_against_reference_goldens(golden_dir, name, precision, tol):
    g = _load(golden_dir, "images_160x90.npz")
    img, tree, batch, hits, beam, cnt = _render(build_scene(name, 160, 90), precision=precision)
    assert np.abs(img - g[name + "_image"]).max() <= tol
    if precision == "f64":
        assert cnt["fallback_samples"] == 0
    _check_counts(tree, batch, hits, beam, g, name)
    r = g[name + "_rays"]
    assert cnt["rays"] == int(r[r >= 3].sum())
    assert cnt["hits"] == int(g[name + "_hits"].sum())


@pytest.mark.parametrize("depths", [(0, 0), (1, 0), (0, 2), (2, 2), (1, 3)])
@pytest.mark.parametrize("name", ["example_2", "example_3"])
def test_depth_limits(golden_dir, name, depths):
    g = _load(golden_dir, "depths_96x54.npz")
    a, b = depths
    key = f"{name}_{a}_{b}"
    params = RenderParams(width=96, height=54, max_reflection_depth=a, max_refraction_depth=b)
    img, tree, batch, hits, beam, _ = _render(build_scene(name, 96, 54), params)
    assert np.abs(img - g[key + "_image"]).max() <= TOL
    _check_counts(tree, batch, hits, beam, g, key)


@pytest.mark.parametrize("precision,tol", PRECISIONS)
@pytest.mark.parametrize("mode", ["none", "linear", "quadratic", "cubic", "exponential"])
def test_spotlight_easing_modes(golden_dir, mode, precision, tol):
    g = _load(golden_dir, "spot_128x72.npz")
    img, tree, batch, hits, beam, _ = _render(synthetic.spotlight_scene(mode, 64, 128, 72), precision=precision)
    assert np.abs(img - g[mode + "_image"]).max() <= tol
    _check_counts(tree, batch, hits, beam, g, mode)


def test_many_spheres_257(golden_dir):
    g = _load(golden_dir, "many_257_64x36.npz")
    img, tree, batch, hits, beam, _ = _render(synthetic.many_spheres_scene(257, 64, 36))
    assert np.abs(img - g["image"]).max() <= TOL
    _check_counts(tree, batch, hits, beam, g, "trace")


def test_many_spheres_4096_through_the_hierarchy(golden_dir):
    g = _load(golden_dir, "many_4096_64x36.npz")
    img, tree, batch, hits, beam, _ = _render(synthetic.many_spheres_scene(4096, 64, 36))
    assert np.abs(img - g["image"]).max() <= TOL
    _check_counts(tree, batch, hits, beam, g, "trace")


def test_row_strips_through_get_pixel_color(golden_dir):
    """get_pixel_color on row strips reproduces the reference's strip renders,
    including the pixels that differ from the whole-frame render because the
    small-batch rule sees smaller batches."""
    g = _load(golden_dir, "strips_160x90.npz")
    for name in ("example_1", "example_3"):
        scene = build_scene(name, 160, 90)
        bounds = g[name + "_bounds"]
        rows = []
        for r0, r1 in zip(bounds[:-1], bounds[1:]):
            Y, X = np.indices((r1 - r0, 160), dtype=float)
            rows.append(get_pixel_color(X.reshape(-1, 1), (Y + r0).reshape(-1, 1), scene, precision="f64").reshape(r1 - r0, 160, 3))
        assert np.abs(np.concatenate(rows) - g[name + "_strips"]).max() <= TOL


@pytest.mark.parametrize("name,size", [("example_1", (333, 187)), ("example_2", (257, 129)),
                                        ("example_3", (500, 500)), ("example_4", (200, 113))])
def test_against_oracle_at_ragged_sizes(name, size):
    """Image sizes that are not multiples of the 64-ray tile."""
    w, h = size
    scene = build_scene(name, w, h)
    img, tree, batch, hits, beam, cnt = _render(scene, seed=3)
    want, stats = O.render(scene, seed=3)
    assert np.abs(img - want).max() <= TOL
    if name == "example_4":
        mixed = _render(scene, seed=3, precision="mixed")
        assert np.abs(mixed[0] - want).max() <= TOL_MIXED
        assert mixed[5]["shaded_samples"] == cnt["shaded_samples"]
        assert 0 <= mixed[5]["fallback_samples"] < 1e-3 * cnt["shaded_samples"]
    assert cnt["rays"] == stats.traced_rays()
    assert cnt["hits"] == stats.hits()


def test_tiny_frames_and_the_small_batch_rule():
    """1 and 2 pixel frames are below the small-batch limit: background only."""
    scene = build_scene("example_1", 160, 90)
    bg = RenderParams().background_color
    for n in (1, 2):
        X = np.full((n, 1), 80.0)
        Y = np.full((n, 1), 60.0)
        assert np.array_equal(get_pixel_color(X, Y, scene), np.tile(bg, (n, 1)))
    X = np.array([80.0, 80.0, 81.0, 10.0])[:, None]
    Y = np.array([60.0, 60.5, 60.0, 2.0])[:, None]
    got = get_pixel_color(X, Y, scene, precision="f64")
    flat_params = RenderParams(width=160, height=90)
    want = O.Oracle(__import__("raytracing_amd.flatten", fromlist=["x"]).flatten_scene(scene), flat_params).pixel_colors(X, Y)
    assert np.abs(got - want).max() <= TOL
    assert get_pixel_color(np.zeros((0, 1)), np.zeros((0, 1)), scene).shape == (0, 3)


def test_arena_overflow_is_detected_and_retried():
    scene = build_scene("example_3", 160, 90)
    with Renderer(scene) as r:
        r.begin_rows(0, 90, arena_rays=160 * 90 + 512)     # far too small: about 21 rays per pixel are needed
        r.ctx.trace_geometry()
        with pytest.raises(_native.RtOverflow):
            r.ctx.node_counts()
        r.ctx.set_precision("f64")
        img = r.render_pixels()                             # retries with a doubled arena until it fits
        assert r.arena_rays > 160 * 90 + 512
    want, _ = O.render(scene, seed=0)
    assert np.abs(img.reshape(90, 160, 3) - want).max() <= TOL


def test_reference_objects_are_accepted(golden_dir):
    """raytrace() takes the scene by attribute, so any object with the
    reference's attribute names works; exercised here with plain namespaces."""
    from types import SimpleNamespace as NS
    src = build_scene("example_2", 96, 54)
    objs = [NS(position=s.position, radius=s.radius, material=NS(**vars(s.material))) for s in src.objects]
    lights = [NS(position=l.position, intensity=l.intensity, diffuse_color=l.diffuse_color,
                 specular_color=l.specular_color) for l in src.light_sources]
    dup = NS(objects=objs, light_sources=lights, ambient_light=src.ambient_light, camera=src.camera, screen=src.screen)
    a = raytrace(dup, precision="f64")
    b = raytrace(src, precision="f64")
    assert np.array_equal(a, b)
    assert a.shape == (54, 96, 3)
    g = _load(golden_dir, "images_160x90.npz")                  # and what both give is the reference's image
    dup.camera = build_scene("example_2", 160, 90).camera
    dup.screen = dup.camera.screen
    assert np.abs(raytrace(dup, precision="f64") - g["example_2_image"]).max() <= TOL


def test_call_order_errors():
    scene = build_scene("example_4", 32, 18)
    with Renderer(scene) as r:
        with pytest.raises(_native.RtError):
            r.ctx.trace_geometry()          # no frame begun
        r.begin_rows(0, 18)
        r.ctx.trace_geometry()
        with pytest.raises(_native.RtError):
            r.ctx.shade()                   # disk light samples missing


def test_batch_sizes_in_caller_owned_device_memory(golden_dir):
    """rt_set_batch_buffer: the whole-image batch sizes live in a torch tensor, the
    levels are traced one by one and nothing is copied through the host (on several
    GPUs an all-reduce of the level's slice sits between two levels)."""
    import torch
    g = _load(golden_dir, "images_160x90.npz")
    scene = build_scene("example_3", 160, 90)
    with Renderer(scene, precision="f64") as r:
        buf = torch.zeros(r.tree.n_nodes, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()        # the fill ran on torch's stream, the renderer has its own
        r.ctx.set_batch_buffer(buf.data_ptr())
        r.begin_rows(0, 90)
        for level in range(r.tree.n_levels):
            r.ctx.trace_level(level)
        batch, hits, beam = r.ctx.node_counts()
        r.shade()
        img = r.image().reshape(90, 160, 3)
        r.ctx.synchronize()
        assert np.array_equal(buf.cpu().numpy(), batch)          # one GPU: global == local
        # shrinking a batch below the limit in the buffer kills that node on the next frame
        assert np.abs(img - g["example_3_image"]).max() <= TOL
        r.ctx.set_batch_buffer(None)
        r.ctx.trace_geometry()
        r.shade()
        assert np.array_equal(r.image().reshape(90, 160, 3), img)


# ---------------------------------------------------------------------------------------------
# Branches the example scenes hardly reach, on fixtures made by running the reference.
@pytest.mark.parametrize("precision,tol", PRECISIONS)
def test_total_internal_reflection_scene(golden_dir, precision, tol):
    """reference raytrace.py:122-127: ~4000 rays in calls reached through the "i" branch (camera
    inside a glass sphere, a sphere of index 0.8, overlapping glass spheres)."""
    g = _load(golden_dir, "tir_160x90.npz")
    tir = [int(n) for p, n in zip(g["trace_path"], g["trace_rays"]) if str(p).endswith("i") and n >= 3]
    assert sum(tir) >= 1000
    img, tree, batch, hits, beam, cnt = _render(synthetic.tir_scene(160, 90), precision=precision)
    assert np.abs(img - g["image"]).max() <= tol
    _check_counts(tree, batch, hits, beam, g, "trace")
    paths = _paths(tree)
    assert sum(int(batch[k]) for k in range(tree.n_nodes) if paths[k].endswith("i") and batch[k] >= 3) == sum(tir)


@pytest.mark.parametrize("precision,tol", PRECISIONS)
def test_attenuation_above_one_and_the_per_call_clip(golden_dir, precision, tol):
    """reference raytrace.py:117-120: a transmitted ray that misses the sphere it entered gets the
    path length -1, so its colour is multiplied by exp(+k) > 1; reference raytrace.py:170 clips the
    result of every call, also below the top level (the fixture records values up to 11 entering
    that clip at depth).  A forward accumulation of path weights would differ here; the bottom-up
    colour pass reproduces the reference.  In mixed precision the colours between the calls are
    float32: every one of them has passed the clip, so they lie in [0, 1] whatever entered it."""
    g = _load(golden_dir, "clip_160x90.npz")
    deep = [m for (a, b), m in zip(g["preclip_depth_used"], g["preclip_max"]) if a + b > 0]
    assert max(deep) > 2.0
    params = RenderParams(width=160, height=90, epsilon=float(g["epsilon"]))
    img, tree, batch, hits, beam, cnt = _render(synthetic.clip_scene(160, 90), params, precision=precision)
    assert np.abs(img - g["image"]).max() <= tol
    _check_counts(tree, batch, hits, beam, g, "trace")


def test_disk_light_colour_attributes_are_ignored(golden_dir):
    """reference objects.py:140-144"""
    g = _load(golden_dir, "disk_colour_48x27.npz")
    scene = build_scene("example_4", 48, 27)
    scene.light_sources[0].diffuse_color = np.array([1.0, 0.0, 0.0])
    scene.light_sources[0].specular_color = np.array([0.0, 0.0, 1.0])
    img = _render(scene)[0]
    assert np.abs(img - g["coloured"]).max() <= TOL


# ---------------------------------------------------------------------------------------------
# The benchmark's own frame sizes against the reference: make_golden.py rendered example_1-4 at
# 1920x1080 and example_1-3 at 3840x2160 with the reference and kept a grid of pixels, the per-call
# traces and the image sums.  (example_4 at 3840x2160 needs ~78 GB in the reference: parity of
# that frame is pinned at 1920x1080 and below, at full size only through the properties further down.)
def _check_full_size(g, name, W, H, precision, tol):
    sy, sx = (H // 36, W // 60)                 # the fixtures keep a 36 x 60 grid of pixels
    if "steps" in g:
        assert (sy, sx) == tuple(int(v) for v in g["steps"])
    np.random.seed(0)
    with Renderer(build_scene(name, W, H), precision=precision) as r:
        r.begin_rows(0, H)
        img = r.render_pixels().reshape(H, W, 3)
        cnt = r.counters()
        assert np.abs(img[::sy, ::sx] - g[name + "_pixels"]).max() <= tol
        assert abs(img.sum() - float(g[name + "_sum"])) <= tol * W * H * 3 * 1e-2
        _check_counts(r.tree, r.node_batch, r.node_hits, r.node_beam, g, name)
        rays = g[name + "_rays"]
        assert cnt["rays"] == int(rays[rays >= 3].sum())
        assert cnt["hits"] == int(g[name + "_hits"].sum())
        assert r.samples_total == int(g[name + "_uniforms"])


@pytest.mark.parametrize("name,precision,tol", [("example_1", "f64", TOL), ("example_2", "f64", TOL), ("example_3", "f64", TOL),
                                                ("example_4", "f64", TOL), ("example_4", "mixed", TOL_MIXED)])
def test_full_hd_frames_against_the_reference(golden_dir, name, precision, tol):
    g = _load(golden_dir, "full_1080.npz")
    if name + "_pixels" not in g:
        pytest.skip(f"{name} at 1920x1080 not generated")
    _check_full_size(g, name, 1920, 1080, precision, tol)


@pytest.mark.parametrize("name", ["example_1", "example_2", "example_3"])
def test_4k_frames_against_the_reference(golden_dir, name):
    g = _load(golden_dir, "full_4k.npz")
    if name + "_pixels" not in g:
        pytest.skip(f"{name} at 3840x2160 not generated")
    _check_full_size(g, name, 3840, 2160, "f64", TOL)


# Ray / hit / shadow-ray / random-number counts of the REFERENCE at 1920x1080 (seed 0),
# recorded when the reference itself was run at that size.
FULL_HD_REFERENCE_COUNTS = {
    "example_1": dict(rays=37_336_983, shadow=20_571_993),
    "example_2": dict(rays=24_405_826, shadow=13_154_208),
    "example_3": dict(rays=43_182_942, shadow=25_755_003),
    "example_4": dict(rays=44_323_426, shadow=780_860_790, uniforms=1_561_721_580),
}


@pytest.mark.parametrize("name", EXAMPLES)
def test_full_hd_counts_equal_the_reference_run(name):
    """At 1920x1080 the reference is too slow to re-run in a test, but its integer
    counters were recorded: every nearest-hit decision of 24-44 million rays has to
    agree for these to match.  (The reference casts one shadow ray per hit and light
    sample, also for hits whose colour is later multiplied by zero.)"""
    want = FULL_HD_REFERENCE_COUNTS[name]
    scene = build_scene(name, 1920, 1080)
    np.random.seed(0)
    with Renderer(scene) as r:
        r.begin_rows(0, 1080)
        r.trace()
        n_uniforms = r.draw_samples()
        cnt = r.counters()
        samples_per_hit = sum(l.n_points for l in r.flat.lights)
    assert cnt["rays"] == want["rays"]
    assert cnt["hits"] * samples_per_hit == want["shadow"]
    if "uniforms" in want:
        assert n_uniforms == want["uniforms"]


# ---------------------------------------------------------------------------------------------
# The reference's output step: 8-bit images, quantised on the GPU, against what the reference
# wrote through plt.imsave (goldens made by tests/golden/make_golden.py, generator "rgb8").
@pytest.mark.gpu
@pytest.mark.parametrize("precision,max_levels_off", [("f64", 0.0), ("mixed", 5e-3)])
@pytest.mark.parametrize("name", ["example_1", "example_2", "example_3"])
def test_rgb8_equals_reference_output(golden_dir, name, precision, max_levels_off):
    """Float64: the reference's bytes.  Mixed precision (colours between the recursion's calls kept
    as float32, within 2e-5 of the reference): a channel may land on the other side of a
    quantisation step, never further than one level."""
    from raytracing_amd.raytrace import raytrace_rgb8
    gold = _load(golden_dir, "rgb8.npz")[name]
    h, w, _ = gold.shape
    got = raytrace_rgb8(build_scene(name, w, h), precision=precision)
    assert got.dtype == np.uint8 and got.shape == gold.shape
    diff = np.abs(got.astype(int) - gold.astype(int))
    assert diff.max() <= (1 if max_levels_off else 0)
    assert (diff > 0).mean() <= max_levels_off


@pytest.mark.gpu
@pytest.mark.parametrize("precision,max_levels_off", [("f64", 0.0), ("mixed", 5e-3)])
def test_rgb8_disk_light_scene(golden_dir, precision, max_levels_off):
    """example_4 at the reference's default 500x500, seed 0.  Float64 samples: the same bytes;
    float32 samples (colours within 2e-5): a channel may land on the other side of a
    quantisation step, never further than one level."""
    from raytracing_amd.raytrace import raytrace_rgb8
    gold = _load(golden_dir, "rgb8.npz")["example_4"].astype(int)
    np.random.seed(0)
    got = raytrace_rgb8(build_scene("example_4", 500, 500), precision=precision).astype(int)
    diff = np.abs(got - gold)
    assert diff.max() <= (1 if max_levels_off else 0)
    assert (diff > 0).mean() <= max_levels_off


@pytest.mark.gpu
def test_main_writes_png(tmp_path):
    from raytracing_amd import main as rt_main
    from raytracing_amd.imageio import decode_png
    from raytracing_amd.raytrace import raytrace
    out = tmp_path / "result.png"
    rt_main.main(["example_2", "--width", "96", "--height", "54", "--out", str(out)])
    px = decode_png(out.read_bytes())
    assert px.shape == (54, 96, 4) and (px[..., 3] == 255).all()
    img = raytrace(build_scene("example_2", 96, 54))
    assert (px[..., :3] == (img * 255).astype(np.uint8)).all()


# ---------------------------------------------------------------------------------------------
# NumPy's legacy generator (MT19937, random_sample) on the device.
def _render_with(sampler, scene, state):
    np.random.set_state(state)
    with Renderer(scene, precision="f64", sampler=sampler) as r:
        r.begin_rows(0, r.params.height)
        img = r.render_pixels()
        return img, r.samples_total, np.random.get_state()


@pytest.mark.gpu
@pytest.mark.parametrize("pos", [624, 0, 1, 623, 397])
def test_device_mt19937_equals_numpy_stream(pos):
    """Same image as with host-drawn samples (bit for bit: the packed samples are equal) and the
    global stream is left exactly where numpy.random.random(total) leaves it."""
    scene = build_scene("example_4", 96, 54)
    np.random.seed(123)
    st = np.random.get_state()
    st = (st[0], st[1], pos) + tuple(st[3:])
    host_img, total, host_state = _render_with("host", scene, st)
    dev_img, total_d, dev_state = _render_with("device", scene, st)
    assert total == total_d and total > 100000
    assert np.array_equal(host_img, dev_img)
    assert dev_state[2] == host_state[2] and np.array_equal(dev_state[1], host_state[1])
    # and the stream really continues identically
    np.random.set_state(dev_state)
    a = np.random.random(5)
    np.random.set_state(host_state)
    assert np.array_equal(a, np.random.random(5))


@pytest.mark.gpu
def test_device_mt19937_small_and_spot_streams():
    """Streams shorter than one generator step, and the in-beam column addressing."""
    for scene in (synthetic.spotlight_scene("cubic", 2, 24, 14), synthetic.spotlight_scene("linear", 16, 64, 36)):
        np.random.seed(7)
        st = np.random.get_state()
        host_img, total, host_state = _render_with("host", scene, st)
        dev_img, _, dev_state = _render_with("device", scene, st)
        assert np.array_equal(host_img, dev_img)
        assert dev_state[2] == host_state[2] and np.array_equal(dev_state[1], host_state[1])


@pytest.mark.gpu
def test_device_sampler_needs_a_legacy_stream():
    scene = build_scene("example_4", 32, 18)
    with Renderer(scene, rng=np.random.default_rng(0), sampler="device") as r:
        r.begin_rows(0, 18)
        r.trace()
        with pytest.raises(ValueError):
            r.draw_samples()
    with Renderer(scene, rng=np.random.default_rng(0)) as r:      # auto: falls back to rng.random()
        r.begin_rows(0, 18)
        img = r.render_pixels()
        assert np.isfinite(img).all()
    rs = np.random.RandomState(5)
    with Renderer(scene, rng=rs, precision="f64") as r:           # a RandomState runs on the device
        r.begin_rows(0, 18)
        img = r.render_pixels()
        total = r.samples_total
    ref = np.random.RandomState(5)
    ref.random_sample(total)
    assert np.array_equal(rs.get_state()[1], ref.get_state()[1]) and rs.get_state()[2] == ref.get_state()[2]


@pytest.mark.gpu
@pytest.mark.parametrize("raw_words", [None, 1 << 17, 1 << 16])
def test_device_mt19937_streams_and_passes(monkeypatch, raw_words):
    """A stream long enough for two dozen jump-ahead streams; with a small raw-word budget
    they are generated in several passes (a value may straddle two passes)."""
    if raw_words is None:
        monkeypatch.delenv("RT_MT_RAW_WORDS", raising=False)
    else:
        monkeypatch.setenv("RT_MT_RAW_WORDS", str(raw_words))
    scene = build_scene("example_4", 160, 90)
    np.random.seed(2024)
    st = np.random.get_state()
    st = (st[0], st[1], 311) + tuple(st[3:])        # odd position: values straddle stream boundaries
    host_img, total, host_state = _render_with("host", scene, st)
    dev_img, _, dev_state = _render_with("device", scene, st)
    assert 2 * total > 20 * 65536
    assert np.array_equal(host_img, dev_img)
    assert dev_state[2] == host_state[2] and np.array_equal(dev_state[1], host_state[1])


@pytest.mark.gpu
@pytest.mark.parametrize("raw_words,by_ranges", [(1 << 20, False), (1 << 20, True), (1 << 18, False), (1 << 22, False)])
def test_device_mt19937_passes_in_mixed_precision(monkeypatch, raw_words, by_ranges):
    """A work buffer smaller than the stream, frame shaded in mixed precision: the passes are made of whole
    (node, light) blocks and the disk points come with them (or, RT_MT_RANGE_PASSES=1 and whenever a
    block is larger than the buffer - the 2^18 case -, of ranges of the stream with the points made
    afterwards).  Same frame and final stream state as with host-drawn samples."""
    from raytracing_amd import objects
    monkeypatch.setenv("RT_MT_POISON", "1")
    monkeypatch.setenv("RT_MT_RAW_WORDS", str(raw_words))
    if by_ranges:
        monkeypatch.setenv("RT_MT_RANGE_PASSES", "1")
    else:
        monkeypatch.delenv("RT_MT_RANGE_PASSES", raising=False)
    spot_scene = synthetic.spotlight_scene("linear", 16, 200, 112)
    spot_scene.light_sources = [objects.DiskSource(x=-3, y=1, z=6, radius=1.5, intensity=20, N=5)] + list(spot_scene.light_sources)
    for scene in (build_scene("example_4", 240, 135), spot_scene):
        np.random.seed(17)
        st = np.random.get_state()
        st = (st[0], st[1], 77) + tuple(st[3:])
        frames = {}
        for sampler in ("host", "device"):
            np.random.set_state(st)
            with Renderer(scene, sampler=sampler) as r:
                r.begin_rows(0, r.params.height)
                frames[sampler] = (r.render_pixels().copy(), np.random.get_state(), r.samples_total)
        assert 2 * frames["device"][2] > 3 * raw_words or raw_words >= 1 << 22     # several passes indeed
        assert np.array_equal(frames["host"][0], frames["device"][0])
        assert frames["host"][1][2] == frames["device"][1][2] and np.array_equal(frames["host"][1][1], frames["device"][1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("store_all,max_streams", [(False, None), (True, None), (False, 5), (False, 100000)])
def test_device_mt19937_store_map(monkeypatch, store_all, max_streams):
    """The generator leaves out the words of values no shaded hit reads (default) or stores them all;
    few long streams or many short ones: the frame (mixed precision, disk points made along the way)
    equals the one from host-drawn samples and the stream ends in the same place.  The spotlight's
    blocks are not screened (in-beam numbering) and lie between screened ones."""
    from raytracing_amd import objects
    monkeypatch.setenv("RT_MT_POISON", "1")     # words the generator leaves out hold a pattern, not an earlier frame's values
    for name, value in (("RT_MT_STORE_ALL", "1" if store_all else None), ("RT_MT_MAX_STREAMS", max_streams)):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(value))
    spot_scene = synthetic.spotlight_scene("linear", 16, 200, 112)
    spot_scene.light_sources = [objects.DiskSource(x=-3, y=1, z=6, radius=1.5, intensity=20, N=5)] + list(spot_scene.light_sources)
    crowd = _disk_scene(97, 41, extra_spheres=12, n_points=9)                 # many nodes with few hits: narrow blocks
    crowd_spot = _disk_scene(120, 67, extra_spheres=37, spot=True, n_points=5)
    for scene in (build_scene("example_4", 480, 270), spot_scene, crowd, crowd_spot):
        np.random.seed(99)
        st = np.random.get_state()
        st = (st[0], st[1], 5) + tuple(st[3:])
        images = {}
        for sampler in ("host", "device"):
            np.random.set_state(st)
            with Renderer(scene, sampler=sampler) as r:
                r.begin_rows(0, r.params.height)
                images[sampler] = (r.render_pixels().copy(), np.random.get_state())
        assert np.array_equal(images["host"][0], images["device"][0])
        assert images["host"][1][2] == images["device"][1][2] and np.array_equal(images["host"][1][1], images["device"][1][1])


# ---------------------------------------------------------------------------------------------
# Degenerate scenes: nothing to hit, nothing that lights, no ambient term.
@pytest.mark.gpu
@pytest.mark.parametrize("drop", ["spheres", "lights", "ambient"])
def test_degenerate_scenes(drop):
    from raytracing_amd.Scenes.Scene import Scene
    s = build_scene("example_2", 40, 30)
    scene = Scene(objects=[] if drop == "spheres" else s.objects,
                  light_sources=[] if drop == "lights" else s.light_sources,
                  ambient_light=None if drop == "ambient" else s.ambient_light,
                  camera=s.camera, screen=s.screen)
    want, _ = O.render(scene, seed=0)
    got = raytrace(scene, precision="f64")
    assert got.shape == (30, 40, 3)
    assert np.abs(got - want.reshape(30, 40, 3)).max() <= 1e-8
    if drop == "spheres":
        assert np.all(got == np.array([0.251, 0.624, 0.769]))      # reference constants.py: BACKGROUND_COLOR


# ---------------------------------------------------------------------------------------------
# The float32 screening of ray-sphere pairs (scenes with more than 64 spheres) must never drop a
# pair the float64 test would accept: scenes in the three table regimes, and the same scene far
# from the origin, where float32 coordinates are coarse and the error bound has to grow with them.
def _shifted(flat, shift):
    import copy
    f = copy.deepcopy(flat)
    shift = np.asarray(shift, dtype=float)
    f.sphere_geom = f.sphere_geom.copy()
    f.sphere_geom[:, :3] += shift
    f.camera_position = f.camera_position + shift
    f.screen_position = f.screen_position + shift
    for light in f.lights:
        light.position = light.position + shift
    return f


@pytest.mark.gpu
@pytest.mark.parametrize("n_spheres,shift", [(100, (0, 0, 0)), (300, (0, 0, 0)), (300, (1.0e5, -2.0e5, 3.0e4)),
                                             (300, (-3.0e6, 1.0e6, 2.0e6))])
def test_screening_keeps_every_hit(n_spheres, shift):
    from raytracing_amd.flatten import flatten_scene
    flat = _shifted(flatten_scene(synthetic.many_spheres_scene(n_spheres, 48, 27)), shift)
    want, stats = O.render(flat, seed=0)
    with Renderer(flat, precision="f64") as r:
        r.begin_rows(0, 27)
        got = r.render_pixels()
        c = r.counters()
    assert int(c["rays"]) == stats.traced_rays()
    assert int(c["hits"]) == stats.hits()
    # far from the origin every coordinate carries ~1e-11 of float64 rounding, which the specular
    # term amplifies: the images then agree to 1e-6 instead of 1e-8 (a dropped pair would change
    # the counts above and whole pixels)
    assert np.abs(got - want.reshape(-1, 3)).max() <= (1e-8 if not any(shift) else 1e-6)


@pytest.mark.gpu
def test_float32_samples_general_coefficients():
    """Lights whose diffuse and specular colours differ (or coefficients outside [0, 1]) take the
    float32 sample loop with per-channel sums; it must agree with the float64 evaluation too."""
    from raytracing_amd.flatten import flatten_scene
    flat = flatten_scene(build_scene("example_4", 96, 54))
    flat.lights[0].specular_color = np.array([0.9, 0.7, 1.0])
    flat.sphere_mat = flat.sphere_mat.copy()
    flat.sphere_mat[1, 6:9] = [1.4, 0.2, 0.9]          # a specular coefficient above one
    want, _ = O.render(flat, seed=3)
    imgs = {}
    for precision in ("f64", "mixed"):
        np.random.seed(3)
        with Renderer(flat, precision=precision) as r:
            r.begin_rows(0, 54)
            imgs[precision] = r.render_pixels()
    assert np.abs(imgs["f64"] - want.reshape(-1, 3)).max() <= 1e-8
    assert np.abs(imgs["mixed"] - imgs["f64"]).max() <= 2e-5


# ---------------------------------------------------------------------------------------------
# What is built over many spheres - the uniform grid with the point lights' occluder maps (default),
# the grid alone (RT_LIGHT_MAPS=0), the bounding-sphere hierarchy (RT_ACCEL=hier) - only chooses
# which spheres take the float64 test; with each of them and with none (RT_HIERARCHY=0: every sphere
# screened for every ray) the frame must come out bit for bit the same: images, ray and hit counts.
ACCELERATORS = {"grid": {}, "grid, no light maps": {"RT_LIGHT_MAPS": "0"}, "hierarchy": {"RT_ACCEL": "hier"},
                "none": {"RT_HIERARCHY": "0"}}


def _with_and_without_hierarchy(flat, width, height, which=("grid", "none")):
    """Frame and counts per accelerator; the keys "1" and "0" stand for the first and the last of ``which``."""
    out = {}
    for name in which:
        os.environ.update(ACCELERATORS[name])
        try:
            with Renderer(flat, RenderParams(width=width, height=height), precision="f64") as r:
                r.begin_rows(0, height)
                img = r.render_pixels()
                c = r.counters()
                out[name] = (img, int(c["rays"]), int(c["hits"]), int(c["shaded_hits"]))
        finally:
            for key in ACCELERATORS[name]:
                os.environ.pop(key, None)
    for name in which[1:-1]:
        assert out[name][1:] == out[which[-1]][1:], name
        assert np.array_equal(out[name][0], out[which[-1]][0]), name
    out["1"], out["0"] = out[which[0]], out[which[-1]]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n_spheres", [65, 200, 1000, 5000])   # 5000: more than 64 top entries
def test_hierarchy_matches_plain_screening(n_spheres):
    from raytracing_amd.flatten import flatten_scene
    w, h = (256, 144) if n_spheres <= 1000 else (128, 72)
    flat = flatten_scene(synthetic.many_spheres_scene(n_spheres, w, h))
    out = _with_and_without_hierarchy(flat, w, h, tuple(ACCELERATORS))
    assert out["1"][1:] == out["0"][1:]
    assert np.array_equal(out["1"][0], out["0"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("light", ["disk", "spot", "point + disk"])
def test_many_spheres_under_sampled_lights(light):
    """More than 64 spheres under a disk light or a spotlight: the shadow rays of the samples walk the
    grid (no occluder map: that is for point lights), in float64.  Against the oracle, and every
    accelerator against none."""
    from raytracing_amd import objects
    from raytracing_amd.flatten import flatten_scene
    w, h = 64, 36
    scene = synthetic.many_spheres_scene(90, w, h)
    point = scene.light_sources[0]
    if light == "spot":
        sampled = objects.DirectionalDiskSource(x=20, y=0, z=25, radius=2, intensity=900, angle=40, easing_mode="linear")
        sampled.n_points = 3
    else:
        sampled = objects.DiskSource(x=20, y=5, z=25, radius=3, intensity=900, N=3)
    scene.light_sources = [point, sampled] if light.startswith("point") else [sampled]
    flat = flatten_scene(scene)
    want, stats = O.render(flat, seed=3)
    frames = {}
    for name, env in ACCELERATORS.items():
        os.environ.update(env)
        try:
            np.random.seed(3)
            with Renderer(flat, RenderParams(width=w, height=h), precision="f64") as r:
                r.begin_rows(0, h)
                frames[name] = r.render_pixels()
                c = r.counters()
            assert int(c["rays"]) == stats.traced_rays() and int(c["hits"]) == stats.hits(), name
        finally:
            for key in env:
                os.environ.pop(key, None)
    assert np.abs(frames["grid"] - want.reshape(-1, 3)).max() <= 1e-8
    for name in frames:
        assert np.array_equal(frames[name], frames["none"]), name


@pytest.mark.gpu
def test_hierarchy_far_from_origin_and_list_overflow():
    """The same far from the origin (coarse float32 coordinates), and with 40 glass shells around one
    point in front of the camera: rays towards them get more candidates at once than a lane's list
    holds and take the plain screening of every sphere."""
    from raytracing_amd.flatten import flatten_scene
    flat = flatten_scene(synthetic.many_spheres_scene(300, 96, 54))
    row = np.tile(flat.sphere_geom[5], (40, 1))
    cam, view = flat.camera_position, flat.screen_position - flat.camera_position
    view = view / np.linalg.norm(view)
    for i in range(40):
        row[i, :3] = cam + 8.0 * view
        row[i, 3] = 0.3 + 0.002 * i
    glass = np.tile(flat.sphere_mat[np.argmax(flat.sphere_mat[:, 11])], (40, 1))   # the most transparent material
    flat.sphere_geom = np.vstack([flat.sphere_geom, row])
    flat.sphere_mat = np.vstack([flat.sphere_mat, glass])
    for shift in ((0, 0, 0), (1.0e5, -2.0e5, 3.0e4)):
        f = _shifted(flat, shift)
        out = _with_and_without_hierarchy(f, 96, 54, tuple(ACCELERATORS))
        assert out["1"][1:] == out["0"][1:]
        assert np.array_equal(out["1"][0], out["0"][0])
    want, stats = O.render(flat, seed=0)
    with Renderer(flat, RenderParams(width=96, height=54), precision="f64") as r:
        r.begin_rows(0, 54)
        got = r.render_pixels()
        c = r.counters()
    assert int(c["rays"]) == stats.traced_rays() and int(c["hits"]) == stats.hits()
    assert np.abs(got - want.reshape(-1, 3)).max() <= 1e-8


# ---------------------------------------------------------------------------------------------
# At the benchmark's size (3840x2160) neither the reference nor the oracle finishes in a test, so
# the frame is checked through properties that do not depend on the size: the same pixels rendered as
# part of the whole frame and as part of a block of rows that is given the whole frame's batch sizes
# are bit for bit equal (pixels are independent but for the small-batch rule); rendering again
# reproduces the frame; a sky pixel is exactly the background; the 8-bit frame is the quantised one;
# and the number of rays per pixel is that of the reference's recorded 1920x1080 run (the scene is
# the same, only sampled twice as densely).
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["example_2", "example_4"])
def test_benchmark_size_properties(name):
    from raytracing_amd.imageio import quantise
    W, H = 3840, 2160
    scene = build_scene(name, W, H)
    np.random.seed(0)
    with Renderer(scene) as r:
        r.begin_rows(0, H)
        r.trace()
        r.draw_samples()
        r.shade()
        img = r.image().reshape(H, W, 3).copy()
        cnt = r.counters()
        rgb8 = r.ctx.read_image_rgb8().reshape(H, W, 3).copy()
        r.ctx.trace_geometry()
        r.ctx.shade()
        again = r.image().reshape(H, W, 3)
        assert np.array_equal(again, img)
    assert img.min() >= 0.0 and img.max() <= 1.0
    assert np.array_equal(rgb8, quantise(img))
    assert np.all(img[0, 0] == np.array([0.251, 0.624, 0.769]))       # the corner pixel looks at the sky
    per_pixel = cnt["rays"] / (W * H)
    assert abs(per_pixel / (FULL_HD_REFERENCE_COUNTS[name]["rays"] / (1920 * 1080)) - 1.0) < 2e-3
    if name == "example_2":
        # rows 1000-1099 on their own, with the small-batch rule fed from the whole frame's batch sizes
        with Renderer(scene) as whole, Renderer(scene) as part:
            whole.begin_rows(0, H)
            whole.trace()
            part.begin_rows(1000, 100)
            for level in range(part.tree.n_levels):
                if level:
                    part.ctx.set_global_batches(level, whole.ctx.level_batches(level))
                part.ctx.trace_level(level)
            part.ctx.shade()
            strip = part.ctx.read_image().reshape(100, W, 3)
        assert np.array_equal(strip, img[1000:1100])


@pytest.mark.gpu
def test_hierarchy_matches_plain_screening_at_full_size():
    """The 4096-sphere scene at 3840x2160 (482 M rays): the walk through the grid (with the point
    light's occluder map) and the plain screening of every (ray, sphere) pair - two different ways to
    the candidates of the float64 test - give the same frame bit for bit and the same ray, hit and
    shaded-hit counts.  (The pin of this
    scene to the reference is the 64x36 golden, test_many_spheres_4096_through_the_hierarchy: the
    reference needs 290 s for that size.)"""
    from raytracing_amd.flatten import flatten_scene
    flat = flatten_scene(synthetic.many_spheres_scene(4096, 3840, 2160))
    out = _with_and_without_hierarchy(flat, 3840, 2160)
    assert out["1"][1:] == out["0"][1:]
    assert out["1"][1] > 58 * 3840 * 2160 * 0.95      # the reference traces 58.3 rays per pixel on this scene at 64x36
    assert np.array_equal(out["1"][0], out["0"][0])


# ---------------------------------------------------------------------------------------------
# The float32 sample pass claims that every shadow decision it does not hand to the float64
# re-evaluation equals the float64 decision.  "mixed_checked" re-evaluates every certified
# (sample, sphere) decision in float64 on the device and counts the differences.

def _disk_scene(width, height, offset=(0.0, 0.0, 0.0), light_radius=3.0, light_z=10.0, light_x=4.0, n_points=30,
                extra_spheres=0, spot=False):
    """example_4's content (reference Scenes/example_4.py:6-22), optionally moved away from the origin,
    with another light geometry, or with more small spheres (glass and opaque, some overlapping)."""
    from raytracing_amd import materials, objects
    from raytracing_amd.constants import BLUE, GREEN, GREY, RED, WHITE, YELLOW
    from raytracing_amd.Scenes.Scene import Scene
    ox, oy, oz = offset
    M = materials.Material
    spheres = [objects.Sphere(x=4 + ox, y=oy, z=-10000000 + oz, radius=10000000,
                              material=M(diffuse_color=WHITE, reflection_coefficient=0, smoothness=1)),
               objects.Sphere(x=ox, y=oy, z=1 + oz, radius=1,
                              material=M(diffuse_color=GREEN, reflection_coefficient=0, specular_coefficient=1,
                                         transparency_coefficient=0.9, refractive_index=1.05, smoothness=1,
                                         attenuation_coefficient=0.5)),
               objects.Sphere(x=15 + ox, y=oy, z=2 + oz, radius=4, material=M(diffuse_color=RED, smoothness=0.6))]
    rs = np.random.RandomState(11)
    palette = [BLUE, YELLOW, GREY, RED, GREEN]
    for i in range(extra_spheres):
        x, y, r = rs.uniform(1.5, 9), rs.uniform(-3, 3), rs.uniform(0.25, 0.7)
        glass = i % 2 == 0
        mat = (M(diffuse_color=palette[i % 5], transparency_coefficient=0.8, refractive_index=1.3, smoothness=0.5,
                 attenuation_coefficient=0.4) if glass else M(diffuse_color=palette[i % 5], reflection_coefficient=0.2))
        spheres.append(objects.Sphere(x=x + ox, y=y + oy, z=rs.uniform(0.8, 1.2) * r + oz, radius=r, material=mat))
    if spot:
        light = objects.DirectionalDiskSource(x=light_x + ox, y=oy, z=light_z + oz, radius=light_radius, intensity=100, angle=25,
                                              easing_mode="cubic")
        light.n_points = n_points
    else:
        light = objects.DiskSource(x=light_x + ox, y=oy, z=light_z + oz, radius=light_radius, intensity=100, N=n_points)
    camera = objects.Camera(x=-4 + ox, y=1 + oy, z=1 + oz, viewing_direction=np.array([1.0, 0.0, -0.1]),
                            pixels_x=width, pixels_y=height)
    return Scene(spheres, [light], camera, camera.screen, objects.AmbientLight(intensity=1 / 10, color=WHITE))


CERTIFY_SCENES = {
    "example_4_500": lambda: build_scene("example_4", 500, 500),
    "spot_linear": lambda: synthetic.spotlight_scene("linear", 64, 128, 72),
    "spot_exponential": lambda: synthetic.spotlight_scene("exponential", 64, 128, 72),
    "far_from_origin": lambda: _disk_scene(200, 113, offset=(1.0e5, -2.0e5, 3.0e4)),
    "tiny_light": lambda: _disk_scene(200, 113, light_radius=0.01),            # hard shadows: rays graze the occluders
    "odd_sample_count": lambda: _disk_scene(160, 90, n_points=7),
    "light_near_the_ground": lambda: _disk_scene(200, 113, light_radius=2.0, light_z=1.5, light_x=6.0),  # grazing, wide cone
    "light_touching_hits": lambda: _disk_scene(160, 90, light_radius=6.0, light_z=2.5),   # extent above half the distance
    "nine_spheres": lambda: _disk_scene(200, 113, extra_spheres=6),
    "forty_spheres_spot": lambda: _disk_scene(200, 113, extra_spheres=37, spot=True),
}


@pytest.mark.parametrize("name", sorted(CERTIFY_SCENES))
def test_certified_float32_shadow_decisions_equal_float64(name):
    scene = CERTIFY_SCENES[name]()
    want, _, _, _, _, cnt64 = _render(scene, seed=5, precision="f64")
    got, _, _, _, _, cnt = _render(scene, seed=5, precision="mixed_checked")
    assert cnt["certify_mismatches"] == 0
    assert np.abs(got - want).max() <= TOL_MIXED
    assert cnt["shaded_samples"] == cnt64["shaded_samples"] > 0
    # the float32 pass must do the bulk of the work itself
    assert cnt["fallback_samples"] <= 0.05 * cnt["shaded_samples"], (cnt["fallback_samples"], cnt["shaded_samples"])
    plain = _render(scene, seed=5, precision="mixed")
    assert np.array_equal(plain[0], got)


# ---------------------------------------------------------------------------------------------
# The soft-shadow configuration at its benchmark size: a spotlight with 64 samples per hit, one frame
# per easing mode, 3840x2160.  The reference needs minutes for 128x72 of it, so the full-size frames are
# held to properties: pixel (30 x, 30 y) of the big frame is the same ray as pixel (x, y) of the 128x72
# golden (reference objects.py:45-51 places pixel X at X * width / WIDTH, no half-pixel offset), so that
# sub-grid must equal the golden up to the noise of 64 differently drawn light samples; the geometry does
# not depend on the easing mode; rays, hits and in-beam hits per pixel are the golden's up to the sampling
# density; the stream length is 2 * 64 * in-beam hits; rendering again reproduces the frame; the 8-bit
# frame is the quantised one; the float32 sample pass stays within 2e-5 of the float64 one.
SPOT_MODES = ["linear", "quadratic", "cubic", "exponential"]


@pytest.mark.gpu
def test_spotlight_frames_at_benchmark_size(golden_dir):
    from raytracing_amd.imageio import quantise
    g = _load(golden_dir, "spot_128x72.npz")
    W, H = 3840, 2160
    counts = {}
    for mode in SPOT_MODES:
        scene = synthetic.spotlight_scene(mode, 64, W, H)
        imgs = {}
        for precision in ("mixed", "f64"):
            np.random.seed(0)
            with Renderer(scene, precision=precision) as r:
                r.begin_rows(0, H)
                r.trace()
                n_uniforms = r.draw_samples()
                r.shade()
                img = r.image().reshape(H, W, 3).copy()
                cnt = r.counters()
                if precision == "mixed":
                    rgb8 = r.ctx.read_image_rgb8().reshape(H, W, 3).copy()
                    r.ctx.trace_geometry()
                    r.ctx.shade()
                    assert np.array_equal(r.image().reshape(H, W, 3), img)
                    assert np.array_equal(rgb8, quantise(img))
                    in_beam = int(r.node_beam[:, 0].sum())
                    assert n_uniforms == 2 * 64 * in_beam
                    counts[mode] = (int(cnt["rays"]), int(cnt["hits"]), in_beam)
                    assert 0 <= cnt["fallback_samples"] < 1e-3 * cnt["shaded_samples"]
            imgs[precision] = img
        assert imgs["f64"].min() >= 0.0 and imgs["f64"].max() <= 1.0
        assert np.abs(imgs["mixed"] - imgs["f64"]).max() <= TOL_MIXED
        # the sub-grid of rays the golden frame consists of: same scene, other random numbers
        # (measured: mean difference 5e-3, 99th percentile 3e-2, largest 8e-2 - the spread of a mean of 64 samples)
        diff = np.abs(imgs["f64"][::30, ::30] - g[mode + "_image"])
        assert diff.mean() < 1e-2 and np.quantile(diff, 0.99) < 6e-2 and diff.max() < 0.2, (mode, diff.mean(), np.quantile(diff, 0.99), diff.max())
        rays, hits, beam = g[mode + "_rays"], g[mode + "_hits"], g[mode + "_beam"]
        want = np.array([rays[rays >= 3].sum(), hits.sum(), beam[:, 0].sum()]) / (128 * 72)
        got = np.array(counts[mode]) / (W * H)
        assert np.abs(got / want - 1.0).max() < 1e-2, (mode, got, want)
    assert len(set(counts.values())) == 1          # easing changes the light, not the geometry


# ---------------------------------------------------------------------------------------------
# The frame the multi-GPU scaling target is quoted on, example_4 at 7680x4320, on one GPU.  Same kind of
# properties: pixel (48 x, 48 y) is the ray of pixel (x, y) of the 160x90 golden (other random numbers
# for the disk light); the stream length is 2 * 30 * hits; a sky pixel is exactly the background; a
# second render reproduces the frame; the 8-bit frame is the quantised one; and rows 2000-2539 rendered
# on their own - given the whole frame's batch sizes (small-batch rule) and their columns of the whole
# frame's random stream, which is what one GPU of several does - equal the whole frame bit for bit.
@pytest.mark.gpu
def test_example_4_at_7680x4320_and_a_540_row_block(golden_dir):
    from raytracing_amd.imageio import quantise
    from raytracing_amd.tree import sample_offsets
    g = _load(golden_dir, "images_160x90.npz")
    W, H = 7680, 4320
    R0, ROWS = 2000, 540
    scene = build_scene("example_4", W, H)
    np.random.seed(0)
    state0 = np.random.get_state()
    with Renderer(scene) as r:
        r.begin_rows(0, H)
        r.trace()
        n_uniforms = r.draw_samples()
        r.shade()
        cnt = r.counters()
        block = r.image().reshape(H, W, 3)[R0:R0 + ROWS].copy()
        sub = r.image().reshape(H, W, 3)[::48, ::48].copy()
        rgb8_rows = r.ctx.read_image_rgb8().reshape(H, W, 3)[R0:R0 + ROWS].copy()
        whole_batches = [r.ctx.level_batches(level) for level in range(r.tree.n_levels)]
        whole_hits, whole_beam = r.node_hits.copy(), r.node_beam.copy()
        state_after = np.random.get_state()
        r.ctx.trace_geometry()
        r.ctx.shade()
        assert np.array_equal(r.image().reshape(H, W, 3)[R0:R0 + ROWS], block)
        tree, lights = r.tree, r.flat.lights
    assert n_uniforms == 2 * 30 * int(cnt["hits"])
    assert np.array_equal(rgb8_rows, quantise(block))
    assert np.all(sub[0, 0] == np.array([0.251, 0.624, 0.769]))
    per_pixel = cnt["rays"] / (W * H)
    assert abs(per_pixel / (FULL_HD_REFERENCE_COUNTS["example_4"]["rays"] / (1920 * 1080)) - 1.0) < 2e-3
    diff = np.abs(sub - g["example_4_image"])
    assert diff.mean() < 1e-2 and np.quantile(diff, 0.99) < 6e-2 and diff.max() < 0.2, (diff.mean(), np.quantile(diff, 0.99), diff.max())

    def geometry_of_rows(r, row0, n_rows):
        arena = 24 * n_rows * W
        while True:                                     # rows through the spheres need more rays per pixel than the frame's mean
            r.begin_rows(row0, n_rows, arena_rays=arena)
            for level in range(r.tree.n_levels):
                if level:
                    r.ctx.set_global_batches(level, whole_batches[level])
                r.ctx.trace_level(level)
            try:
                return r.ctx.node_counts()
            except _native.RtOverflow:
                arena *= 2

    with Renderer(scene) as above:                      # rows before the block: how many hits of every node precede it
        _, first_hits, _ = geometry_of_rows(above, 0, R0)
    with Renderer(scene) as part:
        _, hits, beam = geometry_of_rows(part, R0, ROWS)
        g_off, g_cnt, g_total = sample_offsets(tree, lights, whole_hits, whole_beam)
        l_off, _, l_total = sample_offsets(tree, lights, hits, beam)
        assert g_total == n_uniforms
        part.ctx.set_sample_layout(l_off, l_total)
        part.ctx.set_sample_layout_global(g_off, g_cnt, first_hits[:, None].repeat(len(lights), axis=1), g_total)
        key, pos = part.ctx.generate_samples(state0[1], state0[2])
        assert pos == state_after[2] and np.array_equal(key, state_after[1])
        part.ctx.shade()
        assert np.array_equal(part.ctx.read_image().reshape(ROWS, W, 3), block)
