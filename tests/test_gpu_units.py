"""The device functions the frame kernels are built from, evaluated on the GPU through
rt_debug_eval for the input vectors recorded from the reference's own functions
(tests/golden/unit_kats.npz).  Needs an MI355X: run with -m gpu."""
import os

import numpy as np
import pytest

from raytracing_amd import _native
from raytracing_amd import flatten as F
from raytracing_amd.constants import RenderParams
from raytracing_amd.Scenes.scenes import build_scene
from raytracing_amd.tree import build_tree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kats(golden_dir):
    return np.load(os.path.join(golden_dir, "unit_kats.npz"))


def _context(flat, width=160, height=90):
    params = RenderParams(width=width, height=height)
    ctx = _native.Context(0)
    ctx.set_scene(flat, params, build_tree(params.max_reflection_depth, params.max_refraction_depth))
    return ctx


def _flat_from_geom(geom):
    flat = F.flatten_scene(build_scene("example_2", 160, 90))
    geom = np.asarray(geom, dtype=float).reshape(-1, 4)
    flat.sphere_geom = geom.copy()
    flat.sphere_mat = np.tile(flat.sphere_mat[2], (geom.shape[0], 1))
    return flat


def test_sphere_roots_bit_for_bit(kats):
    """Sphere.intersection + solve_quadratic, both modes, on every example sphere including the
    grounds of radius 1e6 and 1e7 (reference objects.py:60-66, :238-271): the device follows the
    reference's rounding order, so the roots are the same doubles."""
    geom = kats["isect_geom"]
    with _context(_flat_from_geom(geom)) as ctx:
        for i in range(geom.shape[0]):
            rows = np.concatenate([kats["isect_o"][i], kats["isect_d"][i], np.full((256, 1), float(i))], axis=1)
            got = ctx.debug_eval("sphere_roots", rows)
            assert np.array_equal(got[:, 0], kats["isect_closest"][i]), i
            assert np.array_equal(got[:, 1], kats["isect_furthest"][i]), i
        assert (kats["isect_closest"] == -1).any() and (kats["isect_closest"] > 0).any()
        with pytest.raises(_native.RtError):
            ctx.debug_eval("sphere_roots", np.array([[0, 0, 0, 1, 0, 0, 99.0]]))


def test_nearest_hit_and_the_tie_rule(kats):
    """find_closest_intersected_object (reference objects.py:288-299): two identical spheres, the
    later one wins; misses keep index -1 and distance inf."""
    with _context(_flat_from_geom(kats["near_geom"])) as ctx:
        got = ctx.debug_eval("nearest", np.concatenate([kats["near_o"], kats["near_d"]], axis=1))
    assert np.array_equal(got[:, 0].astype(int), kats["near_id"])
    assert np.array_equal(got[:, 1], kats["near_t"])
    assert (kats["near_id"] == 1).any() and (kats["near_id"] == -1).any()


def test_shadow_multipliers(kats):
    """shadow_objects_multipliers through the spheres of example_2 (reference objects.py:274-285)."""
    with _context(F.flatten_scene(build_scene("example_2", 160, 90))) as ctx:
        got = ctx.debug_eval("shadow", np.concatenate([kats["shadow_o"], kats["shadow_d"]], axis=1))
    assert np.abs(got - kats["shadow_mult"]).max() <= 1e-15      # exp() may differ in the last bit


def test_refraction_fresnel_reflection(kats):
    """refract_vectors / fresnel_multiplier / reflect_vectors (reference raytrace.py:173-212) incl.
    total internal reflection, n1 = 2.4 and n1 == n2.  For totally reflected rays the device
    returns R = 1, which is what the tracer uses (raytrace.py:99); the recorded Fresnel value of
    those rows is the formula with T = 0 and is not compared."""
    rows = np.concatenate([kats["opt_normal"], kats["opt_incident"], kats["opt_n1"][:, None], kats["opt_n2"][:, None]], axis=1)
    with _context(F.flatten_scene(build_scene("example_2", 160, 90))) as ctx:
        got = ctx.debug_eval("optics", rows)
    mask = kats["opt_mask"]
    assert np.array_equal(got[:, 3] == 1, mask) and (~mask).sum() > 50
    assert np.abs(got[:, 0:3] - kats["opt_transmitted"]).max() <= 1e-15
    assert np.abs(got[mask, 4] - kats["opt_fresnel"][mask]).max() <= 1e-15   # x^5 by multiplication vs pow()
    assert (got[~mask, 4] == 1).all()
    assert np.abs(got[:, 5:8] - kats["opt_reflect"]).max() <= 1e-15


@pytest.mark.parametrize("mode,code", [("none", 0), ("linear", 1), ("quadratic", 2), ("cubic", 3), ("exponential", 4)])
def test_easing(kats, mode, code):
    """ease_fall_off_beam with its d == 0 guard (reference objects.py:176-191, :302-338)."""
    n = kats["ease_x"].shape[0]
    with _context(F.flatten_scene(build_scene("example_2", 160, 90))) as ctx:
        rows = np.stack([np.full(n, float(code)), kats["ease_x"], kats["ease_a"], kats["ease_dz"]], axis=1)
        guarded = ctx.debug_eval("ease", rows)[:, 0]
        rows[:, 3] = kats["ease_d"]
        plain = ctx.debug_eval("ease", rows)[:, 0]
    assert np.abs(guarded - kats["beam_" + mode]).max() <= 1e-14
    assert (guarded[kats["ease_dz"] == 0] == 1).all()
    if mode != "none":
        # linear is clamped and the polynomial modes are split into regions inside the reference's
        # functions themselves; exponential is never clamped
        assert np.abs(plain - kats["ease_" + mode]).max() <= 1e-14


def test_screen_points_and_primary_directions(kats):
    """Screen.index_to_position for fractional pixel indices and the four example cameras
    (reference objects.py:45-51); the direction is the normalised difference to the camera
    (raytrace.py:8-10)."""
    rows = np.concatenate([kats["screen_X"], kats["screen_Y"]], axis=1)
    for i, name in enumerate(["example_1", "example_2", "example_3", "example_4"]):
        flat = F.flatten_scene(build_scene(name, 160, 90))
        with _context(flat) as ctx:
            got = ctx.debug_eval("screen", rows)
        want = kats["screen_points"][i]
        assert np.array_equal(got[:, :3], want), name
        d = want - flat.camera_position
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        assert np.array_equal(got[:, 3:], d), name


def test_trim_gives_the_sample_work_buffers_back():
    """rt_trim releases what the sample calls keep between frames (raw words of the generator, staging),
    and the next frame simply allocates again and reproduces the image."""
    import torch
    from raytracing_amd.raytrace import Renderer
    scene = build_scene("example_4", 320, 180)
    with Renderer(scene) as r:
        r.begin_rows(0, 180)
        np.random.seed(5)
        first = r.render_pixels().copy()
        r.ctx.synchronize()
        held = torch.cuda.mem_get_info(0)[0]
        r.ctx.trim()
        assert torch.cuda.mem_get_info(0)[0] > held
        np.random.seed(5)
        assert np.array_equal(r.render_pixels(), first)


def test_generator_stores_what_shaded_hits_read(monkeypatch):
    """RtCounters.generator_words / generator_words_stored: with the store map (default) well under half of
    the stream's words reach the work buffer in the example scene and in a spotlight scene, all of them
    with RT_MT_STORE_ALL=1; the frames are the same."""
    from raytracing_amd.raytrace import Renderer
    from raytracing_amd.Scenes.scenes import build_scene as by_name
    for name in ("example_4", "spot:quadratic:16"):
        frames = {}
        for store_all in (False, True):
            if store_all:
                monkeypatch.setenv("RT_MT_STORE_ALL", "1")
            else:
                monkeypatch.delenv("RT_MT_STORE_ALL", raising=False)
            np.random.seed(21)
            with Renderer(by_name(name, 480, 270)) as r:
                r.begin_rows(0, 270)
                img = r.render_pixels().copy()
                c = r.ctx.counters()
            frames[store_all] = img
            assert c["generator_words"] == 2 * r.samples_total + 624
            if store_all:
                assert c["generator_words_stored"] == c["generator_words"]
            else:
                assert 0 < c["generator_words_stored"] < 0.5 * c["generator_words"], (name, c)
        assert np.array_equal(frames[False], frames[True])


def test_whole_frame_on_one_gpu_with_a_late_overflow_check():
    """ShardedRenderer.frame(): frames without sampled lights are queued as a whole and the arena's
    overflow flag is looked at with the image; too small an arena is doubled and the frame repeated."""
    from raytracing_amd.distributed import Comm, ShardedRenderer
    from raytracing_amd.raytrace import raytrace
    for name in ("example_3", "example_4"):
        scene = build_scene(name, 160, 90)
        np.random.seed(4)
        want = raytrace(scene)
        ctx = _native.Context(0)
        try:
            r = ShardedRenderer(scene, None, Comm(None), ctx)
            r.choose_rows()
            r.arena_rays = 160 * 90 + 512
            np.random.seed(4)
            got = r.frame().reshape(90, 160, 3)
            assert r.arena_rays > 160 * 90 + 512
            assert np.array_equal(got, want)
            assert int(r.global_hits.sum()) == ctx.counters()["hits"]
        finally:
            ctx.close()


def test_samples_kept_as_raw_words_follow_every_later_call(monkeypatch):
    """Generated samples are packed as disk points only; the float64 readers (re-evaluation of hits the
    float32 pass cannot certify, the checking pass, float64 shading) take the uniforms from the raw words,
    or from the packed array once something has moved them there (rt_trim, the checking mode).  Every
    route gives the frame of RT_PACK_UNIFORMS=1 bit for bit - also a shading graph recorded before rt_trim."""
    import torch
    from raytracing_amd.raytrace import Renderer
    scene = build_scene("example_4", 320, 180)
    stream = torch.cuda.Stream()

    def frames(packed):
        if packed:
            monkeypatch.setenv("RT_PACK_UNIFORMS", "1")
        else:
            monkeypatch.delenv("RT_PACK_UNIFORMS", raising=False)
        out = {}
        with torch.cuda.stream(stream):
            with Renderer(scene, stream=stream.cuda_stream) as r:
                r.begin_rows(0, 180)
                np.random.seed(8)
                r.trace()
                r.draw_samples()
                r.ctx.shade()
                out["mixed"] = r.image().copy()
                out["fallback"] = r.ctx.counters()["fallback_samples"]
                g = torch.cuda.CUDAGraph()
                g.capture_begin(capture_error_mode="relaxed")
                try:
                    r.ctx.shade()
                finally:
                    g.capture_end()
                r.ctx.set_precision("f64")               # float64 lighting over the float32 colour arrays of this frame
                r.ctx.shade()
                out["f64"] = r.image().copy()
                r.ctx.set_precision("mixed")
                r.ctx.trim()                             # the raw words go away: the uniforms are unpacked first
                g.replay()
                stream.synchronize()
                out["replayed after trim"] = r.image().copy()
                r.ctx.set_precision("mixed_checked")
                r.ctx.shade()
                out["checked"] = r.image().copy()
                out["mismatches"] = r.ctx.counters()["certify_mismatches"]
        return out

    lean, full = frames(False), frames(True)
    assert lean["fallback"] > 0 and lean["fallback"] == full["fallback"]      # some hits do take the float64 route
    assert lean["mismatches"] == 0 and full["mismatches"] == 0
    for key in ("mixed", "f64", "replayed after trim", "checked"):
        assert np.array_equal(lean[key], full[key]), key
    assert np.array_equal(lean["mixed"], lean["replayed after trim"]) and np.array_equal(lean["mixed"], lean["checked"])


def test_first_shade_after_new_samples_is_refused_under_stream_capture():
    """The first rt_shade after uploaded disk-light samples converts them for the float32 pass; recorded
    into a graph that conversion would not run when a later, uncaptured rt_shade expects it.  The call
    refuses (nothing is launched), and after one eager rt_shade the capture works and replays the frame.
    (Samples generated on the device in one pass come with their converted form: nothing is pending.)"""
    import torch
    from raytracing_amd.raytrace import Renderer
    scene = build_scene("example_4", 160, 90)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        with Renderer(scene, stream=stream.cuda_stream) as r:
            r.begin_rows(0, 90)
            np.random.seed(2)
            r.trace()
            r.draw_samples()
            g = torch.cuda.CUDAGraph()
            g.capture_begin(capture_error_mode="relaxed")
            try:
                r.ctx.shade()
            finally:
                g.capture_end()
            g.replay()
            stream.synchronize()
            generated = r.image().copy()
        with Renderer(scene, stream=stream.cuda_stream, sampler="host") as r:
            r.begin_rows(0, 90)
            np.random.seed(2)
            r.trace()
            r.draw_samples()
            g = torch.cuda.CUDAGraph()
            g.capture_begin(capture_error_mode="relaxed")
            try:
                with pytest.raises(_native.RtError, match="captured"):
                    r.ctx.shade()
            finally:
                g.capture_end()
            r.ctx.shade()
            want = r.image().copy()
            g2 = torch.cuda.CUDAGraph()
            g2.capture_begin(capture_error_mode="relaxed")
            try:
                r.ctx.trace_geometry()
                r.ctx.shade()
            finally:
                g2.capture_end()
            g2.replay()
            stream.synchronize()
            assert np.array_equal(r.image(), want)
            assert np.array_equal(generated, want)
