"""rt_upload_samples_noalloc: the disk-light sample upload that allocates no device memory (its
staging array is part of the library's code object) must leave exactly what rt_upload_samples
leaves.  Two frames in a row through the same array, the second one smaller, each compared bit
for bit with the allocating entry point; oversized pieces are refused."""
import numpy as np
import pytest

from raytracing_amd import _native
from raytracing_amd.constants import RenderParams
from raytracing_amd.flatten import flatten_scene
from raytracing_amd.Scenes.scenes import build_scene
from raytracing_amd.tree import build_tree, sample_offsets

pytestmark = pytest.mark.gpu


def _render(width, height, precision, upload, piece):
    """example_4 with host-drawn samples (seed 0) handed over through ``upload(ctx, offset, values)``
    in pieces of ``piece`` values."""
    flat = flatten_scene(build_scene("example_4", width, height))
    params = RenderParams(width=width, height=height)
    tree = build_tree(params.max_reflection_depth, params.max_refraction_depth)
    with _native.Context(0) as ctx:
        ctx.set_scene(flat, params, tree)
        ctx.set_precision(precision)
        ctx.begin_rows(0, height, width)
        ctx.trace_geometry()
        _, hits, beam = ctx.node_counts()
        offsets, _, total = sample_offsets(tree, flat.lights, hits, beam)
        ctx.set_sample_layout(offsets, total)
        u = np.random.RandomState(0).random_sample(total)
        for pos in range(0, total, piece):
            upload(ctx, pos, u[pos:pos + piece])
        ctx.shade()
        return ctx.read_image(), total


@pytest.mark.parametrize("precision", ["f64", "mixed"])
def test_noalloc_upload_matches_the_allocating_one(precision):
    plain = lambda ctx, pos, v: ctx.upload_samples(pos, v)
    noalloc = lambda ctx, pos, v: ctx.upload_samples_noalloc(pos, v)
    # first call(s): a frame whose stream needs several full pieces of the staging array ...
    big_ref, total_big = _render(160, 90, precision, plain, 1 << 22)
    big, _ = _render(160, 90, precision, noalloc, _native.NOALLOC_MAX_SAMPLES)
    assert total_big > 3 * _native.NOALLOC_MAX_SAMPLES
    assert np.array_equal(big, big_ref)
    # ... then a smaller frame in odd-sized, much shorter pieces: the array still holds the tail of
    # the first frame's stream, none of which may reach the second frame
    small_ref, total_small = _render(64, 36, precision, plain, 1 << 22)
    small, _ = _render(64, 36, precision, noalloc, 99_991)
    assert total_small < total_big
    assert np.array_equal(small, small_ref)


def test_noalloc_upload_leaves_free_device_memory_unchanged():
    """hipMemGetInfo before and after the calls: the allocating entry point grows its staging
    buffer on the first call of a context, the other one takes nothing."""
    import torch
    flat = flatten_scene(build_scene("example_4", 160, 90))
    params = RenderParams(width=160, height=90)
    tree = build_tree(params.max_reflection_depth, params.max_refraction_depth)
    u = np.random.RandomState(1).random_sample(_native.NOALLOC_MAX_SAMPLES)
    with _native.Context(0) as ctx:
        ctx.set_scene(flat, params, tree)
        ctx.begin_rows(0, 90, 160)
        ctx.trace_geometry()
        _, hits, beam = ctx.node_counts()
        offsets, _, total = sample_offsets(tree, flat.lights, hits, beam)
        ctx.set_sample_layout(offsets, total)
        assert total > 2 * u.size
        ctx.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        ctx.upload_samples_noalloc(0, u)
        ctx.upload_samples_noalloc(u.size, u[:12345])
        ctx.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free_before
        ctx.upload_samples(0, u)                       # control: this one allocates its staging buffer
        ctx.synchronize()
        assert torch.cuda.mem_get_info(0)[0] < free_before


def test_noalloc_upload_refuses_more_than_its_array_holds():
    flat = flatten_scene(build_scene("example_4", 160, 90))
    params = RenderParams(width=160, height=90)
    tree = build_tree(params.max_reflection_depth, params.max_refraction_depth)
    with _native.Context(0) as ctx:
        ctx.set_scene(flat, params, tree)
        ctx.begin_rows(0, 90, 160)
        ctx.trace_geometry()
        _, hits, beam = ctx.node_counts()
        offsets, _, total = sample_offsets(tree, flat.lights, hits, beam)
        ctx.set_sample_layout(offsets, total)
        with pytest.raises(_native.RtError, match="at most"):
            ctx.upload_samples_noalloc(0, np.zeros(_native.NOALLOC_MAX_SAMPLES + 1))
        # and shading with an incomplete stream is refused instead of reading unwritten samples
        ctx.upload_samples_noalloc(0, np.zeros(1000))
        with pytest.raises(_native.RtError, match="sample values"):
            ctx.shade()
