"""The multi-GPU host path on real device contexts: two processes share the one
GPU of the test box (gloo carries the host-side exchanges, so no second GPU is
needed), each renders its block of rows through the level-by-level C ABI calls
(rt_trace_level / rt_get_level_batches / rt_set_global_batches), the per-rank
sample columns are uploaded, and rank 0 assembles the frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, w, h, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracing_amd import _native
        from raytracing_amd.distributed import Comm, ShardedRenderer
        from raytracing_amd.Scenes.scenes import build_scene
        comm = Comm(dist.group.WORLD)               # CPU tensors over gloo
        ctx = _native.Context(0)
        np.random.seed(0)
        r = ShardedRenderer(build_scene(name, w, h), None, comm, ctx)
        bounds = r.choose_rows()
        r.trace_checked()
        total = r.draw_samples()
        # the samples come from the generator running on the GPU over the whole image's stream;
        # every rank's NumPy stream must be where numpy.random.random(total) would leave it
        want = np.random.RandomState(0)
        want.random_sample(total)
        got = np.random.get_state()
        assert got[2] == want.get_state()[2] and np.array_equal(got[1], want.get_state()[1])
        r.shade()
        r0, r1 = r.my_rows
        tile = torch.from_numpy(ctx.read_image()[:(r1 - r0) * w].copy())
        full = torch.empty((h * w, 3), dtype=torch.float64) if rank == 0 else None
        if rank == 0:
            full[r0 * w:r1 * w] = tile
        comm.gather_tiles(full, tile, bounds, w)
        stats = r.global_stats()
        comm.barrier()
        if rank == 0:
            np.savez(out_path, image=full.numpy().reshape(h, w, 3), bounds=bounds, rays=stats["rays"], hits=stats["hits"])
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,tol", [("example_3", 1e-8), ("example_4", 2e-5)])
def test_two_ranks_on_one_gpu(tmp_path, name, tol):
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), name, 160, 90, out), nprocs=2, join=True)
    res = np.load(out)
    assert 0 < res["bounds"][1] < 90
    assert np.abs(res["image"] - g[name + "_image"]).max() <= tol
    r = g[name + "_rays"]
    assert int(res["rays"]) == int(r[r >= 3].sum())
    assert int(res["hits"]) == int(g[name + "_hits"].sum())


def _worker_rgb8(rank, world, port, name, w, h, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracing_amd import _native
        from raytracing_amd.distributed import Comm, raytrace_sharded
        from raytracing_amd.Scenes.scenes import build_scene
        ctx = _native.Context(0)
        np.random.seed(0)
        img = raytrace_sharded(build_scene(name, w, h), None, Comm(dist.group.WORLD), ctx, output="rgb8")
        if rank == 0:
            np.save(out_path, img)
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_gather_8bit_tiles(tmp_path):
    """Each rank quantises its rows on the GPU; rank 0 assembles the 8-bit frame."""
    from raytracing_amd.imageio import quantise
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    out = str(tmp_path / "rgb8.npy")
    mp.spawn(_worker_rgb8, args=(2, _free_port(), "example_1", 160, 90, out), nprocs=2, join=True)
    img = np.load(out)
    assert img.dtype == np.uint8 and img.shape == (90, 160, 3)
    assert (img == quantise(g["example_1_image"])).all()
