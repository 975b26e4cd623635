"""Host-side logic of the multi-GPU path, run on CPUs: the level-synchronous CPU
stand-in replaces the device context, torch.distributed's gloo backend replaces
RCCL, two processes replace two GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import numpy_oracle as O
from oracle.level_backend import CpuContext
from raytracing_amd.constants import RenderParams
from raytracing_amd.distributed import Comm, ShardedRenderer, raytrace_sharded, split_rows
from raytracing_amd.flatten import flatten_scene
from raytracing_amd.Scenes import synthetic
from raytracing_amd.Scenes.scenes import build_scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _scene(kind, w, h):
    if kind.startswith("spot:"):
        return synthetic.spotlight_scene(kind[5:], 16, w, h)
    return build_scene(kind, w, h)


# ------------------------------------------------------------------ single process
@pytest.mark.parametrize("name", ["example_1", "example_2", "example_3", "example_4"])
def test_level_synchronous_stand_in_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    np.random.seed(0)
    img = raytrace_sharded(build_scene(name, 160, 90), None, Comm(None), CpuContext())
    assert np.abs(img - g[name + "_image"]).max() <= 1e-9


@pytest.mark.parametrize("mode", ["none", "cubic"])
def test_level_synchronous_stand_in_spotlight(mode):
    g = np.load(os.path.join(GOLDEN, "spot_128x72.npz"))
    np.random.seed(0)
    img = raytrace_sharded(synthetic.spotlight_scene(mode, 64, 128, 72), None, Comm(None), CpuContext())
    assert np.abs(img - g[mode + "_image"]).max() <= 1e-9


def test_stand_in_reports_arena_overflow_and_the_renderer_retries():
    scene = build_scene("example_3", 96, 54)
    r = ShardedRenderer(scene, None, Comm(None), CpuContext())
    r.choose_rows()
    r.arena_rays = 96 * 54 + 512
    r.trace_checked()
    assert r.arena_rays > 96 * 54 + 512
    r.shade()
    want, _ = O.render(scene, seed=0)
    assert np.abs(r.ctx.read_image().reshape(54, 96, 3) - want).max() <= 1e-12


def test_split_rows_balances_cost():
    rows = np.array([0, 10, 20, 30, 40])
    b = split_rows(40, 2, rows, [0.0, 0.0, 100.0, 100.0])
    assert list(b) == [0, 30, 40]
    b = split_rows(40, 4, rows, [1.0, 1.0, 1.0, 1.0])
    assert list(b) == [0, 10, 20, 30, 40]
    b = split_rows(8, 8, np.array([0, 8]), [0.0])          # no cost anywhere: still one row each
    assert list(b) == list(range(9))
    b = split_rows(100, 3, np.array([0, 50, 100]), [0.0, 9.0])
    assert b[0] == 0 and b[-1] == 100 and np.all(np.diff(b) >= 1) and b[1] > 50


# ------------------------------------------------------------------ two processes, gloo
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _watchdog(seconds=240):
    """A worker that is still alive after this long dumps its stacks and exits: a rank stuck in an
    exchange must not outlive the test and keep the GPU (or the port) busy."""
    import faulthandler
    faulthandler.dump_traceback_later(seconds, exit=True)


def _worker(rank, world, port, kind, w, h, seed, out_path):
    _watchdog()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = Comm(dist.group.WORLD)
        np.random.seed(seed)
        r = ShardedRenderer(_scene(kind, w, h), None, comm, CpuContext())
        bounds = r.choose_rows()
        r.trace_checked()
        total = r.draw_samples()
        r.shade()
        r0, r1 = r.my_rows
        tile = torch.from_numpy(r.ctx.read_image()[:(r1 - r0) * w].copy())
        full = torch.empty((h * w, 3), dtype=torch.float64) if rank == 0 else None
        if rank == 0:
            full[r0 * w:r1 * w] = tile
        comm.gather_tiles(full, tile, bounds, w)
        stats = r.global_stats()
        comm.barrier()
        if rank == 0:
            np.savez(out_path, image=full.numpy().reshape(h, w, 3), bounds=bounds, rays=stats["rays"],
                     hits=stats["hits"], uniforms=total)
    finally:
        dist.destroy_process_group()


def _run_two_ranks(tmp_path, kind, w, h, seed=0):
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), kind, w, h, seed, out), nprocs=2, join=True)
    return np.load(out)


def test_two_ranks_whole_image_batches(tmp_path):
    """example_3 rendered in row strips WITHOUT the batch exchange differs from the
    whole-frame render (see the strips fixture); with the exchange two ranks
    reproduce the whole-frame image and ray counts exactly."""
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    res = _run_two_ranks(tmp_path, "example_3", 160, 90)
    assert 0 < res["bounds"][1] < 90
    assert np.abs(res["image"] - g["example_3_image"]).max() <= 1e-9
    r = g["example_3_rays"]
    assert int(res["rays"]) == int(r[r >= 3].sum())
    assert int(res["hits"]) == int(g["example_3_hits"].sum())


def test_two_ranks_disk_light_samples(tmp_path):
    """Each rank keeps its own columns of the global random stream."""
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    res = _run_two_ranks(tmp_path, "example_4", 160, 90)
    assert np.abs(res["image"] - g["example_4_image"]).max() <= 1e-9
    assert int(res["uniforms"]) == int(g["example_4_uniforms"])


def test_two_ranks_spotlight_in_beam_ranks(tmp_path):
    scene = synthetic.spotlight_scene("quadratic", 16, 96, 54)
    want, stats = O.render(scene, seed=5)
    res = _run_two_ranks(tmp_path, "spot:quadratic", 96, 54, seed=5)
    assert np.abs(res["image"] - want).max() <= 1e-12
    assert int(res["uniforms"]) == stats.uniforms_drawn


def _worker_rgb8(rank, world, port, name, w, h, out_path):
    _watchdog()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        np.random.seed(0)
        img = raytrace_sharded(build_scene(name, w, h), None, Comm(dist.group.WORLD), CpuContext(), output="rgb8")
        if rank == 0:
            np.save(out_path, img)
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


def test_two_ranks_gather_8bit_tiles(tmp_path):
    """The tiles travel as the 8-bit image of the reference's output step."""
    from raytracing_amd.imageio import quantise
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    out = str(tmp_path / "rgb8.npy")
    mp.spawn(_worker_rgb8, args=(2, _free_port(), "example_2", 160, 90, out), nprocs=2, join=True)
    img = np.load(out)
    assert img.dtype == np.uint8 and img.shape == (90, 160, 3)
    assert (img == quantise(g["example_2_image"])).all()


def _worker_too_many_ranks(rank, world, port, out_dir):
    _watchdog(120)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r = ShardedRenderer(build_scene("example_2", 16, 1), None, Comm(dist.group.WORLD), CpuContext())
        try:
            r.choose_rows()
            verdict = "no error"
        except ValueError as exc:
            verdict = str(exc)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(verdict)
    finally:
        dist.destroy_process_group()


def test_more_ranks_than_rows_is_refused_on_every_rank(tmp_path):
    """A rank without rows would still have to take part in every exchange; the renderer refuses the
    split on all ranks before the first collective, so none is left waiting for the others."""
    mp.spawn(_worker_too_many_ranks, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert "every rank needs at least one row" in open(tmp_path / f"rank{rank}.txt").read()


# ------------------------------------------------------------------ raytrace(scene, devices=[...])
@pytest.mark.parametrize("name", ["example_3", "example_4"])
def test_one_call_starts_its_ranks_and_returns_the_frame(name):
    """The single-call form: the call starts one fresh process per entry of ``devices`` (here two
    CPU ranks over gloo with the level-synchronous stand-in for the device), rank 0 gathers the
    tiles, the caller gets the frame, and NumPy's global stream is left where the reference's own
    draws would leave it."""
    from raytracing_amd.launcher import raytrace_on_devices
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    np.random.seed(0)
    img = raytrace_on_devices(build_scene(name, 160, 90), [0, 1], backend="oracle.level_backend:CpuContext", timeout=240)
    assert img.shape == (90, 160, 3) and img.dtype == np.float64
    assert np.abs(img - g[name + "_image"]).max() <= 1e-9
    want = np.random.RandomState(0)
    want.random_sample(int(g[name + "_uniforms"]))
    got = np.random.get_state()
    assert got[2] == want.get_state()[2] and np.array_equal(got[1], want.get_state()[1])


def test_one_call_reports_a_failing_rank():
    from raytracing_amd.launcher import raytrace_on_devices
    with pytest.raises(RuntimeError, match="rank [01] ended with code"):
        raytrace_on_devices(build_scene("example_2", 32, 18), [0, 1], backend="oracle.level_backend:NoSuchFactory", timeout=120)
    with pytest.raises(ValueError, match="at least one row"):
        raytrace_on_devices(build_scene("example_2", 16, 1), [0, 1], backend="oracle.level_backend:CpuContext")
