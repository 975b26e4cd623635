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


def _watchdog(seconds=240):
    """A worker that is still alive after this long dumps its stacks and exits: a rank stuck in an
    exchange must not outlive the test and keep the GPU (or the port) busy."""
    import faulthandler
    faulthandler.dump_traceback_later(seconds, exit=True)


def _worker(rank, world, port, name, w, h, out_path, precision="mixed", env=None):
    _watchdog()
    os.environ["RT_MT_POISON"] = "1"    # raw words the generator does not store hold a pattern, not an earlier frame's values
    os.environ.update(env or {})
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracing_amd import _native
        from raytracing_amd.distributed import Comm, ShardedRenderer
        from raytracing_amd.Scenes.scenes import build_scene
        comm = Comm(dist.group.WORLD)               # CPU tensors over gloo
        ctx = _native.Context(0)
        ctx.set_precision(precision)
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
        c = ctx.counters()
        generator = comm.all_gather(np.array([c["generator_words"], c["generator_words_stored"]], dtype=np.int64))
        comm.barrier()
        if rank == 0:
            np.savez(out_path, image=full.numpy().reshape(h, w, 3), bounds=bounds, rays=stats["rays"], hits=stats["hits"],
                     generator=generator, total=total)
        ctx.close()
    finally:
        dist.destroy_process_group()


# float64 mode: 1e-8; mixed (default) mode, float32 samples and float32 colours between the levels: 2e-5
@pytest.mark.parametrize("name,precision,tol", [("example_3", "f64", 1e-8), ("example_3", "mixed", 2e-5), ("example_4", "mixed", 2e-5)])
def test_two_ranks_on_one_gpu(tmp_path, name, precision, tol):
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), name, 160, 90, out, precision), nprocs=2, join=True)
    res = np.load(out)
    assert 0 < res["bounds"][1] < 90
    assert np.abs(res["image"] - g[name + "_image"]).max() <= tol
    r = g[name + "_rays"]
    assert int(res["rays"]) == int(r[r >= 3].sum())
    assert int(res["hits"]) == int(g[name + "_hits"].sum())


def test_two_ranks_make_the_stream_in_passes(tmp_path):
    """Row blocks on two ranks with a work buffer far smaller than the whole image's stream: passes of whole
    (node, light) blocks (8 M words) and passes by ranges (2 M words: the root's block does not fit)."""
    from raytracing_amd.raytrace import Renderer
    from raytracing_amd.Scenes.scenes import build_scene
    w, h = 320, 180
    np.random.seed(0)
    with Renderer(build_scene("example_4", w, h)) as r:
        r.begin_rows(0, h)
        one = r.render_pixels().reshape(h, w, 3).copy()
    for words in (1 << 23, 1 << 21):
        out = str(tmp_path / f"out{words}.npz")
        mp.spawn(_worker, args=(2, _free_port(), "example_4", w, h, out, "mixed", {"RT_MT_RAW_WORDS": str(words)}), nprocs=2, join=True)
        res = np.load(out)
        assert 2 * int(res["total"]) > 4 * words
        assert np.abs(res["image"] - one).max() <= 2e-5


@pytest.mark.parametrize("name", ["example_4", "spot:cubic:16"])
def test_two_ranks_store_their_own_part_of_the_stream(tmp_path, name):
    """Every rank runs the generator over the whole image's stream (the same final state everywhere) but
    stores only the words of values its own shaded hits read: the memory traffic of the sample set-up
    divides between the ranks, its arithmetic does not.  (The spotlight's columns are numbered among
    the hits inside its beam.)"""
    from raytracing_amd.raytrace import Renderer
    from raytracing_amd.Scenes.scenes import build_scene
    w, h = 320, 180
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), name, w, h, out, "mixed"), nprocs=2, join=True)
    res = np.load(out)
    np.random.seed(0)
    with Renderer(build_scene(name, w, h)) as r:
        r.begin_rows(0, h)
        one = r.render_pixels().reshape(h, w, 3).copy()
        alone = r.ctx.counters()
    assert np.abs(res["image"] - one).max() <= 2e-5
    words = 2 * int(res["total"])
    assert alone["generator_words"] == words + 624        # the stream starts behind a fresh key block
    for made, stored in res["generator"]:
        assert made == alone["generator_words"]            # not divided
        assert stored <= 0.6 * made
        assert stored <= 0.8 * alone["generator_words_stored"]
    assert res["generator"][:, 1].sum() <= 1.25 * alone["generator_words_stored"]


def _worker_rgb8(rank, world, port, name, w, h, out_path):
    _watchdog()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracing_amd import _native
        from raytracing_amd.distributed import Comm, raytrace_sharded
        from raytracing_amd.Scenes.scenes import build_scene
        ctx = _native.Context(0)
        ctx.set_precision("f64")                    # the reference's bytes exactly
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


# ----------------------------------------------------------------------------------------------
# FrameSet: the benchmark's way of issuing a step (graphs, one sum per level for all frames,
# tiles travelling while the next step runs)
SET_SCENES = ["example_2", "example_4", "example_3"]


def _worker_frame_set(rank, world, port, w, h, out_path, env):
    _watchdog()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracing_amd.distributed import FrameSet
        torch.cuda.set_device(0)
        frames = FrameSet(SET_SCENES, w, h, device=0, rank=rank, world=world, group=dist.group.WORLD)
        frames.prepare(seed=0)
        assert frames.pipelined == (env.get("RT_BATCH_EXCHANGE") != "host")
        assert (frames.graphs is not None) == (frames.pipelined and env.get("RT_STEP_GRAPH") != "0")
        per_step = []
        for _ in range(3):                      # both output buffers, and the first one twice
            frames.step()
            per_step.append(frames.images())
        frames.step()                           # two steps in flight without a host synchronisation between
        frames.step()
        per_step.append(frames.images())
        if frames.speculate:
            # the steps after the first went by the previous step's batch sizes and were confirmed
            assert frames.speculation == {"steps": 4, "redone": 0}, frames.speculation
            # a guess that puts nodes on the wrong side of SMALL_NUM is noticed and the step issued again
            frames.guess.zero_()
            frames.step()
            per_step.append(frames.images())
            assert frames.speculation == {"steps": 5, "redone": 1}, frames.speculation
            frames.step()
            per_step.append(frames.images())
            assert frames.speculation == {"steps": 6, "redone": 1}, frames.speculation
            # the host looks at a step's outcome only after it has queued the next one: a wrong guess is
            # still noticed, both steps end up exact, and the frame on offer is the later step's
            frames.guess.zero_()
            frames.step()
            frames.step()
            per_step.append(frames.images())
            assert frames.speculation["steps"] == 8 and frames.speculation["redone"] >= 2, frames.speculation
            assert all(job.buffer == frames.latest_buffer for job in frames.jobs)
        else:
            assert frames.speculation["steps"] == 0
        stats = [job.stats() for job in frames.jobs]
        if rank == 0:
            np.savez(out_path, images=np.array(per_step), rays=[s["rays"] for s in stats],
                     bounds=np.array([job.r.bounds for job in frames.jobs]))
        frames.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("env", [{}, {"RT_STEP_GRAPH": "0"}, {"RT_BATCH_EXCHANGE": "host"}, {"RT_SPECULATE": "0"}],
                         ids=["graphs", "no-graphs", "host-exchange", "no-speculation"])
def test_frame_set_two_ranks_on_one_gpu(tmp_path, env):
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    out = str(tmp_path / "set.npz")
    mp.spawn(_worker_frame_set, args=(2, _free_port(), 160, 90, out, env), nprocs=2, join=True)
    res = np.load(out)
    n_sets = res["images"].shape[0]
    assert res["images"].shape == (n_sets, len(SET_SCENES), 90, 160, 3) and n_sets in (4, 7)
    for f, name in enumerate(SET_SCENES):
        tol = 2e-5                                  # FrameSet runs in the default mixed precision
        assert 0 < res["bounds"][f][1] < 90
        for step in range(n_sets):
            assert np.abs(res["images"][step, f] - g[name + "_image"]).max() <= tol, (name, step)
            assert np.array_equal(res["images"][step, f], res["images"][0, f])
        r = g[name + "_rays"]
        assert int(res["rays"][f]) == int(r[r >= 3].sum())


def _worker_frame_set_full_hd(rank, world, port, out_path):
    _watchdog()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracing_amd.distributed import FrameSet
        torch.cuda.set_device(0)
        names = ["example_1", "example_2", "example_3", "example_4"]
        frames = FrameSet(names, 1920, 1080, device=0, rank=rank, world=world, group=dist.group.WORLD)
        frames.prepare(seed=0)
        assert frames.pipelined and frames.graphs is not None
        frames.step()                           # level by level, with the sums of the batch sizes between the levels
        first = frames.images()
        frames.step()                           # speculative steps, two in flight
        frames.step()
        second = frames.images()
        assert frames.speculation["steps"] == 2
        stats = [job.stats() for job in frames.jobs]
        if rank == 0:
            np.savez(out_path, images=np.array([[im[::30, ::32] for im in first], [im[::30, ::32] for im in second]]),
                     sums=[im.sum() for im in second], rays=[s["rays"] for s in stats], hits=[s["hits"] for s in stats],
                     bounds=np.array([job.r.bounds for job in frames.jobs]), redone=frames.speculation["redone"])
        frames.close()
    finally:
        dist.destroy_process_group()


def test_frame_set_two_ranks_at_1920x1080(tmp_path):
    """The benchmark's way of issuing a step (level graphs, one sum per level for all frames, speculative
    steps, double-buffered tiles) on two ranks at 1920x1080, against the pixels, sums and counts the
    reference itself produced at that size."""
    g = np.load(os.path.join(GOLDEN, "full_1080.npz"))
    assert tuple(int(v) for v in g["steps"]) == (30, 32)
    out = str(tmp_path / "set.npz")
    mp.spawn(_worker_frame_set_full_hd, args=(2, _free_port(), out), nprocs=2, join=True)
    res = np.load(out)
    assert int(res["redone"]) == 0
    for f, name in enumerate(["example_1", "example_2", "example_3", "example_4"]):
        assert 0 < res["bounds"][f][1] < 1080
        for step in range(2):
            assert np.abs(res["images"][step, f] - g[name + "_pixels"]).max() <= 2e-5, (name, step)
        assert np.array_equal(res["images"][0, f], res["images"][1, f])
        assert abs(float(res["sums"][f]) - float(g[name + "_sum"])) <= 2e-5 * 1920 * 1080 * 3 * 1e-2
        r = g[name + "_rays"]
        assert int(res["rays"][f]) == int(r[r >= 3].sum())
        assert int(res["hits"][f]) == int(g[name + "_hits"].sum())


def test_frame_set_one_gpu_graph_matches_plain_issue():
    """One GPU: the step replayed from the HIP graph gives bit for bit what the same calls give
    when issued one by one."""
    from raytracing_amd.distributed import FrameSet
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    frames = FrameSet(SET_SCENES, 160, 90)
    frames.prepare(seed=0)
    assert frames.graphs is not None
    plain = [job.image().copy() for job in frames.jobs]          # prepare() ran one plain step
    for _ in range(2):
        frames.step()
    replayed = frames.images()
    frames.close()
    for name, a, b in zip(SET_SCENES, plain, replayed):
        assert np.array_equal(a, b)
        assert np.abs(b - g[name + "_image"]).max() <= 2e-5


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher environment: the script starts both ranks (here
    on the one GPU of the box, gloo carrying the exchanges) and rank 0 reports n_gpus = 2."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RT_DIST_BACKEND="gloo", RT_BENCH_DEVICE="0", RT_BENCH_WATCHDOG="200")
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--width", "160", "--height", "90", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    want = sum(int(g[n + "_rays"][g[n + "_rays"] >= 3].sum()) for n in ("example_1", "example_2", "example_3", "example_4"))
    assert out["config"]["rays_per_step"] == want


def test_bench_one_rank_through_the_rccl_path(tmp_path):
    """The multi-rank path with its RCCL calls (device all-reduce between the level graphs, a
    communicator of their own for the tile transfers, the barrier and the maximum over ranks of
    the benchmark) on the one GPU of the box: a process group of a single rank, sent through that
    path by RT_SHARDED_PATH=1.  Counts and throughput fields must come out as on the plain path."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RT_DIST_BACKEND", "RT_BENCH_DEVICE")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), RT_SHARDED_PATH="1", RT_BENCH_WATCHDOG="200")
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--width", "160", "--height", "90", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert "one all-reduce per bounce level" in out["config"]["issue"] and "HIP graph" in out["config"]["issue"]
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    want = sum(int(g[n + "_rays"][g[n + "_rays"] >= 3].sum()) for n in ("example_1", "example_2", "example_3", "example_4"))
    assert out["config"]["rays_per_step"] == want


def _worker_one_rank_rccl(rank, world, port, w, h, out_path):
    _watchdog()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RT_SHARDED_PATH"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from raytracing_amd.distributed import FrameSet
        frames = FrameSet(SET_SCENES, w, h, device=0, rank=rank, world=world, group=dist.group.WORLD)
        frames.prepare(seed=0)
        assert frames.sharded and frames.pipelined and frames.graphs is not None and frames.speculate
        per_step = []
        for _ in range(3):
            frames.step()
            per_step.append(frames.images())
        frames.step()
        frames.step()
        per_step.append(frames.images())
        assert frames.speculation == {"steps": 4, "redone": 0}, frames.speculation
        np.savez(out_path, images=np.array(per_step))
        frames.close()
    finally:
        dist.destroy_process_group()


def test_frame_set_one_rank_over_rccl(tmp_path):
    """FrameSet's multi-rank issue order (level graphs with the batch-size sums between them queued
    on the stream by RCCL, double-buffered tiles) with device tensors and the nccl backend."""
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    out = str(tmp_path / "set.npz")
    mp.spawn(_worker_one_rank_rccl, args=(1, _free_port(), 160, 90, out), nprocs=1, join=True)
    res = np.load(out)
    for f, name in enumerate(SET_SCENES):
        for step in range(4):
            assert np.abs(res["images"][step, f] - g[name + "_image"]).max() <= 2e-5
            assert np.array_equal(res["images"][step, f], res["images"][0, f])


def test_raytrace_with_a_device_list(tmp_path):
    """raytrace(scene, devices=[...]): the call starts its own ranks (two on the one GPU of the box,
    so gloo carries the exchanges - RCCL refuses two ranks on one device) and returns the frame."""
    from raytracing_amd.launcher import raytrace_on_devices
    from raytracing_amd.raytrace import raytrace
    from raytracing_amd.Scenes.scenes import build_scene
    g = np.load(os.path.join(GOLDEN, "images_160x90.npz"))
    for name, precision, tol in (("example_3", "f64", 1e-8), ("example_4", "mixed", 2e-5)):
        np.random.seed(0)
        img = raytrace_on_devices(build_scene(name, 160, 90), [0, 0], precision=precision, dist_backend="gloo", timeout=240)
        assert np.abs(img - g[name + "_image"]).max() <= tol
        after = np.random.get_state()
        np.random.seed(0)
        one = raytrace(build_scene(name, 160, 90), devices=[0], precision=precision)
        assert np.abs(one - img).max() <= (0 if precision == "f64" else tol)
        assert after[2] == np.random.get_state()[2] and np.array_equal(after[1], np.random.get_state()[1])
