"""``raytrace(scene, devices=[...])``: one call that renders a frame on several GPUs.

The reference's entry point is a plain function call (reference raytrace.py:273-279).  Rendering
on several GPUs needs one process per GPU (one RCCL rank each), so the call starts them itself:
fresh interpreter processes (never forks of the caller, which may already hold a GPU), one per
entry of ``devices``, joined into a ``torch.distributed`` process group over the loopback
interface.  Every rank renders its block of rows (:mod:`raytracing_amd.distributed`), rank 0
gathers the tiles and writes the frame where the caller picks it up.

The caller's NumPy random stream is part of the call's meaning: disk-light samples are taken from
``numpy.random`` exactly as the reference takes them (reference objects.py:132-133).  The stream's
state travels to the ranks with the job, and the state the draws leave behind is set back in the
caller, so the stream continues as it would after the reference's own call.

The job (flattened scene, parameters, stream state) and the result travel through a private
temporary directory as files this module writes itself.
"""
import os
import pickle
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

from .constants import RenderParams
from .flatten import FlatScene, flatten_scene


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def raytrace_on_devices(scene, devices, *, params=None, precision="mixed", output="f64", backend=None,
                        dist_backend=None, timeout=600.0):
    """Render ``scene`` split over ``devices`` (GPU indices, one rank each); returns the
    ``(height, width, 3)`` image (float64, or uint8 with ``output="rgb8"``).

    ``dist_backend``: ``"nccl"`` (RCCL, the default on GPUs) or ``"gloo"`` (needed when two ranks
    share one GPU, which RCCL refuses).  ``backend`` is a test seam: ``"module:factory"`` of an object
    with the methods of :class:`raytracing_amd._native.Context` that stands in for the device;
    the default is the HIP library on the rank's GPU."""
    devices = [int(d) for d in devices]
    if not devices:
        raise ValueError("devices must name at least one GPU")
    if output not in ("f64", "rgb8"):
        raise ValueError("output must be 'f64' or 'rgb8'")
    flat = scene if isinstance(scene, FlatScene) else flatten_scene(scene)
    if params is None:
        params = RenderParams(width=int(flat.pixels_x), height=int(flat.pixels_y))
    if len(devices) > params.height:
        raise ValueError(f"{len(devices)} ranks for an image of {params.height} rows: every rank needs at least one row")
    work = tempfile.mkdtemp(prefix="raytracing_amd_")
    try:
        job = {"flat": flat, "params": params, "precision": precision, "output": output, "backend": backend,
               "dist_backend": dist_backend or ("gloo" if backend else "nccl"), "rng_state": np.random.get_state()}
        with open(os.path.join(work, "job.pkl"), "wb") as f:
            pickle.dump(job, f)
        port = _free_port()
        repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        procs, logs = [], []
        for rank, device in enumerate(devices):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(device), WORLD_SIZE=str(len(devices)),
                       LOCAL_WORLD_SIZE=str(len(devices)), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env["PYTHONPATH"] = repo_root + os.pathsep + env.get("PYTHONPATH", "")
            log = open(os.path.join(work, f"rank{rank}.log"), "wb")
            logs.append(log)
            procs.append(subprocess.Popen([sys.executable, "-m", "raytracing_amd.launcher", work],
                                          env=env, stdout=log, stderr=subprocess.STDOUT))
        failed = _wait(procs, timeout)
        for log in logs:
            log.close()
        if failed is not None:
            rank, code = failed
            with open(os.path.join(work, f"rank{rank}.log"), "rb") as f:
                tail = f.read()[-3000:].decode("utf-8", "replace")
            raise RuntimeError(f"raytrace on devices {devices}: rank {rank} ended with code {code}\n{tail}")
        image = np.load(os.path.join(work, "image.npy"))
        with open(os.path.join(work, "rng_state.pkl"), "rb") as f:
            np.random.set_state(pickle.load(f))
        return image
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _wait(procs, timeout):
    """Wait for all ranks; if one fails (or time runs out) end the others, which would wait for it
    in a collective for ever.  Returns ``(rank, exit code)`` of the first failure, or None."""
    deadline = time.monotonic() + timeout
    pending = set(range(len(procs)))
    failed = None
    while pending and failed is None:
        for rank in sorted(pending):
            code = procs[rank].poll()
            if code is None:
                continue
            pending.discard(rank)
            if code != 0:
                failed = (rank, code)
                break
        if failed is None and pending:
            if time.monotonic() > deadline:
                failed = (min(pending), "timeout")
                break
            time.sleep(0.02)
    if failed is not None:
        for rank in pending:
            procs[rank].terminate()
        for rank in pending:
            try:
                procs[rank].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[rank].kill()
    return failed


def _rank_main(work):
    """Body of one rank (a fresh process started by raytrace_on_devices)."""
    import importlib
    import torch
    import torch.distributed as dist
    from .distributed import Comm, raytrace_sharded
    with open(os.path.join(work, "job.pkl"), "rb") as f:
        job = pickle.load(f)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ["LOCAL_RANK"])
    on_gpu = job["backend"] is None
    if on_gpu:
        torch.cuda.set_device(device)
    dist.init_process_group(job["dist_backend"], rank=rank, world_size=world)
    try:
        if on_gpu:
            from . import _native
            ctx = _native.Context(device, torch.cuda.current_stream().cuda_stream)
            comm = Comm(dist.group.WORLD, torch.device("cuda", device) if world > 1 else None)
        else:
            module, _, factory = job["backend"].partition(":")
            ctx = getattr(importlib.import_module(module), factory)()
            comm = Comm(dist.group.WORLD)
        if hasattr(ctx, "set_precision"):
            ctx.set_precision(job["precision"])
        np.random.set_state(job["rng_state"])
        image = raytrace_sharded(job["flat"], job["params"], comm, ctx, output=job["output"])
        if rank == 0:
            np.save(os.path.join(work, "image.npy"), image)
            with open(os.path.join(work, "rng_state.pkl"), "wb") as f:
                pickle.dump(np.random.get_state(), f)
        ctx.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    _rank_main(sys.argv[1])
