"""Rendering one frame on several GPUs of a node (one process per GPU).

The image is cut into contiguous blocks of rows, one per rank.  Pixels are
independent except for two whole-image couplings of the reference's batch
recursion, and both are exchanged here:

* the small-batch rule (reference raytrace.py:26-27) compares the size of a
  tree node's batch *over the whole image* with ``SMALL_NUM``: after every
  bounce level the ranks sum their per-node batch sizes (a few hundred
  integers) and hand the totals to the device before the next level is traced;
* disk-light random numbers (reference objects.py:132-133) are consumed in one
  global order: sample ``s`` of the ``j``-th hit (in pixel order) of a node
  reads element ``s * h + j`` of the node's block, ``h`` being the node's hit
  count over the whole image.  With contiguous row blocks the hits of rank
  ``r`` are the index range ``[base_r, base_r + h_r)`` of every node, so each
  rank draws the stream and keeps the columns that belong to its rows.

Row blocks are sized by a low-resolution probe render so that every rank gets
about the same estimated work (sky rows cost almost nothing).  The finished
tiles are sent to rank 0 with point-to-point transfers (RCCL over xGMI when
the tensors live on GPUs).

All collective traffic goes through :class:`Comm`, a thin wrapper over
``torch.distributed`` that works with the ``nccl`` (= RCCL) backend on GPUs and
with ``gloo`` on CPUs; the device work goes through a *backend* object with the
methods of :class:`raytracing_amd._native.Context`.
"""
import os

import numpy as np

from .constants import RenderParams
from .flatten import FlatScene, flatten_scene, LIGHT_SPOT
from .raytrace import legacy_mt19937_state
from .tree import build_tree, sample_offsets

PROBE_SCALE = 8       # the probe frame is 1/8 of the resolution in each direction
PROBE_BANDS = 64
SAMPLE_CHUNK = 1 << 24
# same FLOP weights as the benchmark's model
F_ISECT, F_HIT, F_LIGHT, F_SHADOW = 22, 100, 50, 24


def is_sharded(world, group):
    """Does the frame go through the multi-rank path (row blocks, per-level sums, tile transfers)?
    Yes with more than one rank; RT_SHARDED_PATH=1 also sends a process group of ONE rank through
    it, which rehearses that path (RCCL calls included) on a box with a single GPU."""
    return world > 1 or (group is not None and os.environ.get("RT_SHARDED_PATH") == "1")


# ----------------------------------------------------------------------------------------------
class Comm:
    """The handful of collectives the renderer needs, on NumPy arrays."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.tile_group = group       # group the tile transfers run on (see separate_tile_group)
        self.device = device          # torch device the backend's tensors must live on
        self.host_staged = False      # device tensors go through host copies (gloo)
        self.sharded = False          # the multi-rank path is taken (see is_sharded)
        self._side = None
        if group is None:
            self.rank, self.world = 0, 1
        else:
            import torch.distributed as dist
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            self.sharded = is_sharded(self.world, group)
            if device is not None and dist.get_backend(group) != "nccl":
                # gloo's handling of device tensors is unreliable here; small exchanges use host
                # tensors and tiles are copied through the host
                self.device = None
                self.host_staged = True

    def _t(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device) if self.device is not None else t

    def _aside(self):
        """Stream context for exchanges that start and end on the host: a stream of this object's own,
        which nobody ever captures into a graph.  (RCCL's watchdog thread polls the end event of every
        collective until it has seen it finished; HIP refuses such a query while the stream the event
        was recorded on is being captured, and the process aborts.)"""
        import contextlib
        if self.device is None:
            return contextlib.nullcontext()
        import torch
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        return torch.cuda.stream(self._side)

    def all_reduce_sum(self, a):
        if not self.sharded:
            return np.asarray(a)
        import torch.distributed as dist
        with self._aside():
            t = self._t(a)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t.cpu().numpy()

    def all_reduce_device(self, t):
        """In-place sum of a torch tensor over the ranks.  Device tensors with RCCL: queued behind
        the work of the current stream, nothing waits on the host.  Backends that cannot take
        device tensors: through a host copy (synchronises the current stream)."""
        if not self.sharded:
            return
        import torch.distributed as dist
        if t.is_cuda and self.host_staged:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def separate_tile_group(self):
        """Give the tile transfers a process group (an RCCL communicator and stream) of their
        own, so that a frame's tiles can travel while the next frame's per-level sums run.
        Collective: every rank of the group must call it."""
        if self.sharded and self.tile_group is self.group:
            import torch.distributed as dist
            ranks = dist.get_process_group_ranks(self.group)
            self.tile_group = dist.new_group(ranks=ranks, backend=dist.get_backend(self.group))
            # set the communicator up now, with every rank taking part, not at the first transfer
            import torch
            with self._aside():
                one = torch.zeros(1, dtype=torch.int32, device=self.device if self.device is not None else "cpu")
                dist.all_reduce(one, group=self.tile_group)

    def all_gather(self, a):
        """Stack every rank's array (same shape everywhere) along a new first axis."""
        a = np.asarray(a)
        if not self.sharded:
            return a[None]
        import torch
        import torch.distributed as dist
        with self._aside():
            t = self._t(a)
            out = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(out, t, group=self.group)
            return np.stack([o.cpu().numpy() for o in out])

    def barrier(self):
        if self.sharded:
            import torch.distributed as dist
            with self._aside():
                dist.barrier(group=self.group)

    def gather_tiles(self, full, tile, bounds, width):
        """Point-to-point gather of row tiles into ``full`` on rank 0.  ``full`` and
        ``tile`` are torch tensors of shape (rows * width, 3); rank 0's own tile is
        expected to be a view of ``full`` already."""
        self.gather_tiles_many([(full, tile, bounds, width)])

    def gather_tiles_many(self, frames):
        """gather_tiles for several frames, ``[(full, tile, bounds, width), ...]``, as one batch
        of transfers (one group call for all frames and ranks)."""
        if not self.sharded:
            return
        import torch
        import torch.distributed as dist
        group = self.tile_group
        staged = []
        if self.host_staged and any(t.is_cuda for _, t, _, _ in frames):
            for full, tile, bounds, width in frames:
                host_full = torch.empty(full.shape, dtype=full.dtype) if self.rank == 0 else None
                staged.append((full, host_full))
            frames = [(hf, tile.cpu(), bounds, width) for (_, tile, bounds, width), (_, hf) in zip(frames, staged)]
        ops = []
        for full, tile, bounds, width in frames:
            if self.rank == 0:
                for r in range(1, self.world):
                    lo, hi = int(bounds[r]) * width, int(bounds[r + 1]) * width
                    if hi > lo:
                        ops.append(dist.P2POp(dist.irecv, full[lo:hi], dist.get_global_rank(group, r), group=group))
            elif tile.numel():
                ops.append(dist.P2POp(dist.isend, tile, dist.get_global_rank(group, 0), group=group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged and self.rank == 0:
            for (full, host_full), (_, _, bounds, width) in zip(staged, frames):
                for r in range(1, self.world):
                    lo, hi = int(bounds[r]) * width, int(bounds[r + 1]) * width
                    full[lo:hi].copy_(host_full[lo:hi])


def init_group(n_gpus):
    """Join the process group created by ``torch.distributed.run`` (one rank per
    GPU, RCCL).  Returns ``(rank, world, group)``; a single process gets
    ``(0, 1, None)`` without touching torch.distributed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and os.environ.get("RT_SHARDED_PATH") != "1":
        return 0, 1, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        # RT_DIST_BACKEND overrides the choice (rehearsing several ranks on one GPU needs gloo:
        # RCCL refuses two ranks on the same device)
        backend = os.environ.get("RT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size(), dist.group.WORLD


# ----------------------------------------------------------------------------------------------
def frame_cost(flat, rays, hits):
    """Estimated work of a pixel set from its ray and hit counts."""
    S = flat.n_spheres
    per_hit = F_HIT + sum(l.n_points * (F_LIGHT + F_SHADOW * S) for l in flat.lights)
    return F_ISECT * S * rays + per_hit * hits


def split_rows(height, world, band_rows, band_cost):
    """Row boundaries ``(world + 1,)`` that give every rank the same share of the
    piecewise-constant cost density described by bands starting at ``band_rows``
    (plus the end row) with total cost ``band_cost`` each.  Every rank gets at
    least one row when ``height >= world``."""
    band_rows = np.asarray(band_rows, dtype=np.int64)
    cost = np.maximum(np.asarray(band_cost, dtype=np.float64), 0.0)
    n_rows = np.diff(band_rows)
    per_row = np.repeat(cost / np.maximum(n_rows, 1), n_rows) + 1e-9 * max(cost.sum(), 1.0) / height
    cum = np.concatenate([[0.0], np.cumsum(per_row)])
    targets = cum[-1] * np.arange(1, world) / world
    cuts = np.searchsorted(cum, targets - 1e-7 * cum[-1], side="left")
    bounds = np.concatenate([[0], cuts, [height]]).astype(np.int64)
    if height >= world:
        for r in range(1, world + 1):                     # at least one row each, monotone
            bounds[r] = max(bounds[r], bounds[r - 1] + 1)
        for r in range(world - 1, -1, -1):
            bounds[r] = min(bounds[r], bounds[r + 1] - 1)
    return np.maximum.accumulate(bounds)


class ShardedRenderer:
    """One rank's part of a frame that is split over ``comm.world`` ranks."""

    def __init__(self, scene, params, comm, backend, probe_backend=None):
        self.flat = scene if isinstance(scene, FlatScene) else flatten_scene(scene)
        if params is None:
            params = RenderParams(width=int(self.flat.pixels_x), height=int(self.flat.pixels_y))
        self.params = params
        self.comm = comm
        self.tree = build_tree(params.max_reflection_depth, params.max_refraction_depth)
        self.ctx = backend
        self.probe_ctx = probe_backend if probe_backend is not None else backend
        self.bounds = None
        self.arena_rays = 0
        self.rays_per_pixel = 24
        self.batch_tensor = None      # torch int32 tensor on the GPU holding whole-image batch sizes,
        self.batch_slot = 0           # possibly shared by batch_slots frames (this one is batch_slot):
        self.batch_slots = 1          # level by level, frame by frame inside a level
        self.global_hits = self.global_beam = self.global_batch = None
        self.local_hits = self.local_beam = self.local_batch = None
        self.samples_total = 0

    # ---- row split -----------------------------------------------------------------------
    def probe_costs(self):
        """Render the frame at 1/PROBE_SCALE resolution in bands of rows (geometry only)
        and return ``(band_rows_fullres, band_cost)``."""
        import copy
        p, f = self.params, self.flat
        pw, ph = max(1, p.width // PROBE_SCALE), max(1, p.height // PROBE_SCALE)
        probe_flat = copy.copy(f)
        probe_flat.pixels_x, probe_flat.pixels_y = float(pw), float(ph)
        probe_params = copy.copy(p)
        probe_params.width, probe_params.height = pw, ph
        ctx = self.probe_ctx
        ctx.set_scene(probe_flat, probe_params, self.tree)
        n_bands = min(PROBE_BANDS, ph)
        edges = np.linspace(0, ph, n_bands + 1).round().astype(np.int64)
        cost = np.zeros(n_bands)
        for i in range(n_bands):
            r0, r1 = int(edges[i]), int(edges[i + 1])
            if r1 <= r0:
                continue
            arena = 64 * (r1 - r0) * pw + 4096
            while True:
                ctx.begin_rows(r0, r1 - r0, pw, arena)
                ctx.trace_geometry()
                c = ctx.counters()
                if not c["overflow"]:
                    break
                arena *= 2
            cost[i] = frame_cost(f, c["rays"], c["hits"])
        rows = np.minimum(edges * p.height // ph, p.height)
        rows[-1] = p.height
        return rows, cost

    def choose_rows(self):
        if self.comm.world > self.params.height:
            # a rank without rows would still have to take part in every exchange with zero counts;
            # not supported: every rank raises here, before any collective, so none is left waiting
            raise ValueError(f"{self.comm.world} ranks for an image of {self.params.height} rows: "
                             "every rank needs at least one row")
        if not self.comm.sharded:
            self.bounds = np.array([0, self.params.height], dtype=np.int64)
        else:
            rows, cost = self.probe_costs()
            self.bounds = split_rows(self.params.height, self.comm.world, rows, cost)
        self.ctx.set_scene(self.flat, self.params, self.tree)
        if self.batch_tensor is not None:
            self.ctx.set_batch_buffer(self.batch_tensor.data_ptr(), self.batch_slot, self.batch_slots)
        return self.bounds

    def level_batches(self, level, every_frame=False):
        """View of the device tensor with this frame's (or every sharing frame's) whole-image
        batch sizes of one level."""
        a, b = int(self.tree.level_start[level]), int(self.tree.level_start[level + 1])
        n = self.batch_slots
        if every_frame:
            return self.batch_tensor[n * a:n * b]
        first = n * a + self.batch_slot * (b - a)
        return self.batch_tensor[first:first + (b - a)]

    @property
    def my_rows(self):
        return int(self.bounds[self.comm.rank]), int(self.bounds[self.comm.rank + 1])

    # ---- phases --------------------------------------------------------------------------
    def begin(self, output_ptr=None):
        r0, r1 = self.my_rows
        n = r1 - r0
        if n < 1:
            raise RuntimeError(f"rank {self.comm.rank} has no rows (bounds {self.bounds.tolist()}): choose_rows() gives "
                               "every rank at least one")
        if not self.arena_rays:
            self.arena_rays = self.rays_per_pixel * n * self.params.width
        self.ctx.begin_rows(r0, n, self.params.width, self.arena_rays)
        if output_ptr:
            self.ctx.set_output(output_ptr)

    def trace(self):
        """Geometry phase with the per-level exchange of batch sizes."""
        if not self.comm.sharded:
            self.ctx.trace_geometry()
            return
        for level in range(self.tree.n_levels):
            self.trace_level(level)

    def trace_level(self, level):
        """One bounce level: whole-image batch sizes of the level's nodes (sum over the
        ranks, the small-batch rule needs them), then the level's kernels."""
        ctx, comm = self.ctx, self.comm
        if level > 0 and comm.sharded:
            if self.batch_tensor is not None:
                # batch sizes stay on the GPU: the all-reduce is queued on the kernels' stream
                # between two levels and (with RCCL) nothing waits on the host
                comm.all_reduce_device(self.level_batches(level))
            else:
                ctx.set_global_batches(level, comm.all_reduce_sum(ctx.level_batches(level)))
        ctx.trace_level(level)

    def trace_checked(self, output_ptr=None):
        """trace() with arena-overflow handling: every rank doubles its arena and
        retries while any rank overflowed."""
        while True:
            self.begin(output_ptr)
            self.trace()
            from ._native import RtOverflow
            try:
                counts = self.ctx.node_counts()     # (one read-back: the counts and the overflow flag)
                overflow = 0
            except RtOverflow:
                counts, overflow = None, 1
            if int(self.comm.all_reduce_sum(np.array([overflow], dtype=np.int64))[0]) == 0:
                break
            self.arena_rays *= 2
        self.collect_counts(counts)

    def frame(self, out=None):
        """One whole frame on this GPU alone (not for sharded frames): geometry, samples, lighting, the
        image on the host (``out``: a page-locked array to read it into).  Frames without sampled
        lights are queued as a whole and the host waits once, for the image."""
        if self.comm.sharded:
            raise RuntimeError("frame() renders on one GPU; sharded frames go through raytrace_sharded()")
        if self.flat.has_sampled_lights:
            self.trace_checked()
            self.draw_samples()
            self.shade()
            return self.ctx.read_image(out=out)
        from ._native import RtOverflow
        while True:
            self.begin()
            self.trace()
            self.shade()
            try:
                img = self.ctx.read_image(out=out)     # (reports the overflow too)
                self.collect_counts(self.ctx.node_counts())
                return img
            except RtOverflow:
                self.arena_rays *= 2

    def collect_counts(self, counts=None):
        batch, hits, beam = self.ctx.node_counts() if counts is None else counts
        self.local_batch, self.local_hits, self.local_beam = batch, hits, beam
        packed = np.concatenate([batch[:, None], hits[:, None], beam], axis=1).astype(np.int64)
        every = self.comm.all_gather(packed)                      # (world, K, 2 + L)
        self.rank_hits = every[:, :, 1]
        self.rank_beam = every[:, :, 2:]
        self.global_batch = every[:, :, 0].sum(axis=0)
        self.global_hits = self.rank_hits.sum(axis=0)
        self.global_beam = self.rank_beam.sum(axis=0)

    def draw_samples(self, rng=None):
        """Draw the whole image's disk-light uniforms in the reference's order from
        ``rng`` (default: NumPy's global stream) and upload this rank's columns."""
        if not self.flat.has_sampled_lights:
            return 0
        rng = np.random if rng is None else rng
        lights, r = self.flat.lights, self.comm.rank
        g_off, g_cnt, g_total = sample_offsets(self.tree, lights, self.global_hits, self.global_beam)
        if self.comm.sharded:
            l_off, l_cnt, l_total = sample_offsets(self.tree, lights, self.local_hits, self.local_beam)
        else:
            l_off, l_cnt, l_total = g_off, g_cnt, g_total
        self.ctx.set_sample_layout(l_off, l_total)
        self.samples_total = l_total
        if not self.comm.sharded:
            state = legacy_mt19937_state(rng) if hasattr(self.ctx, "generate_samples") else None
            if state is not None:      # MT19937 runs on the GPU; the stream is advanced as host draws would
                key, pos = self.ctx.generate_samples(state[1], state[2])
                rng.set_state((state[0], key, pos) + tuple(state[3:]))
                return g_total
            for pos in range(0, g_total, SAMPLE_CHUNK):
                self.ctx.upload_samples(pos, rng.random(min(SAMPLE_CHUNK, g_total - pos)))
            return g_total
        state = legacy_mt19937_state(rng) if hasattr(self.ctx, "generate_samples") else None
        if state is not None:
            # every rank runs the generator over the whole image's stream on its GPU and keeps
            # the columns of its own hits; all ranks leave their streams in the same state
            first = np.zeros_like(g_cnt)
            for l, light in enumerate(lights):
                per_rank = self.rank_beam[:, :, l] if light.kind == LIGHT_SPOT else self.rank_hits
                first[:, l] = per_rank[:r].sum(axis=0)
            self.ctx.set_sample_layout_global(g_off, g_cnt, first, g_total)
            key, pos = self.ctx.generate_samples(state[1], state[2])
            rng.set_state((state[0], key, pos) + tuple(state[3:]))
            return g_total
        for node in self.tree.post_order():
            for l, light in enumerate(lights):
                h = int(g_cnt[node, l])
                if h == 0:
                    continue
                per_rank = self.rank_beam[:, node, l] if light.kind == LIGHT_SPOT else self.rank_hits[:, node]
                base, mine = int(per_rank[:r].sum()), int(per_rank[r])
                N = light.n_points
                rows_per_draw = max(1, SAMPLE_CHUNK // h)
                for part in range(2):                              # all angles, then all radii
                    for s0 in range(0, N, rows_per_draw):
                        s1 = min(N, s0 + rows_per_draw)
                        u = rng.random((s1 - s0) * h)
                        if mine:
                            cols = u.reshape(s1 - s0, h)[:, base:base + mine]
                            self.ctx.upload_samples(int(l_off[node, l]) + (part * N + s0) * mine, cols)
        return g_total

    def shade(self):
        self.ctx.shade()

    # ---- statistics ------------------------------------------------------------------------
    def global_stats(self):
        small = self.params.small_num
        rays = int(self.global_batch[self.global_batch >= small].sum())
        c = self.ctx.counters()
        extra = self.comm.all_reduce_sum(np.array([c["shaded_hits"], c["shaded_samples"]], dtype=np.int64))
        return {"rays": rays, "hits": int(self.global_hits.sum()), "shaded_hits": int(extra[0]),
                "shadow_rays": int(extra[1]), "spheres": self.flat.n_spheres}

    def local_stats(self):
        c = self.ctx.counters()
        out = {k: int(v) for k, v in c.items()}
        out["spheres"] = self.flat.n_spheres
        out["lights"] = [(l.kind, l.n_points) for l in self.flat.lights]
        return out


def raytrace_sharded(scene, params, comm, backend, rng=None, output="f64"):
    """Render one frame split over the ranks of ``comm``; returns the whole
    ``(height, width, 3)`` image on rank 0 and ``None`` on the other ranks.
    Every rank must call it with the same scene and, for scenes with disk lights,
    with identically seeded random streams.

    ``output="rgb8"`` gathers the tiles as the 8-bit image of the reference's output
    step (reference main.py:32): each rank quantises its own rows, 3 instead of 24
    bytes per pixel cross the links."""
    import torch
    if output not in ("f64", "rgb8"):
        raise ValueError("output must be 'f64' or 'rgb8'")
    r = ShardedRenderer(scene, params, comm, backend)
    r.choose_rows()
    r.trace_checked()
    r.draw_samples(rng)
    r.shade()
    W, H = r.params.width, r.params.height
    r0, r1 = r.my_rows
    local = backend.read_image_rgb8() if output == "rgb8" else backend.read_image()
    tile = torch.from_numpy(np.ascontiguousarray(local[:(r1 - r0) * W]))
    if comm.device is not None:
        tile = tile.to(comm.device)
    full = None
    if comm.rank == 0:
        full = torch.empty((H * W, 3), dtype=tile.dtype, device=tile.device)
        full[r0 * W:r1 * W] = tile
    comm.gather_tiles(full, tile, r.bounds, W)
    comm.barrier()
    return full.cpu().numpy().reshape(H, W, 3) if comm.rank == 0 else None


def raytrace(scene, *, params=None, device=None, output="f64"):
    """``raytrace(scene)`` for a process group started by ``torch.distributed.run``
    (one rank per GPU): rank 0 returns the image, the others ``None``.
    ``output="rgb8"``: the 8-bit image, see :func:`raytrace_sharded`."""
    import torch
    from . import _native
    rank, world, group = init_group(None)
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    comm = Comm(group, torch.device("cuda", device) if is_sharded(world, group) else None)
    ctx = _native.Context(device, torch.cuda.current_stream().cuda_stream)
    try:
        return raytrace_sharded(scene, params, comm, ctx, output=output)
    finally:
        ctx.close()


# ----------------------------------------------------------------------------------------------
class FrameJob:
    """A benchmark frame on the GPUs of this process group: set-up once, then
    ``step()`` re-renders it with the samples already resident in HBM.  Every job
    owns a HIP stream (kernels, collectives and tile transfers of the frame are
    ordered on it), so the steps of different jobs overlap unless the caller
    synchronises between them.

    ``batches=(tensor, slot, n_slots)``: keep the whole-image batch sizes in a device tensor
    shared with other frames (:class:`FrameSet`).  ``n_buffers=2``: two output tiles (and, on
    rank 0, two whole frames), so that one can travel while the next step fills the other."""

    def __init__(self, scene_name, width, height, device=0, rank=0, world=1, group=None, precision="mixed",
                 comm=None, batches=None, n_buffers=1):
        import torch
        from . import _native
        from .Scenes.scenes import build_scene
        self.torch = torch
        self.name = scene_name
        self.device = torch.device("cuda", device)
        sharded = is_sharded(world, group)
        self.comm = comm if comm is not None else Comm(group, self.device if sharded else None)
        scene = build_scene(scene_name, width, height)
        params = RenderParams(width=width, height=height)
        self.stream = torch.cuda.Stream(self.device)
        self.ctx = _native.Context(device, self.stream.cuda_stream)
        self.precision = precision
        self.r = ShardedRenderer(scene, params, self.comm, self.ctx)
        if sharded and batches is not None:
            self.r.batch_tensor, self.r.batch_slot, self.r.batch_slots = batches
        elif sharded and batch_exchange_mode(group) == "device":
            with torch.cuda.stream(self.stream):
                self.r.batch_tensor = torch.zeros(self.r.tree.n_nodes, dtype=torch.int32, device=self.device)
        self.width, self.height = width, height
        self.n_buffers = n_buffers if sharded else 1
        self.fulls, self.tiles = [], []
        self.full = self.tile = None

    def use_buffer(self, b):
        """Direct the following shading passes to output buffer ``b``."""
        self.buffer = b
        if self.tiles:
            self.full, self.tile = self.fulls[b], self.tiles[b]
            self.ctx.set_output(self.tile.data_ptr())

    def prepare(self):
        torch, r = self.torch, self.r
        with torch.cuda.stream(self.stream):
            r.choose_rows()
            r0, r1 = r.my_rows
            W = self.width
            if self.comm.sharded:
                for _ in range(self.n_buffers):
                    if self.comm.rank == 0:
                        full = torch.empty((self.height * W, 3), dtype=torch.float64, device=self.device)
                        self.fulls.append(full)
                        self.tiles.append(full[r0 * W:r1 * W])
                    else:
                        self.fulls.append(None)
                        self.tiles.append(torch.empty(((r1 - r0) * W, 3), dtype=torch.float64, device=self.device))
                self.full, self.tile = self.fulls[0], self.tiles[0]
            self.ctx.set_precision(self.precision)
            r.trace_checked(self.tile.data_ptr() if self.tile is not None else None)
            r.draw_samples()
        self.step()
        self.synchronize()
        self._stats = r.global_stats()
        self._local = r.local_stats()

    def step(self):
        r = self.r
        with self.torch.cuda.stream(self.stream):
            r.trace()
            r.shade()
            if self.comm.sharded:
                self.comm.gather_tiles(self.full, self.tile, r.bounds, self.width)

    # The same step in pieces, for callers that interleave several jobs level by level: the
    # collectives of one process group run in issue order, so with whole steps issued one job
    # after the other a job's first exchange would queue behind the previous job's last one.
    @property
    def n_levels(self):
        return self.r.tree.n_levels

    def step_level(self, level):
        with self.torch.cuda.stream(self.stream):
            self.r.trace_level(level)

    def step_finish(self):
        with self.torch.cuda.stream(self.stream):
            self.r.shade()
            if self.comm.sharded:
                self.comm.gather_tiles(self.full, self.tile, self.r.bounds, self.width)

    def synchronize(self):
        self.stream.synchronize()

    def image(self):
        """Rank 0: the whole frame as a NumPy array (other ranks: their tile)."""
        if not self.comm.sharded:
            return self.ctx.read_image().reshape(self.height, self.width, 3)
        self.synchronize()
        t = self.full if self.comm.rank == 0 else self.tile
        return t.cpu().numpy().reshape(-1, self.width, 3)

    def profile_enable(self, on=True):
        self.ctx.profile_enable(on)

    def profile(self, reset=False):
        return self.ctx.profile(reset)

    def stats(self):
        return self._stats

    def local_stats(self):
        return self._local

    def close(self):
        self.ctx.close()


def batch_exchange_mode(group):
    """Where the per-level batch sizes are summed: ``"device"`` (a device tensor, the sum queued
    on the stream) or ``"host"`` (read back, summed, written again).  RT_BATCH_EXCHANGE overrides
    the default, which is the device wherever there is a process group."""
    return os.environ.get("RT_BATCH_EXCHANGE") or ("device" if group is not None else "host")


class FrameSet:
    """The frames of one benchmark step, rendered together on this process group.

    One GPU: the step (every kernel of every frame, each frame on its own stream) is captured
    once into a HIP graph and replayed, one launch per step.

    Several GPUs: the kernels of one bounce level of all frames are one graph (frames side by
    side on their streams); between two levels ONE all-reduce sums the whole-image batch sizes
    of all frames (they share a device tensor laid out level by level).  Shading writes the
    output tiles alternately into two buffers: the tiles of step k travel to rank 0 on a stream
    and a process group of their own while step k + 1 is computed; a buffer is reused only
    after its transfer has finished.

    Steps after the first speculate (RT_SPECULATE=0: never): the whole-image batch sizes matter
    only through the small-batch rule, i.e. through which nodes fall below SMALL_NUM, and for
    frames that resemble the previous step's those are the same nodes.  The whole step (all levels
    and the lighting of all frames) is then ONE graph that goes by the previous step's sizes
    (rt_set_batch_guess), ONE all-reduce behind it sums the sizes it produced, and if every node
    lies on the same side of SMALL_NUM as guessed, the step is exactly what the level-by-level
    exchange would have given.  The host learns the outcome one step late - when it comes back for
    the next step it queues that step first and then looks at the previous one's outcome - or when it
    synchronises; a step whose guess was wrong is then issued again level by level.

    RT_STEP_GRAPH=0 issues the same work without graphs.  With RT_BATCH_EXCHANGE=host the frames
    fall back to per-frame host-side exchanges, issued level by level."""

    def __init__(self, scene_names, width, height, device=0, rank=0, world=1, group=None, precision="mixed"):
        import torch
        self.torch = torch
        self.world, self.rank = world, rank
        self.sharded = is_sharded(world, group)
        self.device = torch.device("cuda", device)
        self.comm = Comm(group, self.device if self.sharded else None)
        self.use_graphs = os.environ.get("RT_STEP_GRAPH", "1") != "0"
        self.pipelined = self.sharded and batch_exchange_mode(group) == "device"
        self.stream = torch.cuda.Stream(self.device)          # levels are joined here, collectives queue here
        self.tile_stream = torch.cuda.Stream(self.device)     # tile transfers
        F = len(scene_names)
        self.batches = None
        shared = [None] * F
        if self.pipelined:
            from .tree import build_tree
            p = RenderParams(width=width, height=height)
            K = build_tree(p.max_reflection_depth, p.max_refraction_depth).n_nodes
            with torch.cuda.stream(self.stream):
                self.batches = torch.zeros(F * K, dtype=torch.int32, device=self.device)
                self.guess_buffer = torch.zeros(F * K, dtype=torch.int32, device=self.device)
            self.stream.synchronize()
            shared = [(self.batches, f, F) for f in range(F)]
            self.comm.separate_tile_group()
        self.jobs = [FrameJob(name, width, height, device=device, rank=rank, world=world, group=group,
                              precision=precision, comm=self.comm, batches=shared[f],
                              n_buffers=2 if self.pipelined else 1)
                     for f, name in enumerate(scene_names)]
        self.graphs = None
        # one GPU: frames as independent pipelines (RT_FRAME_PIPELINES=0: one graph per step that joins the frames)
        self.frame_pipelines = os.environ.get("RT_FRAME_PIPELINES", "1") != "0"
        self.steps_issued = 0
        self.tiles_sent = [None, None]        # event per output buffer: its last transfer has finished
        self.speculate = self.pipelined and os.environ.get("RT_SPECULATE", "1") != "0"
        self.guess = getattr(self, "guess_buffer", None) if self.speculate else None   # the last step's whole-image batch sizes
        self.guess_valid = False
        self.pending = None                   # (event, output buffer) of a speculative step whose outcome is not known yet
        self.wrong_host = torch.zeros(2, dtype=torch.bool).pin_memory() if self.speculate else None   # per output buffer
        self.latest_buffer = 0
        self.speculation = {"steps": 0, "redone": 0}
        self.small_num = int(RenderParams(width=width, height=height).small_num)

    def prepare(self, seed=0):
        """Set every frame up (row split, first geometry pass, samples resident in HBM); every
        frame starts from NumPy's global stream seeded with ``seed``, as separate runs of the
        reference would."""
        for job in self.jobs:
            np.random.seed(seed)
            job.prepare()
        self.synchronize()
        if self.use_graphs and (not self.sharded or self.pipelined):
            self.capture()

    # ---- issuing work -------------------------------------------------------------------------
    def _side_by_side(self, body):
        """Run ``body(job)`` for every frame on the frame's stream, forked from and joined into
        the set's stream (also the way the frames enter a graph under capture)."""
        torch = self.torch
        fork = torch.cuda.Event()
        fork.record(self.stream)
        for job in self.jobs:
            job.stream.wait_event(fork)
            body(job)
            done = torch.cuda.Event()
            done.record(job.stream)
            self.stream.wait_event(done)

    def _captured(self, body):
        torch = self.torch
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.stream):
            # "relaxed": calls other threads make while this one captures must not end the capture -
            # RCCL's watchdog thread polls the events of finished collectives at any time
            g.capture_begin(capture_error_mode="relaxed")
            try:
                self._side_by_side(body)
            finally:
                g.capture_end()
        return g

    @staticmethod
    def _level(level):
        return lambda job: job.ctx.trace_level(level)

    @staticmethod
    def _shade(buffer):
        def body(job):
            job.use_buffer(buffer)
            job.ctx.shade()
        return body

    @staticmethod
    def _whole(job):
        job.ctx.trace_geometry()
        job.ctx.shade()

    def _speculative(self, buffer):
        def body(job):
            # the kernels of these launches (and of the graph captured from them) read the guess
            job.ctx.set_batch_guess(self.guess.data_ptr())
            job.use_buffer(buffer)
            job.ctx.trace_geometry()
            job.ctx.shade()
            job.ctx.set_batch_guess(None)
        return body

    def _captured_frame(self, job):
        """One frame's whole step as a graph of its own, captured on (and replayed into) the frame's stream."""
        torch = self.torch
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(job.stream):
            g.capture_begin(capture_error_mode="relaxed")
            try:
                self._whole(job)
            finally:
                g.capture_end()
        return g

    def capture(self):
        if self.sharded:
            # The collectives queued on these streams so far (the first geometry pass's sums) have
            # finished, but RCCL's watchdog thread may not have looked at their end events yet - it
            # does so every 100 ms - and HIP fails that look (and torch aborts the process) while the
            # stream such an event was recorded on is capturing.  Give the watchdog time for its round.
            import time
            self.synchronize()
            time.sleep(0.35)
        try:
            if not self.sharded:
                if self.frame_pipelines:
                    self.graphs = {"frames": [self._captured_frame(job) for job in self.jobs]}
                else:
                    self.graphs = {"step": self._captured(self._whole)}
            else:
                self.graphs = {"level": [self._captured(self._level(l)) for l in range(self.jobs[0].n_levels)],
                               "shade": [self._captured(self._shade(b)) for b in range(2)]}
                if self.speculate:
                    self.graphs["speculative"] = [self._captured(self._speculative(b)) for b in range(2)]
        except RuntimeError as exc:
            # the same work is then issued kernel by kernel (same order of collectives, so ranks
            # may differ in this); say so, the timing is affected
            import sys
            print(f"[raytracing_amd] HIP graph capture failed, issuing kernels directly: {exc}", file=sys.stderr)
            self.graphs = None
        self.synchronize()

    def _run(self, kind, index, body):
        if self.graphs is not None:
            g = self.graphs[kind]
            (g if index is None else g[index]).replay()
        else:
            self._side_by_side(body)

    def step(self):
        torch = self.torch
        if self.sharded and not self.pipelined:
            for level in range(self.jobs[0].n_levels):
                for job in self.jobs:
                    job.step_level(level)
            for job in self.jobs:
                job.step_finish()
            return
        if not self.sharded and self.frame_pipelines:
            # every frame is a pipeline of its own: its step k + 1 follows its step k on the frame's
            # stream and waits for no other frame, so a frame with a short chain of kernels does not
            # idle until the longest one has finished the step
            for f, job in enumerate(self.jobs):
                with torch.cuda.stream(job.stream):
                    if self.graphs is not None:
                        self.graphs["frames"][f].replay()
                    else:
                        self._whole(job)
            return
        if not self.sharded:
            with torch.cuda.stream(self.stream):
                self._run("step", None, self._whole)
            return
        b = self.steps_issued & 1
        self.steps_issued += 1
        self.latest_buffer = b
        if self.speculate and self.guess_valid:
            # Geometry of all levels by the last step's sizes and the lighting right behind it, every
            # frame on its own stream as on one GPU; then one sum of the batch sizes and a look at
            # the outcome.  Nothing on the device waits for the collective, and the host takes the
            # outcome of the step BEFORE this one only now, with this step already queued behind it:
            # the GPU never waits for the host to come back.  (A step that passes its own look is exact
            # whatever its guess was - the sizes of a level follow from the decisions of the levels
            # above it, so sizes that confirm the decisions they were made with are the exact ones - and
            # this step's guess, copied on the stream, is what the last step really summed.)
            with torch.cuda.stream(self.stream):
                if self.tiles_sent[b] is not None:
                    self.stream.wait_event(self.tiles_sent[b])
                self._run("speculative", b, self._speculative(b))
                shaded = torch.cuda.Event()
                shaded.record(self.stream)
                self.comm.all_reduce_device(self.batches)
                wrong = ((self.batches < self.small_num) != (self.guess < self.small_num)).any()
                self.wrong_host[b].copy_(wrong, non_blocking=True)
                looked = torch.cuda.Event()
                looked.record(self.stream)
                self.guess.copy_(self.batches)
                self.speculation["steps"] += 1
            self._send(b, shaded)
            self._settle()
            self.pending = (looked, b)
        else:
            self._settle()
            with torch.cuda.stream(self.stream):
                self._exact_levels()
            self._finish(b)

    def _exact_levels(self):
        """Geometry of all frames level by level, one sum of the batch sizes per level (on the set's stream)."""
        first = self.jobs[0].r
        for level in range(first.tree.n_levels):
            if level > 0:
                self.comm.all_reduce_device(first.level_batches(level, every_frame=True))
            self._run("level", level, self._level(level))
        if self.speculate:
            self.guess.copy_(self.batches)
            self.guess_valid = True

    def _finish(self, b):
        """Lighting and colours into output buffer ``b``, then its tiles on their way to rank 0."""
        torch = self.torch
        with torch.cuda.stream(self.stream):
            if self.tiles_sent[b] is not None:
                self.stream.wait_event(self.tiles_sent[b])
            self._run("shade", b, self._shade(b))
            shaded = torch.cuda.Event()
            shaded.record(self.stream)
        self._send(b, shaded)

    def _send(self, b, shaded):
        """The tiles of output buffer ``b`` travel to rank 0 on the tile stream once ``shaded`` has happened."""
        torch = self.torch
        with torch.cuda.stream(self.tile_stream):
            self.tile_stream.wait_event(shaded)
            self.comm.gather_tiles_many([(j.fulls[b], j.tiles[b], j.r.bounds, j.width) for j in self.jobs])
            sent = torch.cuda.Event()
            sent.record(self.tile_stream)
            self.tiles_sent[b] = sent
        for job in self.jobs:                      # what image() reads and a later FrameJob.step() fills
            job.use_buffer(b)

    def _settle(self):
        """The outcome of the last speculative step: if some node fell on the other side of SMALL_NUM
        than guessed, that step is issued again level by level, into the same output buffer (every
        rank sees the same sums, so every rank does)."""
        if self.pending is None:
            return
        looked, b = self.pending
        self.pending = None
        looked.synchronize()
        if bool(self.wrong_host[b].item()):
            self.speculation["redone"] += 1
            with self.torch.cuda.stream(self.stream):
                self._exact_levels()
            self._finish(b)
            for job in self.jobs:                  # a later step may be queued already: its buffer is the current one
                job.use_buffer(self.latest_buffer)

    def synchronize(self):
        if self.sharded and self.pipelined:
            self._settle()
        for job in self.jobs:
            job.synchronize()
        self.stream.synchronize()
        self.tile_stream.synchronize()

    def images(self):
        self.synchronize()
        return [job.image() for job in self.jobs]

    def close(self):
        self.synchronize()
        self.graphs = None
        for job in self.jobs:
            job.close()
