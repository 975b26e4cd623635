"""CPU stand-in for the device context, for tests of the host-side logic.

TEST INFRASTRUCTURE ONLY (see ``numpy_oracle``).  ``CpuContext`` has the
methods of ``raytracing_amd._native.Context`` but evaluates the frame with the
NumPy oracle's arithmetic, organised the way the device organises it: level by
level, one queue per tree node in ascending pixel order, whole-image batch
sizes injected between levels, disk-light uniforms read from an uploaded buffer
at ``offset + s * h + j`` (angles) and ``offset + N * h + s * h + j`` (radii),
colours combined bottom-up.  It lets the multi-process row-splitting code
(``raytracing_amd.distributed``) run under ``gloo`` without a GPU, and it is a
second, independent formulation of the reference's recursion: the tests hold
it to the recursive oracle and to the reference's golden images.
"""
import numpy as np

from raytracing_amd import flatten as F

from .numpy_oracle import Oracle, _reflect


class _BufferedOracle(Oracle):
    """Oracle whose disk-light uniforms come from an uploaded buffer."""

    def __init__(self, flat, params):
        super().__init__(flat, params)
        self.block = None      # (offset, h) of the (node, light) block being evaluated, per light

    def _draw(self, light_index, count):
        off, h, N, buf = self.block[light_index]
        if count != N * h:
            raise AssertionError("sample block size mismatch")
        return buf[off:off + N * h], buf[off + N * h:off + 2 * N * h]


class CpuContext:
    def __init__(self, device=0, stream=None):
        self.orc = None
        self.tree = None
        self.samples = None
        self.sample_off = None
        self.image_out = None

    def close(self):
        pass

    # -- scene / frame ---------------------------------------------------------------------
    def set_scene(self, flat, params, tree):
        self.flat, self.params, self.tree = flat, params, tree
        self.orc = _BufferedOracle(flat, params)
        self.n_nodes, self.n_levels, self.level_start = tree.n_nodes, tree.n_levels, tree.level_start

    def begin_rows(self, row0, n_rows, width, arena_rays=0):
        Y, X = np.indices((n_rows, width), dtype=float)
        self.begin_pixels(X.reshape(-1), Y.reshape(-1) + row0, arena_rays)

    def begin_pixels(self, X, Y, arena_rays=0):
        self.X = np.asarray(X, dtype=float).reshape(-1, 1)
        self.Y = np.asarray(Y, dtype=float).reshape(-1, 1)
        self.npix = self.X.shape[0]
        self.arena_rays = int(arena_rays) if arena_rays else 24 * self.npix
        self.next_level = 0

    def set_output(self, ptr):
        raise NotImplementedError("the CPU stand-in has no device memory")

    # -- geometry --------------------------------------------------------------------------
    def _restart(self):
        K = self.n_nodes
        s, orc = self.flat, self.orc
        sx = self.X * float(s.screen_width) / s.pixels_x - float(s.screen_width) / 2
        sy = (s.pixels_y - self.Y) * float(s.screen_height) / s.pixels_y - float(s.screen_height) / 2
        pix = sx * s.screen_x + sy * s.screen_y + s.screen_position
        d = pix - s.camera_position
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        self.queue = {0: dict(o=np.full(d.shape, s.camera_position), d=d)}
        self.rec = {}
        self.n_local = np.zeros(K, dtype=np.int64)
        self.n_global = np.zeros(K, dtype=np.int64)
        self.hits = np.zeros(K, dtype=np.int64)
        self.beam = np.zeros((K, 8), dtype=np.int64)
        self.n_local[0] = self.n_global[0] = self.npix
        self.used = self.npix
        self.overflow = 0

    def trace_geometry(self):
        for level in range(self.n_levels):
            self.trace_level(level)

    def trace_level(self, level):
        if level == 0:
            self._restart()
        elif level != self.next_level:
            raise RuntimeError("levels must be traced in order")
        t, orc, p = self.tree, self.orc, self.params
        for k in range(t.level_start[level], t.level_start[level + 1]):
            q = self.queue.get(k)
            if q is None or q["o"].shape[0] == 0 or self.overflow:
                continue
            n = q["o"].shape[0]
            rec = dict(n=n, hit=np.zeros(n, dtype=bool))
            self.rec[k] = rec
            if self.n_global[k] < p.small_num:
                continue
            sid, tt = orc._nearest(q["o"], q["d"])
            hit = sid != -1
            rec["hit"] = hit
            h = int(hit.sum())
            self.hits[k] = h
            if h == 0:
                continue
            o, d, tt, sid = q["o"][hit], q["d"][hit], tt[hit][:, None], sid[hit]
            P, nrm, inside, N, Pb, trans, T, R = orc._hit_record(o, d, tt, sid)
            rec.update(sid=sid, d=d, nrm=nrm, inside=inside, N=N, Pb=Pb, trans=trans, T=T, R=R)
            for l, light in enumerate(self.flat.lights):
                if light.kind == F.LIGHT_SPOT:
                    rec.setdefault("in_beam", {})[l] = orc.beam_geometry(light, Pb)[3]
                    self.beam[k, l] = int(rec["in_beam"][l].sum())
            a, b = int(t.a[k]), int(t.b[k])
            ct, ci, cr = (int(c) for c in t.child[k])
            if b != 0:
                self._spawn(ct, Pb[trans] + 2 * p.epsilon * N[trans], T[trans])
            if b != 0 and a != 0 and (~trans).any():
                self._spawn(ci, Pb[~trans], _reflect(d[~trans], -N[~trans]))
            if a != 0:
                self._spawn(cr, Pb, _reflect(d, -N))
        # the next level's queues are complete: check the arena like the device does
        if level + 1 < self.n_levels:
            nxt = range(t.level_start[level + 1], t.level_start[level + 2])
            need = sum(-(-self.n_local[k] // 64) * 64 for k in nxt)
            if self.used + need > self.arena_rays:
                self.overflow = 1
                for k in nxt:
                    self.queue.pop(k, None)
                    self.n_local[k] = self.n_global[k] = 0
            else:
                self.used += need
        self.next_level = level + 1

    def _spawn(self, child, o, d):
        self.queue[child] = dict(o=o, d=d)
        self.n_local[child] = self.n_global[child] = o.shape[0]

    def level_batches(self, level):
        t = self.tree
        return self.n_local[t.level_start[level]:t.level_start[level + 1]].copy()

    def set_global_batches(self, level, n_global):
        t = self.tree
        self.n_global[t.level_start[level]:t.level_start[level + 1]] = np.asarray(n_global, dtype=np.int64)

    def node_counts(self):
        if self.overflow:
            from raytracing_amd._native import RtOverflow
            raise RtOverflow(-3, "ray arena too small (CPU stand-in)")
        return self.n_local.copy(), self.hits.copy(), self.beam.copy()

    def counters(self):
        small = self.params.small_num
        return dict(pixels=self.npix, rays=int(self.n_local[self.n_global >= small].sum()),
                    hits=int(self.hits.sum()), shaded_hits=0, shaded_samples=0, fallback_samples=0,
                    stored_records=0, live_hits=0,
                    arena_rays_used=self.used, arena_capacity=self.arena_rays,
                    n_nodes=self.n_nodes, n_levels=self.n_levels, overflow=self.overflow)

    # -- samples ---------------------------------------------------------------------------
    def set_sample_layout(self, offsets, total):
        self.sample_off = np.array(offsets, dtype=np.int64)
        self.samples = np.full(int(total), np.nan)

    def upload_samples(self, dst_offset, u):
        u = np.asarray(u, dtype=float).reshape(-1)
        self.samples[dst_offset:dst_offset + u.size] = u

    # -- shading ---------------------------------------------------------------------------
    def shade(self):
        t, orc, p = self.tree, self.orc, self.params
        bg = np.asarray(p.background_color, dtype=float)
        col = {}
        for level in range(self.n_levels - 1, -1, -1):
            for k in range(t.level_start[level], t.level_start[level + 1]):
                rec = self.rec.get(k)
                if rec is None:
                    continue
                out = np.zeros((rec["n"], 3)) + bg
                hit = rec["hit"]
                if hit.any():
                    h = int(hit.sum())
                    sid, trans, inside = rec["sid"], rec["trans"], rec["inside"]
                    refr, refl = np.zeros((h, 3)), np.zeros((h, 3))
                    a, b = int(t.a[k]), int(t.b[k])
                    ct, ci, cr = (int(c) for c in t.child[k])
                    if b != 0:
                        if trans.any():
                            refr[trans] = col[ct]
                        entering = trans & ~inside
                        if entering.any():
                            refr[entering] *= orc._absorption(sid[entering], rec["Pb"][entering], rec["T"][entering])
                    if b != 0 and a != 0 and (~trans).any():
                        refr[~trans] = col[ci]
                    if a != 0:
                        refl = col[cr]
                    orc.block = {}
                    for l, light in enumerate(self.flat.lights):
                        if light.draws_samples:
                            hh = int(self.beam[k, l]) if light.kind == F.LIGHT_SPOT else h
                            orc.block[l] = (int(self.sample_off[k, l]) if hh else 0, hh, light.n_points, self.samples)
                    surf, _ = orc._lighting(sid, rec["d"], rec["nrm"], rec["Pb"])
                    out[hit] = orc._combine(sid, inside, rec["R"], surf, refr, refl)
                col[k] = np.clip(out, 0, 1)
        self.image = col[0]

    def read_image(self):
        return self.image.copy()

    def read_image_rgb8(self):
        from raytracing_amd.imageio import quantise
        return quantise(self.image)

    def synchronize(self):
        pass

    def profile_enable(self, on=True):
        pass

    def profile(self, reset=False):
        return {}
