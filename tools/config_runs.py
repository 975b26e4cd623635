"""Full-size runs of the benchmark configurations other than the headline one:

    python3 tools/config_runs.py spot [MODE]     3840x2160 spotlight scene, 64 samples (soft-shadow heavy)
    python3 tools/config_runs.py many [S]        3840x2160 many-sphere scene (intersection bound), default S = 4096
    python3 tools/config_runs.py example NAME W H   an example scene at any size (e.g. example_4 7680 4320)

Prints one JSON line: rays, time per frame per kernel kind, Mrays/s, image mean."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.raytrace import Renderer
from raytracing_amd.Scenes import synthetic

W, H = 3840, 2160
kind = sys.argv[1]
if kind == "example":
    from raytracing_amd.Scenes.scenes import build_scene
    W, H = int(sys.argv[3]), int(sys.argv[4])
    scene, label = build_scene(sys.argv[2], W, H), f"{sys.argv[2]}:{W}x{H}"
elif kind == "spot":
    mode = sys.argv[2] if len(sys.argv) > 2 else "linear"
    scene, label = synthetic.spotlight_scene(mode, 64, W, H), f"spot:{mode}:64"
else:
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    scene, label = synthetic.many_spheres_scene(S, W, H), f"many:{S}"
np.random.seed(0)
t0 = time.perf_counter()
with Renderer(scene) as r:
    r.begin_rows(0, H)
    r.trace()
    t1 = time.perf_counter()
    n_samples = r.draw_samples()
    t2 = time.perf_counter()
    r.shade(); r.ctx.synchronize()
    r.ctx.profile_enable(True); r.ctx.profile(reset=True)
    steps = 2
    t3 = time.perf_counter()
    for _ in range(steps):
        r.ctx.trace_geometry(); r.ctx.shade()
    r.ctx.synchronize()
    t4 = time.perf_counter()
    prof = r.ctx.profile(reset=False)
    c = r.counters()
    img = r.image()
print(json.dumps({"config": label, "rays": int(c["rays"]), "hits": int(c["hits"]), "samples": int(n_samples),
                  "shaded_hits": int(c["shaded_hits"]), "live_hits": int(c["live_hits"]),
                  "ms_per_frame": (t4 - t3) / steps * 1e3, "Mrays_per_s": c["rays"] / ((t4 - t3) / steps) / 1e6,
                  "kernel_ms": {k: v["ms"] / steps for k, v in prof.items()},
                  "first_trace_s": t1 - t0, "draw_upload_s": t2 - t1, "arena_rays_used": int(c["arena_rays_used"]),
                  "image_mean": img.mean(axis=0).tolist()}))
