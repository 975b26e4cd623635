"""Per-scene kernel times and counters at a given size (GPU)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.raytrace import Renderer
from raytracing_amd.Scenes.scenes import build_scene

W, H = int(sys.argv[1]), int(sys.argv[2])
for name in sys.argv[3:] or ["example_1", "example_2", "example_3", "example_4"]:
    np.random.seed(0)
    with Renderer(build_scene(name, W, H), precision=os.environ.get("RT_PRECISION", "mixed")) as r:
        r.begin_rows(0, H)
        t0 = time.perf_counter(); r.trace(); t1 = time.perf_counter(); n = r.draw_samples(); t2 = time.perf_counter()
        r.shade(); r.ctx.synchronize()
        r.ctx.profile_enable(True)
        for _ in range(3):
            r.ctx.trace_geometry(); r.ctx.shade()
        r.ctx.synchronize()
        p = r.ctx.profile()
        c = r.counters()
        print(name, f"{W}x{H}", "rays", c["rays"], "hits", c["hits"], "shaded_hits", c["shaded_hits"], "samples", c["shaded_samples"], "fallback", c["fallback_samples"],
              "uniforms", n, f"draw {t2-t1:.1f}s", {k: round(v["ms"]/3, 3) for k, v in p.items()}, flush=True)
