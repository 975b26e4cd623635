"""Render a scene a few times (for rocprofv3 runs): python3 tools/profile_scene.py NAME W H STEPS [PRECISION]
NAME: example_1..4 or many:S (the synthetic S-sphere scene)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.raytrace import Renderer
from raytracing_amd.Scenes.scenes import build_scene
name, W, H, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
prec = sys.argv[5] if len(sys.argv) > 5 else "mixed"
np.random.seed(0)
if name.startswith("many:"):
    from raytracing_amd.Scenes import synthetic
    scene = synthetic.many_spheres_scene(int(name.split(":")[1]), W, H)
else:
    scene = build_scene(name, W, H)
with Renderer(scene, precision=prec) as r:
    r.begin_rows(0, H)
    r.trace(); r.draw_samples(); r.shade(); r.ctx.synchronize()
    for _ in range(steps):
        r.ctx.trace_geometry(); r.ctx.shade()
    r.ctx.synchronize()
    print(r.counters())
