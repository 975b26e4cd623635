"""Render a scene a few times (for rocprofv3 runs): python3 tools/profile_scene.py NAME W H STEPS [PRECISION]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.raytrace import Renderer
from raytracing_amd.Scenes.scenes import build_scene
name, W, H, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
prec = sys.argv[5] if len(sys.argv) > 5 else "mixed"
np.random.seed(0)
with Renderer(build_scene(name, W, H), precision=prec) as r:
    r.begin_rows(0, H)
    r.trace(); r.draw_samples(); r.shade(); r.ctx.synchronize()
    for _ in range(steps):
        r.ctx.trace_geometry(); r.ctx.shade()
    r.ctx.synchronize()
    print(r.counters())
