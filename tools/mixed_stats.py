"""Debug: waves of k_surface_mixed by number of candidate spheres (libraries built with -DRT_MIXED_STATS=k as ab/stK.so).
python3 tools/mixed_stats.py W H SCENE"""
import os, subprocess, sys
W, H, name = sys.argv[1], sys.argv[2], sys.argv[3]
code = f"""
import numpy as np, sys
sys.path.insert(0, '.')
from raytracing_amd.raytrace import Renderer
from raytracing_amd.Scenes.scenes import build_scene
np.random.seed(0)
with Renderer(build_scene('{name}', {W}, {H})) as r:
    r.begin_rows(0, {H}); r.trace(); r.draw_samples(); r.shade(); r.ctx.synchronize()
    c = r.counters()
    print(c['certify_mismatches'], c['shaded_hits'], c['fallback_samples'])
"""
for k in range(4):
    env = dict(os.environ, RT_HIP_LIB=os.path.abspath(f"ab/st{k}.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("T =", k, out.stdout.strip(), out.stderr.strip()[-300:], flush=True)
