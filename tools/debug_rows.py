"""One process renders a row block with whole-image batch sizes injected between levels from a torch
tensor (what the all-reduce would deliver), many times, without host synchronisation in between."""
import faulthandler, os, sys
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import _native
from raytracing_amd.raytrace import Renderer
from raytracing_amd.Scenes.scenes import build_scene
name, W, H, r0, r1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
scene = build_scene(name, W, H)
with Renderer(scene) as full:
    full.begin_rows(0, H); full.trace()
    glob = torch.tensor(np.asarray(full.node_batch), dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    r = Renderer(scene, stream=stream.cuda_stream)
    buf = torch.zeros(r.tree.n_nodes, dtype=torch.int32, device="cuda")
    r.ctx.set_batch_buffer(buf.data_ptr())
    r.begin_rows(r0, r1 - r0)
    ls = [int(x) for x in r.tree.level_start]
    for rep in range(int(os.environ.get("REPS", "200"))):
        for level in range(r.tree.n_levels):
            if level > 0:
                buf[ls[level]:ls[level + 1]] = glob[ls[level]:ls[level + 1]]
            r.ctx.trace_level(level)
        r.ctx.shade()
        if rep % 20 == 0:
            stream.synchronize()
            print("rep", rep, "ok", flush=True)
    stream.synchronize()
print("done", flush=True)
