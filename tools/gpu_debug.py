"""Ad-hoc GPU diagnostics: compare device renders with goldens / oracle and print summaries."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.raytrace import Renderer
from raytracing_amd.Scenes.scenes import build_scene

G = np.load("tests/golden/images_160x90.npz")
sym = "tir"
PREC = os.environ.get("RT_PRECISION", "mixed")
for name in sys.argv[1:] or ["example_1", "example_2", "example_3", "example_4"]:
    np.random.seed(0)
    with Renderer(build_scene(name, 160, 90), precision=PREC) as r:
        r.begin_rows(0, 90)
        img = r.render_pixels().reshape(90, 160, 3)
        tree = r.tree
        paths = [""] * tree.n_nodes
        for k in range(1, tree.n_nodes):
            paths[k] = paths[tree.parent[k]] + sym[tree.branch[k]]
        by = {p: i for i, p in enumerate(G[name + "_path"])}
        bad = 0
        for k, p in enumerate(paths):
            i = by.get(p)
            wn = int(G[name + "_rays"][i]) if i is not None else 0
            wh = int(G[name + "_hits"][i]) if i is not None else 0
            if r.node_batch[k] != wn or r.node_hits[k] != wh:
                bad += 1
                if bad < 8:
                    print("  node", repr(p), "batch", r.node_batch[k], wn, "hits", r.node_hits[k], wh)
        diff = np.abs(img - G[name + "_image"]).max(axis=-1)
        ys, xs = np.nonzero(diff > 1e-8)
        print(name, "maxdiff", diff.max(), "bad pixels", len(ys), "bad nodes", bad, "counters", r.counters())
        for y, x in list(zip(ys, xs))[:10]:
            print("   px", y, x, img[y, x], G[name + "_image"][y, x])
