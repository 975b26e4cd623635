"""Per-level ray counts of one frame, then a few repeated frames for a profiler to time.

    rocprofv3 --kernel-trace --output-format csv -d OUT -o lp -- python3 tools/level_profile.py NAME W H
    python3 tools/level_profile.py --join OUT/lp_kernel_trace.csv LEVELS.json

The first form writes gpurun_out/levels_NAME.json (rays, hits per level); the second prints, per level,
the duration of every kernel of the last repeated frame next to those counts."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(name, W, H):
    from raytracing_amd.raytrace import Renderer
    from raytracing_amd.Scenes.scenes import build_scene
    np.random.seed(0)
    with Renderer(build_scene(name, W, H)) as r:
        r.begin_rows(0, H)
        r.trace(); r.draw_samples(); r.shade(); r.ctx.synchronize()
        t = r.tree
        levels = []
        for l in range(len(t.level_start) - 1):
            ks = range(t.level_start[l], t.level_start[l + 1])
            small = r.params.small_num
            levels.append({"level": l, "nodes": len(ks),
                           "rays": int(sum(r.node_batch[k] for k in ks)),
                           "traced": int(sum(r.node_batch[k] for k in ks if r.node_batch[k] >= small)),
                           "hits": int(sum(r.node_hits[k] for k in ks)),
                           "nonempty_nodes": int(sum(1 for k in ks if r.node_batch[k] > 0))})
        for _ in range(3):
            r.ctx.trace_geometry(); r.ctx.shade()
        r.ctx.synchronize()
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump({"levels": levels, "counters": r.counters()}, open(f"gpurun_out/levels_{name}.json", "w"), indent=1)
        print(json.dumps(levels))


def join(trace_csv, levels_json):
    import csv
    rows = list(csv.DictReader(open(trace_csv)))
    lv = json.load(open(levels_json))["levels"]
    n = len(lv)
    out = {}
    for kind in ("k_trace", "k_scan", "k_link", "k_combine"):
        x = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if kind in r["Kernel_Name"]][-n:]
        out[kind] = x[::-1] if kind == "k_combine" else x
    print(f"{'lvl':>3} {'nodes':>5} {'rays':>10} {'hits':>10} {'trace us':>9} {'ns/ray':>7} {'link us':>8} {'comb us':>8} {'scan us':>8}")
    for i, L in enumerate(lv):
        tr = out["k_trace"][i] / 1e3
        print(f"{i:3d} {L['nonempty_nodes']:5d} {L['rays']:10d} {L['hits']:10d} {tr:9.1f} {1e3 * tr / max(L['rays'], 1):7.4f} "
              f"{out['k_link'][i] / 1e3:8.1f} {out['k_combine'][i] / 1e3:8.1f} {out['k_scan'][i] / 1e3:8.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "--join":
        join(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
