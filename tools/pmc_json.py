"""Turn rocprofv3 --pmc passes of bench.py into per-kernel means per launch, as JSON that bench.py reads back
(roofline.traffic, executed FLOPs).

    python3 tools/pmc_json.py traffic BENCH_LINE.json OUT.json FETCH_DIR WRITE_DIR
    python3 tools/pmc_json.py flops   BENCH_LINE.json OUT.json FLOPS_DIR [SQ_DIR]

BENCH_LINE.json: the JSON line bench.py printed in one of the passes (names the workload).
traffic: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes read (calibrated
with tools/pmc_calib: 524298 KiB for 1 GiB at 4, 8 and 16 B per lane) and is doubled; WRITE_SIZE is exact.
flops: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_{F32,F64} count wave instructions; an instruction is 64 lanes wide
whatever the execution mask, an FMA is two operations, a packed float32 instruction is counted by the hardware
as the two instructions it stands for.  lanes_per_valu = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (SQ_DIR)."""
import collections, csv, glob, json, sys


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main(kind, bench_json, out_path, *dirs):
    line = json.load(open(bench_json))
    cfg = line["config"]
    out = {"workload": {"scenes": cfg["scenes"], "width": cfg["width"], "height": cfg["height"],
                        "precision": "mixed" if "f32" in cfg["precision"] or "mixed" in cfg["precision"] else "f64", "n_gpus": line["n_gpus"]},
           "unit": "per launch, mean over all launches of the kernel in the pass", "kernels": {}}
    if kind == "traffic":
        f, w = load(dirs[0]), load(dirs[1])
        out["corrections"] = "KiB -> bytes; FETCH_SIZE x2 (gfx950 reports half), WRITE_SIZE x1; see tools/pmc_calib.hip"
        for k in sorted(f):
            if k.startswith("k_") and k in w:
                fb, wb = 2 * 1024 * mean(f[k]["FETCH_SIZE"]), 1024 * mean(w[k]["WRITE_SIZE"])
                out["kernels"][k] = {"fetch": fb, "write": wb, "total": fb + wb, "launches": len(f[k]["FETCH_SIZE"])}
    else:
        c = load(dirs[0])
        d = load(dirs[1]) if len(dirs) > 1 else {}
        for k in sorted(c):
            if not k.startswith("k_"):
                continue
            g = lambda n: mean(c[k].get(n, []))
            f64 = 64 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64") + 2 * g("SQ_INSTS_VALU_FMA_F64"))
            f32 = 64 * (g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_TRANS_F32") + 2 * g("SQ_INSTS_VALU_FMA_F32"))
            rec = {"flop_f32": f32, "flop_f64": f64, "trans_f32": 64 * g("SQ_INSTS_VALU_TRANS_F32"), "trans_f64": 64 * g("SQ_INSTS_VALU_TRANS_F64"),
                   "launches": len(c[k].get("SQ_INSTS_VALU_FMA_F64", []))}
            if k in d:
                h = lambda n: mean(d[k].get(n, []))
                if h("SQ_ACTIVE_INST_VALU"):
                    rec["lanes_per_valu"] = h("SQ_THREAD_CYCLES_VALU") / h("SQ_ACTIVE_INST_VALU")
                if h("SQ_WAVE_CYCLES"):
                    # share of a wave's resident time in which it issues VALU work / waits at an instruction
                    rec["valu_share_of_wave_time"] = h("SQ_ACTIVE_INST_VALU") / h("SQ_WAVE_CYCLES")
                    rec["wait_share_of_wave_time"] = h("SQ_WAIT_INST_ANY") / h("SQ_WAVE_CYCLES") if h("SQ_WAIT_INST_ANY") else None
                rec["waves"] = h("SQ_WAVES")
                rec["valu_instructions_per_wave"] = h("SQ_INSTS_VALU") / h("SQ_WAVES") if h("SQ_WAVES") else None
            out["kernels"][k] = rec
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
