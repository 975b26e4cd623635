"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into bytes per launch per kernel.

    python3 tools/pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json

Corrections (calibrated with tools/pmc_calib, which streams 1 GiB with 4, 8 and 16 B per lane): both
counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes read (524298 KiB for 1 GiB at every
width) and is doubled; WRITE_SIZE is exact (1048576 KiB)."""
import collections, csv, glob, json, sys


def load(path, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")
                acc[name].append(float(r["Counter_Value"]))
    return acc


def main(fetch_dir, write_dir, out_path):
    f, w = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --output-format csv -- python3 bench.py "
                      "--steps 1 --warmup 0 --no-cpu-baseline --frames sequential",
           "workload": {"scenes": ["example_1", "example_2", "example_3", "example_4"], "width": 3840, "height": 2160,
                        "precision": "mixed", "n_gpus": 1},
           "unit": "bytes per launch, mean over all launches of the kernel in the run",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 reports half), WRITE_SIZE x1; see tools/pmc_calib.hip",
           "kernels": {}}
    for k in sorted(f):
        if k.startswith("k_") and k in w:
            fb, wb = 2 * 1024 * sum(f[k]) / len(f[k]), 1024 * sum(w[k]) / len(w[k])
            out["kernels"][k] = {"fetch": fb, "write": wb, "total": fb + wb, "launches": len(f[k])}
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
