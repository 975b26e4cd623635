#!/bin/bash
# where the waves of the sample set-up kernels spend their time (one counter pass over a 4K example_4 frame)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_gen; rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/a -o c -- python3 bench.py --scenes example_4 --no-cpu-baseline --steps 1 --warmup 0 --frames sequential > $O/a.json 2> $O/a.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -o c -- python3 bench.py --scenes example_4 --no-cpu-baseline --steps 1 --warmup 0 --frames sequential > $O/b.json 2> $O/b.err || exit 1
python3 - $O <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    if not k.startswith("k_mt") and not k.startswith("k_pack") and not k.startswith("k_store"): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, {n: round(v / m.get("SQ_WAVE_CYCLES", 1), 4) if n.startswith(("SQ_ACTIVE", "SQ_WAIT")) else round(v) for n, v in sorted(m.items())})
PY
rm -rf $O/a $O/b
