#!/bin/bash
# share of the waves' time by instruction kind, every kernel of a configuration: bash tools/pmc_scalar.sh "ARGS OF bench.py" OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${2:-pmc_scalar}; rm -rf $O; mkdir -p $O
timeout -s KILL 250 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $O/a -o c -- python3 bench.py $1 --no-cpu-baseline --steps 1 --warmup 0 --frames sequential > $O/a.json 2> $O/a.err || exit 1
timeout -s KILL 250 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d $O/b -o c -- python3 bench.py $1 --no-cpu-baseline --steps 1 --warmup 0 --frames sequential > $O/b.json 2> $O/b.err || exit 1
python3 - $O <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("rt::", "").split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    if not k.startswith("k_"): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(f"{k:34s} valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.3f} scalar {m.get('SQ_ACTIVE_INST_SCA',0)/wc:.3f} lds {m.get('SQ_ACTIVE_INST_LDS',0)/wc:.3f} vmem {m.get('SQ_ACTIVE_INST_VMEM',0)/wc:.3f} "
          f"parked {m.get('SQ_WAIT_ANY',0)/wc:.3f} issue-stall {m.get('SQ_WAIT_INST_ANY',0)/wc:.3f} | per wave: valu {m.get('SQ_INSTS_VALU',0)/m['SQ_WAVES']:.0f} salu {m.get('SQ_INSTS_SALU',0)/m['SQ_WAVES']:.0f} "
          f"smem {m.get('SQ_INSTS_SMEM',0)/m['SQ_WAVES']:.0f} lds {m.get('SQ_INSTS_LDS',0)/m['SQ_WAVES']:.0f} vmem {(m.get('SQ_INSTS_VMEM_RD',0)+m.get('SQ_INSTS_VMEM_WR',0))/m['SQ_WAVES']:.0f}")
PY
rm -rf $O/a $O/b
