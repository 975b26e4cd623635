# Occupancy / lane counters of one configuration: bash tools/pmc_sq_many.sh CONFIG OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
C=${1:-many}; O=$GRAFT_REPO_ROOT/gpurun_out/${2:-pmc_sq_$C}; rm -rf $O; mkdir -p $O
B="python3 bench.py --config $C --no-cpu-baseline"
timeout -s KILL 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o c -- $B --steps 1 --warmup 0 --frames sequential > $O/pmc_sq.json 2> $O/pmc_sq.err || exit 1
python3 - $O <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    if not k.startswith("k_"): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if not m.get("SQ_ACTIVE_INST_VALU"): continue
    print(f"{k:28s} launches {len(c['SQ_WAVES']):3d} lanes/valu {m['SQ_THREAD_CYCLES_VALU'] / m['SQ_ACTIVE_INST_VALU']:5.1f} valu/wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:9.0f} "
          f"valu share {m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']:.3f} wait share {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} waves {m['SQ_WAVES']:.0f}")
PY
rm -rf $O/pmc_sq
