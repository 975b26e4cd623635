"""Table of the floating-point instruction counters of tools/pmc_sq_flops.sh: python3 tools/pmc_sq_flops.py LABEL DIR..."""
import collections
import csv
import glob
import sys

label, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0].replace("void rt::", "").replace("rt::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"## {label}")
print("# mean per launch; instruction counts are per wave (64 lanes each, whatever the execution mask)")
print(f"{'kernel':32s} {'waves':>6s} {'f64 add':>9s} {'f64 mul':>9s} {'f64 fma':>9s} {'f64 trans':>9s} {'f32 add':>9s} {'f32 mul':>9s} {'f32 fma':>9s} "
      f"{'f32 trans':>9s} {'int32':>9s} {'GFLOP f64':>10s} {'GFLOP f32':>10s} {'lanes/VALU':>10s} {'LDS conflict':>12s}")
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    g = lambda n: m.get(n, 0.0)
    waves = max(g("SQ_WAVES"), 1.0)
    f64 = 64 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64") + 2 * g("SQ_INSTS_VALU_FMA_F64"))
    f32 = 64 * (g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_TRANS_F32") + 2 * g("SQ_INSTS_VALU_FMA_F32"))
    lanes = g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") if g("SQ_ACTIVE_INST_VALU") else 0.0
    conflict = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else 0.0
    per = lambda n: g(n) / waves
    print(f"{k[:32]:32s} {waves:6.0f} {per('SQ_INSTS_VALU_ADD_F64'):9.0f} {per('SQ_INSTS_VALU_MUL_F64'):9.0f} {per('SQ_INSTS_VALU_FMA_F64'):9.0f} "
          f"{per('SQ_INSTS_VALU_TRANS_F64'):9.0f} {per('SQ_INSTS_VALU_ADD_F32'):9.0f} {per('SQ_INSTS_VALU_MUL_F32'):9.0f} {per('SQ_INSTS_VALU_FMA_F32'):9.0f} "
          f"{per('SQ_INSTS_VALU_TRANS_F32'):9.0f} {per('SQ_INSTS_VALU_INT32'):9.0f} {f64 / 1e9:10.2f} {f32 / 1e9:10.2f} {lanes:10.1f} {conflict:12.3f}")
