"""Summarise a rocprofv3 --pmc csv: per kernel name, mean of each counter per dispatch."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, d in sorted(acc.items()):
            print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
