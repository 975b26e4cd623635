"""Concurrency of the kernels in a rocprofv3 kernel trace (csv): python3 tools/overlap_stats.py TRACE.csv
Prints, for the last third of the trace (steady state): wall span, time with >= 1 kernel running, the
sum of kernel durations, and per kernel name its summed duration."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", "")) for r in rows]
ev.sort()
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 2 // 3
ev = [e for e in ev if e[0] >= t_lo]
span = ev[-1][1] - ev[0][0]
pts = []
for s, e, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0; depth = 0; last = pts[0][0]; hist = collections.Counter()
for t, d in pts:
    if depth > 0: busy += t - last
    hist[min(depth, 6)] += t - last
    depth += d; last = t
tot = sum(e - s for s, e, _ in ev)
print(f"span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({busy/span:.1%}), sum of durations {tot/1e6:.2f} ms, mean concurrency {tot/busy:.2f}")
print("time by number of kernels running:", {k: f"{v/span:.1%}" for k, v in sorted(hist.items())})
by = collections.Counter()
for s, e, n in ev: by[n] += e - s
for n, v in by.most_common(8): print(f"  {n:40s} {v/1e6:8.2f} ms")
