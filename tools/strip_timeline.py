"""Timeline of the last step in a rocprofv3 kernel trace (csv): python3 tools/strip_timeline.py TRACE.csv [N_KERNELS_PER_STEP]
Prints, for the kernels of the last step, start (us from the step's first kernel), duration, gap to the
previous kernel on the same queue, queue, grid and name; then per kernel name the summed duration and
the summed gaps that precede it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt::", ""),
             r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in rows)
# steps are separated by the longest gaps between consecutive kernel starts
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else None
if n_last is None:
    gaps = sorted(((ev[i + 1][0] - max(e[1] for e in ev[: i + 1][-50:]), i) for i in range(len(ev) - 1)), reverse=True)
    cut = max(i for g, i in gaps[:12] if i < len(ev) - 10)
    ev = ev[cut + 1:]
else:
    ev = ev[-n_last:]
t0 = ev[0][0]
last_end = {}
by = collections.Counter(); gapby = collections.Counter()
for s, e, n, q, g in ev:
    gap = s - last_end.get(q, s)
    last_end[q] = e
    by[n] += e - s; gapby[n] += gap
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:6.1f}  q{q}  grid {g:>8}  {n}")
print(f"span {(max(e[1] for e in ev) - t0) / 1e3:.1f} us, {len(ev)} kernels")
for n, v in by.most_common(): print(f"  {n:40s} {v / 1e3:9.1f} us   gaps before it {gapby[n] / 1e3:8.1f} us")
