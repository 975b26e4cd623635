#!/usr/bin/env python3
"""Timeline of the last end-to-end frame (the last one with disk-light samples, if any) in a rocprofv3 results database
(bench.py's end_to_end loop): kernels longer than 0.3 ms or following an idle gap of more than 50 us,
as (start ms, duration ms, gap before ms, kernel)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
generated = [i for i, r in enumerate(rows) if 'k_mt_generate' in r[2]]
first = generated[-1] if generated else len(rows) - 1      # (frames without disk lights: the last frame of the run)
while first > 0 and 'k_init' not in rows[first][2]:
    first -= 1
t0 = prev_end = rows[first][0]


def short(n):
    m = re.search(r'rt\d+(k_\w+?)(I|E)', n)
    return n[:40] if not m else m.group(1)


every = len(sys.argv) > 2
for s, e, n in rows[first:]:
    gap = (s - prev_end) / 1e6
    if every or gap > 0.05 or (e - s) > 0.3e6:
        print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f} {gap:7.3f} {short(n)}")
    prev_end = max(prev_end, e)
    if 'copyBuffer' in n and (e - s) > 1e6:
        break
