#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 results database: name, launches, total ms, average ms."""
import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc").fetchall()
    print(path)
    for r in rows[:16]:
        print(f"  {r[0][:88]:88s} {r[1]:6d} {r[2]:9.2f} {r[3]:8.3f}")
