#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max us, share) of a rocprofv3 --kernel-trace results.db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                   "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':64s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:64]:64s} {r[1]:6d} {r[2]:11.1f} {r[3]:10.2f} {r[4]:9.2f} {r[5]:9.2f} {100*r[2]/tot:6.2f}")
