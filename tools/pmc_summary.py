#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 --pmc results.db files."""
import sqlite3, sys
for f in sys.argv[1:]:
    db = sqlite3.connect(f)
    cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration), max(vgpr_count), max(lds_block_size) from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
    print("#", f)
    for r in rows:
        if "simulate" in r[0] or "place" in r[0]:
            print(f"{r[0][:40]:40s} {r[1]:24s} {r[2]:16.1f}  n={r[3]} dur_us={r[4]/1e3:.1f} vgpr={r[5]} lds={r[6]}")
