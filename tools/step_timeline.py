#!/usr/bin/env python3
"""tools/step_timeline.py <results.db> [n] -- analysis only: from a `rocprofv3 --kernel-trace --memory-copy-trace` database of a bench.py run: everything the GPU
did around the last n (default 3) k_simulate launches, in start order, with the gap to the previous k_simulate's end -- where a step's time outside its
k_simulate goes."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def cols(t): return [r[1] for r in cur.execute(f"pragma table_info({t})").fetchall()]
kc = cols("kernels")
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in kc), None)
rows = [(s, e, nm, q, "K") for s, e, nm, q in cur.execute(f"select start, end, name, {qcol or 'NULL'} from kernels").fetchall()]
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
mc = next((t for t in tabs if t == "memory_copies"), None)
if mc:
    c = cols(mc); nm = "name" if "name" in c else "NULL"; sz = "size" if "size" in c else "0"
    rows += [(s, e, f"copy {k} {b} B", None, "C") for s, e, k, b in cur.execute(f"select start, end, {nm}, {sz} from {mc}").fetchall()]
rows.sort()
sims = [r for r in rows if "k_simulate" in r[2]]
if len(sims) < n + 1: sys.exit("too few k_simulate launches")
t_from = sims[-n - 1][1] - 200_000; prev_end = sims[-n - 1][1]
print(f"{'start_us':>10s} {'dur_us':>9s} {'queue':>6s}  what   (times relative to the end of the k_simulate before the last {n})")
for s, e, nm, q, kind in rows:
    if s < t_from: continue
    print(f"{(s - prev_end)/1e3:10.1f} {(e - s)/1e3:9.1f} {str(q):>6s}  {nm[:110]}")
gaps = [(sims[i + 1][0] - sims[i][1]) / 1e3 for i in range(len(sims) - n - 1, len(sims) - 1)]
print("gaps between consecutive k_simulate launches (end -> start), us:", ", ".join(f"{g:.1f}" for g in gaps))
