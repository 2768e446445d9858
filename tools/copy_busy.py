#!/usr/bin/env python3
"""tools/copy_busy.py <results.db> -- analysis only: from a `rocprofv3 --kernel-trace --memory-copy-trace` database: how busy the copy engine
(device-to-host copies) and the compute queue were over the run -- union of the intervals / span first..last --, the same per 250 ms bin, the
largest gaps between copies, bytes moved and the rate while a copy was in flight."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
def cols(t): return [r[1] for r in cur.execute(f"pragma table_info({t})").fetchall()]
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
mc = next((t for t in tabs if t == "memory_copies"), None) or next(t for t in tabs if "memory_cop" in t)
c = cols(mc)
size_col = "size" if "size" in c else next((x for x in c if "size" in x or "bytes" in x), None)
name_col = "name" if "name" in c else next((x for x in c if "name" in x or "kind" in x or "direction" in x), None)
rows = cur.execute(f"select start, end, {size_col or 0}, {name_col or 'NULL'} from {mc} order by start").fetchall()
kinds = {}
for s, e, n, k in rows:
    a = kinds.setdefault(str(k), [0, 0, 0]); a[0] += 1; a[1] += n or 0; a[2] += e - s
print("copies by kind: " + "; ".join(f"{k}: {v[0]} copies, {v[1]/1e9:.2f} GB, {v[2]/1e9:.3f} s in flight" for k, v in kinds.items()))
d2h = [(s, e, n) for s, e, n, k in rows if "DEVICE_TO_HOST" in str(k).upper() or "DTOH" in str(k).upper()] or [(s, e, n) for s, e, n, k in rows]
big = [x for x in d2h if (x[2] or 0) >= (1 << 20)]
kern = cur.execute("select start, end from kernels order by start").fetchall()
def union(iv):
    tot = 0; cs = ce = None; merged = []
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: merged.append((cs, ce)); cs, ce = s, e
    if cs is not None: merged.append((cs, ce))
    return merged
def report(label, iv):
    if not iv: print(label, "none"); return
    m = union(sorted(iv)); busy = sum(e - s for s, e in m); span = m[-1][1] - m[0][0]
    print(f"{label}: busy {busy/1e9:.3f} s of {span/1e9:.3f} s = {busy/span:.3f}")
    return m
t0 = min([x[0] for x in big] + [kern[0][0]]); t1 = max([x[1] for x in big] + [kern[-1][1]])
print(f"run (first GPU activity to last): {(t1-t0)/1e9:.3f} s; copies >= 1 MiB device to host: {len(big)}, {sum(x[2] for x in big)/1e9:.2f} GB")
mcopy = report("copy engine (D2H copies >= 1 MiB)", [(s, e) for s, e, n in big])
mk = report("kernels", kern)
busy = sum(e - s for s, e in mcopy); print(f"rate while a copy is in flight: {sum(x[2] for x in big)/busy:.2f} GB/s")
first = mcopy[0][0]
print(f"first big copy starts {(first-t0)/1e9:.3f} s after the first GPU activity; after it the copy engine is busy {sum(e-s for s,e in mcopy)/(mcopy[-1][1]-first):.3f} of the time")
gaps = sorted(((mcopy[i+1][0] - mcopy[i][1], mcopy[i][1] - t0) for i in range(len(mcopy) - 1)), reverse=True)
print("largest gaps between copies (ms @ s into the run): " + ", ".join(f"{g/1e6:.1f}@{at/1e9:.2f}" for g, at in gaps[:12]))
print(f"gaps: {len(gaps)} in all, {sum(g for g, _ in gaps)/1e9:.3f} s; of it in gaps > 1 ms: {sum(g for g, _ in gaps if g > 1e6)/1e9:.3f} s ({sum(1 for g, _ in gaps if g > 1e6)}), 0.1-1 ms: {sum(g for g, _ in gaps if 1e5 < g <= 1e6)/1e9:.3f} s ({sum(1 for g, _ in gaps if 1e5 < g <= 1e6)})")
binw = 250e6; nb = int((t1 - t0) / binw) + 1
def per_bin(m):
    out = [0.0] * nb
    for s, e in m:
        b0 = int((s - t0) / binw); b1 = int((e - t0) / binw)
        for b in range(b0, b1 + 1):
            lo = max(s, t0 + b * binw); hi = min(e, t0 + (b + 1) * binw)
            if hi > lo: out[b] += (hi - lo) / binw
    return out
print("per 250 ms: copy busy   " + " ".join(f"{x:.2f}" for x in per_bin(mcopy)))
print("per 250 ms: kernel busy " + " ".join(f"{x:.2f}" for x in per_bin(mk)))
rows = cur.execute("select name, count(*), sum(end-start)/1e6 from kernels group by name order by 3 desc limit 14").fetchall()
for r in rows: print(f"  {r[0][:70]:70s} {r[1]:6d} {r[2]:9.1f} ms")
