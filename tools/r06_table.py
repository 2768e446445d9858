#!/usr/bin/env python3
"""tools/r06_table.py [dir] -- analysis only: the variants table of DESIGN.md section 5 from the profiles/r06_<name>_kernel_stats_pmc.txt files (tools/r06_final_profiles.sh):
launch time = rocprofv3 average of the k_simulate kernel(s), fraction = algorithmic bytes per launch / that time / 8 TB/s, traffic = FETCH_SIZE x 2 + WRITE_SIZE (KiB),
VALU-active = SQ_ACTIVE_INST_VALU x 4 / (1024 x GRBM_GUI_ACTIVE / 8)."""
import json, re, sys, os
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rows = [("chr20", "2 × 150 `-o 1` (the bench line)"), ("ecoli", "the same, E. coli-sized contig"), ("o0", "2 × 150 `-o 0`, the reference's default output"), ("solid50", "SOLiD 2 × 50 `-o 0`"),
        ("long2000", "2 000-base reads, single-end"), ("ion", "Ion Torrent 400 bp, E. coli-sized"), ("ion_chr20", "Ion Torrent 400 bp")]
print("| workload (chr20-sized contig unless said) | kernel(s) | launch (rocprofv3 avg) | algorithmic | fraction | traffic | VALU + SALU per wave | VALU-active |")
print("|---|---|---|---|---|---|---|---|")
for name, label in rows:
    p = os.path.join(d, f"r06_{name}_kernel_stats_pmc.txt")
    if not os.path.exists(p): continue
    txt = open(p).read().split("\n")
    bench = json.loads(next(l for l in txt if l.startswith("{")))
    kern, ctr = {}, {}
    for l in txt:
        m = re.match(r"^(void dw::k_simulate<[^>]*>)\(dw::SimArgs\)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)", l)
        if m: kern[m.group(1)] = float(m.group(4))
        m = re.match(r"^(void dw::k_simulate<[^>]*>)\(.*?\s([A-Z][A-Z0-9_]+)\s+([0-9.]+)\s+n=", l)
        if m: ctr.setdefault(m.group(2), {})[m.group(1)] = float(m.group(3))
    tot = lambda c: sum(ctr.get(c, {}).values())
    launches = max(bench["config"].get("launches_per_gpu_per_step", 1), 1)
    two = len(kern) > 1
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    t_us = sum(kern.values())
    waves = max(ctr.get("SQ_WAVES", {"": 1}).values())
    traffic = (tot("FETCH_SIZE") * 2 + tot("WRITE_SIZE")) * 1024
    va = 100.0 * tot("SQ_ACTIVE_INST_VALU") * 4 / (1024 * tot("GRBM_GUI_ACTIVE") / 8) if tot("GRBM_GUI_ACTIVE") else float("nan")
    ks = " + ".join(k.replace("void dw::", "").replace(", ", ",") for k in kern)
    print(f"| {label} | `{ks}` | {' + '.join('%.2f' % (v / 1e3) for v in kern.values())} ms | {alg / 1e9:.2f} GB | **{alg / (t_us * 1e-6) / 8e12:.3f}** | {traffic / 1e9:.1f} GB ({traffic / alg:.2f} ×) | {tot('SQ_INSTS_VALU') / waves / 1e3:.1f} k + {tot('SQ_INSTS_SALU') / waves / 1e3:.1f} k | {va:.0f} % |")
