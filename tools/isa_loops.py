#!/usr/bin/env python3
"""tools/isa_loops.py -- analysis only: static instruction mix of one kernel of a hipcc --save-temps assembly file (.s), per basic block and per LOOP
(a backward branch closes a loop: the blocks from its target to it).  usage: isa_loops.py file.s <substring of the kernel's mangled name> [min_valu]
Prints the loops sorted by their VALU count: label range, VALU / SALU / LDS / VMEM / SMEM instructions of one trip through ALL their blocks (an
upper bound of an iteration: blocks of rare branches inside the loop are counted too)."""
import re, sys

def classify(op):
    if op.startswith(("v_",)): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path, key = sys.argv[1], sys.argv[2]
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split("\n")
    # the kernel's body: from "<name>:" to the matching ".Lfunc_end"
    start = next(i for i, l in enumerate(lines) if key in l.split(":")[0] and re.match(r"^[A-Za-z_][^\s]*:", l) and not l.startswith(".L"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": lines[start][:-1][:40], "n": {}, "br": []}
    order = {}
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith((";", "//", ".")) and not re.match(r"^\.LBB\d+_\d+:", s):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur); cur = {"label": m.group(1), "n": {}, "br": []}
            continue
        op = s.split()[0]
        c = classify(op)
        cur["n"][c] = cur["n"].get(c, 0) + 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur["br"].append(s.split()[1])
    blocks.append(cur)
    for i, b in enumerate(blocks): order[b["label"]] = i
    tot = {}
    for b in blocks:
        for k, v in b["n"].items(): tot[k] = tot.get(k, 0) + v
    print("kernel", lines[start].split(":")[0][:90], "blocks", len(blocks), "static:", tot)
    loops = []
    for i, b in enumerate(blocks):
        for t in b["br"]:
            if t in order and order[t] <= i:
                n = {}
                for bb in blocks[order[t]:i + 1]:
                    for k, v in bb["n"].items(): n[k] = n.get(k, 0) + v
                loops.append((n.get("valu", 0), t, b["label"], i - order[t] + 1, n))
    loops.sort(reverse=True)
    for v, t, e, nb, n in loops:
        if v >= min_valu:
            print(f"loop {t:>12} .. {e:<12} blocks {nb:4d}  VALU {n.get('valu',0):6d} SALU {n.get('salu',0):6d} LDS {n.get('lds',0):5d} VMEM {n.get('vmem',0):4d} SMEM {n.get('smem',0):3d}")

main()
