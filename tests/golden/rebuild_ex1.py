#!/usr/bin/env python3
"""Rebuild samtools/examples/ex1.fa (absent: empty submodule in the reference checkout) from the
reference's own golden FASTQ (testdata/ex1.test.bfast.fastq.gz) by majority vote, as described in
SURVEY.md Appendix C.  Run once in the dev container; the result is committed as tests/golden/ex1.fa.

usage: rebuild_ex1.py <testdata dir> <out.fa>
"""
import gzip, sys, collections, hashlib

def main(testdata, out):
    comp = str.maketrans("ACGTN", "TGCAN")
    lens = {}
    for line in gzip.open(f"{testdata}/ex1.test.mutations.vcf.gz", "rt"):
        if line.startswith("##contig=<ID="):
            body = line.strip()[len("##contig=<ID="):-1]
            name, ln = body.split(",length=")
            lens[name] = int(ln)
    votes = {n: [collections.Counter() for _ in range(l)] for n, l in lens.items()}
    f = gzip.open(f"{testdata}/ex1.test.bfast.fastq.gz", "rt")
    recno = 0
    while True:
        h = f.readline()
        if not h:
            break
        s = f.readline().strip(); f.readline(); f.readline()
        name = h[1:].strip()
        if name.startswith("rand"):
            recno += 1
            continue
        # contig_pos1_pos2_str1_str2_0_0_e:s:i_e:s:i_hex ; contig may hold '_' -> parse from the right
        parts = name.rsplit("_", 9)
        contig, p1, p2, s1, s2 = parts[0], int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4])
        e1, e2 = parts[7].split(":"), parts[8].split(":")
        j = recno & 1   # bfast is interleaved read1, read2 (random pairs also 2 records)
        pos, strand, indel = (p1, s1, int(e1[2])) if j == 0 else (p2, s2, int(e2[2]))
        recno += 1
        if indel != 0:
            continue
        if strand == 1:
            s = s.translate(comp)[::-1]
        for k, ch in enumerate(s):
            q = pos - 1 + k
            if 0 <= q < lens[contig]:
                votes[contig][q][ch] += 1
    txt = {}
    for line in gzip.open(f"{testdata}/ex1.test.mutations.txt.gz", "rt"):
        c, p, ref, alt, hap = line.rstrip("\n").split("\t")
        txt[(c, int(p) - 1)] = ref
    with open(out, "w") as fo:
        for n, l in lens.items():
            seq = []
            for q in range(l):
                if votes[n][q]:
                    seq.append(votes[n][q].most_common(1)[0][0])
                else:
                    seq.append(txt[(n, q)])   # uncovered base: taken from golden mutations.txt
            seq = "".join(seq)
            fo.write(f">{n}\n")
            for i in range(0, l, 60):
                fo.write(seq[i:i + 60] + "\n")
    print(hashlib.sha256(open(out, "rb").read()).hexdigest())

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
