"""Mode A (the reference's sequential drand48 stream, pinned byte for byte to the reference) against mode B (Philox at fixed counters, what
the HIP kernels reproduce byte for byte) on DISTRIBUTIONS.  The two modes share every line of the simulation and differ in where a uniform
comes from -- and mode B takes some of them in narrower forms (16-bit polar operands for the quality normals, 16 + 16 bit base-error draws,
32-bit substitution draws, hit-bit draws of the flow model).  Each such re-definition is a place where the oracle and the kernels could be
wrong TOGETHER; this test keeps them honest: on 10^6 pairs the observable statistics of the two modes must agree within sampling bounds
(two independent samples of the same distribution: differences of counts within 5 sigma of their binomial / Poisson spread).  CPU only."""
import os, subprocess
import numpy as np
import pytest

from dwgsim_amd import synth

N_PAIRS = 1_000_000
FLAGS = f"-N {N_PAIRS} -1 100 -2 100 -d 400 -s 40 -r 0.004 -R 0.3 -X 0.5 -e 0.005-0.03 -E 0.02 -y 0.04 -Q 3 -o 1"


def run_mode(oracle_bin, fasta, mode, seed, workdir):
    prefix = os.path.join(workdir, mode)
    subprocess.run([oracle_bin, "--rng", mode, "-z", str(seed)] + FLAGS.split() + [fasta, prefix], check=True, stderr=subprocess.DEVNULL)
    st = {}
    qual = np.zeros(256, dtype=np.int64); base = np.zeros(256, dtype=np.int64)
    qual_by_pos = np.zeros((2, 100), dtype=np.float64)
    for end in (0, 1):
        lines = open(f"{prefix}.bwa.read{end + 1}.fastq", "rb").read().split(b"\n")
        q = np.frombuffer(b"".join(lines[3::4]), dtype=np.uint8)
        qual += np.bincount(q, minlength=256)
        qual_by_pos[end] = q.reshape(-1, 100).mean(axis=0)
        base += np.bincount(np.frombuffer(b"".join(lines[1::4]), dtype=np.uint8), minlength=256)
        if end == 0:
            names = lines[0::4][:-1] if lines[-1] == b"" and len(lines) % 4 == 1 else lines[0::4]
    n_rand = 0
    ins, err, sub, ind, strand = [], [0, 0], [0, 0], [0, 0], 0
    for nm in names:
        f = nm[1:-2].rsplit(b"_", 9)
        if f[5] == b"1":
            n_rand += 1
            continue
        ins.append(abs(int(f[2]) - int(f[1])))
        strand += f[3] == b"1"
        for e in (0, 1):
            a, b, c = f[7 + e].split(b":")
            err[e] += int(a); sub[e] += int(b); ind[e] += int(c)
    ins = np.asarray(ins, dtype=np.float64)
    st.update(n=len(names), n_rand=n_rand, ins_mean=ins.mean(), ins_std=ins.std(), ins_n=len(ins), err=err, sub=sub, ind=ind, strand=strand, qual=qual, base=base, qual_by_pos=qual_by_pos)
    # the mutation walk: counts by type, indel length histogram (mutations.txt: deletions one line per base, insertions one line per event)
    ins_len = np.zeros(64, dtype=np.int64); n_sub = n_del_bases = n_ins = n_het = 0
    for ln in open(prefix + ".mutations.txt", "rb"):
        c = ln.rstrip(b"\n").split(b"\t")
        n_het += c[4] != b"3"
        if c[2] == b"-":
            n_ins += 1; ins_len[min(len(c[3]), 63)] += 1
        elif c[3] == b"-":
            n_del_bases += 1
        else:
            n_sub += 1
    st.update(n_sub=n_sub, n_del_bases=n_del_bases, n_ins=n_ins, n_het=n_het, ins_len=ins_len)
    for suf in ("bwa.read1.fastq", "bwa.read2.fastq", "bfast.fastq", "mutations.txt", "mutations.vcf"):
        try:
            os.remove(f"{prefix}.{suf}")
        except OSError:
            pass
    return st


def close_counts(a, b, what, k=5.0, floor=30.0):
    """two Poisson / binomial counts of the same expectation: their difference within k sigma (counts below `floor` say nothing)"""
    assert abs(a - b) <= k * np.sqrt(a + b) or (a + b) < floor, (what, a, b, abs(a - b) / max(np.sqrt(a + b), 1))


def test_mode_b_draws_from_the_same_distributions_as_mode_a(oracle_bin, tmp_path):
    fa = str(tmp_path / "ref.fa")
    synth.write_fasta(fa, [("c1", synth.random_contig(3_000_000, 5, [(700_000, 700_500)]))])
    A = run_mode(oracle_bin, fa, "drand48", 11, str(tmp_path))
    B = run_mode(oracle_bin, fa, "philox", 12, str(tmp_path))
    assert A["n"] == B["n"] == N_PAIRS
    # random reads (dwgsim.c:649), strands (:723)
    close_counts(A["n_rand"], B["n_rand"], "random reads")
    close_counts(A["strand"], B["strand"], "reverse-strand first ends")
    # sequencing errors of both ends (dwgsim.c:233-244; ramp on end 1, constant on end 2): ~1.75 M and 2 M errors per mode
    for e in (0, 1):
        close_counts(A["err"][e], B["err"][e], f"sequencing errors of end {e + 1}")
        # SNPs and indel cells crossed by the reads depend on the two (different) mutation walks: the same rates, compared loosely
        assert abs(A["sub"][e] - B["sub"][e]) < 0.05 * (A["sub"][e] + B["sub"][e]), ("n_sub", e, A["sub"][e], B["sub"][e])
        assert abs(A["ind"][e] - B["ind"][e]) < 0.08 * (A["ind"][e] + B["ind"][e]), ("n_indel", e, A["ind"][e], B["ind"][e])
    # insert sizes (dwgsim.c:657-663): mean within 5 standard errors, spread within 1 %
    se = np.hypot(A["ins_std"], B["ins_std"]) / np.sqrt(min(A["ins_n"], B["ins_n"]))
    assert abs(A["ins_mean"] - B["ins_mean"]) < 5 * se, (A["ins_mean"], B["ins_mean"], se)
    assert abs(A["ins_std"] / B["ins_std"] - 1) < 0.01, (A["ins_std"], B["ins_std"])
    assert 39.0 < B["ins_std"] < 41.0 and 299.0 < B["ins_mean"] < 301.5          # (names carry the leftmost coordinate of both ends: d - 100)
    # quality characters (dwgsim.c:899-918; the 16-bit polar operands of mode B): every character of '!' .. 'I', 2 x 10^8 of them per mode
    qa, qb = A["qual"], B["qual"]
    assert qa.sum() == qb.sum() == 2 * 100 * N_PAIRS and qa[:33].sum() == 0 and qa[74:].sum() == 0 and qb[:33].sum() == 0 and qb[74:].sum() == 0
    for c in range(33, 74):
        close_counts(int(qa[c]), int(qb[c]), f"quality character {chr(c)!r}", k=5.5)
    chi2 = float((((qa - qb) ** 2) / np.maximum(qa + qb, 1))[33:74].sum())
    assert chi2 < 41 + 6 * np.sqrt(2 * 41), chi2                    # chi-square with ~40 degrees of freedom
    assert np.abs(A["qual_by_pos"] - B["qual_by_pos"]).max() < 0.02, np.abs(A["qual_by_pos"] - B["qual_by_pos"]).max()      # mean quality per position (the ramp)
    # bases (substitution draws dwgsim.c:238, random-read bases :1000): A C G T N counts
    for ch in b"ACGTN":
        close_counts(int(A["base"][ch]), int(B["base"][ch]), f"base {chr(ch)}", k=6.0)
    # the mutation walk (mut.c:607-642): substitutions, deleted bases, insertions, heterozygous share, insertion lengths (geometric, -X 0.5)
    close_counts(A["n_sub"], B["n_sub"], "substitutions")
    close_counts(A["n_ins"], B["n_ins"], "insertions")
    assert abs(A["n_del_bases"] - B["n_del_bases"]) < 6 * np.sqrt(3 * (A["n_del_bases"] + B["n_del_bases"])), (A["n_del_bases"], B["n_del_bases"])      # (runs of geometric length: over-dispersed)
    close_counts(A["n_het"], B["n_het"], "heterozygous lines", k=6.0)
    for L in range(1, 8):
        close_counts(int(A["ins_len"][L]), int(B["ins_len"][L]), f"insertions of length {L}")
    assert 0.45 < A["ins_len"][1] / A["n_ins"] < 0.55 and 0.45 < B["ins_len"][1] / B["n_ins"] < 0.55


@pytest.mark.parametrize("e,flow", [(0.01, "TACGTACGTCTGAGCATCGATCGATGTACAGC"), (0.06, "TACG")])
def test_the_gap_drawn_flow_model_has_the_law_of_the_reference(oracle_bin, tmp_path, e, flow):
    """Ion Torrent (dwgsim.c:246-417).  Mode B does not draw the first uniform of every homopolymer start / empty flow: it draws the GAPS between the scoring
    ones (Geometric(e'), by inversion in integer arithmetic: oracle flow_first) -- the same Bernoulli process in law, a dozen draws per read
    instead of 1 400.  Replay parity shows that the unmodified reference, given those decisions, writes mode B's bytes; THIS shows that the decisions have
    the reference's distribution: 150 000 reads of 200 bases per mode (mode A = the reference's sequential stream, byte-pinned), the per-read error counts
    (histogram, mean, variance: clustering would show there) and the read lengths after errors (insertions against deletions)."""
    fa = str(tmp_path / "ref.fa")
    synth.write_fasta(fa, [("c1", synth.random_contig(400_000, 9))])
    n = 150_000
    out = {}
    for mode, seed in (("drand48", 21), ("philox", 22)):
        prefix = str(tmp_path / mode)
        subprocess.run([oracle_bin, "--rng", mode, "-z", str(seed), "-N", str(n), "-c", "2", "-f", flow, "-1", "200", "-2", "0", "-e", str(e), "-r", "0", "-y", "0", "-o", "1", fa, prefix],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(prefix + ".bwa.read1.fastq", "rb").read().split(b"\n")
        names, seqs = lines[0::4][:n], lines[1::4][:n]
        errs = np.array([int(nm.rsplit(b"_", 3)[1].split(b":")[0]) for nm in names])
        lens = np.array([len(s) for s in seqs])
        out[mode] = (errs, lens)
        os.remove(prefix + ".bwa.read1.fastq")
    (ea, la), (eb, lb) = out["drand48"], out["philox"]
    assert len(ea) == len(eb) == n
    # error counts per read: mean within 5 standard errors, variance within 3 %, every bin of the histogram within 5 sigma
    se = np.hypot(ea.std(), eb.std()) / np.sqrt(n)
    assert abs(ea.mean() - eb.mean()) < 5 * se, (ea.mean(), eb.mean(), se)
    assert abs(ea.var() / eb.var() - 1) < 0.03, (ea.var(), eb.var())
    ha, hb = np.bincount(ea, minlength=64)[:64], np.bincount(eb, minlength=64)[:64]
    for k in range(64):
        close_counts(int(ha[k]), int(hb[k]), f"reads with {k} flow errors")
    # read lengths after errors (insertions lengthen, deletions shorten): mean, spread, histogram around 200
    assert abs(la.mean() - lb.mean()) < 5 * np.hypot(la.std(), lb.std()) / np.sqrt(n), (la.mean(), lb.mean())
    assert abs(la.std() / lb.std() - 1) < 0.02, (la.std(), lb.std())
    for L in range(190, 215):
        close_counts(int((la == L).sum()), int((lb == L).sum()), f"reads of length {L}")
    assert ea.mean() > 50 * e          # (the test is not vacuous: errors do happen, ~ 200 (1 + 2.4) e of them per read)
