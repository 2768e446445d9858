"""-m gpu: BASELINE configs[3] and configs[4] at their size on ONE GPU -- the whole S4 genome (24 contigs with the GRCh38
primary-assembly lengths, 3.09 Gb, ~5.6 % N in telomere / centromere / short-arm blocks; dwgsim_amd/synth.py).

  configs[3]  2x150 bp PE, 30x   (-z 13 -1 150 -2 150 -C 30 -o 1):  ~325 M pairs, ~232 GB of FASTQ text
  configs[4]  Ion Torrent, 400 bp SE, 50x (-c 2 -f <flow> -1 400 -2 0 -C 50 -e 0.01; the flow model takes a uniform -e only,
              dwgsim_opt.c:338-343):  ~406 M reads

The oracle cannot simulate 3 x 10^8 pairs, so per job:
  (a) every contig runs through the batched API exactly as dwgsim_core would drive it (rand_ii and n_sim chained over the contigs,
      dwgsim.c:519-625, :1042, :1096); size-independent properties of the WHOLE output are checked on the device: four newlines per
      read, no 'N' anywhere in the text (no N base survives the filter, no name or quality holds one), the two paired streams have equal sizes, random reads ~ 5 %;
  (b) read-index windows at the start, in the middle and at the end of the first, a middle and the LAST contig are compared byte for byte
      with the oracle (its --as-contig / --range-rand-base mode: one contig of the genome, one window, no walk of the other 23), the
      rand_ii base coming from the chained counts of all earlier contigs + count_random of the contig's own prefix;
  (c) mutations.txt / .vcf of a mid-size contig (46.7 Mb) against the oracle;
  (d) one contig in a single 4.9 M-pair call equals the same contig in 1 M-pair batches on the other slot;
  (e) the -N remainder rule (dwgsim.c:535-537, :584-585) at genome scale: the last contig's window under -N against the oracle."""
import os, subprocess
import numpy as np
import pytest

from dwgsim_amd import api, synth

pytestmark = pytest.mark.gpu
FLOW = "TACGTACGTCTGAGCATCGATCGATGTACAGC"


@pytest.fixture(scope="module")
def lib():
    return api.load()


@pytest.fixture(scope="module")
def genome():
    return synth.workload_contigs("grch38")


def _oracle_window(oracle_bin, tmp_path, flags, contigs, k, n_sim_before, first, cnt, rand_base, fasta_cache):
    tot = sum(len(a) for _, a in contigs)
    fa = fasta_cache.get(k)
    if fa is None:
        fa = str(tmp_path / f"contig{k}.fa")
        synth.write_fasta(fa, [contigs[k]])
        fasta_cache[k] = fa
    pre = str(tmp_path / f"w{k}_{first}")
    subprocess.run([oracle_bin, "--rng", "philox", "--as-contig", f"{k},{tot},{len(contigs) - 1 - k},{n_sim_before}", "--emit-range", f"{first}:{cnt}",
                    "--range-rand-base", str(rand_base)] + flags.split() + [fa, pre], check=True, stderr=subprocess.DEVNULL)
    return pre


def _run_genome(lib, oracle_bin, tmp_path, contigs, flags, window_contigs, mut_contig=None, two_batchings_contig=None, batch=1 << 22, cnt=2000):
    params = api.parse_flags(flags, lib)
    paired = params.length[1] > 0
    tot = sum(len(a) for _, a in contigs)
    n_sim = 0; rand_ii = 0; tot_bytes = [0, 0]; tot_nl = [0, 0]; n_N = 0
    fasta_cache = {}
    report = []
    with api.Context(params, 0, lib) as ctx:
        for k, (name, arr) in enumerate(contigs):
            n_pairs = api.pairs_for_contig(params, len(arr), tot, k == len(contigs) - 1, n_sim, lib)
            assert n_pairs > 0, name
            cid = ctx.add_contig(name, arr, k)
            ctx.mutate(cid)
            if k == mut_contig:      # (c)
                pre = _oracle_window(oracle_bin, tmp_path, flags, contigs, k, n_sim, 0, 1, rand_ii, fasta_cache)
                txt, vcf = ctx.mutations_text(cid)
                assert txt == open(pre + ".mutations.txt", "rb").read(), name
                assert vcf == open(pre + ".mutations.vcf", "rb").read().split(b"INFO\n", 1)[1], name
                assert len(txt) > 100000
            # (b) windows against the oracle while the contig is resident
            if k in window_contigs:
                for first in (0, n_pairs // 2 + 12345, n_pairs - cnt):
                    before = ctx.count_random(cid, 0, first) if first else 0
                    b = ctx.simulate(cid, first, cnt, rand_ii + before, 1)
                    pre = _oracle_window(oracle_bin, tmp_path, flags, contigs, k, n_sim, first, cnt, rand_ii + before, fasta_cache)
                    for s, suf in ((0, "bwa.read1.fastq"), (1, "bwa.read2.fastq")):
                        want = open(pre + "." + suf, "rb").read() if os.path.exists(pre + "." + suf) else b""
                        assert ctx.fetch(1, s, b.bytes[s]) == want, (name, first, suf)
            # (a) the whole contig in batches, text checked on the device
            c_rand = 0
            first = 0
            while first < n_pairs:
                n = min(batch, n_pairs - first)
                b = ctx.simulate(cid, first, n, rand_ii + c_rand, 0)
                assert b.n_pairs == n
                for s in (0, 1) if paired else (0,):
                    nl = ctx.count_byte(0, s, 10)
                    assert nl == 4 * n, (name, first, s, nl)
                    tot_nl[s] += nl; tot_bytes[s] += int(b.bytes[s])
                    n_N += ctx.count_byte(0, s, ord("N"))
                if paired:
                    assert b.bytes[0] == b.bytes[1]
                c_rand += int(b.n_random); first += n
            if k in window_contigs:                          # the sharding primitive at this size
                assert ctx.count_random(cid, 0, n_pairs) == c_rand, name
            if k == two_batchings_contig:                    # (d)
                one = ctx.simulate(cid, 0, n_pairs, rand_ii, 0)
                keep = [ctx.fetch_np(0, s, one.bytes[s]) for s in ((0, 1) if paired else (0,))]
                off = [0, 0]; first = 0; rr = rand_ii
                while first < n_pairs:
                    n = min(1_000_003, n_pairs - first)
                    b = ctx.simulate(cid, first, n, rr, 1)
                    for s in range(len(keep)):
                        t = ctx.fetch_np(1, s, b.bytes[s])
                        assert np.array_equal(keep[s][off[s]:off[s] + int(b.bytes[s])], t), (name, first, s)
                        off[s] += int(b.bytes[s])
                    rr += int(b.n_random); first += n
                assert off[0] == len(keep[0]) and rr - rand_ii == one.n_random == c_rand
                del keep
            report.append((name, n_pairs, c_rand))
            rand_ii += c_rand; n_sim += n_pairs
            ctx.drop_contig(cid)
    return n_sim, rand_ii, tot_bytes, tot_nl, n_N, report


def test_whole_grch38_2x150_30x_on_one_gpu(lib, oracle_bin, genome, tmp_path):
    """BASELINE configs[3] at its size."""
    flags = "-z 13 -1 150 -2 150 -C 30 -o 1"
    n_sim, n_rand, tot_bytes, tot_nl, n_N, report = _run_genome(lib, oracle_bin, tmp_path, genome, flags, window_contigs=(0, 11, 23), mut_contig=20, two_batchings_contig=20)
    assert n_sim == sum(r[1] for r in report) and 320e6 < n_sim < 330e6, n_sim
    assert tot_nl[0] == tot_nl[1] == 4 * n_sim and tot_bytes[0] == tot_bytes[1] and tot_bytes[0] > 100e9
    assert n_N == 0                                           # -n 0: no read with an N base is emitted
    # a pair whose genomic attempt fails (N blocks: ~5.6 % of the positions) is retried FROM the random-read test (dwgsim.c:649, :833-842),
    # so random reads make up y / (y + (1 - y)(1 - f)) of the output, f = the failing share of genomic attempts
    f_lo, f_hi = 0.050, 0.065
    assert 0.05 / (0.05 + 0.95 * (1 - f_lo)) < n_rand / n_sim < 0.05 / (0.05 + 0.95 * (1 - f_hi)), n_rand / n_sim


def test_whole_grch38_iontorrent_400bp_50x_on_one_gpu(lib, oracle_bin, genome, tmp_path):
    """BASELINE configs[4] at its size (uniform per-flow error: the reference rejects ramps for -c 2)."""
    flags = f"-z 13 -c 2 -f {FLOW} -1 400 -2 0 -C 50 -e 0.01 -o 1"
    n_sim, n_rand, tot_bytes, tot_nl, n_N, report = _run_genome(lib, oracle_bin, tmp_path, genome, flags, window_contigs=(0, 23), batch=1 << 21, cnt=1000)
    assert 400e6 < n_sim < 412e6, n_sim
    assert tot_nl[0] == 4 * n_sim and tot_bytes[0] > 300e9
    assert 0.0520 < n_rand / n_sim < 0.0540, n_rand / n_sim


def test_grch38_last_contig_takes_the_remainder_under_N(lib, oracle_bin, genome, tmp_path):
    """-N: pairs per contig from the long-double share of the genome (dwgsim.c:582-586), the LAST contig takes what is left (:535-537).
    The window at the very end of the job against the oracle, which is told how many pairs came before (n_sim)."""
    flags = "-z 29 -N 40000000 -1 150 -2 150 -o 1"
    params = api.parse_flags(flags, lib)
    tot = sum(len(a) for _, a in genome)
    n_sim = 0; per = []
    for k, (name, arr) in enumerate(genome):
        n = api.pairs_for_contig(params, len(arr), tot, k == len(genome) - 1, n_sim, lib)
        ld = np.longdouble(len(arr)) / np.longdouble(tot) * np.longdouble(40000000) + np.longdouble(0.5)
        assert n == (int(ld) if k < len(genome) - 1 else 40000000 - n_sim), name
        per.append(n); n_sim += n
    assert n_sim == 40000000
    with api.Context(params, 0, lib) as ctx:
        rand_ii = 0
        for k, (name, arr) in enumerate(genome):
            cid = ctx.add_contig(name, arr, k)
            ctx.mutate(cid)
            if k == len(genome) - 1:
                first, cnt = per[k] - 1500, 1500
                before = ctx.count_random(cid, 0, first)
                b = ctx.simulate(cid, first, cnt, rand_ii + before, 0)
                pre = _oracle_window(oracle_bin, tmp_path, flags, genome, k, sum(per[:k]), first, cnt, rand_ii + before, {})
                for s, suf in ((0, "bwa.read1.fastq"), (1, "bwa.read2.fastq")):
                    assert ctx.fetch(0, s, b.bytes[s]) == open(pre + "." + suf, "rb").read(), suf
            else:
                rand_ii += ctx.count_random(cid, 0, per[k])
            ctx.drop_contig(cid)
