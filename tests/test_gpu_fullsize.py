"""-m gpu: parity at BASELINE.json's full sizes.

configs[1] (E. coli-sized S2, 2x150, 30x, 488 595 pairs): the whole job is compared bit-for-bit with the
oracle (sha256 of both FASTQ streams and the mutation files).
configs[2] (chr20-sized S3 with N blocks, 64 Mb, 6.78 M pairs): the oracle is too slow for the whole job,
so (a) windows of read indices at the start, middle and end are compared bit-for-bit with the oracle's
--emit-range output (the API call a shard would make: count_random for the prefix, simulate for the
window), (b) one-call vs many-batch runs must give identical stream hashes, (c) structural invariants
(record count, line structure, read lengths, N filter) hold over the full output."""
import hashlib, os, subprocess
import numpy as np
import pytest

from dwgsim_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return api.load()


def _oracle(oracle_bin, flags, fasta, prefix, extra=()):
    subprocess.run([oracle_bin, "--rng", "philox", *extra] + flags.split() + [fasta, prefix], check=True, stderr=subprocess.DEVNULL)


def test_ecoli_full_job_bit_exact(lib, oracle_bin, tmp_path):
    flags = "-z 13 -1 150 -2 150 -C 30 -o 1"
    fa = str(tmp_path / "ecoli.fa")
    contigs = synth.workload_contigs("ecoli")
    synth.write_fasta(fa, contigs)
    _oracle(oracle_bin, flags, fa, str(tmp_path / "o"))
    res = api.run_job(api.parse_flags(flags, lib), contigs, lib=lib)
    assert res.n_pairs == 488595
    for k, suf in ((0, "bwa.read1.fastq"), (1, "bwa.read2.fastq")):
        want = hashlib.sha256(open(str(tmp_path / ("o." + suf)), "rb").read()).hexdigest()
        assert hashlib.sha256(res.streams[k]).hexdigest() == want, suf
    assert res.mutations_txt == open(str(tmp_path / "o.mutations.txt"), "rb").read()
    assert res.mutations_vcf == open(str(tmp_path / "o.mutations.vcf"), "rb").read()


def test_iontorrent_config5_shape_bit_exact(lib, oracle_bin, tmp_path):
    """BASELINE configs[4] shape on one contig: -c 2 -f <flow> -1 400 -2 0 -e 0.01 (uniform error, as the reference requires),
    4.6 Mb, 6x (73 289 reads): FASTQ (variable read lengths) and mutation files bit-for-bit against the oracle."""
    flow = "TACGTACGTCTGAGCATCGATCGATGTACAGC"
    flags = f"-z 17 -c 2 -f {flow} -1 400 -2 0 -C 6 -e 0.01 -o 1"
    fa = str(tmp_path / "ecoli.fa")
    contigs = synth.workload_contigs("ecoli")
    synth.write_fasta(fa, contigs)
    _oracle(oracle_bin, flags, fa, str(tmp_path / "o"))
    res = api.run_job(api.parse_flags(flags, lib), contigs, lib=lib)
    want = open(str(tmp_path / "o.bwa.read1.fastq"), "rb").read()
    assert res.n_pairs == 73289 and len(res.streams[0]) == len(want)
    assert hashlib.sha256(res.streams[0]).hexdigest() == hashlib.sha256(want).hexdigest()
    lens = np.diff(np.flatnonzero(np.frombuffer(res.streams[0], dtype=np.uint8) == 10))[0::4]   # sequence-line lengths (+1)
    assert lens.min() < 401 and lens.max() > 402       # flow errors really change read lengths
    assert res.mutations_vcf == open(str(tmp_path / "o.mutations.vcf"), "rb").read()


def test_chr20_sized_windows_and_invariants(lib, oracle_bin, tmp_path):
    flags = "-z 20 -1 150 -2 150 -C 30 -o 1 -r 0.001 -R 0.1"
    fa = str(tmp_path / "chr20.fa")
    contigs = synth.workload_contigs("chr20")
    synth.write_fasta(fa, contigs)
    params = api.parse_flags(flags, lib)
    name, arr = contigs[0]
    n_pairs = api.pairs_for_contig(params, len(arr), len(arr), True, 0, lib)
    assert n_pairs == 6783597
    with api.Context(params, 0, lib) as ctx:
        cid = ctx.add_contig(name, arr, 0)
        ctx.mutate(cid)
        txt, vcf = ctx.mutations_text(cid)
        # (a) three windows against the oracle (the walk of 64 Mb + the attempt loop runs once per window on the CPU)
        for first in (0, 3_400_000, n_pairs - 5000):
            cnt = 5000
            pre = str(tmp_path / f"w{first}")
            _oracle(oracle_bin, flags, fa, pre, extra=("--emit-range", f"{first}:{cnt}"))
            rand_base = ctx.count_random(cid, 0, first) if first else 0
            b = ctx.simulate(cid, first, cnt, rand_base, 0)
            for s, suf in ((0, "bwa.read1.fastq"), (1, "bwa.read2.fastq")):
                assert ctx.fetch(0, s, b.bytes[s]) == open(pre + "." + suf, "rb").read(), (first, suf)
            if first == 0:
                assert txt == open(pre + ".mutations.txt", "rb").read() and vcf.endswith(open(pre + ".mutations.vcf", "rb").read().split(b"INFO\n", 1)[1])
        # (b) one call vs batches of 1 000 003 pairs
        one = ctx.simulate(cid, 0, n_pairs, 0, 0)
        h_one = [hashlib.sha256(ctx.fetch(0, s, one.bytes[s])).hexdigest() for s in (0, 1)]
        hs = [hashlib.sha256(), hashlib.sha256()]
        first, rand_ii = 0, 0
        last_stream = None
        while first < n_pairs:
            n = min(1_000_003, n_pairs - first)
            b = ctx.simulate(cid, first, n, rand_ii, 1)
            for s in (0, 1):
                last_stream = ctx.fetch(1, s, b.bytes[s]); hs[s].update(last_stream)
            rand_ii += b.n_random; first += n
        assert [h.hexdigest() for h in hs] == h_one and rand_ii == one.n_random
        # (c) structure of the full read-1 stream
        data = np.frombuffer(ctx.fetch(0, 0, one.bytes[0]), dtype=np.uint8)
        nl = np.flatnonzero(data == 10)
        assert len(nl) == 4 * n_pairs
        starts = np.concatenate(([0], nl[:-1] + 1))
        assert (data[starts[0::4]] == ord("@")).all() and (data[starts[2::4]] == ord("+")).all()
        assert ((nl[1::4] - starts[1::4]) == 150).all() and ((nl[3::4] - starts[3::4]) == 150).all()
        seq_bytes = np.zeros(256, dtype=bool); seq_bytes[[65, 67, 71, 84]] = True      # -n 0: no N survives the filter
        body = np.concatenate([data[a:a + 150] for a in starts[1::4][:: max(1, n_pairs // 20000)]])
        assert seq_bytes[body].all()
        q = np.concatenate([data[a:a + 150] for a in starts[3::4][:: max(1, n_pairs // 20000)]])
        assert q.min() >= 33 and q.max() <= 73


FLOW32 = "TACGTACGTCTGAGCATCGATCGATGTACAGC"


@pytest.mark.parametrize("workload,flags,windows", [
    # sequence with a genome's composition (synth.genome_like_contig: GC content, SINE- / LINE-like repeat families, microsatellites, homopolymers,
    # soft-masked lower case): the whole E. coli-sized job, and windows + the mutation files of the chr20-sized one, 2 x 150 and Ion Torrent
    ("ecoli_like", "-z 13 -1 150 -2 150 -C 30 -o 1", None),
    ("ecoli_like", f"-z 17 -c 2 -f {FLOW32} -1 400 -2 0 -C 6 -e 0.01 -o 1", None),
    ("chr20_like", "-z 20 -1 150 -2 150 -C 30 -o 1 -r 0.001 -R 0.1", (0, 3_400_000, -4000)),
    ("chr20_like", "-z 21 -1 150 -2 150 -C 30 -r 0.001 -R 0.1", (2_000_000,)),                      # -o 0, the reference's default: both output families
    ("chr20_like", f"-z 22 -c 2 -f {FLOW32} -1 400 -2 0 -C 50 -e 0.01 -o 1", (0, 2_100_000, -4000)),
])
def test_genome_like_sequence_bit_exact(lib, oracle_bin, tmp_path, workload, flags, windows):
    """Homopolymers and microsatellites drive the flow model (one event test per homopolymer, the dot-fill rule: dwgsim.c:281-364) and the reach of
    left-justification (mut.c:482-589); lower-case bases go through nst_nt4_table like upper-case ones (dwgsim.c:56-73)."""
    fa = str(tmp_path / "ref.fa")
    contigs = synth.workload_contigs(workload)
    synth.write_fasta(fa, contigs)
    params = api.parse_flags(flags, lib)
    name, arr = contigs[0]
    assert (arr >= 97).any() or workload == "ecoli_like"
    sufs = ((0, "bwa.read1.fastq"), (1, "bwa.read2.fastq"), (2, "bfast.fastq"))
    if windows is None:
        _oracle(oracle_bin, flags, fa, str(tmp_path / "o"))
        res = api.run_job(api.parse_flags(flags, lib), contigs, lib=lib)
        for k, suf in sufs:
            p = str(tmp_path / ("o." + suf))
            want = open(p, "rb").read() if os.path.exists(p) else b""
            assert len(res.streams[k]) == len(want) and hashlib.sha256(res.streams[k]).hexdigest() == hashlib.sha256(want).hexdigest(), suf
        assert res.mutations_txt == open(str(tmp_path / "o.mutations.txt"), "rb").read()
        assert res.mutations_vcf == open(str(tmp_path / "o.mutations.vcf"), "rb").read()
        return
    n_pairs = api.pairs_for_contig(params, len(arr), len(arr), True, 0, lib)
    with api.Context(params, 0, lib) as ctx:
        cid = ctx.add_contig(name, arr, 0)
        ctx.mutate(cid)
        txt, vcf = ctx.mutations_text(cid)
        for k, first in enumerate(windows):
            cnt = 4000
            if first < 0:
                first = n_pairs + first
            pre = str(tmp_path / f"w{first}")
            _oracle(oracle_bin, flags, fa, pre, extra=("--emit-range", f"{first}:{cnt}"))
            rand_base = ctx.count_random(cid, 0, first) if first else 0
            b = ctx.simulate(cid, first, cnt, rand_base, 0)
            for s, suf in sufs:
                want = open(pre + "." + suf, "rb").read() if os.path.exists(pre + "." + suf) else b""
                got = ctx.fetch(0, s, b.bytes[s]) if b.bytes[s] else b""
                assert got == want, (first, suf)
            if k == 0:
                assert txt == open(pre + ".mutations.txt", "rb").read() and vcf.endswith(open(pre + ".mutations.vcf", "rb").read().split(b"INFO\n", 1)[1])


def _assembly_like():
    """A scaffold-level assembly in miniature: 14 contigs from 150 bp to 3 Mb, names with underscores / dots / a 200-character
    name, N-rich and all-N contigs, contigs shorter than a read or a fragment (skip rules #2-#5, dwgsim.c:595-618)."""
    rc = synth.random_contig
    return [
        ("chr1_synth", rc(3_000_000, 101, [(0, 10_000), (1_500_000, 1_500_600)])),
        ("scaffold_0001.1", rc(800_000, 102)),
        ("tiny_a", rc(150, 103)),                                   # shorter than the fragment: skipped
        ("all_N", np.full(50_000, ord("N"), dtype=np.uint8)),       # every attempt fails on it unless it gets no pairs
        ("x" * 200, rc(400_000, 104, [(100, 400)])),
        ("scaffold_0002", rc(1_200_000, 105, [(600_000, 600_050)])),
        ("tiny_b", rc(420, 106)),
        ("unplaced_17_random", rc(90_000, 107)),
        ("scaffold_0003", rc(2_000_000, 108)),
        ("mito", rc(16_569, 109)),
        ("tiny_c", rc(299, 110)),
        ("scaffold_0004", rc(650_000, 111, [(0, 200), (649_000, 650_000)])),
        ("scaffold_0005", rc(33_000, 112)),
        ("last_one", rc(500_000, 113)),
    ]


@pytest.mark.parametrize("flags", [
    "-z 77 -N 60000 -1 100 -2 100 -d 350 -s 30 -r 0.002 -R 0.2 -y 0.03 -n 2",     # -N: the last contig takes the remainder (dwgsim.c:535-537)
    "-z 78 -C 1.2 -1 125 -2 0 -r 0.001 -e 0.01-0.03 -o 2 -P lib1",                  # -C, single end, BFAST output only, read prefix
])
def test_assembly_like_multi_contig_job_bit_exact(lib, oracle_bin, tmp_path, flags):
    """rand_ii and n_sim chained over many contigs, per-contig RNG keys, skip rules in the middle of the genome, long names."""
    contigs = [c for c in _assembly_like() if c[0] != "all_N"] if "-N" in flags else _assembly_like()
    fa = str(tmp_path / "asm.fa")
    synth.write_fasta(fa, contigs)
    _oracle(oracle_bin, flags, fa, str(tmp_path / "o"))
    res = api.run_job(api.parse_flags(flags, lib), contigs, lib=lib)
    for k, suf in ((0, "bwa.read1.fastq"), (1, "bwa.read2.fastq"), (2, "bfast.fastq")):
        p = str(tmp_path / ("o." + suf))
        want = open(p, "rb").read() if os.path.exists(p) else b""
        assert len(res.streams[k]) == len(want) and hashlib.sha256(res.streams[k]).hexdigest() == hashlib.sha256(want).hexdigest(), suf
    assert sum(len(s) for s in res.streams.values()) > 10_000_000
    assert res.mutations_txt == open(str(tmp_path / "o.mutations.txt"), "rb").read()
    assert res.mutations_vcf == open(str(tmp_path / "o.mutations.vcf"), "rb").read()


def test_gzip_members_of_a_full_batch(lib):
    """GPU gzip at the size dwgsim-hip runs it: one 2^20-pair batch of the chr20-sized job (2 x 380 MB of text, 11 600 members per stream made
    by one k_gzip launch each, member offsets by look-back across all of them): gunzip(members) == text, for both writers' slots."""
    import zlib
    contigs = synth.workload_contigs("chr20")
    params = api.parse_flags("-z 13 -1 150 -2 150 -C 30 -o 1", lib)
    name, arr = contigs[0]
    with api.Context(params, 0, lib) as ctx:
        ctx.set_gzip(True)
        cid = ctx.add_contig(name, arr, 0)
        ctx.mutate(cid)
        for slot, first in ((0, 0), (1, 3_000_000)):
            b = ctx.simulate(cid, first, 1 << 20, 0, slot)
            for s in (0, 1):
                txt = ctx.fetch_np(slot, s, b.bytes[s])
                gz = ctx.fetch_gz(slot, s, b.gz_bytes[s])
                assert 0.4 * len(txt) < len(gz) < 0.55 * len(txt)
                out = []; n_members = 0; off = 0; view = memoryview(gz)
                while off < len(gz):                          # member by member (zlib stops at the end of each; a member is < 40 000 bytes)
                    d = zlib.decompressobj(31)
                    chunk = bytes(view[off:off + 40000])
                    out.append(d.decompress(chunk)); assert d.eof
                    off += len(chunk) - len(d.unused_data); n_members += 1
                back = np.frombuffer(b"".join(out), dtype=np.uint8)
                assert n_members == (len(txt) + 32767) // 32768
                assert len(back) == len(txt) and np.array_equal(back, txt), (slot, s)
