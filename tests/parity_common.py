"""Shared parity checker: product path (C-ABI library) vs the oracle in Philox mode, byte for byte.

Used by tests/test_gpu_parity.py (-m gpu, real MI355X, libdwgsim_hip.so) and by
tests/test_emu_parity.py (CPU-only SIMT emulation build of the same kernel sources, tests/emu)."""
import os, subprocess, tempfile

from dwgsim_amd import api

STREAMS = {0: "bwa.read1.fastq", 1: "bwa.read2.fastq", 2: "bfast.fastq"}


def run_oracle(oracle_bin, fasta, flags, workdir):
    prefix = os.path.join(workdir, "ora")
    subprocess.run([oracle_bin, "--rng", "philox"] + flags.split() + [fasta, prefix], check=True, stderr=subprocess.DEVNULL)
    out = {}
    for k, suf in list(STREAMS.items()) + [("txt", "mutations.txt"), ("vcf", "mutations.vcf")]:
        p = prefix + "." + suf
        out[k] = open(p, "rb").read() if os.path.exists(p) else b""
    return out


def first_diff(a: bytes, b: bytes):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            lo = max(0, i - 120)
            return f"first difference at byte {i}: got …{a[lo:i + 60]!r} want …{b[lo:i + 60]!r}"
    return f"length differs: got {len(a)} want {len(b)}"


IN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs")


def compare_case(lib, oracle_bin, fasta, flags, batch_pairs=1 << 22, debug_options=None, group_bp=0):
    """Returns the JobResult; raises AssertionError with a readable diff on any mismatch."""
    flags = flags.replace("{IN}", IN_DIR)
    with tempfile.TemporaryDirectory() as t:
        want = run_oracle(oracle_bin, fasta, flags, t)
    params = api.parse_flags(flags, lib)
    contigs = api.read_fasta(fasta)
    res = api.run_job(params, contigs, batch_pairs=batch_pairs, lib=lib, debug_options=debug_options, group_bp=group_bp, check_list_form=True)
    assert res.mutations_txt == want["txt"], "mutations.txt: " + first_diff(res.mutations_txt, want["txt"])
    assert res.mutations_vcf == want["vcf"], "mutations.vcf: " + first_diff(res.mutations_vcf, want["vcf"])
    for k in STREAMS:
        assert res.streams[k] == want[k], f"{STREAMS[k]}: " + first_diff(res.streams[k], want[k])
    return res


def compare_job_api(lib, oracle_bin, fasta, flags, **kw):
    """The job level of the C-ABI (dwgsim_hip_job_*: the library schedules, groups, shards over the devices and orders) against the oracle."""
    flags = flags.replace("{IN}", IN_DIR)
    with tempfile.TemporaryDirectory() as t:
        want = run_oracle(oracle_bin, fasta, flags, t)
    res = api.run_job_api(api.parse_flags(flags, lib), api.read_fasta(fasta), lib=lib, **kw)
    assert res.mutations_txt == want["txt"], "mutations.txt: " + first_diff(res.mutations_txt, want["txt"])
    assert res.mutations_vcf == want["vcf"], "mutations.vcf: " + first_diff(res.mutations_vcf, want["vcf"])
    for k in STREAMS:
        assert res.streams[k] == want[k], f"{STREAMS[k]}: " + first_diff(res.streams[k], want[k])
    return res


FLOW = "TACGTACGTCTGAGCATCGATCGATGTACAGC"

# reads beyond every LDS staging limit (round 3 returned DWGSIM_HIP_ERR_UNSUP from ~4 600 bases on): the one-wave blocks stage them in scratch slots
# of global memory (dw_simulate.hip GS); the reference works on realloc'ed buffers and has no limit (dwgsim.c:75-153).  Run on the repeat-rich contigs.
LONG_READ_CASES = [
    "-z 5 -N 300 -1 10000 -2 0 -n 200 -r 0.01 -R 0.3 -X 0.5",
    "-z 5 -N 200 -c 1 -1 6000 -2 0 -n 150 -r 0.005",
    "-z 6 -N 150 -1 7000 -2 5000 -d 20000 -s 300 -n 300 -y 0.05 -o 0",
    "-z 7 -N 120 -1 25000 -2 0 -n 1000 -e 0.001-0.01 -Q 4",
]

# (fasta under tests/golden, flags): the option surface of the accelerated path (Illumina, SOLiD and Ion Torrent)
CASES = [
    ("ex1.fa", "-z 13 -N 10000"),                                  # the reference's bundled test configuration
    ("ex1.fa", "-z 13 -N 10000 -1 100 -2 100"),                   # BASELINE configs[0]
    ("tiny.fa", "-z 5 -C 20 -1 150 -2 150 -o 1"),
    ("tiny.fa", "-z 3 -N 5000 -r 0.01 -R 0.3 -X 0.5"),
    ("tiny.fa", "-z 4 -N 5000 -r 0.02 -R 0.5 -I 30 -X 0.6"),
    ("odd.fa", "-z 3 -N 5000 -1 50 -2 50 -d 200 -s 20 -r 0.1 -R 1.0 -X 0.7 -n 50"),
    ("odd.fa", "-z 2 -N 5000 -1 50 -2 50 -d 200 -s 20 -r 0.05 -R 0.7 -X 0.5 -n 3"),
    # a deletion that is left-justified THROUGH substituted cells (mut.c:515-516) into the footprint of the events in front of it: the one case
    # in 2 400 of the round-3 GPU fuzz run that differed -- the parallel justification had given it too short a reach (dw_walk.hip reach_del)
    ("odd.fa", "-z 8384 -1 7 -2 1 -d 900 -s 1 -C 0.5 -r 0.3 -y 0.3 -n 1000 -S 1 -H -o 1"),
    ("tiny.fa", "-z 9 -N 3000 -H"),
    ("tiny.fa", "-z 9 -N 3000 -S 1"),
    ("tiny.fa", "-z 9 -N 3000 -S 2 -A 1"),
    ("tiny.fa", "-z 9 -N 3000 -A 2"),
    ("tiny.fa", "-z 9 -N 3000 -i -d 100"),
    ("tiny.fa", "-z 9 -N 3000 -y 0"),
    ("tiny.fa", "-z 9 -N 3000 -n 2"),
    ("tiny.fa", "-z 9 -N 3000 -e 0.001-0.05 -E 0.01"),
    ("tiny.fa", "-z 9 -N 3000 -q I"),
    ("tiny.fa", "-z 9 -N 3000 -Q 0"),
    ("tiny.fa", "-z 9 -N 3000 -2 0"),
    ("tiny.fa", "-z 9 -N 3000 -P pfx"),
    ("tiny.fa", "-z 9 -N 3000 -o 2"),
    ("tiny.fa", "-z 9 -N 3000 -M 2"),
    ("tiny.fa", "-z 9 -N 300 -a"),
    ("tiny.fa", "-z 21 -N 3000 -F 0.2 -y 0.3 -Q 10"),
    ("tiny.fa", "-z 8 -N 2000 -1 33 -2 77 -d 300 -Q 40"),
    # Ion Torrent flow-space error model (BASELINE configs[4] shape: -c 2 -f <flow> -1 400 -2 0)
    ("tiny.fa", f"-z 9 -N 2000 -c 2 -f {FLOW} -1 400 -2 0 -e 0.01"),
    ("tiny.fa", f"-z 9 -N 2000 -c 2 -f {FLOW} -1 200 -2 100 -e 0.02 -E 0.03 -d 600"),
    ("tiny.fa", "-z 9 -N 1000 -c 2 -f TACG -1 100 -2 0 -e 0.2 -o 1"),
    ("odd.fa", f"-z 6 -N 3000 -c 2 -f {FLOW} -1 120 -2 0 -e 0.05 -n 10 -r 0.05 -R 0.5 -y 0.2"),
    ("ex1.fa", f"-z 6472 -N 1500 -c 2 -f {FLOW} -1 150 -2 0 -e 0.3 -A 1"),
    # flow orders that keep a base away for 32 and more flows (up to 63 are accepted): the hit bitmap of pass 2 is read in pieces
    ("tiny.fa", "-z 9 -N 1500 -c 2 -f TCG" + "A" * 37 + " -1 120 -2 0 -e 0.03"),
    ("tiny.fa", "-z 9 -N 1200 -c 2 -f " + "TACG" * 4 + "A" * 34 + " -1 90 -2 60 -e 0.02 -E 0.01 -d 300 -o 1"),                       # absurd per-flow error: deep insertion cascades in pass 2
    # reads too long to stage in LDS at 256 lanes per block: the one-wave-per-block variants
    ("tiny.fa", "-z 3 -N 400 -1 1500 -2 0 -n 40 -r 0.01 -R 0.3"),
    ("tiny.fa", "-z 3 -N 300 -1 1300 -2 1400 -d 3600 -s 40 -n 60 -y 0.1"),
    ("tiny.fa", "-z 3 -N 200 -c 1 -1 1400 -2 0 -n 60 -o 2"),
    # SOLiD colour space: first-colour bookkeeping in the BWA names, "/2"-"/1" suffix swap, 'A'+digits for BFAST
    ("ex1.fa", "-z 13 -N 5000 -c 1"),
    ("tiny.fa", "-z 8 -N 4000 -c 1 -1 50 -2 35 -d 300 -r 0.02 -R 0.5 -e 0.05 -E 0.03 -y 0.1"),
    ("tiny.fa", "-z 8 -N 3000 -c 1 -2 0 -o 1 -n 3"),
    ("odd.fa", "-z 6 -N 3000 -c 1 -1 40 -2 40 -d 150 -s 10 -r 0.08 -R 0.8 -n 20 -o 2 -q 5"),
    ("tiny.fa", "-z 8 -N 2000 -c 1 -1 1 -2 1 -d 50 -s 5 -Q 0"),
    # -B: the flow error is calibrated on 10^6 random reads before the run (dwgsim_opt.c:415-457)
    ("tiny.fa", f"-z 9 -N 1500 -c 2 -f {FLOW} -1 100 -2 0 -e 0.02 -B"),
    ("tiny.fa", "-z 9 -N 1000 -c 2 -f TACG -1 60 -2 40 -e 0.03 -E 0.01 -d 300 -B -o 1"),
    ("tiny.fa", "-z 9 -N 800 -c 2 -f TACG -1 50 -2 50 -e 0.02 -E 0.04 -d 300 -B"),            # equal lengths: end 2 takes end 1's rate (dwgsim_opt.c:425-431)
    # mutation-input files: -m txt, -v vcf, -b bed
    ("tiny.fa", "-z 5 -N 3000 -m {IN}/muts_generated.txt"),
    ("tiny.fa", "-z 5 -N 3000 -m {IN}/muts_edge.txt"),
    ("tiny.fa", "-z 5 -N 3000 -m {IN}/muts_edge.txt -H"),
    ("tiny.fa", "-z 5 -N 3000 -v {IN}/muts_generated.vcf"),
    ("tiny.fa", "-z 5 -N 3000 -v {IN}/muts_edge.vcf"),
    ("tiny.fa", "-z 5 -N 3000 -b {IN}/muts_edge.bed"),
    ("tiny.fa", "-z 5 -N 3000 -b {IN}/muts_edge.bed -H -o 1"),
    # target regions -x
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -N 3000"),
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -C 8"),
    ("tiny.fa", "-z 5 -x {IN}/regions_b.bed -N 2000 -d 200 -s 10 -1 50 -2 50 -n 5"),
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -N 2000 -2 0"),
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -C 5 -m {IN}/muts_edge.txt"),
]


# ---- the read-name contract dwgsim_eval consumes (src/dwgsim_eval.c; dwgsim.c:923-929): checked against the FASTA itself,
# ---- independently of the oracle
_COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


def check_read_names_tell_the_truth(res, contigs, lengths, min_checked=200):
    """@contig_pos1_pos2_strand1_strand2_0_0_e1:s1:i1_e2:s2:i2_hex/end -- for every non-random read without indels the
    Hamming distance between the read and the stated reference window is at most its stated errors + SNPs (0 => identical)."""
    ref = {name: bytes(arr).upper().translate(bytes.maketrans(b"RYMKWSBDHVX", b"N" * 11)) for name, arr in contigs}
    checked = exact = 0
    for end in (0, 1):
        lines = res.streams[end].split(b"\n")
        for k in range(0, len(lines) - 3, 4):
            name, seq = lines[k], lines[k + 1]
            assert name.startswith(b"@") and name.endswith(b"/%d" % (end + 1)) and lines[k + 2] == b"+" and len(lines[k + 3]) == len(seq)
            f = name[1:-2].rsplit(b"_", 9)           # contig names may contain '_': split from the right as dwgsim_eval does
            contig, pos, strand = f[0].decode(), [int(f[1]), int(f[2])], [int(f[3]), int(f[4])]
            counts = [tuple(int(x) for x in f[7 + e].split(b":")) for e in (0, 1)]
            int(f[9], 16)
            if f[5] == b"1":                         # random read: "rand_0_0_0_0_1_1_0:0:0_0:0:0_<hex>"
                assert contig.endswith("rand") and pos == [0, 0] and counts == [(0, 0, 0), (0, 0, 0)]
                continue
            assert len(seq) == lengths[end]
            n_err, n_sub, n_indel = counts[end]
            if n_indel:
                continue
            window = ref[contig][pos[end] - 1: pos[end] - 1 + len(seq)]
            if strand[end]:
                window = window.translate(_COMP)[::-1]
            ham = sum(1 for a, b in zip(seq, window) if a != b and b != ord("N"))      # reads print N wherever the reference has a non-ACGT base
            assert ham <= n_err + n_sub, (name, seq, window)
            checked += 1
            exact += (n_err + n_sub == 0)
    assert checked >= min_checked and exact > 0
    return checked, exact


def check_count_random_matches_simulate(lib, fasta, flags, ranges=((0, None), (17, 900), (1000, 1500))):
    """dwgsim_hip_count_random (k_place: accepts clean windows from the haplotype summaries, walks the rest) must agree with
    the random reads k_simulate itself produces for the same read-index range, retries included."""
    flags = flags.replace("{IN}", IN_DIR)
    params = api.parse_flags(flags, lib)
    contigs = api.read_fasta(fasta)
    tot = sum(len(a) for _, a in contigs)
    ctx = api.Context(params, 0, lib)
    have_regions = bool(getattr(params, "_regions", None))
    if have_regions:
        tot = ctx.set_regions(params._regions, contigs)
    checked = 0
    for idx, (name, arr) in enumerate(contigs):
        l_eff = ctx.region_length(idx, arr) if have_regions else len(arr)
        if l_eff < 0:
            continue
        n_pairs = api.pairs_for_contig(params, l_eff, tot, False, 0, lib)
        if n_pairs <= 0:
            continue
        cid = ctx.add_contig(name, arr, idx)
        if have_regions:
            ctx.set_placement_length(cid, l_eff)
        ctx.mutate(cid)
        for first, n in ranges:
            first = min(first, n_pairs - 1)
            n = n_pairs - first if n is None else min(n, n_pairs - first)
            want = ctx.simulate(cid, first, n, 0, 0)
            assert ctx.count_random(cid, first, n) == int(want.n_random), (name, first, n)
            checked += 1
        ctx.drop_contig(cid)
    assert checked > 0


def check_count_random_fast_path(lib, length=60000, n=2500, flags="-z 17 -1 50 -2 50 -d 300 -s 20 -C 10 -y 0.1 -r 0.01 -R 0.5 -n 1", ranges=((0, None), (333, 1111))):
    """dwgsim_hip_count_random on contigs where BOTH of its paths matter: most pairs are settled by k_place from two Philox blocks and the
    coarse (1024-cell) haplotype summaries under any insert size the pair can still draw, the pairs near the N runs, the contig ends and dense
    indels go through k_place_rest.  The counts must equal the random reads k_simulate itself produces for the same ranges; with the lists of
    open pairs made too small ("place_cap") the second, full-size run must give the same answer."""
    from dwgsim_amd import synth
    params = api.parse_flags(flags, lib)
    contigs = [("f1", synth.random_contig(length, 5, n_runs=[(0, 700), (length // 3, length // 3 + 2500), (length - 300, length)])),
               ("f2", synth.random_contig(length // 2, 6, n_runs=[(9000, 9040)]))]
    tot = sum(len(a) for _, a in contigs)
    with api.Context(params, 0, lib) as ctx:
        h0 = ctx.add_contigs(contigs, indices=[0, 1])
        ctx.mutate(h0)
        for k, (name, arr) in enumerate(contigs):
            n_pairs = min(n, api.pairs_for_contig(params, len(arr), tot, False, 0, lib))
            for first, cnt in ranges:
                first = min(first, n_pairs - 1)
                cnt = n_pairs - first if cnt is None else min(cnt, n_pairs - first)
                want = int(ctx.simulate(h0 + k, first, cnt, 0, 0).n_random)
                for cap in (-1, 0, 3):
                    ctx.debug_option("place_cap", cap)
                    assert ctx.count_random(h0 + k, first, cnt) == want, (name, first, cnt, cap)
                    opened = ctx.debug_get("place_open")
                    assert 0 < opened < cnt // 2, (name, first, cnt, opened)      # both paths were taken
        # several ranges of both contigs in one call
        ctx.debug_option("place_cap", -1)
        rr = [(h0, 0, 700), (h0, 700, 300), (h0 + 1, 5, 900)]
        per = ctx.count_random_ranges(rr, per_range=True)
        assert list(per) == [int(ctx.simulate(c, f, m, 0, 0).n_random) for c, f, m in rr]


def check_both_abort(lib, oracle_bin, fasta, flags, group_bp=0):
    """Jobs the reference gives up on ("failed to generate a read after 10001 trials", dwgsim.c:833-843: one counter of failed attempts
    over the pairs of a contig, reset only by a genomic read): the oracle exits non-zero and the HIP path must return the same error."""
    with tempfile.TemporaryDirectory() as t:
        r = subprocess.run([oracle_bin, "--rng", "philox"] + flags.split() + [fasta, os.path.join(t, "o")], capture_output=True, text=True)
    assert r.returncode != 0 and "failed to generate a read" in r.stderr, r.stderr[-300:]
    try:
        api.run_job(api.parse_flags(flags, lib), api.read_fasta(fasta), lib=lib, group_bp=group_bp)
    except api.DwgsimError as e:
        assert "failed to generate a read after 10001 trials" in str(e), str(e)
        return
    raise AssertionError("the HIP path produced output where the reference aborts: " + flags)


def check_mut_debug_aborts(lib, oracle_bin, golden_dir):
    """Mutation inputs on which the reference's mut_debug asserts end the run (mut.c:379-425; reference-pinned in tests/golden/MANIFEST.json,
    cases G6_abort_*): the oracle aborts on the same assert, and the product returns an error carrying the assert's text instead of output."""
    import json
    man = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["cases"]
    n = 0
    for name, c in sorted(man.items()):
        if "aborts" not in c:
            continue
        flags = c["flags"].replace("{IN}", IN_DIR)
        fasta = os.path.join(golden_dir, c["fasta"])
        with tempfile.TemporaryDirectory() as t:
            r = subprocess.run([oracle_bin, "--rng", "philox"] + flags.split() + [fasta, os.path.join(t, "o")], capture_output=True, text=True)
        assert r.returncode == -6 and c["aborts"]["assertion"] in r.stderr
        try:
            api.run_job(api.parse_flags(flags, lib), api.read_fasta(fasta), lib=lib)
        except api.DwgsimError as e:
            assert "mut_debug: Assertion `" + c["aborts"]["assertion"] + "' failed" in str(e), str(e)
            n += 1
            continue
        raise AssertionError("the product produced output where the reference aborts: " + name)
    assert n == 2


def check_gpu_gzip(lib, fasta, flags, sizes):
    """dwgsim_hip_set_gzip: for every stream of every batch, the bytes fetch_gz returns are complete gzip members whose decompressed bytes are
    exactly the text of that stream (the property the reference's own test checks of its .gz files, testdata/test.sh:21-26); switching it
    off again leaves the text path untouched.  `sizes`: pairs per call (1 pair: one short member; many pairs: several 32 KiB members)."""
    import gzip, zlib
    contigs = api.read_fasta(fasta)
    params = api.parse_flags(flags, lib)
    name, arr = contigs[0]
    with api.Context(params, 0, lib) as ctx:
        cid = ctx.add_contig(name, arr, 0)
        ctx.mutate(cid)
        plain = ctx.simulate(cid, 0, sizes[0], 0, 0)
        keep = [ctx.fetch(0, s, plain.bytes[s]) for s in range(3)]
        assert list(plain.gz_bytes) == [0, 0, 0]
        ctx.set_gzip(True)
        for k, n in enumerate(sizes):
            b = ctx.simulate(cid, 0, n, 0, k & 1)
            for s in range(3):
                txt = ctx.fetch(k & 1, s, b.bytes[s])
                gz = ctx.fetch_gz(k & 1, s, b.gz_bytes[s])
                assert (len(gz) == 0) == (len(txt) == 0)
                if k == 0:
                    assert txt == keep[s]
                if not gz:
                    continue
                assert gzip.decompress(gz) == txt, (n, s)
                # member by member: each is a complete gzip file of at most 32 KiB of text (a multiple of 4 bytes long)
                off, n_members, total, view = 0, 0, 0, memoryview(gz)
                while off < len(gz):
                    d = zlib.decompressobj(31)
                    chunk = bytes(view[off:off + 40000])          # (a member is < 40 000 bytes)
                    part = d.decompress(chunk)
                    used = len(chunk) - len(d.unused_data)
                    assert d.eof and 0 < len(part) <= 32768 and used % 4 == 0
                    total += len(part); n_members += 1; off += used
                assert total == len(txt) and n_members == (len(txt) + 32767) // 32768
                assert len(gz) < 0.6 * len(txt) + 300 * n_members
        ctx.set_gzip(False)
        b = ctx.simulate(cid, 0, sizes[0], 0, 0)
        assert list(b.gz_bytes) == [0, 0, 0] and [ctx.fetch(0, s, b.bytes[s]) for s in range(3)] == keep


WRITER_CASES = [      # (flags, contig-name length): record sizes, read lengths and name lengths that move every piece of the record writers
    ("-z 31 -N 1500 -1 150 -2 150 -y 0.1", 0),
    ("-z 32 -N 1500 -1 33 -2 17 -d 120 -s 8 -o 0", 0),
    ("-z 33 -N 1200 -1 1 -2 1 -d 40 -s 3 -o 1", 0),              # records shorter than one burst
    ("-z 34 -N 1200 -1 64 -2 31 -d 200 -P some_prefix -o 2", 0),
    ("-z 35 -N 1000 -1 129 -2 0 -e 0.02 -o 0", 200),             # a name longer than the part kept in LDS
    ("-z 36 -N 1000 -1 250 -2 250 -d 600 -o 1", 131),
]


def check_record_writers(lib, oracle_bin, tmpdir, flags, name_len):
    """Both record writers of the Illumina kernels (dw_read.hpp: the LDS FIFO with 32-byte aligned bursts, and 16-byte pieces from
    registers; the host picks one by LDS occupancy, the "writer" hook forces it) against the oracle, byte for byte."""
    import random
    rnd = random.Random(name_len + 7)
    seq = "".join(rnd.choice("ACGT") for _ in range(6000))
    name = ("ctg" + "x" * name_len)[:max(3, name_len)]
    fa = os.path.join(tmpdir, f"w{name_len}.fa")
    with open(fa, "w") as f:
        f.write(f">{name}\n")
        for i in range(0, len(seq), 70):
            f.write(seq[i:i + 70] + "\n")
    for writer in (0, 1):
        compare_case(lib, oracle_bin, fa, flags, batch_pairs=700, debug_options={"writer": writer})


def check_gzip_kernel_on_hard_inputs(lib, scale=0):
    """k_gzip on bytes the simulator never produces: one repeated byte, every byte value, random bytes (incompressible: the member must still
    fit its image), a Fibonacci histogram (unlimited Huffman codes would be 20+ bits deep: the counts are halved until 15 suffice), sizes
    around the 32 KiB member boundary.  gunzip(members) must give the input back."""
    import gzip, random
    rnd = random.Random(5)
    fib = [1, 1]
    while len(fib) < 22:
        fib.append(fib[-1] + fib[-2])
    fib_bytes = b"".join(bytes([65 + k]) * f for k, f in enumerate(fib))        # 28 656 bytes, depth 21 without a limit
    fib_mixed = bytearray(fib_bytes); rnd.shuffle(fib_mixed)
    cases = [b"A", b"AB", b"\n" * 7, b"G" * 32769, bytes(range(256)) * 3, bytes(rnd.randrange(256) for _ in range(33000)), bytes(fib_mixed)]
    if scale:       # (the emulation needs seconds per member: the long cases run on the GPU only)
        cases += [b"G" * 32768, bytes(rnd.randrange(256) for _ in range(40000 * scale)), bytes(fib_mixed) * (3 * scale), b"ACGT" * 8191 + b"N",
                  (b"@r\nACGT\n+\nIIII\n") * (2100 * scale), bytes(rnd.choice(b"ACGT") for _ in range(32767)), bytes(rnd.choice(b"ACGT") for _ in range(65536 * scale + 1))]
    params = api.parse_flags("-z 1 -N 10", lib)
    with api.Context(params, 0, lib) as ctx:
        assert ctx.debug_gzip(b"") == b""
        for data in cases:
            gz = ctx.debug_gzip(data)
            assert gzip.decompress(gz) == data, (len(data), data[:16])
            assert len(gz) % 4 == 0 and len(gz) <= len(data) + 300 * (len(data) // 32768 + 1)


def check_walking_a_contig_again(lib, fasta, flags, n=400):
    """dwgsim_hip_mutate_contig on a contig that was walked before starts again from the pristine copies: the 64-cell chunks the previous walk
    wrote are set back (dw_walk.hip k_dirty_chunks), the walk runs, the views and summaries of the chunks it wrote are made again.  Same mutation
    files, same reads, same random-read counts (they come from the summaries) -- also when a walk has to be re-run for capacity (round 2), when the
    views are made from every cell as rounds 1-4 did ("dense_view", round 3) and on the sparse walk behind that (round 4: everything is set back)."""
    contigs = api.read_fasta(fasta)
    params = api.parse_flags(flags, lib)
    name, arr = contigs[0]
    with api.Context(params, 0, lib) as ctx:
        cid = ctx.add_contig(name, arr, 0)
        got = []
        for rnd in range(5):
            ctx.debug_option("walk_cap", 5 if rnd == 2 else -1)
            ctx.debug_option("dense_view", 1 if rnd == 3 else 0)
            ctx.mutate(cid)
            cnt = ctx.count_random(cid, 7, n - 7)
            b = ctx.simulate(cid, 0, n, 0, rnd & 1)
            got.append((ctx.mutations_text(cid), [ctx.fetch(rnd & 1, s, b.bytes[s]) for s in range(3)], cnt))
        assert got[0] == got[1] == got[2] == got[3] == got[4]
        assert len(got[0][0][0]) > 200
