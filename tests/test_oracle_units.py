"""Known-answer tests for the oracle's primitives."""
import ctypes, math, random, struct


def test_philox_known_answers(oracle_lib):
    f = oracle_lib.oracle_philox4x32_10
    A4, A2 = ctypes.c_uint32 * 4, ctypes.c_uint32 * 2
    for ctr, key, want in [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]:
        out = A4()
        f(A4(*ctr), A2(*key), out)
        assert tuple(out) == want


def test_drand48_known_answers(oracle_lib):
    x = ctypes.c_uint64((13 << 16) | 0x330E)
    got = [oracle_lib.oracle_drand48_next(ctypes.byref(x)) for _ in range(3)]
    assert got == [0.49125804875894019, 0.9095780156526132, 0.69616396868708463]


def test_philox_uniform_layout(oracle_lib):
    # u = ((w_hi << 21) | (w_lo >> 11)) * 2^-53 of block slot>>1, half slot&1; counter layout of DESIGN.md
    A4, A2 = ctypes.c_uint32 * 4, ctypes.c_uint32 * 2
    seed, contig, dom, idx, att, retry, slot = 13, 2, 8, (5 << 32) | 77, 3, 2, 9
    out = A4()
    oracle_lib.oracle_philox4x32_10(A4(idx & 0xffffffff, ((idx >> 32) & 0xffff) | (retry << 16), (dom << 24) | att, slot >> 1), A2(seed, contig), out)
    want = ((out[2] << 21) | (out[3] >> 11)) * 2.0 ** -53
    assert oracle_lib.oracle_philox_uniform(seed, contig, dom, idx, att, retry, slot) == want
    assert 0.0 <= want < 1.0


def test_det_log_accuracy(oracle_lib):
    rnd = random.Random(1)
    worst = 0
    for _ in range(200000):
        x = rnd.random() if rnd.random() < 0.8 else math.ldexp(rnd.random() + 0.5, rnd.randint(-110, 3))
        if x <= 0:
            continue
        a, b = oracle_lib.oracle_det_log(x), math.log(x)
        ia, ib = struct.unpack("<q", struct.pack("<d", a))[0], struct.unpack("<q", struct.pack("<d", b))[0]
        worst = max(worst, abs(ia - ib))
    assert worst <= 1, worst   # within 1 ulp of glibc everywhere
    assert oracle_lib.oracle_det_log(1.0) == 0.0


def test_single_contig_window_equals_the_slice_of_the_whole_genome_run(oracle_bin, golden_dir, tmp_path):
    """--as-contig / --range-rand-base (mode B): one contig of a larger genome, one read-index window of it, simulated without walking
    the rest -- must be byte-identical to that window of the whole-genome run (what the whole-GRCh38 GPU tests rely on)."""
    import os, subprocess
    from dwgsim_amd import api, synth
    contigs = api.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    flags = "-z 31 -N 3000 -1 60 -2 60 -d 200 -s 15 -y 0.2 -r 0.01 -o 1"
    full = str(tmp_path / "full")
    subprocess.run([oracle_bin, "--rng", "philox"] + flags.split() + [os.path.join(golden_dir, "tiny.fa"), full], check=True, stderr=subprocess.DEVNULL)
    recs = open(full + ".bwa.read1.fastq", "rb").read().split(b"\n")
    recs = [b"\n".join(recs[k:k + 4]) + b"\n" for k in range(0, len(recs) - 1, 4)]
    assert len(recs) == 3000
    lib = None
    try:
        lib = api.load()
    except RuntimeError:
        pass
    tot = sum(len(a) for _, a in contigs)
    # pairs per contig (dwgsim.c:582-586; the last contig takes the remainder, the 300-bp contig is skipped by rule #3)
    n1 = int(len(contigs[0][1]) / tot * 3000 + 0.5)
    k, first, cnt = 1, 100, 250                                   # contig t2, pairs [100, 350)
    lo = n1 + first
    rand_base = sum(1 for r in recs[:lo] if r.startswith(b"@rand_"))
    fa1 = str(tmp_path / "t2.fa")
    synth.write_fasta(fa1, [contigs[k]])
    win = str(tmp_path / "win")
    subprocess.run([oracle_bin, "--rng", "philox", "--as-contig", f"{k},{tot},{len(contigs) - 1 - k},{n1}", "--emit-range", f"{first}:{cnt}",
                    "--range-rand-base", str(rand_base)] + flags.split() + [fa1, win], check=True, stderr=subprocess.DEVNULL)
    got = open(win + ".bwa.read1.fastq", "rb").read()
    assert got == b"".join(recs[lo:lo + cnt])
    assert any(r.startswith(b"@rand_") for r in recs[lo:lo + cnt])
    # the mutation body lines of that contig are those of the whole-genome run
    want_txt = b"".join(l + b"\n" for l in open(full + ".mutations.txt", "rb").read().split(b"\n") if l.startswith(contigs[k][0].encode() + b"\t"))
    assert open(win + ".mutations.txt", "rb").read() == want_txt and len(want_txt) > 0


def test_every_possible_quality_try_gives_the_offset_glibc_log_would(oracle_bin):
    """The link between the two oracle modes on the headline path (paired ends, -Q 2).  A quality normal of mode B is a polar try on two 16-bit
    operands (D_QUAL0): 2^32 possible tries, 3 373 258 460 of them accepted.  The reference's ran_normal (dwgsim.c:156-175) uses
    glibc's log, the oracle and the kernels det_log; their last bits differ on ~7 % of the accepted radii -- and for NO try does that change
    (int)(nrm * sigma + 0.5) (dwgsim.c:912), for any -Q below.  oracle/exhaust_log.c folds the tries into the 2^29 pairs (|s1| <= |s2|) that share
    rsq and compares the truncations of all four signed variates wherever the logs differ bitwise.  Together with the replay cases of even-length
    reads (tests/replay_common.py) this retires "same code, other provider" for the quality path."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(oracle_bin), "exhaust_log")
    sig = ["0.3", "0.5", "1", "2", "3", "3.7", "10", "40", "60", "5000"]
    out = subprocess.run([exe, str(min(16, os.cpu_count() or 1))] + sig, check=True, capture_output=True, text=True).stdout.split()
    kv = dict(zip(out[0::2], out[1::2]))
    assert int(kv["tries_accepted"]) == 3373258460            # |{(s1, s2) in [-32768, 32767]^2 : 0 < s1^2 + s2^2 < 2^30}|
    assert int(kv["log_bits_differ"]) > 10**7                 # (the check is not vacuous: the two logs do differ, on 30 M of the 422 M distinct radii)
    for s in sig:
        assert int(kv[f"offsets_differ[{float(s):g}]"]) == 0, (s, kv)
