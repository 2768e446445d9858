"""CPU-only check of the kernels' logic: the SAME kernel sources (dwgsim_amd/csrc) compiled against
the SIMT emulation shim in tests/emu (one OS thread per GPU thread) must match the oracle in Philox
mode byte for byte.  This is test infrastructure -- the product library has no CPU path -- and only
a subset of the GPU parity cases is run (the emulation is slow)."""
import os, subprocess, sys
import pytest

from dwgsim_amd import api
from parity_common import compare_case

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_CASES = [
    ("ex1.fa", "-z 13 -N 1500"),
    ("tiny.fa", "-z 4 -N 1200 -r 0.02 -R 0.5 -I 30 -X 0.6"),
    ("odd.fa", "-z 3 -N 1200 -1 50 -2 50 -d 200 -s 20 -r 0.1 -R 1.0 -X 0.7 -n 50"),
    ("tiny.fa", "-z 9 -N 1000 -2 0 -n 2"),
    ("tiny.fa", "-z 9 -N 700 -o 1 -y 0.3 -P pfx -A 2"),
    ("tiny.fa", "-z 9 -N 600 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -e 0.01"),
    ("tiny.fa", "-z 9 -N 400 -c 2 -f TACG -1 100 -2 60 -e 0.2 -E 0.1 -d 300 -o 1"),
    ("tiny.fa", "-z 9 -N 300 -c 2 -f TCG" + "A" * 37 + " -1 120 -2 0 -e 0.03"),                      # 39 empty flows in front of a T: more than one 31-bit piece of the hit bitmap
    ("tiny.fa", "-z 8 -N 700 -c 1 -1 50 -2 35 -d 300 -r 0.02 -R 0.5 -e 0.05 -E 0.03 -y 0.1"),
    ("odd.fa", "-z 6 -N 500 -c 1 -2 0 -1 40 -r 0.08 -R 0.8 -n 20"),
    ("tiny.fa", "-z 3 -N 200 -1 1300 -2 1400 -d 3600 -s 40 -n 60 -y 0.1"),
    ("tiny.fa", "-z 5 -N 600 -m {IN}/muts_edge.txt"),
    ("tiny.fa", "-z 5 -N 600 -v {IN}/muts_edge.vcf"),
    ("tiny.fa", "-z 5 -N 600 -b {IN}/muts_edge.bed"),
    ("tiny.fa", "-z 5 -M 2 -m {IN}/muts_generated.txt"),
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -N 700"),
    ("tiny.fa", "-z 5 -x {IN}/regions_b.bed -C 4 -d 200 -s 10 -1 50 -2 50 -n 5"),
]


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.run([os.path.join(HERE, "emu", "build.sh")], check=True, stdout=subprocess.DEVNULL)
    return api.load(os.path.join(HERE, "emu", "libdwgsim_emu.so"))


def test_number_text_of_the_name_line_on_cpu_emulation(emu_lib):
    """put_dec / put_hex against one division per digit (tests/test_gpu_parity.py runs every 32-bit value): here the values around every power of ten and
    of sixteen, and a few million spread over the range."""
    import ctypes as C
    out = (C.c_uint64 * 4)()
    emu_lib.dwgsim_hip_selftest_text.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    emu_lib.dwgsim_hip_selftest_text.restype = C.c_int
    spans = [(0, 200000, 1), (0, 1 << 21, 2039), (0, 1 << 20, 0xFFF_FFF1_0003)]
    spans += [(10 ** k - 300, 600, 1) for k in range(3, 10)] + [(m * 10 ** 8 - 300, 600, 1) for m in (2, 9, 10, 11, 42)] + [((1 << 32) - 600, 1200, 1)]
    spans += [((1 << (4 * k)) - 300, 600, 1) for k in range(3, 16)] + [((1 << 64) - 600, 600, 1)]
    for first, n, stride in spans:
        assert emu_lib.dwgsim_hip_selftest_text(0, first, n, stride, out) == 0
        assert out[2] == n and (out[0], out[1]) == (0, 0), (first, n, stride, list(out))


@pytest.mark.parametrize("fasta,flags", EMU_CASES, ids=[f"{f}:{fl}" for f, fl in EMU_CASES])
def test_kernel_logic_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, fasta, flags):
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, fasta), flags, batch_pairs=700)


ION_HOMES = [{"ion_lds": 1}, {"ion_lds": 1, "split": 0}, {"ion_lds": 2}, {"ion_lds": 0}]      # as tests/test_gpu_parity.py


@pytest.mark.parametrize("home", ION_HOMES, ids=[",".join(f"{k}={v}" for k, v in h.items()) for h in ION_HOMES])
@pytest.mark.parametrize("fasta,flags,cap", [
    ("tiny.fa", "-z 9 -N 500 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 400 -2 0 -e 0.01 -y 0.1", 0),
    ("tiny.fa", "-z 9 -N 400 -c 2 -f TACG -1 100 -2 60 -e 0.2 -E 0.1 -d 300 -o 0", 0),
    ("odd.fa", "-z 6 -N 500 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 120 -2 0 -e 0.05 -n 10 -r 0.05 -R 0.5 -y 0.2 -A 2", 0),
    ("tiny.fa", "-z 9 -N 600 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 100 -2 0 -e 0.05 -y 0.1", 104),       # a starting capacity that reads outgrow: the batch runs again with twice the room
])
def test_ion_torrent_read_buffer_homes_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, fasta, flags, cap, home):
    """The Ion Torrent read buffers in LDS (two kernels / one), in LDS with the smaller blocks, in scratch slots: dw_read.hpp flow_errors, dw_host.cpp fill_sim_args."""
    res = compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, fasta), flags, batch_pairs=300, debug_options=dict(home, **({"flow_cap": cap} if cap else {})))
    if cap: assert res.flow_cap_mult >= 2


GROUPED_CASES = [      # the same jobs with all contigs resident together (one walk chain, batches that run across contig boundaries)
    ("ex1.fa", "-z 13 -N 1500"),
    ("tiny.fa", "-z 4 -N 1200 -r 0.02 -R 0.5 -I 30 -X 0.6"),
    ("odd.fa", "-z 3 -N 1200 -1 50 -2 50 -d 200 -s 20 -r 0.1 -R 1.0 -X 0.7 -n 50"),
    ("tiny.fa", "-z 9 -C 4 -2 0 -n 2 -y 0.3 -P pfx"),
    ("tiny.fa", "-z 9 -N 400 -c 2 -f TACG -1 100 -2 60 -e 0.2 -E 0.1 -d 300 -o 1"),
    ("tiny.fa", "-z 8 -N 700 -c 1 -1 50 -2 35 -d 300 -r 0.02 -R 0.5 -e 0.05 -E 0.03 -y 0.1"),
    ("tiny.fa", "-z 5 -N 600 -m {IN}/muts_edge.txt"),
    ("tiny.fa", "-z 5 -N 600 -b {IN}/muts_edge.bed"),
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -N 700"),
    ("tiny.fa", "-z 5 -x {IN}/regions_b.bed -C 4 -d 200 -s 10 -1 50 -2 50 -n 5"),
]


@pytest.mark.parametrize("fasta,flags", GROUPED_CASES, ids=[f"{f}:{fl}" for f, fl in GROUPED_CASES])
def test_contigs_resident_together_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, fasta, flags):
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, fasta), flags, batch_pairs=333, group_bp=1 << 30)


def write_many_contigs(path, n, seed, repeats=True):
    """n short contigs: random bases with homopolymers / tandem repeats (left-justification runs into contig ends) and N runs; some are
    too short to be simulated at all (skip rules), lengths are not multiples of anything"""
    import random
    rnd = random.Random(seed)
    with open(path, "w") as f:
        for k in range(n):
            l = rnd.choice([60, 150, 333, 700, 1024, 1500, 2100, 4096, 4097])
            seq = [rnd.choice("ACGT") for _ in range(l)]
            pos = 0
            while repeats and pos < l - 40:
                pos += rnd.randrange(10, 200)
                unit = rnd.choice(["A", "T", "CA", "GGC", "N"])
                reps = rnd.randrange(3, 25)
                for q, ch in enumerate((unit * reps)[: max(0, l - pos)]):
                    seq[pos + q] = ch
                pos += len(unit) * reps
            if k % 7 == 3:
                seq[:9] = "AAAAAAAAA"          # a homopolymer at the very start: shifts that would run off the contig
            if k % 5 == 2:
                seq[-7:] = "TTTTTTT"
            f.write(f">ctg{k}_{l}\n")
            for q in range(0, l, 61):
                f.write("".join(seq[q:q + 61]) + "\n")


@pytest.mark.parametrize("flags,group_bp", [
    ("-z 11 -C 6 -1 50 -2 50 -d 200 -s 15 -r 0.03 -R 0.6 -X 0.6 -n 8 -y 0.1", 1 << 30),
    ("-z 12 -N 4000 -1 60 -2 40 -d 220 -s 10 -r 0.05 -R 0.9 -X 0.7 -I 2 -n 20", 9000),         # several groups, -N remainder on the last contig
    ("-z 13 -C 3 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 80 -2 0 -e 0.02 -n 5 -r 0.02 -R 0.5", 1 << 30),
    ("-z 14 -C 3 -c 1 -1 40 -2 40 -d 150 -s 10 -r 0.04 -R 0.5 -n 10 -o 0", 20000),
])
def test_many_small_contigs_on_cpu_emulation(emu_lib, oracle_bin, tmp_path, flags, group_bp):
    """A scaffold-like job (dwgsim.c:519-625 loops over any number of contigs): 60 short contigs in groups -- one walk chain per group, the
    justification clusters, deletion runs and read windows must stop at every contig's own ends, read names, rand_ii and the abort rule's
    per-contig counter run on across the batch."""
    fa = str(tmp_path / "many.fa")
    write_many_contigs(fa, 60, seed=len(flags))
    compare_case(emu_lib, oracle_bin, fa, flags, batch_pairs=500, group_bp=group_bp)
    compare_case(emu_lib, oracle_bin, fa, flags, batch_pairs=1 << 20, group_bp=group_bp, debug_options={"justify_seq": 1})


def test_read_names_tell_the_truth_on_cpu_emulation(emu_lib, golden_dir):
    """Oracle-independent check of the name contract (see parity_common.check_read_names_tell_the_truth)."""
    from parity_common import check_read_names_tell_the_truth
    params = api.parse_flags("-z 21 -N 800 -r 0.01 -R 0.2 -e 0.01 -E 0.02", emu_lib)
    contigs = api.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    res = api.run_job(params, contigs, batch_pairs=700, lib=emu_lib)
    check_read_names_tell_the_truth(res, contigs, [params.length[0], params.length[1]], min_checked=100)


@pytest.mark.parametrize("fasta,flags", [
    ("tiny.fa", "-z 9 -C 3 -y 0.15 -n 0"),
    ("odd.fa", "-z 6 -C 3 -1 40 -2 40 -d 150 -s 10 -r 0.08 -R 0.8 -X 0.6 -n 1 -y 0.1"),
])
def test_count_random_matches_simulate_on_cpu_emulation(emu_lib, golden_dir, fasta, flags):
    from parity_common import check_count_random_matches_simulate
    check_count_random_matches_simulate(emu_lib, os.path.join(golden_dir, fasta), flags, ranges=((0, None), (17, 300)))


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("fasta,flags", [c for c in EMU_CASES if "-c " not in c[1]][:7], ids=lambda v: str(v))
def test_both_forms_of_the_illumina_read_kernel_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, fasta, flags, split):
    """k_simulate as one kernel (look-backs) and as two (first half | offsets | second half): see tests/test_gpu_parity.py"""
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, fasta), flags, batch_pairs=333, debug_options={"split": split})


@pytest.mark.parametrize("slots", [1, 2])
def test_ion_torrent_scratch_slots_change_hands_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, slots):
    """The Ion Torrent read buffers are scratch SLOTS taken and released by the blocks of an XCD (dw_simulate.hip scratch_slot_take): with one or
    two slots per XCD and 12 blocks (the emulation puts block b on XCD b % 8) a slot's second owner writes over its first owner's buffers."""
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 9 -N 3000 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 100 -2 0 -e 0.02 -y 0.05",
                 debug_options={"flow_slots": slots})


@pytest.mark.parametrize("flags,slots", [("-z 5 -N 130 -1 5200 -2 0 -n 100 -r 0.01 -R 0.3", 0), ("-z 5 -N 100 -c 1 -1 2500 -2 2000 -d 5800 -s 20 -n 60 -o 0", 1)])
def test_reads_beyond_the_lds_staging_limit_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, flags, slots):
    """The one-wave blocks stage their reads in scratch slots of global memory (no DWGSIM_HIP_ERR_UNSUP for long reads): see tests/test_gpu_parity.py"""
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), flags, debug_options={"flow_slots": slots} if slots else None)


def test_ion_torrent_read_outgrows_its_buffers_on_cpu_emulation(emu_lib, oracle_bin, golden_dir):
    """A read that outgrows its flow-space buffers makes the batch run again with twice the room (the reference doubles its buffers, dwgsim.c:296-311):
    forced here with a starting capacity of 104 bases for 100-base reads at e = 0.05; small batches, so that several batches meet the limit."""
    res = compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 9 -N 900 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 100 -2 0 -e 0.05 -y 0.1",
                       batch_pairs=300, debug_options={"flow_cap": 104})
    assert res.flow_cap_mult >= 2
    # ... and 17-base reads that grow to 2 059 bases (twelve empty flows in front of every T at e = 0.19): 128 x the starting capacity -- rounds 3-4 gave
    # up at 16 x, the reference keeps doubling (dwgsim.c:296-311)
    res = compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 12 -N 400 -c 2 -f TCG" + "A" * 12 + " -1 17 -2 0 -e 0.19", batch_pairs=300, debug_options={"flow_cap": 20})
    assert res.flow_cap_mult >= 128


def test_count_random_fast_and_long_path_on_cpu_emulation(emu_lib):
    from parity_common import check_count_random_fast_path
    check_count_random_fast_path(emu_lib, n=1200)


def test_walk_reruns_when_a_capacity_is_exceeded(emu_lib, oracle_bin, golden_dir):
    """The walk is enqueued with estimated capacities and checked once at the end; too small a candidate list or inserted-base
    pool must lead to an exact re-run with the same result."""
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 4 -N 600 -r 0.02 -R 0.5 -I 30 -X 0.6", batch_pairs=700, debug_options={"walk_cap": 7})


@pytest.mark.parametrize("opts", [{"site_slots": 0}, {"site_slots": 1}, {"site_slots": 1, "site_slot_cap": 3}, {"site_slots": 1, "site_slot_cap": 3, "walk_cap": 7}],
                         ids=["look-back", "slots", "slot-outgrown", "slot-and-list-outgrown"])
def test_both_forms_of_the_site_scan(emu_lib, oracle_bin, golden_dir, opts):
    """Candidate sites into the ordered list: every block into a slot of its own + scan + gather (round 6: no block waits for another), or one kernel with a
    decoupled look-back (re-runs, high mutation rates).  A block that outgrows its slot -- three entries here -- makes the host run the walk again through the
    look-back form; mutations and reads as the oracle's either way, also with several contigs in one group (contig boundaries inside a block's tiles)."""
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 4 -N 600 -r 0.02 -R 0.5 -I 3 -X 0.6", batch_pairs=700, debug_options=opts)
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "odd.fa"), "-z 3 -N 300 -1 50 -2 50 -d 200 -s 20 -r 0.1 -R 1.0 -X 0.7 -n 50 -y 0.1", batch_pairs=90, debug_options=opts, group_bp=1 << 30)


def test_walk_scans_in_their_segmented_form(emu_lib, oracle_bin, golden_dir):
    """k_scan4 / k_sufmin cut into 32 segments + fix-up kernels (what long contigs use), forced on a small one: mutations and reads as the oracle's."""
    try:
        compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 4 -N 600 -r 0.05 -R 0.6 -I 3 -X 0.5", batch_pairs=700, debug_options={"walk_seg_min": 1})
    finally:
        with api.Context(api.parse_flags("-z 1 -N 1", emu_lib), 0, emu_lib) as ctx:
            ctx.debug_option("walk_seg_min", 0)


@pytest.mark.parametrize("dirty_map", ["chunk", "word"])
def test_walking_a_contig_again_on_cpu_emulation(emu_lib, golden_dir, monkeypatch, dirty_map):
    """... with both mappings of k_dirty_chunks (a lane per chunk: small groups; a thread per bitmap word: groups beyond 128 Mb)"""
    from parity_common import check_walking_a_contig_again
    monkeypatch.setenv("DWGSIM_HIP_DIRTY_MAP", dirty_map)
    check_walking_a_contig_again(emu_lib, os.path.join(golden_dir, "tiny.fa"), "-z 4 -N 600 -r 0.02 -R 0.5 -I 3 -X 0.6")


def test_both_mappings_of_the_dirty_chunk_kernel_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, monkeypatch):
    """parity of a mutation-rich job and of count_random (the summaries k_place reads) with the thread-per-word form that large groups take"""
    monkeypatch.setenv("DWGSIM_HIP_DIRTY_MAP", "word")
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 8 -N 700 -1 50 -2 35 -d 300 -r 0.05 -R 0.5 -X 0.6 -y 0.1", batch_pairs=700)
    from parity_common import check_count_random_matches_simulate
    check_count_random_matches_simulate(emu_lib, os.path.join(golden_dir, "odd.fa"), "-z 6 -C 3 -1 40 -2 40 -d 150 -s 10 -r 0.08 -R 0.8 -X 0.6 -n 1 -y 0.1", ranges=((0, None), (17, 300)))


def test_abort_rule_matches_the_reference(emu_lib, oracle_bin, golden_dir):
    """Amplicon mode on a contig whose read-1 window always holds an N: every genomic attempt fails, only random reads come out and
    never reset the counter -- the reference dies at the 10 001st failure, and so must the HIP path (no pair exceeds the limit alone)."""
    from parity_common import check_both_abort
    check_both_abort(emu_lib, oracle_bin, os.path.join(golden_dir, "odd.fa"), "-z 6466 -1 33 -2 150 -d 900 -s 50 -N 1200 -r 0 -e 0.0-0.1 -Q 0 -a")
    # ... while the same job with fewer pairs stays under the limit and must match byte for byte
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "odd.fa"), "-z 6466 -1 33 -2 150 -d 900 -s 50 -N 600 -r 0 -e 0.0-0.1 -Q 0 -a", batch_pairs=100)


@pytest.mark.parametrize("fasta,flags,kw", [
    ("tiny.fa", "-z 9 -N 900 -P pfx -r 0.01 -R 0.3 -y 0.2", dict(devices=[0, 0, 0], gzip_on_gpu=False, batch_pairs=100, min_share=40)),
    ("tiny.fa", "-z 9 -N 300 -y 0.2 -o 1", dict(devices=[0, 0], gzip_on_gpu=True, batch_pairs=100, min_share=40)),
    ("tiny.fa", "-z 9 -C 3 -y 0.1", dict(devices=[0, 0, 0], gzip_on_gpu=False, batch_pairs=64, min_share=1, group_bp=5000)),      # contig by contig (groups of one)
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -N 700 -m {IN}/muts_edge.txt", dict(devices=[0, 0], gzip_on_gpu=False, batch_pairs=90, min_share=1)),
    ("odd.fa", "-z 6 -N 700 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 120 -2 0 -e 0.05 -n 10 -r 0.05 -R 0.5 -y 0.2", dict(devices=[0, 0, 0, 0], gzip_on_gpu=False, batch_pairs=77, min_share=1)),
    ("ex1.fa", "-z 13 -N 800 -M 2", dict(devices=[0, 0], gzip_on_gpu=False)),
    ("ex1.fa", "-z 13 -N 800 -M 1", dict(devices=[0], gzip_on_gpu=False, batch_pairs=300)),
])
def test_job_level_of_the_abi_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, fasta, flags, kw):
    """dwgsim_hip_job_* through ctypes: several contexts (here all on the one emulated device), batches dealt round-robin, counted random
    reads as rand_ii bases, ordered delivery -- every byte as the oracle's single-process run."""
    from parity_common import compare_job_api
    compare_job_api(emu_lib, oracle_bin, os.path.join(golden_dir, fasta), flags, **kw)


@pytest.mark.parametrize("fasta,flags,kw", [
    ("tiny.fa", "-z 9 -N 900 -P pfx -r 0.01 -R 0.3 -y 0.2", dict(devices=[0, 0, 0], gzip_on_gpu=False, batch_pairs=100, min_share=40)),
    ("tiny.fa", "-z 9 -N 300 -y 0.2 -o 1", dict(devices=[0, 0], gzip_on_gpu=True, batch_pairs=100, min_share=40)),
    ("tiny.fa", "-z 9 -C 3 -y 0.1", dict(devices=[0, 0, 0], gzip_on_gpu=False, batch_pairs=64, min_share=1, group_bp=5000)),
    ("odd.fa", "-z 6 -N 700 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 120 -2 0 -e 0.05 -n 10 -r 0.05 -R 0.5 -y 0.2", dict(devices=[0, 0, 0, 0], gzip_on_gpu=False, batch_pairs=77, min_share=1)),
    ("ex1.fa", "-z 13 -N 800 -M 1", dict(devices=[0], gzip_on_gpu=False, batch_pairs=300)),
])
def test_job_level_offset_sink_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, fasta, flags, kw):
    """dwgsim_hip_job_sink_t::reads_at (ABI 5): every device's batches handed over by threads of their own, each piece with its offset in the stream, in any
    order -- the pieces must tile each stream exactly (api.run_job_api checks: no gap, no overlap) and, put in order, be the oracle's bytes."""
    from parity_common import compare_job_api
    res = compare_job_api(emu_lib, oracle_bin, os.path.join(golden_dir, fasta), flags, offset_sink=True, **kw)
    assert res.delivery_threads >= 1


def test_job_level_offset_sink_in_the_shape_of_a_whole_node_on_cpu_emulation(emu_lib, oracle_bin, tmp_path):
    from parity_common import compare_job_api
    fa = str(tmp_path / "many.fa")
    write_many_contigs(fa, 60, seed=8)
    res = compare_job_api(emu_lib, oracle_bin, fa, "-z 12 -C 5 -1 60 -2 40 -d 220 -s 10 -r 0.02 -R 0.5 -n 10 -y 0.15", devices=[0] * 8, batch_pairs=23, min_share=1, group_bp=3000, offset_sink=True)
    assert res.delivery_threads > 3      # more deliverers than the ordered sink's one per stream


def test_job_level_many_small_contigs_on_cpu_emulation(emu_lib, oracle_bin, tmp_path):
    from parity_common import compare_job_api
    fa = str(tmp_path / "many.fa")
    write_many_contigs(fa, 60, seed=5)
    compare_job_api(emu_lib, oracle_bin, fa, "-z 11 -C 6 -1 50 -2 50 -d 200 -s 15 -r 0.03 -R 0.6 -X 0.6 -n 8 -y 0.1", devices=[0, 0, 0], gzip_on_gpu=False, batch_pairs=256, min_share=100, group_bp=30000)


def test_job_level_in_the_shape_of_a_whole_node_on_cpu_emulation(emu_lib, oracle_bin, tmp_path):
    """Eight workers, 24+ groups, tiny batches, three delivery threads: the shape of the 8-GPU node (here eight emulated contexts)."""
    from parity_common import compare_job_api
    fa = str(tmp_path / "many.fa")
    write_many_contigs(fa, 60, seed=8)
    compare_job_api(emu_lib, oracle_bin, fa, "-z 12 -C 5 -1 60 -2 40 -d 220 -s 10 -r 0.02 -R 0.5 -n 10 -y 0.15", devices=[0] * 8, batch_pairs=23, min_share=1, group_bp=3000)


def test_job_level_abort_rule_across_devices_on_cpu_emulation(emu_lib, oracle_bin, golden_dir):
    """The failure counter over the pairs of a contig (dwgsim.c:635) when its batches are dealt to three contexts: the joined summaries give the
    reference's verdict -- no abort at 600 pairs, abort at 1200."""
    from parity_common import compare_job_api
    fa = os.path.join(golden_dir, "odd.fa")
    compare_job_api(emu_lib, oracle_bin, fa, "-z 6466 -1 33 -2 150 -d 900 -s 50 -N 600 -r 0 -e 0.0-0.1 -Q 0 -a", devices=[0, 0, 0], gzip_on_gpu=False, batch_pairs=50, min_share=1)
    with pytest.raises(api.DwgsimError, match="failed to generate a read after 10001 trials"):
        api.run_job_api(api.parse_flags("-z 6466 -1 33 -2 150 -d 900 -s 50 -N 1200 -r 0 -e 0.0-0.1 -Q 0 -a", emu_lib), api.read_fasta(fa), devices=[0, 0, 0], gzip_on_gpu=False, batch_pairs=50, min_share=1, lib=emu_lib)


def test_abort_rule_counts_per_contig_inside_a_group(emu_lib, oracle_bin, golden_dir, tmp_path):
    """`int num_failed = 0` sits inside the contig loop (dwgsim.c:635): a job whose failures add up to more than 10 000 over TWO contigs, but
    not inside either, runs to the end.  With both contigs in one group -- one launch across the boundary -- the counter must still start
    again at the second contig; and it must still abort when one contig alone passes the limit."""
    from parity_common import check_both_abort
    src = api.read_fasta(os.path.join(golden_dir, "odd.fa"))
    fa = str(tmp_path / "twice.fa")
    with open(fa, "w") as f:
        for tag in ("a", "b"):
            for name, arr in src:
                if name != "tiny":
                    f.write(f">{name}{tag}\n{bytes(arr).decode()}\n")
    flags = "-z 6466 -1 33 -2 150 -d 900 -s 50 -N 1600 -r 0 -e 0.0-0.1 -Q 0 -a"
    res = compare_case(emu_lib, oracle_bin, fa, flags, batch_pairs=1 << 20, group_bp=1 << 30)
    assert res.n_retries > 10000
    compare_case(emu_lib, oracle_bin, fa, flags, batch_pairs=250, group_bp=1 << 30)
    check_both_abort(emu_lib, oracle_bin, fa, flags.replace("-N 1600", "-N 2400"), group_bp=1 << 30)


def test_hopeless_target_regions_end_with_an_error(emu_lib, golden_dir, tmp_path):
    """Regions that pass the length checks but can never hold a fragment: the reference spins forever (dwgsim.c:677-713); here the
    placement gives up after 2^20 tries, the rest of the batch stops early, and the call returns an error."""
    bed = tmp_path / "r.bed"
    bed.write_text("t1\t100\t500\nt1\t900\t1300\n")
    with pytest.raises(api.DwgsimError, match="no fragment placement satisfied the target regions"):
        api.run_job(api.parse_flags(f"-z 3 -N 130 -1 50 -2 50 -d 500 -s 5 -x {bed}", emu_lib), api.read_fasta(os.path.join(golden_dir, "tiny.fa")), lib=emu_lib)


@pytest.mark.parametrize("fasta,flags,gz", [
    ("tiny.fa", "-z 9 -N 360 -P pfx -r 0.01 -R 0.3 -y 0.2", "gpu"),       # (the gzip kernel is slow under the emulation: thousands of barriers per member)
    ("tiny.fa", "-z 9 -N 900 -P pfx -r 0.01 -R 0.3 -y 0.2", "cpu"),
    ("tiny.fa", "-z 9 -N 500 -c 2 -f TACG -1 100 -2 60 -e 0.05 -E 0.02 -d 300", "cpu"),
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -C 3 -m {IN}/muts_edge.txt -o 1", "cpu"),
])
def test_command_line_on_several_contexts_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, tmp_path, fasta, flags, gz):
    """The dwgsim-hip host code (three contexts on host threads, read-index ranges with count_random bases, two batches in flight per
    context, deflate pool, ordered merge) linked against the emulated library: the five files equal the oracle's after gunzip."""
    import gzip
    from parity_common import run_oracle, IN_DIR
    flags = flags.replace("{IN}", IN_DIR)
    want = run_oracle(oracle_bin, os.path.join(golden_dir, fasta), flags, str(tmp_path))
    env = dict(os.environ, DWGSIM_HIP_DEVICES="0,0,0", DWGSIM_HIP_MIN_SHARE="40", DWGSIM_HIP_BATCH="150", DWGSIM_HIP_THREADS="3", DWGSIM_HIP_GZIP=gz)
    subprocess.run([os.path.join(HERE, "emu", "dwgsim-emu")] + flags.split() + [os.path.join(golden_dir, fasta), str(tmp_path / "cli")], check=True, stderr=subprocess.DEVNULL, env=env)
    for k, suf in [(0, "bwa.read1.fastq.gz"), (1, "bwa.read2.fastq.gz"), (2, "bfast.fastq.gz")]:
        p = str(tmp_path / ("cli." + suf))
        assert (gzip.open(p, "rb").read() if os.path.exists(p) else b"") == want[k], suf
    assert open(str(tmp_path / "cli.mutations.txt"), "rb").read() == want["txt"]
    assert open(str(tmp_path / "cli.mutations.vcf"), "rb").read() == want["vcf"]


def test_solo_rank_of_the_job_level_makes_exactly_that_devices_share(emu_lib, oracle_bin, golden_dir, tmp_path):
    """DWGSIM_HIP_SOLO=r/W (measurement aid, dw_job.cpp): ONE device plays device r of a W-device job.  The records the W solo runs make are, together, the
    records of the whole job -- every batch made exactly once, none twice -- apart from the running number in the names of random reads (the other devices'
    random-read counts are taken as zero in a solo run); only device 0 writes the mutation files.  Several contigs, groups the devices share unequally."""
    import gzip
    from parity_common import run_oracle
    flags = "-z 9 -N 3000 -y 0.2 -r 0.02 -R 0.5 -o 1"
    fa = os.path.join(golden_dir, "tiny.fa")
    want = run_oracle(oracle_bin, fa, flags, str(tmp_path))

    def recs(b):
        L = b.split(b"\n")
        out = [b"\n".join(L[k:k + 4]) for k in range(0, len(L) - 1, 4)]
        return [r if not r.startswith(b"@rand") else b"@rand" + r.split(b"\n", 1)[1] for r in out]
    for W in (2, 3):
        got = {0: [], 1: []}
        for r in range(W):
            env = dict(os.environ, DWGSIM_HIP_SOLO=f"{r}/{W}", DWGSIM_HIP_MIN_SHARE="20", DWGSIM_HIP_BATCH="170", DWGSIM_HIP_GZIP="cpu", DWGSIM_HIP_THREADS="2")
            subprocess.run([os.path.join(HERE, "emu", "dwgsim-emu")] + flags.split() + [fa, str(tmp_path / f"s{W}_{r}")], check=True, stderr=subprocess.DEVNULL, env=env, timeout=300)
            for k, suf in [(0, "bwa.read1.fastq.gz"), (1, "bwa.read2.fastq.gz")]:
                got[k] += recs(gzip.open(str(tmp_path / f"s{W}_{r}.{suf}"), "rb").read())
            txt = open(str(tmp_path / f"s{W}_{r}.mutations.txt"), "rb").read()
            assert txt == (want["txt"] if r == 0 else b""), (W, r)
        for k in (0, 1):
            assert sorted(got[k]) == sorted(recs(want[k])), (W, k)


@pytest.mark.parametrize("k", range(6))
def test_both_record_writers_on_cpu_emulation(emu_lib, oracle_bin, tmp_path, k):
    from parity_common import WRITER_CASES, check_record_writers
    check_record_writers(emu_lib, oracle_bin, str(tmp_path), *WRITER_CASES[k])


def test_gzip_members_made_by_the_kernels_on_cpu_emulation(emu_lib, golden_dir):
    from parity_common import check_gpu_gzip
    check_gpu_gzip(emu_lib, os.path.join(golden_dir, "tiny.fa"), "-z 9 -N 900 -1 70 -2 50 -r 0.01 -y 0.1", sizes=(250, 1))


def test_gzip_kernel_on_hard_inputs_on_cpu_emulation(emu_lib):
    from parity_common import check_gzip_kernel_on_hard_inputs
    check_gzip_kernel_on_hard_inputs(emu_lib)


def test_mut_debug_aborts_on_cpu_emulation(emu_lib, oracle_bin, golden_dir):
    from parity_common import check_mut_debug_aborts
    check_mut_debug_aborts(emu_lib, oracle_bin, golden_dir)


def test_command_line_honours_a_fai_index_as_the_reference_does(emu_lib, oracle_bin, golden_dir, tmp_path):
    """dwgsim.c:465-478: with <in.fa>.fai present, the contig table (VCF header, total length, number of contigs -- hence pairs per contig
    and which contig "is the last") comes from the index, not from the FASTA.  An index that disagrees with the FASTA shows it."""
    import gzip, shutil
    from parity_common import run_oracle
    fa = str(tmp_path / "g.fa")
    shutil.copy(os.path.join(golden_dir, "tiny.fa"), fa)
    open(fa + ".fai", "w").write("t1\t5000\t4\t60\t61\nt2\t4000\t6110\t60\t61\nshort\t300\t10180\t60\t61\nphantom\t2500\t10500\t60\t61\n")
    flags = "-z 11 -N 1500 -1 50 -2 50 -d 200 -s 10 -o 1"
    want = run_oracle(oracle_bin, fa, flags, str(tmp_path))
    subprocess.run([os.path.join(HERE, "emu", "dwgsim-emu")] + flags.split() + [fa, str(tmp_path / "cli")], check=True, stderr=subprocess.DEVNULL,
                   env=dict(os.environ, DWGSIM_HIP_THREADS="2", DWGSIM_HIP_GZIP="cpu"))
    for k, suf in [(0, "bwa.read1.fastq.gz"), (1, "bwa.read2.fastq.gz")]:
        assert gzip.open(str(tmp_path / ("cli." + suf)), "rb").read() == want[k], suf
    assert open(str(tmp_path / "cli.mutations.vcf"), "rb").read() == want["vcf"] and b"phantom" in want["vcf"]
    assert open(str(tmp_path / "cli.mutations.txt"), "rb").read() == want["txt"]


@pytest.mark.parametrize("threads,chunk,fai", [(1, 0, False), (3, 64, False), (16, 7, False), (3, 100, True), (16, 1, True)])
def test_command_line_reads_an_awkward_fasta_as_the_reference_does(emu_lib, oracle_bin, tmp_path, threads, chunk, fai):
    """mut.c:49-87 seq_read_fasta: text before the first '>', CR LF, blank lines, lower case, '-' and '.', digits and blanks inside the
    sequence, a '>' in the middle of a line, lines of every length around the reader's eight-byte steps, bytes with the high bit set, no
    newline at the end.  The command line's reader (mapped file, memchr, eight letters verified at a time) against the oracle's."""
    import gzip, random
    from parity_common import run_oracle
    rng = random.Random(5)
    def seq(n): return "".join(rng.choice("ACGT") for _ in range(n))
    body = ["junk before the first record", ">c1 first comment\r"]
    for L in list(range(1, 20)) + [60, 61, 64, 65]:
        body.append(seq(L) + ("\r" if L % 3 == 0 else ""))
    body += ["", seq(30).lower(), seq(10) + "-." + seq(7), seq(9) + " 12 " + seq(9), seq(8) + "\xe9" + seq(8), seq(40)]
    body += [">c2\tsecond", seq(700), seq(33) + ">c3 opens mid-line", seq(900), ">c5 with a > in its header >x", seq(450)]
    body += [">c4"] + [seq(61) for _ in range(30)]
    # regular records (what every FASTA writer produces: the path that is verified and copied by all cores), ending in every way they can
    body += [">r1 regular, last line shorter"] + [seq(60) for _ in range(40)] + [seq(17)]
    body += [">r2 regular, all lines full"] + [seq(70) for _ in range(25)]
    body += [">r3 one line", seq(333)]
    body += [">r4 a ragged line in the middle"] + [seq(50) for _ in range(20)] + [seq(49)] + [seq(50) for _ in range(20)]
    body += [">r5 a digit among the letters"] + [seq(50) for _ in range(10)] + [seq(25) + "7" + seq(24)] + [seq(50) for _ in range(10)]
    body += [">r6 empty", ">r7"] + [seq(80) for _ in range(12)]
    fa = str(tmp_path / "awkward.fa")
    open(fa, "w", encoding="latin-1").write("\n".join(body) + "\n" + seq(50))      # (no newline at the end)
    flags = "-z 21 -N 1200 -1 50 -2 50 -d 200 -s 10 -o 1 -n 50"
    if fai:      # with an index the records are parsed straight into the job's staging, one after the other (the index's own numbers are the reference's
                 # contig table, dwgsim.c:465-478: here they are the true ones)
        from dwgsim_amd import api as _api
        open(fa + ".fai", "w").write("".join(f"{n}\t{len(a)}\t0\t60\t61\n" for n, a in _api.read_fasta(fa)))
    env = dict(os.environ, DWGSIM_HIP_THREADS="2", DWGSIM_HIP_GZIP="cpu", DWGSIM_HIP_READ_THREADS=str(threads))
    if chunk:
        env["DWGSIM_HIP_READ_CHUNK"] = str(chunk)
    want = run_oracle(oracle_bin, fa, flags, str(tmp_path))
    subprocess.run([os.path.join(HERE, "emu", "dwgsim-emu")] + flags.split() + [fa, str(tmp_path / "cli")], check=True, stderr=subprocess.DEVNULL, env=env)
    for k, suf in [(0, "bwa.read1.fastq.gz"), (1, "bwa.read2.fastq.gz")]:
        assert gzip.open(str(tmp_path / ("cli." + suf)), "rb").read() == want[k], suf
    assert open(str(tmp_path / "cli.mutations.vcf"), "rb").read() == want["vcf"] and b"c3" in want["vcf"]
    assert open(str(tmp_path / "cli.mutations.txt"), "rb").read() == want["txt"]
    # the same file through a pipe (no mapping: the chunked path of the reader)
    with open(fa, "rb") as f:
        subprocess.run([os.path.join(HERE, "emu", "dwgsim-emu")] + flags.split() + ["-", str(tmp_path / "pipe")], check=True, stdin=f, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, DWGSIM_HIP_THREADS="2", DWGSIM_HIP_GZIP="cpu"))
    assert gzip.open(str(tmp_path / "pipe.bwa.read1.fastq.gz"), "rb").read() == want[0]


@pytest.mark.parametrize("seed,count,mode", [(201, 160, ""), (202, 100, "inputs"), (203, 60, "cli"), (204, 30, "inputs cli"), (205, 100, "shards"), (206, 60, "inputs shards")])
def test_random_option_sets_on_cpu_emulation(emu_lib, oracle_bin, seed, count, mode):
    """tests/fuzz_flags.py (what test_gpu_fuzz.py runs on the GPU) through the emulated library and command line: random option combinations --
    all three technologies, every length from 1, ramps, mutation inputs, regions, read-index ranges simulated out of order -- against the
    oracle, byte for byte; option sets the oracle itself rejects are skipped.  (-B is left to its own cases above: its calibration runs 10^5 reads
    through the flow model, minutes under emulation.)"""
    env = dict(os.environ, DWGSIM_HIP_LIB=os.path.join(HERE, "emu", "libdwgsim_emu.so"), DWGSIM_HIP_CLI=os.path.join(HERE, "emu", "dwgsim-emu"), DWGSIM_HIP_GZIP="cpu",
               DWGSIM_FUZZ_ORACLE_TIMEOUT="3", DWGSIM_FUZZ_NO_B="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_flags.py"), str(seed), str(count)] + mode.split(), capture_output=True, text=True, timeout=1400, env=env)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and last.endswith(" 0 bad"), r.stdout[-3000:]


INDEPENDENT_THREADS = "k_jrun,k_apply,k_resolve,k_jreach,k_jbound,k_events,k_apply_patches,k_gather,k_pack,k_compact,k_make_view,k_collect_mask"


def test_independent_threads_in_reverse_order_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, tmp_path, monkeypatch):
    """Kernels whose threads are meant to be independent -- above all k_jrun, one cluster of the left-justification per thread -- with their
    blocks and lanes run from the last to the first (HIPEMU_REVERSE, tests/emu/hip_emu.cpp).  In index order an overlap between two clusters
    looks like the sequential algorithm and stays hidden; this order shows it, as the GPU does.  The case the GPU found, dense indels in
    homopolymers and tandem repeats, and 150 of the fuzzer's option sets at high mutation rates."""
    monkeypatch.setenv("HIPEMU_REVERSE", INDEPENDENT_THREADS)
    compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, "odd.fa"), "-z 8384 -1 7 -2 1 -d 900 -s 1 -C 0.5 -r 0.3 -y 0.3 -n 1000 -S 1 -H -o 1")
    fa = str(tmp_path / "many.fa")
    write_many_contigs(fa, 40, 77)
    for flags in ("-z 21 -N 3000 -1 40 -2 40 -d 150 -s 10 -r 0.2 -R 0.7 -X 0.6 -n 40", "-z 22 -N 3000 -1 40 -2 40 -d 150 -s 10 -r 0.3 -R 0.3 -X 0.3 -n 40 -H",
                  "-z 23 -N 2000 -1 30 -2 30 -d 120 -s 5 -r 0.1 -R 1.0 -X 0.9 -I 3 -n 40"):
        compare_case(emu_lib, oracle_bin, fa, flags, group_bp=1 << 30)
    env = dict(os.environ, DWGSIM_HIP_LIB=os.path.join(HERE, "emu", "libdwgsim_emu.so"), DWGSIM_FUZZ_ORACLE_TIMEOUT="3", DWGSIM_FUZZ_NO_B="1", DWGSIM_FUZZ_MUT="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_flags.py"), "207", "150"], capture_output=True, text=True, timeout=1400, env=env)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].endswith(" 0 bad"), r.stdout[-3000:]


# every kernel whose blocks do not wait for one another: the whole mutation walk, the random-read count, the scans, the abort rule, the calibration
# (k_simulate and k_gzip in their single-kernel forms look back at the blocks in front of them: their LANES are reversed, HIPEMU_REVERSE_LANES)
ORDER_FREE_KERNELS = ("k_pack,k_site_scan,k_site_scan_slots,k_slot_scan,k_slot_gather,k_mark_dirty,k_dirty_chunks,k_scan_excl,k_compact,k_events,k_resolve,k_scan4,k_scan4_fix,k_apply,k_jreach,k_sufmin,k_sufmin_fix,k_jbound,k_jrun,k_apply_patches,"
                      "k_collect_mask,k_gather,k_mut_debug,k_make_view,k_place,k_place_rest,k_range_counts,k_split_scan1,k_split_scan2,k_failrule_a,k_failrule_b,k_calibrate")
DENSE_CASES = [      # dense indels, long insertions, homopolymers and tandem repeats, N runs, file-driven mutations, regions: what makes the threads of the walk meet
    ("odd.fa", "-z 8384 -1 7 -2 1 -d 900 -s 1 -C 0.5 -r 0.3 -y 0.3 -n 1000 -S 1 -H -o 1", 0),
    ("odd.fa", "-z 3 -N 1200 -1 50 -2 50 -d 200 -s 20 -r 0.1 -R 1.0 -X 0.7 -n 50", 0),
    ("tiny.fa", "-z 4 -N 1200 -r 0.02 -R 0.5 -I 30 -X 0.6", 0),
    ("tiny.fa", "-z 5 -N 600 -m {IN}/muts_edge.txt", 0),
    ("tiny.fa", "-z 5 -N 600 -v {IN}/muts_edge.vcf -H", 0),
    ("tiny.fa", "-z 5 -N 600 -b {IN}/muts_edge.bed", 0),
    ("tiny.fa", "-z 5 -x {IN}/regions_b.bed -C 4 -d 200 -s 10 -1 50 -2 50 -n 5 -r 0.05 -R 0.5", 0),
    ("tiny.fa", "-z 9 -N 400 -c 2 -f TACG -1 100 -2 60 -e 0.2 -E 0.1 -d 300 -o 1 -r 0.05 -R 0.6", 0),
    ("tiny.fa", "-z 8 -N 700 -c 1 -1 50 -2 35 -d 300 -r 0.05 -R 0.5 -e 0.05 -E 0.03 -y 0.1", 0),
    ("many", "-z 21 -N 3000 -1 40 -2 40 -d 150 -s 10 -r 0.2 -R 0.7 -X 0.6 -n 40", 1 << 30),
    ("many", "-z 22 -N 3000 -1 40 -2 40 -d 150 -s 10 -r 0.3 -R 0.3 -X 0.3 -n 40 -H", 1 << 30),
    ("many", "-z 23 -N 2000 -1 30 -2 30 -d 120 -s 5 -r 0.1 -R 1.0 -X 0.9 -I 3 -n 40", 9000),
]


@pytest.mark.parametrize("fasta,flags,group_bp", DENSE_CASES, ids=[f"{f}:{fl}" for f, fl, _ in DENSE_CASES])
def test_every_order_free_kernel_in_reverse_order_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, tmp_path, monkeypatch, fasta, flags, group_bp):
    """ALWAYS both orders: every dense-indel case runs in index order elsewhere in this file, and here with the blocks AND lanes of every kernel
    that has no look-back run from the last to the first, and the lanes of the two that have one reversed too.  Threads that are meant to be
    independent have to be: in index order an overlap between two of them looks like the sequential algorithm and stays hidden (it did for two
    rounds: dw_walk.hip reach_del)."""
    monkeypatch.setenv("HIPEMU_REVERSE", ORDER_FREE_KERNELS)
    monkeypatch.setenv("HIPEMU_REVERSE_LANES", "all")
    if fasta == "many":
        fa = str(tmp_path / "many.fa")
        write_many_contigs(fa, 40, 77)
    else:
        fa = os.path.join(golden_dir, fasta)
    compare_case(emu_lib, oracle_bin, fa, flags, batch_pairs=700, group_bp=group_bp)
    if "-c" not in flags:      # the Illumina read kernel also in its single-kernel form (look-backs; blocks in order, lanes reversed)
        compare_case(emu_lib, oracle_bin, fa, flags, batch_pairs=700, group_bp=group_bp, debug_options={"split": 0})


def test_lanes_in_reverse_order_on_cpu_emulation(emu_lib, oracle_bin, golden_dir, monkeypatch):
    """Every kernel with the lanes of a block run from the last to the first (HIPEMU_REVERSE_LANES=all; blocks stay in order for the look-backs):
    nothing may count on the lock step of a wave where no wave operation enforces it."""
    from parity_common import CASES
    monkeypatch.setenv("HIPEMU_REVERSE_LANES", "all")
    picked = [c for c in CASES if any(t in c[1] for t in ("-z 13 -N 10000 -1 100", "-I 30", "-c 1 -1 40", "-e 0.05 -n 10", "-o 2 -q 5", "-z 8384"))]
    assert len(picked) >= 5
    import re
    for fasta, flags in picked:
        compare_case(emu_lib, oracle_bin, os.path.join(golden_dir, fasta), re.sub(r"-N \d+", "-N 1200", flags))


def test_ion_torrent_random_flow_orders_on_cpu_emulation(emu_lib, oracle_bin):
    """tests/fuzz_ion_flows.py (the GPU suite runs it too): the flow model under flow orders of 4 .. 64 flows with long gaps, read lengths 1 .. 400,
    per-flow error rates up to 0.2; at most one case may end with a read that outgrew its buffers."""
    env = dict(os.environ, DWGSIM_HIP_LIB=os.path.join(HERE, "emu", "libdwgsim_emu.so"), DWGSIM_FUZZ_ORACLE_TIMEOUT="3", DWGSIM_FUZZ_NO_B="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_ion_flows.py"), "32", "80"], capture_output=True, text=True, timeout=1400, env=env)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0 and last.endswith(" 0 bad"), r.stdout[-3000:]
    assert int(last.split(" outgrew")[0].split()[-1]) <= 1, last      # (buffers are doubled up to 16 x before a case may count as that)


@pytest.mark.parametrize("flags", ["-z 3 -1 150 -2 150", "-z 3 -c 1 -1 50 -2 50"])
def test_a_call_whose_look_back_word_would_overflow_is_refused_on_cpu_emulation(emu_lib, golden_dir, flags):
    """The single Illumina / SOLiD kernel carries the random reads and the bytes of the first stream in front of a block in ONE 62-bit word (dw_simulate.hip ONE_LB);
    a call with so many pairs that the two sums could not share it (2^27 pairs of 2 x 150 bp: 90 GB of text per stream) is refused before anything is allocated
    (include/dwgsim_hip.h dwgsim_hip_simulate_ranges_async).  A read-index range is not bound by the contig's length: the tiny contig will do."""
    params = api.parse_flags(flags, emu_lib)
    contigs = api.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    with api.Context(params, 0, emu_lib) as ctx:
        h0 = ctx.add_contigs(contigs[:1], indices=[0])
        ctx.mutate(h0)
        with pytest.raises(api.DwgsimError, match="too many pairs in one call"):
            ctx.simulate_ranges([(h0, 0, 1 << 34)], 0, 0)
        b = ctx.simulate_ranges([(h0, 0, 100)], 0, 0)      # (and the context is still good)
        assert b.n_pairs == 100
