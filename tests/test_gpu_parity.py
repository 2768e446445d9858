"""-m gpu: the HIP path (libdwgsim_hip.so through the C-ABI) against the oracle in Philox mode,
bit-exact on FASTQ text (names, bases, qualities) and on mutations.txt / .vcf."""
import os
import pytest

from dwgsim_amd import api
FLOW_ORDER = "TACGTACGTCTGAGCATCGATCGATGTACAGC"
from parity_common import CASES, FLOW, compare_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return api.load()


@pytest.mark.parametrize("fasta,flags", CASES, ids=[f"{f}:{fl}" for f, fl in CASES])
def test_bit_exact_vs_oracle(lib, oracle_bin, golden_dir, fasta, flags):
    compare_case(lib, oracle_bin, os.path.join(golden_dir, fasta), flags)


GROUPABLE = [c for c in CASES if " -B" not in c[1] and "-1 1500" not in c[1] and "-1 1300" not in c[1] and "-1 1400" not in c[1]]


@pytest.mark.parametrize("fasta,flags", GROUPABLE, ids=[f"{f}:{fl}" for f, fl in GROUPABLE])
def test_bit_exact_vs_oracle_with_the_contigs_resident_together(lib, oracle_bin, golden_dir, fasta, flags):
    """The same option surface with all contigs of the FASTA in ONE group (dwgsim_hip_add_contigs): one chain of walk kernels for all of
    them, batches of 777 pairs that run across the contig boundaries (dwgsim_hip_simulate_ranges_async)."""
    compare_case(lib, oracle_bin, os.path.join(golden_dir, fasta), flags, batch_pairs=777, group_bp=1 << 30)


@pytest.fixture(scope="module")
def many_fa(tmp_path_factory):
    from test_emu_parity import write_many_contigs
    p = str(tmp_path_factory.mktemp("many") / "many.fa")
    write_many_contigs(p, 200, seed=77)
    return p


@pytest.mark.parametrize("flags,group_bp,batch", [
    ("-z 11 -C 20 -1 50 -2 50 -d 200 -s 15 -r 0.03 -R 0.6 -X 0.6 -n 8 -y 0.1", 1 << 30, 1 << 22),
    ("-z 12 -N 60000 -1 60 -2 40 -d 220 -s 10 -r 0.05 -R 0.9 -X 0.7 -I 2 -n 20", 50000, 4099),      # several groups, -N remainder on the last contig
    ("-z 13 -C 10 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 80 -2 0 -e 0.02 -n 5 -r 0.02 -R 0.5", 1 << 30, 1 << 22),
    ("-z 14 -C 10 -c 1 -1 40 -2 40 -d 150 -s 10 -r 0.04 -R 0.5 -n 10 -o 0", 100000, 1 << 22),
    ("-z 15 -C 15 -1 150 -2 150 -o 1", 1 << 30, 1 << 22),
])
def test_two_hundred_small_contigs(lib, oracle_bin, many_fa, flags, group_bp, batch):
    """A scaffold-like job (dwgsim.c:519-625 loops over any number of contigs at no fixed cost): 200 short contigs in groups, every byte as
    the oracle writes it -- left-justification, deletion runs and read windows stop at every contig's own ends; read names, rand_ii and the
    abort rule's per-contig counter run on across a launch."""
    compare_case(lib, oracle_bin, many_fa, flags, batch_pairs=batch, group_bp=group_bp)


def test_abort_rule_counts_per_contig_inside_a_group(lib, oracle_bin, golden_dir, tmp_path):
    from test_emu_parity import test_abort_rule_counts_per_contig_inside_a_group as body
    body(lib, oracle_bin, golden_dir, tmp_path)


def test_batches_and_shards_are_order_independent(lib, oracle_bin, golden_dir):
    """Read-index ranges are independent: tiny batches (many simulate() calls, rand_base chained by the
    host) give the same bytes as one call -- the property multi-GPU sharding relies on."""
    compare_case(lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 9 -N 3000 -y 0.2", batch_pairs=257)


@pytest.mark.parametrize("flags", ["-z 21 -N 4000 -r 0.01 -R 0.2 -e 0.01 -E 0.02", "-z 22 -N 3000 -r 0 -e 0 -E 0 -1 70 -2 50 -S 1 -y 0.1"])
def test_read_names_tell_the_truth(lib, golden_dir, flags):
    """Oracle-independent: every read is found where its name says it is (the contract dwgsim_eval relies on)."""
    from parity_common import check_read_names_tell_the_truth
    params = api.parse_flags(flags, lib)
    contigs = api.read_fasta(os.path.join(golden_dir, "tiny.fa"))
    res = api.run_job(params, contigs, lib=lib)
    check_read_names_tell_the_truth(res, contigs, [params.length[0], params.length[1]])


@pytest.fixture(scope="module")
def repeats_fa(tmp_path_factory):
    from dwgsim_amd import synth
    p = str(tmp_path_factory.mktemp("rep") / "repeats.fa")
    synth.write_fasta(p, synth.workload_contigs("repeats"))
    return p


@pytest.mark.parametrize("flags", [
    "-z 31 -M 2 -r 0.02 -R 0.6 -X 0.5",
    "-z 32 -M 2 -r 0.2 -R 0.9 -X 0.3 -I 2",
    "-z 33 -M 2 -r 0.05 -R 1.0 -X 0.8 -H",
    "-z 34 -N 20000 -1 100 -2 100 -r 0.03 -R 0.5 -X 0.4 -n 5",
])
def test_mutation_walk_on_repeat_rich_contigs(lib, oracle_bin, repeats_fa, flags):
    """1.8 Mb of homopolymers / tandem repeats / N blocks at high indel rates: the cluster-parallel
    left-justification must reproduce the oracle's sequential pass exactly."""
    compare_case(lib, oracle_bin, repeats_fa, flags)


def test_parallel_justify_equals_sequential_crosscheck(lib, repeats_fa):
    params = api.parse_flags("-z 35 -M 2 -r 0.1 -R 0.8 -X 0.6", lib)
    contigs = api.read_fasta(repeats_fa)
    par = api.run_job(params, contigs, lib=lib)
    seq = api.run_job(params, contigs, lib=lib, debug_options={"justify_seq": 1})
    assert par.mutations_txt == seq.mutations_txt and par.mutations_vcf == seq.mutations_vcf
    assert len(par.mutations_txt) > 100000


def test_cli_is_a_drop_in_for_the_dwgsim_command(oracle_bin, golden_dir, tmp_path):
    """dwgsim-hip <options> ref.fa prefix writes the reference's five files; after gunzip they equal the
    oracle's (Philox mode) outputs, as the reference's own test compares them (testdata/test.sh:21-26)."""
    import gzip, subprocess
    from parity_common import run_oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "dwgsim_amd", "dwgsim-hip")
    fasta, flags = os.path.join(golden_dir, "tiny.fa"), "-z 9 -N 3000 -P pfx -r 0.01 -R 0.3"
    want = run_oracle(oracle_bin, fasta, flags, str(tmp_path))
    subprocess.run([cli] + flags.split() + [fasta, str(tmp_path / "cli")], check=True, stderr=subprocess.DEVNULL)
    for k, suf in [(0, "bwa.read1.fastq.gz"), (1, "bwa.read2.fastq.gz"), (2, "bfast.fastq.gz")]:
        assert gzip.open(str(tmp_path / ("cli." + suf)), "rb").read() == want[k]
    assert open(str(tmp_path / "cli.mutations.txt"), "rb").read() == want["txt"]
    assert open(str(tmp_path / "cli.mutations.vcf"), "rb").read() == want["vcf"]


@pytest.mark.parametrize("fasta,flags", [
    ("tiny.fa", "-z 9 -N 3000 -P pfx -r 0.01 -R 0.3 -y 0.2"),
    ("tiny.fa", "-z 8 -N 4000 -c 1 -1 50 -2 35 -d 300 -r 0.02 -R 0.5 -e 0.05 -E 0.03 -y 0.1"),
    ("tiny.fa", f"-z 9 -N 2000 -c 2 -f {FLOW_ORDER} -1 200 -2 100 -e 0.02 -E 0.03 -d 600"),
    ("odd.fa", "-z 6 -C 30 -2 0 -1 120 -r 0.05 -R 0.9 -I 40 -y 0.3 -n 3"),
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -C 8 -m {IN}/muts_edge.txt"),
])
@pytest.mark.parametrize("gz", ["gpu", "cpu"])
def test_cli_on_several_contexts_writes_the_same_files(oracle_bin, golden_dir, tmp_path, fasta, flags, gz):
    """dwgsim-hip with three contexts (here all on GPU 0: the multi-GPU code path on a 1-GPU box), tiny batches and every contig
    split into read-index ranges: host threads, per-range rand_ii bases from count_random, ordered merge of the deflated
    members -- the five files must equal the oracle's (= the single-context run's) after gunzip."""
    import gzip, subprocess
    from parity_common import run_oracle, IN_DIR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "dwgsim_amd", "dwgsim-hip")
    flags = flags.replace("{IN}", IN_DIR)
    want = run_oracle(oracle_bin, os.path.join(golden_dir, fasta), flags, str(tmp_path))
    env = dict(os.environ, DWGSIM_HIP_DEVICES="0,0,0", DWGSIM_HIP_MIN_SHARE="40", DWGSIM_HIP_BATCH="333", DWGSIM_HIP_THREADS="4", DWGSIM_HIP_GZIP=gz)
    subprocess.run([cli] + flags.split() + [os.path.join(golden_dir, fasta), str(tmp_path / "cli")], check=True, stderr=subprocess.DEVNULL, env=env)
    for k, suf in [(0, "bwa.read1.fastq.gz"), (1, "bwa.read2.fastq.gz"), (2, "bfast.fastq.gz")]:
        assert gzip.open(str(tmp_path / ("cli." + suf)), "rb").read() == want[k], suf
    assert open(str(tmp_path / "cli.mutations.txt"), "rb").read() == want["txt"]
    assert open(str(tmp_path / "cli.mutations.vcf"), "rb").read() == want["vcf"]


@pytest.mark.parametrize("flags,sizes", [
    ("-z 9 -N 900 -1 70 -2 50 -r 0.01 -y 0.1", (40000, 1, 900, 333333)),
    (f"-z 3 -c 2 -f {FLOW_ORDER} -1 200 -2 0 -e 0.02 -o 1", (20000, 7)),
    ("-z 4 -c 1 -1 50 -2 50 -o 2", (50000, 129)),
])
def test_gzip_members_made_on_the_gpu(lib, golden_dir, flags, sizes):
    from parity_common import check_gpu_gzip
    check_gpu_gzip(lib, os.path.join(golden_dir, "tiny.fa"), flags, sizes)


@pytest.mark.parametrize("k", range(6))
def test_both_record_writers(lib, oracle_bin, tmp_path, k):
    from parity_common import WRITER_CASES, check_record_writers
    check_record_writers(lib, oracle_bin, str(tmp_path), *WRITER_CASES[k])


@pytest.mark.parametrize("dirty_map", ["chunk", "word"])
def test_walking_a_contig_again(lib, golden_dir, monkeypatch, dirty_map):
    """... with both mappings of k_dirty_chunks (a lane per chunk: small groups; a thread per bitmap word: groups beyond 128 Mb)"""
    from parity_common import check_walking_a_contig_again
    monkeypatch.setenv("DWGSIM_HIP_DIRTY_MAP", dirty_map)
    check_walking_a_contig_again(lib, os.path.join(golden_dir, "tiny.fa"), "-z 4 -N 600 -r 0.02 -R 0.5 -I 3 -X 0.6", n=600)


def test_both_mappings_of_the_dirty_chunk_kernel(lib, oracle_bin, golden_dir, monkeypatch):
    """parity of a mutation-rich job and of count_random (the summaries k_place reads) with the thread-per-word form that large groups take"""
    from parity_common import check_count_random_matches_simulate
    monkeypatch.setenv("DWGSIM_HIP_DIRTY_MAP", "word")
    compare_case(lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), "-z 8 -N 3000 -1 50 -2 35 -d 300 -r 0.05 -R 0.5 -X 0.6 -y 0.1")
    check_count_random_matches_simulate(lib, os.path.join(golden_dir, "odd.fa"), "-z 6 -C 3 -1 40 -2 40 -d 150 -s 10 -r 0.08 -R 0.8 -X 0.6 -n 1 -y 0.1")


def test_gzip_kernel_on_hard_inputs(lib):
    from parity_common import check_gzip_kernel_on_hard_inputs
    check_gzip_kernel_on_hard_inputs(lib, scale=8)


JOB_CASES = [
    ("tiny.fa", "-z 9 -N 9000 -P pfx -r 0.01 -R 0.3 -y 0.2", dict(devices=[0, 0, 0], batch_pairs=700, min_share=40)),
    ("tiny.fa", "-z 9 -N 9000 -y 0.2 -o 1", dict(devices=[0, 0], gzip_on_gpu=False, batch_pairs=1000, min_share=40)),
    ("tiny.fa", "-z 9 -C 30 -y 0.1", dict(devices=[0, 0, 0], batch_pairs=64, min_share=1, group_bp=5000)),       # contig by contig (groups of one)
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -N 7000 -m {IN}/muts_edge.txt", dict(devices=[0, 0], batch_pairs=900, min_share=1)),
    ("odd.fa", "-z 6 -N 7000 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 120 -2 0 -e 0.05 -n 10 -r 0.05 -R 0.5 -y 0.2", dict(devices=[0, 0, 0, 0], batch_pairs=777, min_share=1)),
    ("tiny.fa", "-z 8 -N 4000 -c 1 -1 50 -2 35 -d 300 -r 0.02 -R 0.5 -e 0.05 -E 0.03 -y 0.1", dict(devices=[0, 0, 0], batch_pairs=500, min_share=1)),
    ("ex1.fa", "-z 13 -N 10000 -1 100 -2 100", dict(devices=None)),                                                    # every device the process sees, defaults
    ("ex1.fa", "-z 13 -N 8000 -M 2", dict(devices=[0, 0])),
    ("ex1.fa", "-z 13 -N 8000 -M 1", dict(devices=[0], batch_pairs=3000)),
]


@pytest.mark.parametrize("fasta,flags,kw", JOB_CASES, ids=[f"{f}:{fl}" for f, fl, _ in JOB_CASES])
def test_job_level_of_the_abi(lib, oracle_bin, golden_dir, fasta, flags, kw):
    """dwgsim_hip_job_* through ctypes (what a binding at the reference's seam calls): several contexts on the one GPU of the
    box, batches dealt round-robin, counted random reads as rand_ii bases, gzip members made on the GPU, ordered delivery -- every byte as
    the oracle's single-process run."""
    from parity_common import compare_job_api
    compare_job_api(lib, oracle_bin, os.path.join(golden_dir, fasta), flags, **kw)


def test_job_level_in_the_shape_of_a_whole_node(lib, oracle_bin, many_fa):
    """Eight workers (here eight contexts on the one GPU of the box), 30+ groups, tiny batches, the three delivery threads: the shape of the
    8-GPU node the scaling run uses -- every worker takes part in every group, counts its batches' random reads, waits for the other seven."""
    from parity_common import compare_job_api
    res = compare_job_api(lib, oracle_bin, many_fa, "-z 12 -C 25 -1 60 -2 40 -d 220 -s 10 -r 0.02 -R 0.5 -n 10 -y 0.15", devices=[0] * 8, batch_pairs=97, min_share=1, group_bp=12000)
    assert res.n_pairs > 20000
    compare_job_api(lib, oracle_bin, many_fa, "-z 13 -N 30000 -1 50 -2 0 -o 0 -y 0.3", devices=[0] * 8, batch_pairs=512, min_share=64, group_bp=40000, gzip_on_gpu=False)


def test_job_level_two_hundred_small_contigs(lib, oracle_bin, many_fa):
    from parity_common import compare_job_api
    compare_job_api(lib, oracle_bin, many_fa, "-z 11 -C 20 -1 50 -2 50 -d 200 -s 15 -r 0.03 -R 0.6 -X 0.6 -n 8 -y 0.1", devices=[0, 0, 0], batch_pairs=4096, min_share=100, group_bp=100000)
    compare_job_api(lib, oracle_bin, many_fa, "-z 15 -C 15 -1 150 -2 150 -o 1", devices=[0])


def test_job_level_abort_rule_across_devices(lib, oracle_bin, golden_dir):
    from test_emu_parity import test_job_level_abort_rule_across_devices_on_cpu_emulation as body
    body(lib, oracle_bin, golden_dir)


def test_two_ranks_over_gloo_on_the_real_library(oracle_bin, golden_dir, tmp_path):
    """The N>1 flow of bench.py / dw_job.cpp with two PROCESSES that share the box's one GPU: each rank walks every contig itself, counts the
    random reads of its batches (k_place), one gloo all-gather per group, batches dealt round-robin; the rank-ordered interleaving of the
    batches equals the oracle's single-process output.  (tests/test_sharding_gloo.py runs the same flow on the CPU emulation.)"""
    import pickle, subprocess, sys
    from parity_common import run_oracle, STREAMS, first_diff
    from test_sharding_gloo import WORKER_BATCHES
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = "-z 9 -N 20000 -y 0.25 -1 50 -2 50 -d 200 -s 20 -r 0.01 -R 0.3"
    fasta = os.path.join(golden_dir, "tiny.fa")
    want = run_oracle(oracle_bin, fasta, flags, str(tmp_path))
    w = tmp_path / "worker.py"
    w.write_text(WORKER_BATCHES)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(w), root, flags, fasta, str(tmp_path), os.path.join(root, "dwgsim_amd", "libdwgsim_hip.so"), "1500"], env=dict(env, RANK=str(r))) for r in range(2)]
    for p_ in procs:
        assert p_.wait(timeout=600) == 0
    parts = [pickle.load(open(str(tmp_path / f"rank{r}.pkl"), "rb")) for r in range(2)]
    merged = sorted(parts[0] + parts[1], key=lambda m: (m[0], m[1]))          # (group, batch, {stream: bytes})
    for s_ in STREAMS:
        got = b"".join(m[2][s_] for m in merged)
        assert got == want[s_], f"{STREAMS[s_]}: " + first_diff(got, want[s_])


def test_cli_abort_rule_across_contexts(oracle_bin, golden_dir, tmp_path):
    """The failure counter of dwgsim.c:635 runs over the pairs of a contig in index order; with the contig split over contexts no single
    range reaches 10 000 failures in this job, the joined summaries do: dwgsim-hip must die as the reference does -- and must not when
    the job is shortened below the limit."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "dwgsim_amd", "dwgsim-hip")
    env = dict(os.environ, DWGSIM_HIP_DEVICES="0,0", DWGSIM_HIP_MIN_SHARE="10", DWGSIM_HIP_BATCH="150", DWGSIM_HIP_THREADS="2")
    fa = os.path.join(golden_dir, "odd.fa")
    base = "-z 6466 -1 33 -2 150 -d 900 -s 50 -r 0 -e 0.0-0.1 -Q 0 -a"
    r = subprocess.run([cli] + (base + " -N 1200").split() + [fa, str(tmp_path / "a")], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "failed to generate a read after 10001 trials" in r.stderr, r.stderr[-300:]
    r = subprocess.run([cli] + (base + " -N 600").split() + [fa, str(tmp_path / "b")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-300:]


def test_async_two_slot_pipeline_equals_blocking_calls(lib, golden_dir):
    """simulate_async / wait / fetch_async / fetch_wait with two batches in flight and the device-chained rand_ii (DWGSIM_HIP_RAND_CHAIN)
    against the blocking simulate / fetch calls on the same ranges."""
    import ctypes as C
    params = api.parse_flags("-z 41 -C 60 -1 100 -2 100 -y 0.2 -r 0.01 -R 0.3", lib)
    name, arr = api.read_fasta(os.path.join(golden_dir, "tiny.fa"))[0]
    n_pairs, batch = 1800, 250
    with api.Context(params, 0, lib) as ctx:
        cid = ctx.add_contig(name, arr, 0)
        ctx.mutate(cid)
        want = [b"", b"", b""]; rr = 7
        for first in range(0, n_pairs, batch):
            b = ctx.simulate(cid, first, min(batch, n_pairs - first), rr, 0)
            for s in range(3):
                want[s] += ctx.fetch(0, s, b.bytes[s])
            rr += b.n_random
        got = [b"", b"", b""]
        cap = 1 << 20
        bufs = [[lib.dwgsim_hip_host_alloc(cap) for _ in range(3)] for _ in range(2)]
        pending = []

        def finish(slot):
            b = ctx.wait(slot)
            for s in range(3):
                ctx.fetch_async(slot, s, bufs[slot][s], cap)
            ctx.fetch_wait(slot)
            for s in range(3):
                got[s] += C.string_at(bufs[slot][s], b.bytes[s])
        for k, first in enumerate(range(0, n_pairs, batch)):
            ctx.simulate_async(cid, first, min(batch, n_pairs - first), 7 if k == 0 else api.RAND_CHAIN, k & 1)
            if k > 0:
                finish((k - 1) & 1)
        finish(k & 1)
        for slot in range(2):
            for s in range(3):
                lib.dwgsim_hip_host_free(bufs[slot][s])
    assert got == want and len(want[2]) > 100000


def test_range_restricted_fp64_forms_equal_the_general_ones(lib):
    """The quality path uses a division / sqrt / log specialised to its operand range (dw_common.hpp: div_mid, sqrt_mid,
    det_log<true>): on 2^28 operand sets drawn as that path draws them (plus a 2^-70..2^70 exponent sweep) they must give the
    same bits as the compiler's `/`, sqrt() and the general det_log."""
    import ctypes as C
    out = (C.c_uint64 * 4)()
    lib.dwgsim_hip_selftest_fp64.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.dwgsim_hip_selftest_fp64.restype = C.c_int
    for seed in (1, 2):
        assert lib.dwgsim_hip_selftest_fp64(0, seed, 1 << 28, out) == 0
        assert out[3] > (1 << 29) and (out[0], out[1], out[2]) == (0, 0, 0), list(out)


def test_number_text_of_the_name_line_equals_one_division_per_digit(lib):
    """put_dec / put_hex (dw_read.hpp: four digits per multiplication, hexadecimal digits by nibble spreading) against the plain per-digit loop: EVERY
    32-bit value in decimal (positions, counts), and in hexadecimal (the read index) every value below 2^32 plus 2^31 values spread over 64 bits."""
    import ctypes as C
    out = (C.c_uint64 * 4)()
    lib.dwgsim_hip_selftest_text.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.dwgsim_hip_selftest_text.restype = C.c_int
    for first, n, stride in ((0, 1 << 32, 1), (0, 1 << 31, 0x1_0000_0003), (0xFFFF_FFFF_0000_0000, 1 << 24, 257), ((1 << 60) - 5000, 10000, 1)):
        assert lib.dwgsim_hip_selftest_text(0, first, n, stride, out) == 0
        assert out[2] == n and (out[0], out[1]) == (0, 0), (first, n, stride, list(out))


def test_lazy_quality_normals_decide_exactly_what_the_exact_form_decides(lib):
    """quality_try_lazy (fp32 estimate + proven error bound, exact fp64 path only near a rounding boundary) against quality_try_exact on EVERY
    possible try -- a try of the quality stream is one 32-bit word, so 2^32 of them per quality_std is all there is -- and the hardware
    operations the bound rests on (v_log_f32, v_rcp_f32, v_sqrt_f32) against fp64 on EVERY float of their operand ranges."""
    import ctypes as C, struct
    out = (C.c_uint64 * 12)()
    lib.dwgsim_hip_selftest_lazy.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_double, C.c_int, C.POINTER(C.c_uint64)]
    lib.dwgsim_hip_selftest_lazy.restype = C.c_int
    dbl = lambda u: struct.unpack("<d", struct.pack("<Q", u))[0]
    for k, sigma in enumerate((2.0, 0.3, 1.0, 3.7, 10.0, 40.0, 5000.0, 1e7)):
        assert lib.dwgsim_hip_selftest_lazy(0, 0, 1 << 32, sigma, 1 if k == 0 else 0, out) == 0
        assert out[4] == 1 << 32, list(out)
        assert out[0] == 0 and out[1] == 0, (sigma, list(out))                  # not one offset, not one accept / reject verdict differs
        assert 3.3e9 < out[2] + out[3] < 3.4e9                                   # pi / 4 of the tries are accepted
        assert dbl(out[5]) < 0.5, (sigma, dbl(out[5]))                           # the estimate stays well inside its proven bound
        if sigma <= 10:
            assert out[3] < 0.01 * out[4], (sigma, out[3] / out[4])              # ... and the exact path is rare at realistic -Q
        if k == 0:
            assert 0 < dbl(out[6]) <= 1.0 and 0 < dbl(out[7]) <= 1.0 and 0 < dbl(out[8]) <= 1.0, [dbl(out[q]) for q in (6, 7, 8)]
        print(f"sigma {sigma}: exact-path share {out[3] / max(out[2] + out[3], 1):.5f}, max |y - x| / eps {dbl(out[5]):.3f}", [round(dbl(out[q]), 3) for q in (6, 7, 8)] if k == 0 else "")


@pytest.mark.parametrize("which,flags", [
    ("tiny", "-z 9 -C 40 -y 0.15 -n 0"),
    ("odd", "-z 6 -C 40 -1 40 -2 40 -d 150 -s 10 -r 0.08 -R 0.8 -X 0.6 -n 1 -y 0.1"),
    ("odd", "-z 6 -C 30 -2 0 -1 120 -r 0.05 -R 0.9 -I 40 -y 0.3 -n 3"),
    ("repeats", "-z 34 -C 2 -1 100 -2 100 -r 0.03 -R 0.5 -X 0.4 -n 5 -y 0.05"),
    ("tiny", "-z 5 -x {IN}/regions_a.bed -C 30 -y 0.2"),
])
def test_count_random_matches_simulate(lib, golden_dir, repeats_fa, which, flags):
    """The sharding primitive: k_place (summary fast path + exact walk) against k_simulate's own count, on N-rich contigs,
    dense long indels, contig ends and target regions."""
    from parity_common import check_count_random_matches_simulate
    fasta = repeats_fa if which == "repeats" else os.path.join(golden_dir, which + ".fa")
    check_count_random_matches_simulate(lib, fasta, flags)


ILLUMINA_CASES = [c for c in CASES if "-c 1" not in c[1] and "-c 2" not in c[1]]


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("fasta,flags", ILLUMINA_CASES, ids=[f"{f}:{fl}" for f, fl in ILLUMINA_CASES])
def test_both_forms_of_the_illumina_read_kernel(lib, oracle_bin, golden_dir, fasta, flags, split):
    """k_simulate as ONE kernel (look-backs over the blocks in front) and as TWO (first half | offsets | second half, dw_simulate.hip SPLIT): the
    library picks one by read length; here every Illumina option set runs through both, batches of 777 pairs, byte for byte against the oracle."""
    compare_case(lib, oracle_bin, os.path.join(golden_dir, fasta), flags, batch_pairs=777, debug_options={"split": split})


@pytest.mark.parametrize("flags", ["-z 17 -1 50 -2 50 -d 300 -s 20 -C 10 -y 0.1 -r 0.01 -R 0.5 -n 1", "-z 18 -1 150 -2 150 -C 8 -n 0", "-z 19 -1 100 -2 0 -C 4 -y 0.02 -r 0.003",
                                   "-z 20 -1 100 -2 100 -i -d 100 -s 30 -C 6 -S 1", "-z 21 -1 120 -2 80 -d 5000 -s 700 -C 6 -S 2 -A 2"])
def test_count_random_fast_and_long_path(lib, flags):
    """k_place's two-Philox-block decision (coarse summaries, any insert size within dist +- 12.1 sigma) + k_place_rest against k_simulate's own
    count: 3 Mb contigs with N runs at the ends and inside, paired / single-end / inner-distance / mate-pair geometries, forced list overflow."""
    from parity_common import check_count_random_fast_path
    check_count_random_fast_path(lib, length=3000000, n=150000, flags=flags, ranges=((0, None), (33333, 55555)))


@pytest.mark.parametrize("slots", [1, 3, 0])
def test_ion_torrent_scratch_slots_change_hands(lib, oracle_bin, repeats_fa, slots):
    """The Ion Torrent read buffers are scratch slots handed from block to block inside an XCD (dw_simulate.hip scratch_slot_take).  One or three
    slots per XCD make almost every one of the ~ 350 blocks WAIT for a slot that another block of its XCD releases (the path that a normal launch,
    with a slot per resident block, takes only when a release is still on its way); 0 = the library's own count."""
    compare_case(lib, oracle_bin, repeats_fa, "-z 41 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 200 -2 0 -C 10 -e 0.02 -y 0.03 -r 0.01",
                 debug_options={"ion_lds": 0, "flow_slots": slots} if slots else {"ion_lds": 0})      # (ion_lds = 0: the buffers in scratch slots -- since round 5 the fallback for reads LDS cannot hold)


ION_HOMES = [{"ion_lds": 1}, {"ion_lds": 1, "split": 0}, {"ion_lds": 2}, {"ion_lds": 0}]      # LDS as two kernels (the default) / as one kernel / the smaller blocks / scratch slots
ION_HOME_CASES = [
    ("tiny.fa", f"-z 9 -N 3000 -c 2 -f {FLOW} -1 400 -2 0 -e 0.01 -y 0.1"),
    ("tiny.fa", f"-z 9 -N 2000 -c 2 -f {FLOW} -1 200 -2 100 -e 0.02 -E 0.03 -d 600 -o 0"),
    ("odd.fa", f"-z 6 -N 3000 -c 2 -f {FLOW} -1 120 -2 0 -e 0.05 -n 10 -r 0.05 -R 0.5 -y 0.2 -A 2"),
    ("tiny.fa", "-z 9 -N 1200 -c 2 -f " + "TACG" * 4 + "A" * 34 + " -1 90 -2 60 -e 0.02 -E 0.01 -d 300 -o 1"),
    ("ex1.fa", f"-z 6472 -N 1500 -c 2 -f {FLOW} -1 150 -2 0 -e 0.3 -A 1"),
]


@pytest.mark.parametrize("home", ION_HOMES, ids=[",".join(f"{k}={v}" for k, v in h.items()) for h in ION_HOMES])
@pytest.mark.parametrize("fasta,flags", ION_HOME_CASES, ids=[f"{f}:{fl}" for f, fl in ION_HOME_CASES])
def test_ion_torrent_in_every_home_of_its_read_buffers(lib, oracle_bin, golden_dir, fasta, flags, home):
    """The flow model's one in-place 2-bit buffer per lane (dw_read.hpp flow_errors) in LDS -- as two kernels (flow model | qualities + text: the default) or
    as one -- in LDS with the smaller blocks, and in scratch slots of global memory (what reads too long for LDS fall back to): the same bytes everywhere."""
    compare_case(lib, oracle_bin, os.path.join(golden_dir, fasta), flags, batch_pairs=700, debug_options=home)


@pytest.mark.parametrize("flags", __import__("parity_common").LONG_READ_CASES)
def test_reads_beyond_the_lds_staging_limit(lib, oracle_bin, repeats_fa, flags):
    """-1 10000, -c 1 -1 6000, 7 000 + 5 000 paired, 25 000: see parity_common.LONG_READ_CASES; also with two scratch slots per XCD (blocks wait for slots)."""
    compare_case(lib, oracle_bin, repeats_fa, flags)
    compare_case(lib, oracle_bin, repeats_fa, flags, batch_pairs=77, debug_options={"flow_slots": 2})


@pytest.mark.parametrize("cap,flags", [(104, "-1 100 -2 0 -e 0.05 -y 0.1"), (60, "-1 50 -2 50 -d 300 -e 0.1 -E 0.02 -o 0"), (20, "-1 17 -2 0 -e 0.1 -f TCG" + "A" * 12),
                                       (20, "-1 17 -2 0 -e 0.19 -f TCG" + "A" * 12)])      # (the last: 17-base reads that grow to 2 059 bases -- 128 x the starting capacity; rounds 3-4 gave up at 16 x)
def test_ion_torrent_read_outgrows_its_buffers(lib, oracle_bin, golden_dir, cap, flags):
    """A read that outgrows its flow-space buffers makes the batch run again with twice the room (dw_host.cpp dwgsim_hip_wait; the reference doubles its
    buffers, dwgsim.c:296-311): forced with a small starting capacity, batch by batch and through the job level with two batches in flight per
    context; the third case: a flow order that keeps T away for twelve flows at e = 0.1 (17-base reads that grow up to 112 bases: three doublings from 20)."""
    # (-e 0.19 with twelve empty flows in front of every T is SUPERCRITICAL -- insertions breed faster than they are examined -- and how far a read gets is a
    # matter of the seed: -z 12 stays inside the pass-2 run stack of 64 pending runs, -z 9 .. 11 do not with round 6's gap-drawn stream)
    fl = f"-z {12 if '0.19' in flags else 9} -N {400 if '0.19' in flags else 2500} -c 2 {'' if ' -f ' in flags else '-f ' + FLOW} {flags}"
    for home in ({}, {"ion_lds": 1, "split": 0}, {"ion_lds": 0}):
        res = compare_case(lib, oracle_bin, os.path.join(golden_dir, "tiny.fa"), fl, batch_pairs=700, debug_options=dict(home, flow_cap=cap))
        assert res.flow_cap_mult >= (128 if '0.19' in flags else 2)


def test_walk_reruns_when_a_capacity_is_exceeded(lib, oracle_bin, repeats_fa):
    """See tests/test_emu_parity.py: forced tiny capacities on the repeat-rich 1.8 Mb contigs."""
    compare_case(lib, oracle_bin, repeats_fa, "-z 32 -M 2 -r 0.2 -R 0.9 -X 0.3 -I 2", debug_options={"walk_cap": 100})


def test_abort_rule_matches_the_reference(lib, oracle_bin, golden_dir):
    """See tests/test_emu_parity.py; here also with small batches, so the counter is carried between simulate() calls."""
    from parity_common import check_both_abort
    flags = "-z 6466 -1 33 -2 150 -d 900 -s 50 -N 1200 -r 0 -e 0.0-0.1 -Q 0 -a"
    check_both_abort(lib, oracle_bin, os.path.join(golden_dir, "odd.fa"), flags)
    with pytest.raises(api.DwgsimError, match="failed to generate a read after 10001 trials"):
        api.run_job(api.parse_flags(flags, lib), api.read_fasta(os.path.join(golden_dir, "odd.fa")), batch_pairs=37, lib=lib)
    compare_case(lib, oracle_bin, os.path.join(golden_dir, "odd.fa"), flags.replace("-N 1200", "-N 600"), batch_pairs=37)


def test_hopeless_target_regions_end_with_an_error(lib, golden_dir, tmp_path):
    """Regions that pass the length checks but can never hold a fragment: the reference spins forever (dwgsim.c:677-713); here the
    placement gives up after 2^20 tries, the rest of the batch stops early, and the call returns an error."""
    bed = tmp_path / "r.bed"
    bed.write_text("t1\t100\t500\nt1\t900\t1300\n")
    with pytest.raises(api.DwgsimError, match="no fragment placement satisfied the target regions"):
        api.run_job(api.parse_flags(f"-z 3 -N 130 -1 50 -2 50 -d 500 -s 5 -x {bed}", lib), api.read_fasta(os.path.join(golden_dir, "tiny.fa")), lib=lib)


def test_mut_debug_aborts_like_the_reference(lib, oracle_bin, golden_dir):
    from parity_common import check_mut_debug_aborts
    check_mut_debug_aborts(lib, oracle_bin, golden_dir)


from replay_common import REPLAY_CASES, have_reference


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref/dwgsim (the unmodified reference; it travels to the GPU box as a prebuilt binary) is not here")
@pytest.mark.parametrize("fasta,flags", REPLAY_CASES, ids=[f"{f}:{fl}" for f, fl in REPLAY_CASES])
def test_hip_path_equals_the_replayed_reference(lib, oracle_bin, golden_dir, tmp_path, fasta, flags):
    """The HIP path against bytes the UNMODIFIED REFERENCE wrote: oracle/_ref/dwgsim with its drand48() replaying the Philox stream (tests/replay_common.py;
    single-end configurations: Illumina, SOLiD, the whole Ion Torrent flow model, heavy -r / -R, -x, -m / -b / -v, and -- reads of even length -- quality
    noise -Q 0.5 ... 60, i.e. the 16-bit polar tries and lazy normals of the headline kernel against the reference's own ran_normal with glibc log) -- no oracle
    output is compared here, the oracle only dumps the stream.  The last case also goes through the dwgsim-hip executable."""
    import gzip, subprocess
    from replay_common import run_replay, SUFFIXES, IN_DIR
    fa = os.path.join(golden_dir, fasta)
    orc, rrc, _, ref, served, avail = run_replay(oracle_bin, fa, flags, str(tmp_path))
    assert orc == 0 and rrc == 0 and served == avail > 0
    fl = flags.replace("{IN}", IN_DIR)
    res = api.run_job(api.parse_flags(fl, lib), api.read_fasta(fa), lib=lib)
    assert res.streams[0] == ref["bwa.read1.fastq"] and res.streams[1] == ref["bwa.read2.fastq"] and res.streams[2] == ref["bfast.fastq"]
    assert res.mutations_txt == ref["mutations.txt"] and res.mutations_vcf == ref["mutations.vcf"]
    if (fasta, flags) == REPLAY_CASES[-1]:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run([os.path.join(root, "dwgsim_amd", "dwgsim-hip")] + fl.split() + [fa, str(tmp_path / "cli")], check=True, stderr=subprocess.DEVNULL)
        for suf in SUFFIXES:
            p = str(tmp_path / ("cli." + suf + (".gz" if suf.endswith("fastq") else "")))
            got = (gzip.open(p, "rb").read() if p.endswith(".gz") else open(p, "rb").read()) if os.path.exists(p) else b""
            assert got == ref[suf], suf
