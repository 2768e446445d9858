"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
declared in include/dwgsim_hip.h, option defaults/checks and the contig-scheduling arithmetic mirror
the reference (dwgsim_opt.c:40-80, :307-371; dwgsim.c:535-618), and the product refuses to run
without a GPU (no CPU fallback).  No compute calls are made here."""
import ctypes, os, re, subprocess
import pytest

from dwgsim_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(api.LIB_PATH):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "dwgsim_amd", "csrc"), "../libdwgsim_hip.so"], check=True)
    return api.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dwgsim_hip.h")).read()
    declared = set(re.findall(r"\b(dwgsim_hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.EXPORTS)
    for s in declared:
        assert hasattr(lib, s), s


def test_defaults_match_dwgsim_opt_init(lib):
    p = api.default_params(lib)
    assert (p.dist, p.std_dev, p.N, p.C, p.length[0], p.length[1]) == (500, 50.0, -1, 100.0, 70, 70)
    assert (p.mut_rate, p.mut_freq, p.indel_frac, p.indel_extend, p.indel_min) == (0.001, 0.5, 0.1, 0.3, 1)
    assert (p.rand_read, p.max_n, p.data_type, p.seed, p.quality_std, p.fixed_quality) == (0.05, 0, 0, -1, 2.0, -1)
    assert list(p.e_start) == [0.02, 0.02] and list(p.e_end) == [0.02, 0.02]


@pytest.mark.parametrize("flags,code", [
    ("-z 1 -N 10", 0), ("-z 1 -N 10 -C 5", 0), ("-z 1 -1 0 -N 5", -1), ("-z 1 -N 5 -r 1.5", -1), ("-z 1 -N 5 -y -0.1", -1),
    ("-z 1 -N 5 -S 3", -1), ("-z 1 -N 5 -c 1", 0), ("-z 1 -N 5 -c 2 -f TACG -2 0", 0), ("-z 1 -N 5 -c 2 -f TAC -2 0", -4), ("-z 1 -N 5 -c 2 -f TACG -e 0.01-0.02", -1), ("-z 1 -N 5 -e 1.2", -1), ("-N 5", -1), ("-z 1 -N 5 -o 3", -1),
])
def test_option_checks(lib, flags, code):
    p = api.parse_flags(flags, lib)
    msg = ctypes.create_string_buffer(512)
    assert lib.dwgsim_hip_params_check(ctypes.byref(p), msg, 512) == code
    if code:
        assert msg.value


def test_pairs_per_contig_matches_reference_arithmetic(lib):
    # ex1.fa -N 10000 -> 4986 + 5014; E. coli 30x 2x150 default -y 0.05 -> 488595; -y 0 -> 464165
    p = api.parse_flags("-z 13 -N 10000", lib)
    assert api.pairs_for_contig(p, 1575, 3159, False, 0, lib) == 4986
    assert api.pairs_for_contig(p, 1584, 3159, True, 4986, lib) == 5014
    p = api.parse_flags("-z 13 -1 150 -2 150 -C 30", lib)
    assert api.pairs_for_contig(p, 4641652, 4641652, True, 0, lib) == 488595
    p = api.parse_flags("-z 13 -1 150 -2 150 -C 30 -y 0", lib)
    assert api.pairs_for_contig(p, 4641652, 4641652, True, 0, lib) == 464165
    assert api.pairs_for_contig(p, 64444167, 64444167, True, 0, lib) == 6444417
    # skip rules #3 (shorter than d + 3 sigma), #4 (shorter than a read), #2 (amplicon shorter than read)
    assert api.pairs_for_contig(p, 600, 10000, False, 0, lib) == api.SKIP_SHORT_INSERT
    q = api.parse_flags("-z 1 -N 100 -2 0 -1 70", lib)
    assert api.pairs_for_contig(q, 50, 1000, False, 0, lib) == api.SKIP_SHORT_READ
    q = api.parse_flags("-z 1 -N 100 -a", lib)
    assert api.pairs_for_contig(q, 50, 1000, False, 0, lib) == api.SKIP_AMPLICON


def test_no_cpu_fallback(lib):
    """Without a HIP device the product refuses to create a context (this container has no GPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.DwgsimError):
        api.Context(api.parse_flags("-z 1 -N 10", lib), 0, lib)


def test_product_does_not_link_or_import_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "dwgsim_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(root, f), errors="replace").read()
                assert "oracle/" not in src and "liboracle" not in src and "dwgsim_oracle" not in src, f
    out = subprocess.run(["ldd", api.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_fasta_reader_follows_seq_read_fasta(tmp_path):
    p = tmp_path / "x.fa"
    p.write_bytes(b"garbage\n>c1 comment > here\r\nACGT\r\nac-g.t1 2\n>c2\tdesc\nNNNN>c3\nAC\n")
    got = api.read_fasta(str(p))
    assert [(n, bytes(a)) for n, a in got] == [("c1", b"ACGTac-g.t"), ("c2", b"NNNN"), ("c3", b"AC")]
