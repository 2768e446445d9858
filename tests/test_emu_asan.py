"""CPU-only memory-safety check of the kernels' logic: the kernel sources compiled against the SIMT emulation shim with
AddressSanitizer, driven through the C-ABI on a few option sets (including flow-model failures, where reads emit nothing).
An out-of-bounds access on the GPU is silent corruption; here it is a hard failure.  Test infrastructure only."""
import os, subprocess, sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "dwgsim_amd", "csrc")

DRIVER = r'''
import os, re, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dwgsim_amd import api
from parity_common import compare_case, compare_job_api, check_gpu_gzip, check_record_writers, WRITER_CASES
lib = api.load(LIB)
oracle = os.path.join(ROOT, "oracle", "build", "dwgsim_oracle")
g = os.path.join(ROOT, "tests", "golden")
for fasta, flags in [("ex1.fa", "-z 13 -N 160"), ("odd.fa", "-z 3 -N 200 -1 50 -2 50 -d 200 -s 20 -r 0.1 -R 1.0 -X 0.7 -n 50 -y 0.1"),
                     ("tiny.fa", "-z 8 -N 200 -c 1 -1 50 -2 35 -d 300 -r 0.02 -R 0.5 -e 0.05"),
                     ("tiny.fa", "-z 9 -N 200 -c 2 -f TACGTACGTCTGAGCATCGATCGATGTACAGC -1 200 -2 100 -e 0.02 -E 0.03 -d 600"),
                     ("tiny.fa", "-z 3 -N 40 -1 1300 -2 1400 -d 3600 -s 40 -n 60")]:
    compare_case(lib, oracle, os.path.join(g, fasta), flags, batch_pairs=128)
# a contig whose length is 16 mod 64 with a mutation in its LAST base: the dirty 64-cell chunk of the sparse walk ends 48 cells past the contig
# (round 5's advisor: with 32 cells of padding k_dirty_chunks read and wrote 16 bytes past the allocations; the emulator's hipMalloc is exact now)
import numpy as np, tempfile
from dwgsim_amd import synth
with tempfile.TemporaryDirectory() as td:
    for L in (1040, 4112 + 64):
        fa = os.path.join(td, f"l{L}.fa"); mf = os.path.join(td, f"m{L}.txt")
        arr = synth.random_contig(L, 5)
        synth.write_fasta(fa, [("c16", arr)])
        alt = "A" if chr(arr[L - 1]) != "A" else "C"
        open(mf, "w").write(f"c16\t{L}\t{chr(arr[L - 1])}\t{alt}\t3\n")
        compare_case(lib, oracle, fa, f"-z 5 -N 100 -1 50 -2 50 -d 200 -s 10 -m {mf}", batch_pairs=128)
        compare_case(lib, oracle, fa, "-z 7 -N 100 -1 50 -2 50 -d 200 -s 10 -r 0.2 -R 0.3", batch_pairs=128)
# contigs resident together (one coordinate space, launches across contig boundaries), and the job level on three contexts
compare_case(lib, oracle, os.path.join(g, "odd.fa"), "-z 3 -N 300 -1 50 -2 50 -d 200 -s 20 -r 0.1 -R 1.0 -X 0.7 -n 50 -y 0.1", batch_pairs=90, group_bp=1 << 30)
compare_job_api(lib, oracle, os.path.join(g, "tiny.fa"), "-z 9 -N 300 -y 0.2 -r 0.02 -R 0.5", devices=[0, 0, 0], gzip_on_gpu=False, batch_pairs=64, min_share=20)
# reads the flow model gives up on (absurd per-flow error): the call must fail cleanly, with every write in bounds
compare_case(lib, oracle, os.path.join(g, "ex1.fa"), "-z 8397 -1 100 -2 0 -N 64 -e 0.3 -o 0 -c 2 -f GATC", batch_pairs=128)      # deep insertion cascades
# reads that grow to 128 x their starting capacity: the buffers are doubled again and again, from LDS into scratch slots (rounds 3-4 stopped at 16 x)
compare_case(lib, oracle, os.path.join(g, "tiny.fa"), "-z 12 -N 400 -c 2 -f TCG" + "A" * 12 + " -1 17 -2 0 -e 0.19", batch_pairs=300, debug_options={"flow_cap": 20})
for flags in ["-z 4246 -1 9 -2 0 -N 300 -e 1.0 -y 0.3 -n 20 -c 2 -f TACG", "-z 4246 -1 9 -2 9 -d 40 -N 300 -e 1.0 -o 0 -c 2 -f TACG"]:
    try:
        api.run_job(api.parse_flags(flags, lib), api.read_fasta(os.path.join(g, "ex1.fa")), lib=lib)
        raise SystemExit("expected an error for " + flags)
    except api.DwgsimError as e:
        assert "flow-error model" in str(e), str(e)
# the gzip kernel (members of a long and of a one-record stream) and both record writers
check_gpu_gzip(lib, os.path.join(g, "tiny.fa"), "-z 9 -N 900 -1 70 -2 50 -r 0.01 -y 0.1", sizes=(60, 1))
import tempfile
with tempfile.TemporaryDirectory() as td:
    check_record_writers(lib, oracle, td, WRITER_CASES[1][0].replace("-N 1500", "-N 150"), 0)
    check_record_writers(lib, oracle, td, WRITER_CASES[4][0].replace("-N 1000", "-N 100"), 200)
print("ASAN-CLEAN")
'''


def test_kernel_sources_are_asan_clean_on_the_emulator(oracle_bin, tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not available")
    lib = str(tmp_path / "libdwgsim_emu_asan.so")
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
                    "-I" + os.path.join(HERE, "emu"), "-I" + SRC, "-x", "c++", os.path.join(SRC, "dw_walk.hip"), os.path.join(SRC, "dw_gzip.hip"), os.path.join(SRC, "dw_simulate.hip"),
                    os.path.join(SRC, "dw_host.cpp"), os.path.join(SRC, "dw_mutin.cpp"), os.path.join(SRC, "dw_job.cpp"), os.path.join(HERE, "emu", "hip_emu.cpp"), "-o", lib], check=True)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\nLIB = {lib!r}\n" + DRIVER], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0 and "ASAN-CLEAN" in r.stdout, (r.stdout[-800:], r.stderr[-3000:])
