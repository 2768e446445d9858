"""Replay parity (test infrastructure): the UNMODIFIED reference binary fed the Philox stream.

`dwgsim_oracle --rng philox --dump-draws F` writes every uniform of mode B in the order it is consumed; oracle/replay48.c, preloaded into
oracle/_ref/dwgsim, serves them as drand48().  On configurations that draw no normal (single-end; -Q 0 or -q, or an even read length: the reference's Box-Muller cache
leaks a variate from pair to pair, dwgsim.c:158-159, which a counter-based stream deliberately does not reproduce) the reference must then write
exactly the files mode B writes -- and so must the HIP path -- and consume exactly the draws that were dumped.

Round 6: quality NOISE is replayed too where it can be.  ran_normal (dwgsim.c:156-175) draws two uniforms per polar try and delivers two normals per
accepted try; a single-end read of EVEN length with -Q > 0 takes L normals = L / 2 accepted tries, so the static cache is empty at every read boundary
and nothing leaks from read to read: the unmodified reference (glibc log) then consumes the 16-bit polar tries of the D_QUAL0 stream exactly
as mode B does.  This puts the headline quality path (16-bit operands, det_log, lazy fp32 estimates on the GPU) under the reference itself.  Odd lengths
and paired ends (the insert-size normal shares the cache, dwgsim.c:657) stay on the two-link chain."""
import gzip, os, random, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "dwgsim")
SHIM = os.path.join(ROOT, "oracle", "build", "libreplay48.so")
IN_DIR = os.path.join(ROOT, "tests", "golden", "inputs")
SUFFIXES = ("bwa.read1.fastq", "bwa.read2.fastq", "bfast.fastq", "mutations.txt", "mutations.vcf")
FLOW = "TACGTACGTCTGAGCATCGATCGATGTACAGC"

# (fasta, flags): single-end, no quality normals; Illumina, SOLiD, Ion Torrent incl. heavy mutation rates, regions, mutation inputs
REPLAY_CASES = [
    ("ex1.fa", "-z 13 -N 3000 -1 70 -2 0 -Q 0"),
    ("tiny.fa", "-z 9 -N 3000 -1 100 -2 0 -q I -r 0.01 -R 0.3 -X 0.5"),
    ("odd.fa", "-z 3 -N 3000 -1 50 -2 0 -Q 0 -r 0.1 -R 1.0 -X 0.7 -n 50"),
    ("tiny.fa", "-z 4 -N 3000 -1 80 -2 0 -Q 0 -r 0.02 -R 0.5 -I 30 -X 0.6 -H"),
    ("tiny.fa", "-z 21 -N 3000 -1 60 -2 0 -q 5 -F 0.2 -y 0.3 -A 1 -e 0.001-0.05 -P pfx -o 1"),
    ("tiny.fa", "-z 8 -N 3000 -c 1 -1 50 -2 0 -Q 0 -r 0.02 -R 0.5 -e 0.05 -y 0.1 -n 3"),
    ("odd.fa", "-z 6 -N 2500 -c 1 -1 40 -2 0 -q 5 -r 0.08 -R 0.8 -n 20 -o 2"),
    ("tiny.fa", f"-z 9 -N 1500 -c 2 -f {FLOW} -1 400 -2 0 -e 0.01 -Q 0"),
    ("odd.fa", f"-z 6 -N 2000 -c 2 -f {FLOW} -1 120 -2 0 -e 0.05 -q 9 -n 10 -r 0.05 -R 0.5 -y 0.2"),
    ("ex1.fa", f"-z 6472 -N 1200 -c 2 -f {FLOW} -1 150 -2 0 -e 0.3 -A 1 -Q 0"),
    ("tiny.fa", "-z 9 -N 1000 -c 2 -f TACG -1 100 -2 0 -e 0.2 -o 1 -Q 0 -r 0.05 -R 0.9"),
    # even-length single-end reads WITH quality noise (module docstring): the normals of the quality line under the unmodified reference
    ("tiny.fa", "-z 9 -N 3000 -1 100 -2 0 -Q 2"),
    ("ex1.fa", "-z 13 -N 3000 -1 150 -2 0 -o 1"),                      # -Q 2 is the default (dwgsim_opt.c:80): the headline read length
    ("odd.fa", "-z 3 -N 3000 -1 50 -2 0 -Q 10 -r 0.05 -R 0.5 -e 0.001-0.05"),
    ("tiny.fa", "-z 9 -N 2000 -c 1 -1 50 -2 0 -Q 3 -y 0.1"),
    ("tiny.fa", "-z 77 -N 2000 -1 36 -2 0 -Q 60 -e 0.3"),              # sigma 60: offsets clamp at both ends of the quality range
    ("tiny.fa", "-z 5 -x {IN}/regions_a.bed -N 2000 -1 70 -2 0 -Q 0"),
    ("tiny.fa", "-z 5 -N 2000 -1 70 -2 0 -q I -m {IN}/muts_edge.txt"),
    ("tiny.fa", "-z 5 -N 2000 -1 70 -2 0 -Q 0 -b {IN}/muts_edge.bed"),
    ("tiny.fa", "-z 5 -N 2000 -1 70 -2 0 -Q 0 -v {IN}/muts_generated.vcf -C 3"),
]


def have_reference():
    return os.path.exists(REF_BIN)


def run_replay(oracle_bin, fasta, flags, workdir, timeout=120):
    """-> (oracle rc, reference rc, {suffix: bytes} of mode B, {suffix: bytes} of the replayed reference, draws served, draws dumped)"""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    flags = flags.replace("{IN}", IN_DIR)
    dump = os.path.join(workdir, "draws.bin"); rep = os.path.join(workdir, "report.txt")
    o = subprocess.run([oracle_bin, "--rng", "philox", "--dump-draws", dump] + flags.split() + [fasta, os.path.join(workdir, "o")], capture_output=True, timeout=timeout)
    env = dict(os.environ, LD_PRELOAD=SHIM, REPLAY48_FILE=dump, REPLAY48_REPORT=rep)
    r = subprocess.run([REF_BIN] + flags.split() + [fasta, os.path.join(workdir, "r")], capture_output=True, env=env, timeout=timeout)
    want, got = {}, {}
    for suf in SUFFIXES:
        po = os.path.join(workdir, "o." + suf)
        want[suf] = open(po, "rb").read() if os.path.exists(po) else b""
        pr = os.path.join(workdir, "r." + suf + (".gz" if suf.endswith("fastq") else ""))
        if not os.path.exists(pr): got[suf] = b""
        elif pr.endswith(".gz"):
            try: got[suf] = gzip.open(pr, "rb").read()
            except EOFError: got[suf] = b"<truncated gzip>"
        else: got[suf] = open(pr, "rb").read()
    served = avail = -1
    if os.path.exists(rep):
        served, avail = (int(x) for x in open(rep).read().split())
    return o.returncode, r.returncode, want, got, served, avail


def random_replay_flags(rng: random.Random):
    """A random single-end option set without quality normals (a subset of tests/fuzz_flags.py's surface)."""
    f = [f"-z {rng.randrange(1, 10000)}"]
    model = rng.choice(["illumina"] * 3 + ["solid", "ion", "ion"])
    if model == "ion":
        l1 = rng.choice([40, 100, 150, 251, 400])           # (long enough to visit every flow: the reference's flow mask persists from read to read, mode B's is per read)
        f += ["-c 2", f"-f {rng.choice(['TACG', FLOW, 'GATC', 'TACGTACGTCTGAGCATCGATCGATGTACAGC'])}", f"-e {rng.choice(['0', '0.001', '0.01', '0.05', '0.2'])}"]
    else:
        l1 = rng.choice([1, 2, 7, 8, 9, 16, 33, 36, 50, 100, 150, 250, 251])
        if model == "solid": f.append("-c 1")
        if rng.random() < 0.6: f.append(f"-e {rng.choice(['0', '0.001', '0.02', '0.0-0.1', '0.3', '0.05-0.001'])}")
    if model != "ion" and l1 % 2 == 0 and rng.random() < 0.6:      # even length: quality noise replays exactly (module docstring)
        f += [f"-1 {l1}", "-2 0"] + ([f"-Q {rng.choice([0.5, 2, 3.7, 10, 60])}"] if rng.random() < 0.8 else [])      # (no -Q: the default, 2)
    else:
        f += [f"-1 {l1}", "-2 0", rng.choice(["-Q 0", f"-q {rng.choice(['5', 'I', '!'])}"])]
    f.append(rng.choice([f"-N {rng.choice([1, 2, 63, 64, 65, 257, 1000, 3000])}", f"-C {rng.choice([0.5, 2, 7])}"]))
    if rng.random() < 0.6: f.append(f"-r {rng.choice([0, 0.0001, 0.001, 0.01, 0.05, 0.3])}")
    if rng.random() < 0.5: f.append(f"-R {rng.choice([0, 0.1, 0.5, 1.0])}")
    if rng.random() < 0.4: f.append(f"-X {rng.choice([0, 0.3, 0.8, 0.95])}")
    if rng.random() < 0.3: f.append(f"-I {rng.choice([1, 2, 10, 40])}")
    if rng.random() < 0.4: f.append(f"-F {rng.choice([0, 0.3, 0.5, 1.0])}")
    if rng.random() < 0.4: f.append(f"-y {rng.choice([0, 0.01, 0.3, 1.0])}")
    if rng.random() < 0.5: f.append(f"-n {rng.choice([0, 1, 3, 20, 1000])}")
    if rng.random() < 0.3: f.append(f"-A {rng.choice([0, 1, 2])}")
    if rng.random() < 0.2: f.append("-H")
    if rng.random() < 0.3: f.append(f"-o {rng.choice([0, 1, 2])}")
    if rng.random() < 0.2: f.append(f"-P {rng.choice(['p', 'lib_1'])}")
    if rng.random() < 0.15: f.append(rng.choice([f"-m {IN_DIR}/muts_generated.txt", f"-m {IN_DIR}/muts_edge.txt", f"-v {IN_DIR}/muts_edge.vcf", f"-b {IN_DIR}/muts_edge.bed"]))
    if rng.random() < 0.15: f.append(rng.choice([f"-x {IN_DIR}/regions_a.bed", f"-x {IN_DIR}/regions_b.bed"]))
    fasta = "tiny.fa" if any(x.startswith(("-m ", "-v ", "-b ", "-x ")) for x in f) else rng.choice(["tiny.fa", "odd.fa", "ex1.fa"])
    return fasta, " ".join(f)
